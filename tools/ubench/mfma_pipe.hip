// Micro-benchmark of the SCAN main-loop STRUCTURE (one 256-thread workgroup per CU):
// per "half chunk": 36 fp32 MFMAs consuming fragments read from LDS one half earlier, 10 ds_read_b128 for the
// next half; every other half: 7 ds_write_b128 + 7 global loads (prefetch distance 2 chunks) + a barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void k(int iters, const char *__restrict__ gsrc, float *out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f32x4 acc[9];
    for (int m = 0; m < 9; ++m) acc[m] = f32x4{0, 0, 0, 0};
    const int tid = threadIdx.x;
    f32x4 fa[9], fb, ga[9], gb;  // two fragment sets
    f32x4 s0[7], s1[7];
    for (int i = 0; i < 7; ++i) { s0[i] = f32x4{1, 1, 1, 1}; s1[i] = f32x4{2, 2, 2, 2}; }
    for (int i = tid; i < 2 * 28672 / 16; i += 256) reinterpret_cast<f32x4 *>(lds)[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    const unsigned ra = (tid & 63) * 16, wa = tid * 16;
    const unsigned vo = tid * 16;
#define FREAD(A, B, OFF) { B = *reinterpret_cast<const f32x4 *>(lds + ra + (OFF) + 2304); \
    _Pragma("unroll") for (int m = 0; m < 9; ++m) A[m] = *reinterpret_cast<const f32x4 *>(lds + ra + (OFF) + m * 256); }
#define FMFMA(A, B) { _Pragma("unroll") for (int c = 0; c < 4; ++c) _Pragma("unroll") for (int m = 0; m < 9; ++m) \
    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m][c], B[c], acc[m], 0, 0, 0); }
#define LSTORE(S, OFF) { _Pragma("unroll") for (int i = 0; i < 7; ++i) *reinterpret_cast<f32x4 *>(lds + wa + (OFF) + i * 4096) = S[i]; }
#define GLOAD(S, KC) { const char *b_ = gsrc + (size_t)((KC) & 31) * 28672; \
    _Pragma("unroll") for (int i = 0; i < 7; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(S[i]) : "v"(vo + i * 4096), "s"(b_) : "memory"); }
#define VMWAIT(S) asm volatile("s_waitcnt vmcnt(7)" : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(S[4]), "+v"(S[5]), "+v"(S[6])::"memory");
#define PIN __builtin_amdgcn_sched_barrier(0);
#define SGB(m, n) __builtin_amdgcn_sched_group_barrier(m, n, 0);
#define ILV_A { _Pragma("unroll") for (int i_ = 0; i_ < 7; ++i_) { SGB(0x008, 1) SGB(0x200, 1) } \
    _Pragma("unroll") for (int i_ = 0; i_ < 7; ++i_) { SGB(0x008, 1) SGB(0x020, 1) } \
    _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) { SGB(0x008, 1) SGB(0x100, 1) } SGB(0x008, 12) }
#define ILV_B { _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) { SGB(0x008, 1) SGB(0x100, 1) } SGB(0x008, 26) }
#define BARRIER { if (VARIANT == 3 || VARIANT == 4 || VARIANT == 10) __syncthreads(); \
    if (VARIANT == 5) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } \
    if (VARIANT == 6) { asm volatile("s_barrier" ::: "memory"); } \
    if (VARIANT == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } \
    if (VARIANT == 8) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } }
    GLOAD(s0, 0) GLOAD(s1, 1)
    FREAD(fa, fb, 0)
    for (int it = 0; it < iters; it += 2) {
        // chunk A (buffer 0 -> park into buffer 1)
        if (VARIANT >= 2) { VMWAIT(s1) LSTORE(s1, 28672) GLOAD(s1, it + 3) }
        if (VARIANT >= 1) FREAD(ga, gb, 1024)
        if (VARIANT == 4) PIN
        FMFMA(fa, fb)
        if (VARIANT >= 9) { ILV_A }
        if (VARIANT >= 4) PIN
        BARRIER
        if (VARIANT >= 1) FREAD(fa, fb, 28672)
        if (VARIANT == 4) PIN
        FMFMA(ga, gb)
        if (VARIANT >= 9) { ILV_B }
        if (VARIANT == 4) PIN
        // chunk B (buffer 1 -> park into buffer 0)
        if (VARIANT >= 2) { VMWAIT(s0) LSTORE(s0, 0) GLOAD(s0, it + 4) }
        if (VARIANT >= 1) FREAD(ga, gb, 28672 + 1024)
        if (VARIANT == 4) PIN
        FMFMA(fa, fb)
        if (VARIANT >= 9) { ILV_A }
        if (VARIANT == 4) PIN
        BARRIER
        if (VARIANT >= 1) FREAD(fa, fb, 0)
        if (VARIANT == 4) PIN
        FMFMA(ga, gb)
        if (VARIANT >= 9) { ILV_B }
        if (VARIANT == 4) PIN
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int m = 0; m < 9; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int i = 0; i < 7; ++i) s += s0[i][0] + s1[i][0];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(const char *name, int iters, const char *g, float *out) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 100 * 1024, 0, iters, g, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (rep) printf("%-44s %8.3f ms -> %7.1f cycles per chunk of 72 MFMAs @2.35GHz (ideal 2304)\n", name, t, t * 1e-3 * 2.35e9 / iters);
    }
}
int main() {
    char *g; float *out;
    hipMalloc(&g, 64 * 28672 + 65536); hipMemset(g, 0, 64 * 28672 + 65536); hipMalloc(&out, 256 * 256 * 4);
    const int it = 20000;
    run<0>("mfma only", it, g, out);
    run<1>("+ pipelined fragment reads", it, g, out);
    run<2>("+ park (7 ds_write) + 7 gloads per chunk", it, g, out);
    run<3>("+ barrier per chunk", it, g, out);
    run<4>("+ sched_barrier pins", it, g, out);
    run<5>("asm lgkmcnt(0)+s_barrier", it, g, out);
    run<6>("asm s_barrier only (racy)", it, g, out);
    run<7>("asm lgkmcnt(0) only (no barrier)", it, g, out);
    run<8>("fence+s_barrier builtin", it, g, out);
    run<9>("interleaved 1 mem op per MFMA, no barrier", it, g, out);
    run<10>("interleaved + __syncthreads", it, g, out);
    return 0;
}
