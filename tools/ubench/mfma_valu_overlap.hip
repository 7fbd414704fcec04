// Micro-benchmark: do fp32 MFMA (v_mfma_f32_16x16x4_f32) and fp32 VALU FMAs issued by DIFFERENT
// waves of the same SIMD overlap on gfx950?  Two 256-thread workgroups per CU; even ones run an MFMA
// loop, odd ones a VALU loop (FMA / or transcendental / or LDS traffic).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 1 = fma, 2 = v_exp, 3 = lds reads
__device__ void side_work(int iters, float *out) {
    __shared__ float buf[4096];
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    const float m = 1.0001f, c = 0.5f;
    buf[threadIdx.x] = a0; buf[threadIdx.x + 256] = a1;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 1) {
                a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
                a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
            } else if (MODE == 2) {
                a0 = __expf(a0 * 1e-3f); a1 = __expf(a1 * 1e-3f); a2 = __expf(a2 * 1e-3f); a3 = __expf(a3 * 1e-3f);
                a4 = __expf(a4 * 1e-3f); a5 = __expf(a5 * 1e-3f); a6 = __expf(a6 * 1e-3f); a7 = __expf(a7 * 1e-3f);
            } else {
                a0 += buf[(threadIdx.x + u * 64) & 4095]; a1 += buf[(threadIdx.x + u * 64 + 1) & 4095];
                a2 += buf[(threadIdx.x + u * 64 + 2) & 4095]; a3 += buf[(threadIdx.x + u * 64 + 3) & 4095];
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(int do_mfma, int do_side, int iters, float *out) {
    __shared__ float force_lds[8192];  // 32 KB so that the LDS base tells the two residents apart
    force_lds[threadIdx.x] = 0.f;
    const unsigned lds_alloc = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));  // HW_REG_LDS_ALLOC
    const bool second = (lds_alloc & 0xff) != 0;
    if (threadIdx.x == 0) { atomicAdd(reinterpret_cast<unsigned *>(out + 1024 * 256), second ? 1u : 0u); out[1024 * 256 + 1 + (blockIdx.x & 7)] = __uint_as_float(lds_alloc); }
    if (!second) {
        if (!do_mfma) return;
        f32x4 acc[9];
        for (int m = 0; m < 9; ++m) acc[m] = f32x4{0, 0, 0, 0};
        float a = threadIdx.x * 1e-3f, b = 1.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 9; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
        }
        float s = 0;
        for (int m = 0; m < 9; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        if (!do_side) return;
        side_work<MODE>(iters * SIDE_SCALE, out);
    }
}

template <int MODE>
void run(const char *name, int iters) {
    float *out; hipMalloc(&out, 1024 * 256 * 4 + 64); hipMemset(out, 0, 1024 * 256 * 4 + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float t[3];
    for (int c = 0; c < 3; ++c) {
        int dm = c != 1, ds = c != 0;
        hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, dm, ds, iters, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, dm, ds, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&t[c], e0, e1);
    }
    unsigned h[9]; hipMemcpy(h, out + 1024 * 256, 36, hipMemcpyDeviceToHost);
    printf("second-slot blocks counted: %u (of 6 launches x 512)  sample LDS_ALLOC regs: %08x %08x %08x %08x\n", h[0], h[1], h[2], h[3], h[4]);
    printf("%-10s mfma only %.3f ms | side only %.3f ms | both %.3f ms  (sum %.3f, max %.3f)\n", name, t[0], t[1], t[2],
           t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
    hipFree(out);
}
int main() {
    run<1>("valu_fma", 20000);
    run<2>("v_exp", 20000);
    run<3>("lds_read", 20000);
    return 0;
}
