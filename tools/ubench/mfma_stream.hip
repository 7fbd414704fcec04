// Micro-benchmark: what do LDS / VMEM instructions cost when issued by the SAME wave inside a stream of fp32
// MFMAs (v_mfma_f32_16x16x4_f32)?  One 256-thread workgroup per CU (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NR, int NW, int NG, int NBAR>
__global__ __launch_bounds__(256, 1) void k(int iters, const float4 *__restrict__ gsrc, float *out) {
    __shared__ float4 lds[4096];
    f32x4 acc[9];
    for (int m = 0; m < 9; ++m) acc[m] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.f;
    float4 r[10], gq[7];
    for (int i = 0; i < 10; ++i) r[i] = make_float4(a, a, a, a);
    for (int i = 0; i < 7; ++i) gq[i] = make_float4(b, b, b, b);
    lds[threadIdx.x] = r[0];
    __syncthreads();
    const float4 *lp = &lds[threadIdx.x & 63];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NW; ++i) lds[threadIdx.x + 256 * i] = gq[i];
#pragma unroll
        for (int i = 0; i < NG; ++i) gq[i] = gsrc[(it & 15) * 4096 + threadIdx.x + 256 * i];
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = lp[64 * i + (it & 1) * 1024];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < 9; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[m].x + a, r[9].y, acc[m], 0, 0, 0);
        if (NBAR) __syncthreads();
    }
    float s = 0;
    for (int m = 0; m < 9; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int i = 0; i < 7; ++i) s += gq[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NR, int NW, int NG, int NBAR>
void run(const char *name, int iters, const float4 *g, float *out) {
    hipLaunchKernelGGL((k<NR, NW, NG, NBAR>), dim3(256), dim3(256), 0, 0, iters, g, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NR, NW, NG, NBAR>), dim3(256), dim3(256), 0, 0, iters, g, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    printf("%-34s %8.3f ms  -> %7.1f cycles per 36-MFMA group @2.35GHz (ideal 1152)\n", name, t, t * 1e-3 * 2.35e9 / iters);
}
int main() {
    float4 *g; float *out;
    hipMalloc(&g, 16 * 4096 * 16); hipMemset(g, 0, 16 * 4096 * 16); hipMalloc(&out, 256 * 256 * 4);
    const int it = 20000;
    run<0, 0, 0, 0>("mfma only", it, g, out);
    run<10, 0, 0, 0>("+10 ds_read_b128", it, g, out);
    run<10, 7, 0, 0>("+10 ds_read +7 ds_write", it, g, out);
    run<10, 7, 7, 0>("+10 read +7 write +7 gload", it, g, out);
    run<10, 7, 7, 1>("+10 read +7 write +7 gload +barrier", it, g, out);
    run<0, 0, 0, 1>("mfma + barrier", it, g, out);
    run<0, 0, 7, 0>("mfma +7 gload", it, g, out);
    run<0, 7, 0, 0>("mfma +7 ds_write", it, g, out);
    return 0;
}
