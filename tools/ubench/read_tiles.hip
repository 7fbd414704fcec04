// How fast can a 5 000 x 25 000 fp32 matrix be streamed ONCE when a workgroup owns a tile of R rows x C columns and walks it row by
// row (the access pattern of a one-pass two-direction ranker: per-column state wants few column blocks, per-row state wants few
// row blocks)?  256 threads, one float4 per thread and chunk of 1 024 columns, 8 chunks in flight.  Prints GB/s per (R, C).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/read_tiles tools/ubench/read_tiles.hip && /tmp/read_tiles
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void read_tiles(const float *__restrict__ S, long ldS, long nrows, long ncols, int R, int C, float *out) {
    const long cb = (long)blockIdx.x * C, rb = (long)blockIdx.y * R;
    const int kpr = C / 1024;                                     // chunks per row of the tile
    const long r_end = rb + R < nrows ? rb + R : nrows;
    const long nchunk = (r_end - rb) * kpr;
    float acc = 0.f;
    for (long q = 0; q < nchunk; q += U) {
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            long qq = q + k < nchunk ? q + k : nchunk - 1;
            const long r = rb + qq / kpr, c = cb + (qq % kpr) * 1024 + threadIdx.x * 4;
            v[k] = c + 3 < ncols ? *reinterpret_cast<const float4 *>(S + r * ldS + c) : float4{0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const long Ni = 5000, Nc = 25000;
    float *S, *out;
    hipMalloc(&S, Ni * Nc * 4); hipMalloc(&out, 4);
    hipMemset(S, 0, Ni * Nc * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int Rs[] = {32, 64, 128, 256, 512, 5000}, Cs[] = {1024, 2048, 4096, 8192, 25600};
    printf("GB/s (median of 9), rows per workgroup down, columns per workgroup across; workgroups in brackets\n        ");
    for (int C : Cs) printf("%14d", C);
    printf("\n");
    for (int R : Rs) {
        printf("R=%5d ", R);
        for (int C : Cs) {
            dim3 grid((unsigned)((Nc + C - 1) / C), (unsigned)((Ni + R - 1) / R));
            std::vector<float> ts;
            for (int it = 0; it < 12; ++it) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(read_tiles<8>, grid, dim3(256), 0, 0, S, Nc, Ni, Nc, R, C, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 3) ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            printf("%8.0f[%4u]", Ni * Nc * 4 / ts[ts.size() / 2] / 1e6, grid.x * grid.y);
        }
        printf("\n");
    }
    return 0;
}
