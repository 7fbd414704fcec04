// VSRN region-relationship reasoning (itr/modalmodule/vsrn_.py:50-71, Rs_GCN.forward), the part between the 1x1
// convolutions: per image   R = theta_v phi_v^T / N   (N x N affinity of the N = 36 regions),   y = R g_v.
// The three convolutions in front (theta, phi, g: ONE MFMA GEMM with the stacked [3D, D] weight -> rows [theta|phi|g])
// and the W convolution + BatchNorm + residual behind (a GEMM accumulating into a copy of v, BN folded into W) are
// gemm_nt_kernel launches; this kernel is the small per-image piece in between: 2 * 2 * N * N * D flops per image
// (10.6 MFLOP at D = 2048, against 1.5 GFLOP in the layer's GEMMs) and one read of the [N, 3D] block -- HBM-bound.
//
// One workgroup per image.  Phase 1: theta / phi stream through LDS in 64-column chunks, 144 threads hold 3 x 3
// register tiles of R.  Phase 2: R (scaled) sits in LDS, thread = output column, 36 accumulators, g read coalesced.
#include "itr_common.h"

namespace itr {

constexpr int GCN_MAXN = 36;   // regions per image (precomp features: 36)
constexpr int GCN_KC = 64;     // columns of theta / phi per LDS chunk
constexpr int GCN_LD = GCN_KC + 4;

__global__ __launch_bounds__(256) void gcn_relation_kernel(const float *__restrict__ tpg, int64_t ld, float *__restrict__ y,
                                                           int64_t ldy, int N, int D) {
    __shared__ __attribute__((aligned(16))) float th[GCN_MAXN][GCN_LD];
    __shared__ __attribute__((aligned(16))) float ph[GCN_MAXN][GCN_LD];
    __shared__ __attribute__((aligned(16))) float Rs[GCN_MAXN][GCN_MAXN];
    const int t = threadIdx.x;
    const float *base = tpg + (int64_t)blockIdx.x * N * ld;
    const int tn = t / 12, tm = t % 12;       // 12 x 12 threads own 3 x 3 tiles of R (t < 144)
    float acc[3][3] = {};
    for (int k0 = 0; k0 < D; k0 += GCN_KC) {
        // stage theta[:, k0:k0+64] and phi[:, k0:k0+64]; rows >= N are zero
        for (int i = t; i < 2 * GCN_MAXN * (GCN_KC / 4); i += 256) {
            const int which = i / (GCN_MAXN * (GCN_KC / 4));
            const int r = (i / (GCN_KC / 4)) % GCN_MAXN, c = (i % (GCN_KC / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < N) {
                const float *src = base + (int64_t)r * ld + (which ? D : 0) + k0 + c;
                if (k0 + c + 3 < D) v = *reinterpret_cast<const float4 *>(src);
                else {
                    if (k0 + c < D) v.x = src[0];
                    if (k0 + c + 1 < D) v.y = src[1];
                    if (k0 + c + 2 < D) v.z = src[2];
                }
            }
            *reinterpret_cast<float4 *>(which ? &ph[r][c] : &th[r][c]) = v;
        }
        __syncthreads();
        if (t < 144) {
#pragma unroll 4
            for (int k = 0; k < GCN_KC; k += 4) {
                float4 a[3], b[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    a[i] = *reinterpret_cast<const float4 *>(&th[3 * tn + i][k]);
                    b[i] = *reinterpret_cast<const float4 *>(&ph[3 * tm + i][k]);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        acc[i][j] = fmaf(a[i].x, b[j].x, acc[i][j]);
                        acc[i][j] = fmaf(a[i].y, b[j].y, acc[i][j]);
                        acc[i][j] = fmaf(a[i].z, b[j].z, acc[i][j]);
                        acc[i][j] = fmaf(a[i].w, b[j].w, acc[i][j]);
                    }
            }
        }
        __syncthreads();
    }
    if (t < 144) {
        const float inv = 1.f / (float)N;     // R_div_C = R / N  (vsrn_.py:62-63)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Rs[3 * tn + i][3 * tm + j] = acc[i][j] * inv;
    }
    __syncthreads();
    // phase 2: y[n, d] = sum_m Rs[n][m] g[m, d]
    float *yo = y + (int64_t)blockIdx.x * N * ldy;
    for (int d = t; d < D; d += 256) {
        float g[GCN_MAXN];
#pragma unroll
        for (int m = 0; m < GCN_MAXN; ++m) g[m] = m < N ? base[(int64_t)m * ld + 2 * (int64_t)D + d] : 0.f;
#pragma unroll 4
        for (int n = 0; n < GCN_MAXN; ++n) {
            if (n >= N) break;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < GCN_MAXN; m += 4) {
                const float4 r = *reinterpret_cast<const float4 *>(&Rs[n][m]);
                s = fmaf(r.x, g[m], s);
                s = fmaf(r.y, g[m + 1], s);
                s = fmaf(r.z, g[m + 2], s);
                s = fmaf(r.w, g[m + 3], s);
            }
            yo[(int64_t)n * ldy + d] = s;
        }
    }
}

}  // namespace itr

extern "C" int itr_gcn_relation(const float *tpg, int64_t ld, float *y, int64_t ldy, int64_t n_img, int N, int D,
                                itr_stream_t stream) {
    ITR_REQUIRE(n_img >= 0 && N >= 1 && N <= itr::GCN_MAXN && D >= 1, "itr_gcn_relation: bad shape (at most 36 regions)");
    if (n_img == 0) return ITR_OK;
    ITR_REQUIRE(tpg && y, "itr_gcn_relation: null pointer");
    ITR_REQUIRE(ld >= 3 * (int64_t)D && ldy >= D && ld % 4 == 0 && D % 4 == 0, "itr_gcn_relation: strides (ld >= 3 D, multiples of 4)");
    ITR_REQUIRE((reinterpret_cast<uintptr_t>(tpg) & 15) == 0, "itr_gcn_relation: unaligned");
    ITR_REQUIRE(n_img <= 0x7fffffff, "itr_gcn_relation: too many images");
    hipLaunchKernelGGL(itr::gcn_relation_kernel, dim3((unsigned)n_img), dim3(256), 0, itr::as_stream(stream), tpg, ld, y, ldy, N, D);
    ITR_CHECK_LAUNCH("gcn_relation");
    return ITR_OK;
}
