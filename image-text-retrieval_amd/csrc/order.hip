// The two non-cosine pooled measures of the reference (measure='order'):
//   itr_order_scores   order_sim (Objectives.py:24-30):   S[i, c] = -sqrt( sum_d max(0, s[c, d] - im[i, d])^2 )
//   itr_order_bwd      its gradient for train_emb
//   itr_pdist_finish   SAEM's pdist (Objectives.py:297-307): S = sqrt(|x1|^2 - 2 x1.x2 + |x2|^2 + 1e-4) on top of the
//                      x1 x2^T GEMM (gemm_nt_kernel) -- the squared row norms come from itr_row_sqnorm
// order_sim is NOT a contraction (the clamp sits inside the sum): 3 VALU operations per (pair, dimension), 64 x 64 pair
// tiles with 4 x 4 register blocks, both operands staged k-major in LDS.  Bound: VALU (3 * Ni * Nc * D lane-ops;
// 5k x 25k x 1024 -> 384 G lane-ops ~ 10 ms at 256 CUs x 64 lanes x 2.3 GHz), not HBM (inputs are re-read from L2).
#include "itr_common.h"

namespace itr {

constexpr int OT = 64;        // pairs tile: 64 images x 64 captions
constexpr int OK_ = 32;       // dimensions per LDS chunk
constexpr int OLD = OT + 1;

__device__ __forceinline__ void order_stage(const float *__restrict__ src, int64_t n_rows, int64_t row0, int D, int k0,
                                            float (*dst)[OLD], int t) {
    // lane -> (8 consecutive float4 of one row = 128 contiguous bytes); 32 rows per pass
    const int kq = t & 7, r = t >> 3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = r + 32 * p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < n_rows) {
            const float *q = src + (row0 + row) * D + k0 + 4 * kq;
            if (k0 + 4 * kq + 3 < D) v = *reinterpret_cast<const float4 *>(q);
            else {
                if (k0 + 4 * kq < D) v.x = q[0];
                if (k0 + 4 * kq + 1 < D) v.y = q[1];
                if (k0 + 4 * kq + 2 < D) v.z = q[2];
            }
        }
        dst[4 * kq][row] = v.x;
        dst[4 * kq + 1][row] = v.y;
        dst[4 * kq + 2][row] = v.z;
        dst[4 * kq + 3][row] = v.w;
    }
}

__global__ __launch_bounds__(256) void order_scores_kernel(const float *__restrict__ im, const float *__restrict__ s,
                                                           float *__restrict__ out, int64_t Ni, int64_t Nc, int D, int64_t ldo) {
    __shared__ float As[OK_][OLD];
    __shared__ float Bs[OK_][OLD];
    const int t = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;        // 16 x 16 threads, 4 x 4 pairs each: images 4ty.., captions 4tx..
    const int64_t i0 = (int64_t)blockIdx.y * OT, c0 = (int64_t)blockIdx.x * OT;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < D; k0 += OK_) {
        order_stage(im, Ni, i0, D, k0, As, t);
        order_stage(s, Nc, c0, D, k0, Bs, t);
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < OK_; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = As[k][4 * ty + j];
                b[j] = Bs[k][4 * tx + j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float r = fmaxf(b[j] - a[i], 0.f);
                    acc[i][j] = fmaf(r, r, acc[i][j]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = i0 + 4 * ty + i;
        if (row >= Ni) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t col = c0 + 4 * tx + j;
            if (col < Nc) out[row * ldo + col] = -sqrtf(acc[i][j]);
        }
    }
}

// dX[r, d] = sign * sum_q W(r, q) * max(0, dir * (Y[q, d] - X[r, d])),   W(r, q) = w[r * ws_r + q * ws_q]
//   d im:  X = im, Y = s,  dir = +1, sign = +1, W(i, c) = dS[i, c] / n[i, c]
//   d s :  X = s,  Y = im, dir = -1, sign = -1, W(c, i) = dS[i, c] / n[i, c]          (n = -S; pairs with n == 0 carry no gradient)
__global__ __launch_bounds__(256) void order_bwd_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                        const float *__restrict__ dS, const float *__restrict__ S, int64_t ws_r,
                                                        int64_t ws_q, int64_t nX, int64_t nY, int D, float dir, float sign,
                                                        float *__restrict__ dX) {
    const int64_t r = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const float x = X[r * D + d];
    float acc = 0.f;
    for (int64_t q = 0; q < nY; ++q) {
        const float n = -S[r * ws_r + q * ws_q];
        const float w = n > 0.f ? dS[r * ws_r + q * ws_q] / n : 0.f;
        acc = fmaf(w, fmaxf(dir * (Y[q * D + d] - x), 0.f), acc);
    }
    dX[r * D + d] = sign * acc;
}

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float *__restrict__ x, float *__restrict__ out, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float sacc = 0.f;
    for (int c = lane; c < D; c += 64) sacc = fmaf(x[row * D + c], x[row * D + c], sacc);
    sacc = wave_sum(sacc);
    if (lane == 0) out[row] = sacc;
}

__global__ __launch_bounds__(256) void pdist_finish_kernel(float *__restrict__ S, const float *__restrict__ n1,
                                                           const float *__restrict__ n2, int64_t Ni, int64_t Nc) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= Ni * Nc) return;
    const int64_t i = idx / Nc, c = idx - i * Nc;
    S[idx] = sqrtf(n1[i] - 2.f * S[idx] + n2[c] + 1e-4f);      // evaluation order of Objectives.py:307
}

}  // namespace itr

extern "C" int itr_order_scores(const float *im, const float *s, float *out, int64_t Ni, int64_t Nc, int D, itr_stream_t stream) {
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && D >= 1, "itr_order_scores: bad shape");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    ITR_REQUIRE(im && s && out, "itr_order_scores: null pointer");
    ITR_REQUIRE(D % 4 == 0 && ((reinterpret_cast<uintptr_t>(im) | reinterpret_cast<uintptr_t>(s)) & 15) == 0,
                "itr_order_scores: D must be a multiple of 4 and the operands 16-byte aligned");
    ITR_REQUIRE(itr::ceil_div(Ni, (int64_t)itr::OT) <= 65535, "itr_order_scores: too many image rows per call");
    dim3 grid((unsigned)itr::ceil_div(Nc, (int64_t)itr::OT), (unsigned)itr::ceil_div(Ni, (int64_t)itr::OT));
    hipLaunchKernelGGL(itr::order_scores_kernel, grid, dim3(256), 0, itr::as_stream(stream), im, s, out, Ni, Nc, D, Nc);
    ITR_CHECK_LAUNCH("order_scores");
    return ITR_OK;
}

extern "C" int itr_order_bwd(const float *im, const float *s, const float *S, const float *dS, float *d_im, float *d_s, int64_t Ni,
                             int64_t Nc, int D, itr_stream_t stream) {
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && D >= 1, "itr_order_bwd: bad shape");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    ITR_REQUIRE(im && s && S && dS && d_im && d_s, "itr_order_bwd: null pointer");
    ITR_REQUIRE(Ni <= 65535 && Nc <= 65535, "itr_order_bwd: a training batch (at most 65535 rows)");
    const unsigned gx = (unsigned)itr::ceil_div(D, 256);
    hipLaunchKernelGGL(itr::order_bwd_kernel, dim3(gx, (unsigned)Ni), dim3(256), 0, itr::as_stream(stream), im, s, dS, S, Nc,
                       (int64_t)1, Ni, Nc, D, 1.f, 1.f, d_im);
    hipLaunchKernelGGL(itr::order_bwd_kernel, dim3(gx, (unsigned)Nc), dim3(256), 0, itr::as_stream(stream), s, im, dS, S,
                       (int64_t)1, Nc, Nc, Ni, D, -1.f, -1.f, d_s);
    ITR_CHECK_LAUNCH("order_bwd");
    return ITR_OK;
}

extern "C" int itr_row_sqnorm(const float *x, float *out, int64_t rows, int D, itr_stream_t stream) {
    ITR_REQUIRE(rows >= 0 && D >= 1, "itr_row_sqnorm: bad shape");
    if (rows == 0) return ITR_OK;
    ITR_REQUIRE(x && out, "itr_row_sqnorm: null pointer");
    hipLaunchKernelGGL(itr::row_sqnorm_kernel, dim3((unsigned)itr::ceil_div(rows, (int64_t)4)), dim3(256), 0, itr::as_stream(stream), x,
                       out, rows, D);
    ITR_CHECK_LAUNCH("row_sqnorm");
    return ITR_OK;
}

extern "C" int itr_pdist_finish(float *S, const float *n1, const float *n2, int64_t Ni, int64_t Nc, itr_stream_t stream) {
    ITR_REQUIRE(Ni >= 0 && Nc >= 0, "itr_pdist_finish: bad shape");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    ITR_REQUIRE(S && n1 && n2, "itr_pdist_finish: null pointer");
    ITR_REQUIRE(itr::ceil_div(Ni * Nc, (int64_t)256) <= 0x7fffffff, "itr_pdist_finish: matrix too large for one call");
    hipLaunchKernelGGL(itr::pdist_finish_kernel, dim3((unsigned)itr::ceil_div(Ni * Nc, (int64_t)256)), dim3(256), 0,
                       itr::as_stream(stream), S, n1, n2, Ni, Nc);
    ITR_CHECK_LAUNCH("pdist_finish");
    return ITR_OK;
}
