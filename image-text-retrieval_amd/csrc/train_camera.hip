// Training-step primitives of the CAMERA towers (itr/modalmodule/camera_.py:14-114, ImgEncoder.py:355-389, TextEncoder.py:162-192,
// Fusionmodule.py:674-692) -- the pieces between the dense layers (which are gemm_nt_kernel, autograd.py: _Linear):
//   itr_ew_mul                       out = a * b                                   (rgn_emb * pos_emb, fc_q(q) * fc_k(k); backward = two more calls)
//   itr_act_bwd                      dx = dy * act'(.) from the OUTPUT y of sigmoid / relu / tanh
//   itr_gate_apply / _bwd            q' = q * M[:, :dk], k' = k * M[:, dk:]       (GatedQueryAttLayer, camera_.py:41-44)
//   itr_bn_train_fwd / _bwd          nn.BatchNorm1d in training mode on [N, C]: batch statistics per column (biased variance)
//   itr_l2norm_mid_fwd / _bwd        utils.l2norm with its DEFAULT dim=1 on [B, R, D]: normalises across the R regions (ImgEncoder.py:378,384)
//   itr_smry_fwd / _bwd              L = softmax(smry_mat, dim=1);  out[b, v, :] = sum_r L[b, r, v] x[b, r, :]   (ImgEncoder.py:386-387)
//   itr_groupmax_fwd / _bwd          MultiViewMatching: S[i, c] = max_v T[i * k + v, c] with the arg-max view kept
// All HBM-bound elementwise / short-reduction kernels.
#include "itr_common.h"

namespace itr {

__global__ __launch_bounds__(256) void ew_mul_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out,
                                                     int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy, float *__restrict__ dx,
                                                      int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = y[i];
    float d;
    if (act == 1) d = v > 0.f ? 1.f : 0.f;            // relu
    else if (act == 2) d = 1.f - v * v;               // tanh
    else if (act == 4) d = v > 0.f ? 1.f : 0.1f;      // LeakyReLU(0.1)
    else d = v * (1.f - v);                           // sigmoid
    dx[i] = dy[i] * d;
}

// rows x dk operands, M rows x 2 dk
__global__ __launch_bounds__(256) void gate_apply_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ M,
                                                         float *__restrict__ qo, float *__restrict__ ko, int64_t rows, int dk) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * dk) return;
    const int64_t r = i / dk;
    const int d = (int)(i - r * dk);
    qo[i] = q[i] * M[r * 2 * dk + d];
    ko[i] = k[i] * M[r * 2 * dk + dk + d];
}
__global__ __launch_bounds__(256) void gate_apply_bwd_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                             const float *__restrict__ M, const float *__restrict__ dqo,
                                                             const float *__restrict__ dko, float *__restrict__ dq, float *__restrict__ dk_,
                                                             float *__restrict__ dM, int64_t rows, int dk) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * dk) return;
    const int64_t r = i / dk;
    const int d = (int)(i - r * dk);
    const float mq = M[r * 2 * dk + d], mk = M[r * 2 * dk + dk + d];
    dq[i] = dqo[i] * mq;
    dk_[i] = dko[i] * mk;
    dM[r * 2 * dk + d] = dqo[i] * q[i];
    dM[r * 2 * dk + dk + d] = dko[i] * k[i];
}

// ---- BatchNorm1d, training mode.  Grid = (column blocks of 64, row slices): with few columns and many rows (VisualSA's
// BatchNorm1d(36) over B * D = 131 072 rows) one workgroup per column block would leave the chip idle, so the column statistics are
// reduced in two stages: every (column block, row slice) workgroup writes partial sums, the consumers add the <= 64 partials of
// their columns in a fixed order (deterministic).  Two passes for the variance (mean first), like torch.
constexpr int BN_MAXSLICES = 64;
__device__ __forceinline__ void bn_rows(int64_t N, int slices, int64_t *r0, int64_t *r1) {
    const int64_t per = (N + slices - 1) / slices;
    *r0 = (int64_t)blockIdx.y * per;
    *r1 = *r0 + per < N ? *r0 + per : N;
}
// mode 0: part[slice][c] = sum x;  mode 1: sum (x - mean)^2 with mean from the mode-0 partials (sum0 / N)
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, const float *__restrict__ sum0, float *__restrict__ part,
                                                         int64_t N, int C, int slices, int mode) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    int64_t r0, r1;
    bn_rows(N, slices, &r0, &r1);
    float u = 0.f;
    if (mode == 1 && ok) {
        for (int s = 0; s < slices; ++s) u += sum0[(int64_t)s * C + c];
        u /= (float)N;
    }
    // (8 rows in flight per lane in every row loop of this file: one load per loop trip made each of them a chain of L2 round trips --
    // bn_bwd_partial 111 us for 38 MB in round 6)
    float a = 0.f;
    if (ok) {
        int64_t r = r0 + rl;
        float a1 = 0.f;
        for (; r + 28 < r1; r += 32) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = x[(r + 4 * q) * C + c] - u;
#pragma unroll
            for (int q = 0; q < 8; q += 2) { a += mode == 0 ? v[q] : v[q] * v[q]; a1 += mode == 0 ? v[q + 1] : v[q + 1] * v[q + 1]; }
        }
        for (; r < r1; r += 4) {
            const float d = x[r * C + c] - u;
            a += mode == 0 ? d : d * d;
        }
        a += a1;
    }
    red[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && ok) part[(int64_t)blockIdx.y * C + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}
__global__ __launch_bounds__(256) void bn_train_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, const float *__restrict__ sum0,
                                                             const float *__restrict__ sum1, float *__restrict__ y, float *__restrict__ mean,
                                                             float *__restrict__ invstd, int64_t N, int C, int slices, float eps) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (c >= C) return;
    float u = 0.f, q = 0.f;
    for (int s = 0; s < slices; ++s) { u += sum0[(int64_t)s * C + c]; q += sum1[(int64_t)s * C + c]; }
    u /= (float)N;
    const float is = 1.f / sqrtf(q / (float)N + eps);      // biased variance, as the normalisation uses
    int64_t r0, r1;
    bn_rows(N, slices, &r0, &r1);
    const float g = gamma[c], b = beta[c];
    int64_t r = r0 + rl;
    for (; r + 28 < r1; r += 32) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[(r + 4 * q) * C + c];
#pragma unroll
        for (int q = 0; q < 8; ++q) y[(r + 4 * q) * C + c] = (v[q] - u) * is * g + b;
    }
    for (; r < r1; r += 4) y[r * C + c] = (x[r * C + c] - u) * is * g + b;
    if (blockIdx.y == 0 && rl == 0) { mean[c] = u; invstd[c] = is; }
}
// backward partials: pa = sum dy, pb = sum dy * xhat per (row slice, column)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                             const float *__restrict__ mean, const float *__restrict__ invstd,
                                                             float *__restrict__ pa, float *__restrict__ pb, int64_t N, int C, int slices) {
    __shared__ float ra[4][64], rb[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    const float u = ok ? mean[c] : 0.f, is = ok ? invstd[c] : 0.f;
    int64_t r0, r1;
    bn_rows(N, slices, &r0, &r1);
    float a = 0.f, b = 0.f;
    if (ok) {
        int64_t r = r0 + rl;
        for (; r + 28 < r1; r += 32) {
            float g[8], xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { g[q] = dy[(r + 4 * q) * C + c]; xv[q] = x[(r + 4 * q) * C + c]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) { a += g[q]; b += g[q] * (xv[q] - u) * is; }
        }
        for (; r < r1; r += 4) {
            const float g = dy[r * C + c];
            a += g;
            b += g * (x[r * C + c] - u) * is;
        }
    }
    ra[rl][cl] = a; rb[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && ok) {
        pa[(int64_t)blockIdx.y * C + c] = ra[0][cl] + ra[1][cl] + ra[2][cl] + ra[3][cl];
        pb[(int64_t)blockIdx.y * C + c] = rb[0][cl] + rb[1][cl] + rb[2][cl] + rb[3][cl];
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma, const float *__restrict__ pa,
                                                           const float *__restrict__ pb, float *__restrict__ dx, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta, int64_t N, int C, int slices) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (c >= C) return;
    float sa = 0.f, sb = 0.f;
    for (int s = 0; s < slices; ++s) { sa += pa[(int64_t)s * C + c]; sb += pb[(int64_t)s * C + c]; }
    const float u = mean[c], is = invstd[c], g = gamma[c], ma = sa / (float)N, mb = sb / (float)N;
    int64_t r0, r1;
    bn_rows(N, slices, &r0, &r1);
    int64_t r = r0 + rl;
    for (; r + 28 < r1; r += 32) {
        float gv[8], xv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { gv[q] = dy[(r + 4 * q) * C + c]; xv[q] = x[(r + 4 * q) * C + c]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) dx[(r + 4 * q) * C + c] = g * is * (gv[q] - ma - (xv[q] - u) * is * mb);
    }
    for (; r < r1; r += 4) {
        const float xh = (x[r * C + c] - u) * is;
        dx[r * C + c] = g * is * (dy[r * C + c] - ma - xh * mb);
    }
    if (blockIdx.y == 0 && rl == 0) { dgamma[c] = sb; dbeta[c] = sa; }
}
static int bn_slices(int64_t N, int C) {
    const int64_t col_blocks = ceil_div((int64_t)C, (int64_t)64);
    int64_t s = ceil_div((int64_t)1024, col_blocks);            // aim at ~1024 workgroups
    const int64_t by_rows = ceil_div(N, (int64_t)256);          // at least 256 rows per slice
    if (s > by_rows) s = by_rows;
    if (s > BN_MAXSLICES) s = BN_MAXSLICES;
    return (int)(s < 1 ? 1 : s);
}

// ---- l2norm across the middle axis of [B, R, D]: one thread per (b, d)
__global__ __launch_bounds__(256) void l2norm_mid_fwd_kernel(const float *__restrict__ x, float *__restrict__ z, float *__restrict__ nrm,
                                                             int R, int D, float eps) {
    const int64_t b = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const float *p = x + b * R * D + d;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += p[(int64_t)r * D] * p[(int64_t)r * D];
    const float n = sqrtf(s);
    for (int r = 0; r < R; ++r) z[b * R * D + (int64_t)r * D + d] = p[(int64_t)r * D] / (n + eps);
    nrm[b * D + d] = n;
}
__global__ __launch_bounds__(256) void l2norm_mid_bwd_kernel(const float *__restrict__ dz, const float *__restrict__ z,
                                                             const float *__restrict__ nrm, float *__restrict__ dx, int R, int D, float eps) {
    const int64_t b = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const int64_t o = b * R * D + d;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += dz[o + (int64_t)r * D] * z[o + (int64_t)r * D];
    const float n = nrm[b * D + d];
    const float inv = 1.f / (n + eps), kk = n > 0.f ? s / n : 0.f;
    for (int r = 0; r < R; ++r) dx[o + (int64_t)r * D] = dz[o + (int64_t)r * D] * inv - z[o + (int64_t)r * D] * kk;
}

// ---- multi-view summarisation: L = softmax over the R regions of smry [B, R, K]; out[b, v, d] = sum_r L[b, r, v] x[b, r, d]
int allow_dynamic_lds(const void *kernel, size_t bytes);      // scan_train.hip
constexpr int SM_R = 192, SM_K = 192;    // (SGRAF: K = words of a caption / graph nodes; the longest Flickr30k caption has 82 tokens.  The R x K softmax block is
                                         //  dynamic LDS: 36 KB at 96 x 96, 144 KB at the limit)
__global__ __launch_bounds__(256) void smry_fwd_kernel(const float *__restrict__ smry, const float *__restrict__ x, float *__restrict__ Lout,
                                                       float *__restrict__ out, int R, int K, int D) {
    extern __shared__ __attribute__((aligned(16))) float sl[];      // [R][K]
    const int64_t b = blockIdx.y;
    const int t = threadIdx.x;
    if (t < K) {
        float m = -INFINITY;
        for (int r = 0; r < R; ++r) m = fmaxf(m, smry[(b * R + r) * K + t]);
        float den = 0.f;
        for (int r = 0; r < R; ++r) den += expf(smry[(b * R + r) * K + t] - m);
        for (int r = 0; r < R; ++r) {
            const float p = expf(smry[(b * R + r) * K + t] - m) / den;
            sl[r * K + t] = p;
            if (blockIdx.x == 0) Lout[(b * R + r) * K + t] = p;
        }
    }
    __syncthreads();
    const int d = blockIdx.x * 256 + t;
    if (d >= D) return;
    for (int v = 0; v < K; ++v) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s = fmaf(sl[r * K + v], x[(b * R + r) * D + d], s);
        out[(b * K + v) * D + d] = s;
    }
}
// dx[b, r, d] = sum_v L[b, r, v] dout[b, v, d];   dLraw[b, r, v] = sum_d dout[b, v, d] x[b, r, d]  (one workgroup per (b, r), second kernel)
__global__ __launch_bounds__(256) void smry_bwd_x_kernel(const float *__restrict__ L, const float *__restrict__ dout, float *__restrict__ dx,
                                                         int R, int K, int D) {
    const int64_t b = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
        for (int v = 0; v < K; ++v) s = fmaf(L[(b * R + r) * K + v], dout[(b * K + v) * D + d], s);
        dx[(b * R + r) * D + d] = s;
    }
}
__global__ __launch_bounds__(256) void smry_bwd_l_kernel(const float *__restrict__ x, const float *__restrict__ dout, float *__restrict__ dLraw,
                                                         int R, int K, int D) {
    __shared__ float red[4];
    const int64_t br = blockIdx.x;            // b * R + r
    const int64_t b = br / R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int v = 0; v < K; ++v) {
        float s = 0.f;
        for (int d = threadIdx.x; d < D; d += 256) s = fmaf(dout[(b * K + v) * D + d], x[br * D + d], s);
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) dLraw[br * K + v] = red[0] + red[1] + red[2] + red[3];
        __syncthreads();
    }
}
// softmax backward over the R axis: dsmry[b, r, v] = L (dLraw - sum_r dLraw L)
__global__ __launch_bounds__(256) void smry_bwd_softmax_kernel(const float *__restrict__ L, const float *__restrict__ dLraw,
                                                              float *__restrict__ dsmry, int R, int K) {
    const int64_t b = blockIdx.x;
    const int v = threadIdx.x;
    if (v >= K) return;
    float dot = 0.f;
    for (int r = 0; r < R; ++r) dot += dLraw[(b * R + r) * K + v] * L[(b * R + r) * K + v];
    for (int r = 0; r < R; ++r) dsmry[(b * R + r) * K + v] = L[(b * R + r) * K + v] * (dLraw[(b * R + r) * K + v] - dot);
}

// ---- max over the k view rows of each image: T [Ni * k, Nc] -> S [Ni, Nc], arg [Ni, Nc]
__global__ __launch_bounds__(256) void groupmax_fwd_kernel(const float *__restrict__ T, int k, int64_t Nc, float *__restrict__ S,
                                                           int32_t *__restrict__ arg) {
    const int64_t i = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= Nc) return;
    float m = T[(i * k) * Nc + c];
    int am = 0;
    for (int v = 1; v < k; ++v) {
        const float t = T[(i * k + v) * Nc + c];
        if (t > m) { m = t; am = v; }
    }
    S[i * Nc + c] = m;
    arg[i * Nc + c] = am;
}
__global__ __launch_bounds__(256) void groupmax_bwd_kernel(const float *__restrict__ dS, const int32_t *__restrict__ arg, int k, int64_t Nc,
                                                           float *__restrict__ dT) {
    const int64_t i = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= Nc) return;
    const int am = arg[i * Nc + c];
    for (int v = 0; v < k; ++v) dT[(i * k + v) * Nc + c] = v == am ? dS[i * Nc + c] : 0.f;
}

// ---- batched small matrix products (SGRAF graph reasoning: n <= 64 nodes, 256-wide similarity vectors):
//   C[b] (M x N) = op(A[b]) op(B[b]),  op = identity or transpose, all operands row-major and dense per batch item
__global__ __launch_bounds__(256) void bmm_small_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M,
                                                        int N, int K, int ta, int tb) {
    const int64_t b = blockIdx.y;
    const float *a = A + b * (int64_t)M * K, *bb = B + b * (int64_t)K * N;
    float *c = C + b * (int64_t)M * N;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < M * N; e += gridDim.x * 256) {
        const int m = e / N, n = e % N;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(ta ? a[k * M + m] : a[m * K + k], tb ? bb[n * K + k] : bb[k * N + n], s);
        c[e] = s;
    }
}

}  // namespace itr

#define EW_GRID(n) dim3((unsigned)itr::ceil_div((int64_t)(n), (int64_t)256)), dim3(256), 0, itr::as_stream(stream)

extern "C" int itr_ew_mul(const float *a, const float *b, float *out, int64_t n, itr_stream_t stream) {
    ITR_REQUIRE(n >= 0 && itr::ceil_div(n, (int64_t)256) <= 0x7fffffff, "itr_ew_mul: bad size");
    if (n == 0) return ITR_OK;
    ITR_REQUIRE(a && b && out, "itr_ew_mul: null pointer");
    hipLaunchKernelGGL(itr::ew_mul_kernel, EW_GRID(n), a, b, out, n);
    ITR_CHECK_LAUNCH("ew_mul");
    return ITR_OK;
}

extern "C" int itr_act_bwd(const float *y, const float *dy, float *dx, int64_t n, int act, itr_stream_t stream) {
    ITR_REQUIRE(n >= 0 && itr::ceil_div(n, (int64_t)256) <= 0x7fffffff && act >= 1 && act <= 4, "itr_act_bwd: bad size or activation (1 relu, 2 tanh, 3 sigmoid, 4 leaky_relu(0.1))");
    if (n == 0) return ITR_OK;
    ITR_REQUIRE(y && dy && dx, "itr_act_bwd: null pointer");
    hipLaunchKernelGGL(itr::act_bwd_kernel, EW_GRID(n), y, dy, dx, n, act);
    ITR_CHECK_LAUNCH("act_bwd");
    return ITR_OK;
}

extern "C" int itr_gate_apply(const float *q, const float *k, const float *M, float *qo, float *ko, int64_t rows, int dk, itr_stream_t stream) {
    ITR_REQUIRE(rows >= 0 && dk >= 1 && itr::ceil_div(rows * dk, (int64_t)256) <= 0x7fffffff, "itr_gate_apply: bad shape");
    if (rows == 0) return ITR_OK;
    ITR_REQUIRE(q && k && M && qo && ko, "itr_gate_apply: null pointer");
    hipLaunchKernelGGL(itr::gate_apply_kernel, EW_GRID(rows * dk), q, k, M, qo, ko, rows, dk);
    ITR_CHECK_LAUNCH("gate_apply");
    return ITR_OK;
}

extern "C" int itr_gate_apply_bwd(const float *q, const float *k, const float *M, const float *dqo, const float *dko, float *dq, float *dk_out,
                                  float *dM, int64_t rows, int dk, itr_stream_t stream) {
    ITR_REQUIRE(rows >= 0 && dk >= 1 && itr::ceil_div(rows * dk, (int64_t)256) <= 0x7fffffff, "itr_gate_apply_bwd: bad shape");
    if (rows == 0) return ITR_OK;
    ITR_REQUIRE(q && k && M && dqo && dko && dq && dk_out && dM, "itr_gate_apply_bwd: null pointer");
    hipLaunchKernelGGL(itr::gate_apply_bwd_kernel, EW_GRID(rows * dk), q, k, M, dqo, dko, dq, dk_out, dM, rows, dk);
    ITR_CHECK_LAUNCH("gate_apply_bwd");
    return ITR_OK;
}

extern "C" size_t itr_bn_train_scratch_bytes(int64_t N, int C) { return (size_t)2 * itr::BN_MAXSLICES * (size_t)C * sizeof(float); }

extern "C" int itr_bn_train_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *invstd, int64_t N, int C,
                                float eps, void *scratch, itr_stream_t stream) {
    ITR_REQUIRE(N >= 1 && C >= 1, "itr_bn_train_fwd: bad shape");
    ITR_REQUIRE(x && gamma && beta && y && mean && invstd && scratch, "itr_bn_train_fwd: null pointer (scratch: itr_bn_train_scratch_bytes)");
    const int slices = itr::bn_slices(N, C);
    float *s0 = static_cast<float *>(scratch), *s1 = s0 + (size_t)itr::BN_MAXSLICES * C;
    const dim3 grid((unsigned)itr::ceil_div(C, 64), (unsigned)slices);
    hipLaunchKernelGGL(itr::bn_partial_kernel, grid, dim3(256), 0, itr::as_stream(stream), x, (const float *)nullptr, s0, N, C, slices, 0);
    hipLaunchKernelGGL(itr::bn_partial_kernel, grid, dim3(256), 0, itr::as_stream(stream), x, (const float *)s0, s1, N, C, slices, 1);
    hipLaunchKernelGGL(itr::bn_train_apply_kernel, grid, dim3(256), 0, itr::as_stream(stream), x, gamma, beta, (const float *)s0, (const float *)s1, y,
                       mean, invstd, N, C, slices, eps);
    ITR_CHECK_LAUNCH("bn_train_fwd");
    return ITR_OK;
}

extern "C" int itr_bn_train_bwd(const float *dy, const float *x, const float *mean, const float *invstd, const float *gamma, float *dx,
                                float *dgamma, float *dbeta, int64_t N, int C, void *scratch, itr_stream_t stream) {
    ITR_REQUIRE(N >= 1 && C >= 1, "itr_bn_train_bwd: bad shape");
    ITR_REQUIRE(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && scratch, "itr_bn_train_bwd: null pointer (scratch: itr_bn_train_scratch_bytes)");
    const int slices = itr::bn_slices(N, C);
    float *pa = static_cast<float *>(scratch), *pb = pa + (size_t)itr::BN_MAXSLICES * C;
    const dim3 grid((unsigned)itr::ceil_div(C, 64), (unsigned)slices);
    hipLaunchKernelGGL(itr::bn_bwd_partial_kernel, grid, dim3(256), 0, itr::as_stream(stream), dy, x, mean, invstd, pa, pb, N, C, slices);
    hipLaunchKernelGGL(itr::bn_bwd_apply_kernel, grid, dim3(256), 0, itr::as_stream(stream), dy, x, mean, invstd, gamma, (const float *)pa,
                       (const float *)pb, dx, dgamma, dbeta, N, C, slices);
    ITR_CHECK_LAUNCH("bn_train_bwd");
    return ITR_OK;
}

extern "C" int itr_l2norm_mid_fwd(const float *x, float *z, float *norms, int64_t B, int R, int D, float eps, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && R >= 1 && D >= 1, "itr_l2norm_mid_fwd: bad shape (at most 65535 groups)");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(x && z && norms, "itr_l2norm_mid_fwd: null pointer");
    hipLaunchKernelGGL(itr::l2norm_mid_fwd_kernel, dim3((unsigned)itr::ceil_div(D, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), x, z,
                       norms, R, D, eps);
    ITR_CHECK_LAUNCH("l2norm_mid_fwd");
    return ITR_OK;
}

extern "C" int itr_l2norm_mid_bwd(const float *dz, const float *z, const float *norms, float *dx, int64_t B, int R, int D, float eps,
                                  itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && R >= 1 && D >= 1, "itr_l2norm_mid_bwd: bad shape (at most 65535 groups)");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(dz && z && norms && dx, "itr_l2norm_mid_bwd: null pointer");
    hipLaunchKernelGGL(itr::l2norm_mid_bwd_kernel, dim3((unsigned)itr::ceil_div(D, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), dz, z,
                       norms, dx, R, D, eps);
    ITR_CHECK_LAUNCH("l2norm_mid_bwd");
    return ITR_OK;
}

extern "C" int itr_smry_fwd(const float *smry, const float *x, float *L, float *out, int64_t B, int R, int K, int D, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && R >= 1 && R <= itr::SM_R && K >= 1 && K <= itr::SM_K && D >= 1, "itr_smry_fwd: at most 192 rows, 192 columns, 65535 groups");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(smry && x && L && out, "itr_smry_fwd: null pointer");
    const size_t lds = (size_t)R * K * sizeof(float);
    if (lds > 48 * 1024) {
        const int rc = itr::allow_dynamic_lds(reinterpret_cast<const void *>(itr::smry_fwd_kernel), (size_t)itr::SM_R * itr::SM_K * sizeof(float));
        if (rc != ITR_OK) return rc;
    }
    hipLaunchKernelGGL(itr::smry_fwd_kernel, dim3((unsigned)itr::ceil_div(D, 256), (unsigned)B), dim3(256), lds, itr::as_stream(stream), smry, x, L, out,
                       R, K, D);
    ITR_CHECK_LAUNCH("smry_fwd");
    return ITR_OK;
}

extern "C" int itr_smry_bwd(const float *x, const float *L, const float *dout, float *dx, float *dsmry, float *scratch, int64_t B, int R, int K,
                            int D, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && R >= 1 && R <= itr::SM_R && K >= 1 && K <= itr::SM_K && D >= 1, "itr_smry_bwd: at most 192 rows, 192 columns, 65535 groups");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(x && L && dout && dx && dsmry && scratch, "itr_smry_bwd: null pointer (scratch: B * R * K floats)");
    ITR_REQUIRE(B * R <= 0x7fffffff, "itr_smry_bwd: too many rows");
    hipLaunchKernelGGL(itr::smry_bwd_x_kernel, dim3((unsigned)itr::ceil_div(D, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), L, dout, dx, R,
                       K, D);
    hipLaunchKernelGGL(itr::smry_bwd_l_kernel, dim3((unsigned)(B * R)), dim3(256), 0, itr::as_stream(stream), x, dout, scratch, R, K, D);
    hipLaunchKernelGGL(itr::smry_bwd_softmax_kernel, dim3((unsigned)B), dim3(256), 0, itr::as_stream(stream), L, scratch, dsmry, R, K);
    ITR_CHECK_LAUNCH("smry_bwd");
    return ITR_OK;
}

extern "C" int itr_groupmax_fwd(const float *T, int64_t Ni, int k, int64_t Nc, float *S, int32_t *arg, itr_stream_t stream) {
    ITR_REQUIRE(Ni >= 0 && Ni <= 65535 && k >= 1 && Nc >= 0, "itr_groupmax_fwd: bad shape (at most 65535 images per call)");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    ITR_REQUIRE(T && S && arg, "itr_groupmax_fwd: null pointer");
    hipLaunchKernelGGL(itr::groupmax_fwd_kernel, dim3((unsigned)itr::ceil_div(Nc, (int64_t)256), (unsigned)Ni), dim3(256), 0, itr::as_stream(stream), T, k,
                       Nc, S, arg);
    ITR_CHECK_LAUNCH("groupmax_fwd");
    return ITR_OK;
}

extern "C" int itr_groupmax_bwd(const float *dS, const int32_t *arg, int64_t Ni, int k, int64_t Nc, float *dT, itr_stream_t stream) {
    ITR_REQUIRE(Ni >= 0 && Ni <= 65535 && k >= 1 && Nc >= 0, "itr_groupmax_bwd: bad shape (at most 65535 images per call)");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    ITR_REQUIRE(dS && arg && dT, "itr_groupmax_bwd: null pointer");
    hipLaunchKernelGGL(itr::groupmax_bwd_kernel, dim3((unsigned)itr::ceil_div(Nc, (int64_t)256), (unsigned)Ni), dim3(256), 0, itr::as_stream(stream), dS,
                       arg, k, Nc, dT);
    ITR_CHECK_LAUNCH("groupmax_bwd");
    return ITR_OK;
}

extern "C" int itr_bmm_small(const float *A, const float *B, float *C, int64_t batch, int M, int N, int K, int trans_a, int trans_b,
                             itr_stream_t stream) {
    ITR_REQUIRE(batch >= 0 && batch <= 65535 && M >= 1 && N >= 1 && K >= 1, "itr_bmm_small: bad shape (at most 65535 batch items)");
    if (batch == 0) return ITR_OK;
    ITR_REQUIRE(A && B && C, "itr_bmm_small: null pointer");
    const int64_t gq = itr::ceil_div((int64_t)M * N, (int64_t)256);
    const unsigned gx = (unsigned)(gq < 64 ? gq : 64);
    hipLaunchKernelGGL(itr::bmm_small_kernel, dim3(gx, (unsigned)batch), dim3(256), 0, itr::as_stream(stream), A, B, C, M, N, K, trans_a, trans_b);
    ITR_CHECK_LAUNCH("bmm_small");
    return ITR_OK;
}
