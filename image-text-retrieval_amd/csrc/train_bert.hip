// Training-step primitives of the transformer towers (SAEM: TransformerMapping / BertMapping under autograd,
// itr/modalmodule/ImgEncoder.py:324-350, TextEncoder.py:75-152, bert.py:113-300).  Small-sequence kernels (36 regions / 32
// tokens): everything here is HBM- or latency-bound; the dense layers around them are gemm_nt_kernel (autograd.py: _Linear).
//   itr_dropout                     y = x * keep / (1 - p), keep from a counter-based hash of (seed, element index) -- stateless,
//                                   so the backward pass is the same call on dy (nn.Dropout; masks are not stored)
//   itr_add_ln_fwd / itr_ln_bwd     BERTLayerNorm(x + residual) (bert.py:113-126, TF style: eps inside the sqrt) keeping
//                                   z = x + residual, mean, rstd;  dz = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma
//   itr_gelu_fwd / itr_gelu_bwd     x * 0.5 * (1 + erf(x / sqrt 2))  (bert.py:104-110)
//   itr_mha_train_fwd / _bwd        softmax(Q K^T / sqrt(dk) + (1 - mask) * -10000) -> dropout -> . V per (sequence, head)
//                                   (bert.py:175-215), probabilities kept for the backward pass
//   itr_relu_maxpool_arg / _bwd     max over positions of relu(x) with the arg-max kept (F.relu + F.max_pool1d, TextEncoder.py:122-124)
//   itr_bcast_mid                   dx[b, r, :] = scale * dy[b, :]   (backward of torch.mean(x, 1))
#include "itr_common.h"

namespace itr {

__device__ __forceinline__ uint32_t hash32(uint64_t idx, uint64_t seed) {      // splitmix64 finaliser, upper half
    uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
__device__ __forceinline__ float drop_scale(uint64_t idx, uint64_t seed, uint32_t thresh, float inv_keep) {
    return hash32(idx, seed) >= thresh ? inv_keep : 0.f;                          // P(keep) = 1 - thresh / 2^32
}

__global__ __launch_bounds__(256) void dropout_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n, uint32_t thresh,
                                                      float inv_keep, uint64_t seed, uint64_t offset) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] * drop_scale(offset + i, seed, thresh, inv_keep);
}

// one wave per row
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         float *__restrict__ z, float *__restrict__ out, float *__restrict__ mean,
                                                         float *__restrict__ rstd, int64_t rows, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float v = x[row * H + c] + (res ? res[row * H + c] : 0.f);
        z[row * H + c] = v;
        s += v;
    }
    const float u = wave_sum(s) / (float)H;
    float q = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float d = z[row * H + c] - u;
        q += d * d;
    }
    const float r = 1.f / sqrtf(wave_sum(q) / (float)H + eps);
    for (int c = lane; c < H; c += 64) out[row * H + c] = gamma[c] * ((z[row * H + c] - u) * r) + beta[c];
    if (lane == 0) { mean[row] = u; rstd[row] = r; }
}

// dz and t = dy * xhat (for the gamma gradient: dgamma = colsum(t), dbeta = colsum(dy))
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ z,
                                                     const float *__restrict__ mean, const float *__restrict__ rstd,
                                                     const float *__restrict__ gamma, float *__restrict__ dz, float *__restrict__ t,
                                                     int64_t rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float u = mean[row], r = rstd[row];
    float a = 0.f, b = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float xh = (z[row * H + c] - u) * r, g = dy[row * H + c] * gamma[c];
        a += g;
        b += g * xh;
        t[row * H + c] = dy[row * H + c] * xh;
    }
    a = wave_sum(a) / (float)H;
    b = wave_sum(b) / (float)H;
    for (int c = lane; c < H; c += 64) {
        const float xh = (z[row * H + c] - u) * r, g = dy[row * H + c] * gamma[c];
        dz[row * H + c] = r * (g - a - xh * b);
    }
}

__global__ __launch_bounds__(256) void gelu_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ out,
                                                   int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
    if (!dy) out[i] = v * cdf;
    else out[i] = dy[i] * (cdf + v * 0.39894228040143268f * expf(-0.5f * v * v));
}

// ---- multi-head attention of short sequences, one workgroup per (sequence, head) --------------------------------------------
constexpr int MT_L = 64, MT_D = 64;      // at most 64 positions, head size at most 64
struct MhaT {
    const float *q, *k, *v;              // row (b * L + i), column head * dk + d, row stride ld
    int64_t ld;
    const float *mask01;                 // [B, L] or null
    float *P;                            // [B, heads, L, L] softmax probabilities (before dropout)
    int L, heads, dk;
    float scale, inv_keep;
    uint32_t thresh;
    uint64_t seed;
};

__global__ __launch_bounds__(256) void mha_train_fwd_kernel(MhaT a, float *__restrict__ ctx, int64_t ldc) {
    extern __shared__ float mha_lds[];
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads, t = threadIdx.x, L = a.L, dk = a.dk;
    const int sd_ = dk + 1, sl_ = L + 1;                       // padded row strides
    float *const sq_ = mha_lds, *const sk_ = sq_ + L * sd_, *const sv_ = sk_ + L * sd_, *const sp_ = sv_ + L * sd_;
#define sq(i, d) sq_[(i) * sd_ + (d)]
#define sk(i, d) sk_[(i) * sd_ + (d)]
#define sv(i, d) sv_[(i) * sd_ + (d)]
#define sp(i, j) sp_[(i) * sl_ + (j)]
    for (int e = t; e < L * dk; e += 256) {
        const int i = e / dk, d = e % dk;
        const int64_t o = ((int64_t)b * L + i) * a.ld + h * dk + d;
        sq(i, d) = a.q[o]; sk(i, d) = a.k[o]; sv(i, d) = a.v[o];
    }
    __syncthreads();
    for (int e = t; e < L * L; e += 256) {
        const int i = e / L, j = e % L;
        float s = 0.f;
        for (int d = 0; d < dk; ++d) s = fmaf(sq(i, d), sk(j, d), s);
        s = s * a.scale + (a.mask01 ? (1.f - a.mask01[(int64_t)b * L + j]) * -10000.f : 0.f);
        sp(i, j) = s;
    }
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    for (int i = wave; i < L; i += 4) {                       // softmax of row i by one wave
        const float v = lane < L ? sp(i, lane) : -INFINITY;
        const float m = wave_max(v);
        const float ex = lane < L ? expf(v - m) : 0.f;
        const float den = wave_sum(ex);
        if (lane < L) {
            const float p = ex / den;
            a.P[(((int64_t)b * a.heads + h) * L + i) * L + lane] = p;
            sp(i, lane) = p * drop_scale((((uint64_t)b * a.heads + h) * L + i) * L + lane, a.seed, a.thresh, a.inv_keep);
        }
    }
    __syncthreads();
    for (int e = t; e < L * dk; e += 256) {
        const int i = e / dk, d = e % dk;
        float s = 0.f;
        for (int j = 0; j < L; ++j) s = fmaf(sp(i, j), sv(j, d), s);
        ctx[((int64_t)b * L + i) * ldc + h * dk + d] = s;
    }
}

// dctx [B*L, ldc] -> dq, dk, dv written with the layout of q / k / v (row stride ldg)
__global__ __launch_bounds__(256) void mha_train_bwd_kernel(MhaT a, const float *__restrict__ dctx, int64_t ldc, float *__restrict__ dq,
                                                            float *__restrict__ dkk, float *__restrict__ dv, int64_t ldg) {
    extern __shared__ float mha_lds[];
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads, t = threadIdx.x, L = a.L, dk = a.dk;
    const int sd_ = dk + 1, sl_ = L + 1;
    float *const sq_ = mha_lds, *const sk_ = sq_ + L * sd_, *const sv_ = sk_ + L * sd_, *const sg_ = sv_ + L * sd_, *const sp_ = sg_ + L * sd_;
#define sg(i, d) sg_[(i) * sd_ + (d)]
    for (int e = t; e < L * dk; e += 256) {
        const int i = e / dk, d = e % dk;
        const int64_t o = ((int64_t)b * L + i) * a.ld + h * dk + d;
        sq(i, d) = a.q[o]; sk(i, d) = a.k[o]; sv(i, d) = a.v[o];
        sg(i, d) = dctx[((int64_t)b * L + i) * ldc + h * dk + d];
    }
    __syncthreads();
    // dv[j, d] = sum_i Pd[i, j] dctx[i, d]  with Pd = P * dropout scale;   sp <- dPd[i, j] = dctx[i, :] . v[j, :]
    const float *Pg = a.P + ((int64_t)b * a.heads + h) * L * L;
    const uint64_t pbase = ((uint64_t)b * a.heads + h) * L * L;
    for (int e = t; e < L * dk; e += 256) {
        const int j = e / dk, d = e % dk;
        float s = 0.f;
        for (int i = 0; i < L; ++i)
            s = fmaf(Pg[i * L + j] * drop_scale(pbase + (uint64_t)i * L + j, a.seed, a.thresh, a.inv_keep), sg(i, d), s);
        dv[((int64_t)b * L + j) * ldg + h * dk + d] = s;
    }
    for (int e = t; e < L * L; e += 256) {
        const int i = e / L, j = e % L;
        float s = 0.f;
        for (int d = 0; d < dk; ++d) s = fmaf(sg(i, d), sv(j, d), s);
        sp(i, j) = s * drop_scale(pbase + (uint64_t)i * L + j, a.seed, a.thresh, a.inv_keep);     // dP (pre-dropout probabilities)
    }
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    for (int i = wave; i < L; i += 4) {                       // dS = P (dP - sum_j dP P), times the score scale
        const float p = lane < L ? Pg[i * L + lane] : 0.f, g = lane < L ? sp(i, lane) : 0.f;
        const float dot = wave_sum(p * g);
        if (lane < L) sp(i, lane) = p * (g - dot) * a.scale;
    }
    __syncthreads();
    for (int e = t; e < L * dk; e += 256) {
        const int i = e / dk, d = e % dk;
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < L; ++j) {
            s1 = fmaf(sp(i, j), sk(j, d), s1);               // dq[i] = sum_j dS[i, j] k[j]
            s2 = fmaf(sp(j, i), sq(j, d), s2);               // dk[i] = sum_j dS[j, i] q[j]
        }
        dq[((int64_t)b * L + i) * ldg + h * dk + d] = s1;
        dkk[((int64_t)b * L + i) * ldg + h * dk + d] = s2;
    }
}
#undef sq
#undef sk
#undef sv
#undef sg
#undef sp

// x [B, npos, C] -> out[b, c] = max_p relu(x[b, p, c]), arg[b, c] = first position of the maximum (-1 when the maximum is 0)
__global__ __launch_bounds__(256) void relu_maxpool_arg_kernel(const float *__restrict__ x, int npos, int C, float *__restrict__ out,
                                                               int32_t *__restrict__ arg) {
    const int64_t b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float m = 0.f;
    int am = -1;
    for (int p = 0; p < npos; ++p) {
        const float v = x[(b * npos + p) * C + c];
        if (v > m) { m = v; am = p; }
    }
    out[b * C + c] = m;
    arg[b * C + c] = am;
}
__global__ __launch_bounds__(256) void relu_maxpool_bwd_kernel(const float *__restrict__ dy, const int32_t *__restrict__ arg, int npos,
                                                               int C, float *__restrict__ dx) {
    const int64_t b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int am = arg[b * C + c];
    for (int p = 0; p < npos; ++p) dx[(b * npos + p) * C + c] = p == am ? dy[b * C + c] : 0.f;
}

__global__ __launch_bounds__(256) void bcast_mid_kernel(const float *__restrict__ dy, float *__restrict__ dx, int R, int F, float scale) {
    const int64_t b = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const float v = dy[b * F + f] * scale;
    for (int r = 0; r < R; ++r) dx[(b * R + r) * F + f] = v;
}

static bool drop_params(float p, uint32_t *thresh, float *inv_keep) {
    if (!(p >= 0.f && p < 1.f)) return false;
    const double t = (double)p * 4294967296.0;
    *thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    *inv_keep = 1.f / (1.f - p);
    return true;
}

}  // namespace itr

extern "C" int itr_dropout(const float *x, float *y, int64_t n, float p, uint64_t seed, uint64_t offset, itr_stream_t stream) {
    uint32_t th; float ik;
    ITR_REQUIRE(n >= 0 && itr::drop_params(p, &th, &ik), "itr_dropout: bad size or p outside [0, 1)");
    if (n == 0) return ITR_OK;
    ITR_REQUIRE(x && y, "itr_dropout: null pointer");
    ITR_REQUIRE(itr::ceil_div(n, (int64_t)256) <= 0x7fffffff, "itr_dropout: too many elements");
    hipLaunchKernelGGL(itr::dropout_kernel, dim3((unsigned)itr::ceil_div(n, (int64_t)256)), dim3(256), 0, itr::as_stream(stream), x, y, n, th,
                       ik, seed, offset);
    ITR_CHECK_LAUNCH("dropout");
    return ITR_OK;
}

extern "C" int itr_add_ln_fwd(const float *x, const float *residual, const float *gamma, const float *beta, float *z, float *out,
                              float *mean, float *rstd, int64_t rows, int H, float eps, itr_stream_t stream) {
    ITR_REQUIRE(rows >= 0 && H >= 1, "itr_add_ln_fwd: bad shape");
    if (rows == 0) return ITR_OK;
    ITR_REQUIRE(x && gamma && beta && z && out && mean && rstd, "itr_add_ln_fwd: null pointer");
    hipLaunchKernelGGL(itr::add_ln_fwd_kernel, dim3((unsigned)itr::ceil_div(rows, (int64_t)4)), dim3(256), 0, itr::as_stream(stream), x,
                       residual, gamma, beta, z, out, mean, rstd, rows, H, eps);
    ITR_CHECK_LAUNCH("add_ln_fwd");
    return ITR_OK;
}

extern "C" int itr_ln_bwd(const float *dy, const float *z, const float *mean, const float *rstd, const float *gamma, float *dz, float *t,
                          int64_t rows, int H, itr_stream_t stream) {
    ITR_REQUIRE(rows >= 0 && H >= 1, "itr_ln_bwd: bad shape");
    if (rows == 0) return ITR_OK;
    ITR_REQUIRE(dy && z && mean && rstd && gamma && dz && t, "itr_ln_bwd: null pointer");
    hipLaunchKernelGGL(itr::ln_bwd_kernel, dim3((unsigned)itr::ceil_div(rows, (int64_t)4)), dim3(256), 0, itr::as_stream(stream), dy, z, mean,
                       rstd, gamma, dz, t, rows, H);
    ITR_CHECK_LAUNCH("ln_bwd");
    return ITR_OK;
}

extern "C" int itr_gelu(const float *x, const float *dy, float *out, int64_t n, itr_stream_t stream) {
    ITR_REQUIRE(n >= 0, "itr_gelu: bad size");
    if (n == 0) return ITR_OK;
    ITR_REQUIRE(x && out, "itr_gelu: null pointer");
    hipLaunchKernelGGL(itr::gelu_kernel, dim3((unsigned)itr::ceil_div(n, (int64_t)256)), dim3(256), 0, itr::as_stream(stream), x, dy, out, n);
    ITR_CHECK_LAUNCH("gelu");
    return ITR_OK;
}

static int mha_args(itr::MhaT *a, const float *q, const float *k, const float *v, int64_t ld, const float *mask01, float *P, int64_t B,
                    int L, int heads, int dk, float scale, float p_drop, uint64_t seed) {
    ITR_REQUIRE(B >= 0 && L >= 1 && L <= itr::MT_L && heads >= 1 && dk >= 1 && dk <= itr::MT_D, "itr_mha_train: at most 64 positions, head size at most 64");
    ITR_REQUIRE(B * heads <= 0x7fffffff, "itr_mha_train: too many (sequence, head) pairs");
    ITR_REQUIRE(itr::drop_params(p_drop, &a->thresh, &a->inv_keep), "itr_mha_train: p outside [0, 1)");
    a->q = q; a->k = k; a->v = v; a->ld = ld; a->mask01 = mask01; a->P = P; a->L = L; a->heads = heads; a->dk = dk; a->scale = scale;
    a->seed = seed;
    return ITR_OK;
}

extern "C" int itr_mha_train_fwd(const float *q, const float *k, const float *v, int64_t ld, const float *mask01, int64_t B, int L,
                                 int heads, int dk, float scale, float p_drop, uint64_t seed, float *P, float *ctx, int64_t ldc,
                                 itr_stream_t stream) {
    itr::MhaT a{};
    const int rc = mha_args(&a, q, k, v, ld, mask01, P, B, L, heads, dk, scale, p_drop, seed);
    if (rc != ITR_OK) return rc;
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(q && k && v && P && ctx, "itr_mha_train_fwd: null pointer");
    const size_t lds = (size_t)(3 * L * (dk + 1) + L * (L + 1)) * 4;
    static bool attr = false;
    if (!attr) {
        ITR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(itr::mha_train_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(itr::mha_train_fwd_kernel, dim3((unsigned)(B * heads)), dim3(256), lds, itr::as_stream(stream), a, ctx, ldc);
    ITR_CHECK_LAUNCH("mha_train_fwd");
    return ITR_OK;
}

extern "C" int itr_mha_train_bwd(const float *q, const float *k, const float *v, int64_t ld, const float *mask01, int64_t B, int L,
                                 int heads, int dk, float scale, float p_drop, uint64_t seed, const float *P, const float *dctx,
                                 int64_t ldc, float *dq, float *dk_out, float *dv, int64_t ldg, itr_stream_t stream) {
    itr::MhaT a{};
    const int rc = mha_args(&a, q, k, v, ld, mask01, const_cast<float *>(P), B, L, heads, dk, scale, p_drop, seed);
    if (rc != ITR_OK) return rc;
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(q && k && v && P && dctx && dq && dk_out && dv, "itr_mha_train_bwd: null pointer");
    const size_t lds = (size_t)(4 * L * (dk + 1) + L * (L + 1)) * 4;
    static bool attr = false;
    if (!attr) {
        ITR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(itr::mha_train_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(itr::mha_train_bwd_kernel, dim3((unsigned)(B * heads)), dim3(256), lds, itr::as_stream(stream), a, dctx, ldc, dq, dk_out,
                       dv, ldg);
    ITR_CHECK_LAUNCH("mha_train_bwd");
    return ITR_OK;
}

extern "C" int itr_relu_maxpool_arg(const float *x, int64_t B, int npos, int C, float *out, int32_t *arg, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && npos >= 1 && C >= 1 && B <= 65535, "itr_relu_maxpool_arg: bad shape (at most 65535 rows)");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(x && out && arg, "itr_relu_maxpool_arg: null pointer");
    hipLaunchKernelGGL(itr::relu_maxpool_arg_kernel, dim3((unsigned)itr::ceil_div(C, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), x,
                       npos, C, out, arg);
    ITR_CHECK_LAUNCH("relu_maxpool_arg");
    return ITR_OK;
}

extern "C" int itr_relu_maxpool_bwd(const float *dy, const int32_t *arg, int64_t B, int npos, int C, float *dx, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && npos >= 1 && C >= 1 && B <= 65535, "itr_relu_maxpool_bwd: bad shape (at most 65535 rows)");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(dy && arg && dx, "itr_relu_maxpool_bwd: null pointer");
    hipLaunchKernelGGL(itr::relu_maxpool_bwd_kernel, dim3((unsigned)itr::ceil_div(C, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), dy,
                       arg, npos, C, dx);
    ITR_CHECK_LAUNCH("relu_maxpool_bwd");
    return ITR_OK;
}

extern "C" int itr_bcast_mid(const float *dy, float *dx, int64_t B, int R, int F, float scale, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && R >= 1 && F >= 1 && B <= 65535, "itr_bcast_mid: bad shape (at most 65535 groups)");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(dy && dx, "itr_bcast_mid: null pointer");
    hipLaunchKernelGGL(itr::bcast_mid_kernel, dim3((unsigned)itr::ceil_div(F, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), dy, dx, R,
                       F, scale);
    ITR_CHECK_LAUNCH("bcast_mid");
    return ITR_OK;
}
