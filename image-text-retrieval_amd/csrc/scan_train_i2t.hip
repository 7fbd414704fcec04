// SCAN i2t similarity for TRAINING batches (xattn_score_i2t + func_attention + cosine_similarity,
// Objectives.py:376-417, :420-476, :10-15, under autograd): the 36 regions of an image attend over the W words of a
// caption.  Same decomposition as scan_train.hip (t2i) with the roles of regions and words exchanged:
//       A[B_i * 36, n_tok] = V E^T  (one GEMM),  H_c = E_c E_c^T  (W x W per caption),  ||v_r||
//       b = f(a);  u[r, w] = first norm of b[:, w] over the 36 REGIONS, per word      (Objectives.py:436-457, dim 2 = queryL)
//       p[r, w] = softmax_w(lambda_s u[r, w])                                         (over the caption's words)
//       num_r = sum_w p a     q_r = p_r^T H_c p_r     s_r = num_r / max(||v_r|| sqrt(q_r), 1e-8)
//       S = LSE_r(lambda_lse s_r) / lambda_lse | max | sum | mean                      (over the 36 regions)
// Backward per (image, caption) pair: dA block (disjoint), dH partial [W x W] (summed over images by itr_colsum),
// d||v_r|| partial (summed over captions by itr_colsum).  Caller:  dV = dA E + d||v|| v / ||v||,
// dE = dA^T V + (dH + dH^T) E_c.
#include "scan_common.h"

namespace itr {

constexpr int SI_RMAX = 100;   // as ST_RMAX (scan_train.hip)
constexpr int SI_MAXW = 96;    // as scan_train.hip; the W x W caption Gram makes the pair 80 KB -> dynamic LDS

struct ScanI2TArgs {
    const float *A;        // [Bi*36, ldA] raw dot products
    int64_t ldA;
    const float *H;        // packed caption Grams: caption c at H + h_off[c], W_c x W_c row-major
    const int64_t *h_off;  // [Bc]
    const float *vnorm;    // [Bi*36]
    const int64_t *cap_off;
    const int32_t *cap_len;
    int64_t Bi, Bc, h_total;
    int R;                 // regions per image: 36 in every reference configuration (FIXED instantiation), 1..SI_RMAX otherwise
    int norm, agg;
    float ls, ll;
    float *S;              // [Bi, Bc]
    const float *dS;
    float *dA;             // [Bi*36, ldA]
    float *dHp;            // [Bi, h_total]     per-pair partials of dH_c
    float *dvn;            // [Bc, Bi*36]       per-pair d||v_r||
};

template <int RMAX>
struct I2TSmem {
    float a[RMAX][SI_MAXW + 1];    // raw
    float p[RMAX][SI_MAXW + 1];    // u, then attention weights
    float hp[RMAX][SI_MAXW + 1];   // H p_r; the backward reuses it for dp / du / da
    float h[SI_MAXW][SI_MAXW + 1]; // caption Gram
    float wn[SI_MAXW];             // first-norm statistic per word: 1/(||b|| + eps) | 1/(sum|b| + eps) | 1/sum exp
    float ws[SI_MAXW];             // ||b||, sum |b|, or the column maximum (softmax)
    float s[RMAX], num[RMAX], q[RMAX], ds[RMAX];
    float dqs[RMAX], dnums[RMAX];  // backward
};

template <int RMAX, bool FIXED>
__device__ __forceinline__ void i2t_pair_forward(const ScanI2TArgs &g, I2TSmem<RMAX> &sm, int64_t i, int64_t c, int W, int64_t off) {
    const int tid = threadIdx.x;
    const int R = FIXED ? RMAX : g.R;
    for (int idx = tid; idx < R * W; idx += 256) {
        const int r = idx / W, w = idx - r * W;
        sm.a[r][w] = g.A[(i * R + r) * g.ldA + off + w];
    }
    const float *H = g.H + g.h_off[c];
    for (int idx = tid; idx < W * W; idx += 256) sm.h[idx / W][idx % W] = H[idx];
    __syncthreads();
    const bool clip = (g.norm == 0 || g.norm == 4 || g.norm == 6), l2 = (g.norm == 0 || g.norm == 1), l1 = (g.norm == 5 || g.norm == 6);
    if (tid < W) {   // one lane per word: first normalisation over the regions
        const int w = tid;
        float st = 0.f, mxa = -INFINITY;
        for (int r = 0; r < R; ++r) {
            const float b = clip ? leaky(sm.a[r][w]) : sm.a[r][w];
            st += l2 ? b * b : (l1 ? fabsf(b) : 0.f);
            mxa = fmaxf(mxa, sm.a[r][w]);
        }
        if (l2) st = sqrtf(st);
        if (g.norm == 2) {
            st = 0.f;
            for (int r = 0; r < R; ++r) st += expf(sm.a[r][w] - mxa);
        }
        sm.ws[w] = (g.norm == 2) ? mxa : st;
        const float rn = (l2 || l1) ? 1.f / (st + 1e-8f) : (g.norm == 2 ? 1.f / st : 1.f);
        sm.wn[w] = rn;
        for (int r = 0; r < R; ++r) {
            const float araw = sm.a[r][w];
            sm.p[r][w] = (g.norm == 2) ? expf(araw - mxa) * rn : (clip ? leaky(araw) : araw) * rn;
        }
    }
    __syncthreads();
    if (tid < R) {   // one lane per region: softmax over the words, cosine terms
        const int r = tid;
        float mx = -INFINITY;
        for (int w = 0; w < W; ++w) {
            const float u = sm.p[r][w] * g.ls;
            sm.p[r][w] = u;
            mx = fmaxf(mx, u);
        }
        float den = 0.f;
        for (int w = 0; w < W; ++w) {
            const float e = expf(sm.p[r][w] - mx);
            sm.p[r][w] = e;
            den += e;
        }
        float num = 0.f;
        for (int w = 0; w < W; ++w) {
            const float pv = sm.p[r][w] / den;
            sm.p[r][w] = pv;
            num += pv * sm.a[r][w];
        }
        float q = 0.f;
        for (int w = 0; w < W; ++w) {
            float t = 0.f;
            for (int v = 0; v < W; ++v) t += sm.h[w][v] * sm.p[r][v];
            sm.hp[r][w] = t;
            q += sm.p[r][w] * t;
        }
        q = fmaxf(q, 0.f);
        sm.num[r] = num;
        sm.q[r] = q;
        sm.s[r] = num / fmaxf(g.vnorm[i * R + r] * sqrtf(q), 1e-8f);
    }
    __syncthreads();
}

template <int RMAX, bool FIXED>
__global__ __launch_bounds__(256) void scan_train_i2t_fwd_kernel(ScanI2TArgs g) {
    extern __shared__ __attribute__((aligned(16))) char i2t_smem[];
    I2TSmem<RMAX> &sm = *reinterpret_cast<I2TSmem<RMAX> *>(i2t_smem);
    const int R = FIXED ? RMAX : g.R;
    const int64_t c = blockIdx.x, i = blockIdx.y;
    const int W = g.cap_len[c];
    i2t_pair_forward<RMAX, FIXED>(g, sm, i, c, W, g.cap_off[c]);
    if (threadIdx.x == 0) {
        float r;
        if (g.agg == 0) {
            float mx = -INFINITY;
            for (int t = 0; t < R; ++t) mx = fmaxf(mx, sm.s[t] * g.ll);
            float acc = 0.f;
            for (int t = 0; t < R; ++t) acc += expf(sm.s[t] * g.ll - mx);
            r = (logf(acc) + mx) / g.ll;
        } else if (g.agg == 1) {
            r = -INFINITY;
            for (int t = 0; t < R; ++t) r = fmaxf(r, sm.s[t]);
        } else {
            r = 0.f;
            for (int t = 0; t < R; ++t) r += sm.s[t];
            if (g.agg == 3) r /= (float)R;
        }
        g.S[i * g.Bc + c] = r;
    }
}

template <int RMAX, bool FIXED>
__global__ __launch_bounds__(256) void scan_train_i2t_bwd_kernel(ScanI2TArgs g) {
    extern __shared__ __attribute__((aligned(16))) char i2t_smem[];
    I2TSmem<RMAX> &sm = *reinterpret_cast<I2TSmem<RMAX> *>(i2t_smem);
    float(&dqs)[RMAX] = sm.dqs;
    float(&dnums)[RMAX] = sm.dnums;
    const int R = FIXED ? RMAX : g.R;
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x, i = blockIdx.y;
    const int W = g.cap_len[c];
    const int64_t off = g.cap_off[c];
    i2t_pair_forward<RMAX, FIXED>(g, sm, i, c, W, off);
    const float dS = g.dS[i * g.Bc + c];
    if (tid == 0) {
        if (g.agg == 0) {
            float mx = -INFINITY;
            for (int t = 0; t < R; ++t) mx = fmaxf(mx, sm.s[t] * g.ll);
            float acc = 0.f;
            for (int t = 0; t < R; ++t) acc += expf(sm.s[t] * g.ll - mx);
            for (int t = 0; t < R; ++t) sm.ds[t] = dS * expf(sm.s[t] * g.ll - mx) / acc;
        } else if (g.agg == 1) {
            int best = 0;
            for (int t = 1; t < R; ++t)
                if (sm.s[t] > sm.s[best]) best = t;
            for (int t = 0; t < R; ++t) sm.ds[t] = (t == best) ? dS : 0.f;
        } else {
            const float k = (g.agg == 3) ? dS / (float)R : dS;
            for (int t = 0; t < R; ++t) sm.ds[t] = k;
        }
    }
    __syncthreads();
    const bool clip = (g.norm == 0 || g.norm == 4 || g.norm == 6), l2 = (g.norm == 0 || g.norm == 1), l1 = (g.norm == 5 || g.norm == 6);
    // ---- per region: cosine backward, softmax-over-words backward -> du (kept in hp)
    if (tid < R) {
        const int r = tid;
        const float vn = g.vnorm[i * R + r];
        const float sq = sqrtf(sm.q[r]);
        const float den = vn * sq;
        float dnum = 0.f, dq = 0.f, dvn = 0.f;
        if (den > 1e-8f) {
            dnum = sm.ds[r] / den;
            const float dden = -sm.ds[r] * sm.num[r] / (den * den);
            dvn = dden * sq;
            dq = sq > 0.f ? dden * vn / (2.f * sq) : 0.f;
        } else {
            dnum = sm.ds[r] / 1e-8f;
        }
        dqs[r] = dq;
        dnums[r] = dnum;
        g.dvn[c * (g.Bi * R) + i * R + r] = dvn;
        float dot = 0.f;
        for (int w = 0; w < W; ++w) {
            const float dp = dnum * sm.a[r][w] + 2.f * dq * sm.hp[r][w];
            sm.hp[r][w] = dp;
            dot += sm.p[r][w] * dp;
        }
        for (int w = 0; w < W; ++w) sm.hp[r][w] = g.ls * sm.p[r][w] * (sm.hp[r][w] - dot);     // du[r][w]
    }
    __syncthreads();
    // ---- per word: first-norm backward over the regions
    if (tid < W) {
        const int w = tid;
        const float rn = sm.wn[w], rt = sm.ws[w];
        float dot = 0.f;
        for (int r = 0; r < R; ++r) {
            const float araw = sm.a[r][w];
            const float b = clip ? leaky(araw) : araw;
            if (l2 || l1) dot += sm.hp[r][w] * b;
            else if (g.norm == 2) dot += sm.hp[r][w] * expf(araw - rt) * rn;
        }
        for (int r = 0; r < R; ++r) {
            const float araw = sm.a[r][w];
            const float b = clip ? leaky(araw) : araw;
            const float du = sm.hp[r][w];
            float db;
            if (l2) db = du * rn - (rt > 0.f ? b * dot * rn * rn / rt : 0.f);
            else if (l1) db = du * rn - (b > 0.f ? 1.f : (b < 0.f ? -1.f : 0.f)) * dot * rn * rn;
            else if (g.norm == 2) { const float u = expf(araw - rt) * rn; db = u * (du - dot); }
            else db = du;
            if (clip) db *= (araw > 0.f) ? 1.f : 0.1f;
            sm.hp[r][w] = db + dnums[r] * sm.p[r][w];          // + the direct path of num = sum_w p a
        }
    }
    __syncthreads();
    for (int idx = tid; idx < R * W; idx += 256) {
        const int r = idx / W, w = idx - r * W;
        g.dA[(i * R + r) * g.ldA + off + w] = sm.hp[r][w];
    }
    // ---- dH partial of this pair: sum_r dq_r p_r p_r^T
    float *dhp = g.dHp + i * g.h_total + g.h_off[c];
    for (int idx = tid; idx < W * W; idx += 256) {
        const int u = idx / W, v = idx - u * W;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += dqs[r] * sm.p[r][u] * sm.p[r][v];
        dhp[idx] = acc;
    }
}

// dE_c += (dH_c + dH_c^T) E_c, one workgroup per caption;  dH_c = the colsum over images of the pair partials
__global__ __launch_bounds__(256) void i2t_gram_bwd_kernel(const float *__restrict__ dH, const int64_t *__restrict__ h_off,
                                                           const int64_t *__restrict__ cap_off, const int32_t *__restrict__ cap_len,
                                                           const float *__restrict__ E, int D, float *__restrict__ dE) {
    __shared__ float dh[SI_MAXW][SI_MAXW + 1];
    const int64_t c = blockIdx.x;
    const int W = cap_len[c];
    const float *src = dH + h_off[c];
    for (int idx = threadIdx.x; idx < W * W; idx += 256) dh[idx / W][idx % W] = src[idx];
    __syncthreads();
    const float *Ec = E + cap_off[c] * (int64_t)D;
    float *dEc = dE + cap_off[c] * (int64_t)D;
    for (int d = threadIdx.x; d < D; d += 256) {
        for (int u = 0; u < W; ++u) {
            float acc = 0.f;
            for (int v = 0; v < W; ++v) acc += (dh[u][v] + dh[v][u]) * Ec[(int64_t)v * D + d];
            dEc[(int64_t)u * D + d] += acc;
        }
    }
}

__global__ __launch_bounds__(256) void rowscale_add_kernel(const float *__restrict__ X, const float *__restrict__ xnorm, const float *__restrict__ dn,
                                                           int64_t rows, int D, float *__restrict__ dX) {
    const int64_t row = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows || d >= D) return;
    const float n = xnorm[row];
    if (n > 0.f) dX[row * D + d] += dn[row] * X[row * D + d] / n;
}

__global__ __launch_bounds__(256) void rownorm_i2t_kernel(const float *__restrict__ x, int64_t rows, int D, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += x[row * D + d] * x[row * D + d];
    s = wave_sum(s);
    if (lane == 0) out[row] = sqrtf(s);
}

// H_c = E_c E_c^T for captions of up to SI_MAXW words (the evaluation path's gram_kernel stops at 64 rows):
// one workgroup per caption, 32-wide slices of D through LDS, up to 36 (row, row) pairs per thread.
__global__ __launch_bounds__(256) void caption_gram_kernel(const float *__restrict__ E, const int64_t *__restrict__ cap_off,
                                                           const int32_t *__restrict__ cap_len, int D, float *__restrict__ H,
                                                           const int64_t *__restrict__ h_off) {
    __shared__ float xs[SI_MAXW][33];
    const int64_t c = blockIdx.x;
    const int W = cap_len[c];
    const float *x = E + cap_off[c] * (int64_t)D;
    constexpr int NP = (SI_MAXW * SI_MAXW + 255) / 256;
    float acc[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < D; k0 += 32) {
        for (int idx = threadIdx.x; idx < W * 32; idx += 256) {
            const int r = idx >> 5, k = idx & 31;
            xs[r][k] = (k0 + k < D) ? x[(int64_t)r * D + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            const int pidx = threadIdx.x + 256 * e;
            if (pidx < W * W) {
                const int r1 = pidx / W, r2 = pidx % W;
                float s_ = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) s_ += xs[r1][k] * xs[r2][k];
                acc[e] += s_;
            }
        }
        __syncthreads();
    }
    float *out = H + h_off[c];
#pragma unroll
    for (int e = 0; e < NP; ++e) {
        const int pidx = threadIdx.x + 256 * e;
        if (pidx < W * W) out[pidx] = acc[e];
    }
}

int allow_dynamic_lds(const void *kernel, size_t bytes);      // scan_train.hip

static int check_i2t(const char *who, int64_t Bi, int64_t Bc, int64_t n_tok, int R, int D, int norm, int agg, int max_len) {
    ITR_REQUIRE(Bi >= 1 && Bc >= 1 && n_tok >= 1 && D > 0, "%s: bad shape", who);
    ITR_UNSUPPORTED(R < 1 || R > SI_RMAX, "%s: 1..%d regions per image are supported, got %d", who, SI_RMAX, R);
    ITR_UNSUPPORTED(max_len > SI_MAXW, "%s: captions of at most %d words are supported, got %d", who, SI_MAXW, max_len);
    ITR_REQUIRE(norm >= 0 && norm <= 6, "%s: unknown first norm %d", who, norm);
    ITR_REQUIRE(agg >= 0 && agg <= 3, "%s: unknown aggregation %d", who, agg);
    ITR_UNSUPPORTED(Bi > 65535, "%s: at most 65535 images per training batch", who);
    return ITR_OK;
}

template <int RMAX>
static int launch_i2t(void (*kernel)(ScanI2TArgs), const char *what, const ScanI2TArgs &g, hipStream_t st) {
    static_assert(sizeof(I2TSmem<RMAX>) <= 160 * 1024, "a pair's LDS block must fit one CU");
    const int rc = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), sizeof(I2TSmem<RMAX>));
    if (rc != ITR_OK) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)g.Bc, (unsigned)g.Bi), dim3(256), sizeof(I2TSmem<RMAX>), st, g);
    ITR_CHECK_LAUNCH(what);
    return ITR_OK;
}

}  // namespace itr

using namespace itr;

extern "C" int itr_scan_train_i2t_prepare(const float *V, const float *E, const int64_t *cap_off, const int32_t *cap_len, const int64_t *h_off,
                                          int64_t Bi, int64_t Bc, int R, int D, float *H, float *vnorm, itr_stream_t stream) {
    ITR_REQUIRE(V && E && cap_off && cap_len && h_off && H && vnorm, "itr_scan_train_i2t_prepare: null pointer");
    ITR_REQUIRE(Bi >= 1 && Bc >= 1 && D > 0, "itr_scan_train_i2t_prepare: bad shape");
    ITR_UNSUPPORTED(R < 1 || R > SI_RMAX, "itr_scan_train_i2t_prepare: 1..%d regions per image are supported, got %d", SI_RMAX, R);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(caption_gram_kernel, dim3((unsigned)Bc), dim3(256), 0, st, E, cap_off, cap_len, D, H, h_off);
    ITR_CHECK_LAUNCH("scan_train_i2t gram");
    hipLaunchKernelGGL(rownorm_i2t_kernel, dim3((unsigned)ceil_div(Bi * R, 4)), dim3(256), 0, st, V, Bi * R, D, vnorm);
    ITR_CHECK_LAUNCH("scan_train_i2t rownorm");
    return ITR_OK;
}

extern "C" int itr_scan_train_i2t_fwd(const float *A, int64_t ldA, const float *H, const int64_t *h_off, const float *vnorm,
                                      const int64_t *cap_off, const int32_t *cap_len, int64_t Bi, int64_t Bc, int64_t n_tok, int R, int D,
                                      int max_len, int norm, int agg, float lambda_softmax, float lambda_lse, float *S, itr_stream_t stream) {
    ITR_REQUIRE(A && H && h_off && vnorm && cap_off && cap_len && S, "itr_scan_train_i2t_fwd: null pointer");
    ITR_REQUIRE(ldA >= n_tok, "itr_scan_train_i2t_fwd: ldA < n_tok");
    int rc = check_i2t("itr_scan_train_i2t_fwd", Bi, Bc, n_tok, R, D, norm, agg, max_len);
    if (rc != ITR_OK) return rc;
    ScanI2TArgs g{A, ldA, H, h_off, vnorm, cap_off, cap_len, Bi, Bc, 0, R, norm, agg, lambda_softmax, lambda_lse, S, nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = as_stream(stream);
    if (R == SC_R) return launch_i2t<SC_R>(scan_train_i2t_fwd_kernel<SC_R, true>, "scan_train_i2t_fwd", g, st);
    if (R < SC_R) return launch_i2t<SC_R>(scan_train_i2t_fwd_kernel<SC_R, false>, "scan_train_i2t_fwd", g, st);
    return launch_i2t<SI_RMAX>(scan_train_i2t_fwd_kernel<SI_RMAX, false>, "scan_train_i2t_fwd", g, st);
}

extern "C" int itr_scan_train_i2t_bwd(const float *A, int64_t ldA, const float *H, const int64_t *h_off, int64_t h_total, const float *vnorm,
                                      const int64_t *cap_off, const int32_t *cap_len, int64_t Bi, int64_t Bc, int64_t n_tok, int R, int D,
                                      int max_len, int norm, int agg, float lambda_softmax, float lambda_lse, const float *dS, float *dA,
                                      float *dH_pairs, float *d_vnorm_pairs, itr_stream_t stream) {
    ITR_REQUIRE(A && H && h_off && vnorm && cap_off && cap_len && dS && dA && dH_pairs && d_vnorm_pairs, "itr_scan_train_i2t_bwd: null pointer");
    ITR_REQUIRE(ldA >= n_tok && h_total >= 1, "itr_scan_train_i2t_bwd: bad leading dimension / Gram size");
    int rc = check_i2t("itr_scan_train_i2t_bwd", Bi, Bc, n_tok, R, D, norm, agg, max_len);
    if (rc != ITR_OK) return rc;
    ScanI2TArgs g{A, ldA, H, h_off, vnorm, cap_off, cap_len, Bi, Bc, h_total, R, norm, agg, lambda_softmax, lambda_lse, nullptr, dS, dA, dH_pairs,
                  d_vnorm_pairs};
    hipStream_t st = as_stream(stream);
    if (R == SC_R) return launch_i2t<SC_R>(scan_train_i2t_bwd_kernel<SC_R, true>, "scan_train_i2t_bwd", g, st);
    if (R < SC_R) return launch_i2t<SC_R>(scan_train_i2t_bwd_kernel<SC_R, false>, "scan_train_i2t_bwd", g, st);
    return launch_i2t<SI_RMAX>(scan_train_i2t_bwd_kernel<SI_RMAX, false>, "scan_train_i2t_bwd", g, st);
}

extern "C" int itr_scan_train_i2t_finish(const float *dH, const int64_t *h_off, const int64_t *cap_off, const int32_t *cap_len, int64_t Bc,
                                         const float *E, const float *V, const float *vnorm, const float *d_vnorm, int64_t Bi, int R, int D,
                                         float *dV, float *dE, itr_stream_t stream) {
    ITR_REQUIRE(dH && h_off && cap_off && cap_len && E && V && vnorm && d_vnorm && dV && dE, "itr_scan_train_i2t_finish: null pointer");
    ITR_REQUIRE(Bi >= 1 && Bc >= 1 && D > 0, "itr_scan_train_i2t_finish: bad shape");
    ITR_UNSUPPORTED(R < 1 || R > SI_RMAX, "itr_scan_train_i2t_finish: 1..%d regions per image are supported, got %d", SI_RMAX, R);
    ITR_UNSUPPORTED(Bi * R > 65535, "itr_scan_train_i2t_finish: at most 65535 region rows per training batch");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(i2t_gram_bwd_kernel, dim3((unsigned)Bc), dim3(256), 0, st, dH, h_off, cap_off, cap_len, E, D, dE);
    ITR_CHECK_LAUNCH("scan_train_i2t gram_bwd");
    hipLaunchKernelGGL(rowscale_add_kernel, dim3((unsigned)ceil_div(D, 256), (unsigned)(Bi * R)), dim3(256), 0, st, V, vnorm, d_vnorm,
                       Bi * R, D, dV);
    ITR_CHECK_LAUNCH("scan_train_i2t vnorm_bwd");
    return ITR_OK;
}
