// SCAN t2i similarity for TRAINING batches: forward that keeps what the backward needs, and the backward
// (xattn_score_t2i + func_attention + cosine_similarity, Objectives.py:329-372, :420-476, :10-15, under autograd).
//
// A training batch is small (128 images x 128 captions), so the layout differs from the 5k x 25k evaluation
// kernel (scan_xattn.hip): the raw dot products of ALL pairs are one plain GEMM
//       A[B_i * 36, n_tok] = V[B_i * 36, D] . E[n_tok, D]^T                      (gemm_nt_kernel)
// and one workgroup per (image, caption) pair does the per-pair arithmetic on its 36 x W block of A in LDS, using the
// same Gram-matrix identity as the evaluation kernel (G_i = V_i V_i^T):
//       b = f(a)                                   f: leaky(0.1) | id          (clipped_* | plain)
//       u[r, w] = b[r, w] / (||b[r, :]||_w + eps)   first norm over the caption's words, per region (l2 forms) | b
//       p[r, w] = softmax_r(lambda_s u[r, w])
//       num_w = sum_r p a      q_w = p^T G p      s_w = num_w / max(||e_w|| sqrt(q_w), 1e-8)
//       S = LSE_w(lambda_lse s_w) / lambda_lse | max | sum | mean
// Backward (dS given): ds_w -> dnum_w, dq_w, d||e_w||;  dp = dnum a + 2 dq (G p);  da += dnum p;  softmax and norm
// backward to da;  outputs per pair:  dA block (disjoint, no atomics), dG partial [36 x 36] (summed over captions by a
// second kernel), d||e_w|| partial (summed over images by itr_colsum).  The D-long contractions of the backward are
// again plain GEMMs done by the caller:  dV = dA E + (dG + dG^T) V,   dE = dA^T V + d||e|| e / ||e||.
#include "scan_common.h"
#include <mutex>
#include <set>
#include <utility>

namespace itr {

constexpr int ST_RMAX = 100;  // regions per image the general instantiations hold (bottom-up features come as 36 fixed or 10..100 adaptive boxes)
constexpr int ST_MAXW = 96;   // words per caption: the longest Flickr30k caption has 82 tokens; 96 keeps the pair in 64 KB of static LDS.
                              // Kernels are instantiated for 64 (3 workgroups per CU: every BASELINE batch) and 96.

struct ScanTrainArgs {
    const float *A;        // [Bi*36, ldA]  raw dot products
    int64_t ldA;           //               (= n_tok)
    const float *G;        // [Bi, 36, 36]
    const float *enorm;    // [n_tok]       ||e_w||
    const int64_t *cap_off;
    const int32_t *cap_len;
    int64_t Bi, Bc;
    int R;                 // regions per image: 36 in every reference configuration (the FIXED instantiations), 1..ST_RMAX otherwise
    int norm, agg;         // norm: 0 clipped_l2norm, 1 l2norm, 2 softmax, 3 no_norm, 4 clipped, 5 l1norm, 6 clipped_l1norm;  agg: 0 LSE, 1 Max, 2 Sum, 3 Mean
    float ls, ll;
    float *S;              // [Bi, Bc]
    // backward only
    const float *dS;       // [Bi, Bc]
    float *dA;             // [Bi*36, ldA]
    float *dGp;            // [Bi, Bc, 36, 36]   per-pair partials
    float *den;            // [Bi, n_tok]        per (image, word) d||e_w||
};

// RMAX: the region count the arrays hold; FIXED: the count IS RMAX (compile-time loop bounds: the reference's 36), otherwise
// g.R <= RMAX at run time (round 3: any region count).
template <int MAXW, int RMAX>
struct PairSmem {
    float a[RMAX][MAXW + 1];   // raw
    float p[RMAX][MAXW + 1];   // attention weights
    float g[RMAX][RMAX + 1];   // Gram
    float gp[RMAX][MAXW + 1];  // G p
    float rn[RMAX];            // 1 / (||b[r,:]|| + eps)
    float rs[RMAX];            // sqrt(sum_w b^2) per region (norm backward)
    float s[MAXW], num[MAXW], q[MAXW], ds[MAXW];
    float red;
};
template <int MAXW, int RMAX>
struct PairBwdSmem {
    PairSmem<MAXW, RMAX> sm;
    float da[RMAX][MAXW + 1];
    float dqs[MAXW];
};

// everything up to s_w; returns with sm.{a,p,g,gp,rn,rs,s,num,q} valid
template <int MAXW, int RMAX, bool FIXED>
__device__ __forceinline__ void pair_forward(const ScanTrainArgs &g, PairSmem<MAXW, RMAX> &sm, int64_t i, int64_t c, int W, int64_t off) {
    const int tid = threadIdx.x;
    const int R = FIXED ? RMAX : g.R;
    for (int idx = tid; idx < R * W; idx += 256) {
        const int r = idx / W, w = idx - r * W;
        sm.a[r][w] = g.A[(i * R + r) * g.ldA + off + w];
    }
    for (int idx = tid; idx < R * R; idx += 256) sm.g[idx / R][idx % R] = g.G[i * R * R + idx];
    __syncthreads();
    // first normalisation along the caption's words, one lane per region row: u[r][w] -> sm.p (Objectives.py:436-457)
    const bool clip = (g.norm == 0 || g.norm == 4 || g.norm == 6), l2 = (g.norm == 0 || g.norm == 1), l1 = (g.norm == 5 || g.norm == 6);
    if (tid < R) {
        const int r = tid;
        float st = 0.f, mxa = -INFINITY;
        for (int w = 0; w < W; ++w) {
            const float b = clip ? leaky(sm.a[r][w]) : sm.a[r][w];
            st += l2 ? b * b : (l1 ? fabsf(b) : 0.f);
            mxa = fmaxf(mxa, sm.a[r][w]);
        }
        if (l2) st = sqrtf(st);
        if (g.norm == 2) {                       // softmax over the words
            st = 0.f;
            for (int w = 0; w < W; ++w) st += expf(sm.a[r][w] - mxa);
        }
        sm.rs[r] = (g.norm == 2) ? mxa : st;     // l2: ||b||, l1: sum |b|, softmax: the row maximum
        const float rn = (l2 || l1) ? 1.f / (st + 1e-8f) : (g.norm == 2 ? 1.f / st : 1.f);
        sm.rn[r] = rn;
        for (int w = 0; w < W; ++w) {
            const float araw = sm.a[r][w];
            sm.p[r][w] = (g.norm == 2) ? expf(araw - mxa) * rn : (clip ? leaky(araw) : araw) * rn;
        }
    }
    __syncthreads();
    if (tid < W) {   // one lane per word: softmax over the regions
        const int w = tid;
        float mx = -INFINITY;
        for (int r = 0; r < R; ++r) {
            const float u = sm.p[r][w] * g.ls;
            sm.p[r][w] = u;
            mx = fmaxf(mx, u);
        }
        float den = 0.f;
        for (int r = 0; r < R; ++r) {
            const float e = expf(sm.p[r][w] - mx);
            sm.p[r][w] = e;
            den += e;
        }
        float num = 0.f;
        for (int r = 0; r < R; ++r) {
            const float pv = sm.p[r][w] / den;
            sm.p[r][w] = pv;
            num += pv * sm.a[r][w];
        }
        sm.num[w] = num;
    }
    __syncthreads();
    // G p for every (region, word) -- one element per thread (round 6; before, the lane of a word walked all R x R products itself:
    // a dozen active lanes of 256 for 1 296 dependent steps, most of the 0.55 + 0.97 ms the two pair kernels took per step)
    for (int idx = tid; idx < R * W; idx += 256) {
        const int r = idx / W, w = idx - r * W;
        float t = 0.f;
        for (int s2 = 0; s2 < R; ++s2) t += sm.g[r][s2] * sm.p[s2][w];
        sm.gp[r][w] = t;
    }
    __syncthreads();
    if (tid < W) {
        const int w = tid;
        float q = 0.f;
        for (int r = 0; r < R; ++r) q += sm.p[r][w] * sm.gp[r][w];
        q = fmaxf(q, 0.f);
        sm.q[w] = q;
        sm.s[w] = sm.num[w] / fmaxf(g.enorm[off + w] * sqrtf(q), 1e-8f);
    }
    __syncthreads();
}

template <int MAXW, int RMAX, bool FIXED>
__global__ __launch_bounds__(256) void scan_train_fwd_kernel(ScanTrainArgs g) {
    extern __shared__ __attribute__((aligned(16))) char pair_smem[];
    PairSmem<MAXW, RMAX> &sm = *reinterpret_cast<PairSmem<MAXW, RMAX> *>(pair_smem);
    const int64_t c = blockIdx.x, i = blockIdx.y;
    const int W = g.cap_len[c];
    const int64_t off = g.cap_off[c];
    pair_forward<MAXW, RMAX, FIXED>(g, sm, i, c, W, off);
    if (threadIdx.x == 0) {
        float r;
        if (g.agg == 0) {
            float mx = -INFINITY;
            for (int w = 0; w < W; ++w) mx = fmaxf(mx, sm.s[w] * g.ll);
            float acc = 0.f;
            for (int w = 0; w < W; ++w) acc += expf(sm.s[w] * g.ll - mx);
            r = (logf(acc) + mx) / g.ll;
        } else if (g.agg == 1) {
            r = -INFINITY;
            for (int w = 0; w < W; ++w) r = fmaxf(r, sm.s[w]);
        } else {
            r = 0.f;
            for (int w = 0; w < W; ++w) r += sm.s[w];
            if (g.agg == 3) r /= (float)W;
        }
        g.S[i * g.Bc + c] = r;
    }
}

template <int MAXW, int RMAX, bool FIXED>
__device__ __forceinline__ void scan_train_bwd_body(const ScanTrainArgs &g) {
    extern __shared__ __attribute__((aligned(16))) char pair_smem[];
    PairBwdSmem<MAXW, RMAX> &bs = *reinterpret_cast<PairBwdSmem<MAXW, RMAX> *>(pair_smem);
    PairSmem<MAXW, RMAX> &sm = bs.sm;
    float(&da)[RMAX][MAXW + 1] = bs.da;
    float(&dqs)[MAXW] = bs.dqs;
    const int R = FIXED ? RMAX : g.R;
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x, i = blockIdx.y;
    const int W = g.cap_len[c];
    const int64_t off = g.cap_off[c];
    pair_forward<MAXW, RMAX, FIXED>(g, sm, i, c, W, off);
    const float dS = g.dS[i * g.Bc + c];
    // ---- aggregation backward: ds_w
    if (tid == 0) {
        if (g.agg == 0) {
            float mx = -INFINITY;
            for (int w = 0; w < W; ++w) mx = fmaxf(mx, sm.s[w] * g.ll);
            float acc = 0.f;
            for (int w = 0; w < W; ++w) acc += expf(sm.s[w] * g.ll - mx);
            for (int w = 0; w < W; ++w) sm.ds[w] = dS * expf(sm.s[w] * g.ll - mx) / acc;
        } else if (g.agg == 1) {
            int best = 0;
            for (int w = 1; w < W; ++w)
                if (sm.s[w] > sm.s[best]) best = w;      // torch.max: the first maximal index receives the gradient
            for (int w = 0; w < W; ++w) sm.ds[w] = (w == best) ? dS : 0.f;
        } else {
            const float k = (g.agg == 3) ? dS / (float)W : dS;
            for (int w = 0; w < W; ++w) sm.ds[w] = k;
        }
    }
    __syncthreads();
    const bool clip = (g.norm == 0 || g.norm == 4 || g.norm == 6), l2 = (g.norm == 0 || g.norm == 1), l1 = (g.norm == 5 || g.norm == 6);
    // ---- per word: cosine backward, attention backward through the softmax -> du (kept in da)
    if (tid < W) {
        const int w = tid;
        const float ew = g.enorm[off + w];
        const float sq = sqrtf(sm.q[w]);
        const float den = ew * sq;
        float dnum = 0.f, dq = 0.f, dew = 0.f;
        if (den > 1e-8f) {   // the clamp of cosine_similarity is inactive
            dnum = sm.ds[w] / den;
            const float dden = -sm.ds[w] * sm.num[w] / (den * den);
            dew = dden * sq;
            dq = sq > 0.f ? dden * ew / (2.f * sq) : 0.f;
        } else {
            dnum = sm.ds[w] / 1e-8f;
        }
        dqs[w] = dq;
        g.den[i * g.ldA + off + w] = dew;
        // dp = dnum a + 2 dq (G p);  du = ls * p (dp - sum_r p dp)
        float dot = 0.f;
        for (int r = 0; r < R; ++r) {
            const float dp = dnum * sm.a[r][w] + 2.f * dq * sm.gp[r][w];
            da[r][w] = dp;
            dot += sm.p[r][w] * dp;
        }
        for (int r = 0; r < R; ++r) da[r][w] = g.ls * sm.p[r][w] * (da[r][w] - dot);   // = du[r][w]
        sm.num[w] = dnum;   // reuse: dnum per word
    }
    __syncthreads();
    // ---- per region: first-norm backward  u = b * rn,  rn = 1 / (sqrt(sum b^2) + eps)
    if (tid < R) {
        const int r = tid;
        const float rn = sm.rn[r], rt = sm.rs[r];
        // dot = sum_w du u-like term of the normalisation's Jacobian
        float dot = 0.f;
        for (int w = 0; w < W; ++w) {
            const float araw = sm.a[r][w];
            const float b = clip ? leaky(araw) : araw;
            if (l2 || l1) dot += da[r][w] * b;
            else if (g.norm == 2) dot += da[r][w] * expf(araw - rt) * rn;      // du . u
        }
        for (int w = 0; w < W; ++w) {
            const float araw = sm.a[r][w];
            const float b = clip ? leaky(araw) : araw;
            float db;
            if (l2) db = da[r][w] * rn - (rt > 0.f ? b * dot * rn * rn / rt : 0.f);       // u = b / (||b|| + eps)
            else if (l1) db = da[r][w] * rn - (b > 0.f ? 1.f : (b < 0.f ? -1.f : 0.f)) * dot * rn * rn;   // u = b / (sum |b| + eps)
            else if (g.norm == 2) { const float u = expf(araw - rt) * rn; db = u * (da[r][w] - dot); }    // u = softmax_w(a)
            else db = da[r][w];
            if (clip) db *= (araw > 0.f) ? 1.f : 0.1f;    // LeakyReLU(0.1); slope at exactly 0 = 0.1 like torch
            da[r][w] = db + sm.num[w] * sm.p[r][w];        // + the direct path of num = sum p a
        }
    }
    __syncthreads();
    for (int idx = tid; idx < R * W; idx += 256) {
        const int r = idx / W, w = idx - r * W;
        g.dA[(i * R + r) * g.ldA + off + w] = da[r][w];
    }
    // ---- dG partial of this pair:  sum_w dq_w p_w p_w^T
    float *dgp = g.dGp + (i * g.Bc + c) * (R * R);
    for (int idx = tid; idx < R * R; idx += 256) {
        const int r = idx / R, s2 = idx - r * R;
        float acc = 0.f;
        for (int w = 0; w < W; ++w) acc += dqs[w] * sm.p[r][w] * sm.p[s2][w];
        dgp[idx] = acc;
    }
}
template <int MAXW, int RMAX, bool FIXED>
__global__ __launch_bounds__(256) void scan_train_bwd_kernel(ScanTrainArgs g) { scan_train_bwd_body<MAXW, RMAX, FIXED>(g); }
// the 32-word instantiation: 25 KB of LDS per pair admit six resident workgroups, 110 registers four -- capped at 80 (a few spills) it
// measured 3.56 -> 3.51 ms on the SCAN step; the larger instantiations are LDS-limited and keep their registers
template <int MAXW, int RMAX, bool FIXED>
__global__ __launch_bounds__(256, 6) void scan_train_bwd_dense_kernel(ScanTrainArgs g) { scan_train_bwd_body<MAXW, RMAX, FIXED>(g); }

// dG[i] = sum_c dGp[i, c]  (summed into the slot of caption 0: a thread touches only its own element of every slot), then
// dV[i] += (dG + dG^T) V[i]   (G = V V^T).  Round 6: two launches of (elements / features) x images workgroups -- the one-workgroup-per-
// image form left half of the chip idle on a 128-image batch (0.34 ms of a 5 ms SCAN step).
__global__ __launch_bounds__(256) void scan_train_dg_reduce_kernel(float *__restrict__ dGp, int64_t Bc, int RR) {
    const int64_t i = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= RR) return;
    // (16 loads in flight, four running sums: one load per loop trip was 128 L2 round trips per thread, 55 us)
    const float *src = dGp + (i * Bc) * RR + idx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int64_t c = 0;
    for (; c + 16 <= Bc; c += 16) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = src[(c + q) * RR];
#pragma unroll
        for (int q = 0; q < 16; q += 4) { a0 += v[q]; a1 += v[q + 1]; a2 += v[q + 2]; a3 += v[q + 3]; }
    }
    for (; c < Bc; ++c) a0 += src[c * RR];
    dGp[(i * Bc) * RR + idx] = (a0 + a1) + (a2 + a3);
}
template <int RMAX, bool FIXED>
__global__ __launch_bounds__(256) void scan_train_gram_bwd_kernel(const float *__restrict__ dGp, int64_t Bc, const float *__restrict__ V, int D,
                                                                  float *__restrict__ dV, int R_) {
    __shared__ float dg[RMAX][RMAX + 1];
    const int R = FIXED ? RMAX : R_;
    const int64_t i = blockIdx.y;
    for (int idx = threadIdx.x; idx < R * R; idx += 256) dg[idx / R][idx % R] = dGp[(i * Bc) * (R * R) + idx];
    __syncthreads();
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    if constexpr (FIXED) {
        float v[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) v[r] = V[(i * RMAX + r) * D + d];
        for (int r = 0; r < RMAX; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < RMAX; ++s2) acc += (dg[r][s2] + dg[s2][r]) * v[s2];
            dV[(i * RMAX + r) * D + d] += acc;
        }
    } else {
        for (int r = 0; r < R; ++r) {
            float acc = 0.f;
            for (int s2 = 0; s2 < R; ++s2) acc += (dg[r][s2] + dg[s2][r]) * V[(i * R + s2) * D + d];
            dV[(i * R + r) * D + d] += acc;
        }
    }
}

// G[i] = V[i] V[i]^T  (36 x 36, one workgroup per image) and ||e_w||
__global__ __launch_bounds__(256) void scan_train_gram_kernel(const float *__restrict__ V, int D, float *__restrict__ G, int R) {
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int idx = wave; idx < R * R; idx += 4) {
        const int r = idx / R, s2 = idx % R;
        const float *a = V + (i * R + r) * D, *b = V + (i * R + s2) * D;
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) acc += a[d] * b[d];
        acc = wave_sum(acc);
        if (lane == 0) G[i * R * R + idx] = acc;
    }
}
__global__ __launch_bounds__(256) void rownorm_train_kernel(const float *__restrict__ x, int64_t rows, int D, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += x[row * D + d] * x[row * D + d];
    s = wave_sum(s);
    if (lane == 0) out[row] = sqrtf(s);
}
// dE[w, :] += den[w] * e_w / ||e_w||
__global__ __launch_bounds__(256) void enorm_bwd_kernel(const float *__restrict__ E, const float *__restrict__ enorm, const float *__restrict__ den,
                                                        int64_t rows, int D, float *__restrict__ dE) {
    const int64_t row = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows || d >= D) return;
    const float n = enorm[row];
    if (n > 0.f) dE[row * D + d] += den[row] * E[row * D + d] / n;
}

// LDS-tiled Gram kernel of the evaluation path (scan_xattn.hip): G[n] = X_n X_n^T
__global__ void gram_kernel(const float *__restrict__ X, const int64_t *__restrict__ row_off, const int32_t *__restrict__ row_cnt, int fixed_rows,
                            int D, float *__restrict__ G, const int64_t *__restrict__ g_off, int upper2);
__global__ void gram_mfma_kernel(const float *__restrict__ X, int rows, int D, float *__restrict__ G, int upper2);

static int check_train_args(const char *who, int64_t Bi, int64_t Bc, int64_t n_tok, int R, int D, int norm, int agg, int max_len) {
    ITR_REQUIRE(Bi >= 1 && Bc >= 1 && n_tok >= 1 && D > 0, "%s: bad shape", who);
    ITR_UNSUPPORTED(R < 1 || R > ST_RMAX, "%s: 1..%d regions per image are supported, got %d", who, ST_RMAX, R);
    ITR_UNSUPPORTED(max_len > ST_MAXW, "%s: captions of at most %d words are supported, got %d", who, ST_MAXW, max_len);
    ITR_REQUIRE(norm >= 0 && norm <= 6, "%s: unknown first norm %d", who, norm);
    ITR_REQUIRE(agg >= 0 && agg <= 3, "%s: unknown aggregation %d", who, agg);
    ITR_UNSUPPORTED(Bi > 65535, "%s: at most 65535 images per training batch", who);
    return ITR_OK;
}

// One launch of a pair kernel instantiation with its LDS block as dynamic shared memory (the general-R ones need up to 159 KB:
// more than the 64 KB a static declaration may have).  The attribute is set once per kernel and device.
int allow_dynamic_lds(const void *kernel, size_t bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    ITR_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (!done.count({kernel, dev})) {
        ITR_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        done.insert({kernel, dev});
    }
    return ITR_OK;
}
template <typename Smem>
static int launch_pair(void (*kernel)(ScanTrainArgs), const char *what, const ScanTrainArgs &g, hipStream_t st) {
    static_assert(sizeof(Smem) <= 160 * 1024, "a pair's LDS block must fit one CU");
    const int rc = allow_dynamic_lds(reinterpret_cast<const void *>(kernel), sizeof(Smem));
    if (rc != ITR_OK) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)g.Bc, (unsigned)g.Bi), dim3(256), sizeof(Smem), st, g);
    ITR_CHECK_LAUNCH(what);
    return ITR_OK;
}

}  // namespace itr

using namespace itr;

extern "C" int itr_scan_train_prepare(const float *V, const float *E, int64_t Bi, int64_t n_tok, int R, int D, float *G, float *enorm,
                                      itr_stream_t stream) {
    ITR_REQUIRE(V && E && G && enorm, "itr_scan_train_prepare: null pointer");
    ITR_REQUIRE(Bi >= 1 && n_tok >= 1 && D > 0, "itr_scan_train_prepare: bad shape");
    ITR_UNSUPPORTED(R < 1 || R > ST_RMAX, "itr_scan_train_prepare: 1..%d regions per image are supported, got %d", ST_RMAX, R);
    hipStream_t st = as_stream(stream);
    if (R <= 48 && D % 4 == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0)      // (the evaluation path's matrix-core Gram kernel, scan_xattn.hip)
        hipLaunchKernelGGL(gram_mfma_kernel, dim3((unsigned)Bi), dim3(256), 0, st, V, R, D, G, 0);
    else if (R == SC_R)
        hipLaunchKernelGGL(gram_kernel, dim3((unsigned)Bi), dim3(256), 0, st, V, (const int64_t *)nullptr, (const int32_t *)nullptr, SC_R, D, G,
                           (const int64_t *)nullptr, 0);
    else
        hipLaunchKernelGGL(scan_train_gram_kernel, dim3((unsigned)Bi), dim3(256), 0, st, V, D, G, R);
    ITR_CHECK_LAUNCH("scan_train_gram");
    hipLaunchKernelGGL(rownorm_train_kernel, dim3((unsigned)ceil_div(n_tok, 4)), dim3(256), 0, st, E, n_tok, D, enorm);
    ITR_CHECK_LAUNCH("scan_train_rownorm");
    return ITR_OK;
}

extern "C" int itr_scan_train_fwd(const float *A, int64_t ldA, const float *G, const float *enorm, const int64_t *cap_off, const int32_t *cap_len,
                                  int64_t Bi, int64_t Bc, int64_t n_tok, int R, int D, int max_len, int norm, int agg, float lambda_softmax,
                                  float lambda_lse, float *S, itr_stream_t stream) {
    ITR_REQUIRE(A && G && enorm && cap_off && cap_len && S, "itr_scan_train_fwd: null pointer");
    ITR_REQUIRE(ldA >= n_tok, "itr_scan_train_fwd: ldA < n_tok");
    int rc = check_train_args("itr_scan_train_fwd", Bi, Bc, n_tok, R, D, norm, agg, max_len);
    if (rc != ITR_OK) return rc;
    ScanTrainArgs g{A, ldA, G, enorm, cap_off, cap_len, Bi, Bc, R, norm, agg, lambda_softmax, lambda_lse, S, nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = as_stream(stream);
    const bool w64 = max_len <= 64;
    // captions of at most 32 words (every COCO / F30k training batch but a handful): 20 KB of LDS per pair instead of 35 -- eight resident
    // pairs per CU instead of four; a pair's life is a chain of short, barrier-separated phases, so residency is what hides it
    if (R == SC_R && max_len <= 32) return launch_pair<PairSmem<32, SC_R>>(scan_train_fwd_kernel<32, SC_R, true>, "scan_train_fwd", g, st);
    if (R == SC_R)
        return w64 ? launch_pair<PairSmem<64, SC_R>>(scan_train_fwd_kernel<64, SC_R, true>, "scan_train_fwd", g, st)
                   : launch_pair<PairSmem<ST_MAXW, SC_R>>(scan_train_fwd_kernel<ST_MAXW, SC_R, true>, "scan_train_fwd", g, st);
    if (R < SC_R)
        return w64 ? launch_pair<PairSmem<64, SC_R>>(scan_train_fwd_kernel<64, SC_R, false>, "scan_train_fwd", g, st)
                   : launch_pair<PairSmem<ST_MAXW, SC_R>>(scan_train_fwd_kernel<ST_MAXW, SC_R, false>, "scan_train_fwd", g, st);
    return w64 ? launch_pair<PairSmem<64, ST_RMAX>>(scan_train_fwd_kernel<64, ST_RMAX, false>, "scan_train_fwd", g, st)
               : launch_pair<PairSmem<ST_MAXW, ST_RMAX>>(scan_train_fwd_kernel<ST_MAXW, ST_RMAX, false>, "scan_train_fwd", g, st);
}

extern "C" int itr_scan_train_bwd(const float *A, int64_t ldA, const float *G, const float *enorm, const int64_t *cap_off, const int32_t *cap_len,
                                  int64_t Bi, int64_t Bc, int64_t n_tok, int R, int D, int max_len, int norm, int agg, float lambda_softmax,
                                  float lambda_lse, const float *dS, float *dA, float *dG_pairs, float *d_enorm_pairs, itr_stream_t stream) {
    ITR_REQUIRE(A && G && enorm && cap_off && cap_len && dS && dA && dG_pairs && d_enorm_pairs, "itr_scan_train_bwd: null pointer");
    ITR_REQUIRE(ldA >= n_tok, "itr_scan_train_bwd: ldA < n_tok");
    int rc = check_train_args("itr_scan_train_bwd", Bi, Bc, n_tok, R, D, norm, agg, max_len);
    if (rc != ITR_OK) return rc;
    // the backward keeps one more regions x words block: more than 36 regions AND more than 64 words do not fit a CU's 160 KB of LDS
    ITR_UNSUPPORTED(R > SC_R && max_len > 64, "itr_scan_train_bwd: more than %d regions with captions of more than 64 words (got %d, %d)", SC_R, R, max_len);
    ScanTrainArgs g{A, ldA, G, enorm, cap_off, cap_len, Bi, Bc, R, norm, agg, lambda_softmax, lambda_lse, nullptr, dS, dA, dG_pairs, d_enorm_pairs};
    hipStream_t st = as_stream(stream);
    const bool w64 = max_len <= 64;
    if (R == SC_R && max_len <= 32) return launch_pair<PairBwdSmem<32, SC_R>>(scan_train_bwd_dense_kernel<32, SC_R, true>, "scan_train_bwd", g, st);
    if (R == SC_R)
        return w64 ? launch_pair<PairBwdSmem<64, SC_R>>(scan_train_bwd_kernel<64, SC_R, true>, "scan_train_bwd", g, st)
                   : launch_pair<PairBwdSmem<ST_MAXW, SC_R>>(scan_train_bwd_kernel<ST_MAXW, SC_R, true>, "scan_train_bwd", g, st);
    if (R < SC_R)
        return w64 ? launch_pair<PairBwdSmem<64, SC_R>>(scan_train_bwd_kernel<64, SC_R, false>, "scan_train_bwd", g, st)
                   : launch_pair<PairBwdSmem<ST_MAXW, SC_R>>(scan_train_bwd_kernel<ST_MAXW, SC_R, false>, "scan_train_bwd", g, st);
    return launch_pair<PairBwdSmem<64, ST_RMAX>>(scan_train_bwd_kernel<64, ST_RMAX, false>, "scan_train_bwd", g, st);
}

extern "C" int itr_scan_train_finish(float *dG_pairs, int64_t Bi, int64_t Bc, const float *V, const float *E, const float *enorm,
                                     const float *d_enorm, int64_t n_tok, int R, int D, float *dV, float *dE, itr_stream_t stream) {
    ITR_REQUIRE(dG_pairs && V && E && enorm && d_enorm && dV && dE, "itr_scan_train_finish: null pointer");
    ITR_REQUIRE(Bi >= 1 && Bc >= 1 && n_tok >= 1 && D > 0, "itr_scan_train_finish: bad shape");
    ITR_UNSUPPORTED(R < 1 || R > ST_RMAX, "itr_scan_train_finish: 1..%d regions per image are supported, got %d", ST_RMAX, R);
    ITR_UNSUPPORTED(n_tok > 65535, "itr_scan_train_finish: at most 65535 words per training batch");
    hipStream_t st = as_stream(stream);
    ITR_UNSUPPORTED(Bi > 65535, "itr_scan_train_finish: at most 65535 images per training batch");
    hipLaunchKernelGGL(scan_train_dg_reduce_kernel, dim3((unsigned)ceil_div(R * R, 256), (unsigned)Bi), dim3(256), 0, st, dG_pairs, Bc,
                       R * R);
    ITR_CHECK_LAUNCH("scan_train_dg_reduce");
    const dim3 gb((unsigned)ceil_div(D, 256), (unsigned)Bi);
    if (R == SC_R) hipLaunchKernelGGL((scan_train_gram_bwd_kernel<SC_R, true>), gb, dim3(256), 0, st, dG_pairs, Bc, V, D, dV, R);
    else if (R < SC_R) hipLaunchKernelGGL((scan_train_gram_bwd_kernel<SC_R, false>), gb, dim3(256), 0, st, dG_pairs, Bc, V, D, dV, R);
    else hipLaunchKernelGGL((scan_train_gram_bwd_kernel<ST_RMAX, false>), gb, dim3(256), 0, st, dG_pairs, Bc, V, D, dV, R);
    ITR_CHECK_LAUNCH("scan_train_gram_bwd");
    hipLaunchKernelGGL(enorm_bwd_kernel, dim3((unsigned)ceil_div(D, 256), (unsigned)n_tok), dim3(256), 0, st, E, enorm, d_enorm, n_tok, D, dE);
    ITR_CHECK_LAUNCH("scan_train_enorm_bwd");
    return ITR_OK;
}
