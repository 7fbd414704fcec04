// SGRAF similarity: EncoderSimilarity.forward + VisualSA / TextSA / SCAN_attention / AttentionFiltration /
// GraphReasoning (itr/modalmodule/Fusionmodule.py:373-664), eval mode (BatchNorm running statistics, no dropout).
//
// Structure (all heavy arithmetic on the fp32 matrix core), per block of 16 images (4 on the unfused path):
//   1. SCAN kernel (scan_xattn.hip) in emit mode  -> attention weights P[i, word, 36] and 1/(||ctx||+eps)
//      (SCAN_attention :632-664 is SCAN t2i with clipped_l2norm and smooth = 9)
//   2+3. sim_loc = l2norm(W_loc (l2norm(P V) - E)^2 + b)   fused, (ctx - E)^2 stays on chip: sgraf_loc.hip (:425-427)
//        (sim_dim != 256: per image GEMM [words,36]x[36,D] with the squared-difference epilogue, GEMM [., D]x[D, S], l2norm)
//   4. sim_glo = l2norm(W_glo (img_glo - cap_glo)^2 + b)  -> elementwise + GEMM + row l2norm          (:429-430)
//   5. SAF (:615-619): one pair kernel;  SGR x sgr_step (:581-587): ONE folded query projection per node
//      (q' = (Wk^T Wq) x + Wk^T bq, see below), MFMA pair kernel softmax(q' X^T) X, graph GEMM; sigmoid(sim_eval_w) (:443-444)
// The global nodes img_glo / cap_glo (VisualSA :491-507, TextSA :543-559) are computed once per call.
#include "scan_common.h"
#include <stdlib.h>

namespace itr {

int gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc,
            int64_t M, int64_t N, int64_t K, int act, hipStream_t st);
int gemm_nt_sqdiff(const float *A, int64_t lda, const float *B, int64_t ldb, const float *rowscale, const float *Z,
                   int64_t ldz, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t st);
int norm_rows(const float *x, float *y, int64_t rows, int dim, float eps, int kind, int take_abs, hipStream_t st);
int scan_prepare_impl(const float *img, const float *words, const int64_t *cap_off, const int32_t *cap_len,
                      const int32_t *tile_begin_dev, const int32_t *cap_order_dev, int64_t n_tiles, int64_t Ni,
                      int64_t Nc, int64_t n_rows, int R, int D, int mode, void *workspace, size_t workspace_bytes,
                      int32_t *cap_col, itr_stream_t stream);
int scan_scores_impl(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows, int R, int D,
                     int mode, int norm, int agg, float lambda_softmax, float lambda_lse, float *S, int64_t ldS,
                     void *workspace, size_t workspace_bytes, float *emit_p, float *emit_cn, int64_t img_index0,
                     int64_t img_count, itr_stream_t stream);

int sgraf_loc_fused(const float *P, const float *cn, const float *img, const float *wtiled, const float *W, const float *bias,
                    float *X, int64_t nb, int64_t n_tiles, int D, hipStream_t st);
// sgr_fused.hip: all graph-reasoning steps of a group of captions in one workgroup
size_t sgr_fused_workspace_bytes(int64_t n_groups, int64_t n_caps, int sgr_step);
int sgr_fused_prepare(const int32_t *grp_begin, const int32_t *grp_order, int64_t n_groups, int64_t n_caps, const int32_t *cap_len,
                      const int32_t *cap_col, const float *const *wq, const float *const *wg, int sgr_step, void *ws, int *bad_flag, hipStream_t st);
int sgr_fused_scores(const float *xloc, const float *xglo, void *ws, int64_t n_groups, int64_t n_caps, int64_t nb, int64_t Nc, int64_t ncols,
                     const float *const *vq, const float *const *bg, int sgr_step, float *y0, bool persistent_walk, hipStream_t st);
int sgr_fused_finish(void *ws, int64_t n_groups, int64_t n_caps, int sgr_step, int64_t Ni, float *S, int64_t ldS, hipStream_t st);

constexpr float BN_EPS = 1e-5f;

// y = tanh(bn(x)) in place; the BN channel is (row % period) when by_row, else the column (eval statistics)
__global__ __launch_bounds__(256) void bn_tanh_kernel(float *__restrict__ x, int64_t rows, int cols, int by_row, int period,
                                                      const float *__restrict__ w, const float *__restrict__ b,
                                                      const float *__restrict__ mean, const float *__restrict__ var) {
    const int64_t row = blockIdx.y;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows || col >= cols) return;
    const int ch = by_row ? (int)(row % period) : col;
    const float v = (x[row * cols + col] - mean[ch]) / sqrtf(var[ch] + BN_EPS) * w[ch] + b[ch];
    x[row * cols + col] = tanhf(v);
}

// out[g, :] = mean over the rows of group g (ragged: rows off[g] .. off[g]+len[g]-1)
__global__ __launch_bounds__(256) void seg_mean_kernel(const float *__restrict__ x, const int64_t *__restrict__ off,
                                                       const int32_t *__restrict__ len, int D, float *__restrict__ out) {
    const int64_t gidx = blockIdx.x;
    const int n = len[gidx];
    const float *p = x + off[gidx] * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float s = 0.f;
        for (int r = 0; r < n; ++r) s += p[(int64_t)r * D + d];
        out[gidx * D + d] = s / (float)n;
    }
}

// Self-attention pooling of VisualSA / TextSA (Fusionmodule.py:499-507 / :551-559):
//   w_r = softmax_r( sum_d l_emb[r,d] * g_emb[d] * wc[d] + bc );  out = l2norm( sum_r w_r * local[r,:] )
// one workgroup per group (image: 36 rows, caption: len rows <= 64)
__global__ __launch_bounds__(256) void sa_pool_kernel(const float *__restrict__ local, const float *__restrict__ l_emb,
                                                      const float *__restrict__ g_emb, const float *__restrict__ wc,
                                                      const float *__restrict__ bc, const int64_t *__restrict__ off,
                                                      const int32_t *__restrict__ len, int fixed_rows, int D,
                                                      float *__restrict__ out) {
    __shared__ float s_w[64];
    __shared__ float s_red[4];
    const int64_t gidx = blockIdx.x;
    const int n = len ? len[gidx] : fixed_rows;
    const int64_t r0 = off ? off[gidx] : gidx * fixed_rows;
    const float *ge = g_emb + gidx * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < n; r += 4) {
        const float *le = l_emb + (r0 + r) * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += le[d] * ge[d] * wc[d];
        s = wave_sum(s);
        if (lane == 0) s_w[r] = s + bc[0];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = -INFINITY;
        for (int r = 0; r < n; ++r) mx = fmaxf(mx, s_w[r]);
        float den = 0.f;
        for (int r = 0; r < n; ++r) { s_w[r] = expf(s_w[r] - mx); den += s_w[r]; }
        for (int r = 0; r < n; ++r) s_w[r] /= den;
    }
    __syncthreads();
    float ss = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = 0.f;
        for (int r = 0; r < n; ++r) v += s_w[r] * local[(r0 + r) * D + d];
        out[gidx * D + d] = v;
        ss += v * v;
    }
    ss = wave_sum(ss);
    if (lane == 0) s_red[wave] = ss;
    __syncthreads();
    const float nrm = sqrtf(s_red[0] + s_red[1] + s_red[2] + s_red[3]) + 1e-8f;
    for (int d = threadIdx.x; d < D; d += 256) out[gidx * D + d] /= nrm;
}

// imgT[i, d, r] = img[i, r, d]  (B operand of the context GEMM, K = 36 contiguous)
__global__ __launch_bounds__(256) void transpose_img_kernel(const float *__restrict__ img, int R, int D, float *__restrict__ out) {
    __shared__ float t[36][65];
    const int64_t i = blockIdx.y;
    const int d0 = blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < R * 64; idx += 256) {
        const int r = idx / 64, d = idx % 64;
        t[r][d] = (d0 + d < D) ? img[(i * R + r) * D + d0 + d] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * R; idx += 256) {
        const int d = idx / R, r = idx % R;
        if (d0 + d < D) out[(i * D + d0 + d) * R + r] = t[r][d];
    }
}

// out[c, r] = in[r, c] for the small (sim_dim x sim_dim) weight matrices; runs 2 x sgr_step times per call
__global__ __launch_bounds__(256) void transpose_small_kernel(const float *__restrict__ in, int rows, int cols, float *__restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < rows * cols) { const int r = idx / cols, c = idx % cols; out[c * rows + r] = in[idx]; }
}

// Aglo[(ii, c), d] = (img_glo[ii, d] - cap_glo[c, d])^2
__global__ __launch_bounds__(256) void glo_sqdiff_kernel(const float *__restrict__ img_glo, const float *__restrict__ cap_glo,
                                                         int64_t Nc, int D, float *__restrict__ out) {
    const int64_t c = blockIdx.x, ii = blockIdx.y;
    const float *a = img_glo + ii * D, *b = cap_glo + c * D;
    float *o = out + (ii * Nc + c) * D;
    for (int d = threadIdx.x; d < D; d += 256) { const float v = a[d] - b[d]; o[d] = v * v; }
}

// The global similarity node l2norm(W_glo (img_glo - cap_glo)^2 + b) through the LOCAL-node kernel (round 3): per image a "region
// set" whose row 0 is img_glo and whose other rows are zero, attention weights that select row 0 and a context norm of 1 make the
// kernel's context exactly img_glo; the captions' global vectors are its "words" (tiles of 64 captions).
__global__ __launch_bounds__(256) void glo_onehot_kernel(float *__restrict__ P, float *__restrict__ cn, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * SC_R) return;
    P[i] = (i % SC_R == 0) ? 1.f : 0.f;
    if (i < rows) cn[i] = 1.f;
}

struct PairArgs {
    const float *xglo, *xloc;      // [nb*ldg, S], [nb*ncols, S]
    const int32_t *cap_col, *cap_len;
    int64_t Nc, ncols;
    int S;
    int64_t ldg;                   // rows per image in xglo: Nc, or Nc rounded up to whole 64-row tiles (global nodes from the local-node kernel)
};

__device__ __forceinline__ const float *node_row(const PairArgs &p, const float *glo, const float *loc, int64_t ii,
                                                 int64_t c, int col0, int n) {
    return n == 0 ? glo + (ii * p.ldg + c) * p.S : loc + (ii * p.ncols + col0 + n - 1) * p.S;
}

// AttentionFiltration + final score (Fusionmodule.py:615-619, :443-444); one wave per (image, caption) pair
__global__ __launch_bounds__(256) void saf_pair_kernel(PairArgs p, const float *__restrict__ saf_w, const float *__restrict__ saf_b,
                                                       const float *__restrict__ bn_w, const float *__restrict__ bn_b,
                                                       const float *__restrict__ bn_mean, const float *__restrict__ bn_var,
                                                       const float *__restrict__ eval_w, const float *__restrict__ eval_b,
                                                       int64_t npairs, float *__restrict__ S, int64_t ldS, int64_t img_index0) {
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= npairs) return;
    const int lane = threadIdx.x & 63;
    const int64_t ii = pair / p.Nc, c = pair % p.Nc;
    const int nn = p.cap_len[c] + 1, col0 = p.cap_col[c];
    const float bscale = bn_w[0] / sqrtf(bn_var[0] + BN_EPS);
    float asum = 0.f;
    if (p.S == 256) {
        // The configured sim_dim: one 16-byte load per lane and node row (lane l holds features 4l .. 4l+3), four node rows in
        // flight per trip -- the kernel is a chain of (load, wave reduction, sigmoid) per node and waits on memory otherwise.
        const float4 wv = *reinterpret_cast<const float4 *>(saf_w + 4 * lane);
        const float sb = saf_b[0], bm = bn_mean[0], bb = bn_b[0];
        float4 acc = float4{0.f, 0.f, 0.f, 0.f};
        for (int n0 = 0; n0 < nn; n0 += 4) {
            float4 xr[4];
            float sr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = n0 + k < nn ? n0 + k : nn - 1;            // the tail re-reads the last node; masked below
                xr[k] = *reinterpret_cast<const float4 *>(node_row(p, p.xglo, p.xloc, ii, c, col0, n) + 4 * lane);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) sr[k] = xr[k].x * wv.x + xr[k].y * wv.y + xr[k].z * wv.z + xr[k].w * wv.w;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int k = 0; k < 4; ++k) sr[k] += __shfl_xor(sr[k], o, 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = 1.f / (1.f + expf(-(((sr[k] + sb) - bm) * bscale + bb)));
                a = n0 + k < nn ? a : 0.f;
                asum += fabsf(a);
                acc.x += a * xr[k].x; acc.y += a * xr[k].y; acc.z += a * xr[k].z; acc.w += a * xr[k].w;
            }
        }
        const float inv = 1.f / (asum + 1e-8f);   // l1norm
        acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
        const float4 ev = *reinterpret_cast<const float4 *>(eval_w + 4 * lane);
        float ss = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
        float dot = acc.x * ev.x + acc.y * ev.y + acc.z * ev.z + acc.w * ev.w;
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float score = dot / (sqrtf(ss) + 1e-8f) + eval_b[0];     // eval_w . l2norm(vec) + b
        if (lane == 0) S[(img_index0 + ii) * ldS + c] = 1.f / (1.f + expf(-score));
        return;
    }
    // ONE pass over the node rows (they are the HBM traffic of this kernel): the l1 normalisation of the attention
    // weights is linear, so the un-normalised aggregate sum_n a_n x_n is divided by sum_n |a_n| at the end.
    // This lane's columns: d = lane, lane + 64, ... (S <= 1024)
    float vec[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) vec[u] = 0.f;
    for (int n = 0; n < nn; ++n) {
        const float *x = node_row(p, p.xglo, p.xloc, ii, c, col0, n);
        float xv[16];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int d = lane + 64 * u;
            xv[u] = d < p.S ? x[d] : 0.f;
            s += d < p.S ? xv[u] * saf_w[d] : 0.f;
        }
        s = wave_sum(s) + saf_b[0];
        const float a = 1.f / (1.f + expf(-((s - bn_mean[0]) * bscale + bn_b[0])));
        asum += fabsf(a);
#pragma unroll
        for (int u = 0; u < 16; ++u) vec[u] += a * xv[u];
    }
    const float inv = 1.f / (asum + 1e-8f);   // l1norm
#pragma unroll
    for (int u = 0; u < 16; ++u) vec[u] *= inv;
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int d = lane + 64 * u; if (d < p.S) { ss += vec[u] * vec[u]; dot += vec[u] * eval_w[d]; } }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float score = dot / (sqrtf(ss) + 1e-8f) + eval_b[0];     // eval_w . l2norm(vec) + b
    if (lane == 0) S[(img_index0 + ii) * ldS + c] = 1.f / (1.f + expf(-score));
}

// One GraphReasoning step per pair (Fusionmodule.py:581-587): edge = softmax(Q K^T), Y = edge X, on the matrix
// core.  One wave per (image, caption) pair; NT = ceil((W+1)/16) node tiles.
//   E  = Q K^T   : A / B fragments straight from L2 (lane (i, g) reads row i, floats 16u+4g..+3), K dim = S
//   P  = softmax over the key axis in the accumulator layout (16-lane shuffles), parked in wave-private LDS
//   Y  = P X     : A = P from LDS (one ds_read_b128 per tile), B = X rows from L2, written row by row
template <int NT>
__device__ __forceinline__ void sgr_pair_body(const PairArgs &p, const float *qglo, const float *qloc, const float *kglo,
                                              const float *kloc, float *yglo, float *yloc, int64_t ii, int64_t c, int col0,
                                              int nn, int glo_only, float *pl /* [NT*16][NT*16+4] wave-private */) {
    const int lane = threadIdx.x & 63, fi = lane & 15, fg = lane >> 4;
    const int S = p.S;
    constexpr int LDP = NT * 16 + 4;
    const int TI = glo_only ? 1 : NT;        // the last step only needs node 0 (row tile 0)
    // ---- E = Q K^T
    f32x4 e[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) e[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *qrow[NT], *krow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int n = t * 16 + fi;
        n = n < nn ? n : nn - 1;             // rows past the graph re-read the last node; masked below
        // the last step only needs the query of node 0 (rows 1..15 of its tile are computed and dropped): every lane
        // reads the global node's q', and the local q' projection of that step is never run (itr_sgraf_scores)
        qrow[t] = node_row(p, qglo, qloc, ii, c, col0, glo_only ? 0 : n);
        krow[t] = node_row(p, kglo, kloc, ii, c, col0, n);
    }
    for (int u = 0; u < S / 16; ++u) {
        float4 qf[NT], kf[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < TI) qf[t] = *reinterpret_cast<const float4 *>(qrow[t] + 16 * u + 4 * fg);
            kf[t] = *reinterpret_cast<const float4 *>(krow[t] + 16 * u + 4 * fg);
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
            if (a < TI) {
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    e[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[a].x, kf[b].x, e[a][b], 0, 0, 0);
                    e[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[a].y, kf[b].y, e[a][b], 0, 0, 0);
                    e[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[a].z, kf[b].z, e[a][b], 0, 0, 0);
                    e[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[a].w, kf[b].w, e[a][b], 0, 0, 0);
                }
            }
    }
    // ---- softmax over keys (columns): this lane holds, per tile, column fi of rows 4fg + r
#pragma unroll
    for (int a = 0; a < NT; ++a)
        if (a < TI) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float mx = -INFINITY;
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    if (b * 16 + fi >= nn) e[a][b][r] = -INFINITY;
                    mx = fmaxf(mx, e[a][b][r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 4, 64)); mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
                float den = 0.f;
#pragma unroll
                for (int b = 0; b < NT; ++b) { e[a][b][r] = expf(e[a][b][r] - mx); den += e[a][b][r]; }
                den += __shfl_xor(den, 1, 64); den += __shfl_xor(den, 2, 64);
                den += __shfl_xor(den, 4, 64); den += __shfl_xor(den, 8, 64);
                const float inv = 1.f / den;
#pragma unroll
                for (int b = 0; b < NT; ++b) pl[(a * 16 + 4 * fg + r) * LDP + b * 16 + fi] = e[a][b][r] * inv;
            }
        }
    // (wave-private LDS: ds operations of one wave are ordered, no barrier needed)
    // ---- Y = P X, 16 output columns at a time
    const float *xrow[NT][4];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = b * 16 + 4 * fg + j;
            n = n < nn ? n : nn - 1;         // P is exactly 0 there
            xrow[b][j] = node_row(p, p.xglo, p.xloc, ii, c, col0, n);
        }
    float4 pf[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
        if (a < TI) {
#pragma unroll
            for (int b = 0; b < NT; ++b) pf[a][b] = *reinterpret_cast<const float4 *>(&pl[(a * 16 + fi) * LDP + b * 16 + 4 * fg]);
        }
    for (int nt = 0; nt < S / 16; ++nt) {
        f32x4 y[NT];
#pragma unroll
        for (int a = 0; a < NT; ++a) y[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const float x0 = xrow[b][0][nt * 16 + fi], x1 = xrow[b][1][nt * 16 + fi];
            const float x2 = xrow[b][2][nt * 16 + fi], x3 = xrow[b][3][nt * 16 + fi];
#pragma unroll
            for (int a = 0; a < NT; ++a)
                if (a < TI) {
                    y[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[a][b].x, x0, y[a], 0, 0, 0);
                    y[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[a][b].y, x1, y[a], 0, 0, 0);
                    y[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[a][b].z, x2, y[a], 0, 0, 0);
                    y[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[a][b].w, x3, y[a], 0, 0, 0);
                }
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
            if (a < TI) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = a * 16 + 4 * fg + r;
                    if (n < (glo_only ? 1 : nn))
                        const_cast<float *>(node_row(p, yglo, yloc, ii, c, col0, n))[nt * 16 + fi] = y[a][r];
                }
            }
    }
}

// NTMAX = the largest node-tile count this launch handles.  The kernel carries the registers of its largest body -- 232 VGPRs,
// two waves per SIMD, for 64-node graphs -- and a graph of <= 16 / <= 32 nodes (captions of up to 15 / 31 words: all but a
// handful of any test split) needs 36 / 76, i.e. three to four times the waves in flight for a kernel that waits on memory.
// A caption set whose longest caption needs more than two tiles is served by TWO launches over the same pair grid: <2> takes
// the pairs of 1..2 tiles, <4> (nt_lo = 3) the rest; a wave whose pair belongs to the other launch leaves at once.
template <int NTMAX>
__global__ __launch_bounds__(256) void sgr_pair_kernel(PairArgs p, const float *__restrict__ qglo, const float *__restrict__ qloc,
                                                       const float *__restrict__ kglo, const float *__restrict__ kloc,
                                                       float *__restrict__ yglo, float *__restrict__ yloc, int64_t npairs,
                                                       int nt_lo, int glo_only) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= npairs) return;
    const int64_t ii = pair / p.Nc, c = pair % p.Nc;
    const int nn = p.cap_len[c] + 1, col0 = p.cap_col[c];
    float *pl = sm + (size_t)(threadIdx.x >> 6) * (NTMAX * 16) * (NTMAX * 16 + 4);
    const int NT = (nn + 15) / 16;
    if (NT < nt_lo || NT > NTMAX) return;               // this pair belongs to the other launch (mixed-length caption sets)
    if (NT == 1) sgr_pair_body<1>(p, qglo, qloc, kglo, kloc, yglo, yloc, ii, c, col0, nn, glo_only, pl);
    else if constexpr (NTMAX >= 2) {
        if (NT == 2) sgr_pair_body<2>(p, qglo, qloc, kglo, kloc, yglo, yloc, ii, c, col0, nn, glo_only, pl);
        else if constexpr (NTMAX >= 3) {
            if (NT == 3) sgr_pair_body<3>(p, qglo, qloc, kglo, kloc, yglo, yloc, ii, c, col0, nn, glo_only, pl);
            else sgr_pair_body<4>(p, qglo, qloc, kglo, kloc, yglo, yloc, ii, c, col0, nn, glo_only, pl);
        }
    }
}

__global__ __launch_bounds__(256) void sgr_final_kernel(const float *__restrict__ xglo, int64_t Nc, int S_, const float *__restrict__ eval_w,
                                                        const float *__restrict__ eval_b, int64_t npairs, float *__restrict__ S,
                                                        int64_t ldS, int64_t img_index0, int64_t ldg) {
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= npairs) return;
    const int lane = threadIdx.x & 63;
    const float *x = xglo + ((pair / Nc) * ldg + pair % Nc) * S_;
    float s = 0.f;
    for (int d = lane; d < S_; d += 64) s += x[d] * eval_w[d];
    s = wave_sum(s) + eval_b[0];
    if (lane == 0) S[(img_index0 + pair / Nc) * ldS + pair % Nc] = 1.f / (1.f + expf(-s));
}

static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
// Images per block of the pair stage (a multiple of the SCAN image tile).  64 by default since round 4 (rounds 1-3: 16) -- a quarter of
// the launches, and the persistent SGR kernel's tail (the last, partial round of items over the CUs) is a quarter as large: SGR 1k x 5k
// 735.8 -> 723.5 ms, SAF 373.3 -> 371.3 (same box; 32: 732.9 / 372.7).  The workspace grows with it (5k x 25k at 64: SAF 38 GB, SGR
// 42 GB fused), so the block is an ARGUMENT: the caller sizes it to the memory it has (itr_sgraf_pick_image_block) -- a validation pass
// inside a training process does not own the whole HBM.  Never larger than the call's image count rounded up to the image tile: small
// calls do not pay for 64 images.  Scores do not depend on it.  The unfused fallback (sim_dim != 256) materialises (ctx - E)^2: 4.
static inline int64_t sgraf_ib(int S, int64_t Ni, int image_block) {
    if (S != 256) return 4;
    int64_t ib = image_block > 0 ? image_block : ITR_SGRAF_DEFAULT_IMAGE_BLOCK;
    const int64_t need = (Ni + SC_IMGS - 1) / SC_IMGS * SC_IMGS;
    if (ib > need) ib = need;
    return ib < SC_IMGS ? SC_IMGS : ib;
}
static inline bool sgraf_block_ok(int image_block) {
    return image_block == 0 || (image_block >= SC_IMGS && image_block <= 64 && image_block % SC_IMGS == 0);
}
// (ctx_glo - cap_glo)^2 rows of the global nodes: only the GEMM chain of sim_dim != 256 (experiment builds can select that chain at 256 too)
static inline bool sgraf_needs_aglo(int S) {
#ifdef ITR_EXPERIMENT
    (void)S;
    return true;
#else
    return S != 256;
#endif
}
// the fused graph steps need neither the query nor the aggregate rows of the word nodes in memory (they live in the workgroup's LDS)
static inline bool sgraf_fused_layout(int module, int S, int flags) { return module == 1 && S == 256 && !(flags & ITR_SGRAF_UNFUSED_STEPS); }

}  // namespace itr

extern "C" size_t itr_sgraf_workspace_bytes(int64_t Ni, int64_t Nc, int64_t n_rows, int64_t n_tiles, int D, int S,
                                            int module, int image_block, int flags) {
    using namespace itr;
    const int64_t ncols = n_tiles * SC_NT, IB = sgraf_ib(S, Ni, image_block);
    const bool fused = sgraf_fused_layout(module, S, flags);
    size_t b = itr_scan_workspace_bytes(Ni, SC_R, n_rows, Nc, n_tiles, D) + 256;
    b += al((size_t)Ni * D * 4) * 3 + al((size_t)Ni * SC_R * D * 4) * 2;        // img_ave, g_emb_v, img_glo; l_emb_v, imgT
    b += al((size_t)n_rows * D * 4) + al((size_t)Nc * D * 4) * 3;                // l_emb_t; cap_ave, g_emb_t, cap_glo
    b += al((size_t)Nc * 4) + al((size_t)Nc * 8);                                // cap_col, seg offsets (unused slot)
    b += al((size_t)IB * ncols * SC_R * 4) + al((size_t)IB * ncols * 4) + al((size_t)IB * Nc * 4);   // P, cn, scan scratch
    const int64_t NcP = (Nc + SC_NT - 1) / SC_NT * SC_NT;                        // captions rounded up to whole 64-row tiles
    b += (S == 256 ? 0 : al((size_t)IB * ncols * D * 4)) + (sgraf_needs_aglo(S) ? al((size_t)IB * Nc * D * 4) : 0);   // Aloc, Aglo (sim_dim != 256 only)
    b += al((size_t)(NcP - Nc) * D * 4) + al((size_t)IB * NcP * SC_R * 4) + al((size_t)IB * NcP * 4) + al((size_t)IB * SC_R * D * 4);   // cap_glo tail, one-hot weights, unit norms, global "regions"
    b += al((size_t)IB * ncols * S * 4) + al((size_t)IB * NcP * S * 4);         // Xloc, Xglo
    if (module == 1) {
        b += al((size_t)IB * NcP * S * 4);                                       // Yglo
        if (!fused) b += al((size_t)IB * ncols * S * 4) * 2 + al((size_t)IB * NcP * S * 4);   // Qloc, Yloc, Qglo (step-by-step chain only)
        b += al((size_t)S * S * 4) * 2 + (al((size_t)S * S * 4) + al((size_t)S * 4)) * 8;   // W^T scratch, folded query weights
        if (S == 256) b += al(sgr_fused_workspace_bytes(Nc, Nc, 8)) + 256;       // group records (<= one per caption), weight fragments, flag
    }
    return b;
}

extern "C" int itr_sgraf_pick_image_block(int64_t Ni, int64_t Nc, int64_t n_rows, int64_t n_tiles, int D, int S, int module, int flags,
                                          size_t max_workspace_bytes, size_t *workspace_bytes) {
    using namespace itr;
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && n_rows >= 0 && n_tiles >= 0 && D > 0 && S > 0, "itr_sgraf_pick_image_block: bad shape");
    static const int cand[] = {64, 32, 16, 8, 4};
    int last = 0;
    for (int c : cand) {
        const int ib = (int)sgraf_ib(S, Ni, c);
        if (ib == last) continue;               // (clamped to the image count: the same block as the previous candidate)
        last = ib;
        const size_t b = itr_sgraf_workspace_bytes(Ni, Nc, n_rows, n_tiles, D, S, module, ib, flags);
        if (b <= max_workspace_bytes) {
            if (workspace_bytes) *workspace_bytes = b;
            return ib;
        }
    }
    set_error("itr_sgraf_pick_image_block: even a %d-image block needs more than the %zu bytes allowed", last, max_workspace_bytes);
    return ITR_ERR_UNSUPPORTED;
}

extern "C" int itr_sgraf_scores(const float *img, const float *words, const int64_t *cap_off, const int32_t *cap_len,
                                const int32_t *tile_begin_dev, const int32_t *cap_order_dev, int64_t n_tiles,
                                int64_t Ni, int64_t Nc, int64_t n_rows, int max_len, int R, int D, int S, int module,
                                int sgr_step, const itr_sgraf_weights *w, const int32_t *node_group_begin_dev,
                                const int32_t *node_group_order_dev, int64_t n_node_groups, int image_block, int flags, float *Sout,
                                int64_t ldS, void *workspace, size_t workspace_bytes, itr_stream_t stream) {
    using namespace itr;
    ITR_REQUIRE(img && words && cap_off && cap_len && tile_begin_dev && cap_order_dev && w && Sout && workspace,
                "itr_sgraf_scores: null pointer");
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && n_rows >= 0 && D > 0 && S > 0 && ldS >= Nc, "itr_sgraf_scores: bad shape");
    if (module != 0 && module != 1) { set_error("Invalid input of config.module_name in configs.py"); return ITR_ERR_BADARG; }
    ITR_UNSUPPORTED(R != SC_R, "itr_sgraf_scores: VisualSA is built for %d regions (BatchNorm1d(36)), got %d", SC_R, R);
    ITR_UNSUPPORTED(S > 1024 || (D % SC_BK) != 0, "itr_sgraf_scores: need sim_dim <= 1024 and embed dim %% 32 == 0");
    ITR_UNSUPPORTED(module == 1 && (sgr_step < 1 || sgr_step > 8), "itr_sgraf_scores: sgr_step must be in [1, 8]");
    ITR_UNSUPPORTED(max_len < 1 || max_len > 63, "itr_sgraf_scores: captions of 1..63 words are supported");
    ITR_REQUIRE(sgraf_block_ok(image_block), "itr_sgraf_scores: image_block must be 0 (default) or a multiple of %d in [%d, 64], got %d", SC_IMGS,
                SC_IMGS, image_block);
    ITR_REQUIRE((flags & ~(ITR_SGRAF_UNFUSED_STEPS | ITR_SGRAF_NON_PERSISTENT)) == 0, "itr_sgraf_scores: unknown flag bits %d", flags);
    ITR_REQUIRE(n_node_groups >= 0 && n_node_groups <= Nc, "itr_sgraf_scores: bad node-group count");
    // SGR with the configured sim_dim and a node-group plan: the graph steps run fused (sgr_fused.hip); otherwise step by step
    const bool have_plan = node_group_begin_dev && node_group_order_dev && n_node_groups > 0;
    if (module == 1 && S == 256 && !have_plan) flags |= ITR_SGRAF_UNFUSED_STEPS;      // no plan: the chain (and its larger workspace)
    const bool fused_sgr = sgraf_fused_layout(module, S, flags);      // (the step-by-step chain is the ABI flag ITR_SGRAF_UNFUSED_STEPS: layout and path agree)
    ITR_REQUIRE(workspace_bytes >= itr_sgraf_workspace_bytes(Ni, Nc, n_rows, n_tiles, D, S, module, image_block, flags),
                "itr_sgraf_scores: workspace too small (size it with the same image_block and flags; without a node-group plan the step-by-step "
                "chain runs: ITR_SGRAF_UNFUSED_STEPS)");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    hipStream_t st = as_stream(stream);
    const int64_t ncols = n_tiles * SC_NT, IB = sgraf_ib(S, Ni, image_block);
    const bool fused_layout = sgraf_fused_layout(module, S, flags);      // (what the workspace was sized for)

    // ---- carve
    char *p = static_cast<char *>(workspace);
    auto take = [&](size_t bytes) { char *q = p; p += al(bytes); return q; };
    const size_t scan_bytes = itr_scan_workspace_bytes(Ni, SC_R, n_rows, Nc, n_tiles, D);
    void *scan_ws = take(scan_bytes + 256);
    float *img_ave = (float *)take((size_t)Ni * D * 4), *g_emb_v = (float *)take((size_t)Ni * D * 4);
    float *img_glo = (float *)take((size_t)Ni * D * 4);
    float *l_emb_v = (float *)take((size_t)Ni * SC_R * D * 4), *imgT = (float *)take((size_t)Ni * SC_R * D * 4);
    float *l_emb_t = (float *)take((size_t)n_rows * D * 4);
    float *cap_ave = (float *)take((size_t)Nc * D * 4), *g_emb_t = (float *)take((size_t)Nc * D * 4);
    const int64_t NcP = (Nc + SC_NT - 1) / SC_NT * SC_NT;
    // sim_dim 256: the global nodes come from the local-node kernel (ITR_SGRAF_GLO_GEMM=1: the (a - b)^2 kernel + GEMM + l2norm chain
    // of rounds 1-2, for A/B timing); their rows then lie in whole 64-caption tiles: ldg = NcP rows per image
    const bool glo_loc = (S == 256) && !ITR_EXP_ENV("ITR_SGRAF_GLO_GEMM");
    const int64_t ldg = glo_loc ? NcP : Nc;
    float *cap_glo = (float *)take((size_t)Nc * D * 4 + (size_t)(NcP - Nc) * D * 4);       // + zero rows up to the last tile
    int32_t *cap_col = (int32_t *)take((size_t)Nc * 4);
    take((size_t)Nc * 8);
    float *P = (float *)take((size_t)IB * ncols * SC_R * 4), *cn = (float *)take((size_t)IB * ncols * 4);
    float *sscr = (float *)take((size_t)IB * Nc * 4);
    float *Aloc = (S == 256) ? nullptr : (float *)take((size_t)IB * ncols * D * 4);
    float *Aglo = sgraf_needs_aglo(S) ? (float *)take((size_t)IB * Nc * D * 4) : nullptr;
    float *Pg = (float *)take((size_t)IB * NcP * SC_R * 4), *cng = (float *)take((size_t)IB * NcP * 4);
    float *gimg = (float *)take((size_t)IB * SC_R * D * 4);
    float *Xloc = (float *)take((size_t)IB * ncols * S * 4), *Xglo = (float *)take((size_t)IB * NcP * S * 4);
    float *Qloc = nullptr, *Qglo = nullptr, *Yloc = nullptr, *Yglo = nullptr;
    float *WqT = nullptr, *WkT = nullptr, *Wfold[8] = {nullptr}, *vfold[8] = {nullptr};
    if (module == 1) {
        Yglo = (float *)take((size_t)IB * NcP * S * 4);
        if (!fused_layout) {
            Qloc = (float *)take((size_t)IB * ncols * S * 4); Yloc = (float *)take((size_t)IB * ncols * S * 4);
            Qglo = (float *)take((size_t)IB * NcP * S * 4);
        }
        WqT = (float *)take((size_t)S * S * 4); WkT = (float *)take((size_t)S * S * 4);
        for (int k = 0; k < 8; ++k) { Wfold[k] = (float *)take((size_t)S * S * 4); vfold[k] = (float *)take((size_t)S * 4); }
    }
    void *fused_ws = nullptr;
    int *fused_bad = nullptr;
    if (module == 1 && S == 256) {
        fused_ws = take(sgr_fused_workspace_bytes(Nc, Nc, 8));
        fused_bad = (int *)take(256);
    }
    int rc;
#define SG_TRY(x) { rc = (x); if (rc != ITR_OK) return rc; }

    // ---- global nodes: VisualSA (:491-507)
    ITR_REQUIRE(Ni <= 65535, "itr_sgraf_scores: at most 65535 images per call");
    SG_TRY(itr_mean_mid(img, img_ave, Ni, SC_R, D, stream));
    SG_TRY(gemm_nt(img, D, w->v_loc_w, D, w->v_loc_b, l_emb_v, D, Ni * SC_R, D, D, 0, st));
    {
        dim3 grid((unsigned)ceil_div(D, 256), (unsigned)1);
        // rows can exceed 65535: launch in slabs
        for (int64_t r0 = 0; r0 < Ni * SC_R; r0 += 65520) {   // 65520 is a multiple of 36
            const int64_t nr = (Ni * SC_R - r0 < 65520) ? Ni * SC_R - r0 : 65520;
            hipLaunchKernelGGL(bn_tanh_kernel, dim3((unsigned)ceil_div(D, 256), (unsigned)nr), dim3(256), 0, st,
                               l_emb_v + r0 * D, nr, D, 1, SC_R, w->v_loc_bn_w, w->v_loc_bn_b, w->v_loc_bn_mean, w->v_loc_bn_var);
        }
        ITR_CHECK_LAUNCH("sgraf bn_tanh (local)");
        (void)grid;
    }
    SG_TRY(gemm_nt(img_ave, D, w->v_glo_w, D, w->v_glo_b, g_emb_v, D, Ni, D, D, 0, st));
    hipLaunchKernelGGL(bn_tanh_kernel, dim3((unsigned)ceil_div(D, 256), (unsigned)Ni), dim3(256), 0, st, g_emb_v, Ni, D, 0, 1,
                       w->v_glo_bn_w, w->v_glo_bn_b, w->v_glo_bn_mean, w->v_glo_bn_var);
    ITR_CHECK_LAUNCH("sgraf bn_tanh (global)");
    hipLaunchKernelGGL(sa_pool_kernel, dim3((unsigned)Ni), dim3(256), 0, st, img, l_emb_v, g_emb_v, w->v_com_w, w->v_com_b,
                       (const int64_t *)nullptr, (const int32_t *)nullptr, SC_R, D, img_glo);
    ITR_CHECK_LAUNCH("sgraf sa_pool (image)");
    // ---- TextSA (:543-559)
    SG_TRY(gemm_nt(words, D, w->t_loc_w, D, w->t_loc_b, l_emb_t, D, n_rows, D, D, 2 /*tanh*/, st));
    hipLaunchKernelGGL(seg_mean_kernel, dim3((unsigned)Nc), dim3(256), 0, st, words, cap_off, cap_len, D, cap_ave);
    ITR_CHECK_LAUNCH("sgraf seg_mean");
    SG_TRY(gemm_nt(cap_ave, D, w->t_glo_w, D, w->t_glo_b, g_emb_t, D, Nc, D, D, 2, st));
    hipLaunchKernelGGL(sa_pool_kernel, dim3((unsigned)Nc), dim3(256), 0, st, words, l_emb_t, g_emb_t, w->t_com_w, w->t_com_b, cap_off,
                       cap_len, 0, D, cap_glo);
    ITR_CHECK_LAUNCH("sgraf sa_pool (caption)");
    // ---- operand prep for the pair stage
    hipLaunchKernelGGL(transpose_img_kernel, dim3((unsigned)ceil_div(D, 64), (unsigned)Ni), dim3(256), 0, st, img, SC_R, D, imgT);
    ITR_CHECK_LAUNCH("sgraf transpose");
    SG_TRY(scan_prepare_impl(img, words, cap_off, cap_len, tile_begin_dev, cap_order_dev, n_tiles, Ni, Nc, n_rows, SC_R, D, 0,
                             scan_ws, scan_bytes, cap_col, stream));
    // the tile-packed words live at a fixed place of the scan workspace (scan_carve: meta first, then wtiled)
    const float *wtiled = reinterpret_cast<const float *>(static_cast<char *>(scan_ws) + al((size_t)n_tiles * sizeof(ScanTileMeta)));

    // ---- SGR: fold the key projection into the query projection (GraphReasoning, Fusionmodule.py:589-597).
    // edge = softmax_j(q_i . k_j) with q = Wq x + bq, k = Wk x + bk.  q_i . k_j = (Wk^T q_i) . x_j + q_i . bk and the
    // second term does not depend on j, so it cancels in the softmax:  edge = softmax_j(q'_i . x_j) with
    //     q' = (Wk^T Wq) x + Wk^T bq.
    // One S x S projection per node and step instead of two; the pair kernel reads the nodes themselves as keys.
    if (module == 1) {
        const unsigned tb = (unsigned)ceil_div((int64_t)S * S, 256);
        for (int k = 0; k < sgr_step; ++k) {
            hipLaunchKernelGGL(transpose_small_kernel, dim3(tb), dim3(256), 0, st, w->sgr_q_w[k], S, S, WqT);
            hipLaunchKernelGGL(transpose_small_kernel, dim3(tb), dim3(256), 0, st, w->sgr_k_w[k], S, S, WkT);
            ITR_CHECK_LAUNCH("sgraf weight transpose");
            // Wfold[b][a] = sum_o Wk[o][b] Wq[o][a];  vfold[b] = sum_o bq[o] Wk[o][b]
            SG_TRY(gemm_nt(WkT, S, WqT, S, nullptr, Wfold[k], S, S, S, S, 0, st));
            SG_TRY(gemm_nt(w->sgr_q_b[k], S, WkT, S, nullptr, vfold[k], S, 1, S, S, 0, st));
        }
    }

    if (module == 1) {
        // rows of captions Nc .. NcP - 1 (and of refused groups) are never written by the pair / fused kernels but run through the GEMMs of
        // the steps: defined values, not uninitialised workspace
        // (Qglo -- step-by-step chain only -- is a GEMM output: every row the pair kernel reads is written first)
        ITR_CHECK_HIP(hipMemsetAsync(Yglo, 0, (size_t)IB * NcP * S * 4, st));
    }
    if (fused_sgr) {
        ITR_CHECK_HIP(hipMemsetAsync(fused_bad, 0, sizeof(int), st));
        SG_TRY(sgr_fused_prepare(node_group_begin_dev, node_group_order_dev, n_node_groups, Nc, cap_len, cap_col, Wfold, w->sgr_g_w, sgr_step,
                                 fused_ws, fused_bad, st));
    }

    if (glo_loc) {
        if (NcP > Nc) ITR_CHECK_HIP(hipMemsetAsync(cap_glo + Nc * D, 0, (size_t)(NcP - Nc) * D * 4, st));
        ITR_CHECK_HIP(hipMemsetAsync(gimg, 0, (size_t)IB * SC_R * D * 4, st));
        hipLaunchKernelGGL(glo_onehot_kernel, dim3((unsigned)ceil_div(IB * NcP * SC_R, (int64_t)256)), dim3(256), 0, st, Pg, cng, IB * NcP);
        ITR_CHECK_LAUNCH("sgraf glo one-hot");
    }
    PairArgs pa{Xglo, Xloc, cap_col, cap_len, Nc, ncols, S, ldg};
    for (int64_t i0 = 0; i0 < Ni; i0 += IB) {
        const int64_t nb = (Ni - i0 < IB) ? Ni - i0 : IB;
        // 1. attention weights + context norms  (SCAN_attention: clipped_l2norm, smooth 9)
        SG_TRY(scan_scores_impl(img, n_tiles, Ni, Nc, n_rows, SC_R, D, 0, 0, 0, 9.0f, 6.0f, sscr, Nc, scan_ws, scan_bytes, P, cn, i0,
                                nb, stream));
        // 2. (l2norm(ctx) - E)^2 per image, 3. sim_loc
        if (S == 256) {   // the configured sim_dim: fused, (ctx - E)^2 never leaves the chip (sgraf_loc.hip)
            SG_TRY(sgraf_loc_fused(P, cn, img + i0 * SC_R * D, wtiled, w->loc_w, w->loc_b, Xloc, nb, n_tiles, D, st));
        } else {
            for (int64_t ii = 0; ii < nb; ++ii)
                SG_TRY(gemm_nt_sqdiff(P + ii * ncols * SC_R, SC_R, imgT + (i0 + ii) * D * SC_R, SC_R, cn + ii * ncols, wtiled, D,
                                      Aloc + ii * ncols * D, D, ncols, D, SC_R, st));
            SG_TRY(gemm_nt(Aloc, D, w->loc_w, D, w->loc_b, Xloc, S, nb * ncols, S, D, 0, st));
            SG_TRY(norm_rows(Xloc, Xloc, nb * ncols, S, 1e-8f, 0, 0, st));
        }
        // 4. sim_glo
        if (glo_loc) {
            // row 0 of image ii's "region set" <- img_glo[i0 + ii] (the other 35 rows stay zero), then the local-node kernel
            ITR_CHECK_HIP(hipMemcpy2DAsync(gimg, (size_t)SC_R * D * 4, img_glo + i0 * D, (size_t)D * 4, (size_t)D * 4, (size_t)nb,
                                           hipMemcpyDeviceToDevice, st));
            SG_TRY(sgraf_loc_fused(Pg, cng, gimg, cap_glo, w->glo_w, w->glo_b, Xglo, nb, NcP / SC_NT, D, st));
        } else {
            hipLaunchKernelGGL(glo_sqdiff_kernel, dim3((unsigned)Nc, (unsigned)nb), dim3(256), 0, st, img_glo + i0 * D, cap_glo, Nc, D, Aglo);
            ITR_CHECK_LAUNCH("sgraf glo_sqdiff");
            SG_TRY(gemm_nt(Aglo, D, w->glo_w, D, w->glo_b, Xglo, S, nb * Nc, S, D, 0, st));
            SG_TRY(norm_rows(Xglo, Xglo, nb * Nc, S, 1e-8f, 0, 0, st));
        }
        const int64_t npairs = nb * Nc;
        if (module == 0) {
            hipLaunchKernelGGL(saf_pair_kernel, dim3((unsigned)ceil_div(npairs, 4)), dim3(256), 0, st, pa, w->saf_w, w->saf_b, w->saf_bn_w,
                               w->saf_bn_b, w->saf_bn_mean, w->saf_bn_var, w->eval_w, w->eval_b, npairs, Sout, ldS, i0);
            ITR_CHECK_LAUNCH("sgraf saf_pair");
        } else if (fused_sgr) {
            // all the graph steps up to the last step's attention in one workgroup per group of captions; the last step's graph
            // projection of node 0 (the only node read afterwards, Fusionmodule.py:443) for ALL the graphs of the block as one GEMM
            SG_TRY(sgr_fused_scores(Xloc, Xglo, fused_ws, n_node_groups, Nc, nb, ldg, ncols, vfold, w->sgr_g_b, sgr_step, Yglo,
                                    !(flags & ITR_SGRAF_NON_PERSISTENT), st));
            SG_TRY(gemm_nt(Yglo, S, w->sgr_g_w[sgr_step - 1], S, w->sgr_g_b[sgr_step - 1], Xglo, S, nb * ldg, S, S, 1 /*relu*/, st));
            hipLaunchKernelGGL(sgr_final_kernel, dim3((unsigned)ceil_div(npairs, 4)), dim3(256), 0, st, Xglo, Nc, S, w->eval_w, w->eval_b,
                               npairs, Sout, ldS, i0, ldg);
            ITR_CHECK_LAUNCH("sgraf sgr_final (fused steps)");
        } else {
            const int ntmax = (max_len + 1 + 15) / 16;
            ITR_UNSUPPORTED(S % 16 != 0, "itr_sgraf_scores: SGR needs sim_dim %% 16 == 0");
            for (int k = 0; k < sgr_step; ++k) {
                const int last = (k == sgr_step - 1);
                if (!last) SG_TRY(gemm_nt(Xloc, S, Wfold[k], S, vfold[k], Qloc, S, nb * ncols, S, S, 0, st));   // last: only node 0 queries
                SG_TRY(gemm_nt(Xglo, S, Wfold[k], S, vfold[k], Qglo, S, nb * ldg, S, S, 0, st));
                const dim3 pgrid((unsigned)ceil_div(npairs, 4));
                auto plds = [](int nt) { return (size_t)4 * (nt * 16) * (nt * 16 + 4) * 4; };   // 4 waves x P[NT*16][NT*16+4]
                if (ntmax == 1) {
                    hipLaunchKernelGGL(sgr_pair_kernel<1>, pgrid, dim3(256), plds(1), st, pa, Qglo, Qloc, Xglo, Xloc, Yglo, Yloc, npairs, 1, last);
                } else {
                    hipLaunchKernelGGL(sgr_pair_kernel<2>, pgrid, dim3(256), plds(2), st, pa, Qglo, Qloc, Xglo, Xloc, Yglo, Yloc, npairs, 1, last);
                    if (ntmax > 2) {
                        ITR_CHECK_LAUNCH("sgraf sgr_pair");
                        hipLaunchKernelGGL(sgr_pair_kernel<4>, pgrid, dim3(256), plds(4), st, pa, Qglo, Qloc, Xglo, Xloc, Yglo, Yloc, npairs, 3, last);
                    }
                }
                ITR_CHECK_LAUNCH("sgraf sgr_pair");
                // NOTE: a word node is shared by all captions... it is NOT: node rows are per (image, word) and a word
                // belongs to one caption, so writing Yloc rows per pair is race-free.
                if (!last) SG_TRY(gemm_nt(Yloc, S, w->sgr_g_w[k], S, w->sgr_g_b[k], Xloc, S, nb * ncols, S, S, 1 /*relu*/, st));
                SG_TRY(gemm_nt(Yglo, S, w->sgr_g_w[k], S, w->sgr_g_b[k], Xglo, S, nb * ldg, S, S, 1, st));
            }
            hipLaunchKernelGGL(sgr_final_kernel, dim3((unsigned)ceil_div(npairs, 4)), dim3(256), 0, st, Xglo, Nc, S, w->eval_w, w->eval_b,
                               npairs, Sout, ldS, i0, ldg);
            ITR_CHECK_LAUNCH("sgraf sgr_final");
        }
    }
    // a refused group of a hand-made node-group plan: NaN in its captions' columns (never uninitialised memory; sgr_fused.hip)
    if (fused_sgr) SG_TRY(sgr_fused_finish(fused_ws, n_node_groups, Nc, sgr_step, Ni, Sout, ldS, st));
#undef SG_TRY
    return ITR_OK;
}
