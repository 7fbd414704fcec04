// SGRAF similarity module, TRAINING step, all (image, caption) pairs of a batch at once
// (EncoderSimilarity.forward in train mode + its backward: Fusionmodule.py:406-451, SCAN_attention :632-664, GraphReasoning :579-586,
// AttentionFiltration :613-618, TextSA :549-564; called from SGRAF.train_emb, Models.py:518-546).
//
// The reference loops over the captions of the batch and, per caption, over ~30 small ATen launches; rounds 1-5 restated that loop on
// the autograd tape in groups of equal-length captions (85 / 224 ms per 128 x 128 step, 3 % of the fp32 MFMA peak).  Here the whole
// B x C block of pairs lives in IMAGE-MAJOR ragged matrices and every stage is one launch:
//
//   words  [T, D]           packed word embeddings of the C captions, caption c = rows cap_off[c] .. cap_off[c + 1]
//   local  [B * T, .]       row b T + t            one row per (image, word): attention weights, context, local alignment
//   nodes  [B * (T + C), S] row b (T + C) + cap_off[c] + c + j   graph of pair (b, c): j = 0 the global alignment, j = 1 .. W_c the words
//   pairs  [B * C, .]       row b C + c            (the similarity matrix itself)
//
// Image-major because everything between the attention logits and the local alignment vectors is then a plain matrix product per
// image against ALL words of the batch (caption boundaries matter only to the two small softmax / l2norm kernels), and the reductions
// of the backward pass run along whole axes: d words = sum over b, d regions = sum over t.  The dense layers between these stages are
// the library's GEMMs (autograd.linear: NT forward / dx, split-row TN for dW); this file is the ragged glue, forward and backward.
#include "itr_common.h"
#include <mutex>

namespace itr {

int allow_dynamic_lds(const void *kernel, size_t bytes);      // scan_train.hip

__device__ __forceinline__ float block_sum_256(float v, float *red4) {       // red4: 4 floats of LDS; all 256 threads call
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
    __syncthreads();
    return red4[0] + red4[1] + red4[2] + red4[3];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K1  attention weights of SCAN_attention (Fusionmodule.py:641-652) for pair (b, c):  a = LeakyReLU_0.1(A[b, r, w]); l2norm over the
//     words of the caption; softmax over the regions of smooth * a  ->  P[(b, t), r].   A [B R, ldA] = regions . words^T (one GEMM).
__global__ __launch_bounds__(64) void sgt_attn_fwd_kernel(const float *__restrict__ A, int64_t ldA, const int32_t *__restrict__ cap_off, int T, int R,
                                                          float smooth, float eps, float *__restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int off = cap_off[c], W = cap_off[c + 1] - off;
    float *a = sm, *p = sm + R * W, *nrm = p + R * W;
    for (int idx = lane; idx < R * W; idx += 64) {
        const int r = idx / W, w = idx - r * W;
        const float v = A[((int64_t)b * R + r) * ldA + off + w];
        a[idx] = v > 0.f ? v : 0.1f * v;
    }
    __syncthreads();
    for (int r = lane; r < R; r += 64) {
        float s = 0.f;
        for (int w = 0; w < W; ++w) s = fmaf(a[r * W + w], a[r * W + w], s);
        nrm[r] = sqrtf(s) + eps;
    }
    __syncthreads();
    for (int w = lane; w < W; w += 64) {
        float m = -INFINITY;
        for (int r = 0; r < R; ++r) m = fmaxf(m, smooth * (a[r * W + w] / nrm[r]));
        float den = 0.f;
        for (int r = 0; r < R; ++r) {
            const float e = expf(smooth * (a[r * W + w] / nrm[r]) - m);
            p[w * R + r] = e;
            den += e;
        }
        for (int r = 0; r < R; ++r) p[w * R + r] /= den;
    }
    __syncthreads();
    float *out = P + ((int64_t)b * T + off) * R;
    for (int idx = lane; idx < R * W; idx += 64) out[idx] = p[idx];
}

// backward of K1: dP [B T, R] -> dA [B R, ldA] (every element written exactly once)
__global__ __launch_bounds__(64) void sgt_attn_bwd_kernel(const float *__restrict__ A, int64_t ldA, const float *__restrict__ P,
                                                          const float *__restrict__ dP, const int32_t *__restrict__ cap_off, int T, int R, float smooth,
                                                          float eps, float *__restrict__ dA) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int off = cap_off[c], W = cap_off[c + 1] - off;
    float *a = sm, *pl = sm + R * W, *gl = pl + R * W, *nrm = gl + R * W;
    const float *Pb = P + ((int64_t)b * T + off) * R, *Gb = dP + ((int64_t)b * T + off) * R;
    for (int idx = lane; idx < R * W; idx += 64) {
        const int r = idx / W, w = idx - r * W;
        const float v = A[((int64_t)b * R + r) * ldA + off + w];
        a[idx] = v > 0.f ? v : 0.1f * v;
        pl[idx] = Pb[idx];
        gl[idx] = Gb[idx];
    }
    __syncthreads();
    for (int r = lane; r < R; r += 64) {
        float s = 0.f;
        for (int w = 0; w < W; ++w) s = fmaf(a[r * W + w], a[r * W + w], s);
        nrm[r] = sqrtf(s);
    }
    for (int w = lane; w < W; w += 64) {          // softmax backward over the regions, times smooth: d (a / norm)
        float dot = 0.f;
        for (int r = 0; r < R; ++r) dot = fmaf(pl[w * R + r], gl[w * R + r], dot);
        for (int r = 0; r < R; ++r) gl[w * R + r] = pl[w * R + r] * (gl[w * R + r] - dot) * smooth;
    }
    __syncthreads();
    for (int r = lane; r < R; r += 64) {          // l2norm backward over the words, then LeakyReLU'
        const float n = nrm[r], ne = n + eps;
        float sd = 0.f;
        for (int w = 0; w < W; ++w) sd = fmaf(gl[w * R + r], a[r * W + w], sd);
        const float coef = n > 0.f ? sd / (n * ne * ne) : 0.f;
        for (int w = 0; w < W; ++w) {
            const float x = a[r * W + w];
            const float dx = gl[w * R + r] / ne - x * coef;
            a[r * W + w] = x > 0.f ? dx : 0.1f * dx;
        }
    }
    __syncthreads();
    for (int idx = lane; idx < R * W; idx += 64) {
        const int r = idx / W, w = idx - r * W;
        dA[((int64_t)b * R + r) * ldA + off + w] = a[idx];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K2  X[(b, t), :] = (l2norm(sum_r P[(b, t), r] img[b, r, :]) - words[t, :])^2   (Fusionmodule.py:654-662 + :426): the weighted context is
//     never stored; its norm is (cnorm) for the backward pass.  One workgroup = 16 words of one image; a thread owns VEC float4 columns.
constexpr int SGT_TW = 16;
template <int VEC>
__global__ __launch_bounds__(256) void sgt_ctx_fwd_kernel(const float *__restrict__ P, const float *__restrict__ img, const float *__restrict__ words,
                                                          int T, int R, int D, float eps, float *__restrict__ X, float *__restrict__ cnorm) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // Ptt[R][16] | red[16][4]
    float *Ptt = sm, *red = sm + R * SGT_TW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * SGT_TW, b = blockIdx.y;
    const int nt = T - t0 < SGT_TW ? T - t0 : SGT_TW;
    const float *Pb = P + ((int64_t)b * T + t0) * R;
    for (int idx = tid; idx < SGT_TW * R; idx += 256) {
        const int i = idx / R, r = idx - i * R;
        Ptt[r * SGT_TW + i] = i < nt ? Pb[idx] : 0.f;
    }
    __syncthreads();
    const int D4 = D >> 2;
    float4 acc[SGT_TW][VEC];
#pragma unroll
    for (int i = 0; i < SGT_TW; ++i)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[i][v] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *img4 = reinterpret_cast<const float4 *>(img) + (int64_t)b * R * D4;
    for (int r0 = 0; r0 < R; r0 += 4) {        // four regions' columns requested together (one memory latency per four regions, not per region)
        float4 xb[4][VEC];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                xb[j][v] = (r0 + j < R && tid + v * 256 < D4) ? img4[(int64_t)(r0 + j) * D4 + tid + v * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + j < R ? r0 + j : R - 1;
            const float4 *pr = reinterpret_cast<const float4 *>(Ptt + r * SGT_TW);
#pragma unroll
            for (int q = 0; q < SGT_TW / 4; ++q) {
                const float4 p4 = pr[q];
                const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        float4 &c = acc[q * 4 + k][v];
                        c.x = fmaf(pp[k], xb[j][v].x, c.x);
                        c.y = fmaf(pp[k], xb[j][v].y, c.y);
                        c.z = fmaf(pp[k], xb[j][v].z, c.z);
                        c.w = fmaf(pp[k], xb[j][v].w, c.w);
                    }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < SGT_TW; ++i) {
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) s += acc[i][v].x * acc[i][v].x + acc[i][v].y * acc[i][v].y + acc[i][v].z * acc[i][v].z + acc[i][v].w * acc[i][v].w;
        s = wave_sum(s);
        if (lane == 0) red[i * 4 + wave] = s;
    }
    __syncthreads();
    const float4 *w4 = reinterpret_cast<const float4 *>(words);
    float4 *X4 = reinterpret_cast<float4 *>(X);
#pragma unroll
    for (int i = 0; i < SGT_TW; ++i) {
        if (i >= nt) continue;
        const float n = sqrtf(red[i * 4] + red[i * 4 + 1] + red[i * 4 + 2] + red[i * 4 + 3]);
        if (tid == 0) cnorm[(int64_t)b * T + t0 + i] = n;
        const float ne = n + eps;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int col = tid + v * 256;
            if (col >= D4) continue;
            const float4 w = w4[(int64_t)(t0 + i) * D4 + col];
            float4 o;
            o.x = acc[i][v].x / ne - w.x;
            o.y = acc[i][v].y / ne - w.y;
            o.z = acc[i][v].z / ne - w.z;
            o.w = acc[i][v].w / ne - w.w;
            o.x *= o.x; o.y *= o.y; o.z *= o.z; o.w *= o.w;
            X4[((int64_t)b * T + t0 + i) * D4 + col] = o;
        }
    }
}

// K3  backward of K2 through the square and the l2norm, for a tile of 8 words and a range of images (the workgroup walks the images so that
//     d words = -sum_b g accumulates in registers):  g = 2 (ctx / (n + eps) - w) dX;  d ctx = g / (n + eps) - ctx (g . ctx) / (n (n + eps)^2).
//     Writes d ctx [B T, D] (the operand of the two contractions that follow: K4 d regions, K5 d P) and one partial d words per image range.
constexpr int SGT_TB = 8;
template <int VEC>
__global__ __launch_bounds__(256) void sgt_ctx_bwd_kernel(const float *__restrict__ P, const float *__restrict__ img, const float *__restrict__ words,
                                                          const float *__restrict__ cnorm, const float *__restrict__ dX, int B, int T, int R, int D,
                                                          float eps, int b_per_split, float *__restrict__ dctx, float *__restrict__ dwpart) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // Ptt[R][8] | red[8][4]
    float *Ptt = sm, *red = sm + R * SGT_TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * SGT_TB;
    const int nt = T - t0 < SGT_TB ? T - t0 : SGT_TB;
    const int b0 = blockIdx.y * b_per_split, b1 = b0 + b_per_split < B ? b0 + b_per_split : B;
    const int D4 = D >> 2;
    const float4 *w4 = reinterpret_cast<const float4 *>(words);
    const float4 *dX4 = reinterpret_cast<const float4 *>(dX);
    float4 *dc4 = reinterpret_cast<float4 *>(dctx);
    float4 dw[SGT_TB][VEC];      // (the word rows are re-read per image from the cache: 32 registers less -> one more wave per SIMD)
#pragma unroll
    for (int i = 0; i < SGT_TB; ++i)
#pragma unroll
        for (int v = 0; v < VEC; ++v) dw[i][v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = b0; b < b1; ++b) {
        __syncthreads();
        const float *Pb = P + ((int64_t)b * T + t0) * R;
        for (int idx = tid; idx < SGT_TB * R; idx += 256) {
            const int i = idx / R, r = idx - i * R;
            Ptt[r * SGT_TB + i] = i < nt ? Pb[idx] : 0.f;
        }
        __syncthreads();
        float4 acc[SGT_TB][VEC], g[SGT_TB][VEC];
#pragma unroll
        for (int i = 0; i < SGT_TB; ++i)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[i][v] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 *img4 = reinterpret_cast<const float4 *>(img) + (int64_t)b * R * D4;
        // four regions requested together (one per trip was one L2 round trip per 32 fmaf; at 174 registers the same change measured slower --
        // it fits since the word rows left the register file)
        for (int r0 = 0; r0 < R; r0 += 4) {
            float4 x[4][VEC];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    x[j][v] = (r0 + j < R && tid + v * 256 < D4) ? img4[(int64_t)(r0 + j) * D4 + tid + v * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (r0 + j >= R) continue;
                const float4 *pr = reinterpret_cast<const float4 *>(Ptt + (r0 + j) * SGT_TB);
#pragma unroll
                for (int q = 0; q < SGT_TB / 4; ++q) {
                    const float4 p4 = pr[q];
                    const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            float4 &c = acc[q * 4 + k][v];
                            c.x = fmaf(pp[k], x[j][v].x, c.x);
                            c.y = fmaf(pp[k], x[j][v].y, c.y);
                            c.z = fmaf(pp[k], x[j][v].z, c.z);
                            c.w = fmaf(pp[k], x[j][v].w, c.w);
                        }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SGT_TB; ++i) {
            float s = 0.f;
            if (i < nt) {
                const float ne = cnorm[(int64_t)b * T + t0 + i] + eps;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const int col = tid + v * 256;
                    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (col < D4) d = dX4[((int64_t)b * T + t0 + i) * D4 + col];
                    float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (col < D4) wv = w4[(int64_t)(t0 + i) * D4 + col];
                    float4 q;
                    q.x = 2.f * (acc[i][v].x / ne - wv.x) * d.x;
                    q.y = 2.f * (acc[i][v].y / ne - wv.y) * d.y;
                    q.z = 2.f * (acc[i][v].z / ne - wv.z) * d.z;
                    q.w = 2.f * (acc[i][v].w / ne - wv.w) * d.w;
                    g[i][v] = q;
                    dw[i][v].x -= q.x; dw[i][v].y -= q.y; dw[i][v].z -= q.z; dw[i][v].w -= q.w;
                    s += q.x * acc[i][v].x + q.y * acc[i][v].y + q.z * acc[i][v].z + q.w * acc[i][v].w;
                }
            }
            s = wave_sum(s);
            if (lane == 0) red[i * 4 + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SGT_TB; ++i) {
            if (i >= nt) continue;
            const float n = cnorm[(int64_t)b * T + t0 + i], ne = n + eps;
            const float sd = red[i * 4] + red[i * 4 + 1] + red[i * 4 + 2] + red[i * 4 + 3];
            const float coef = n > 0.f ? sd / (n * ne * ne) : 0.f;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int col = tid + v * 256;
                if (col >= D4) continue;
                float4 o;
                o.x = g[i][v].x / ne - acc[i][v].x * coef;
                o.y = g[i][v].y / ne - acc[i][v].y * coef;
                o.z = g[i][v].z / ne - acc[i][v].z * coef;
                o.w = g[i][v].w / ne - acc[i][v].w * coef;
                dc4[((int64_t)b * T + t0 + i) * D4 + col] = o;
            }
        }
    }
    float4 *dwp = reinterpret_cast<float4 *>(dwpart) + (int64_t)blockIdx.y * T * D4;
#pragma unroll
    for (int i = 0; i < SGT_TB; ++i) {
        if (i >= nt) continue;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int col = tid + v * 256;
            if (col < D4) dwp[(int64_t)(t0 + i) * D4 + col] = dw[i][v];
        }
    }
}

// out[e] = sum over the slices, in slice order
__global__ __launch_bounds__(256) void sgt_sum_slices_kernel(const float *__restrict__ part, int nsl, int64_t n, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int k = 0; k < nsl; ++k) s += part[(int64_t)k * n + e];
    out[e] = s;
}

// K5  dP[(b, t), r] = d ctx[(b, t), :] . img[b, r, :]  -- per image a [T x D] . [D x R] product on v_mfma_f32_16x16x4_f32: a wave owns 16 words
//     x NT region tiles, the 32-column chunks of both operands are staged in LDS (row stride 36 floats: conflict-free for the lane = 16 k + i map).
template <int NT>
__global__ __launch_bounds__(256) void sgt_dp_kernel(const float *__restrict__ dctx, const float *__restrict__ img, int T, int R, int D,
                                                     float *__restrict__ dP) {
    __shared__ __attribute__((aligned(16))) float Ds[64][36];
    __shared__ __attribute__((aligned(16))) float Is[NT * 16][36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * 64, b = blockIdx.y;
    const int D4 = D >> 2;
    const float4 *dc4 = reinterpret_cast<const float4 *>(dctx) + ((int64_t)b * T + t0) * D4;
    const float4 *im4 = reinterpret_cast<const float4 *>(img) + (int64_t)b * R * D4;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < D4; d0 += 8) {
        __syncthreads();
        for (int idx = tid; idx < 64 * 8; idx += 256) {
            const int row = idx >> 3, c4 = idx & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + row < T && d0 + c4 < D4) v = dc4[(int64_t)row * D4 + d0 + c4];
            *reinterpret_cast<float4 *>(&Ds[row][c4 * 4]) = v;
        }
        for (int idx = tid; idx < NT * 16 * 8; idx += 256) {
            const int row = idx >> 3, c4 = idx & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < R && d0 + c4 < D4) v = im4[(int64_t)row * D4 + d0 + c4];
            *reinterpret_cast<float4 *>(&Is[row][c4 * 4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = kk * 4 + (lane >> 4);
            const float a = Ds[wave * 16 + (lane & 15)][k];
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Is[j * 16 + (lane & 15)][k], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = j * 16 + (lane & 15);
        if (r >= R) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = t0 + wave * 16 + 4 * (lane >> 4) + q;
            if (t < T) dP[((int64_t)b * T + t) * R + r] = acc[j][q];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K6  global alignment input (Fusionmodule.py:429): Xg[(b, c), :] = (img_glo[b] - cap_glo[c])^2, and its two reductions
__global__ __launch_bounds__(256) void sgt_pair_sqdiff_fwd_kernel(const float *__restrict__ ig, const float *__restrict__ cg, int C, int D,
                                                                  float *__restrict__ X) {
    const int c = blockIdx.x, b = blockIdx.y;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float v = ig[(int64_t)b * D + d] - cg[(int64_t)c * D + d];
        X[((int64_t)b * C + c) * D + d] = v * v;
    }
}
__global__ __launch_bounds__(256) void sgt_pair_sqdiff_bwd_img_kernel(const float *__restrict__ ig, const float *__restrict__ cg,
                                                                      const float *__restrict__ dX, int C, int D, float *__restrict__ dig) {
    const int b = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const float x = ig[(int64_t)b * D + d];
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(2.f * (x - cg[(int64_t)c * D + d]), dX[((int64_t)b * C + c) * D + d], s);
    dig[(int64_t)b * D + d] = s;
}
__global__ __launch_bounds__(256) void sgt_pair_sqdiff_bwd_cap_kernel(const float *__restrict__ ig, const float *__restrict__ cg,
                                                                      const float *__restrict__ dX, int B, int C, int D, float *__restrict__ dcg) {
    const int c = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const float y = cg[(int64_t)c * D + d];
    float s = 0.f;
    for (int b = 0; b < B; ++b) s = fmaf(-2.f * (ig[(int64_t)b * D + d] - y), dX[((int64_t)b * C + c) * D + d], s);
    dcg[(int64_t)c * D + d] = s;
}

// K7  the graph of every pair (Fusionmodule.py:433): node 0 = global alignment, nodes 1 .. W = local alignments.  node_cap[q] = caption of
//     node column q (q in [0, T + C)); the word of a local node is q - caption - 1.  dir 0: (glo, loc) -> nodes; dir 1: nodes -> (glo, loc).
__global__ __launch_bounds__(64) void sgt_nodes_kernel(float *__restrict__ glo, float *__restrict__ loc, float *__restrict__ nodes,
                                                       const int32_t *__restrict__ cap_off, const int32_t *__restrict__ node_cap, int C, int T, int S,
                                                       int dir) {
    const int q = blockIdx.x, b = blockIdx.y;
    const int c = node_cap[q];
    const int j = q - cap_off[c] - c;
    float *src = j == 0 ? glo + ((int64_t)b * C + c) * S : loc + ((int64_t)b * T + (q - c - 1)) * S;
    float *nd = nodes + ((int64_t)b * (T + C) + q) * S;
    if (dir == 0)
        for (int s = threadIdx.x; s < S; s += 64) nd[s] = src[s];
    else
        for (int s = threadIdx.x; s < S; s += 64) src[s] = nd[s];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K8  one GraphReasoning step on every pair (Fusionmodule.py:579-586):  E = softmax_r(q_v . k_r);  Z_v = sum_r E[v, r] x_r.
//     One workgroup per pair; q / k chunks of SC columns in LDS for the n x n dot products, E in LDS, the mixing with a thread per column.
struct SgtGraph {
    const int32_t *cap_off, *e_off;       // e_off[c] = sum_{c' < c} n_{c'}^2;  e_off[C] = the total per image
    int C, T, S, SC;
};
__device__ __forceinline__ void sgt_stage(const float *__restrict__ src, int64_t row0, int n, int S, int s0, int SC, float *dst) {
    const int SCP = SC + 4, c4n = SC >> 2;
    for (int idx = threadIdx.x; idx < n * c4n; idx += 256) {
        const int row = idx / c4n, c4 = idx - row * c4n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s0 + c4 * 4 < S) v = *reinterpret_cast<const float4 *>(src + (row0 + row) * S + s0 + c4 * 4);
        *reinterpret_cast<float4 *>(dst + row * SCP + c4 * 4) = v;
    }
}
// M[v nr + r] (+)= sum over the staged columns of a[v][.] b[r][.]   (v < nv query rows, r < nr key rows)
__device__ __forceinline__ void sgt_dots(const float *a, const float *b, float *M, int nv, int nr, int SC, bool first) {
    const int SCP = SC + 4;
    for (int e = threadIdx.x; e < nv * nr; e += 256) {
        const int v = e / nr, r = e - v * nr;
        const float4 *pa = reinterpret_cast<const float4 *>(a + v * SCP), *pb = reinterpret_cast<const float4 *>(b + r * SCP);
        float s = 0.f;
        for (int k = 0; k < (SC >> 2); ++k) {
            const float4 x = pa[k], y = pb[k];
            s = fmaf(x.x, y.x, s);
            s = fmaf(x.y, y.y, s);
            s = fmaf(x.z, y.z, s);
            s = fmaf(x.w, y.w, s);
        }
        M[e] = first ? s : M[e] + s;
    }
}
// out[o, s] = sum_i Mx(o, i) in[i, s] with M [nv, nr]:  trans == false: o < nv, i < nr, Mx = M[o nr + i];  trans == true: o < nr, i < nv,
// Mx = M[i nr + o].  `in` / `out` point at row 0 of their blocks; a thread owns columns s.
__device__ __forceinline__ void sgt_colmix(const float *M, bool trans, int nv, int nr, const float *__restrict__ in, float *__restrict__ out, int S) {
    const int no = trans ? nr : nv, ni = trans ? nv : nr;
    for (int s = threadIdx.x; s < S; s += 256) {
        for (int o0 = 0; o0 < no; o0 += 16) {
            float acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            // the column values of eight inner rows are REQUESTED together and then consumed (a load -> 16 fmaf -> load chain paid one
            // memory latency per inner row: 21 x 3 products x 2 output chunks per workgroup -- most of the kernel's 1.16 ms, round 6)
            for (int r0 = 0; r0 < ni; r0 += 8) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = r0 + j < ni ? in[(int64_t)(r0 + j) * S + s] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j;
                    if (r < ni) {                                         // (uniform over the workgroup)
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            if (o0 + i < no) acc[i] = fmaf(trans ? M[r * nr + o0 + i] : M[(o0 + i) * nr + r], x[j], acc[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (o0 + i < no) out[(int64_t)(o0 + i) * S + s] = acc[i];
        }
    }
}
// ROW0: only node 0 of every pair asks (the LAST reasoning step: the reference reads sim_emb[:, 0, :] of its output, Fusionmodule.py:437-438) --
// q / Z / dq are then pair rows (b C + c) and E is one row of n edge weights, saved in node order.
template <bool ROW0>
__global__ __launch_bounds__(256) void sgt_graph_fwd_kernel(SgtGraph g, const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ x,
                                                            float *__restrict__ Esave, float *__restrict__ Z) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c = blockIdx.x, b = blockIdx.y;
    const int n = g.cap_off[c + 1] - g.cap_off[c] + 1;
    const int nv = ROW0 ? 1 : n;
    const int64_t row0 = (int64_t)b * (g.T + g.C) + g.cap_off[c] + c;
    const int64_t qrow = ROW0 ? (int64_t)b * g.C + c : row0;
    const int nn4 = (nv * n + 3) & ~3;
    float *E = sm, *qs = sm + nn4, *ks = qs + nv * (g.SC + 4);
    for (int s0 = 0; s0 < g.S; s0 += g.SC) {
        __syncthreads();
        sgt_stage(q, qrow, nv, g.S, s0, g.SC, qs);
        sgt_stage(k, row0, n, g.S, s0, g.SC, ks);
        __syncthreads();
        sgt_dots(qs, ks, E, nv, n, g.SC, s0 == 0);
    }
    __syncthreads();
    for (int v = threadIdx.x; v < nv; v += 256) {
        float m = -INFINITY;
        for (int r = 0; r < n; ++r) m = fmaxf(m, E[v * n + r]);
        float den = 0.f;
        for (int r = 0; r < n; ++r) {
            const float e = expf(E[v * n + r] - m);
            E[v * n + r] = e;
            den += e;
        }
        for (int r = 0; r < n; ++r) E[v * n + r] /= den;
    }
    __syncthreads();
    float *Eo = ROW0 ? Esave + row0 : Esave + (int64_t)b * g.e_off[g.C] + g.e_off[c];
    for (int e = threadIdx.x; e < nv * n; e += 256) Eo[e] = E[e];
    sgt_colmix(E, false, nv, n, x + row0 * g.S, Z + qrow * g.S, g.S);
}
template <bool ROW0>
__global__ __launch_bounds__(256) void sgt_graph_bwd_kernel(SgtGraph g, const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ x,
                                                            const float *__restrict__ Esave, const float *__restrict__ dZ, float *__restrict__ dq,
                                                            float *__restrict__ dk, float *__restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c = blockIdx.x, b = blockIdx.y;
    const int n = g.cap_off[c + 1] - g.cap_off[c] + 1;
    const int nv = ROW0 ? 1 : n;
    const int64_t row0 = (int64_t)b * (g.T + g.C) + g.cap_off[c] + c;
    const int64_t qrow = ROW0 ? (int64_t)b * g.C + c : row0;
    const int nn4 = (nv * n + 3) & ~3;
    float *E = sm, *dE = sm + nn4, *as = dE + nn4, *bs = as + nv * (g.SC + 4);
    const float *Ei = ROW0 ? Esave + row0 : Esave + (int64_t)b * g.e_off[g.C] + g.e_off[c];
    for (int e = threadIdx.x; e < nv * n; e += 256) E[e] = Ei[e];
    for (int s0 = 0; s0 < g.S; s0 += g.SC) {          // dE[v, r] = dZ_v . x_r
        __syncthreads();
        sgt_stage(dZ, qrow, nv, g.S, s0, g.SC, as);
        sgt_stage(x, row0, n, g.S, s0, g.SC, bs);
        __syncthreads();
        sgt_dots(as, bs, dE, nv, n, g.SC, s0 == 0);
    }
    __syncthreads();
    sgt_colmix(E, true, nv, n, dZ + qrow * g.S, dx + row0 * g.S, g.S);      // dx_r = sum_v E[v, r] dZ_v   (the mixing path only; autograd adds the q / k paths)
    for (int v = threadIdx.x; v < nv; v += 256) {                            // softmax backward per query row
        float dot = 0.f;
        for (int r = 0; r < n; ++r) dot = fmaf(dE[v * n + r], E[v * n + r], dot);
        for (int r = 0; r < n; ++r) dE[v * n + r] = E[v * n + r] * (dE[v * n + r] - dot);
    }
    __syncthreads();
    sgt_colmix(dE, false, nv, n, k + row0 * g.S, dq + qrow * g.S, g.S);      // dq_v = sum_r dS[v, r] k_r
    sgt_colmix(dE, true, nv, n, q + qrow * g.S, dk + row0 * g.S, g.S);       // dk_r = sum_v dS[v, r] q_v
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K9  AttentionFiltration (Fusionmodule.py:613-618).  Its BatchNorm1d(1) sees ONE caption per call in the reference: batch statistics over
//     the B (W_c + 1) attention logits of caption c -> one workgroup per caption.  y = gamma (a - mean_c) invstd_c + beta.
__global__ __launch_bounds__(256) void sgt_segbn_fwd_kernel(const float *__restrict__ a, const int32_t *__restrict__ cap_off, int B, int C, int T,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                            float *__restrict__ y, float *__restrict__ mean, float *__restrict__ var,
                                                            float *__restrict__ invstd) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int n = cap_off[c + 1] - cap_off[c] + 1, q0 = cap_off[c] + c, NT = T + C;
    const int N = B * n;
    float s = 0.f;
    for (int e = threadIdx.x; e < N; e += 256) s += a[(int64_t)(e / n) * NT + q0 + e % n];
    const float mu = block_sum_256(s, red) / N;
    float s2 = 0.f;
    for (int e = threadIdx.x; e < N; e += 256) {
        const float d = a[(int64_t)(e / n) * NT + q0 + e % n] - mu;
        s2 = fmaf(d, d, s2);
    }
    const float vr = block_sum_256(s2, red) / N;
    const float is = 1.f / sqrtf(vr + eps);
    if (threadIdx.x == 0) { mean[c] = mu; var[c] = vr; invstd[c] = is; }
    const float ga = gamma[0], be = beta[0];
    for (int e = threadIdx.x; e < N; e += 256) {
        const int64_t o = (int64_t)(e / n) * NT + q0 + e % n;
        y[o] = fmaf(ga * is, a[o] - mu, be);
    }
}
__global__ __launch_bounds__(256) void sgt_segbn_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ a, const int32_t *__restrict__ cap_off,
                                                            int B, int C, int T, const float *__restrict__ gamma, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, float *__restrict__ da, float *__restrict__ dgamma_c,
                                                            float *__restrict__ dbeta_c) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int n = cap_off[c + 1] - cap_off[c] + 1, q0 = cap_off[c] + c, NT = T + C;
    const int N = B * n;
    const float mu = mean[c], is = invstd[c];
    float sb = 0.f, sg = 0.f;
    for (int e = threadIdx.x; e < N; e += 256) {
        const int64_t o = (int64_t)(e / n) * NT + q0 + e % n;
        sb += dy[o];
        sg = fmaf(dy[o], (a[o] - mu) * is, sg);
    }
    const float db = block_sum_256(sb, red);
    const float dg = block_sum_256(sg, red);
    if (threadIdx.x == 0) { dgamma_c[c] = dg; dbeta_c[c] = db; }
    const float ga = gamma[0];
    for (int e = threadIdx.x; e < N; e += 256) {
        const int64_t o = (int64_t)(e / n) * NT + q0 + e % n;
        const float xh = (a[o] - mu) * is;
        da[o] = ga * is * (dy[o] - db / N - xh * dg / N);
    }
}
//     pooling: g = sigmoid(y); weights = g / (sum |g| + eps) over the nodes of the pair; out[(b, c), :] = sum_j weights_j nodes_j
__global__ __launch_bounds__(256) void sgt_saf_pool_fwd_kernel(const float *__restrict__ y, const float *__restrict__ nodes,
                                                               const int32_t *__restrict__ cap_off, int C, int T, int S, float eps,
                                                               float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // wgt[n]
    const int c = blockIdx.x, b = blockIdx.y;
    const int n = cap_off[c + 1] - cap_off[c] + 1;
    const int64_t row0 = (int64_t)b * (T + C) + cap_off[c] + c;
    for (int j = threadIdx.x; j < n; j += 256) sm[j] = 1.f / (1.f + expf(-y[row0 + j]));
    __syncthreads();
    float l1 = 0.f;
    for (int j = 0; j < n; ++j) l1 += fabsf(sm[j]);
    l1 += eps;
    for (int s = threadIdx.x; s < S; s += 256) {
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc = fmaf(sm[j] / l1, nodes[(row0 + j) * S + s], acc);
        out[((int64_t)b * C + c) * S + s] = acc;
    }
}
__global__ __launch_bounds__(256) void sgt_saf_pool_bwd_kernel(const float *__restrict__ y, const float *__restrict__ nodes, const float *__restrict__ dout,
                                                               const int32_t *__restrict__ cap_off, int C, int T, int S, float eps,
                                                               float *__restrict__ dy, float *__restrict__ dnodes) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // g[n] | dwgt[n][4]
    const int c = blockIdx.x, b = blockIdx.y;
    const int n = cap_off[c + 1] - cap_off[c] + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)b * (T + C) + cap_off[c] + c;
    float *gsm = sm, *dw = sm + n;
    for (int j = threadIdx.x; j < n; j += 256) gsm[j] = 1.f / (1.f + expf(-y[row0 + j]));
    __syncthreads();
    float l1 = 0.f;
    for (int j = 0; j < n; ++j) l1 += fabsf(gsm[j]);
    l1 += eps;
    const float *dob = dout + ((int64_t)b * C + c) * S;
    for (int j = 0; j < n; ++j) {
        const float wj = gsm[j] / l1;
        float part = 0.f;
        for (int s = threadIdx.x; s < S; s += 256) {
            const float d = dob[s];
            part = fmaf(d, nodes[(row0 + j) * S + s], part);
            dnodes[(row0 + j) * S + s] = wj * d;
        }
        part = wave_sum(part);
        if (lane == 0) dw[j * 4 + wave] = part;
    }
    __syncthreads();
    float tot = 0.f;          // sum_i dwgt_i wgt_i
    for (int j = 0; j < n; ++j) tot = fmaf(dw[j * 4] + dw[j * 4 + 1] + dw[j * 4 + 2] + dw[j * 4 + 3], gsm[j] / l1, tot);
    for (int j = threadIdx.x; j < n; j += 256) {
        const float dwj = dw[j * 4] + dw[j * 4 + 1] + dw[j * 4 + 2] + dw[j * 4 + 3];
        const float dgj = (dwj - tot) / l1;                  // g > 0: d |g| = d g
        dy[row0 + j] = dgj * gsm[j] * (1.f - gsm[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K10 / K11  TextSA on packed captions (Fusionmodule.py:549-564): the caption mean (or sum) and its transpose -- one caption row spread over
//     the caption's words (the g_emb.repeat of :556, no atomics) --, and softmax(logits over the words of a caption) . words
__global__ __launch_bounds__(256) void sgt_seg_mean_fwd_kernel(const float *__restrict__ words, const int32_t *__restrict__ cap_off, int D, int mean,
                                                               float *__restrict__ out) {
    const int c = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const int off = cap_off[c], W = cap_off[c + 1] - off;
    float s = 0.f;
    for (int w = 0; w < W; ++w) s += words[(int64_t)(off + w) * D + d];
    out[(int64_t)c * D + d] = mean ? s / W : s;
}
__global__ __launch_bounds__(256) void sgt_seg_mean_bwd_kernel(const float *__restrict__ dout, const int32_t *__restrict__ cap_off, int D, int mean,
                                                               float *__restrict__ dwords) {
    const int c = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const int off = cap_off[c], W = cap_off[c + 1] - off;
    const float v = mean ? dout[(int64_t)c * D + d] / W : dout[(int64_t)c * D + d];
    for (int w = 0; w < W; ++w) dwords[(int64_t)(off + w) * D + d] = v;
}
__global__ __launch_bounds__(256) void sgt_seg_smry_fwd_kernel(const float *__restrict__ logit, const float *__restrict__ words,
                                                               const int32_t *__restrict__ cap_off, int D, float *__restrict__ p,
                                                               float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // p[W]
    const int c = blockIdx.x;
    const int off = cap_off[c], W = cap_off[c + 1] - off;
    float m = -INFINITY;
    for (int w = 0; w < W; ++w) m = fmaxf(m, logit[off + w]);
    float den = 0.f;
    for (int w = 0; w < W; ++w) den += expf(logit[off + w] - m);
    for (int w = threadIdx.x; w < W; w += 256) {
        const float v = expf(logit[off + w] - m) / den;
        sm[w] = v;
        p[off + w] = v;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        float s = 0.f;
        for (int w = 0; w < W; ++w) s = fmaf(sm[w], words[(int64_t)(off + w) * D + d], s);
        out[(int64_t)c * D + d] = s;
    }
}
__global__ __launch_bounds__(256) void sgt_seg_smry_bwd_kernel(const float *__restrict__ p, const float *__restrict__ words, const float *__restrict__ dout,
                                                               const int32_t *__restrict__ cap_off, int D, float *__restrict__ dlogit,
                                                               float *__restrict__ dwords) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // dp[W][4]
    const int c = blockIdx.x;
    const int off = cap_off[c], W = cap_off[c + 1] - off;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < W; ++w) {
        const float pw = p[off + w];
        float part = 0.f;
        for (int d = threadIdx.x; d < D; d += 256) {
            const float g = dout[(int64_t)c * D + d];
            part = fmaf(g, words[(int64_t)(off + w) * D + d], part);
            dwords[(int64_t)(off + w) * D + d] = pw * g;
        }
        part = wave_sum(part);
        if (lane == 0) sm[w * 4 + wave] = part;
    }
    __syncthreads();
    float dot = 0.f;
    for (int w = 0; w < W; ++w) dot = fmaf(sm[w * 4] + sm[w * 4 + 1] + sm[w * 4 + 2] + sm[w * 4 + 3], p[off + w], dot);
    for (int w = threadIdx.x; w < W; w += 256) dlogit[off + w] = p[off + w] * (sm[w * 4] + sm[w * 4 + 1] + sm[w * 4 + 2] + sm[w * 4 + 3] - dot);
}

static int sgt_lds(const void *kernel, size_t bytes, const char *what) {
    if (bytes > 150 * 1024) {
        set_error("%s: %zu bytes of LDS needed (caption too long for the batched training kernels)", what, bytes);
        return ITR_ERR_UNSUPPORTED;
    }
    if (bytes > 32 * 1024) return allow_dynamic_lds(kernel, 150 * 1024);
    return ITR_OK;
}

// the column chunk of the graph kernels: as wide as SGT_GRAPH_LDS bytes allow next to the two n x n matrices (at least 32, a multiple of
// 32, or all of S when S < 32).  16 KB: a graph's workgroup is a chain of staged chunks and barriers, so what hides it is the number of
// graphs resident on the CU -- 32 KB (5 per CU, two chunks of S = 256) measured 20.4 ms on the SGR step, 16 KB (8 per CU, four chunks)
// 20.05, 12 KB 20.2
#ifndef SGT_GRAPH_LDS
#define SGT_GRAPH_LDS (16 * 1024)
#endif
static int sgt_graph_chunk(int nmax, int S) {
    if (S <= 32) return S;
    int64_t room = (SGT_GRAPH_LDS / 4 - 2 * (int64_t)((nmax * nmax + 3) & ~3)) / (2 * (int64_t)nmax) - 4;
    int sc = (int)(room / 32 * 32);
    if (sc < 32) sc = 32;
    if (sc > S) sc = (S + 3) & ~3;
    return sc;
}

}  // namespace itr

using namespace itr;

#define SGT_SHAPE(cond, name) ITR_REQUIRE(cond, name ": bad shape")

extern "C" int itr_sgt_attn_fwd(const float *A, int64_t ldA, const int32_t *cap_off, int B, int C, int T, int R, int Wmax, float smooth, float eps,
                                float *P, itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && C >= 0 && C <= 65535 && B <= 65535 && T >= 0 && R >= 1 && Wmax >= 0 && ldA >= T, "itr_sgt_attn_fwd");
    if (B == 0 || C == 0 || T == 0) return ITR_OK;
    ITR_REQUIRE(A && cap_off && P, "itr_sgt_attn_fwd: null pointer");
    const size_t lds = ((size_t)2 * R * Wmax + R) * sizeof(float);
    const int rc = sgt_lds(reinterpret_cast<const void *>(sgt_attn_fwd_kernel), lds, "itr_sgt_attn_fwd");
    if (rc != ITR_OK) return rc;
    hipLaunchKernelGGL(sgt_attn_fwd_kernel, dim3(C, B), dim3(64), lds, as_stream(stream), A, ldA, cap_off, T, R, smooth, eps, P);
    ITR_CHECK_LAUNCH("sgt_attn_fwd");
    return ITR_OK;
}

extern "C" int itr_sgt_attn_bwd(const float *A, int64_t ldA, const float *P, const float *dP, const int32_t *cap_off, int B, int C, int T, int R,
                                int Wmax, float smooth, float eps, float *dA, itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && C >= 0 && C <= 65535 && B <= 65535 && T >= 0 && R >= 1 && Wmax >= 0 && ldA >= T, "itr_sgt_attn_bwd");
    if (B == 0 || C == 0 || T == 0) return ITR_OK;
    ITR_REQUIRE(A && P && dP && cap_off && dA, "itr_sgt_attn_bwd: null pointer");
    const size_t lds = ((size_t)3 * R * Wmax + R) * sizeof(float);
    const int rc = sgt_lds(reinterpret_cast<const void *>(sgt_attn_bwd_kernel), lds, "itr_sgt_attn_bwd");
    if (rc != ITR_OK) return rc;
    hipLaunchKernelGGL(sgt_attn_bwd_kernel, dim3(C, B), dim3(64), lds, as_stream(stream), A, ldA, P, dP, cap_off, T, R, smooth, eps, dA);
    ITR_CHECK_LAUNCH("sgt_attn_bwd");
    return ITR_OK;
}

extern "C" int itr_sgt_ctx_fwd(const float *P, const float *img, const float *words, int B, int T, int R, int D, float eps, float *X, float *cnorm,
                               itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && T >= 0 && R >= 1 && D >= 4, "itr_sgt_ctx_fwd");
    ITR_UNSUPPORTED(D % 4 != 0 || D > 2048, "itr_sgt_ctx_fwd: embedding width %d (a multiple of 4, at most 2048)", D);
    if (B == 0 || T == 0) return ITR_OK;
    ITR_REQUIRE(P && img && words && X && cnorm, "itr_sgt_ctx_fwd: null pointer");
    const size_t lds = ((size_t)R * SGT_TW + SGT_TW * 4) * sizeof(float);
    const dim3 grid((unsigned)ceil_div(T, SGT_TW), (unsigned)B);
    if (D <= 1024) {
        const int rc = sgt_lds(reinterpret_cast<const void *>(sgt_ctx_fwd_kernel<1>), lds, "itr_sgt_ctx_fwd");
        if (rc != ITR_OK) return rc;
        hipLaunchKernelGGL(sgt_ctx_fwd_kernel<1>, grid, dim3(256), lds, as_stream(stream), P, img, words, T, R, D, eps, X, cnorm);
    } else {
        const int rc = sgt_lds(reinterpret_cast<const void *>(sgt_ctx_fwd_kernel<2>), lds, "itr_sgt_ctx_fwd");
        if (rc != ITR_OK) return rc;
        hipLaunchKernelGGL(sgt_ctx_fwd_kernel<2>, grid, dim3(256), lds, as_stream(stream), P, img, words, T, R, D, eps, X, cnorm);
    }
    ITR_CHECK_LAUNCH("sgt_ctx_fwd");
    return ITR_OK;
}

// Image ranges per word tile of the context backward.  A workgroup walks its images for ~0.5 ms, so the grid should be ONE full round of
// the chip's resident workgroups (1.24 rounds -- 955 workgroups on 768 slots -- took as long as two: round 6), never a round and a bit:
// splits = resident slots / word tiles, rounded down (resident = CUs x the occupancy the runtime reports for this kernel, cached per
// device).  The result only changes the order of the partial sums' final addition.
static int sgt_ctx_splits(int B, int T, int D) {
    static std::mutex mu;
    static int64_t resident_of[16][2] = {};
    int dev = 0;
    int64_t resident = 768;
    const int vi = D <= 1024 ? 0 : 1;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) {
        std::lock_guard<std::mutex> lock(mu);
        if (!resident_of[dev][vi]) {
            hipDeviceProp_t prop;
            int per_cu = 0;
            const hipError_t e = vi == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sgt_ctx_bwd_kernel<1>, 256, 4096)
                                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sgt_ctx_bwd_kernel<2>, 256, 4096);
            if (e == hipSuccess && per_cu >= 1 && hipGetDeviceProperties(&prop, dev) == hipSuccess) resident_of[dev][vi] = (int64_t)prop.multiProcessorCount * per_cu;
            else { (void)hipGetLastError(); resident_of[dev][vi] = 768; }
        }
        resident = resident_of[dev][vi];
    }
    const int64_t tiles = ceil_div(T > 0 ? T : 1, SGT_TB);
    int64_t s = resident / tiles;
    if (s > B) s = B;
    if (s > 16) s = 16;
    return (int)(s < 1 ? 1 : s);
}
extern "C" size_t itr_sgt_ctx_bwd_workspace_bytes(int B, int T, int D) {
    if (B < 1 || T < 1 || D < 1) return 0;
    (void)sgt_ctx_splits;      // (the split count depends on the device: size for the most it can be)
    const int s = B < 16 ? B : 16;
    return (size_t)s * T * D * sizeof(float);
}
extern "C" int itr_sgt_ctx_bwd(const float *P, const float *img, const float *words, const float *cnorm, const float *dX, int B, int T, int R, int D,
                               float eps, float *dctx, float *dwords, void *workspace, size_t workspace_bytes, itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && T >= 0 && R >= 1 && D >= 4, "itr_sgt_ctx_bwd");
    ITR_UNSUPPORTED(D % 4 != 0 || D > 2048, "itr_sgt_ctx_bwd: embedding width %d (a multiple of 4, at most 2048)", D);
    if (B == 0 || T == 0) return ITR_OK;
    ITR_REQUIRE(P && img && words && cnorm && dX && dctx && dwords && workspace, "itr_sgt_ctx_bwd: null pointer");
    const int bps = (int)ceil_div(B, sgt_ctx_splits(B, T, D));
    const int ns = (int)ceil_div(B, bps);
    ITR_REQUIRE(workspace_bytes >= (size_t)ns * T * D * sizeof(float), "itr_sgt_ctx_bwd: workspace too small (itr_sgt_ctx_bwd_workspace_bytes)");
    const size_t lds = ((size_t)R * SGT_TB + SGT_TB * 4) * sizeof(float);
    const dim3 grid((unsigned)ceil_div(T, SGT_TB), (unsigned)ns);
    float *part = static_cast<float *>(workspace);
    if (D <= 1024) {
        const int rc = sgt_lds(reinterpret_cast<const void *>(sgt_ctx_bwd_kernel<1>), lds, "itr_sgt_ctx_bwd");
        if (rc != ITR_OK) return rc;
        hipLaunchKernelGGL(sgt_ctx_bwd_kernel<1>, grid, dim3(256), lds, as_stream(stream), P, img, words, cnorm, dX, B, T, R, D, eps, bps, dctx, part);
    } else {
        const int rc = sgt_lds(reinterpret_cast<const void *>(sgt_ctx_bwd_kernel<2>), lds, "itr_sgt_ctx_bwd");
        if (rc != ITR_OK) return rc;
        hipLaunchKernelGGL(sgt_ctx_bwd_kernel<2>, grid, dim3(256), lds, as_stream(stream), P, img, words, cnorm, dX, B, T, R, D, eps, bps, dctx, part);
    }
    ITR_CHECK_LAUNCH("sgt_ctx_bwd");
    const int64_t n = (int64_t)T * D;
    hipLaunchKernelGGL(sgt_sum_slices_kernel, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, as_stream(stream), (const float *)part, ns, n, dwords);
    ITR_CHECK_LAUNCH("sgt_sum_slices");
    return ITR_OK;
}

extern "C" int itr_sgt_dp(const float *dctx, const float *img, int B, int T, int R, int D, float *dP, itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && T >= 0 && R >= 1 && D >= 4, "itr_sgt_dp");
    ITR_UNSUPPORTED(D % 4 != 0 || R > 64, "itr_sgt_dp: D %% 4 == 0 and at most 64 regions (got D %d, R %d)", D, R);
    if (B == 0 || T == 0) return ITR_OK;
    ITR_REQUIRE(dctx && img && dP, "itr_sgt_dp: null pointer");
    const dim3 grid((unsigned)ceil_div(T, 64), (unsigned)B);
    hipStream_t st = as_stream(stream);
    switch ((R + 15) / 16) {
        case 1: hipLaunchKernelGGL(sgt_dp_kernel<1>, grid, dim3(256), 0, st, dctx, img, T, R, D, dP); break;
        case 2: hipLaunchKernelGGL(sgt_dp_kernel<2>, grid, dim3(256), 0, st, dctx, img, T, R, D, dP); break;
        case 3: hipLaunchKernelGGL(sgt_dp_kernel<3>, grid, dim3(256), 0, st, dctx, img, T, R, D, dP); break;
        default: hipLaunchKernelGGL(sgt_dp_kernel<4>, grid, dim3(256), 0, st, dctx, img, T, R, D, dP); break;
    }
    ITR_CHECK_LAUNCH("sgt_dp");
    return ITR_OK;
}

extern "C" int itr_sgt_pair_sqdiff_fwd(const float *img_glo, const float *cap_glo, int B, int C, int D, float *X, itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && C >= 0 && C <= 65535 && D >= 1, "itr_sgt_pair_sqdiff_fwd");
    if (B == 0 || C == 0) return ITR_OK;
    ITR_REQUIRE(img_glo && cap_glo && X, "itr_sgt_pair_sqdiff_fwd: null pointer");
    hipLaunchKernelGGL(sgt_pair_sqdiff_fwd_kernel, dim3(C, B), dim3(256), 0, as_stream(stream), img_glo, cap_glo, C, D, X);
    ITR_CHECK_LAUNCH("sgt_pair_sqdiff_fwd");
    return ITR_OK;
}
extern "C" int itr_sgt_pair_sqdiff_bwd(const float *img_glo, const float *cap_glo, const float *dX, int B, int C, int D, float *dimg_glo,
                                       float *dcap_glo, itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && C >= 0 && C <= 65535 && D >= 1, "itr_sgt_pair_sqdiff_bwd");
    if (B == 0 || C == 0) return ITR_OK;
    ITR_REQUIRE(img_glo && cap_glo && dX && dimg_glo && dcap_glo, "itr_sgt_pair_sqdiff_bwd: null pointer");
    hipLaunchKernelGGL(sgt_pair_sqdiff_bwd_img_kernel, dim3((unsigned)ceil_div(D, 256), B), dim3(256), 0, as_stream(stream), img_glo, cap_glo, dX, C, D,
                       dimg_glo);
    hipLaunchKernelGGL(sgt_pair_sqdiff_bwd_cap_kernel, dim3((unsigned)ceil_div(D, 256), C), dim3(256), 0, as_stream(stream), img_glo, cap_glo, dX, B, C, D,
                       dcap_glo);
    ITR_CHECK_LAUNCH("sgt_pair_sqdiff_bwd");
    return ITR_OK;
}

extern "C" int itr_sgt_nodes(float *glo, float *loc, float *nodes, const int32_t *cap_off, const int32_t *node_cap, int B, int C, int T, int S,
                             int backward, itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && C >= 0 && T >= 0 && S >= 1, "itr_sgt_nodes");
    if (B == 0 || C == 0) return ITR_OK;
    ITR_REQUIRE(glo && (loc || T == 0) && nodes && cap_off && node_cap, "itr_sgt_nodes: null pointer");
    hipLaunchKernelGGL(sgt_nodes_kernel, dim3(T + C, B), dim3(64), 0, as_stream(stream), glo, loc, nodes, cap_off, node_cap, C, T, S, backward ? 1 : 0);
    ITR_CHECK_LAUNCH("sgt_nodes");
    return ITR_OK;
}

extern "C" int itr_sgt_graph_fwd(const float *q, const float *k, const float *x, const int32_t *cap_off, const int32_t *e_off, int B, int C, int T, int S,
                                 int nmax, int row0_only, float *E, float *Z, itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && C >= 0 && C <= 65535 && T >= 0 && S >= 4 && nmax >= 1, "itr_sgt_graph_fwd");
    ITR_UNSUPPORTED(S % 4 != 0, "itr_sgt_graph_fwd: sim_dim %d is not a multiple of 4", S);
    if (B == 0 || C == 0) return ITR_OK;
    ITR_REQUIRE(q && k && x && cap_off && e_off && E && Z, "itr_sgt_graph_fwd: null pointer");
    const int SC = sgt_graph_chunk(nmax, S);
    const size_t lds = ((size_t)((nmax * nmax + 3) & ~3) + 2 * (size_t)nmax * (SC + 4)) * sizeof(float);
    const void *kern = row0_only ? reinterpret_cast<const void *>(sgt_graph_fwd_kernel<true>) : reinterpret_cast<const void *>(sgt_graph_fwd_kernel<false>);
    const int rc = sgt_lds(kern, lds, "itr_sgt_graph_fwd");
    if (rc != ITR_OK) return rc;
    const SgtGraph g{cap_off, e_off, C, T, S, SC};
    if (row0_only) hipLaunchKernelGGL(sgt_graph_fwd_kernel<true>, dim3(C, B), dim3(256), lds, as_stream(stream), g, q, k, x, E, Z);
    else hipLaunchKernelGGL(sgt_graph_fwd_kernel<false>, dim3(C, B), dim3(256), lds, as_stream(stream), g, q, k, x, E, Z);
    ITR_CHECK_LAUNCH("sgt_graph_fwd");
    return ITR_OK;
}
extern "C" int itr_sgt_graph_bwd(const float *q, const float *k, const float *x, const float *E, const float *dZ, const int32_t *cap_off,
                                 const int32_t *e_off, int B, int C, int T, int S, int nmax, int row0_only, float *dq, float *dk, float *dx,
                                 itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && C >= 0 && C <= 65535 && T >= 0 && S >= 4 && nmax >= 1, "itr_sgt_graph_bwd");
    ITR_UNSUPPORTED(S % 4 != 0, "itr_sgt_graph_bwd: sim_dim %d is not a multiple of 4", S);
    if (B == 0 || C == 0) return ITR_OK;
    ITR_REQUIRE(q && k && x && E && dZ && cap_off && e_off && dq && dk && dx, "itr_sgt_graph_bwd: null pointer");
    const int SC = sgt_graph_chunk(nmax, S);
    const size_t lds = (2 * (size_t)((nmax * nmax + 3) & ~3) + 2 * (size_t)nmax * (SC + 4)) * sizeof(float);
    const void *kern = row0_only ? reinterpret_cast<const void *>(sgt_graph_bwd_kernel<true>) : reinterpret_cast<const void *>(sgt_graph_bwd_kernel<false>);
    const int rc = sgt_lds(kern, lds, "itr_sgt_graph_bwd");
    if (rc != ITR_OK) return rc;
    const SgtGraph g{cap_off, e_off, C, T, S, SC};
    if (row0_only) hipLaunchKernelGGL(sgt_graph_bwd_kernel<true>, dim3(C, B), dim3(256), lds, as_stream(stream), g, q, k, x, E, dZ, dq, dk, dx);
    else hipLaunchKernelGGL(sgt_graph_bwd_kernel<false>, dim3(C, B), dim3(256), lds, as_stream(stream), g, q, k, x, E, dZ, dq, dk, dx);
    ITR_CHECK_LAUNCH("sgt_graph_bwd");
    return ITR_OK;
}

extern "C" int itr_sgt_segbn_fwd(const float *a, const int32_t *cap_off, int B, int C, int T, const float *gamma, const float *beta, float eps, float *y,
                                 float *mean, float *var, float *invstd, itr_stream_t stream) {
    SGT_SHAPE(B >= 1 && C >= 0 && T >= 0, "itr_sgt_segbn_fwd");
    if (C == 0) return ITR_OK;
    ITR_REQUIRE(a && cap_off && gamma && beta && y && mean && var && invstd, "itr_sgt_segbn_fwd: null pointer");
    hipLaunchKernelGGL(sgt_segbn_fwd_kernel, dim3(C), dim3(256), 0, as_stream(stream), a, cap_off, B, C, T, gamma, beta, eps, y, mean, var, invstd);
    ITR_CHECK_LAUNCH("sgt_segbn_fwd");
    return ITR_OK;
}
extern "C" int itr_sgt_segbn_bwd(const float *dy, const float *a, const int32_t *cap_off, int B, int C, int T, const float *gamma, const float *mean,
                                 const float *invstd, float *da, float *dgamma_c, float *dbeta_c, itr_stream_t stream) {
    SGT_SHAPE(B >= 1 && C >= 0 && T >= 0, "itr_sgt_segbn_bwd");
    if (C == 0) return ITR_OK;
    ITR_REQUIRE(dy && a && cap_off && gamma && mean && invstd && da && dgamma_c && dbeta_c, "itr_sgt_segbn_bwd: null pointer");
    hipLaunchKernelGGL(sgt_segbn_bwd_kernel, dim3(C), dim3(256), 0, as_stream(stream), dy, a, cap_off, B, C, T, gamma, mean, invstd, da, dgamma_c, dbeta_c);
    ITR_CHECK_LAUNCH("sgt_segbn_bwd");
    return ITR_OK;
}

extern "C" int itr_sgt_saf_pool_fwd(const float *y, const float *nodes, const int32_t *cap_off, int B, int C, int T, int S, int nmax, float eps, float *out,
                                    itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && C >= 0 && C <= 65535 && T >= 0 && S >= 1 && nmax >= 1, "itr_sgt_saf_pool_fwd");
    if (B == 0 || C == 0) return ITR_OK;
    ITR_REQUIRE(y && nodes && cap_off && out, "itr_sgt_saf_pool_fwd: null pointer");
    hipLaunchKernelGGL(sgt_saf_pool_fwd_kernel, dim3(C, B), dim3(256), (size_t)nmax * sizeof(float), as_stream(stream), y, nodes, cap_off, C, T, S, eps, out);
    ITR_CHECK_LAUNCH("sgt_saf_pool_fwd");
    return ITR_OK;
}
extern "C" int itr_sgt_saf_pool_bwd(const float *y, const float *nodes, const float *dout, const int32_t *cap_off, int B, int C, int T, int S, int nmax,
                                    float eps, float *dy, float *dnodes, itr_stream_t stream) {
    SGT_SHAPE(B >= 0 && B <= 65535 && C >= 0 && C <= 65535 && T >= 0 && S >= 1 && nmax >= 1, "itr_sgt_saf_pool_bwd");
    if (B == 0 || C == 0) return ITR_OK;
    ITR_REQUIRE(y && nodes && dout && cap_off && dy && dnodes, "itr_sgt_saf_pool_bwd: null pointer");
    hipLaunchKernelGGL(sgt_saf_pool_bwd_kernel, dim3(C, B), dim3(256), (size_t)nmax * 5 * sizeof(float), as_stream(stream), y, nodes, dout, cap_off, C, T, S,
                       eps, dy, dnodes);
    ITR_CHECK_LAUNCH("sgt_saf_pool_bwd");
    return ITR_OK;
}

extern "C" int itr_sgt_seg_mean(const float *in, const int32_t *cap_off, int C, int D, float *out, int spread, int mean, itr_stream_t stream) {
    SGT_SHAPE(C >= 0 && C <= 65535 && D >= 1, "itr_sgt_seg_mean");
    if (C == 0) return ITR_OK;
    ITR_REQUIRE(in && cap_off && out, "itr_sgt_seg_mean: null pointer");
    const dim3 grid((unsigned)ceil_div(D, 256), (unsigned)C);
    if (spread) hipLaunchKernelGGL(sgt_seg_mean_bwd_kernel, grid, dim3(256), 0, as_stream(stream), in, cap_off, D, mean, out);
    else hipLaunchKernelGGL(sgt_seg_mean_fwd_kernel, grid, dim3(256), 0, as_stream(stream), in, cap_off, D, mean, out);
    ITR_CHECK_LAUNCH("sgt_seg_mean");
    return ITR_OK;
}
extern "C" int itr_sgt_seg_smry_fwd(const float *logit, const float *words, const int32_t *cap_off, int C, int D, int Wmax, float *p, float *out,
                                    itr_stream_t stream) {
    SGT_SHAPE(C >= 0 && D >= 1 && Wmax >= 1, "itr_sgt_seg_smry_fwd");
    if (C == 0) return ITR_OK;
    ITR_REQUIRE(logit && words && cap_off && p && out, "itr_sgt_seg_smry_fwd: null pointer");
    ITR_UNSUPPORTED(Wmax > 8192, "itr_sgt_seg_smry_fwd: captions of at most 8192 words");
    hipLaunchKernelGGL(sgt_seg_smry_fwd_kernel, dim3(C), dim3(256), (size_t)Wmax * sizeof(float), as_stream(stream), logit, words, cap_off, D, p, out);
    ITR_CHECK_LAUNCH("sgt_seg_smry_fwd");
    return ITR_OK;
}
extern "C" int itr_sgt_seg_smry_bwd(const float *p, const float *words, const float *dout, const int32_t *cap_off, int C, int D, int Wmax, float *dlogit,
                                    float *dwords, itr_stream_t stream) {
    SGT_SHAPE(C >= 0 && D >= 1 && Wmax >= 1, "itr_sgt_seg_smry_bwd");
    if (C == 0) return ITR_OK;
    ITR_REQUIRE(p && words && dout && cap_off && dlogit && dwords, "itr_sgt_seg_smry_bwd: null pointer");
    ITR_UNSUPPORTED(Wmax > 2048, "itr_sgt_seg_smry_bwd: captions of at most 2048 words");
    hipLaunchKernelGGL(sgt_seg_smry_bwd_kernel, dim3(C), dim3(256), (size_t)Wmax * 4 * sizeof(float), as_stream(stream), p, words, dout, cap_off, D, dlogit,
                       dwords);
    ITR_CHECK_LAUNCH("sgt_seg_smry_bwd");
    return ITR_OK;
}
