// Transformer building blocks of the BERT text tower (itr/modalmodule/bert.py:113-358) and of the SAEM / CAMERA
// heads.  The dense layers run on gemm_nt (fp32 MFMA, erf-GELU / tanh / relu epilogues); this file holds the
// HBM-bound glue, written as fused row kernels:
//   itr_bert_embed_ln   word + position + token-type embedding sum + TF-style LayerNorm   (bert.py:127-157, :113-126)
//   itr_add_layernorm   LayerNorm(x + residual), epsilon INSIDE the sqrt (1e-12)          (bert.py:216-220, :253-257)
//   itr_mha_small       softmax(Q K^T / sqrt(dk) + (1 - mask) * -10000) V for short sequences (L <= 64), one wave
//                       per (sequence, head), one lane per query row, online softmax       (bert.py:185-207; camera_.py:42-53)
//   itr_relu_maxpool    max over time of relu(x) with a per-sequence valid length          (TextEncoder.py:148-149)
#include <stdlib.h>
#include "itr_common.h"

namespace itr {

constexpr int LN_MAXV = 16;   // float4 per lane -> hidden size <= 4096

__device__ __forceinline__ void ln_finish(float4 (&v)[LN_MAXV], int nv, int lane, int H, const float *gamma,
                                          const float *beta, float eps, float *out) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
        if (lane + 64 * i < nv) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float u = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
        if (lane + 64 * i < nv) {
            const float a = v[i].x - u, b = v[i].y - u, c = v[i].z - u, d = v[i].w - u;
            q += a * a + b * b + c * c + d * d;
        }
    const float denom = sqrtf(wave_sum(q) / (float)H + eps);   // TF style: epsilon inside the sqrt
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) {
            const float4 g = reinterpret_cast<const float4 *>(gamma)[c4], b = reinterpret_cast<const float4 *>(beta)[c4];
            float4 o;
            o.x = g.x * ((v[i].x - u) / denom) + b.x;
            o.y = g.y * ((v[i].y - u) / denom) + b.y;
            o.z = g.z * ((v[i].z - u) / denom) + b.z;
            o.w = g.w * ((v[i].w - u) / denom) + b.w;
            reinterpret_cast<float4 *>(out)[c4] = o;
        }
    }
}

__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t *__restrict__ ids, const int64_t *__restrict__ type_ids,
                                                       const float *__restrict__ word, const float *__restrict__ pos,
                                                       const float *__restrict__ type, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float *__restrict__ out, int64_t rows,
                                                       int L, int H, int64_t V, int P, int Tv, float eps) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, nv = H >> 2;
    int64_t id = ids[row];
    int64_t ty = type_ids ? type_ids[row] : 0;
    const int t = (int)(row % L);
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);        // nn.Embedding would raise; stay memory safe
    ty = ty < 0 ? 0 : (ty >= Tv ? Tv - 1 : ty);
    const float4 *w4 = reinterpret_cast<const float4 *>(word + id * H);
    const float4 *p4 = reinterpret_cast<const float4 *>(pos + (int64_t)(t < P ? t : P - 1) * H);
    const float4 *t4 = reinterpret_cast<const float4 *>(type + ty * H);
    float4 v[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) {
            const float4 a = w4[c4], b = p4[c4], c = t4[c4];
            v[i] = make_float4(a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z, a.w + b.w + c.w);
        }
    }
    ln_finish(v, nv, lane, H, gamma, beta, eps, out + row * H);
}

__global__ __launch_bounds__(256) void add_ln_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     float *__restrict__ out, int64_t rows, int H, float eps) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, nv = H >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x + row * H);
    const float4 *r4 = res ? reinterpret_cast<const float4 *>(res + row * H) : nullptr;
    float4 v[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) {
            v[i] = x4[c4];
            if (r4) { const float4 r = r4[c4]; v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w; }
        }
    }
    ln_finish(v, nv, lane, H, gamma, beta, eps, out + row * H);
}

// One wave per (sequence b, head h); lane i < L owns query row i.  K and V of the head are staged in LDS.
template <int DK>
__global__ __launch_bounds__(64) void mha_small_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                       const float *__restrict__ v, int64_t ldq, int64_t ldk, int64_t ldv,
                                                       const float *__restrict__ mask, float *__restrict__ out, int64_t ldo,
                                                       int L, int heads, float scale) {
    __shared__ float ks[64][DK + 1];
    __shared__ float vs[64][DK + 1];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)b * L;
    for (int idx = lane; idx < L * DK; idx += 64) {
        const int j = idx / DK, d = idx % DK;
        ks[j][d] = k[(row0 + j) * ldk + h * DK + d];
        vs[j][d] = v[(row0 + j) * ldv + h * DK + d];
    }
    __syncthreads();
    if (lane >= L) return;
    float qr[DK], ctx[DK];
#pragma unroll
    for (int d = 0; d < DK; ++d) { qr[d] = q[(row0 + lane) * ldq + h * DK + d]; ctx[d] = 0.f; }
    float mx = -INFINITY, den = 0.f;
    for (int j = 0; j < L; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DK; ++d) s += qr[d] * ks[j][d];
        s = s * scale;
        if (mask) s += (1.0f - mask[row0 + j]) * -10000.0f;     // bert.py:340-341 additive mask
        const float nm = fmaxf(mx, s);
        const float corr = expf(mx - nm), p = expf(s - nm);
        den = den * corr + p;
#pragma unroll
        for (int d = 0; d < DK; ++d) ctx[d] = ctx[d] * corr + p * vs[j][d];
        mx = nm;
    }
    const float inv = 1.f / den;
#pragma unroll
    for (int d = 0; d < DK; ++d) out[(row0 + lane) * ldo + h * DK + d] = ctx[d] * inv;
}

// Matrix-core attention for short sequences: one WAVE per (sequence b, head h), WPB waves per workgroup.
//   S = Q K^T      (L x L, K = DK)   NT x NT tiles of v_mfma_f32_16x16x4_f32; Q and K staged in LDS in the plane layout
//                                    lds[k / 4][row ^ (plane & 7)] (16-byte fragment reads, 4 k-steps per read)
//   P = softmax(S * scale + mask)    in the accumulator layout: a row lives in the 16 lanes that share fg -> 4 xor-shuffles
//   ctx = P V      (L x DK, K = L)   P goes through LDS (accumulator layout -> A-operand layout), V is read row-major
// against the VALU kernel above (one lane per query row, half of the lanes idle at L = 32): 128 MFMAs = 4 096 cycles per
// (sequence, head) at L = 32, DK = 64 instead of ~16 k FMAs per lane.  Waves never synchronise with each other.
template <int DK, int NT, int WPB>
__global__ __launch_bounds__(64 * WPB) void mha_mfma_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                            const float *__restrict__ v, int64_t ldq, int64_t ldk, int64_t ldv,
                                                            const float *__restrict__ mask, float *__restrict__ out, int64_t ldo,
                                                            int L, int heads, float scale, int64_t npairs) {
    constexpr int LP = NT * 16, NPL = DK / 4, VST = DK + 16, PST = LP + 2, NDT = DK / 16;
    constexpr int V_F4 = (LP * VST + 3) / 4, P_F4 = (LP * PST + 3) / 4;
    constexpr int WAVE_F4 = V_F4 + P_F4;
    extern __shared__ float4 mha_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t pair = (int64_t)blockIdx.x * WPB + wave;
    if (pair >= npairs) return;
    const int b = (int)(pair / heads), h = (int)(pair % heads);
    const int64_t row0 = (int64_t)b * L;
    float *Vs = reinterpret_cast<float *>(mha_smem + (size_t)wave * WAVE_F4);
    float *Ps = reinterpret_cast<float *>(mha_smem + (size_t)wave * WAVE_F4 + V_F4);
    const int fi = lane & 15, fg = lane >> 4;
    // ---- Q and K straight from memory in MFMA fragment order (round 3; before, both went through LDS in a plane layout, 30 KB per
    // wave = ONE workgroup of four waves per CU, and a (sequence, head) pair was one exposed memory round trip after another at 1.4
    // TB/s): lane (fi, fg) owns the 16 bytes [4 (4 qq + fg), + 4) of row 16 t + fi -- 64 contiguous bytes per row and instruction.
    // Rows beyond L re-read row L - 1: their scores are masked (keys) or never stored (queries).  V goes to LDS row-major (the B
    // operand of P V is read by key row), zero rows beyond L.  Everything is requested before the first wait.
    float4 qf[NPL / 4][NT], kf[NPL / 4][NT];
#pragma unroll
    for (int qq = 0; qq < NPL / 4; ++qq)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = (t * 16 + fi < L) ? t * 16 + fi : L - 1;
            qf[qq][t] = *reinterpret_cast<const float4 *>(q + (row0 + row) * ldq + h * DK + 4 * (4 * qq + fg));
            kf[qq][t] = *reinterpret_cast<const float4 *>(k + (row0 + row) * ldk + h * DK + 4 * (4 * qq + fg));
        }
    for (int idx = lane; idx < LP * NPL; idx += 64) {
        const int row = idx / NPL, p = idx % NPL;
        float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < L) vv = *reinterpret_cast<const float4 *>(v + (row0 + row) * ldv + h * DK + 4 * p);
        float *vd = Vs + row * VST + 4 * p;
        vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
    }
    // ---- S = Q K^T
    f32x4 sacc[NT][NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sacc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < NPL / 4; ++qq) {
        const float4(&a)[NT] = qf[qq];
        const float4(&bb)[NT] = kf[qq];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, bb[nt].x, sacc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, bb[nt].y, sacc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, bb[nt].z, sacc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, bb[nt].w, sacc[mt][nt], 0, 0, 0);
    }
    // ---- softmax over the keys: sacc[mt][nt][j] = S[mt*16 + 4 fg + j][nt*16 + fi]
    float madd[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int key = nt * 16 + fi;
        madd[nt] = key < L ? (mask ? (1.0f - mask[row0 + key]) * -10000.0f : 0.f) : -INFINITY;   // bert.py:340-341; padding keys drop out
    }
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float mx = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                sacc[mt][nt][j] = sacc[mt][nt][j] * scale + madd[nt];
                mx = fmaxf(mx, sacc[mt][nt][j]);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float den = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                sacc[mt][nt][j] = expf(sacc[mt][nt][j] - mx);
                den += sacc[mt][nt][j];
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) den += __shfl_xor(den, o, 64);
            const float inv = 1.f / den;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sacc[mt][nt][j] *= inv;
        }
    // ---- P -> LDS in [query][key] order
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) Ps[(mt * 16 + 4 * fg + j) * PST + nt * 16 + fi] = sacc[mt][nt][j];
    // ---- ctx = P V
    f32x4 cacc[NT][NDT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) cacc[mt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < LP / 4; ++ks) {
        float a[NT], bv[NDT];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) a[mt] = Ps[(mt * 16 + fi) * PST + 4 * ks + fg];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) bv[dt] = Vs[(4 * ks + fg) * VST + dt * 16 + fi];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) cacc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv[dt], cacc[mt][dt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mt * 16 + 4 * fg + j;
            if (m < L) {
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) out[(row0 + m) * ldo + h * DK + dt * 16 + fi] = cacc[mt][dt][j];
            }
        }
}

// The same attention entirely in registers (round 3): no LDS, no staging pass.
//   S^T = K Q^T   A = K fragments, B = Q fragments, both read from memory in fragment order (lane (fi, fg): 16 bytes at column
//                 4 (4 qq + fg) of row 16 t + fi).  Accumulator j of tile (nt, mt) = S[query 16 mt + fi][key 16 nt + 4 fg + j]:
//                 a query's keys live in ONE lane column -- the softmax reduces over registers and the four 16-lane rows (two
//                 v_permlane swaps), and the normalised weights ARE the B operand of the next product (k index fg <-> key 4 fg + j).
//   ctx^T = V^T P^T   A = V[key 16 nt + 4 fg + j][d 16 dt + fi], one dword per lane straight from memory (64 contiguous bytes per
//                 key row); accumulator j' of tile (dt, mt) = ctx[query 16 mt + fi][d 16 dt + 4 fg + j'] -> ONE 16-byte store.
// Against the LDS kernel above (Q, K, V, P staged per wave: 30 KB at L <= 32 -> four waves per CU, each a chain of exposed memory
// round trips; tools/mha_bench.py, B = 3 571 sequences x 12 heads x 64): L = 12 / 30 / 45 / 64: 133 / 468 / 1 381 / 2 046 us -> 97 / 289 / 492 /
// 933 us (5.4 / 4.6 / 4.0 / 3.0 TB/s of q + k + v + context).  Rows beyond L re-read row L - 1: as keys they are masked to weight 0, as
// queries never stored.  The LDS kernel (itself 20-50 % faster since Q and K skip the LDS) remains for outputs without 16-byte rows.
__device__ __forceinline__ float mha_rows_max(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float mha_rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <int DK, int NT>
__global__ __launch_bounds__(256) void mha_reg_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                      const float *__restrict__ v, int64_t ldq, int64_t ldk, int64_t ldv,
                                                      const float *__restrict__ mask, float *__restrict__ out, int64_t ldo,
                                                      int L, int heads, float scale, int64_t npairs) {
    constexpr int NKQ = DK / 16, NDT = DK / 16;
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= npairs) return;
    const int b = (int)(pair / heads), h = (int)(pair % heads);
    const int64_t row0 = (int64_t)b * L;
    const int fi = lane & 15, fg = lane >> 4;
    int rowt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rowt[t] = (t * 16 + fi < L) ? t * 16 + fi : L - 1;
    float4 qf[NKQ][NT], kf[NKQ][NT];
#pragma unroll
    for (int qq = 0; qq < NKQ; ++qq)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            qf[qq][t] = *reinterpret_cast<const float4 *>(q + (row0 + rowt[t]) * ldq + h * DK + 4 * (4 * qq + fg));
            kf[qq][t] = *reinterpret_cast<const float4 *>(k + (row0 + rowt[t]) * ldk + h * DK + 4 * (4 * qq + fg));
        }
    // additive mask of this lane's keys (bert.py:340-341); keys beyond L drop out
    float madd[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = nt * 16 + 4 * fg + j;
            madd[nt][j] = key < L ? (mask ? (1.0f - mask[row0 + key]) * -10000.0f : 0.f) : -INFINITY;
        }
    f32x4 sacc[NT][NT];      // [key tile][query tile]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) sacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < NKQ; ++qq) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) sacc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[qq][nt].x, qf[qq][mt].x, sacc[nt][mt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) sacc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[qq][nt].y, qf[qq][mt].y, sacc[nt][mt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) sacc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[qq][nt].z, qf[qq][mt].z, sacc[nt][mt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) sacc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[qq][nt].w, qf[qq][mt].w, sacc[nt][mt], 0, 0, 0);
    }
    // V operands of the second product (requested here: in flight during the softmax)
    float vf[NT][4][NDT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = nt * 16 + 4 * fg + j;
            const float *vr = v + (row0 + (key < L ? key : L - 1)) * ldv + h * DK + fi;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) vf[nt][j][dt] = vr[16 * dt];
        }
    // softmax over the keys of query 16 mt + fi
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        float mx = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sacc[nt][mt][j] = sacc[nt][mt][j] * scale + madd[nt][j];
                mx = fmaxf(mx, sacc[nt][mt][j]);
            }
        mx = mha_rows_max(mx);
        float den = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sacc[nt][mt][j] = expf(sacc[nt][mt][j] - mx);
                den += sacc[nt][mt][j];
            }
        den = mha_rows_sum(den);
        const float inv = 1.f / den;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc[nt][mt][j] *= inv;
    }
    // ctx^T = V^T P^T
    f32x4 cacc[NDT][NT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) cacc[dt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) cacc[dt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[nt][j][dt], sacc[nt][mt][j], cacc[dt][mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        const int m = mt * 16 + fi;
        if (m < L) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
                *reinterpret_cast<float4 *>(out + (row0 + m) * ldo + h * DK + dt * 16 + 4 * fg) =
                    float4{cacc[dt][mt][0], cacc[dt][mt][1], cacc[dt][mt][2], cacc[dt][mt][3]};
        }
    }
}

template <int DK, int NT, int WPB>
static int launch_mha_mfma(const float *q, const float *k, const float *v, int64_t ldq, int64_t ldk, int64_t ldv, const float *mask,
                           float *out, int64_t ldo, int64_t B, int L, int heads, float scale, hipStream_t st) {
    constexpr int LP = NT * 16, VST = DK + 16, PST = LP + 2;
    constexpr size_t wave_bytes = (size_t)((LP * VST + 3) / 4 + (LP * PST + 3) / 4) * 16;
    constexpr size_t lds = wave_bytes * WPB;
    static bool attr = false;
    if (!attr) {
        ITR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mha_mfma_kernel<DK, NT, WPB>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const int64_t npairs = B * heads;
    hipLaunchKernelGGL((mha_mfma_kernel<DK, NT, WPB>), dim3((unsigned)ceil_div(npairs, (int64_t)WPB)), dim3(64 * WPB), lds, st, q, k, v, ldq,
                       ldk, ldv, mask, out, ldo, L, heads, scale, npairs);
    ITR_CHECK_LAUNCH("mha_mfma");
    return ITR_OK;
}

template <int DK>
static int dispatch_mha_mfma(const float *q, const float *k, const float *v, int64_t ldq, int64_t ldk, int64_t ldv, const float *mask,
                             float *out, int64_t ldo, int64_t B, int L, int heads, float scale, hipStream_t st) {
    const int nt = (L + 15) / 16;
    // the register-only kernel: 16-byte context stores (ITR_MHA_LDS=1: the LDS kernel, for A/B timing)
    if (ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && !ITR_EXP_ENV("ITR_MHA_LDS")) {
        const int64_t npairs = B * heads;
        const dim3 grid((unsigned)ceil_div(npairs, (int64_t)4));
#define ITR_MHA_REG(NT_)                                                                                                          \
        hipLaunchKernelGGL((mha_reg_kernel<DK, NT_>), grid, dim3(256), 0, st, q, k, v, ldq, ldk, ldv, mask, out, ldo, L, heads, scale, npairs)
        if (nt == 1) ITR_MHA_REG(1);
        else if (nt == 2) ITR_MHA_REG(2);
        else if (nt == 3) ITR_MHA_REG(3);
        else ITR_MHA_REG(4);
#undef ITR_MHA_REG
        ITR_CHECK_LAUNCH("mha_reg");
        return ITR_OK;
    }
    if (nt == 1) return launch_mha_mfma<DK, 1, 4>(q, k, v, ldq, ldk, ldv, mask, out, ldo, B, L, heads, scale, st);
    if (nt == 2) return launch_mha_mfma<DK, 2, 4>(q, k, v, ldq, ldk, ldv, mask, out, ldo, B, L, heads, scale, st);
    if (nt == 3) return launch_mha_mfma<DK, 3, 2>(q, k, v, ldq, ldk, ldv, mask, out, ldo, B, L, heads, scale, st);
    return launch_mha_mfma<DK, 4, 2>(q, k, v, ldq, ldk, ldv, mask, out, ldo, B, L, heads, scale, st);
}

// out[b, c] = max_{t < valid} relu(x[b, t, c])
__global__ __launch_bounds__(256) void relu_maxpool_kernel(const float *__restrict__ x, int L, int C, int valid,
                                                           float *__restrict__ out, int64_t ldo) {
    const int64_t b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float m = 0.f;   // relu >= 0
    for (int t = 0; t < valid; ++t) m = fmaxf(m, x[(b * L + t) * (int64_t)C + c]);
    out[b * ldo + c] = m;
}

}  // namespace itr

extern "C" int itr_bert_embed_ln(const int64_t *ids, const int64_t *type_ids, const float *word_emb, const float *pos_emb,
                                 const float *type_emb, const float *gamma, const float *beta, float *out, int64_t B,
                                 int L, int H, int64_t vocab, int max_pos, int type_vocab, float eps,
                                 itr_stream_t stream) {
    ITR_REQUIRE(ids && word_emb && pos_emb && type_emb && gamma && beta && out, "itr_bert_embed_ln: null pointer");
    ITR_REQUIRE(B >= 0 && L >= 1 && H >= 4 && vocab >= 1 && max_pos >= 1 && type_vocab >= 1, "itr_bert_embed_ln: bad shape");
    ITR_UNSUPPORTED(H % 4 != 0 || H > 64 * 4 * itr::LN_MAXV, "itr_bert_embed_ln: hidden size must be a multiple of 4, <= 4096");
    ITR_REQUIRE(L <= max_pos, "itr_bert_embed_ln: sequence length %d exceeds max_position_embeddings %d", L, max_pos);
    if (B == 0) return ITR_OK;
    const int64_t rows = B * L;
    hipLaunchKernelGGL(itr::embed_ln_kernel, dim3((unsigned)itr::ceil_div(rows, 4)), dim3(256), 0, itr::as_stream(stream), ids,
                       type_ids, word_emb, pos_emb, type_emb, gamma, beta, out, rows, L, H, vocab, max_pos, type_vocab, eps);
    ITR_CHECK_LAUNCH("embed_ln");
    return ITR_OK;
}

extern "C" int itr_add_layernorm(const float *x, const float *residual, const float *gamma, const float *beta, float *out,
                                 int64_t rows, int H, float eps, itr_stream_t stream) {
    ITR_REQUIRE(x && gamma && beta && out, "itr_add_layernorm: null pointer");
    ITR_REQUIRE(rows >= 0 && H >= 4, "itr_add_layernorm: bad shape");
    ITR_UNSUPPORTED(H % 4 != 0 || H > 64 * 4 * itr::LN_MAXV, "itr_add_layernorm: hidden size must be a multiple of 4, <= 4096");
    if (rows == 0) return ITR_OK;
    hipLaunchKernelGGL(itr::add_ln_kernel, dim3((unsigned)itr::ceil_div(rows, 4)), dim3(256), 0, itr::as_stream(stream), x, residual,
                       gamma, beta, out, rows, H, eps);
    ITR_CHECK_LAUNCH("add_layernorm");
    return ITR_OK;
}

extern "C" int itr_mha_small(const float *q, const float *k, const float *v, int64_t ldq, int64_t ldk, int64_t ldv,
                             const float *mask, float *out, int64_t ldo, int64_t B, int L, int heads, int dk,
                             float scale, itr_stream_t stream) {
    ITR_REQUIRE(q && k && v && out, "itr_mha_small: null pointer");
    ITR_REQUIRE(B >= 0 && L >= 1 && heads >= 1, "itr_mha_small: bad shape");
    ITR_UNSUPPORTED(L > 64, "itr_mha_small: sequences of at most 64 positions (got %d)", L);
    ITR_UNSUPPORTED(dk != 16 && dk != 32 && dk != 64, "itr_mha_small: head size must be 16, 32 or 64 (got %d)", dk);
    ITR_REQUIRE(B * heads < 0x7fffffffLL, "itr_mha_small: grid too large");
    if (B == 0) return ITR_OK;
    const dim3 grid((unsigned)(B * heads));
    hipStream_t st = itr::as_stream(stream);
    // matrix-core path: 16-byte aligned head slices (every fused-QKV / separate-projection layout of the path)
    const bool al = (ldq % 4 == 0) && (ldk % 4 == 0) && (ldv % 4 == 0) && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (al && !ITR_EXP_ENV("ITR_MHA_VALU")) {
        if (dk == 16) return itr::dispatch_mha_mfma<16>(q, k, v, ldq, ldk, ldv, mask, out, ldo, B, L, heads, scale, st);
        if (dk == 32) return itr::dispatch_mha_mfma<32>(q, k, v, ldq, ldk, ldv, mask, out, ldo, B, L, heads, scale, st);
        return itr::dispatch_mha_mfma<64>(q, k, v, ldq, ldk, ldv, mask, out, ldo, B, L, heads, scale, st);
    }
    if (dk == 16) hipLaunchKernelGGL(itr::mha_small_kernel<16>, grid, dim3(64), 0, st, q, k, v, ldq, ldk, ldv, mask, out, ldo, L, heads, scale);
    else if (dk == 32) hipLaunchKernelGGL(itr::mha_small_kernel<32>, grid, dim3(64), 0, st, q, k, v, ldq, ldk, ldv, mask, out, ldo, L, heads, scale);
    else hipLaunchKernelGGL(itr::mha_small_kernel<64>, grid, dim3(64), 0, st, q, k, v, ldq, ldk, ldv, mask, out, ldo, L, heads, scale);
    ITR_CHECK_LAUNCH("mha_small");
    return ITR_OK;
}

extern "C" int itr_relu_maxpool(const float *x, float *out, int64_t ldo, int64_t B, int L, int C, int valid,
                                itr_stream_t stream) {
    ITR_REQUIRE(x && out, "itr_relu_maxpool: null pointer");
    ITR_REQUIRE(B >= 0 && B <= 65535 && L >= 1 && C >= 1 && valid >= 1 && valid <= L && ldo >= C, "itr_relu_maxpool: bad shape");
    if (B == 0) return ITR_OK;
    hipLaunchKernelGGL(itr::relu_maxpool_kernel, dim3((unsigned)itr::ceil_div(C, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), x,
                       L, C, valid, out, ldo);
    ITR_CHECK_LAUNCH("relu_maxpool");
    return ITR_OK;
}
