// SCAN stacked cross attention similarity (xattn_score_t2i / xattn_score_i2t,
// itr/modalmodule/Objectives.py:329-476) as ONE fused gfx950 kernel.
//
// The reference loops over captions in Python; per (image i, caption c) pair it does
//     A      = V_i E_c^T                       [R x W]   (bmm, K = D)
//     Ahat   = raw_feature_norm(A)             (LeakyReLU(0.1) + l2norm over the query axis ...)
//     P      = softmax(lambda_s * Ahat)        over the context axis
//     ctx    = P * context                     [W x D] (t2i) or [R x D] (i2t)   (bmm, K = R or W)
//     sim    = cosine(query, ctx)  per query row;   score = LSE / max / sum / mean over rows.
//
// Mapping: all pairs together are one big fp32 GEMM  (Ni*R) x (sum W) x D  whose 36 x W blocks
// never leave the chip.  A workgroup owns 4 images (144 rows = 9 MFMA row tiles of 16) x one
// column tile of <= 64 words (whole captions only, planned host-side), accumulates the raw
// dot products with v_mfma_f32_16x16x4_f32 (exact fp32), parks the 144 x 64 block in LDS and
// finishes every pair in the epilogue.
//
// The second bmm is never executed.  With G_i = V_i V_i^T (36 x 36 Gram matrix per image,
// precomputed once, 5 KB/image) and the raw A already on chip:
//     query . ctx_w   = sum_r P[r,w] * A[r,w]
//     || ctx_w ||^2   = P[:,w]^T G_i P[:,w]
// (t2i; for i2t the caption Gram H_c = E_c E_c^T plays the same role).  This removes half of
// the reference's FLOPs (the 2*W*R*D context GEMM and the 3 D-long reductions per word) and all
// of its HBM traffic for `ctx`; the result differs from the literal evaluation order only by
// fp32 rounding (tests bound it at 2e-5 absolute on scores in (-1, 1)).
//
// LDS operand layout (both operands K-contiguous): 8 planes of float4 per 32-wide K chunk,
// lds[plane][row ^ plane]; see gemm_f32.hip for the bank-conflict argument (the 16x16x4 MFMA
// lane groups need plane = 4q + (lane >> 4) so both planes of a ds_read_b128 lane group share
// bits [3:2] of the XOR).
#include <stdlib.h>
#include <type_traits>
#include <vector>

#include "scan_common.h"
#include "pack_plan.h"

namespace itr {

struct ScanArgs {
    const float *img;        // [Ni, 36, D]
    const float *wtiled;     // [n_tiles * 64, D]   words re-packed tile by tile (zero rows = padding)
    const uint16_t *img_bf;  // bf16x3 variant: split planes of img,    [Ni * 36][D / 32][hi | lo][32] bf16 (scan_split_rows_kernel)
    const uint16_t *wt_bf;   //                 split planes of wtiled, [n_tiles * 64][D / 32][hi | lo][32]
    const ScanTileMeta *meta;  // [n_tiles]
    const float *gram;       // [Ni, 36, 36]        (t2i)   V_i V_i^T in upper-triangular form (diagonal + 2 x upper part)
    const float *wnorm;      // [n_tiles * 64]      (t2i)   ||E_w|| per tiled column
    const float *vnorm;      // [Ni * 36]           (i2t)   ||V_r||
    const float *cgram;      // [sum W_c^2]         (i2t)   E_c E_c^T, caption c at cgram_off[c]
    const int64_t *cgram_off;  // [Nc]
    const float *hblk;       // [n_tiles, 64, 64]   (i2t)   per column tile: the block-diagonal Gram of its captions, zero elsewhere
    float *S;
    int64_t ldS;
    int64_t Ni, Nc, n_tiles;
    int D;
    int mode, norm, agg;
    float lambda_softmax, lambda_lse;
    float *emit_p;    // [Ni, n_tiles*64, 36] normalised attention weights (SGRAF: SCAN_attention), or null
    float *emit_cn;   // [Ni, n_tiles*64]     1 / (||ctx|| + eps)
    int tpw;    // tiles per workgroup (see the work mapping in the kernel)
    int debug;  // ablation switches for tools/scan_ablate.py (env ITR_SCAN_DEBUG); 0 in production
    unsigned long long *dbg_cycles;  // [8] phase cycle sums (debug & 16), normally null
};

struct ScanSmem {
    union {
        float4 stage[2][SC_PLANES][SC_ROWS];  // 57,344 B   main loop operand staging
        float arawt[SC_NT][SC_LDT];           // 37,888 B   parked raw dot products, [word column][region row]
    };
    union {
        float stat[2][SC_MAXCAP][SC_LDT];     // 18,944 B   t2i: per (caption, row) norm statistics s0 / s1
        float rsim2[SC_MT][SC_MAXCAP];        //  9,216 B   i2t: per (region row, caption) term
    };
    float colstat[SC_IMGS][SC_NT][2];         //  2,048 B   i2t: per (image, word) norm statistics
    float rowsim[SC_IMGS][SC_NT];             //  1,024 B   t2i: per (image, word) similarity term
    ScanTileMeta meta;                        //    256 B
};

// value of the normalised attention logit b (before * lambda_softmax) from the raw a and the
// statistics of its normalisation group (Objectives.py:436-457)
template <int NORM>
__device__ __forceinline__ float norm_apply_c(float a, float s0, float s1) {
    if (NORM == 0) return leaky(a) * s0;
    if (NORM == 1) return a * s0;
    if (NORM == 2) return expf(a - s0) * s1;
    if (NORM == 3) return a;
    if (NORM == 4) return leaky(a);
    if (NORM == 5) return a * s0;
    return leaky(a) * s0;
}
// Runs f(std::integral_constant<int, norm>) so that the per-element code is specialised at compile time: with a
// run-time `norm` hipcc emits a tree of scalar branches (and an lgkmcnt(0)) PER ELEMENT.
template <typename F>
__device__ __forceinline__ void dispatch_norm(int norm, F &&f) {
    switch (norm) {
        case 0: f(std::integral_constant<int, 0>{}); break;
        case 1: f(std::integral_constant<int, 1>{}); break;
        case 2: f(std::integral_constant<int, 2>{}); break;
        case 3: f(std::integral_constant<int, 3>{}); break;
        case 4: f(std::integral_constant<int, 4>{}); break;
        case 5: f(std::integral_constant<int, 5>{}); break;
        default: f(std::integral_constant<int, 6>{}); break;
    }
}

__device__ __forceinline__ float norm_apply(float a, int norm, float s0, float s1) {
    switch (norm) {
        case 0: return leaky(a) * s0;             // clipped_l2norm: s0 = 1 / (sqrt(sum leaky^2) + eps)
        case 1: return a * s0;                    // l2norm
        case 2: return expf(a - s0) * s1;         // softmax: s0 = max, s1 = 1 / sum exp
        case 3: return a;                         // no_norm
        case 4: return leaky(a);                  // clipped
        case 5: return a * s0;                    // l1norm: s0 = 1 / (sum |a| + eps)
        default: return leaky(a) * s0;            // clipped_l1norm
    }
}

// accumulate / finish the statistics of one normalisation group
struct NormAcc {
    float s0, s1;
    __device__ __forceinline__ void init(int norm) { s0 = (norm == 2) ? -INFINITY : 0.f; s1 = 0.f; }
    __device__ __forceinline__ void pass1(float a, int norm) {
        switch (norm) {
            case 0: { const float b = leaky(a); s0 += b * b; break; }
            case 1: s0 += a * a; break;
            case 2: s0 = fmaxf(s0, a); break;
            case 5: s0 += fabsf(a); break;
            case 6: s0 += fabsf(leaky(a)); break;
            default: break;
        }
    }
    __device__ __forceinline__ void pass2(float a, int norm) {
        if (norm == 2) s1 += expf(a - s0);
    }
    __device__ __forceinline__ void finish(int norm) {
        // stored as reciprocals: the consumers multiply (<= 1 ulp away from the reference's division)
        if (norm == 0 || norm == 1) s0 = 1.f / (sqrtf(s0) + 1e-8f);
        else if (norm == 5 || norm == 6) s0 = 1.f / (s0 + 1e-8f);
        else if (norm == 2) s1 = 1.f / s1;
    }
};

using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
using f16x8_t = __attribute__((ext_vector_type(8))) _Float16;

// PREC 0: exact fp32 main loop (scan_mainloop.inc).  PREC 1: split-bf16 "bf16x3", PREC 3: split-fp16 "fp16x3" main loop
// (scan_mainloop_bf16.inc; opt-in, reported separately -- STUDY_SPLIT_PRECISION.md); bits 2 / 3: ablation builds.  The epilogue is shared.
// XA: the attention direction as a compile-time constant for the exact fp32 build (0 = t2i, 1 = i2t: two kernels, so the i2t epilogue's
// registers and scalar spills do not weigh on the t2i kernel -- adding 60 lines to the i2t branch cost the t2i kernel 0.3 % while both
// lived in one function); -1 = g.mode at run time (the study variants).
template <int PREC, int XA>
__device__ __forceinline__ void scan_xattn_body(const ScanArgs &g) {
    const int mode = XA >= 0 ? XA : g.mode;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ScanSmem &sm = *reinterpret_cast<ScanSmem *>(smem_raw);

    // ---- XCD-aware work mapping, `tpw` tiles per workgroup.  Workgroup b lives on XCD b % 8 (round-robin dispatch); each XCD
    // walks 8 x 8 patches of (image tile, column tile) so that the ~64 workgroups resident on it share operand panels in its
    // private L2.  A workgroup handles the SAME position of `tpw` consecutive patches of its XCD, one after the other: the
    // resident workgroups still sit on one patch at a time, while the cost of launching a workgroup (its slot stays empty for
    // several microseconds; residency 0.90 with one 75 us tile per workgroup) is spread over tpw tiles.  Workgroups are still
    // launched as slots free up, so the two workgroups of a CU stay out of phase -- one multiplies while the other runs its
    // epilogue; a fully persistent grid (all workgroups started together) locks them in phase and measured 7 % slower.
    const int64_t img_tiles = (g.Ni + SC_IMGS - 1) / SC_IMGS;
    const int64_t PI = (img_tiles + 7) / 8, PJ = (g.n_tiles + 7) / 8;
    const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int64_t pgroup = slot >> 6;
    const int within = (int)(slot & 63);
    int tid_ = threadIdx.x;
  for (int rep = 0; rep < g.tpw; ++rep) {
    const int64_t patch = (pgroup * g.tpw + rep) * 8 + xcd;
    if (patch >= PI * PJ) break;
    const int64_t it = (patch / PJ) * 8 + (within & 7);
    const int64_t ct = (patch % PJ) * 8 + (within >> 3);
    if (it >= img_tiles || ct >= g.n_tiles) continue;
    const int64_t img0 = it * SC_IMGS;
    // everything per-lane is re-derived from this opaque copy each tile: otherwise hipcc hoists the (loop-invariant) address
    // arithmetic out of the tile loop and its live ranges then span main loop AND epilogue -- spills at this VGPR budget
    asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __syncthreads();   // the previous tile's epilogue is done with the LDS block (meta, parked scores)

    unsigned long long tick_ = g.dbg_cycles ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long real0_ = g.dbg_cycles ? __builtin_amdgcn_s_memrealtime() : 0ull;   // 100 MHz
    // t2i: ||E_w|| of this lane's word column is only needed at the very end -- fetch it now
    const float wnorm_pre = (mode == 0) ? g.wnorm[ct * SC_NT + lane] : 0.f;
    // ---- tile metadata: needed by the epilogue only, so its load overlaps the main loop
    if (tid < 64) reinterpret_cast<int32_t *>(&sm.meta)[tid] = reinterpret_cast<const int32_t *>(g.meta + ct)[tid];

    const int fi = lane & 15, fg = lane >> 4;
    if constexpr (PREC == 0) {
#include "scan_mainloop.inc"
    } else {
#include "scan_mainloop_bf16.inc"
    }

    const int norm = g.norm;
    const float ls = g.lambda_softmax;
    const int ncap = sm.meta.ncap;
    if (g.debug & 1) {  // ablation: no epilogue
        if (tid < SC_IMGS && img0 + tid < g.Ni) g.S[(img0 + tid) * g.ldS + sm.meta.cap_id[0]] = sm.arawt[0][tid * SC_R];
        continue;
    }
#define AT(row, col) sm.arawt[col][row]

    if (mode == 0) {
        // ================= t2i: words attend over the 36 regions of every image ============
        // The Gram matrix of this wave's image is the A operand of the ||ctx||^2 product; fetch its fragments
        // now (27 floats per lane), the latency hides behind E1.
        // Fragment (row tile mt, lane (fi, fg)): G[mt*16 + fi][16u + 4fg .. +3] (u = 0, 1) and G[..][32 + fg].
        float4 gfa[3][2];
        float gfb[3];
        {
            const int64_t gimg = (img0 + wave < g.Ni) ? img0 + wave : img0;
            const float *G = g.gram + gimg * (SC_R * SC_R);
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                int row = mt * 16 + fi;
                row = row < SC_R ? row : SC_R - 1;       // rows 36..47 of the last tile: their outputs are ignored
                // G is stored upper-triangular (gram_kernel, upper2): k-blocks left of row tile mt are zero and are skipped
                if (mt < 1) gfa[mt][0] = *reinterpret_cast<const float4 *>(G + row * SC_R + 4 * fg);
                if (mt < 2) gfa[mt][1] = *reinterpret_cast<const float4 *>(G + row * SC_R + 16 + 4 * fg);
                gfb[mt] = G[row * SC_R + 32 + fg];
            }
        }
        // E1: statistics of the first normalisation, along each caption's words, per (region row, caption).
        // From here on the epilogue is WAVE-LOCAL: wave ii owns image img0 + ii -- its 36 rows of the parked block, its
        // statistics, its similarity terms -- so E1 -> E2 -> E3 need no workgroup barrier (LDS operations of one wave
        // execute in order), and the four waves carry the same load.  That matters more than the instruction count: a
        // main-loop wave of the co-resident workgroup that shares its SIMD with a busy epilogue wave falls behind, and
        // its three sibling waves then idle at their per-chunk barrier -- unbalanced epilogue work is paid four times.
        // E1: statistics of the first normalisation, along each caption's words, per (region row, caption):
        // sum_w f(a[row][w]), f = leaky^2 | a^2 | |a| | |leaky|.  One lane per region row (36 of 64) walks the tile's columns
        // caption by caption (bounds are wave-uniform: scalar loop control), ~3 vector instructions per element.  (Round 1
        // ran this as an MFMA product with a [64 x 16] indicator matrix: 6 % of its multiply-adds useful.)
        // The reciprocal is stored multiplied by lambda_softmax * log2(e): E2a needs nothing else (see there).
        const bool img_ok = img0 + wave < g.Ni;      // wave-uniform
        if (img_ok && lane < SC_R) {
            const int row = wave * SC_R + lane;
            const float *colbase = &sm.arawt[0][row];          // (row, column c) at colbase[c * SC_LDT]
            if (norm == 0 || norm == 1 || norm == 5 || norm == 6) {
                const float ls_log2e = ls * 1.44269504088896341f;
                dispatch_norm(norm, [&](auto NC) {
                    constexpr int NORM = decltype(NC)::value;
                    for (int k = 0; k < ncap; ++k) {
                        const int c0 = __builtin_amdgcn_readfirstlane(sm.meta.cap_start[k]);
                        const int c1 = __builtin_amdgcn_readfirstlane(sm.meta.cap_start[k + 1]);
                        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four independent chains: the loop is LDS-latency bound
                        int c = c0;
                        for (; c + 3 < c1; c += 4) {
                            float v0 = colbase[c * SC_LDT], v1 = colbase[(c + 1) * SC_LDT];
                            float v2 = colbase[(c + 2) * SC_LDT], v3 = colbase[(c + 3) * SC_LDT];
                            if (NORM == 0 || NORM == 6) { v0 = leaky(v0); v1 = leaky(v1); v2 = leaky(v2); v3 = leaky(v3); }
                            s0 += (NORM <= 1) ? v0 * v0 : fabsf(v0);
                            s1 += (NORM <= 1) ? v1 * v1 : fabsf(v1);
                            s2 += (NORM <= 1) ? v2 * v2 : fabsf(v2);
                            s3 += (NORM <= 1) ? v3 * v3 : fabsf(v3);
                        }
                        for (; c < c1; ++c) {
                            float v0 = colbase[c * SC_LDT];
                            if (NORM == 0 || NORM == 6) v0 = leaky(v0);
                            s0 += (NORM <= 1) ? v0 * v0 : fabsf(v0);
                        }
                        s0 += s2;
                        s1 += s3;
                        const float sum = s0 + s1;
                        sm.stat[0][k][row] = fast_rcp((NORM <= 1 ? fast_sqrt(sum) : sum) + 1e-8f) * ls_log2e;
                    }
                });
            } else if (norm == 2) {
                for (int k = 0; k < ncap; ++k) {
                    const int c0 = __builtin_amdgcn_readfirstlane(sm.meta.cap_start[k]);
                    const int c1 = __builtin_amdgcn_readfirstlane(sm.meta.cap_start[k + 1]);
                    NormAcc na;
                    na.init(norm);
                    for (int c = c0; c < c1; ++c) na.pass1(colbase[c * SC_LDT], norm);
                    for (int c = c0; c < c1; ++c) na.pass2(colbase[c * SC_LDT], norm);
                    na.finish(norm);
                    sm.stat[0][k][row] = na.s0;
                    sm.stat[1][k][row] = na.s1;
                }
            }
        }
        SC_TICK(2)   // E1
        // E2: one wave per image, one lane per word column.
        //   (a) attention weights e = exp(lambda_s * b [- max]); the max shift is skipped whenever the first
        //       normalisation bounds |b| <= 1 (all l2 / l1 / softmax forms; exp(+-lambda_s) is harmless in fp32);
        //       e overwrites the raw dot products of this lane's column;
        //   (b) ||ctx||^2 = e^T G e / den^2:  T = G' E on the matrix core (G' = upper-triangular form of G: 60 MFMAs per wave
        //       instead of the 108 of the full 3 x 4 tiles x K = 36), then a 12-term dot per lane and a 4-lane reduction.
        {
            const int ii = wave;
            const int w = lane;
            const int64_t img = img0 + ii;
            const int k = sm.meta.col_cap[w];
            float simv = 0.f;
            if (img < g.Ni) {  // wave-uniform
                const int kk = k < 0 ? 0 : k;
                float *colp = &sm.arawt[w][ii * SC_R];
                float a[SC_R], e[SC_R];
                float den = 0.f, num = 0.f;
                dispatch_norm(norm, [&](auto NC) {
                    constexpr int NORM = decltype(NC)::value;
                    // l2 / l1 first norms: E1 stored 1/(norm + eps) already multiplied by lambda_softmax * log2(e), so the
                    // attention weight is one v_exp_f32 of (leaky(a) | a) * stat -- two multiplies fewer per element
                    constexpr bool FOLDED = (NORM == 0 || NORM == 1 || NORM == 5 || NORM == 6);
                    float mx = 0.f;
#pragma unroll
                    for (int r4 = 0; r4 < SC_R / 4; ++r4) {
                        const f32x4 av = *reinterpret_cast<const f32x4 *>(colp + 4 * r4);
                        f32x4 s0 = f32x4{1.f, 1.f, 1.f, 1.f}, s1 = s0;
                        if (NORM != 3 && NORM != 4) s0 = *reinterpret_cast<const f32x4 *>(&sm.stat[0][kk][ii * SC_R + 4 * r4]);
                        if (NORM == 2) s1 = *reinterpret_cast<const f32x4 *>(&sm.stat[1][kk][ii * SC_R + 4 * r4]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            a[4 * r4 + j] = av[j];
                            e[4 * r4 + j] = FOLDED ? norm_apply_c<NORM>(av[j], s0[j], s1[j]) : norm_apply_c<NORM>(av[j], s0[j], s1[j]) * ls;
                        }
                    }
                    if (NORM == 3 || NORM == 4) {
                        mx = e[0];
#pragma unroll
                        for (int r = 1; r < SC_R; ++r) mx = fmaxf(mx, e[r]);
                    }
#pragma unroll
                    for (int r = 0; r < SC_R; ++r) {
                        e[r] = FOLDED ? __builtin_amdgcn_exp2f(e[r]) : fast_exp(e[r] - mx);
                        den += e[r];
                        num += e[r] * a[r];
                    }
                });
#pragma unroll
                for (int r4 = 0; r4 < SC_R / 4; ++r4)
                    *reinterpret_cast<f32x4 *>(colp + 4 * r4) = f32x4{e[4 * r4], e[4 * r4 + 1], e[4 * r4 + 2], e[4 * r4 + 3]};
                const float rden = fast_rcp(den);
                SC_TICK(5)   // E2a: weights
                // T = G E.  B operand (k = region r2, n = word): lane (fi, fg) feeds E[16u + 4fg + j][nt*16 + fi].
                // Only this wave touches rows ii*36 .. +35, and LDS operations of one wave are ordered.
                f32x4 tacc[3][4];
#pragma unroll
                for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) tacc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float *ec = &sm.arawt[nt * 16 + fi][ii * SC_R];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const f32x4 bv = *reinterpret_cast<const f32x4 *>(ec + 16 * u + 4 * fg);
                        // row tile mt only meets k-blocks u >= mt (upper-triangular G): 60 MFMAs per wave instead of 108
#pragma unroll
                        for (int mt = 0; mt <= u; ++mt) tacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gfa[mt][u].x, bv[0], tacc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt <= u; ++mt) tacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gfa[mt][u].y, bv[1], tacc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt <= u; ++mt) tacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gfa[mt][u].z, bv[2], tacc[mt][nt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt <= u; ++mt) tacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gfa[mt][u].w, bv[3], tacc[mt][nt], 0, 0, 0);
                    }
                    const float b4 = ec[32 + fg];
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt)
                        tacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gfb[mt], b4, tacc[mt][nt], 0, 0, 0);
                }
                SC_TICK(6)   // E2b: T = G E
                // q[nt] = sum_r E[r][col] * T[r][col], col = nt*16 + fi; this lane holds rows mt*16 + 4fg + j
                float qn[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float *ec = &sm.arawt[nt * 16 + fi][ii * SC_R];
                    const f32x4 e0 = *reinterpret_cast<const f32x4 *>(ec + 4 * fg);
                    const f32x4 e1 = *reinterpret_cast<const f32x4 *>(ec + 16 + 4 * fg);
                    const f32x4 e2 = *reinterpret_cast<const f32x4 *>(ec + 32);          // rows 32..35, used by fg == 0
                    // (explicit fmaf chains: with "a * b + c * d" hipcc is free to contract either product, and it chose differently
                    // for different column tiles nt in one build of round 3 -- a word's term then depended on the tile column its
                    // caption was packed into, 1 ulp, and the sharded evaluation's partitions were no longer bit-identical)
                    float part = 0.f, tail = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        part = fmaf(e0[j], tacc[0][nt][j], part);
                        part = fmaf(e1[j], tacc[1][nt][j], part);
                        tail = fmaf(e2[j], tacc[2][nt][j], tail);
                    }
                    part += (fg == 0) ? tail : 0.f;
#ifdef ITR_SCAN_SHFL_LDS            // A/B build: the round-3 form (two LDS-crossbar round trips per column tile)
                    part += __shfl_xor(part, 16, 64);
                    part += __shfl_xor(part, 32, 64);
#else
                    part = xor16_add(part);       // (bit-identical to the shuffles: see scan_common.h)
                    part = xor32_add(part);
#endif
                    qn[nt] = part;
                }
                // this lane's own column is w = fg*16 + fi -> n-tile fg
                float q = fg == 0 ? qn[0] : (fg == 1 ? qn[1] : (fg == 2 ? qn[2] : qn[3]));
                q = q * rden * rden;
                num *= rden;
                if (g.emit_p) {   // SGRAF: hand the attention weights and the context norm to the GEMM chain
                    const int64_t orow = img * (g.n_tiles * SC_NT) + ct * SC_NT + w;
                    f32x4 *pd = reinterpret_cast<f32x4 *>(g.emit_p + orow * SC_R);
#pragma unroll
                    for (int r4 = 0; r4 < SC_R / 4; ++r4)
                        pd[r4] = f32x4{e[4 * r4] * rden, e[4 * r4 + 1] * rden, e[4 * r4 + 2] * rden, e[4 * r4 + 3] * rden};
                    g.emit_cn[orow] = 1.f / (sqrtf(fmaxf(q, 0.f)) + 1e-8f);
                }
                const float w1 = wnorm_pre;
                const float w2 = fast_sqrt(fmaxf(q, 0.f));
                simv = num * fast_rcp(fmaxf(w1 * w2, 1e-8f));   // cosine_similarity, Objectives.py:10-15
            }
            // E3: aggregate over the words of each caption (Objectives.py:355-366), still inside the wave: every lane turns
            // its word's term into the summand (one exp per lane for LogSumExp), lanes 0..15 then fold their caption's words
            // (segmented inclusive scan over the 64 lanes: six shuffle steps; lane w adds the value of lane w - off while that lane is
            // still inside w's caption; the caption's last lane then holds its total.  Round 2 first had lanes 0..15 walk their
            // caption's columns in LDS -- up to 20 dependent round trips.)
            {
                const int kcap = sm.meta.col_cap[w];                                     // -1: padding column
                const int cbeg = kcap >= 0 ? sm.meta.cap_start[kcap] : w;
                const int cend = kcap >= 0 ? sm.meta.cap_start[kcap + 1] : w + 1;
                float r = (g.agg == 0) ? fast_exp(simv * g.lambda_lse) : simv;
#pragma unroll
                for (int off = 1; off < SC_NT; off <<= 1) {
                    const float o = __shfl_up(r, off, 64);
                    const bool in = (w - off) >= cbeg;
                    if (g.agg == 1) r = in ? fmaxf(r, o) : r;
                    else r = in ? r + o : r;
                }
                if (img < g.Ni && kcap >= 0 && w == cend - 1) {
                    if (g.agg == 0) r = fast_log(r) / g.lambda_lse;
                    else if (g.agg == 3) r /= (float)(cend - cbeg);
                    g.S[img * g.ldS + sm.meta.cap_id[kcap]] = r;
                }
            }
        }
        SC_TICK(3)   // E2 + E3
        if (g.dbg_cycles && tid == 0) atomicAdd(&g.dbg_cycles[7], __builtin_amdgcn_s_memrealtime() - real0_);
    } else {
        // ================= i2t: regions attend over the words of every caption ============
        float *hbuf = reinterpret_cast<float *>(smem_raw + sizeof(sm.arawt));     // 19 KB of the staging area behind the parked block
        static_assert(sizeof(sm.arawt) + SC_NT * SC_NT * 4 <= sizeof(sm.stage), "caption Gram fits behind the parked block");
        const float *rsim = &sm.rsim2[0][0];
        // Matrix-core path: every first norm that bounds |b| <= 1 (no max shift needed for exp(ls b), ls <= 60).
        const bool mfma_path = (norm == 0 || norm == 1 || norm == 2 || norm == 5 || norm == 6) && fabsf(ls) <= 60.f;
        // E1: first normalisation runs along the 36 regions (query axis), per (image, word): lane = word, its 36 raw scores are
        // nine 16-byte reads of the parked column (conflict-free: the 592-byte column pitch puts 16 lanes on 64 distinct banks)
        {
            const int ii = wave, w = lane;
            if (norm != 3 && norm != 4) {
                f32x4 col[SC_R / 4];
#pragma unroll
                for (int q = 0; q < SC_R / 4; ++q) col[q] = *reinterpret_cast<const f32x4 *>(&AT(ii * SC_R + 4 * q, w));
                dispatch_norm(norm, [&](auto NC) {
                    constexpr int NORM = decltype(NC)::value;
                    NormAcc na;
                    na.init(NORM);
#pragma unroll
                    for (int q = 0; q < SC_R / 4; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) na.pass1(col[q][c], NORM);
                    if (NORM == 2) {
#pragma unroll
                        for (int q = 0; q < SC_R / 4; ++q)
#pragma unroll
                            for (int c = 0; c < 4; ++c) na.pass2(col[q][c], NORM);
                    }
                    na.finish(NORM);
                    // matrix-core path: the reciprocal norm is stored multiplied by lambda_softmax * log2(e), the weight is then ONE
                    // v_exp_f32 of f(a) * stat (as in the t2i epilogue)
                    if (mfma_path && NORM != 2) na.s0 *= ls * 1.44269504088896341f;
                    sm.colstat[ii][w][0] = na.s0;
                    sm.colstat[ii][w][1] = na.s1;
                });
            }
        }
        __syncthreads();
        SC_TICK(2)   // E1 (i2t): per (image, word) statistics over the regions
        if (mfma_path) {
            // One pass per 16-row tile, entirely in the wave's registers (round 3; before: weights written back to the parked block,
            // T = E Hblk read them again as A fragments, E o T written back and read a third time for the sums, a zero-filled and
            // gathered Hblk in LDS, four barriers -- the epilogue shares the CU's issue slots with the co-resident workgroup's main
            // loop, so every instruction removed is time):
            //   P1  e = exp(ls b) for (row fi, word 16 u + 4 fg + j), u, j = 0..3: sixteen registers that are at once the A fragments of
            //       den / num = E * indicator, the B fragments of the next product and the factors of the last one
            //   P3  T^T = Hblk E^T: A = Hblk rows as they lie in memory (lane (fi, fg): 16 bytes at column 16 u + 4 fg of row 16 nt +
            //       fi; Hblk is symmetric), k slot fg of step 4 u + j <-> word 16 u + 4 fg + j, i.e. B = e[4 u + j].  Accumulator j' of
            //       tile nt = T[row fi][word 16 nt + 4 fg + j']: the element that multiplies e[4 nt + j'] --
            //   P4  q = (E o T) * indicator with A = e[4 nt + j'] * acc[nt][j'], then the cosine terms.
            // the tile's block-diagonal caption Gram as A fragments of P3: row 16 nt + fi, 16 bytes at column 16 u + 4 fg
            float4 hfrag[4][4];
            const bool near_only = __builtin_amdgcn_readfirstlane(sm.meta.far) == 0;
            {
                const float *hb = g.hblk + ct * (int64_t)(SC_NT * SC_NT) + fi * SC_NT + 4 * fg;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (nt - u <= 1 && u - nt <= 1) hfrag[nt][u] = *reinterpret_cast<const float4 *>(hb + nt * 16 * SC_NT + 16 * u);
                        else hfrag[nt][u] = near_only ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(hb + nt * 16 * SC_NT + 16 * u);
                    }
            }
            float ind[16];   // indicator fragment: [word 16u + 4fg + j belongs to caption slot fi]
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) ind[4 * u + j] = (sm.meta.col_cap[16 * u + 4 * fg + j] == fi) ? 1.f : 0.f;
            float *rs_out = hbuf;       // [144][16] similarity terms for the aggregation below (LDS behind the parked block)
            dispatch_norm(norm, [&](auto NC) {
                constexpr int NORM = decltype(NC)::value;
                for (int mt = wave; mt < SC_MTILES; mt += 4) {
                    const int row = mt * 16 + fi;
                    const int ii = row / SC_R;
                    float e[16];
                    f32x4 sden = f32x4{0.f, 0.f, 0.f, 0.f}, snum = sden;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int w = 16 * u + 4 * fg + j;
                            const float av = AT(row, w);
                            const float bq = norm_apply_c<NORM>(av, sm.colstat[ii][w][0], sm.colstat[ii][w][1]);
                            e[4 * u + j] = (NORM == 2) ? fast_exp(bq * ls) : __builtin_amdgcn_exp2f(bq);
                            sden = __builtin_amdgcn_mfma_f32_16x16x4f32(e[4 * u + j], ind[4 * u + j], sden, 0, 0, 0);
                            snum = __builtin_amdgcn_mfma_f32_16x16x4f32(e[4 * u + j] * av, ind[4 * u + j], snum, 0, 0, 0);
                        }
                    f32x4 tacc[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) tacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    // Hblk is block diagonal: when no caption of the tile spans three 16-column blocks (`far` == 0: every tile of the
                    // BASELINE length distributions), the 16 x 16 blocks (nt, u) with |nt - u| >= 2 are zero -- 40 instead of 64 MFMAs.
                    // ONE wave-uniform branch per row tile (round 2 tried a branch per k-step: slower than the MFMAs it skipped).
                    auto t_products = [&](auto NEAR) {
                        constexpr bool near = decltype(NEAR)::value;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                if (!near || (nt - u <= 1 && u - nt <= 1)) tacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hfrag[nt][u].x, e[4 * u + 0], tacc[nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                if (!near || (nt - u <= 1 && u - nt <= 1)) tacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hfrag[nt][u].y, e[4 * u + 1], tacc[nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                if (!near || (nt - u <= 1 && u - nt <= 1)) tacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hfrag[nt][u].z, e[4 * u + 2], tacc[nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                if (!near || (nt - u <= 1 && u - nt <= 1)) tacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hfrag[nt][u].w, e[4 * u + 3], tacc[nt], 0, 0, 0);
                        }
                    };
                    if (near_only) t_products(std::true_type{});
                    else t_products(std::false_type{});
                    f32x4 sq = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) sq = __builtin_amdgcn_mfma_f32_16x16x4f32(e[4 * nt + j] * tacc[nt][j], ind[4 * nt + j], sq, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r2 = mt * 16 + 4 * fg + j;
                        const int64_t img = img0 + r2 / SC_R;
                        const float w1 = img < g.Ni ? g.vnorm[img * SC_R + r2 % SC_R] : 0.f;
                        const float rden = sden[j] > 0.f ? fast_rcp(sden[j]) : 0.f;     // caption slots >= ncap: never read
                        const float w2 = fast_sqrt(fmaxf(sq[j], 0.f)) * rden;
                        rs_out[r2 * SC_MAXCAP + fi] = (snum[j] * rden) * fast_rcp(fmaxf(w1 * w2, 1e-8f));
                    }
                }
            });
            rsim = rs_out;
            __syncthreads();
            SC_TICK(5)   // P1 + P3 + P4 per row tile
        } else {
            // E2: one lane per region row, loop over the captions of the tile.  Per caption: the caption Gram H_c (W x W)
            // is pulled into the part of the staging area that the parked block does not cover (19 KB free behind arawt),
            // the un-normalised attention weights e = exp(ls b - max) overwrite the raw block row segment, and
            //   num = sum e a / den,   ||ctx_r||^2 = e^T H_c e / den^2.
            // The per-element code is specialised on the norm mode at compile time (see dispatch_norm).
            for (int k = 0; k < ncap; ++k) {
                const int c0 = sm.meta.cap_start[k], c1 = sm.meta.cap_start[k + 1];
                const int W = c1 - c0;
                {
                    const float *H = g.cgram + g.cgram_off[sm.meta.cap_id[k]];
                    for (int idx = tid; idx < W * W; idx += SC_THREADS) hbuf[idx] = H[idx];
                }
                __syncthreads();
                if (tid < SC_MT) {
                    const int ii = tid / SC_R;
                    const int64_t img = img0 + ii;
                    float mx = -INFINITY, den = 0.f, num = 0.f;
                    dispatch_norm(norm, [&](auto NC) {
                        constexpr int NORM = decltype(NC)::value;
                        for (int c = c0; c < c1; ++c)
                            mx = fmaxf(mx, norm_apply_c<NORM>(AT(tid, c), sm.colstat[ii][c][0], sm.colstat[ii][c][1]) * ls);
                        for (int c = c0; c < c1; ++c) {
                            const float a = AT(tid, c);
                            const float e = expf(norm_apply_c<NORM>(a, sm.colstat[ii][c][0], sm.colstat[ii][c][1]) * ls - mx);
                            den += e;
                            num += e * a;
                            AT(tid, c) = e;  // own row, own caption segment: no other reader left
                        }
                    });
                    float q = 0.f;
                    for (int u = 0; u < W; ++u) {
                        float t = 0.f;
                        for (int v = 0; v < W; ++v) t += hbuf[u * W + v] * AT(tid, c0 + v);
                        q += AT(tid, c0 + u) * t;
                    }
                    const float rden = 1.f / den;
                    const float w1 = img < g.Ni ? g.vnorm[img * SC_R + tid % SC_R] : 0.f;
                    const float w2 = sqrtf(fmaxf(q, 0.f)) * rden;
                    sm.rsim2[tid][k] = (num * rden) / fmaxf(w1 * w2, 1e-8f);
                }
                __syncthreads();   // hbuf is reloaded for the next caption
            }
        }
        // E3: aggregate over the 36 regions.  Wave ii = image ii; lane = (caption slot k = lane & 15, quarter of the regions
        // lane >> 4): nine terms per lane, two xor-shuffles fold the quarters (round 1: one lane per (image, caption) walked all 36).
        {
            const int ii = wave, k = lane & 15, part = lane >> 4;
            const int64_t img = img0 + ii;
            float r;
            if (g.agg == 0) {
                r = 0.f;
#pragma unroll
                for (int t = 0; t < SC_R / 4; ++t) r += fast_exp(rsim[(ii * SC_R + part * (SC_R / 4) + t) * SC_MAXCAP + k] * g.lambda_lse);
            } else if (g.agg == 1) {
                r = -INFINITY;
#pragma unroll
                for (int t = 0; t < SC_R / 4; ++t) r = fmaxf(r, rsim[(ii * SC_R + part * (SC_R / 4) + t) * SC_MAXCAP + k]);
            } else {
                r = 0.f;
#pragma unroll
                for (int t = 0; t < SC_R / 4; ++t) r += rsim[(ii * SC_R + part * (SC_R / 4) + t) * SC_MAXCAP + k];
            }
            if (g.agg == 1) {
                r = xor16_max(r);
                r = xor32_max(r);
            } else {
                r = xor16_add(r);
                r = xor32_add(r);
            }
            if (g.agg == 0) r = fast_log(r) / g.lambda_lse;
            else if (g.agg == 3) r /= (float)SC_R;
            if (part == 0 && k < ncap && img < g.Ni) g.S[img * g.ldS + sm.meta.cap_id[k]] = r;
        }
        SC_TICK(3)   // P4 + E3
        if (g.dbg_cycles && tid == 0) atomicAdd(&g.dbg_cycles[7], __builtin_amdgcn_s_memrealtime() - real0_);
    }
  }   // tile loop
}

template <int PREC, int XA = -1>
__global__ __launch_bounds__(SC_THREADS, 2) void scan_xattn_kernel(ScanArgs g) {
    scan_xattn_body<PREC, XA>(g);
}

// ---- precompute kernels -------------------------------------------------------------------
// G[n] = X_n X_n^T for X_n [rows, D] (rows <= 64): one workgroup per matrix.
// upper2: write the upper-triangular form G'[r][s] = G[r][s] (r == s), 2 G[r][s] (r < s), 0 (r > s): e^T G' e == e^T G e
// (G is symmetric bit for bit: the same products in the same order), and the consumer skips the all-zero blocks.
__global__ __launch_bounds__(256) void gram_kernel(const float *__restrict__ X, const int64_t *__restrict__ row_off,
                                                   const int32_t *__restrict__ row_cnt, int fixed_rows, int D,
                                                   float *__restrict__ G, const int64_t *__restrict__ g_off, int upper2) {
    __shared__ float xs[64][33];
    const int64_t n = blockIdx.x;
    const int rows = row_cnt ? row_cnt[n] : fixed_rows;
    const float *x = X + (row_off ? row_off[n] : n * (int64_t)fixed_rows) * D;
    float *out = G + (g_off ? g_off[n] : n * (int64_t)fixed_rows * fixed_rows);
    const int npair = rows * rows;
    float acc[16];  // up to 64*64/256 pairs per thread
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < D; k0 += 32) {
        for (int idx = threadIdx.x; idx < rows * 32; idx += 256) {
            const int r = idx >> 5, k = idx & 31;
            xs[r][k] = (k0 + k < D) ? x[(int64_t)r * D + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int pidx = threadIdx.x + 256 * e;
            if (pidx < npair) {
                const int r1 = pidx / rows, r2 = pidx % rows;
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) s += xs[r1][k] * xs[r2][k];
                acc[e] += s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int pidx = threadIdx.x + 256 * e;
        if (pidx < npair) {
            float v = acc[e];
            if (upper2) {
                const int r1 = pidx / rows, r2 = pidx % rows;
                v = r2 > r1 ? 2.f * v : (r2 == r1 ? v : 0.f);
            }
            out[pidx] = v;
        }
    }
}

// The same Gram matrices for a FIXED row count <= 48 (the 36 regions of an image) on the matrix cores: one workgroup per matrix, wave w owns
// rows 16 w .. 16 w + 15 and the three 16-column tiles, 32-column chunks of X staged once in LDS and read as both MFMA operands
// (v_mfma_f32_16x16x4_f32: A[i][k] / B[k][j] at lane = 16 k + i; row stride 36 floats = conflict-free).  Round 5's VALU form took 1.07 ms
// for 5 000 images (0.08 of either roof); this one is bound by reading X once.  G[r][s] and G[s][r] are still the same bits (the same
// products in the same k order), which the upper-triangular form relies on.
__global__ __launch_bounds__(256) void gram_mfma_kernel(const float *__restrict__ X, int rows, int D, float *__restrict__ G, int upper2) {
    __shared__ __attribute__((aligned(16))) float xs[48][36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D4 = D >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(X) + (int64_t)blockIdx.x * rows * D4;
    f32x4 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < D4; d0 += 8) {
        __syncthreads();
        for (int idx = tid; idx < 48 * 8; idx += 256) {
            const int row = idx >> 3, c4 = idx & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows && d0 + c4 < D4) v = x4[(int64_t)row * D4 + d0 + c4];
            *reinterpret_cast<float4 *>(&xs[row][c4 * 4]) = v;
        }
        __syncthreads();
        if (wave < 3) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = kk * 4 + (lane >> 4);
                const float a = xs[wave * 16 + (lane & 15)][k];
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xs[j * 16 + (lane & 15)][k], acc[j], 0, 0, 0);
            }
        }
    }
    if (wave >= 3) return;
    float *out = G + (int64_t)blockIdx.x * rows * rows;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int s_ = j * 16 + (lane & 15);
        if (s_ >= rows) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 16 + 4 * (lane >> 4) + q;
            if (r >= rows) continue;
            float v = acc[j][q];
            if (upper2) v = s_ > r ? 2.f * v : (s_ == r ? v : 0.f);
            out[r * rows + s_] = v;
        }
    }
}

// plain row L2 norms (no eps): one wave per row
__global__ __launch_bounds__(256) void rownorm_kernel(const float *__restrict__ X, int64_t rows, int D,
                                                      float *__restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) { const float v = X[row * D + k]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) out[row] = sqrtf(s);
}

// Hblk[t][v][w] = e_v . e_w for words v, w of the same caption of column tile t, 0 elsewhere (padding columns, different captions):
// the dense 64 x 64 form the i2t epilogue multiplies with (round 3: built once per caption set instead of zero-filled and gathered
// into LDS by every workgroup of the tile's 1 250 image blocks).
__global__ __launch_bounds__(256) void scan_hblk_kernel(const ScanTileMeta *__restrict__ meta, const float *__restrict__ cgram,
                                                        const int64_t *__restrict__ coff, float *__restrict__ hblk) {
    const int64_t t = blockIdx.x;
    const ScanTileMeta &m = meta[t];
    for (int idx = threadIdx.x; idx < SC_NT * SC_NT; idx += 256) {
        const int v = idx / SC_NT, w = idx % SC_NT;
        const int k = m.col_cap[w];
        float val = 0.f;
        if (k >= 0 && m.col_cap[v] == k) {
            const int c0 = m.cap_start[k], W = m.cap_start[k + 1] - c0;
            val = cgram[coff[m.cap_id[k]] + (int64_t)(v - c0) * W + (w - c0)];
        }
        hblk[t * (SC_NT * SC_NT) + idx] = val;
    }
}

// exclusive prefix sum of len^2 (caption Gram offsets); single workgroup, Nc is small (<= ~1e5)
__global__ __launch_bounds__(1024) void sq_prefix_kernel(const int32_t *__restrict__ len, int64_t n,
                                                         int64_t *__restrict__ off) {
    __shared__ long long part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = t * per, e = (b + per < n) ? b + per : n;
    long long s = 0;
    for (int64_t i = b; i < e; ++i) s += (long long)len[i] * len[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; ++i) { const long long v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    long long run = part[t];
    for (int64_t i = b; i < e; ++i) { off[i] = run; run += (long long)len[i] * len[i]; }
}

// Re-pack the word embeddings tile by tile: wtiled[t*64 + j] = words[row of column j of tile t] (zero rows
// for padding columns), so that the main kernel addresses its B operand from blockIdx alone; also emits the
// per-tile metadata record and (t2i) the word norms ||E_w||.  One workgroup per tile; HBM-bound copy.
__global__ __launch_bounds__(256) void scan_pack_kernel(const float *__restrict__ words, const int64_t *__restrict__ cap_off,
                                                        const int32_t *__restrict__ cap_len,
                                                        const int32_t *__restrict__ tile_begin,
                                                        const int32_t *__restrict__ cap_order, int D,
                                                        float *__restrict__ wtiled, ScanTileMeta *__restrict__ meta,
                                                        float *__restrict__ wnorm, int32_t *__restrict__ cap_col) {
    __shared__ ScanTileMeta m;
    __shared__ int64_t off[SC_MAXCAP];
    __shared__ int32_t len[SC_MAXCAP];
    __shared__ int64_t col_row[SC_NT];
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x;
    const int c0 = tile_begin[t];
    int n = tile_begin[t + 1] - c0;
    if (n > SC_MAXCAP) n = SC_MAXCAP;
    if (tid < SC_MAXCAP) {
        int32_t cid = -1, l = 0;
        int64_t o = 0;
        if (tid < n) { cid = cap_order[c0 + tid]; l = cap_len[cid]; o = cap_off[cid]; }
        m.cap_id[tid] = cid; len[tid] = l; off[tid] = o;
    }
    if (tid < SC_NT) { m.col_cap[tid] = -1; col_row[tid] = -1; }
    __syncthreads();
    if (tid == 0) {
        int pos = 0;
        int far = 0;
        for (int k = 0; k < n; ++k) {
            m.cap_start[k] = pos;
            int w = len[k];
            if (pos + w > SC_NT) w = SC_NT - pos;  // the planner never lets this happen
            if (w > 0 && ((pos + w - 1) >> 4) - (pos >> 4) >= 2) far = 1;
            pos += w;
        }
        for (int k = n; k <= SC_MAXCAP; ++k) m.cap_start[k] = pos;
        m.ncap = n;
        m.far = far;
        if (cap_col)
            for (int k = 0; k < n; ++k) cap_col[m.cap_id[k]] = (int32_t)(t * SC_NT + m.cap_start[k]);
    }
    __syncthreads();
    for (int idx = tid; idx < n * SC_NT; idx += 256) {
        const int k = idx / SC_NT, w = idx % SC_NT;
        const int st = m.cap_start[k];
        if (w < m.cap_start[k + 1] - st) { col_row[st + w] = off[k] + w; m.col_cap[st + w] = (int8_t)k; }
    }
    __syncthreads();
    if (tid < 64) reinterpret_cast<int32_t *>(meta + t)[tid] = reinterpret_cast<const int32_t *>(&m)[tid];
    const int lane = tid & 63, wave = tid >> 6;
    for (int j = wave; j < SC_NT; j += 4) {
        const int64_t r = col_row[j];
        float *dst = wtiled + (t * SC_NT + j) * (int64_t)D;
        float ss = 0.f;
        if (r >= 0) {
            const float *src = words + r * D;
            for (int k = lane * 4; k < D; k += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(src + k);
                *reinterpret_cast<float4 *>(dst + k) = v;
                ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        } else {
            for (int k = lane * 4; k < D; k += 256) *reinterpret_cast<float4 *>(dst + k) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (wnorm) {
            ss = wave_sum(ss);
            if (lane == 0) wnorm[t * SC_NT + j] = sqrtf(ss);
        }
    }
}

// bf16x3 variant: rows of an fp32 matrix [rows, D] -> split planes interleaved per 32-wide K chunk,
// [rows][D / 32][hi (32 bf16) | lo (32 bf16)]  (hi = bf16(x), lo = bf16(x - hi), round to nearest even): x = hi + lo + O(2^-17 |x|)
__global__ __launch_bounds__(256) void scan_split_rows_kernel(const float *__restrict__ x, uint16_t *__restrict__ out, int64_t rows,
                                                              int D, int f16) {
    const int64_t row = blockIdx.y;
    const int d = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (d >= D) return;
    const float4 v = *reinterpret_cast<const float4 *>(x + row * D + d);
    auto rne = [](float f) -> uint32_t {
        const uint32_t u = __float_as_uint(f);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    const float in[4] = {v.x, v.y, v.z, v.w};
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (f16) {       // fp16 planes of x' = 2^12 x: hi = half(x'), lo' = half((x' - hi) * 2^11).  The matrix core flushes fp16
                         // subnormals, so both planes are kept normal by exact power-of-two scales (undone at the park):
                         // |x| >= 1.5e-8 stays normal, |x| <= 15.99 does not overflow -- unit-norm rows on this path
            const float xs = in[i] * 4096.f;
            const _Float16 hh = (_Float16)xs;
            const _Float16 ll = (_Float16)((xs - (float)hh) * 2048.f);
            h[i] = __builtin_bit_cast(uint16_t, hh);
            l[i] = __builtin_bit_cast(uint16_t, ll);
        } else {
            h[i] = rne(in[i]);
            l[i] = rne(in[i] - __uint_as_float(h[i] << 16));
        }
    }
    uint16_t *o = out + row * 2 * (int64_t)D + (d >> 5) * 64 + (d & 31);
    *reinterpret_cast<uint2 *>(o) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    *reinterpret_cast<uint2 *>(o + 32) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static_assert(sizeof(ScanSmem) <= 80 * 1024, "two workgroups per CU need <= 80 KiB of LDS each");

// Workspace layout (prepare and scores agree on it)
struct ScanWs {
    float *gram, *wnorm, *vnorm, *cgram, *hblk, *wtiled;
    int64_t *coff;
    ScanTileMeta *meta;
};
static size_t scan_ws_bytes(int64_t Ni, int R, int64_t n_rows, int64_t Nc, int64_t n_tiles, int D) {
    const size_t common = align256((size_t)n_tiles * sizeof(ScanTileMeta)) + align256((size_t)n_tiles * SC_NT * D * 4);
    const size_t t2i = align256((size_t)Ni * R * R * 4) + align256((size_t)n_tiles * SC_NT * 4);
    const size_t i2t = align256((size_t)Ni * R * 4) + align256((size_t)Nc * 8) + align256((size_t)n_rows * SC_NT * 4) +
                       align256((size_t)n_tiles * SC_NT * SC_NT * 4);
    return common + (t2i > i2t ? t2i : i2t);
}
static ScanWs scan_carve(void *workspace, int64_t Ni, int R, int64_t n_rows, int64_t Nc, int64_t n_tiles, int D, int mode) {
    char *ws = static_cast<char *>(workspace);
    ScanWs w{};
    w.meta = reinterpret_cast<ScanTileMeta *>(ws); ws += align256((size_t)n_tiles * sizeof(ScanTileMeta));
    w.wtiled = reinterpret_cast<float *>(ws); ws += align256((size_t)n_tiles * SC_NT * D * 4);
    if (mode == 0) {
        w.gram = reinterpret_cast<float *>(ws); ws += align256((size_t)Ni * R * R * 4);
        w.wnorm = reinterpret_cast<float *>(ws);
    } else {
        w.vnorm = reinterpret_cast<float *>(ws); ws += align256((size_t)Ni * R * 4);
        w.coff = reinterpret_cast<int64_t *>(ws); ws += align256((size_t)Nc * 8);
        w.cgram = reinterpret_cast<float *>(ws); ws += align256((size_t)n_rows * SC_NT * 4);
        w.hblk = reinterpret_cast<float *>(ws);
    }
    return w;
}

}  // namespace itr

extern "C" int itr_scan_plan_tiles(const int32_t *len_host, int64_t Nc, int nt, int32_t *tile_begin_host,
                                   int32_t *cap_order_host, int64_t *n_tiles) {
    // Whole captions into column tiles of `nt` words and at most SC_MAXCAP captions, tiles filled exactly where the
    // lengths allow it (pack_plan.h; rounds 1-4: best fit decreasing, 2 % more tiles on the bench's captions).
    // Output: cap_order_host[Nc] = caption ids grouped by tile, tile_begin_host[n_tiles + 1] into it.
    ITR_REQUIRE(len_host && tile_begin_host && cap_order_host && n_tiles, "itr_scan_plan_tiles: null pointer");
    ITR_REQUIRE(nt == ITR_SCAN_NT, "itr_scan_plan_tiles: nt must be %d", ITR_SCAN_NT);
    ITR_REQUIRE(Nc >= 0 && Nc < 0x7fffffffLL, "itr_scan_plan_tiles: bad caption count");
    std::vector<std::vector<int32_t>> by_len(nt + 1);
    for (int64_t c = 0; c < Nc; ++c) {
        const int w = len_host[c];
        ITR_REQUIRE(w >= 1, "itr_scan_plan_tiles: caption %lld has length %d (< 1)", (long long)c, w);
        ITR_UNSUPPORTED(w > nt, "itr_scan_plan_tiles: caption %lld has %d words; this build supports <= %d",
                        (long long)c, w, nt);
        by_len[w].push_back((int32_t)c);
    }
    static_assert(itr::PACK_MAXN == itr::SC_MAXCAP, "planner bins hold SC_MAXCAP captions");
    std::vector<itr::PackBin> tiles;
    itr::pack_bins(by_len, nt, itr::SC_MAXCAP, tiles);
    int64_t pos = 0;
    for (size_t t = 0; t < tiles.size(); ++t) {
        tile_begin_host[t] = (int32_t)pos;
        for (int k = 0; k < tiles[t].n; ++k) cap_order_host[pos++] = tiles[t].item[k];
    }
    tile_begin_host[tiles.size()] = (int32_t)pos;
    *n_tiles = (int64_t)tiles.size();
    return ITR_OK;
}

extern "C" size_t itr_scan_workspace_bytes(int64_t Ni, int R, int64_t n_rows, int64_t Nc, int64_t n_tiles, int D) {
    return itr::scan_ws_bytes(Ni, R, n_rows, Nc, n_tiles, D);
}

namespace itr {
int scan_prepare_impl(const float *img, const float *words, const int64_t *cap_off, const int32_t *cap_len,
                      const int32_t *tile_begin_dev, const int32_t *cap_order_dev, int64_t n_tiles, int64_t Ni,
                      int64_t Nc, int64_t n_rows, int R, int D, int mode, void *workspace, size_t workspace_bytes,
                      int32_t *cap_col, itr_stream_t stream) {
    ITR_REQUIRE(img && words && cap_off && cap_len && tile_begin_dev && cap_order_dev && workspace,
                "itr_scan_prepare: null pointer");
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && n_rows >= 0 && n_tiles >= 0 && D > 0, "itr_scan_prepare: bad shape");
    if (mode != 0 && mode != 1) { set_error("unknown cross_attn mode %d", mode); return ITR_ERR_BADARG; }
    ITR_UNSUPPORTED(R != SC_R, "itr_scan_prepare: this build handles %d regions per image, got %d", SC_R, R);
    ITR_UNSUPPORTED(D % SC_BK != 0, "itr_scan_prepare: embed dim must be a multiple of %d, got %d", SC_BK, D);
    ITR_REQUIRE((reinterpret_cast<uintptr_t>(img) & 15) == 0 && (reinterpret_cast<uintptr_t>(words) & 15) == 0,
                "itr_scan_prepare: operands must be 16-byte aligned");
    ITR_REQUIRE(workspace_bytes >= scan_ws_bytes(Ni, R, n_rows, Nc, n_tiles, D), "itr_scan_prepare: workspace too small");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    hipStream_t st = as_stream(stream);
    ScanWs w = scan_carve(workspace, Ni, R, n_rows, Nc, n_tiles, D, mode);
    hipLaunchKernelGGL(scan_pack_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, words, cap_off, cap_len,
                       tile_begin_dev, cap_order_dev, D, w.wtiled, w.meta, w.wnorm, cap_col);
    ITR_CHECK_LAUNCH("scan pack");
    if (mode == 0) {
        if (R <= 48 && D % 4 == 0)
            hipLaunchKernelGGL(gram_mfma_kernel, dim3((unsigned)Ni), dim3(256), 0, st, img, R, D, w.gram, 1);
        else
            hipLaunchKernelGGL(gram_kernel, dim3((unsigned)Ni), dim3(256), 0, st, img, (const int64_t *)nullptr,
                               (const int32_t *)nullptr, R, D, w.gram, (const int64_t *)nullptr, 1);
        ITR_CHECK_LAUNCH("scan gram");
    } else {
        hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned)ceil_div(Ni * R, 4)), dim3(256), 0, st, img, Ni * R, D, w.vnorm);
        ITR_CHECK_LAUNCH("scan vnorm");
        hipLaunchKernelGGL(sq_prefix_kernel, dim3(1), dim3(1024), 0, st, cap_len, Nc, w.coff);
        ITR_CHECK_LAUNCH("scan cgram offsets");
        hipLaunchKernelGGL(gram_kernel, dim3((unsigned)Nc), dim3(256), 0, st, words, cap_off, cap_len, 0, D, w.cgram,
                           (const int64_t *)w.coff, 0);
        ITR_CHECK_LAUNCH("scan caption gram");
        hipLaunchKernelGGL(scan_hblk_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, w.meta, w.cgram, w.coff, w.hblk);
        ITR_CHECK_LAUNCH("scan tile gram");
    }
    return ITR_OK;
}
}  // namespace itr

extern "C" int itr_scan_prepare(const float *img, const float *words, const int64_t *cap_off,
                                const int32_t *cap_len, const int32_t *tile_begin_dev,
                                const int32_t *cap_order_dev, int64_t n_tiles, int64_t Ni, int64_t Nc,
                                int64_t n_rows, int R, int D, int mode, void *workspace, size_t workspace_bytes,
                                itr_stream_t stream) {
    return itr::scan_prepare_impl(img, words, cap_off, cap_len, tile_begin_dev, cap_order_dev, n_tiles, Ni, Nc, n_rows,
                                  R, D, mode, workspace, workspace_bytes, nullptr, stream);
}

namespace itr {
// emit_p / emit_cn != null: also write the normalised attention weights [Ni, n_tiles*64, 36] and
// 1 / (||ctx|| + eps) [Ni, n_tiles*64] (SGRAF); S may then be null-free scratch of the usual shape.
static int scan_scores_impl2(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows, int R, int D,
                             int mode, int norm, int agg, float lambda_softmax, float lambda_lse, float *S, int64_t ldS,
                             void *workspace, size_t workspace_bytes, float *emit_p, float *emit_cn, int64_t img_index0,
                             int64_t img_count, void *bf16_ws, int f16, itr_stream_t stream, int debug_bits = 0);

int allow_dynamic_lds(const void *kernel, size_t bytes);      // scan_train.hip

int scan_scores_impl(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows, int R, int D,
                     int mode, int norm, int agg, float lambda_softmax, float lambda_lse, float *S, int64_t ldS,
                     void *workspace, size_t workspace_bytes, float *emit_p, float *emit_cn, int64_t img_index0,
                     int64_t img_count, itr_stream_t stream) {
    return scan_scores_impl2(img, n_tiles, Ni, Nc, n_rows, R, D, mode, norm, agg, lambda_softmax, lambda_lse, S, ldS, workspace,
                             workspace_bytes, emit_p, emit_cn, img_index0, img_count, nullptr, 0, stream);
}

static size_t scan_bf16_ws_bytes(int64_t Ni, int R, int64_t n_tiles, int D) {
    return align256((size_t)Ni * R * D * 4) + align256((size_t)n_tiles * SC_NT * D * 4);
}

// bf16_ws != null: the split-bf16 ("bf16x3") main loop; img and the packed word tiles are split into bf16_ws first
static int scan_scores_impl2(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows, int R, int D,
                             int mode, int norm, int agg, float lambda_softmax, float lambda_lse, float *S, int64_t ldS,
                             void *workspace, size_t workspace_bytes, float *emit_p, float *emit_cn, int64_t img_index0,
                             int64_t img_count, void *bf16_ws, int f16, itr_stream_t stream, int debug_bits) {
    ITR_REQUIRE(img && S && workspace, "itr_scan_xattn_scores: null pointer");
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && n_rows >= 0 && n_tiles >= 0 && ldS >= Nc, "itr_scan_xattn_scores: bad shape");
    if (mode != 0 && mode != 1) { set_error("unknown cross_attn mode %d", mode); return ITR_ERR_BADARG; }
    if (norm < 0 || norm > 6) { set_error("unknown first norm type: %d", norm); return ITR_ERR_BADARG; }
    if (agg < 0 || agg > 3) { set_error("unknown aggfunc: %d", agg); return ITR_ERR_BADARG; }
    ITR_UNSUPPORTED(R != SC_R, "itr_scan_xattn_scores: this build handles %d regions per image, got %d", SC_R, R);
    ITR_UNSUPPORTED(D <= 0 || D % SC_BK != 0, "itr_scan_xattn_scores: embed dim must be a multiple of %d, got %d",
                    SC_BK, D);
    ITR_REQUIRE((reinterpret_cast<uintptr_t>(img) & 15) == 0, "itr_scan_xattn_scores: operands must be 16-byte aligned");
    ITR_REQUIRE(workspace_bytes >= scan_ws_bytes(Ni, R, n_rows, Nc, n_tiles, D),
                "itr_scan_xattn_scores: workspace too small");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    hipStream_t st = as_stream(stream);

    ScanWs w = scan_carve(workspace, Ni, R, n_rows, Nc, n_tiles, D, mode);
    ScanArgs a{};
    a.img = img; a.wtiled = w.wtiled; a.meta = w.meta;
    a.S = S; a.ldS = ldS; a.Ni = Ni; a.Nc = Nc; a.n_tiles = n_tiles; a.D = D;
    a.mode = mode; a.norm = norm; a.agg = agg; a.lambda_softmax = lambda_softmax; a.lambda_lse = lambda_lse;
    a.gram = w.gram; a.wnorm = w.wnorm; a.vnorm = w.vnorm; a.cgram = w.cgram; a.cgram_off = w.coff; a.hblk = w.hblk;
    a.emit_p = emit_p; a.emit_cn = emit_cn;
    if (img_count >= 0) {   // score only images [img_index0, img_index0 + img_count) of the prepared set (S, emit_*: local rows)
        ITR_REQUIRE(img_index0 >= 0 && img_index0 + img_count <= Ni, "scan: image sub-range out of bounds");
        a.img = img + img_index0 * (int64_t)R * D;
        if (a.gram) a.gram += img_index0 * (int64_t)R * R;
        if (a.vnorm) a.vnorm += img_index0 * (int64_t)R;
        a.Ni = img_count;
        Ni = img_count;
        if (Ni == 0) return ITR_OK;
    }
    a.debug = debug_bits;      // 0, or 16 from itr_debug_scan_clock_probe; the ablation bits exist in experiment builds only
    if (const char *dbg = ITR_EXP_ENV("ITR_SCAN_DEBUG")) a.debug = atoi(dbg);
    if (a.debug & 16) {   // phase timing: the caller reads the 8 counters placed at the start of S (S is garbage then)
        a.dbg_cycles = reinterpret_cast<unsigned long long *>(S);
        ITR_CHECK_HIP(hipMemsetAsync(S, 0, 64, st));
        a.S = S + 16;
    }

    const int64_t img_tiles = ceil_div(Ni, SC_IMGS);
    const int64_t PI = ceil_div(img_tiles, 8), PJ = ceil_div(n_tiles, 8);
    const int64_t nblk = ceil_div(PI * PJ, 8) * 64 * 8;
    ITR_UNSUPPORTED(nblk > 0x7fffffffLL, "itr_scan_xattn_scores: grid too large; shard the call");
    {   // once per (kernel, device), under a mutex (scan_train.hip): a process may drive several devices
        int rc = allow_dynamic_lds(reinterpret_cast<const void *>(scan_xattn_kernel<0, 0>), 160 * 1024);
        if (rc == ITR_OK) rc = allow_dynamic_lds(reinterpret_cast<const void *>(scan_xattn_kernel<0, 1>), 160 * 1024);
        if (rc == ITR_OK) rc = allow_dynamic_lds(reinterpret_cast<const void *>(scan_xattn_kernel<1>), 160 * 1024);
        if (rc != ITR_OK) return rc;
    }
    size_t lds = sizeof(ScanSmem);
    if (const char *ex = ITR_EXP_ENV("ITR_SCAN_LDS_EXTRA")) lds += (size_t)atoi(ex);   // occupancy experiments only
    // tiles per workgroup (ITR_SCAN_TPW overrides, tools/scan_ablate2.py).  Measured at 1k x 5k: 37.13 / 36.89 / 36.94 / 36.85 ms
    // for 1 / 2 / 4 / 8 -- the launch gaps of one-tile workgroups are already covered by the co-resident workgroup.
    a.tpw = 2;
    if (const char *te = ITR_EXP_ENV("ITR_SCAN_TPW")) a.tpw = atoi(te) > 0 ? atoi(te) : 1;
    const int64_t grid = ceil_div(ceil_div(PI * PJ, 8), (int64_t)a.tpw) * 64 * 8;
    if (bf16_ws) {
        ITR_UNSUPPORTED((uint64_t)Ni * R * D * 4 >= (1ull << 32) || (uint64_t)SC_NT * D * 4 >= (1ull << 32),
                        "itr_scan_xattn_scores_bf16x3: per-launch operand offsets must fit 32 bits; shard the images");
        uint16_t *img_bf = static_cast<uint16_t *>(bf16_ws);
        uint16_t *wt_bf = reinterpret_cast<uint16_t *>(static_cast<char *>(bf16_ws) + align256((size_t)Ni * R * D * 4));
        const dim3 sg((unsigned)ceil_div(D / 4, 256), 1);
        ITR_REQUIRE(Ni * R <= 0x7fffffff / 1 && n_tiles * SC_NT <= 0x7fffffff, "scan split: too many rows");
        for (int64_t r0 = 0; r0 < Ni * R; r0 += 65535)
            hipLaunchKernelGGL(scan_split_rows_kernel, dim3(sg.x, (unsigned)min((int64_t)65535, Ni * R - r0)), dim3(256), 0, st,
                               a.img + r0 * D, img_bf + r0 * 2 * D, Ni * R - r0, D, f16);
        for (int64_t r0 = 0; r0 < n_tiles * SC_NT; r0 += 65535)
            hipLaunchKernelGGL(scan_split_rows_kernel, dim3(sg.x, (unsigned)min((int64_t)65535, n_tiles * SC_NT - r0)), dim3(256), 0,
                               st, w.wtiled + r0 * D, wt_bf + r0 * 2 * D, n_tiles * SC_NT - r0, D, f16);
        ITR_CHECK_LAUNCH("scan split");
        a.img_bf = img_bf;
        a.wt_bf = wt_bf;
        // ablation builds of the main loop (tools/scan_ablate2.py bf16x3): 5 = no global loads, 9 = no MFMAs -- results are garbage
        const int abl = ITR_EXP_ENV("ITR_SCAN_BF16_ABLATE") ? atoi(ITR_EXP_ENV("ITR_SCAN_BF16_ABLATE")) : 0;
        if (abl == 5) {
            ITR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(scan_xattn_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL(scan_xattn_kernel<5>, dim3((unsigned)grid), dim3(SC_THREADS), lds, st, a);
        } else if (abl == 9) {
            ITR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(scan_xattn_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL(scan_xattn_kernel<9>, dim3((unsigned)grid), dim3(SC_THREADS), lds, st, a);
        } else if (f16) {
            {
                const int rc = allow_dynamic_lds(reinterpret_cast<const void *>(scan_xattn_kernel<3>), 160 * 1024);
                if (rc != ITR_OK) return rc;
            }
            hipLaunchKernelGGL(scan_xattn_kernel<3>, dim3((unsigned)grid), dim3(SC_THREADS), lds, st, a);
        } else
            hipLaunchKernelGGL(scan_xattn_kernel<1>, dim3((unsigned)grid), dim3(SC_THREADS), lds, st, a);
    } else if (a.mode == 0) {
        hipLaunchKernelGGL((scan_xattn_kernel<0, 0>), dim3((unsigned)grid), dim3(SC_THREADS), lds, st, a);
    } else {
        hipLaunchKernelGGL((scan_xattn_kernel<0, 1>), dim3((unsigned)grid), dim3(SC_THREADS), lds, st, a);
    }
    ITR_CHECK_LAUNCH("scan_xattn");
    return ITR_OK;
}
}  // namespace itr

extern "C" int itr_scan_xattn_scores(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows,
                                     int R, int D, int mode, int norm, int agg, float lambda_softmax,
                                     float lambda_lse, float *S, int64_t ldS, void *workspace,
                                     size_t workspace_bytes, itr_stream_t stream) {
    return itr::scan_scores_impl(img, n_tiles, Ni, Nc, n_rows, R, D, mode, norm, agg, lambda_softmax, lambda_lse, S,
                                 ldS, workspace, workspace_bytes, nullptr, nullptr, 0, -1, stream);
}

extern "C" size_t itr_scan_bf16_workspace_bytes(int64_t Ni, int R, int64_t n_tiles, int D) {
    return itr::scan_bf16_ws_bytes(Ni, R, n_tiles, D);
}

extern "C" int itr_scan_xattn_scores_bf16x3(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows, int R, int D,
                                            int mode, int norm, int agg, float lambda_softmax, float lambda_lse, float *S,
                                            int64_t ldS, void *workspace, size_t workspace_bytes, void *bf16_workspace,
                                            size_t bf16_workspace_bytes, int split_format, itr_stream_t stream) {
    ITR_REQUIRE(bf16_workspace && bf16_workspace_bytes >= itr::scan_bf16_ws_bytes(Ni, R, n_tiles, D),
                "itr_scan_xattn_scores_bf16x3: split workspace missing or too small");
    ITR_REQUIRE(split_format == 0 || split_format == 1, "itr_scan_xattn_scores_bf16x3: split_format 0 (bf16) or 1 (fp16)");
    return itr::scan_scores_impl2(img, n_tiles, Ni, Nc, n_rows, R, D, mode, norm, agg, lambda_softmax, lambda_lse, S, ldS, workspace,
                                  workspace_bytes, nullptr, nullptr, 0, -1, bf16_workspace, split_format, stream);
}

// Diagnostics: ONE instrumented launch of the SCAN kernel (bench.py's sustained-clock probe).  Same arguments as
// itr_scan_xattn_scores, but `scratch` is NOT a score matrix afterwards: its first 64 bytes hold eight uint64 counters -- [0..6] the
// sum over all workgroups of the s_memtime (shader-clock) cycles spent per phase, [7] the sum of the s_memrealtime (100 MHz) ticks --
// and the rest is undefined.  sustained clock = 100 MHz * sum(c[0..6]) / c[7].  scratch: Ni rows of ld_scratch >= Nc + 64 floats.
extern "C" int itr_debug_scan_clock_probe(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows, int R, int D, int mode,
                                          int norm, int agg, float lambda_softmax, float lambda_lse, float *scratch, int64_t ld_scratch,
                                          void *workspace, size_t workspace_bytes, itr_stream_t stream) {
    ITR_REQUIRE(ld_scratch >= Nc + 64, "itr_debug_scan_clock_probe: scratch rows must hold Nc + 64 floats");
    return itr::scan_scores_impl2(img, n_tiles, Ni, Nc, n_rows, R, D, mode, norm, agg, lambda_softmax, lambda_lse, scratch, ld_scratch, workspace,
                                  workspace_bytes, nullptr, nullptr, 0, -1, nullptr, 0, stream, 16);
}

// Diagnostics for tools/: resident workgroups per CU of the SCAN kernel as the runtime sees it.
extern "C" int itr_debug_scan_occupancy(int *blocks_per_cu, int *lds_bytes) {
    using namespace itr;
    ITR_REQUIRE(blocks_per_cu && lds_bytes, "itr_debug_scan_occupancy: null pointer");
    ITR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(scan_xattn_kernel<0, 0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ScanSmem)));
    ITR_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void *>(scan_xattn_kernel<0, 0>),
                                                               SC_THREADS, sizeof(ScanSmem)));
    *lds_bytes = (int)sizeof(ScanSmem);
    return ITR_OK;
}
