// SGRAF Similarity Graph Reasoning, all steps of a group of captions in ONE workgroup
// (GraphReasoning.forward x sgr_step + sim_eval_w, itr/modalmodule/Fusionmodule.py:564-597, :437-444), sim_dim = 256.
//
// A graph is one (image, caption) pair: node 0 = the global similarity vector, nodes 1..W = the local (word) ones; graphs never
// talk to each other.  The unfused chain (sgraf.hip) runs every step as  GEMM (q' = W' x + v)  ->  pair kernel
// (softmax(q' X^T) X)  ->  GEMM (relu(W_g y + b))  over ALL node rows of a 16-image block, with q', y and the new x crossing HBM
// (3 KB per node and step; the pair kernel is memory-bound, 18 % of the SGR time in round 2).  Here a workgroup owns up to 64
// node rows = a GROUP of whole captions of one image (host plan: best-fit-decreasing bins of len + 1 rows, <= 16 captions) and
// keeps them in LDS through all steps; HBM sees the node rows once (64 KB in) and one score per caption (out).
//
// Everything is computed TRANSPOSED so that each product leaves its result in the register layout the next one needs:
//   P1  Q'^T[o, n] = W'[o, :] . X[n, :]      A = weight fragments streamed from L2 (pre-packed in fragment order: one coalesced
//                                             16-byte load per lane and 16 x 16 k-block), B = X^T from LDS.  Wave w owns the 32
//                                             output features 32 w .. 32 w + 31 (two 16-row tiles) for ALL node rows, so a weight
//                                             element is loaded exactly once per workgroup (512 KB per step against 21 MFLOP).
//   P2  per (caption, 16-query tile), one wave each:
//         E^T[j, i] = X[j, :] . Q'[i, :]      A = X rows, B = Q' rows (both gathered from LDS by node)
//         softmax over j in the accumulator layout (column i = lane & 15: two cross-lane steps), P^T stays in registers
//         Y^T[d, i] = sum_j X[j, d] P^T[j, i]  A = X columns, B = the softmax registers as they are
//       Y overwrites the Q' rows of the same 16 nodes (nobody else reads them).
//   P3  X'^T[o, n] = relu(W_g[o, :] . Y[n, :] + b)    as P1, from the Q'/Y buffer back into the X buffer.
// The last step needs node 0 only (Fusionmodule.py:443 reads sim_emb[:, 0]): its query projection and its one softmax row per caption
// run on the vector ALU (sf_last_project / sf_last_attend below), and y of node 0 leaves the kernel -- the last graph projection and
// sigmoid(sim_eval_w . x_0 + b) are one GEMM + one small kernel over all the graphs of the image block (sgraf.hip).
// v_mfma_f32_16x16x4_f32 throughout (exact fp32).  LDS: two [ROWS][SF_LD = 260] fp32 buffers + the group records + the softmax tiles.
//
// Two sizes of group (round 4).  ROWS = 64 (the planner's default, itr_sgr_plan_node_groups(small_rows = 64)): 157 KB of LDS, one
// 512-thread workgroup per CU.  ROWS = 32 (opt-in: small_rows = 32 puts every caption of at most 31 words into groups of <= 32 node
// rows; the Python layer's ITR_SGR_GROUP_ROWS=32): 79 KB, TWO workgroups per CU -- while one of them sits in its attention phase or at
// a barrier the other one's projection MFMAs can fill the pipe.  Measured in round 4: same scores bit for bit, 740 ms against 731 ms
// at 1k x 5k -- the second resident workgroup does not raise the matrix pipe's busy fraction (the limit is ALU time, and halving the
// group doubles the per-item last step and the weight stream), so 64 stayed the default.  The groups of a call are sorted into the two
// classes ON THE DEVICE (sgr_group_meta_kernel + sgr_group_classify_kernel: no host round trip); a class's persistent launch reads
// its item count from device memory.
#include "scan_common.h"
#include "pack_plan.h"
#include <string.h>
#include <stdlib.h>
#include <vector>

namespace itr {

// Row stride of the two node-row buffers, in floats.  260 = 65 sixteen-byte slots: the 16 rows of a ds_read_b128 fragment read
// (lanes fi = 0..15 of one quarter-wave) start in 16 distinct slots mod 16, and the four rows 4 fq + r of the value gathers of P2
// (ds_read_b32, 4 x 260 floats apart = 16 banks) no longer pair up on the same banks as they did at 264 (4 x 264 = 32 banks: the
// 2-way conflicts of round 3's PMC).  Same-box A/B: 723.7 -> 721.2 ms at 1k x 5k (268: 722.6).
#ifndef ITR_SF_LNR          // rows per pass / k per weight batch of the last step's VALU projection in the one-per-CU kernel
#define ITR_SF_LNR 8
#endif
#ifndef ITR_SF_LKB
#define ITR_SF_LKB 8
#endif
#ifndef ITR_SF_LD
#define ITR_SF_LD 260
#endif
constexpr int SF_ROWS = 64, SF_S = 256, SF_LD = ITR_SF_LD, SF_MAXCAP = 16, SF_MAXUNIT = 24, SF_THREADS = 512, SF_WAVES = 8;
constexpr int SF_SMALL = 32;             // node rows of the small class of groups (two workgroups per CU)

// One record per group of captions (built on the device from the host's bin plan: sgr_group_meta_kernel).
struct alignas(16) SgrGroupMeta {
    int32_t ncap, nrows, nunit, pad0;
    int32_t cap_id[SF_MAXCAP];
    int16_t wstart[SF_MAXCAP];           // LDS row of a caption's first word node (its global node is row = slot)
    int16_t nn[SF_MAXCAP];               // nodes of the caption's graph = words + 1
    uint8_t unit_cap[SF_MAXUNIT];        // P2 work units (caption slot, 16-query tile), largest graphs first
    uint8_t unit_tile[SF_MAXUNIT];
    int32_t row_src[SF_ROWS];            // rows < ncap: caption id (global node); others: tile-packed column of the word
    uint8_t unit_poff[SF_MAXUNIT];       // first 1 KB softmax tile of a unit in the P^T scratch (prefix sum of the units' key-tile counts)
    uint8_t cap_poff[SF_MAXCAP];         // (rounds 1-3: the same for the last step's MFMA units; unused since the last step runs on the vector ALU)
    int32_t pad1[6];
};
static_assert(sizeof(SgrGroupMeta) == 512, "one 512-byte record per group");

// softmax tiles (16 x 16 floats) of one group = sum over its captions of (key tiles)^2.  <= 64 node rows in <= 16 graphs of >= 2 nodes:
// at most 24 (one 33-node graph + fifteen 2-node ones: 9 + 15); <= 32 rows: at most 11 (one 17-node graph + seven 2-node ones: 4 + 7)
constexpr int sf_ptiles(int rows) { return rows > SF_SMALL ? 24 : 12; }
constexpr size_t sf_lds_bytes(int rows) {      // (two group records: the persistent form holds the next item's as well)
    return 2 * (size_t)rows * SF_LD * sizeof(float) + 2 * sizeof(SgrGroupMeta) + (size_t)sf_ptiles(rows) * 1024;
}
static_assert(2 * sf_lds_bytes(SF_SMALL) <= 160 * 1024 && sf_lds_bytes(SF_ROWS) <= 160 * 1024, "LDS budget of a CU");

__global__ __launch_bounds__(256) void sgr_group_meta_kernel(const int32_t *__restrict__ grp_begin, const int32_t *__restrict__ grp_order,
                                                             const int32_t *__restrict__ cap_len, const int32_t *__restrict__ cap_col,
                                                             int64_t n_groups, SgrGroupMeta *__restrict__ meta, int *__restrict__ bad,
                                                             int8_t *__restrict__ cls, int32_t *__restrict__ cap_bad, int64_t n_caps) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= n_groups) return;
    SgrGroupMeta m;
    memset(&m, 0, sizeof(m));
    const int b0 = grp_begin[gi], ncap = grp_begin[gi + 1] - b0;
    int rows = ncap;
    bool ok = ncap >= 1 && ncap <= SF_MAXCAP;
    for (int s = 0; ok && s < ncap; ++s) {
        const int c = grp_order[b0 + s];
        if (c < 0 || c >= n_caps) { ok = false; break; }
        const int len = cap_len[c];
        if (len < 1 || rows + len > SF_ROWS) { ok = false; break; }
        m.cap_id[s] = c;
        m.nn[s] = (int16_t)(len + 1);
        m.wstart[s] = (int16_t)rows;
        m.row_src[s] = c;
        for (int w = 0; w < len; ++w) m.row_src[rows + w] = cap_col[c] + w;
        rows += len;
    }
    // The plan is wrong (more than 16 captions, more than 64 node rows, more softmax tiles than the scratch holds): the group is not
    // scored, and so that this is LOUD and not a column of uninitialised memory, its captions are marked and sgr_poison_kernel writes
    // NaN into their score columns at the end of the call (the call has returned by then: no error code can carry it).
    auto refuse = [&]() {
        atomicExch(bad, 1);
        for (int s = 0; s < ncap && s < 4096; ++s) {
            const int c = grp_order[b0 + s];
            if (c >= 0 && c < n_caps) cap_bad[c] = 1;
        }
        m.ncap = 0; meta[gi] = m; cls[gi] = -1;
    };
    if (!ok) { refuse(); return; }
    m.ncap = ncap;
    m.nrows = rows;
    int nu = 0;
    for (int nt = 4; nt >= 1; --nt)
        for (int s = 0; s < ncap; ++s)
            if ((m.nn[s] + 15) / 16 == nt)
                for (int b = 0; b < nt; ++b) { m.unit_cap[nu] = (uint8_t)s; m.unit_tile[nu] = (uint8_t)b; ++nu; }
    m.nunit = nu;
    int po = 0;
    for (int u = 0; u < nu; ++u) { m.unit_poff[u] = (uint8_t)po; po += (m.nn[m.unit_cap[u]] + 15) / 16; }
    if (po > sf_ptiles(SF_ROWS)) { refuse(); return; }
    const bool small = rows <= SF_SMALL && po <= sf_ptiles(SF_SMALL);
    po = 0;
    for (int s = 0; s < ncap; ++s) { m.cap_poff[s] = (uint8_t)po; po += (m.nn[s] + 15) / 16; }
    meta[gi] = m;
    cls[gi] = small ? 0 : 1;
}

// Stable compaction of the group indices into the two class lists (one workgroup; a call has a few thousand groups).
// glist[0 .. n) = small groups, glist[n_groups .. ) = large ones; gcount[0 / 1] = their numbers.
__global__ __launch_bounds__(1024) void sgr_group_classify_kernel(const int8_t *__restrict__ cls, int64_t n_groups, int32_t *__restrict__ glist,
                                                                  int32_t *__restrict__ gcount) {
    __shared__ int wsum[2][16];
    __shared__ int base[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 2) base[tid] = 0;
    __syncthreads();
    for (int64_t g0 = 0; g0 < n_groups; g0 += 1024) {
        const int64_t gi = g0 + tid;
        const int c = gi < n_groups ? cls[gi] : -1;
        int pos[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned long long bal = __ballot(c == k);
            pos[k] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wsum[k][wv] = __popcll(bal);
        }
        __syncthreads();
        if (c >= 0) {
            int off = base[c];
            for (int w = 0; w < wv; ++w) off += wsum[c][w];
            glist[(c ? n_groups : 0) + off + pos[c]] = (int32_t)gi;
        }
        __syncthreads();
        if (tid < 2) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[tid][w]; base[tid] += t; }
        __syncthreads();
    }
    if (tid < 2) gcount[tid] = base[tid];
}

// A refused group's captions: NaN in their score columns (see sgr_group_meta_kernel).
__global__ __launch_bounds__(256) void sgr_poison_kernel(const int32_t *__restrict__ cap_bad, int64_t Nc, int64_t Ni, float *__restrict__ S, int64_t ldS) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= Nc || !cap_bad[c]) return;
    for (int64_t i = 0; i < Ni; ++i) S[i * ldS + c] = __builtin_nanf("");
}

// W [256][256] row-major -> MFMA A-fragment order: frag[((ot * 16 + ks) * 64 + lane)] (float4) = W[16 ot + (lane & 15)][16 ks + 4 (lane >> 4) + 0..3]
__global__ __launch_bounds__(256) void sgr_pack_weight_kernel(const float *__restrict__ W, float4 *__restrict__ frag) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // 16 * 16 * 64 fragments
    if (idx >= 16 * 16 * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) & 15, ot = idx >> 10;
    frag[idx] = *reinterpret_cast<const float4 *>(W + (16 * ot + (lane & 15)) * SF_S + 16 * ks + 4 * (lane >> 4));
}

// W [256][256] row-major [o][k] -> W^T [k][o]: the operand of the last step's vector-ALU projections (a lane owns four consecutive
// outputs o and reads them with one 16-byte load per k).
__global__ __launch_bounds__(256) void sgr_transpose_weight_kernel(const float *__restrict__ W, float *__restrict__ WT) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // = k * 256 + o
    if (idx >= SF_S * SF_S) return;
    WT[idx] = W[(idx & (SF_S - 1)) * SF_S + (idx >> 8)];
}

struct SgrFusedArgs {
    const float *xloc, *xglo;            // [nb][ncols][256], [nb][Nc][256]
    const SgrGroupMeta *meta;
    const int32_t *glist, *gcount;       // this class's group indices and their number (device memory: sgr_group_classify_kernel)
    int64_t n_groups, nb, Nc, ncols;
    const float4 *wq[8], *wg[8];         // fragment-ordered folded query weight / graph weight of every step
    const float *vq[8], *bg[8];
    const float *wqT_last;               // the LAST step's folded query weight, transposed: [k][o] (sgr_transpose_weight_kernel)
    int steps;
    float *y0;                           // [nb][Nc][256]: y of node 0 after the last step's attention (the last graph projection and the score
                                         // run as ONE GEMM over all the graphs of the image block afterwards: sgraf.hip)
    unsigned long long *trace;           // debug (ITR_SGR_TRACE): [grid][20] = hardware id, group shape, s_memtime at entry / after the load / after every phase
};

#define SF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Weight-fragment loads of the projections go through inline asm with their own s_waitcnt: hipcc sinks a plain load next to its
// first use (the loop then opened every k-block with an exposed L2 round trip: "global_load ...; s_waitcnt vmcnt(2); v_mfma"),
// an asm load stays where it is written -- one k-block (8 NG MFMAs) ahead of its consumer.  The destination is opaque to the
// compiler until SF_WAIT_VM names it ("+v": no consumer is scheduled above the wait; tests/test_isa_audit.py checks that no compiler
// instruction touches the registers in between).
#define SF_GLOAD(dst, ptr, imm) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "i"(imm))
#define SF_LREAD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define SF_WAIT_ALL(x, y) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x), "+v"(y))
#define SF_WAIT_ALL1(x) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x))
#define SF_OPAQUE(x) asm volatile("" : "+v"(x))

// One 16-wide k-block of a projection: requests block KS + 1 (weights from L2, node fragments from LDS -- both through asm: hipcc
// also delays plain LDS reads to the block that uses them), runs the 8 NG MFMAs of block KS, waits.
template <int KS, int NG>
__device__ __forceinline__ void sf_kblock(unsigned baddr, const float4 *const (&wp)[2][4], f32x4 (&acc0)[NG], f32x4 (&acc1)[NG],
                                          f32x4 &a0, f32x4 &a1, f32x4 (&b)[NG], const float4 *nxt, f32x4 &x0, f32x4 &x1) {
    f32x4 n0, n1;
    f32x4 nb[NG];
    // the block's first 2 NG MFMAs go out right behind the wait that closed the previous block; the requests for block KS + 1 are
    // issued while the matrix pipe works on them (six memory instructions ahead of the first MFMA were ~70 idle cycles per block)
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) { acc0[ng] = SF_MFMA(a0[0], b[ng][0], acc0[ng]); acc1[ng] = SF_MFMA(a1[0], b[ng][0], acc1[ng]); }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KS == 15) {
        // the first fragments of the NEXT projection (they depend on nothing computed here): requested behind the last block's
        // MFMAs, waited for right after them, so the next phase opens with MFMAs instead of an L2 round trip after its barrier
        SF_GLOAD(x0, nxt, 0);
        SF_GLOAD(x1, nxt + 16 * 64, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (KS < 15) {
#ifdef SF_EXP_SAME_BLOCK      // timing experiment (tools/ab_build.sh): every k-block re-reads the weights of block 0 (L1 hits; results are garbage)
        SF_GLOAD(n0, wp[0][0], 0);
        SF_GLOAD(n1, wp[1][0], 0);
#else
        SF_GLOAD(n0, wp[0][(KS + 1) / 4], ((KS + 1) % 4) * 1024);
        SF_GLOAD(n1, wp[1][(KS + 1) / 4], ((KS + 1) % 4) * 1024);
#endif
        SF_LREAD(nb[0], baddr, 64 * (KS + 1));
        if constexpr (NG > 1) SF_LREAD(nb[1], baddr, 16 * SF_LD * 4 + 64 * (KS + 1));
        if constexpr (NG > 2) SF_LREAD(nb[2], baddr, 2 * 16 * SF_LD * 4 + 64 * (KS + 1));
        if constexpr (NG > 3) SF_LREAD(nb[3], baddr, 3 * 16 * SF_LD * 4 + 64 * (KS + 1));
        __builtin_amdgcn_sched_barrier(0);      // the requests stay AHEAD of the rest of this block's MFMAs (hipcc otherwise sinks them to the wait)
    }
#pragma unroll
    for (int r = 1; r < 4; ++r)
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) { acc0[ng] = SF_MFMA(a0[r], b[ng][r], acc0[ng]); acc1[ng] = SF_MFMA(a1[r], b[ng][r], acc1[ng]); }
    if constexpr (KS < 15) {
        __builtin_amdgcn_sched_barrier(0);
        SF_WAIT_ALL(n0, n1);
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) SF_OPAQUE(nb[ng]);
        __builtin_amdgcn_sched_barrier(0);
        a0 = n0; a1 = n1;
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) b[ng] = nb[ng];
        sf_kblock<KS + 1, NG>(baddr, wp, acc0, acc1, a0, a1, b, nxt, x0, x1);
    }
    if constexpr (KS == 15) {
        __builtin_amdgcn_sched_barrier(0);
        SF_WAIT_ALL(x0, x1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// dst[n][o] = act(sum_k W[o][k] src[n][k] + bias[o]) for the node rows of the first NG 16-row groups; this wave's 32 features.
// src_lds = LDS byte address of the source buffer.  (a0, a1) in: the fragments of k-block 0 (landed); out: those of the projection
// whose fragment-ordered weight is `wnext`.
template <int NG, bool RELU>
__device__ __forceinline__ void sf_project(unsigned src_lds, float *__restrict__ dst, const float4 *__restrict__ wfrag,
                                           const float *__restrict__ bias, int wave, int lane, f32x4 &a0, f32x4 &a1, const float4 *wnext) {
    const int fi = lane & 15, fq = lane >> 4;
    const float4 *w0 = wfrag + (size_t)(2 * wave) * 16 * 64 + lane, *w1 = w0 + 16 * 64;
    const float4 *const wp[2][4] = {{w0, w0 + 256, w0 + 512, w0 + 768}, {w1, w1 + 256, w1 + 512, w1 + 768}};   // 4 KB apart: 12-bit offsets
    const unsigned baddr = src_lds + (unsigned)(fi * SF_LD + 4 * fq) * 4u;
    // The accumulators START at the bias (element r of tile t is feature 16 (2 wave + t) + 4 fq + r for every node column): its two
    // 16-byte loads are requested here and land with the first node fragments -- a plain load in the epilogue was sunk there by
    // hipcc and cost an L2 round trip at the end of every phase.
    f32x4 bv0, bv1;
    SF_GLOAD(bv0, bias + 32 * wave + 4 * fq, 0);
    SF_GLOAD(bv1, bias + 32 * wave + 16 + 4 * fq, 0);
    f32x4 b[NG];
    SF_LREAD(b[0], baddr, 0);
    if constexpr (NG > 1) SF_LREAD(b[1], baddr, 16 * SF_LD * 4);
    if constexpr (NG > 2) SF_LREAD(b[2], baddr, 2 * 16 * SF_LD * 4);
    if constexpr (NG > 3) SF_LREAD(b[3], baddr, 3 * 16 * SF_LD * 4);
    SF_WAIT_ALL1(b[0]);
#pragma unroll
    for (int ng = 1; ng < NG; ++ng) SF_OPAQUE(b[ng]);
    SF_OPAQUE(bv0);
    SF_OPAQUE(bv1);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0[NG], acc1[NG];
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) { acc0[ng] = bv0; acc1[ng] = bv1; }
    f32x4 x0, x1;
    sf_kblock<0, NG>(baddr, wp, acc0, acc1, a0, a1, b, wnext + (size_t)(2 * wave) * 16 * 64 + lane, x0, x1);
    a0 = x0; a1 = x1;
    // 16-pass results of the last MFMAs: the compiler pads its own consumers (no asm reads the accumulators)
    // accumulator element r of tile t: feature 16 (2 wave + t) + 4 fq + r, node 16 ng + fi  ->  one 16-byte store per tile
    float *drow = dst + fi * SF_LD + 32 * wave + 4 * fq;
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) {
        float4 v0{acc0[ng][0], acc0[ng][1], acc0[ng][2], acc0[ng][3]};
        float4 v1{acc1[ng][0], acc1[ng][1], acc1[ng][2], acc1[ng][3]};
        if (RELU) {
            v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
            v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
        }
        *reinterpret_cast<float4 *>(drow + ng * 16 * SF_LD) = v0;
        *reinterpret_cast<float4 *>(drow + ng * 16 * SF_LD + 16) = v1;
    }
}

template <bool RELU, int ROWS>
__device__ __forceinline__ void sf_project_n(int ng, unsigned src, float *dst, const float4 *wfrag, const float *bias, int wave, int lane,
                                             f32x4 &a0, f32x4 &a1, const float4 *wnext) {
    if constexpr (ROWS > SF_SMALL) {
        switch (ng) {
            case 1: sf_project<1, RELU>(src, dst, wfrag, bias, wave, lane, a0, a1, wnext); break;
            case 2: sf_project<2, RELU>(src, dst, wfrag, bias, wave, lane, a0, a1, wnext); break;
            case 3: sf_project<3, RELU>(src, dst, wfrag, bias, wave, lane, a0, a1, wnext); break;
            default: sf_project<4, RELU>(src, dst, wfrag, bias, wave, lane, a0, a1, wnext); break;
        }
    } else {
        if (ng == 1) sf_project<1, RELU>(src, dst, wfrag, bias, wave, lane, a0, a1, wnext);
        else sf_project<2, RELU>(src, dst, wfrag, bias, wave, lane, a0, a1, wnext);
    }
}

// all-reduce over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48) in registers: v_permlane32_swap / v_permlane16_swap
// exchange halves / odd-even rows of two copies, so two VALU steps replace two LDS-crossbar round trips (ds_bpermute) each
__device__ __forceinline__ float sf_rows_max(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sf_rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

#define SF_LREAD32(dst, addr, imm) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define SF_WAIT_LGKM1(x) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x))

// One 16-feature block of E^T = X Q'^T for a unit: requests the row fragments of block U + 1 (asm LDS reads, see sf_kblock), runs the
// 4 NTC MFMAs of block U on TWO accumulators per tile, alternating (a chain of dependent MFMAs runs at 40, not 32, cycles each).
template <int U, int NTC>
__device__ __forceinline__ void sf_eblock(unsigned qaddr, const unsigned (&kaddr)[NTC], f32x4 (&e0)[NTC], f32x4 (&e1)[NTC], f32x4 &q, f32x4 (&k)[NTC]) {
    f32x4 nq, nk[NTC];
    if constexpr (U < 15) {
        SF_LREAD(nq, qaddr, 64 * (U + 1));
#pragma unroll
        for (int a = 0; a < NTC; ++a) SF_LREAD(nk[a], kaddr[a], 64 * (U + 1));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < NTC; ++a) { e0[a] = SF_MFMA(k[a][0], q[0], e0[a]); e1[a] = SF_MFMA(k[a][1], q[1], e1[a]); }
#pragma unroll
    for (int a = 0; a < NTC; ++a) { e0[a] = SF_MFMA(k[a][2], q[2], e0[a]); e1[a] = SF_MFMA(k[a][3], q[3], e1[a]); }
    if constexpr (U < 15) {
        __builtin_amdgcn_sched_barrier(0);
        SF_WAIT_LGKM1(nq);
#pragma unroll
        for (int a = 0; a < NTC; ++a) SF_OPAQUE(nk[a]);
        __builtin_amdgcn_sched_barrier(0);
        q = nq;
#pragma unroll
        for (int a = 0; a < NTC; ++a) k[a] = nk[a];
        sf_eblock<U + 1, NTC>(qaddr, kaddr, e0, e1, q, k);
    }
}

// P2, first half.  One unit = the 16 query nodes of tile `tile` of caption slot `ci`
// against the caption's NTC key tiles: E^T, softmax over the keys, P^T -> LDS (`pt`: NTC tiles of [4 fq][16 fi] float4 -- a lane
// stores its accumulator registers as they are and the second half reads them back as MFMA B operands).
template <int NTC>
__device__ __forceinline__ void sf_attend_e(unsigned xb_lds, unsigned qy_lds, const SgrGroupMeta &m, int ci, int tile, int lane,
                                            float4 *__restrict__ pt) {
    const int fi = lane & 15, fq = lane >> 4;
    const int nn = m.nn[ci], ws = m.wstart[ci];
    auto row_of = [&](int n) { n = n < nn ? n : nn - 1; return n == 0 ? ci : ws + n - 1; };   // rows past the graph re-read its last node (masked)
    const unsigned qaddr = qy_lds + (unsigned)(row_of(16 * tile + fi) * SF_LD + 4 * fq) * 4u;
    unsigned kaddr[NTC];
#pragma unroll
    for (int a = 0; a < NTC; ++a) kaddr[a] = xb_lds + (unsigned)(row_of(16 * a + fi) * SF_LD + 4 * fq) * 4u;
    f32x4 e0[NTC], e1[NTC], q, k[NTC];
#pragma unroll
    for (int a = 0; a < NTC; ++a) { e0[a] = f32x4{0.f, 0.f, 0.f, 0.f}; e1[a] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    SF_LREAD(q, qaddr, 0);
#pragma unroll
    for (int a = 0; a < NTC; ++a) SF_LREAD(k[a], kaddr[a], 0);
    SF_WAIT_LGKM1(q);
#pragma unroll
    for (int a = 0; a < NTC; ++a) SF_OPAQUE(k[a]);
    __builtin_amdgcn_sched_barrier(0);
    sf_eblock<0, NTC>(qaddr, kaddr, e0, e1, q, k);
    // softmax over the keys j = 16 a + 4 fq + r of column i = fi (Fusionmodule.py:595: softmax(sim_edge, dim=-1))
    float p[NTC][4];
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < NTC; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = (16 * a + 4 * fq + r < nn) ? e0[a][r] + e1[a][r] : -INFINITY;
            p[a][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = sf_rows_max(mx);
    // exp through v_exp_f32 and the reciprocal through v_rcp_f32 (1 ulp each; the weights sum to 1 within 1e-7, far inside the 5e-6
    // parity budget of the scores): the softmax of a unit is ~40 vector instructions instead of ~150, on the unit's critical path
    float den = 0.f;
#pragma unroll
    for (int a = 0; a < NTC; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[a][r] = fast_exp(p[a][r] - mx); den += p[a][r]; }
    den = sf_rows_sum(den);
    const float inv = fast_rcp(den);
#pragma unroll
    for (int a = 0; a < NTC; ++a) pt[a * 64 + lane] = float4{p[a][0] * inv, p[a][1] * inv, p[a][2] * inv, p[a][3] * inv};
}

// One chunk of a Y task: key tile a = C / 2, feature tiles t = 2 (C % 2) and + 1: 8 A operands (asm LDS reads, requested one chunk
// ahead -- the 4-bit lgkmcnt allows 15 in flight), 8 MFMAs alternating between two accumulators.
template <int C, int NTC, int H>
__device__ __forceinline__ void sf_ychunk(const unsigned (&vaddr)[NTC][4], const float4 (&p)[NTC], f32x4 (&y)[4], float (&cur)[4][2]) {
    float nxt[4][2];
    constexpr int a = C / 2, th = C % 2;
    if constexpr (C + 1 < 2 * NTC) {
        constexpr int na = (C + 1) / 2, nth = (C + 1) % 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) { SF_LREAD32(nxt[r][0], vaddr[na][r], H + 128 * nth); SF_LREAD32(nxt[r][1], vaddr[na][r], H + 128 * nth + 64); }
        __builtin_amdgcn_sched_barrier(0);
    }
    y[2 * th] = SF_MFMA(cur[0][0], p[a].x, y[2 * th]); y[2 * th + 1] = SF_MFMA(cur[0][1], p[a].x, y[2 * th + 1]);
    y[2 * th] = SF_MFMA(cur[1][0], p[a].y, y[2 * th]); y[2 * th + 1] = SF_MFMA(cur[1][1], p[a].y, y[2 * th + 1]);
    y[2 * th] = SF_MFMA(cur[2][0], p[a].z, y[2 * th]); y[2 * th + 1] = SF_MFMA(cur[2][1], p[a].z, y[2 * th + 1]);
    y[2 * th] = SF_MFMA(cur[3][0], p[a].w, y[2 * th]); y[2 * th + 1] = SF_MFMA(cur[3][1], p[a].w, y[2 * th + 1]);
    if constexpr (C + 1 < 2 * NTC) {
        __builtin_amdgcn_sched_barrier(0);
        SF_WAIT_LGKM1(nxt[0][0]);
        SF_OPAQUE(nxt[0][1]);
#pragma unroll
        for (int r = 1; r < 4; ++r) { SF_OPAQUE(nxt[r][0]); SF_OPAQUE(nxt[r][1]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { cur[r][0] = nxt[r][0]; cur[r][1] = nxt[r][1]; }
        sf_ychunk<C + 1, NTC, H>(vaddr, p, y, cur);
    }
}

// P2, second half.  One task = a unit x HALF of the 256 features (two quarters of 64, one after the other: the row addresses and
// the softmax registers are set up once): Y^T[d][i] = sum_j X[j][d] P^T[j][i], A = X[row(16 a + 4 fq + r)][d], B = the stored
// softmax registers.  Y overwrites the Q' rows of the unit's query nodes (every E^T of the step is finished: barrier between the
// halves of P2).
template <int NTC, int H>
__device__ __forceinline__ void sf_yquarter(const unsigned (&vaddr)[NTC][4], const float4 (&p)[NTC], float *__restrict__ yrow, bool wr) {
    f32x4 y[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) y[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cur[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) { SF_LREAD32(cur[r][0], vaddr[0][r], H); SF_LREAD32(cur[r][1], vaddr[0][r], H + 64); }
    SF_WAIT_LGKM1(cur[0][0]);
    SF_OPAQUE(cur[0][1]);
#pragma unroll
    for (int r = 1; r < 4; ++r) { SF_OPAQUE(cur[r][0]); SF_OPAQUE(cur[r][1]); }
    __builtin_amdgcn_sched_barrier(0);
    sf_ychunk<0, NTC, H>(vaddr, p, y, cur);
    if (wr) {
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<float4 *>(yrow + H / 4 + 16 * t) = float4{y[t][0], y[t][1], y[t][2], y[t][3]};
    }
}

template <int NTC>
__device__ __forceinline__ void sf_attend_y(unsigned xb_lds, float *__restrict__ qy, const SgrGroupMeta &m, int ci, int tile, int half,
                                            int lane, const float4 *__restrict__ pt) {
    const int fi = lane & 15, fq = lane >> 4;
    const int nn = m.nn[ci], ws = m.wstart[ci];
    auto row_of = [&](int n) { n = n < nn ? n : nn - 1; return n == 0 ? ci : ws + n - 1; };
    unsigned vaddr[NTC][4];
#pragma unroll
    for (int a = 0; a < NTC; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) vaddr[a][r] = xb_lds + (unsigned)(row_of(16 * a + 4 * fq + r) * SF_LD + 128 * half + fi) * 4u;
    float4 p[NTC];
#pragma unroll
    for (int a = 0; a < NTC; ++a) p[a] = pt[a * 64 + lane];
    const bool wr = 16 * tile + fi < nn;
    float *yrow = qy + row_of(16 * tile + fi) * SF_LD + 128 * half + 4 * fq;
    sf_yquarter<NTC, 0>(vaddr, p, yrow, wr);
    sf_yquarter<NTC, 256>(vaddr, p, yrow, wr);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The LAST graph step on the vector ALU (round 4).  Only node 0 of every graph is read after it (Fusionmodule.py:443: sim_emb[:, 0]), so
// the step is, per caption: q' = W' x_0 + v (one row), one softmax row over the graph's nodes, y = sum_j p_j x_j, x_0' = relu(W_g y + b).
// As 16 x 16 x 4 MFMA tiles this was 16 node columns of which ncap (2-5) were live for the two projections, and 16 query columns of
// which ONE was live for the attention: 8.2 k + 6.1 k + 8.2 k cycles of the matrix pipe per 32-row item, a quarter of all the MFMA time
// of the kernel, for 2 % of its useful flop.  fp32 MFMA and the vector ALU are the same ALU on gfx950 (their times add), so the honest
// price of the step is its useful flop at the packed-FMA rate: ~1.1 k cycles per projection, < 1 k for the attention.
//
// Projection: dst[n][o] = act(sum_k WT[k][o] src[n][k] + bias[o]) for the rows n < ncap (LDS rows 0 .. ncap - 1 = the global nodes).
// Wave w owns outputs 32 w .. 32 w + 31; lane = (k slice ks = lane >> 3 of 32 k, output quad oq = lane & 7): one 16-byte weight load per
// k (8 x 128 contiguous bytes per wave instruction), four rows per pass over the weights; the eight k slices are summed across lanes
// (l ^ 8: DPP row_ror 8; l ^ 16, l ^ 32: the permlane swaps of sf_rows_sum) in a fixed order -- a row's result does not depend on which
// group or slot its caption was packed into.
template <int CTRL>
__device__ __forceinline__ float sf_dpp(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sf_row16_sum(float v) {      // all-reduce over the 16 lanes of a row (rotations: every lane adds the same pairs)
    v += sf_dpp<0x128>(v); v += sf_dpp<0x124>(v); v += sf_dpp<0x122>(v); v += sf_dpp<0x121>(v);
    return v;
}
__device__ __forceinline__ float sf_row16_max(float v) {
    v = fmaxf(v, sf_dpp<0x128>(v)); v = fmaxf(v, sf_dpp<0x124>(v)); v = fmaxf(v, sf_dpp<0x122>(v)); v = fmaxf(v, sf_dpp<0x121>(v));
    return v;
}

// NR rows per pass over the weights (row pairs past the group are skipped: wave-uniform branches), KB k per batch of weight loads
// (the next batch is in flight while this one's FMAs run).  The one-workgroup-per-CU kernel has the registers for NR = 8, KB = 8 and
// needs them -- nobody else hides its L2 round trips; with two workgroups per CU (128 registers a wave) NR = KB = 4.  The first batch
// is requested by the caller BEFORE the barrier that precedes the projection (sf_last_weights0).  The order of a row's sum is the
// same for every (NR, KB).
template <int KB>
__device__ __forceinline__ void sf_last_weights0(const float *__restrict__ wT, int wave, int lane, float4 (&w)[KB]) {
    const float4 *wp = reinterpret_cast<const float4 *>(wT + (size_t)(32 * (lane >> 3)) * SF_S + 32 * wave + 4 * (lane & 7));
#pragma unroll
    for (int j = 0; j < KB; ++j) w[j] = wp[j * 64];
}
template <int NR, int KB>
__device__ __forceinline__ void sf_last_project(const float *__restrict__ src, float *__restrict__ dst, const float *__restrict__ wT,
                                                const float *__restrict__ bias, int ncap, int wave, int lane, float4 (&w)[KB]) {
    const int oq = lane & 7, ks = lane >> 3, o0 = 32 * wave + 4 * oq;
    const float4 *wp = reinterpret_cast<const float4 *>(wT + (size_t)(32 * ks) * SF_S + o0);      // + 64 float4 per k
    const float4 bv = *reinterpret_cast<const float4 *>(bias + o0);
    for (int n0 = 0; n0 < ncap; n0 += NR) {
        const int npair = ((ncap - n0 < NR ? ncap - n0 : NR) + 1) >> 1;       // live row pairs of this pass (wave-uniform)
        float4 acc[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = float4{0.f, 0.f, 0.f, 0.f};
        const float *xr[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) xr[r] = src + (n0 + r < ncap ? n0 + r : ncap - 1) * SF_LD + 32 * ks;   // an odd count: the last row twice (dropped)
        if (n0 > 0) {
#pragma unroll
            for (int j = 0; j < KB; ++j) w[j] = wp[j * 64];
        }
        // a rolled loop: fully unrolled, hipcc hoists all 32 loads and the LDS reads of every batch and spills hundreds of registers
#pragma unroll 1
        for (int jb = 0; jb < 32 / KB; ++jb) {
            float4 wn[KB];
            const int jn = jb < 32 / KB - 1 ? jb + 1 : 32 / KB - 1;          // (the last batch is requested twice: branch-free)
#pragma unroll
            for (int j = 0; j < KB; ++j) wn[j] = wp[(KB * jn + j) * 64];
#pragma unroll
            for (int rp = 0; rp < NR / 2; ++rp) {
                if (rp < npair) {
#pragma unroll
                    for (int r = 2 * rp; r < 2 * rp + 2; ++r)
#pragma unroll
                        for (int q = 0; q < KB / 4; ++q) {
                            const float4 xa = *reinterpret_cast<const float4 *>(xr[r] + KB * jb + 4 * q);
                            const float xs[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float4 wj = w[4 * q + j];
                                acc[r].x = fmaf(wj.x, xs[j], acc[r].x); acc[r].y = fmaf(wj.y, xs[j], acc[r].y);
                                acc[r].z = fmaf(wj.z, xs[j], acc[r].z); acc[r].w = fmaf(wj.w, xs[j], acc[r].w);
                            }
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < KB; ++j) w[j] = wn[j];
        }
#pragma unroll
        for (int rp = 0; rp < NR / 2; ++rp) {
            if (rp < npair) {
#pragma unroll
                for (int r = 2 * rp; r < 2 * rp + 2; ++r) {
                    float4 v;
                    v.x = sf_rows_sum(acc[r].x + sf_dpp<0x128>(acc[r].x)) + bv.x;
                    v.y = sf_rows_sum(acc[r].y + sf_dpp<0x128>(acc[r].y)) + bv.y;
                    v.z = sf_rows_sum(acc[r].z + sf_dpp<0x128>(acc[r].z)) + bv.z;
                    v.w = sf_rows_sum(acc[r].w + sf_dpp<0x128>(acc[r].w)) + bv.w;
                    if (ks == 0 && n0 + r < ncap) *reinterpret_cast<float4 *>(dst + (n0 + r) * SF_LD + o0) = v;
                }
            }
        }
    }
}

// (Round 4 also measured a second form for the one-workgroup-per-CU kernel -- wave = k slice, lane = output quad, x broadcast with
// v_readlane, the eight k-slice partials summed through LDS: no 16-byte LDS broadcasts and fully contiguous 1 KB weight loads, but
// two more barriers per item: 725.2 ms against 720.8 for the form above, same box.  Removed; profiles/r04/NOTES.md.)

// Attention of node 0 of caption slot `ci` (one wave): e_j = q' . x_j over the graph's nodes, softmax, y = sum_j p_j x_j -> the Q'/Y row
// of the global node.  Scores: lane = (key fi = lane & 15 of key tile a, feature quarter fq = lane >> 4), 16-byte LDS reads in the
// conflict-free pattern of the MFMA fragment reads, the four quarters summed by sf_rows_sum.  Values: lane = feature quad, the weight
// of node j read from the lane that holds it.  NTC = key tiles of the graph.
template <int NTC, int UNR>
__device__ __forceinline__ void sf_last_attend(const float *__restrict__ xb, const float *__restrict__ qy, const SgrGroupMeta &m, int ci, int lane,
                                               float *__restrict__ yout) {
    const int fi = lane & 15, fq = lane >> 4;
    const int nn = m.nn[ci], ws = m.wstart[ci];
    auto row_of = [&](int n) { n = n < nn ? n : nn - 1; return n == 0 ? ci : ws + n - 1; };
    const float *qrow = qy + ci * SF_LD + 64 * fq;
    const float *krow[NTC];
    float e[NTC];
#pragma unroll
    for (int a = 0; a < NTC; ++a) { krow[a] = xb + row_of(16 * a + fi) * SF_LD + 64 * fq; e[a] = 0.f; }
#pragma unroll UNR
    for (int j = 0; j < 16; ++j) {       // (rolled: fully unrolled, hipcc hoists all 16 (NTC + 1) LDS reads)
        const float4 q = *reinterpret_cast<const float4 *>(qrow + 4 * j);
#pragma unroll
        for (int a = 0; a < NTC; ++a) {
            const float4 x = *reinterpret_cast<const float4 *>(krow[a] + 4 * j);
            e[a] = fmaf(q.x, x.x, e[a]); e[a] = fmaf(q.y, x.y, e[a]); e[a] = fmaf(q.z, x.z, e[a]); e[a] = fmaf(q.w, x.w, e[a]);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < NTC; ++a) {
        e[a] = (16 * a + fi < nn) ? sf_rows_sum(e[a]) : -INFINITY;
        mx = fmaxf(mx, e[a]);
    }
    mx = sf_row16_max(mx);
    float den = 0.f;
#pragma unroll
    for (int a = 0; a < NTC; ++a) { e[a] = fast_exp(e[a] - mx); den += e[a]; }       // (exp(-inf) = 0 for the keys past the graph)
    const float inv = fast_rcp(sf_row16_sum(den));
    float4 y{0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < nn; ++n) {                                                   // (n is wave-uniform)
        float pv = e[0];
#pragma unroll
        for (int a = 1; a < NTC; ++a) pv = (n >> 4) == a ? e[a] : pv;
        const float pn = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(pv), n & 15)) * inv;
        const float4 x = *reinterpret_cast<const float4 *>(xb + (n == 0 ? ci : ws + n - 1) * SF_LD + 4 * lane);
        y.x = fmaf(pn, x.x, y.x); y.y = fmaf(pn, x.y, y.y); y.z = fmaf(pn, x.z, y.z); y.w = fmaf(pn, x.w, y.w);
    }
    *reinterpret_cast<float4 *>(yout + 4 * lane) = y;       // one 1 KB row per caption, straight from the registers
}
template <int ROWS>
__device__ __forceinline__ void sf_last_attend_n(const float *__restrict__ xb, const float *__restrict__ qy, const SgrGroupMeta &m, int ci, int lane,
                                                 float *__restrict__ yout) {
    const int ntc = (m.nn[ci] + 15) >> 4;
    if constexpr (ROWS > SF_SMALL) {
        switch (ntc) {
#ifndef ITR_SF_UNR
#define ITR_SF_UNR 4
#endif
            case 1: sf_last_attend<1, ITR_SF_UNR>(xb, qy, m, ci, lane, yout); break;
            case 2: sf_last_attend<2, ITR_SF_UNR>(xb, qy, m, ci, lane, yout); break;
            case 3: sf_last_attend<3, 2>(xb, qy, m, ci, lane, yout); break;
            default: sf_last_attend<4, 2>(xb, qy, m, ci, lane, yout); break;
        }
    } else {
        if (ntc == 1) sf_last_attend<1, 2>(xb, qy, m, ci, lane, yout);
        else sf_last_attend<2, 2>(xb, qy, m, ci, lane, yout);
    }
}

// P2 dispatch on a caption's key-tile count (a ROWS = 32 group has graphs of at most two tiles: the wider instantiations would only
// raise the kernel's register count past the 128 that two workgroups per CU leave a wave)
template <int ROWS>
__device__ __forceinline__ void sf_attend_e_n(int ntc, unsigned xb_lds, unsigned qy_lds, const SgrGroupMeta &m, int ci, int tile, int lane,
                                              float4 *__restrict__ pt) {
    if constexpr (ROWS > SF_SMALL) {
        switch (ntc) {
            case 1: sf_attend_e<1>(xb_lds, qy_lds, m, ci, tile, lane, pt); break;
            case 2: sf_attend_e<2>(xb_lds, qy_lds, m, ci, tile, lane, pt); break;
            case 3: sf_attend_e<3>(xb_lds, qy_lds, m, ci, tile, lane, pt); break;
            default: sf_attend_e<4>(xb_lds, qy_lds, m, ci, tile, lane, pt); break;
        }
    } else {
        if (ntc == 1) sf_attend_e<1>(xb_lds, qy_lds, m, ci, tile, lane, pt);
        else sf_attend_e<2>(xb_lds, qy_lds, m, ci, tile, lane, pt);
    }
}
template <int ROWS>
__device__ __forceinline__ void sf_attend_y_n(int ntc, unsigned xb_lds, float *__restrict__ qy, const SgrGroupMeta &m, int ci, int tile,
                                              int half, int lane, const float4 *__restrict__ pt) {
    if constexpr (ROWS > SF_SMALL) {
        switch (ntc) {
            case 1: sf_attend_y<1>(xb_lds, qy, m, ci, tile, half, lane, pt); break;
            case 2: sf_attend_y<2>(xb_lds, qy, m, ci, tile, half, lane, pt); break;
            case 3: sf_attend_y<3>(xb_lds, qy, m, ci, tile, half, lane, pt); break;
            default: sf_attend_y<4>(xb_lds, qy, m, ci, tile, half, lane, pt); break;
        }
    } else {
        if (ntc == 1) sf_attend_y<1>(xb_lds, qy, m, ci, tile, half, lane, pt);
        else sf_attend_y<2>(xb_lds, qy, m, ci, tile, half, lane, pt);
    }
}

template <int ROWS, int WG_PER_CU>
__global__ __launch_bounds__(SF_THREADS, 2 * WG_PER_CU) void sgr_fused_kernel(SgrFusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) char sf_smem[];
    float *xb = reinterpret_cast<float *>(sf_smem);
    float *qy = xb + ROWS * SF_LD;
    SgrGroupMeta &m = *reinterpret_cast<SgrGroupMeta *>(qy + ROWS * SF_LD);
    float4 *ptile = reinterpret_cast<float4 *>(reinterpret_cast<char *>(&m) + 2 * sizeof(SgrGroupMeta));      // [sf_ptiles(ROWS)][64] float4
    const unsigned xb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)sf_smem;
    const unsigned qy_lds = xb_lds + (unsigned)(ROWS * SF_LD * sizeof(float));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int64_t)blockIdx.x >= (int64_t)g.gcount[0] * g.nb) return;      // (the grid is sized for all the groups of the call)
    const int64_t grp = g.glist[blockIdx.x / g.nb], ii = blockIdx.x % g.nb;      // consecutive workgroups: one group, the images of the block
    int nstamp = 0;
    // (straight to memory: a local array indexed by a run-time count would live in scratch, inside loops with hand-counted asm loads)
#define SF_STAMP() { if (g.trace && tid == 0 && nstamp < 16) g.trace[(size_t)blockIdx.x * 20 + 2 + nstamp++] = __builtin_amdgcn_s_memtime(); }
    SF_STAMP()
    if (tid < (int)(sizeof(SgrGroupMeta) / 4)) reinterpret_cast<int32_t *>(&m)[tid] = reinterpret_cast<const int32_t *>(g.meta + grp)[tid];
    __syncthreads();
    const int ncap = m.ncap, nrows = m.nrows;
    if (ncap == 0) return;
    // first weight fragments of the first projection (every later projection receives its own from the one before it)
    const size_t wfo = (size_t)(2 * wave) * 16 * 64 + lane;
    f32x4 fa0, fa1;
    {
        const float4 t0 = g.wq[0][wfo], t1 = g.wq[0][wfo + 16 * 64];
        fa0 = f32x4{t0.x, t0.y, t0.z, t0.w};
        fa1 = f32x4{t1.x, t1.y, t1.z, t1.w};
    }
    // ---- node rows -> LDS (one 1 KB row per wave-instruction); rows past the group are zeroed (their products are never read)
    {
        float4 v[ROWS / SF_WAVES];
#pragma unroll
        for (int k = 0; k < ROWS / SF_WAVES; ++k) {
            const int r = wave + SF_WAVES * k;
            v[k] = float4{0.f, 0.f, 0.f, 0.f};
            if (r < nrows) {
                const float *src = r < ncap ? g.xglo + (ii * g.Nc + m.row_src[r]) * SF_S : g.xloc + (ii * g.ncols + m.row_src[r]) * SF_S;
                v[k] = reinterpret_cast<const float4 *>(src)[lane];
            }
        }
#pragma unroll
        for (int k = 0; k < ROWS / SF_WAVES; ++k)
            *reinterpret_cast<float4 *>(xb + (wave + SF_WAVES * k) * SF_LD + 4 * lane) = v[k];
    }
    __syncthreads();
    SF_STAMP()
    const int ng_all = (nrows + 15) >> 4;
    for (int k = 0; k + 1 < g.steps; ++k) {
        sf_project_n<false, ROWS>(ng_all, xb_lds, qy, g.wq[k], g.vq[k], wave, lane, fa0, fa1, g.wg[k]);
        __syncthreads();
        SF_STAMP()
        // P2.  Units are sorted by size; unit u goes to wave (u & 7) for u & 7 < 4 and to wave 11 - (u & 7) otherwise (a snake
        // over the four SIMDs: waves w and w + 4 share one, and share its matrix pipe).
        const int nu = m.nunit;
        for (int u = 0; u < nu; ++u) {
            const int wv = (u & 7) < 4 ? (u & 7) : 11 - (u & 7);
            if (wv != wave) continue;
            const int ci = m.unit_cap[u], tile = m.unit_tile[u];
            float4 *pt = ptile + (size_t)m.unit_poff[u] * 64;
            sf_attend_e_n<ROWS>((m.nn[ci] + 15) >> 4, xb_lds, qy_lds, m, ci, tile, lane, pt);
        }
        __syncthreads();
        SF_STAMP()
        // tasks = unit x feature half, task t -> wave t & 7 (units are sorted by size: waves w and w + 4, which share a SIMD, get the
        // halves of units two places apart)
        for (int t = wave; t < 2 * nu; t += SF_WAVES) {
            const int u = t >> 1, dq = t & 1;
            const int ci = m.unit_cap[u], tile = m.unit_tile[u];
            const float4 *pt = ptile + (size_t)m.unit_poff[u] * 64;
            sf_attend_y_n<ROWS>((m.nn[ci] + 15) >> 4, xb_lds, qy, m, ci, tile, dq, lane, pt);
        }
        __syncthreads();
        SF_STAMP()
        sf_project_n<true, ROWS>(ng_all, qy_lds, xb, g.wg[k], g.bg[k], wave, lane, fa0, fa1, g.wq[k + 2 < g.steps ? k + 1 : 0]);
        __syncthreads();
        SF_STAMP()
    }
    // ---- the last step: node 0 of every graph only, on the vector ALU (see sf_last_project)
    constexpr int LNR = ROWS > SF_SMALL ? ITR_SF_LNR : 4, LKB = ROWS > SF_SMALL ? ITR_SF_LKB : 4;
    float4 lw[LKB];
    sf_last_weights0<LKB>(g.wqT_last, wave, lane, lw);
    sf_last_project<LNR, LKB>(xb, qy, g.wqT_last, g.vq[g.steps - 1], ncap, wave, lane, lw);
    __syncthreads();
    SF_STAMP()
    // y of node 0 -> memory; X'_0 = relu(W_g y + b) and sigmoid(sim_eval_w . x_0 + b) (Fusionmodule.py:443-444) run as one GEMM + one
    // small kernel over ALL the graphs of the image block (sgraf.hip): there the weight is read once per 128 rows, here it was
    // streamed through the CU once per item for 2-5 live rows
    for (int ci = wave; ci < ncap; ci += SF_WAVES)
        sf_last_attend_n<ROWS>(xb, qy, m, ci, lane, g.y0 + (ii * g.Nc + m.cap_id[ci]) * SF_S);
    SF_STAMP()
    SF_STAMP()
    SF_STAMP()
    if (g.trace && tid == 0) {
        SF_STAMP()
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        unsigned long long *t = g.trace + (size_t)blockIdx.x * 20;
        t[0] = ((unsigned long long)xcc << 32) | hw;
        t[1] = ((unsigned long long)nrows << 32) | (unsigned)(m.nunit << 8) | (unsigned)ncap;
    }
#undef SF_STAMP
}

// Persistent form (default; ITR_SGR_PERSISTENT=0 launches one workgroup per (image, group) instead): one workgroup per CU walks
// the (group, image) list with stride gridDim.x.  What it buys is the 64 KB of node rows of the NEXT item, requested when the last
// step's attention starts (11 k cycles with nothing else on the vector-memory counter: the k-blocks of the last projection wait
// with vmcnt(0), and loads complete in order) into 8 registers per lane, and stored into the X buffer after the scores of the
// current item -- instead of a cold load phase (5-6 k cycles at the ~11 B/clk a CU streams from HBM) and a launch gap (2 k) per item
// (SGR 1k x 5k: 763 -> 753 ms).
// The next item's group record is fetched by waves 6 and 7 during the first step's score phase.
template <int ROWS, int WG_PER_CU>
__global__ __launch_bounds__(SF_THREADS, 2 * WG_PER_CU) void sgr_fused_persistent_kernel(SgrFusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) char sf_smem[];
    float *xb = reinterpret_cast<float *>(sf_smem);
    float *qy = xb + ROWS * SF_LD;
    SgrGroupMeta *m2 = reinterpret_cast<SgrGroupMeta *>(qy + ROWS * SF_LD);      // two records: current / next item
    float4 *ptile = reinterpret_cast<float4 *>(reinterpret_cast<char *>(m2) + 2 * sizeof(SgrGroupMeta));
    const unsigned xb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)sf_smem;
    const unsigned qy_lds = xb_lds + (unsigned)(ROWS * SF_LD * sizeof(float));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t total = (int64_t)g.gcount[0] * g.nb;        // this class's groups x the images of the block (device-side count)
    int64_t item = blockIdx.x;
    if (item >= total) return;
    int cur = 0;
    // ---- the first item: group record and node rows the plain way
    if (tid < (int)(sizeof(SgrGroupMeta) / 4))
        reinterpret_cast<int32_t *>(&m2[0])[tid] = reinterpret_cast<const int32_t *>(g.meta + g.glist[item / g.nb])[tid];
    __syncthreads();
    {
        const SgrGroupMeta &m = m2[0];
        const int64_t ii = item % g.nb;
        float4 v[ROWS / SF_WAVES];
#pragma unroll
        for (int k = 0; k < ROWS / SF_WAVES; ++k) {
            const int r = wave + SF_WAVES * k;
            v[k] = float4{0.f, 0.f, 0.f, 0.f};
            if (r < m.nrows) {
                const float *src = r < m.ncap ? g.xglo + (ii * g.Nc + m.row_src[r]) * SF_S : g.xloc + (ii * g.ncols + m.row_src[r]) * SF_S;
                v[k] = reinterpret_cast<const float4 *>(src)[lane];
            }
        }
#pragma unroll
        for (int k = 0; k < ROWS / SF_WAVES; ++k)
            *reinterpret_cast<float4 *>(xb + (wave + SF_WAVES * k) * SF_LD + 4 * lane) = v[k];
    }
    const size_t wfo = (size_t)(2 * wave) * 16 * 64 + lane;
    f32x4 fa0, fa1;
    {
        const float4 t0 = g.wq[0][wfo], t1 = g.wq[0][wfo + 16 * 64];
        fa0 = f32x4{t0.x, t0.y, t0.z, t0.w};
        fa1 = f32x4{t1.x, t1.y, t1.z, t1.w};
    }
    __syncthreads();
    for (;;) {
        const SgrGroupMeta &m = m2[cur];
        const int ncap = m.ncap, nrows = m.nrows;
        const int64_t ii = item % g.nb;
        const int64_t nxt = item + gridDim.x;
        const bool has_next = nxt < total;                      // workgroup-uniform
        const int ng_all = (nrows + 15) >> 4;
        const bool meta_wave = has_next && wave >= 6;
        if (meta_wave && g.steps == 1)      // a single step is also the last one: the record must be there before its attention phase
            reinterpret_cast<int32_t *>(&m2[cur ^ 1])[tid - 384] = reinterpret_cast<const int32_t *>(g.meta + g.glist[nxt / g.nb])[tid - 384];
        constexpr int LNR = ROWS > SF_SMALL ? ITR_SF_LNR : 4, LKB = ROWS > SF_SMALL ? ITR_SF_LKB : 4;
        float4 lw[LKB];          // first batch of the last step's weights: requested ahead of the barrier that precedes their projection
        constexpr bool early = ROWS > SF_SMALL;      // (two workgroups per CU: no registers to spare, and the other workgroup hides the trip)
        if (early && g.steps == 1) sf_last_weights0<LKB>(g.wqT_last, wave, lane, lw);
        for (int k = 0; k + 1 < g.steps; ++k) {
            sf_project_n<false, ROWS>(ncap == 0 ? 1 : ng_all, xb_lds, qy, g.wq[k], g.vq[k], wave, lane, fa0, fa1, g.wg[k]);
            __syncthreads();
            // the next item's group record: one dword per lane of waves 6 and 7, fetched while the other waves run the first step's
            // score units (the unit map gives waves 6 and 7 the smallest units, or none)
            if (k == 0 && meta_wave)
                reinterpret_cast<int32_t *>(&m2[cur ^ 1])[tid - 384] = reinterpret_cast<const int32_t *>(g.meta + g.glist[nxt / g.nb])[tid - 384];
            const int nu = m.nunit;
            for (int u = 0; u < nu; ++u) {
                const int wv = (u & 7) < 4 ? (u & 7) : 11 - (u & 7);
                if (wv != wave) continue;
                const int ci = m.unit_cap[u], tile = m.unit_tile[u];
                float4 *pt = ptile + (size_t)m.unit_poff[u] * 64;
                sf_attend_e_n<ROWS>((m.nn[ci] + 15) >> 4, xb_lds, qy_lds, m, ci, tile, lane, pt);
            }
            __syncthreads();
            for (int t = wave; t < 2 * nu; t += SF_WAVES) {
                const int u = t >> 1, dq = t & 1;
                const int ci = m.unit_cap[u], tile = m.unit_tile[u];
                const float4 *pt = ptile + (size_t)m.unit_poff[u] * 64;
                sf_attend_y_n<ROWS>((m.nn[ci] + 15) >> 4, xb_lds, qy, m, ci, tile, dq, lane, pt);
            }
            __syncthreads();
            // (the last MFMA projection of the item hands over the first fragments of the NEXT item's first projection)
            sf_project_n<true, ROWS>(ncap == 0 ? 1 : ng_all, qy_lds, xb, g.wg[k], g.bg[k], wave, lane, fa0, fa1, g.wq[k + 2 < g.steps ? k + 1 : 0]);
            if (early && k + 2 == g.steps) sf_last_weights0<LKB>(g.wqT_last, wave, lane, lw);
            __syncthreads();
        }
        if (!early) sf_last_weights0<LKB>(g.wqT_last, wave, lane, lw);
        // ---- the last step: node 0 of every graph only, on the vector ALU (see sf_last_project)
        sf_last_project<LNR, LKB>(xb, qy, g.wqT_last, g.vq[g.steps - 1], ncap, wave, lane, lw);
        __syncthreads();
        f32x4 pre[ROWS / SF_WAVES];
        if (has_next) {
            // node rows of the next item -> registers, in flight during the attention.  Branch-free (a conditional
            // load makes hipcc merge "loaded | zero" right here): rows past the next group re-read its last row and are zeroed when
            // they are stored.  Loads hipcc counts: their registers are safe.
            const SgrGroupMeta &mn = m2[cur ^ 1];
            const int64_t iin = nxt % g.nb;
            const int nrn = mn.nrows > 0 ? mn.nrows : 1;
#pragma unroll
            for (int q = 0; q < ROWS / SF_WAVES; ++q) {
                int r = wave + SF_WAVES * q;
                r = r < nrn ? r : nrn - 1;
                const float *src = r < mn.ncap ? g.xglo + (iin * g.Nc + mn.row_src[r]) * SF_S : g.xloc + (iin * g.ncols + mn.row_src[r]) * SF_S;
                pre[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src) + lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // y of node 0 -> memory: the last graph projection and the score are one GEMM + one small kernel over all the graphs of the
        // image block (sgraf.hip)
        for (int ci = wave; ci < ncap; ci += SF_WAVES)
            sf_last_attend_n<ROWS>(xb, qy, m, ci, lane, g.y0 + (ii * g.Nc + m.cap_id[ci]) * SF_S);
        if (!has_next) break;
        __syncthreads();                                        // every wave is done with the X buffer of this item
#pragma unroll
        for (int q = 0; q < ROWS / SF_WAVES; ++q) {
            const bool live = wave + SF_WAVES * q < m2[cur ^ 1].nrows;
            *reinterpret_cast<f32x4 *>(xb + (wave + SF_WAVES * q) * SF_LD + 4 * lane) = live ? pre[q] : f32x4{0.f, 0.f, 0.f, 0.f};   // (landed long ago)
        }
        cur ^= 1;
        item = nxt;
        __syncthreads();
    }
}

// workspace: [group records][class of every group][class lists: 2 x n_groups][counts][bad-caption marks: n_caps][packed weights]
struct SgrWs {
    SgrGroupMeta *meta;
    int8_t *cls;
    int32_t *glist, *gcount, *cap_bad;
    float4 *frag;
    float *wT;                   // [256][256]: the last step's folded query weight, transposed
    size_t bytes;
};
static SgrWs sgr_carve(void *ws, int64_t n_groups, int64_t n_caps, int sgr_step) {
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    char *p = static_cast<char *>(ws);
    SgrWs w;
    w.meta = reinterpret_cast<SgrGroupMeta *>(p); p += al((size_t)n_groups * sizeof(SgrGroupMeta));
    w.cls = reinterpret_cast<int8_t *>(p); p += al((size_t)n_groups);
    w.glist = reinterpret_cast<int32_t *>(p); p += al((size_t)2 * n_groups * 4);
    w.gcount = reinterpret_cast<int32_t *>(p); p += 256;
    w.cap_bad = reinterpret_cast<int32_t *>(p); p += al((size_t)n_caps * 4);
    w.frag = reinterpret_cast<float4 *>(p); p += (size_t)sgr_step * 2 * SF_S * SF_S * 4;
    w.wT = reinterpret_cast<float *>(p); p += (size_t)SF_S * SF_S * 4;
    w.bytes = (size_t)(p - static_cast<char *>(ws)) + 256;
    return w;
}
size_t sgr_fused_workspace_bytes(int64_t n_groups, int64_t n_caps, int sgr_step) { return sgr_carve(nullptr, n_groups, n_caps, sgr_step).bytes; }

// One-time preparation per itr_sgraf_scores call: group records, the two class lists, fragment-ordered weights.
int sgr_fused_prepare(const int32_t *grp_begin, const int32_t *grp_order, int64_t n_groups, int64_t n_caps, const int32_t *cap_len,
                      const int32_t *cap_col, const float *const *wq, const float *const *wg, int sgr_step, void *ws, int *bad_flag, hipStream_t st) {
    const SgrWs w = sgr_carve(ws, n_groups, n_caps, sgr_step);
    ITR_CHECK_HIP(hipMemsetAsync(w.cap_bad, 0, (size_t)n_caps * 4, st));
    hipLaunchKernelGGL(sgr_group_meta_kernel, dim3((unsigned)ceil_div(n_groups, 256)), dim3(256), 0, st, grp_begin, grp_order, cap_len, cap_col,
                       n_groups, w.meta, bad_flag, w.cls, w.cap_bad, n_caps);
    ITR_CHECK_LAUNCH("sgr group meta");
    hipLaunchKernelGGL(sgr_group_classify_kernel, dim3(1), dim3(1024), 0, st, w.cls, n_groups, w.glist, w.gcount);
    ITR_CHECK_LAUNCH("sgr group classes");
    for (int k = 0; k < sgr_step; ++k) {
        hipLaunchKernelGGL(sgr_pack_weight_kernel, dim3(64), dim3(256), 0, st, wq[k], w.frag + (size_t)(2 * k) * SF_S * SF_S / 4);
        hipLaunchKernelGGL(sgr_pack_weight_kernel, dim3(64), dim3(256), 0, st, wg[k], w.frag + (size_t)(2 * k + 1) * SF_S * SF_S / 4);
    }
    ITR_CHECK_LAUNCH("sgr pack weights");
    hipLaunchKernelGGL(sgr_transpose_weight_kernel, dim3(SF_S * SF_S / 256), dim3(256), 0, st, wq[sgr_step - 1], w.wT);
    ITR_CHECK_LAUNCH("sgr transpose weights");
    return ITR_OK;
}

// After the last block of images: NaN into the score columns of the captions whose group was refused (a wrong hand-made plan).
int sgr_fused_finish(void *ws, int64_t n_groups, int64_t n_caps, int sgr_step, int64_t Ni, float *S, int64_t ldS, hipStream_t st) {
    const SgrWs w = sgr_carve(ws, n_groups, n_caps, sgr_step);
    if (n_caps == 0 || Ni == 0) return ITR_OK;
    hipLaunchKernelGGL(sgr_poison_kernel, dim3((unsigned)ceil_div(n_caps, 256)), dim3(256), 0, st, w.cap_bad, n_caps, Ni, S, ldS);
    ITR_CHECK_LAUNCH("sgr refused groups");
    return ITR_OK;
}

int allow_dynamic_lds(const void *kernel, size_t bytes);      // scan_train.hip: once per (kernel, device), under a mutex

static int sgr_cu_count(int64_t *cus) {
    static int cus_of[64] = {};
    int dev = 0;
    ITR_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !cus_of[dev]) {            // (idempotent: racing callers store the same value)
        hipDeviceProp_t prop;
        ITR_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        cus_of[dev] = prop.multiProcessorCount;
    }
    *cus = (dev >= 0 && dev < 64 && cus_of[dev] > 0) ? cus_of[dev] : 256;
    return ITR_OK;
}

// One class of groups.  ROWS = 32: two workgroups per CU; ROWS = 64: one.
template <int ROWS, int WG_PER_CU>
static int sgr_fused_launch_class(SgrFusedArgs g, int cls_index, int64_t n_groups, const SgrWs &w, bool persistent, const char *trace_path, hipStream_t st) {
    constexpr size_t lds = sf_lds_bytes(ROWS);
    int rc = allow_dynamic_lds(reinterpret_cast<const void *>(sgr_fused_kernel<ROWS, WG_PER_CU>), lds);
    if (rc == ITR_OK) rc = allow_dynamic_lds(reinterpret_cast<const void *>(sgr_fused_persistent_kernel<ROWS, WG_PER_CU>), lds);
    if (rc != ITR_OK) return rc;
    g.glist = w.glist + (cls_index ? n_groups : 0);
    g.gcount = w.gcount + cls_index;
    const int64_t grid = n_groups * g.nb;                 // upper bound: the class's own count lives on the device
    ITR_REQUIRE(grid < (1ll << 31), "sgr_fused: grid too large");
    if (trace_path) {
        // Debug only (tools/sgr_trace.py): ITR_SGR_TRACE=<file> makes every launch synchronous and rewrites <file>.<class> with one record
        // per workgroup (hardware id, group shape, s_memtime after every phase).
        const size_t bytes = (size_t)grid * 20 * sizeof(unsigned long long);
        ITR_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&g.trace), bytes));
        ITR_CHECK_HIP(hipMemsetAsync(g.trace, 0, bytes, st));
        hipLaunchKernelGGL((sgr_fused_kernel<ROWS, WG_PER_CU>), dim3((unsigned)grid), dim3(SF_THREADS), lds, st, g);
        ITR_CHECK_LAUNCH("sgr_fused");
        ITR_CHECK_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> host((size_t)grid * 20);
        ITR_CHECK_HIP(hipMemcpy(host.data(), g.trace, bytes, hipMemcpyDeviceToHost));
        ITR_CHECK_HIP(hipFree(g.trace));
        char path[1024];
        snprintf(path, sizeof(path), "%s.%d", trace_path, ROWS);
        if (FILE *f = fopen(path, "wb")) { fwrite(host.data(), 1, bytes, f); fclose(f); }
        return ITR_OK;
    }
    if (persistent) {
        int64_t cus = 256;
        rc = sgr_cu_count(&cus);
        if (rc != ITR_OK) return rc;
        const int64_t resident = cus * WG_PER_CU;
        hipLaunchKernelGGL((sgr_fused_persistent_kernel<ROWS, WG_PER_CU>), dim3((unsigned)(grid < resident ? grid : resident)), dim3(SF_THREADS), lds, st, g);
        ITR_CHECK_LAUNCH("sgr_fused (persistent)");
        return ITR_OK;
    }
    hipLaunchKernelGGL((sgr_fused_kernel<ROWS, WG_PER_CU>), dim3((unsigned)grid), dim3(SF_THREADS), lds, st, g);
    ITR_CHECK_LAUNCH("sgr_fused");
    return ITR_OK;
}

// Runs the graph steps up to the last step's attention; y0 [nb][Nc][256] receives y of node 0 of every graph.  The caller finishes:
// x_0 = relu(W_g y + b) as one GEMM over nb * Nc rows, then sigmoid(sim_eval_w . x_0 + b) (sgraf.hip).
int sgr_fused_scores(const float *xloc, const float *xglo, void *ws, int64_t n_groups, int64_t n_caps, int64_t nb, int64_t Nc, int64_t ncols,
                     const float *const *vq, const float *const *bg, int sgr_step, float *y0, bool persistent_walk, hipStream_t st) {
    if (nb == 0 || n_groups == 0) return ITR_OK;
    const SgrWs w = sgr_carve(ws, n_groups, n_caps, sgr_step);
    SgrFusedArgs g;
    memset(&g, 0, sizeof(g));
    g.xloc = xloc; g.xglo = xglo;
    g.meta = w.meta;
    g.n_groups = n_groups; g.nb = nb; g.Nc = Nc; g.ncols = ncols;
    for (int k = 0; k < sgr_step; ++k) {
        g.wq[k] = w.frag + (size_t)(2 * k) * SF_S * SF_S / 4;
        g.wg[k] = w.frag + (size_t)(2 * k + 1) * SF_S * SF_S / 4;
        g.vq[k] = vq[k];
        g.bg[k] = bg[k];
    }
    g.steps = sgr_step;
    g.wqT_last = w.wT;
    g.y0 = y0;
    static const char *trace_env = ITR_EXP_ENV("ITR_SGR_TRACE");
    const char *trace_path = (trace_env && *trace_env) ? trace_env : nullptr;
    const bool persistent = persistent_walk;      // (itr_sgraf_scores flag ITR_SGRAF_NON_PERSISTENT: one workgroup per (image, group))
    // the large class first (it is empty for every caption set of at most 31 words when the plan comes from itr_sgr_plan_node_groups:
    // its workgroups read a zero count and leave), then the small one
    int rc = sgr_fused_launch_class<SF_ROWS, 1>(g, 1, n_groups, w, persistent, trace_path, st);
    if (rc != ITR_OK) return rc;
    return sgr_fused_launch_class<SF_SMALL, 2>(g, 0, n_groups, w, persistent, trace_path, st);
}

}  // namespace itr

// Pure CPU.  The planner of SGR's fused graph steps: whole captions are bin-packed (exact fill, pack_plan.h, like itr_scan_plan_tiles) by
// NODE count = words + 1 into groups of at most 16 captions and at most 64 node rows (one workgroup per CU).  small_rows = 64: that is
// all (the default of the Python layer: measured fastest).  small_rows = 32: captions of at most 31 words go into
// groups of <= 32 node rows instead, which run two workgroups per CU (kept: same scores, 1.2 % slower on the bench captions).
// group_begin_host[n_groups + 1] indexes group_order_host[Nc].
extern "C" int itr_sgr_plan_node_groups(const int32_t *cap_len_host, int64_t Nc, int small_rows, int32_t *group_begin_host,
                                        int32_t *group_order_host, int64_t *n_groups) {
    using namespace itr;
    ITR_REQUIRE(cap_len_host && group_begin_host && group_order_host && n_groups, "itr_sgr_plan_node_groups: null pointer");
    ITR_REQUIRE(small_rows == SF_SMALL || small_rows == SF_ROWS, "itr_sgr_plan_node_groups: small_rows must be %d or %d", SF_SMALL, SF_ROWS);
    ITR_REQUIRE(Nc >= 0 && Nc < 0x7fffffffLL, "itr_sgr_plan_node_groups: bad caption count");
    static_assert(PACK_MAXN == SF_MAXCAP, "planner bins hold SF_MAXCAP graphs");
    std::vector<PackBin> bins;
    for (int pass = (small_rows == SF_ROWS ? 1 : 0); pass < 2; ++pass) {
        const int cap_rows = pass == 0 ? SF_SMALL : SF_ROWS, lo = (pass == 0 || small_rows == SF_ROWS) ? 2 : SF_SMALL + 1;
        std::vector<std::vector<int32_t>> by_n(cap_rows + 1);
        for (int64_t c = 0; c < Nc; ++c) {
            const int n = cap_len_host[c] + 1;
            ITR_REQUIRE(n >= 2, "itr_sgr_plan_node_groups: caption %lld has length %d (< 1)", (long long)c, n - 1);
            ITR_UNSUPPORTED(n > SF_ROWS, "itr_sgr_plan_node_groups: caption %lld has %d words; the fused graph steps hold <= %d", (long long)c,
                            n - 1, SF_ROWS - 1);
            if (n >= lo && n <= cap_rows) by_n[n].push_back((int32_t)c);
        }
        pack_bins(by_n, cap_rows, SF_MAXCAP, bins);      // (graphs have >= 2 nodes: a last free row stays free)
    }
    int64_t pos = 0;
    for (size_t t = 0; t < bins.size(); ++t) {
        group_begin_host[t] = (int32_t)pos;
        for (int k = 0; k < bins[t].n; ++k) group_order_host[pos++] = bins[t].item[k];
    }
    group_begin_host[bins.size()] = (int32_t)pos;
    *n_groups = (int64_t)bins.size();
    return ITR_OK;
}
