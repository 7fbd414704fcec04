// Recall@K ranker (itr/metricmodule/evaluation.py:156-222) without sorting.
//
// The reference argsorts every row (i2t) and every column (t2i) of the float64 similarity
// matrix on the host.  The rank of a ground-truth item is just a count:
//     rank(q, gt) = #{k : S[q,k] > S[q,gt]} + #{k > gt : S[q,k] == S[q,gt]}
// (with np.argsort(...)[::-1] the higher index wins exact ties), so one streaming
// pass over S per direction is enough: integer compares, HBM-bound (4 bytes/pair/direction).
//
// fp32 matrices: ONE kernel reads every element once and serves both directions (rank_fused_kernel below).
//   t2i : one lane owns 4 consecutive caption columns (float4, coalesced across the wave) and
//         walks down a chunk of image rows; partial counts are added with one atomic per column
//         so row blocks (other workgroups, or other GPUs after an all-reduce) just sum.
//   i2t : the same elements against the row's best ground-truth key, counted with lane masks on the scalar unit.
// float64 matrices (ensemble averages) keep one pass per direction.
// S may be a row block of the global matrix (multi-GPU row sharding): row0 is the global index
// of its first row, and s_gt[] carries the GT score of every caption (gathered over ranks).
#include "itr_common.h"
#include <mutex>

namespace itr {

constexpr int MAX_IMDIV = 8;
constexpr int RANK_THREADS = 256;

__global__ void gather_gt_kernel(const float *__restrict__ S, int64_t ldS, int64_t row0, int64_t nrows,
                                 int64_t Nc, int im_div, float *__restrict__ s_gt) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Nc) return;
    const int64_t g = j / im_div - row0;
    if (g >= 0 && g < nrows) s_gt[j] = S[g * ldS + j];
}

// (score, index) as ONE 64-bit key: larger score first, then the larger index -- exactly the tie rule of the counts
// (#{S_k > S_gt} + #{k > gt : S_k == S_gt} = #{key_k > key_gt}) and of the top-1 (np.argsort(...)[::-1]: the higher index wins).
// Keys of distinct elements are distinct, so the counts are those of a TOTAL order -- which is what makes i2t cheap: the best of an
// image's im_div ground-truth captions is the one with the LARGEST key, and  min_g #{key > gkey_g} = #{key > max_g gkey_g}:
// ONE compare per element instead of im_div.
// score_key canonicalises first: e + 0.0f folds -0.0 into +0.0 (equal as floats, different bit patterns) and fminf(., inf) maps NaN
// to +inf: a NaN score sorts as the LARGEST value, like np.argsort (NaN last ascending = first after [::-1]); the float64 kernels use
// the same rule.
__device__ __forceinline__ uint32_t score_key(float e) {
    const uint32_t u = __float_as_uint(fminf(e + 0.0f, INFINITY));
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);      // = float_order_key, as three integer instructions
}
__device__ __forceinline__ unsigned long long key64(uint32_t hi, uint32_t lo) { return ((unsigned long long)hi << 32) | lo; }
__device__ __forceinline__ unsigned long long rank_key(float e, unsigned idx) { return key64(score_key(e), idx); }

// ---- ONE pass over S for both directions (round 5) ----------------------------------------------------------
// Rounds 1-4 read the score matrix twice (a row pass for i2t, a column pass for t2i: 1.0 GB for the 0.5 GB matrix of 5k x 25k,
// 94 + 122 us).  Now a workgroup owns a tile of RF_ROWS rows x 1 024 columns and every element it loads serves both directions:
//   t2i (columns): a lane owns 4 consecutive columns (one float4 per row, coalesced across the wave) and keeps, per column, the
//       count #{key > gkey_col} and the running best (score key, row) while it walks down the rows -- as before;
//   i2t (rows):    the row's ONE ground-truth key (above) sits in scalar registers; per element one v_cmp_gt_u64 whose lane mask
//       is counted on the SCALAR unit (s_bcnt1) -- no per-lane counters, no cross-lane reduction of counts; the row's top-1 is a
//       6-step DPP max of the lane maxima.  Row results are parked in lane (row mod 64) of three registers (v_writelane) and
//       leave the workgroup once per 64 rows: the four waves are combined through LDS, then one atomicAdd / one 64-bit
//       atomicMax per row (other column blocks -- and nobody else -- add to the same row).
// RF_U rows are requested before the first is consumed (the old kernels had one 16-byte load per lane in flight and relied on
// occupancy alone: they were latency-bound, not ALU- or HBM-bound).
#ifndef ITR_RF_U
#define ITR_RF_U 4
#endif
#ifndef ITR_RF_WAVES              // waves per SIMD the register allocation must admit
#define ITR_RF_WAVES 4
#endif
constexpr int RF_U = ITR_RF_U;    // rows in the register ring of a lane (RF_U - 1 in flight at all times); divides 64
// rows per workgroup: a multiple of 64 chosen per launch (itr_rank_counts) so that the grid fits the resident slots of the chip in ONE
// round where it can (1 000 workgroups of 128 rows on 768 slots were two rounds for 1.3 rounds of work)
constexpr int RF_COLS = RANK_THREADS * 4;

// max over the wave of a u32 (identity 0), returned uniform.  row_shr within the rows of 16 lanes, then the two row broadcasts.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));   // row_shr:1
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));   // row_shr:2
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));   // row_shr:4
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));   // row_shr:8  -> lane 15 of every row: the row's max
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));   // row_bcast:15 into rows 1 and 3
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));   // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// per local row: the key of its best ground-truth caption (~0: the image has none inside the matrix), zeroed accumulators
// ONE launch ahead of the fused kernel: per local row the key of its best ground-truth caption (~0: the image has none inside the
// matrix) and zeroed row accumulators; per column (single-call use: flags) the ground-truth score read from S itself and zeroed
// column accumulators -- a gather launch and three fill launches of ~5 us each went into that before, next to a 113 us kernel.
__global__ void rank_prepare_kernel(const float *__restrict__ S, int64_t ldS, int64_t row0, int64_t nrows, int64_t Nc, int im_div,
                                    unsigned long long *__restrict__ row_gkey, unsigned long long *__restrict__ row_best,
                                    int32_t *__restrict__ i2t_cnt, float *__restrict__ s_gt_out,
                                    int32_t *__restrict__ t2i_cnt, unsigned long long *__restrict__ t2i_best, int init_cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) {
        const int64_t gi = row0 + i;
        unsigned long long gk = 0;
        bool any = false;
        for (int g = 0; g < im_div; ++g) {
            const int64_t c = gi * im_div + g;
            if (c >= Nc) break;
            const unsigned long long k = rank_key(S[i * ldS + c], (unsigned)c);
            gk = k > gk ? k : gk;
            any = true;
        }
        row_gkey[i] = any ? gk : ~0ull;
        row_best[i] = 0;
        i2t_cnt[i] = 0;
    }
    if (i < Nc) {
        if (s_gt_out) s_gt_out[i] = S[(i / im_div - row0) * ldS + i];      // (every ground-truth row is local: checked by the caller)
        if (init_cols) { t2i_cnt[i] = 0; t2i_best[i] = 0; }
    }
}

// FULL: every lane of the workgroup owns four valid columns and the rows are 16-byte aligned (all but the last column block of an
// aligned matrix): plain float4 loads, no per-lane column guards.  Otherwise the guarded form (scalar loads, key 0 for columns
// past the end).
// DIAG: the tile may contain ground-truth pairs (the rows' GT columns / the columns' GT rows fall inside it: 1 tile in 40 at
// 5k x 25k).  Everywhere else the index half of a 64-bit key compare is the SAME for the whole tile -- every column of the tile
// lies on one side of the row's GT column, every row on one side of a column's GT row -- and
//     key64(ok, idx) > key64(g_ok, g_idx)   <=>   ok > g_ok - (idx > g_idx)
// becomes a 32-bit compare against a threshold fixed per row (scalar) / per column (once per tile).
template <bool FULL, bool DIAG, int U>
__device__ __forceinline__ void rank_tile(const float *__restrict__ S, int64_t ldS, int64_t row0, int64_t r_begin, int64_t r_end, int64_t Nc,
                                          int im_div, const float *__restrict__ s_gt, const unsigned long long *__restrict__ row_gkey,
                                          int32_t *__restrict__ i2t_cnt, unsigned long long *__restrict__ row_best,
                                          int32_t *__restrict__ t2i_cnt, unsigned long long *__restrict__ t2i_best,
                                          uint32_t (*s_rows)[RANK_THREADS / 64][64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cb = (int64_t)blockIdx.x * RF_COLS;
    const int64_t cw = cb + wave * 256;      // first column of this wave
    const int64_t c0 = cw + lane * 4;
    const int ncol = FULL ? 4 : (c0 >= Nc ? 0 : ((Nc - c0 >= 4) ? 4 : (int)(Nc - c0)));
    uint32_t cg_ok[4], cg_row[4], c_thr[4], b_ok[4], b_row[4];
    int cnt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {      // (score of the column's GT image, its row): a row counts when its key is larger
        cg_ok[u] = u < ncol ? score_key(s_gt[c0 + u]) : 0xffffffffu;
        cg_row[u] = u < ncol ? (uint32_t)((c0 + u) / im_div) : 0xffffffffu;
        c_thr[u] = u < ncol ? cg_ok[u] - ((uint32_t)(row0 + r_begin) > cg_row[u] ? 1u : 0u) : 0xffffffffu;      // (!DIAG)
        b_ok[u] = 0; b_row[u] = 0; cnt[u] = 0;
    }
    const uint32_t col0 = (uint32_t)c0;
    uint32_t rv_cnt = 0, rv_ok = 0, rv_col = 0;
    int64_t rc = r_begin;                                                     // first row of the 64-row chunk whose results sit in rv_*
    // one row of the tile (v = this lane's four scores of it)
    auto row_step = [&](const float4 &v, int64_t r) {
        const unsigned long long gk = row_gkey[r];                            // uniform address: a scalar load
        const uint32_t grow = (uint32_t)(row0 + r);
        const uint32_t r_thr = (uint32_t)(gk >> 32) - ((uint32_t)cb > (uint32_t)gk ? 1u : 0u);      // (!DIAG; scalar)
        const float e[4] = {v.x, v.y, v.z, v.w};
        uint32_t ok[4];
        int scnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = u < ncol ? score_key(e[u]) : 0u;                          // key 0 is below every real key: never counted, never the best
            // i2t: one compare against the row's GT key; the lane mask is counted on the scalar unit
            const bool above = DIAG ? key64(ok[u], u < ncol ? col0 + u : 0u) > gk : ok[u] > r_thr;
            scnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(above));
            // t2i: count + running best of the column (rows ascend: >= keeps the higher row among equal scores)
            cnt[u] += DIAG ? key64(ok[u], grow) > key64(cg_ok[u], cg_row[u]) : ok[u] > c_thr[u];
            const bool ge = ok[u] >= b_ok[u];
            b_ok[u] = ge ? ok[u] : b_ok[u];
            b_row[u] = ge ? grow : b_row[u];
        }
        // i2t top-1: the wave's maximum, the highest lane holding it, the highest of that lane's columns holding it (scalar unit)
        uint32_t lm = ok[0] > ok[1] ? ok[0] : ok[1];
        const uint32_t lm2 = ok[2] > ok[3] ? ok[2] : ok[3];
        lm = lm > lm2 ? lm : lm2;
        const uint32_t wm = wave_max_u32(lm);
        const int hl = 63 - __builtin_clzll(__builtin_amdgcn_ballot_w64(lm == wm));      // (non-empty: lm == wm on at least one lane)
        const uint32_t o3 = (uint32_t)__builtin_amdgcn_readlane((int)ok[3], hl), o2 = (uint32_t)__builtin_amdgcn_readlane((int)ok[2], hl),
                       o1 = (uint32_t)__builtin_amdgcn_readlane((int)ok[1], hl);
        const uint32_t wcol = (uint32_t)cw + (uint32_t)hl * 4u + (o3 == wm ? 3u : o2 == wm ? 2u : o1 == wm ? 1u : 0u);
        // three uniform values into lane (r - rc) of the row registers (gfx9's v_writelane takes ONE scalar operand besides m0: a lane
        // compare + three selects are as cheap)
        const bool mine = lane == (int)(r - rc);
        rv_cnt = mine ? (uint32_t)scnt : rv_cnt;
        rv_ok = mine ? wm : rv_ok;
        rv_col = mine ? wcol : rv_col;
    };
    auto load_row = [&](int64_t r) -> float4 {
        const float *p = S + r * ldS + c0;
        if (FULL) return *reinterpret_cast<const float4 *>(p);
        float4 v;
        v.x = ncol > 0 ? p[0] : 0.f; v.y = ncol > 1 ? p[1] : 0.f; v.z = ncol > 2 ? p[2] : 0.f; v.w = ncol > 3 ? p[3] : 0.f;
        return v;
    };
    // rows rc .. r_to - 1 (at most 64) are done: the four waves hold them over four column ranges -- combine through LDS, one atomic
    // pair per row; the next chunk starts at r_to
    auto flush = [&](int64_t r_to) {
        __syncthreads();                                                      // (the previous chunk's readers are done)
        s_rows[0][wave][lane] = rv_cnt; s_rows[1][wave][lane] = rv_ok; s_rows[2][wave][lane] = rv_col;
        __syncthreads();
        if (wave == 0 && rc + lane < r_to) {
            uint32_t c = 0;
            unsigned long long best = 0;
#pragma unroll
            for (int w = 0; w < RANK_THREADS / 64; ++w) {
                c += s_rows[0][w][lane];
                const unsigned long long k = key64(s_rows[1][w][lane], s_rows[2][w][lane]);
                best = k > best ? k : best;
            }
            if (c) atomicAdd(&i2t_cnt[rc + lane], (int)c);
            if (best) atomicMax(&row_best[rc + lane], best);
        }
        rc = r_to;
        rv_cnt = 0; rv_ok = 0; rv_col = 0;
    };
    // A ring of U rows in registers: row r + U is requested the moment row r has been consumed, so U - 1 rows are in flight
    // per lane at ALL times.  (v1 loaded a group of rows, then consumed it: every wave idled for a full memory round trip per group --
    // 29 % ALU issue, 50 % of the wave time in s_waitcnt.  Two alternating groups of four left loads in flight only about half
    // of the time: the rate of a streaming read follows the bytes in flight, tools/ubench/read_tiles.hip.)
    // Rows past the end are clamped to the last row (loaded again, never consumed).
    float4 ring[U];
#pragma unroll
    for (int k = 0; k < U; ++k) ring[k] = load_row(r_begin + k < r_end ? r_begin + k : r_end - 1);
    int64_t r = r_begin;
    for (; r + U <= r_end; r += U) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            row_step(ring[k], r + k);
            const int64_t rn = r + U + k;
            ring[k] = load_row(rn < r_end ? rn : r_end - 1);
        }
        if (((r + U - r_begin) & 63) == 0) flush(r + U);
    }
#pragma unroll
    for (int k = 0; k < U; ++k)                                            // the matrix's last rows (nrows % U): already in the ring
        if (r + k < r_end) row_step(ring[k], r + k);
    if (rc < r_end) flush(r_end);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (u < ncol) {
            if (cnt[u]) atomicAdd(&t2i_cnt[c0 + u], cnt[u]);
            atomicMax(&t2i_best[c0 + u], key64(b_ok[u], b_row[u]));
        }
}

__global__ __launch_bounds__(RANK_THREADS, ITR_RF_WAVES) void rank_fused_kernel(const float *__restrict__ S, int64_t ldS, int64_t row0, int64_t nrows,
                                                                  int64_t Nc, int im_div, const float *__restrict__ s_gt,
                                                                  const unsigned long long *__restrict__ row_gkey,
                                                                  int32_t *__restrict__ i2t_cnt, unsigned long long *__restrict__ row_best,
                                                                  int32_t *__restrict__ t2i_cnt, unsigned long long *__restrict__ t2i_best,
                                                                  int rows_per_wg) {
    __shared__ uint32_t s_rows[3][RANK_THREADS / 64][64];
    const int64_t cb = (int64_t)blockIdx.x * RF_COLS;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_wg;
    const int64_t r_end = (r_begin + rows_per_wg < nrows) ? r_begin + rows_per_wg : nrows;
    const bool full = (cb + RF_COLS <= Nc) && ((ldS & 3) == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0);
    // ground-truth pairs inside the tile?  rows R0 .. R1-1 have their GT columns in [im_div R0, im_div R1); columns cb .. cb+1023
    // have their GT rows in [cb / im_div, (cb + 1023) / im_div].  (Conservative: a tile flagged without need only runs the exact form.)
    const int64_t R0 = row0 + r_begin, R1 = row0 + r_end;
    const bool diag = (im_div * R0 <= cb + RF_COLS - 1) && (im_div * R1 - 1 >= cb);
// (the exact-compare and the guarded forms run on a few tiles in a hundred: a ring of 4 keeps them inside the register budget)
#define ITR_RANK_TILE(F, D) rank_tile<F, D, ((F) && !(D)) ? RF_U : 4>(S, ldS, row0, r_begin, r_end, Nc, im_div, s_gt, row_gkey, i2t_cnt, row_best, t2i_cnt, t2i_best, s_rows)
    if (full) { if (diag) ITR_RANK_TILE(true, true); else ITR_RANK_TILE(true, false); }
    else      { if (diag) ITR_RANK_TILE(false, true); else ITR_RANK_TILE(false, false); }
#undef ITR_RANK_TILE
}

// After the pass: the rows' accumulated top-1 keys -> column indices.  (A "last workgroup of the row block finishes" scheme inside the
// pass was built and measured: it needs a device-scope fence per workgroup, which on this multi-XCD part writes back and invalidates the
// XCD's L2 -- 122 -> 287 us.  A 5 us launch is the cheaper fence.)
__global__ void rank_rows_finish_kernel(int64_t row0, int64_t nrows, int64_t Nc, int im_div, const unsigned long long *__restrict__ row_best,
                                        int32_t *__restrict__ i2t_rank, int32_t *__restrict__ i2t_top1) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    i2t_top1[r] = (int32_t)(row_best[r] & 0xffffffffu);
    if ((row0 + r) * im_div >= Nc) i2t_rank[r] = 0x7fffffff;      // no ground-truth caption inside the matrix (as rounds 1-4)
}

// workgroups of rank_fused_kernel the device holds at once (CUs x the occupancy the runtime reports), cached per device
static int rank_resident_slots(int64_t *slots) {
    static std::mutex mu;
    static int64_t slots_of[64] = {};
    int dev = 0;
    ITR_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64) { *slots = 512; return ITR_OK; }
    if (!slots_of[dev]) {
        hipDeviceProp_t prop;
        int per_cu = 0;
        ITR_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        ITR_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rank_fused_kernel, RANK_THREADS, 0));
        slots_of[dev] = (int64_t)prop.multiProcessorCount * (per_cu > 0 ? per_cu : 1);
    }
    *slots = slots_of[dev];
    return ITR_OK;
}

// ---- float64 similarity matrices ------------------------------------------------------------
// The reference ranks the float64 matrix cal_sims returns (evaluation.py:169, :209), and the
// ensemble path averages two models in float64 first (evaluation.py:380, :398): scores that differ
// in float64 may collapse in fp32, so those matrices are counted in float64.  Same scheme as above
// with 16-byte double2 loads; the arg-max key (64-bit ordered score) no longer fits next to the
// index, so t2i's top-1 takes a second pass: max key per column first, then the highest row
// holding it.
constexpr int T2I_ROWS = 128;      // image rows per workgroup of the float64 column pass
// canon_f64: -0.0 -> +0.0 and NaN -> +inf (the fp32 kernels' rule, np.argsort's order); applied to every score the kernels load
__device__ __forceinline__ double canon_f64(double d) { return fmin(d + 0.0, (double)INFINITY); }
__device__ __forceinline__ unsigned long long double_order_key(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__global__ void gather_gt_f64_kernel(const double *__restrict__ S, int64_t ldS, int64_t row0, int64_t nrows,
                                     int64_t Nc, int im_div, double *__restrict__ s_gt) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Nc) return;
    const int64_t g = j / im_div - row0;
    if (g >= 0 && g < nrows) s_gt[j] = S[g * ldS + j];
}

__global__ __launch_bounds__(RANK_THREADS) void i2t_rank_f64_kernel(const double *__restrict__ S, int64_t ldS,
                                                                    int64_t row0, int64_t Nc, int im_div,
                                                                    int32_t *__restrict__ rank_out,
                                                                    int32_t *__restrict__ top1_out) {
    __shared__ int s_cnt[RANK_THREADS / 64][MAX_IMDIV];
    __shared__ unsigned long long s_key[RANK_THREADS / 64];
    __shared__ int s_idx[RANK_THREADS / 64];
    const int64_t r = blockIdx.x;
    const double *row = S + r * ldS;
    const int64_t gi = row0 + r;
    double gt[MAX_IMDIV];
    int64_t gidx[MAX_IMDIV];
    int cnt[MAX_IMDIV];
#pragma unroll
    for (int g = 0; g < MAX_IMDIV; ++g) {
        gidx[g] = gi * im_div + g;
        const bool ok = g < im_div && gidx[g] < Nc;
        gt[g] = ok ? canon_f64(row[gidx[g]]) : (double)INFINITY;
        cnt[g] = 0;
    }
    unsigned long long bkey = 0;
    int bidx = -1;  // columns are visited in increasing order per lane: >= keeps the highest index
    const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    const int64_t nvec = vec ? (Nc >> 1) : 0;
    for (int64_t c = threadIdx.x; c < nvec; c += RANK_THREADS) {
        const double2 v = reinterpret_cast<const double2 *>(row)[c];
        const double e[2] = {canon_f64(v.x), canon_f64(v.y)};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t k = c * 2 + u;
            const unsigned long long key = double_order_key(e[u]);
            if (key >= bkey) { bkey = key; bidx = (int)k; }
#pragma unroll
            for (int g = 0; g < MAX_IMDIV; ++g)
                cnt[g] += (e[u] > gt[g]) || (e[u] == gt[g] && k > gidx[g]);
        }
    }
    for (int64_t k = nvec * 2 + threadIdx.x; k < Nc; k += RANK_THREADS) {
        const double e = canon_f64(row[k]);
        const unsigned long long key = double_order_key(e);
        if (key >= bkey) { bkey = key; bidx = (int)k; }
#pragma unroll
        for (int g = 0; g < MAX_IMDIV; ++g) cnt[g] += (e > gt[g]) || (e == gt[g] && k > gidx[g]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < MAX_IMDIV; ++g) cnt[g] = wave_sum_i(cnt[g]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long okey = __shfl_xor(bkey, o, 64);
        const int oidx = __shfl_xor(bidx, o, 64);
        if (okey > bkey || (okey == bkey && oidx > bidx)) { bkey = okey; bidx = oidx; }
    }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < MAX_IMDIV; ++g) s_cnt[wave][g] = cnt[g];
        s_key[wave] = bkey;
        s_idx[wave] = bidx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int rank = 0x7fffffff;
        for (int g = 0; g < im_div; ++g) {
            if (gi * im_div + g >= Nc) break;
            int t = 0;
            for (int w = 0; w < RANK_THREADS / 64; ++w) t += s_cnt[w][g];
            rank = t < rank ? t : rank;
        }
        unsigned long long k = 0;
        int b = -1;
        for (int w = 0; w < RANK_THREADS / 64; ++w)
            if (s_key[w] > k || (s_key[w] == k && s_idx[w] > b)) { k = s_key[w]; b = s_idx[w]; }
        rank_out[r] = rank;
        top1_out[r] = b;
    }
}

// PASS 0: counts + max key per column; PASS 1: highest global row whose key equals best_key.
template <int PASS>
__global__ __launch_bounds__(RANK_THREADS) void t2i_rank_f64_kernel(const double *__restrict__ S, int64_t ldS,
                                                                    int64_t row0, int64_t nrows, int64_t Nc,
                                                                    int im_div, const double *__restrict__ s_gt,
                                                                    int32_t *__restrict__ rank_acc,
                                                                    unsigned long long *__restrict__ best_key,
                                                                    int32_t *__restrict__ best_row) {
    const int64_t c0 = ((int64_t)blockIdx.x * RANK_THREADS + threadIdx.x) * 2;
    if (c0 >= Nc) return;
    const int64_t r_begin = (int64_t)blockIdx.y * T2I_ROWS;
    const int64_t r_end = (r_begin + T2I_ROWS < nrows) ? r_begin + T2I_ROWS : nrows;
    const int ncol = (Nc - c0 >= 2) ? 2 : 1;
    double gt[2];
    int64_t gimg[2];
    int cnt[2] = {0, 0};
    unsigned long long best[2] = {0, 0};
    int brow[2] = {-1, -1};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        gt[u] = (PASS == 0 && u < ncol) ? canon_f64(s_gt[c0 + u]) : (double)INFINITY;
        gimg[u] = (c0 + u) / im_div;
        if (PASS == 1 && u < ncol) best[u] = best_key[c0 + u];
    }
    const bool vec = (ncol == 2) && ((ldS & 1) == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0);
    for (int64_t r = r_begin; r < r_end; ++r) {
        const double *p = S + r * ldS + c0;
        double e[2];
        if (vec) {
            const double2 v = *reinterpret_cast<const double2 *>(p);
            e[0] = v.x; e[1] = v.y;
        } else {
            e[0] = p[0];
            e[1] = ncol == 2 ? p[1] : -(double)INFINITY;
        }
        e[0] = canon_f64(e[0]); e[1] = canon_f64(e[1]);
        const int64_t gr = row0 + r;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned long long key = double_order_key(e[u]);
            if (PASS == 0) {
                cnt[u] += (e[u] > gt[u]) || (e[u] == gt[u] && gr > gimg[u]);
                best[u] = key > best[u] ? key : best[u];
            } else if (key == best[u]) {
                brow[u] = (int)gr;  // rows ascend: the last match is the highest
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (u < ncol) {
            if (PASS == 0) {
                if (cnt[u]) atomicAdd(&rank_acc[c0 + u], cnt[u]);
                atomicMax(&best_key[c0 + u], best[u]);
            } else if (brow[u] >= 0) {
                atomicMax(&best_row[c0 + u], brow[u]);
            }
        }
}

}  // namespace itr

extern "C" int itr_rank_gather_gt_f64(const double *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                                      int im_div, double *s_gt, itr_stream_t stream) {
    ITR_REQUIRE(S && s_gt, "itr_rank_gather_gt_f64: null pointer");
    ITR_REQUIRE(im_div >= 1 && Nc >= 0 && ldS >= Nc && row0 >= 0 && n_rows_local >= 0,
                "itr_rank_gather_gt_f64: bad shape");
    if (Nc == 0) return ITR_OK;
    hipLaunchKernelGGL(itr::gather_gt_f64_kernel, dim3((unsigned)itr::ceil_div(Nc, 256)), dim3(256), 0,
                       itr::as_stream(stream), S, ldS, row0, n_rows_local, Nc, im_div, s_gt);
    ITR_CHECK_LAUNCH("gather_gt_f64");
    return ITR_OK;
}

extern "C" int itr_rank_counts_f64(const double *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                                   int im_div, const double *s_gt, int32_t *i2t_rank, int32_t *i2t_top1,
                                   int32_t *t2i_rank, uint64_t *t2i_best_key, itr_stream_t stream) {
    ITR_REQUIRE(S && s_gt && i2t_rank && i2t_top1 && t2i_rank && t2i_best_key, "itr_rank_counts_f64: null pointer");
    ITR_REQUIRE(im_div >= 1 && im_div <= itr::MAX_IMDIV, "itr_rank_counts_f64: im_div must be in [1, %d]",
                itr::MAX_IMDIV);
    ITR_REQUIRE(Nc >= 0 && ldS >= Nc && row0 >= 0 && n_rows_local >= 0, "itr_rank_counts_f64: bad shape");
    ITR_REQUIRE(Nc < 0x7fffffffLL && row0 + n_rows_local < 0x7fffffffLL, "itr_rank_counts_f64: index overflow");
    if (Nc == 0 || n_rows_local == 0) return ITR_OK;
    hipStream_t st = itr::as_stream(stream);
    hipLaunchKernelGGL(itr::i2t_rank_f64_kernel, dim3((unsigned)n_rows_local), dim3(itr::RANK_THREADS), 0, st, S,
                       ldS, row0, Nc, im_div, i2t_rank, i2t_top1);
    ITR_CHECK_LAUNCH("i2t_rank_f64");
    dim3 grid((unsigned)itr::ceil_div(Nc, (int64_t)itr::RANK_THREADS * 2),
              (unsigned)itr::ceil_div(n_rows_local, itr::T2I_ROWS));
    hipLaunchKernelGGL(itr::t2i_rank_f64_kernel<0>, grid, dim3(itr::RANK_THREADS), 0, st, S, ldS, row0,
                       n_rows_local, Nc, im_div, s_gt, t2i_rank,
                       reinterpret_cast<unsigned long long *>(t2i_best_key), (int32_t *)nullptr);
    ITR_CHECK_LAUNCH("t2i_rank_f64");
    return ITR_OK;
}

extern "C" int itr_rank_t2i_top1_f64(const double *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                                     const uint64_t *t2i_best_key, int32_t *t2i_top1, itr_stream_t stream) {
    ITR_REQUIRE(S && t2i_best_key && t2i_top1, "itr_rank_t2i_top1_f64: null pointer");
    ITR_REQUIRE(Nc >= 0 && ldS >= Nc && row0 >= 0 && n_rows_local >= 0, "itr_rank_t2i_top1_f64: bad shape");
    ITR_REQUIRE(Nc < 0x7fffffffLL && row0 + n_rows_local < 0x7fffffffLL, "itr_rank_t2i_top1_f64: index overflow");
    if (Nc == 0 || n_rows_local == 0) return ITR_OK;
    dim3 grid((unsigned)itr::ceil_div(Nc, (int64_t)itr::RANK_THREADS * 2),
              (unsigned)itr::ceil_div(n_rows_local, itr::T2I_ROWS));
    hipLaunchKernelGGL(itr::t2i_rank_f64_kernel<1>, grid, dim3(itr::RANK_THREADS), 0, itr::as_stream(stream), S, ldS,
                       row0, n_rows_local, Nc, 1, (const double *)nullptr, (int32_t *)nullptr,
                       reinterpret_cast<unsigned long long *>(const_cast<uint64_t *>(t2i_best_key)), t2i_top1);
    ITR_CHECK_LAUNCH("t2i_top1_f64");
    return ITR_OK;
}

extern "C" int itr_rank_gather_gt(const float *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                                  int im_div, float *s_gt, itr_stream_t stream) {
    ITR_REQUIRE(S && s_gt, "itr_rank_gather_gt: null pointer");
    ITR_REQUIRE(im_div >= 1 && Nc >= 0 && ldS >= Nc && row0 >= 0 && n_rows_local >= 0,
                "itr_rank_gather_gt: bad shape");
    if (Nc == 0) return ITR_OK;
    hipLaunchKernelGGL(itr::gather_gt_kernel, dim3((unsigned)itr::ceil_div(Nc, 256)), dim3(256), 0,
                       itr::as_stream(stream), S, ldS, row0, n_rows_local, Nc, im_div, s_gt);
    ITR_CHECK_LAUNCH("gather_gt");
    return ITR_OK;
}

static inline size_t rank_al(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" size_t itr_rank_workspace_bytes(int64_t n_rows_local, int64_t Nc) {
    const size_t nr = (size_t)(n_rows_local > 0 ? n_rows_local : 0), nc = (size_t)(Nc > 0 ? Nc : 0);
    // per local row: its best GT key + its running top-1 key; per column: the GT score when the call gathers it itself
    return rank_al(nr * 16) + rank_al(nc * 4) + 256;
}

extern "C" int itr_rank_counts(const float *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                               int im_div, const float *s_gt, int32_t *i2t_rank, int32_t *i2t_top1,
                               int32_t *t2i_rank, uint64_t *t2i_best, int flags, void *workspace, size_t workspace_bytes,
                               itr_stream_t stream) {
    ITR_REQUIRE(S && i2t_rank && i2t_top1 && t2i_rank && t2i_best, "itr_rank_counts: null pointer");
    ITR_REQUIRE(im_div >= 1 && im_div <= itr::MAX_IMDIV, "itr_rank_counts: im_div must be in [1, %d]",
                itr::MAX_IMDIV);
    ITR_REQUIRE(Nc >= 0 && ldS >= Nc && row0 >= 0 && n_rows_local >= 0, "itr_rank_counts: bad shape");
    ITR_REQUIRE(Nc < 0x7fffffffLL && row0 + n_rows_local < 0x7fffffffLL, "itr_rank_counts: index overflow");
    ITR_REQUIRE((flags & ~ITR_RANK_INIT_COLUMNS) == 0, "itr_rank_counts: unknown flag bits %d", flags);
    // s_gt == NULL: the ground-truth scores are read from S itself -- every ground-truth row must be local
    ITR_REQUIRE(s_gt || (row0 == 0 && n_rows_local * im_div >= Nc),
                "itr_rank_counts: s_gt may only be NULL when the block holds every ground-truth row (row0 = 0, n_rows_local * im_div >= Nc)");
    if (Nc == 0 || n_rows_local == 0) return ITR_OK;
    ITR_REQUIRE(workspace && workspace_bytes >= itr_rank_workspace_bytes(n_rows_local, Nc) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                "itr_rank_counts: workspace missing, misaligned or smaller than itr_rank_workspace_bytes");
    hipStream_t st = itr::as_stream(stream);
    char *wp = static_cast<char *>(workspace);
    unsigned long long *row_gkey = reinterpret_cast<unsigned long long *>(wp), *row_best = row_gkey + n_rows_local;
    wp += rank_al((size_t)n_rows_local * 16);
    float *gt_own = reinterpret_cast<float *>(wp);
    // rows per workgroup: the smallest multiple of 64 for which the grid fits the chip's resident workgroups in one round
    int64_t slots = 0;
    {
        const int rc_ = itr::rank_resident_slots(&slots);
        if (rc_ != ITR_OK) return rc_;
    }
    const int64_t ncb = itr::ceil_div(Nc, (int64_t)itr::RF_COLS);
    int64_t rpw = itr::ceil_div(itr::ceil_div(n_rows_local * ncb, slots), (int64_t)64) * 64;
    if (rpw < 64) rpw = 64;
    if (rpw > 1024) rpw = 1024;
    const int64_t gy = itr::ceil_div(n_rows_local, rpw);
    ITR_REQUIRE(gy <= 65535, "itr_rank_counts: too many local rows per call");
    const int64_t n_prep = n_rows_local > Nc ? n_rows_local : Nc;
    hipLaunchKernelGGL(itr::rank_prepare_kernel, dim3((unsigned)itr::ceil_div(n_prep, (int64_t)256)), dim3(256), 0, st, S, ldS, row0, n_rows_local,
                       Nc, im_div, row_gkey, row_best, i2t_rank, s_gt ? (float *)nullptr : gt_own, t2i_rank,
                       reinterpret_cast<unsigned long long *>(t2i_best), (flags & ITR_RANK_INIT_COLUMNS) ? 1 : 0);
    ITR_CHECK_LAUNCH("rank_prepare");
    dim3 grid((unsigned)ncb, (unsigned)gy);
    hipLaunchKernelGGL(itr::rank_fused_kernel, grid, dim3(itr::RANK_THREADS), 0, st, S, ldS, row0, n_rows_local, Nc, im_div, s_gt ? s_gt : gt_own,
                       row_gkey, i2t_rank, row_best, t2i_rank, reinterpret_cast<unsigned long long *>(t2i_best), (int)rpw);
    ITR_CHECK_LAUNCH("rank_fused");
    hipLaunchKernelGGL(itr::rank_rows_finish_kernel, dim3((unsigned)itr::ceil_div(n_rows_local, (int64_t)256)), dim3(256), 0, st, row0,
                       n_rows_local, Nc, im_div, row_best, i2t_rank, i2t_top1);
    ITR_CHECK_LAUNCH("rank_rows_finish");
    return ITR_OK;
}

extern "C" int itr_recall_from_ranks(const int32_t *ranks_host, int64_t n, double *out5) {
    ITR_REQUIRE(ranks_host && out5 && n > 0, "itr_recall_from_ranks: bad argument");
    // evaluation.py:181-185: R@K = 100 * #{rank < K} / n ; medr = floor(median) + 1 ; meanr = mean + 1
    int64_t c1 = 0, c5 = 0, c10 = 0;
    double sum = 0;
    int32_t *tmp = new int32_t[n];
    for (int64_t i = 0; i < n; ++i) {
        const int32_t r = ranks_host[i];
        c1 += r < 1; c5 += r < 5; c10 += r < 10;
        sum += r;
        tmp[i] = r;
    }
    // median via counting-free selection (n is small: <= number of queries)
    auto nth = [&](int64_t k) {
        int64_t lo = 0, hi = n - 1;
        while (lo < hi) {
            const int32_t pivot = tmp[(lo + hi) / 2];
            int64_t i = lo, j = hi;
            while (i <= j) {
                while (tmp[i] < pivot) ++i;
                while (tmp[j] > pivot) --j;
                if (i <= j) { const int32_t t = tmp[i]; tmp[i] = tmp[j]; tmp[j] = t; ++i; --j; }
            }
            if (k <= j) hi = j; else if (k >= i) lo = i; else break;
        }
        return (double)tmp[k];
    };
    double med;
    if (n & 1) med = nth(n / 2);
    else { const double a = nth(n / 2 - 1); const double b = nth(n / 2); med = 0.5 * (a + b); }
    delete[] tmp;
    out5[0] = 100.0 * c1 / n; out5[1] = 100.0 * c5 / n; out5[2] = 100.0 * c10 / n;
    out5[3] = (double)(int64_t)med + 1.0;  // floor (ranks are non-negative)
    out5[4] = sum / n + 1.0;
    return ITR_OK;
}
