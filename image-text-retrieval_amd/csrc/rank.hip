// Recall@K ranker (itr/metricmodule/evaluation.py:156-222) without sorting.
//
// The reference argsorts every row (i2t) and every column (t2i) of the float64 similarity
// matrix on the host.  The rank of a ground-truth item is just a count:
//     rank(q, gt) = #{k : S[q,k] > S[q,gt]} + #{k > gt : S[q,k] == S[q,gt]}
// (with np.argsort(...)[::-1] the higher index wins exact ties), so one streaming
// pass over S per direction is enough: integer compares, HBM-bound (4 bytes/pair/direction).
//
//   i2t : one workgroup per image row; 5 GT captions -> 5 counters per lane, float4 loads,
//         wave + LDS reduction, min over the 5.
//   t2i : one lane owns 4 consecutive caption columns (float4, coalesced across the wave) and
//         walks down a chunk of image rows; partial counts are added with one atomic per column
//         so row blocks (other workgroups, or other GPUs after an all-reduce) just sum.
// S may be a row block of the global matrix (multi-GPU row sharding): row0 is the global index
// of its first row, and s_gt[] carries the GT score of every caption (gathered over ranks).
#include "itr_common.h"

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
// e + 0.0f folds -0.0 into +0.0 (they compare equal as floats, their bit patterns do not).  Round 4: the counts were five float
// compare chains with 64-bit index compares per element (~50 vector instructions per element: the row pass was bound by the vector
// ALU at 2.6 TB/s, not by HBM); one v_cmp_gt_u64 + one add per GT caption now, and only im_div of them (template G).
__device__ __forceinline__ unsigned long long rank_key(float e, unsigned idx) {
    return ((unsigned long long)float_order_key(e + 0.0f) << 32) | idx;
}

template <int G>
__global__ __launch_bounds__(RANK_THREADS) void i2t_rank_kernel(const float *__restrict__ S, int64_t ldS,
                                                                int64_t row0, int64_t Nc, int im_div,
                                                                int32_t *__restrict__ rank_out,
                                                                int32_t *__restrict__ top1_out) {
    __shared__ int s_cnt[RANK_THREADS / 64][MAX_IMDIV];
    __shared__ unsigned long long s_best[RANK_THREADS / 64];
    const int64_t r = blockIdx.x;
    const float *row = S + r * ldS;
    const int64_t gi = row0 + r;  // global image index
    unsigned long long gkey[G];
    int cnt[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t gidx = gi * im_div + g;
        // a GT caption past the end of the matrix: nothing is larger than its key (its count is not read)
        gkey[g] = gidx < Nc ? rank_key(row[gidx], (unsigned)gidx) : ~0ull;
        cnt[g] = 0;
    }
    unsigned long long best = 0;  // max => highest index on ties
    const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    const int64_t nvec = vec ? (Nc >> 2) : 0;
    for (int64_t c = threadIdx.x; c < nvec; c += RANK_THREADS) {
        const float4 v = reinterpret_cast<const float4 *>(row)[c];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long key = rank_key(e[u], (unsigned)(c * 4 + u));
            best = key > best ? key : best;
#pragma unroll
            for (int g = 0; g < G; ++g) cnt[g] += key > gkey[g];
        }
    }
    for (int64_t k = nvec * 4 + threadIdx.x; k < Nc; k += RANK_THREADS) {
        const unsigned long long key = rank_key(row[k], (unsigned)k);
        best = key > best ? key : best;
#pragma unroll
        for (int g = 0; g < G; ++g) cnt[g] += key > gkey[g];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < G; ++g) cnt[g] = wave_sum_i(cnt[g]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) s_cnt[wave][g] = cnt[g];
        s_best[wave] = best;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int rank = 0x7fffffff;
        for (int g = 0; g < G; ++g) {
            if (gi * im_div + g >= Nc) break;
            int t = 0;
            for (int w = 0; w < RANK_THREADS / 64; ++w) t += s_cnt[w][g];
            rank = t < rank ? t : rank;
        }
        unsigned long long b = 0;
        for (int w = 0; w < RANK_THREADS / 64; ++w) b = s_best[w] > b ? s_best[w] : b;
        rank_out[r] = rank;
        top1_out[r] = (int32_t)(b & 0xffffffffu);
    }
}

#ifndef ITR_T2I_ROWS
#define ITR_T2I_ROWS 128
#endif
#ifndef ITR_T2I_QUADS
#define ITR_T2I_QUADS 1
#endif
constexpr int T2I_ROWS = ITR_T2I_ROWS;     // image rows per workgroup (one pair of atomics per column and workgroup: fewer rows = more atomics)
constexpr int T2I_QUADS = ITR_T2I_QUADS;   // float4 column groups per lane, 1 024 columns apart (a workgroup reads QUADS x 4 KB contiguous per row)

// Column pass: a workgroup covers QUADS x 1 024 consecutive columns x ROWS rows (lane t owns columns 1024 q + 4 t .. + 3 of the block).
// Round 4 sweep on one box (5 000 x 25 000, us per launch; tools/ab_build.sh -DITR_T2I_QUADS / -DITR_T2I_ROWS): QUADS x ROWS =
// 1 x 64: 127.0 (rounds 1-3), 1 x 128: 121.8 (now), 1 x 256: 176.9, 2 x 64: 155.7, 2 x 128: 188.0, 4 x 64: 220.3, 4 x 16: 331 --
// wider contiguous reads per row LOSE (fewer workgroups in flight), fewer rows per workgroup lose to the atomics (one pair per
// column and workgroup, executed at the memory side).  ~4 TB/s = half of the HBM peak is where this layout ends.
__global__ __launch_bounds__(RANK_THREADS) void t2i_rank_kernel(const float *__restrict__ S, int64_t ldS,
                                                                int64_t row0, int64_t nrows, int64_t Nc,
                                                                int im_div, const float *__restrict__ s_gt,
                                                                int32_t *__restrict__ rank_acc,
                                                                unsigned long long *__restrict__ best_acc) {
    const int64_t cb = (int64_t)blockIdx.x * (RANK_THREADS * 4 * T2I_QUADS) + threadIdx.x * 4;
    const int64_t r_begin = (int64_t)blockIdx.y * T2I_ROWS;
    const int64_t r_end = (r_begin + T2I_ROWS < nrows) ? r_begin + T2I_ROWS : nrows;
    unsigned long long gkey[T2I_QUADS][4], best[T2I_QUADS][4];
    int cnt[T2I_QUADS][4];
    int ncol[T2I_QUADS];
#pragma unroll
    for (int q = 0; q < T2I_QUADS; ++q) {
        const int64_t c0 = cb + (int64_t)q * RANK_THREADS * 4;
        ncol[q] = c0 >= Nc ? 0 : ((Nc - c0 >= 4) ? 4 : (int)(Nc - c0));
#pragma unroll
        for (int u = 0; u < 4; ++u) {      // (score of the GT image, its row): a row counts when its key is larger (rank_key: the tie rule)
            gkey[q][u] = u < ncol[q] ? rank_key(s_gt[c0 + u], (unsigned)((c0 + u) / im_div)) : ~0ull;
            best[q][u] = 0;
            cnt[q][u] = 0;
        }
    }
    const bool vec = ((ldS & 3) == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0);
    for (int64_t r = r_begin; r < r_end; ++r) {
        const float *p = S + r * ldS + cb;
        float e[T2I_QUADS][4];
#pragma unroll
        for (int q = 0; q < T2I_QUADS; ++q) {
            const float *pq = p + q * RANK_THREADS * 4;
            if (vec && ncol[q] == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(pq);
                e[q][0] = v.x; e[q][1] = v.y; e[q][2] = v.z; e[q][3] = v.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) e[q][u] = u < ncol[q] ? pq[u] : -INFINITY;
            }
        }
        const unsigned gr = (unsigned)(row0 + r);
#pragma unroll
        for (int q = 0; q < T2I_QUADS; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long key = rank_key(e[q][u], gr);
                cnt[q][u] += key > gkey[q][u];
                best[q][u] = key > best[q][u] ? key : best[q][u];
            }
    }
#pragma unroll
    for (int q = 0; q < T2I_QUADS; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < ncol[q]) {
                const int64_t c = cb + (int64_t)q * RANK_THREADS * 4 + u;
                if (cnt[q][u]) atomicAdd(&rank_acc[c], cnt[q][u]);
                atomicMax(&best_acc[c], best[q][u]);
            }
}

// ---- float64 similarity matrices ------------------------------------------------------------
// The reference ranks the float64 matrix cal_sims returns (evaluation.py:169, :209), and the
// ensemble path averages two models in float64 first (evaluation.py:380, :398): scores that differ
// in float64 may collapse in fp32, so those matrices are counted in float64.  Same scheme as above
// with 16-byte double2 loads; the arg-max key (64-bit ordered score) no longer fits next to the
// index, so t2i's top-1 takes a second pass: max key per column first, then the highest row
// holding it.
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
        gt[g] = ok ? row[gidx[g]] : (double)INFINITY;
        cnt[g] = 0;
    }
    unsigned long long bkey = 0;
    int bidx = -1;  // columns are visited in increasing order per lane: >= keeps the highest index
    const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    const int64_t nvec = vec ? (Nc >> 1) : 0;
    for (int64_t c = threadIdx.x; c < nvec; c += RANK_THREADS) {
        const double2 v = reinterpret_cast<const double2 *>(row)[c];
        const double e[2] = {v.x, v.y};
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
        const double e = row[k];
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
        gt[u] = (PASS == 0 && u < ncol) ? s_gt[c0 + u] : (double)INFINITY;
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

extern "C" int itr_rank_counts(const float *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                               int im_div, const float *s_gt, int32_t *i2t_rank, int32_t *i2t_top1,
                               int32_t *t2i_rank, uint64_t *t2i_best, itr_stream_t stream) {
    ITR_REQUIRE(S && s_gt && i2t_rank && i2t_top1 && t2i_rank && t2i_best, "itr_rank_counts: null pointer");
    ITR_REQUIRE(im_div >= 1 && im_div <= itr::MAX_IMDIV, "itr_rank_counts: im_div must be in [1, %d]",
                itr::MAX_IMDIV);
    ITR_REQUIRE(Nc >= 0 && ldS >= Nc && row0 >= 0 && n_rows_local >= 0, "itr_rank_counts: bad shape");
    ITR_REQUIRE(Nc < 0x7fffffffLL && row0 + n_rows_local < 0x7fffffffLL, "itr_rank_counts: index overflow");
    if (Nc == 0 || n_rows_local == 0) return ITR_OK;
    hipStream_t st = itr::as_stream(stream);
    switch (im_div) {      // the row pass keeps one counter per GT caption of the image: exactly im_div of them
#define ITR_I2T(G) case G: hipLaunchKernelGGL(itr::i2t_rank_kernel<G>, dim3((unsigned)n_rows_local), dim3(itr::RANK_THREADS), 0, st, S, \
                                             ldS, row0, Nc, im_div, i2t_rank, i2t_top1); break;
        ITR_I2T(1) ITR_I2T(2) ITR_I2T(3) ITR_I2T(4) ITR_I2T(5) ITR_I2T(6) ITR_I2T(7) ITR_I2T(8)
#undef ITR_I2T
    }
    ITR_CHECK_LAUNCH("i2t_rank");
    dim3 grid((unsigned)itr::ceil_div(Nc, (int64_t)itr::RANK_THREADS * 4 * itr::T2I_QUADS),
              (unsigned)itr::ceil_div(n_rows_local, itr::T2I_ROWS));
    hipLaunchKernelGGL(itr::t2i_rank_kernel, grid, dim3(itr::RANK_THREADS), 0, st, S, ldS, row0, n_rows_local,
                       Nc, im_div, s_gt, t2i_rank, reinterpret_cast<unsigned long long *>(t2i_best));
    ITR_CHECK_LAUNCH("t2i_rank");
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
