// fp32 NT GEMM for a SKINNY left operand (M <= 128 rows: the GRU recurrence of a training batch, h [n_act, D] . W_hh^T, and its
// transpose in BPTT -- TextEncoder.py:38-56 under autograd; 80 of them per train_emb step of a bi-GRU model), as raw split-K partials.
//
// The 128 x 128 tile kernel sees ONE row tile here: 24 column tiles x 8 K slices of 128 columns each -- a workgroup's life is its
// prologue and epilogue, 31 us per product against a matrix-core floor of 7 (0.8 GFLOP + tile padding at 157 TFLOP/s).  This kernel
// turns the shape around: a workgroup owns 16 OUTPUT COLUMNS for all rows and one K slice; the rows' 64-wide K chunks stream through LDS
// (double buffered, the next chunk's global loads in flight under the MFMAs of the current one), each wave multiplies two 16-row tiles
// against the one 16-column tile on v_mfma_f32_16x16x4_f32 (operands [row][k] at row stride 68 floats: lane = 16 k + i hits 64 distinct
// banks).  Grid = (N / 16) x slices, slices chosen for ~1.5 workgroups per CU; the slices' partial products are written one after the other
// ([slice][m][n]) and summed -- in slice order, with the bias -- by the GRU gate kernels that consume them, exactly as the tile kernel's
// slices were.  Row tiles past the active prefix are skipped by whole waves.
#include "itr_common.h"

namespace itr {

constexpr int SK_KC = 64, SK_MAXM = 128;

// DIRECT: one slice (the whole K range), the epilogue adds the bias, applies the activation and writes C (leading dimension ldc) -- the
// per-step products of a decoder / a 128-row training batch that have no consumer kernel to sum slices for them.
// NC = 16-column tiles per workgroup, KC = k per LDS chunk.  <1, 64> is the original shape.  <2, 32> (round 6) halves the times the 128
// rows are pulled from L2 -- the kernel's bound: a 128 x 6 144 x 2 048 product moved 442 MB through 384 strips for 50 MB of weights,
// 57 us at 7.8 TB/s -- and reuses every LDS fragment twice; 46 KB of LDS, three workgroups per CU.  The k order inside a slice (and so
// a slice's bits) is the same in both.
template <bool DIRECT, int NC, int KC>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb, int M,
                                                          int64_t N, int K, int kslice, float *__restrict__ part, const float *__restrict__ bias,
                                                          int act, int64_t ldc) {
    constexpr int LD = KC + 4, NA = KC / 8, NB = NC * KC / 64, C4 = KC / 4;      // float4 loads per thread and chunk: rows, weights
    static_assert(NB >= 1 && NA >= 1, "chunk too small for 256 threads");
    __shared__ __attribute__((aligned(16))) float As[2][SK_MAXM][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][16 * NC][LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 16 * NC;
    const int k_begin = blockIdx.y * kslice;
    const int k_end = k_begin + kslice < K ? k_begin + kslice : K;
    const int rtiles = (M + 15) >> 4;
    f32x4 acc[2][NC];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[NA], rb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int idx = tid + 256 * j, row = idx / C4, c4 = (idx % C4) * 4;
            ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < M && k0 + c4 < k_end) ra[j] = *reinterpret_cast<const float4 *>(A + (int64_t)row * lda + k0 + c4);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = tid + 256 * j, row = idx / C4, c4 = (idx % C4) * 4;
            rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + row < N && k0 + c4 < k_end) rb[j] = *reinterpret_cast<const float4 *>(B + (n0 + row) * ldb + k0 + c4);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int idx = tid + 256 * j;
            *reinterpret_cast<float4 *>(&As[buf][idx / C4][(idx % C4) * 4]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = tid + 256 * j;
            *reinterpret_cast<float4 *>(&Bs[buf][idx / C4][(idx % C4) * 4]) = rb[j];
        }
    };
    if (k_begin < k_end) {
        fetch(k_begin);
        park(0);
        __syncthreads();
        int buf = 0;
        const bool t0 = wave < rtiles, t1 = wave + 4 < rtiles;
        for (int k0 = k_begin; k0 < k_end; k0 += KC) {
            const bool more = k0 + KC < k_end;
            if (more) fetch(k0 + KC);
            // (the two shapes of the loop are separate bodies: a branch inside the unrolled loop ends the scheduler's region at every k-step
            // and serialises LDS read -> wait -> MFMA: 21.9 -> 30.7 us on the 32-workgroup decoder products when it was tried)
            if (t1) {
#pragma unroll
                for (int kk = 0; kk < KC / 4; ++kk) {
                    const int k = kk * 4 + (lane >> 4);
                    float bf[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) bf[c] = Bs[buf][16 * c + (lane & 15)][k];
                    const float a0 = As[buf][wave * 16 + (lane & 15)][k], a1 = As[buf][(wave + 4) * 16 + (lane & 15)][k];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bf[c], acc[0][c], 0, 0, 0);
                        acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bf[c], acc[1][c], 0, 0, 0);
                    }
                }
            } else if (t0) {
#pragma unroll
                for (int kk = 0; kk < KC / 4; ++kk) {
                    const int k = kk * 4 + (lane >> 4);
                    const float a0 = As[buf][wave * 16 + (lane & 15)][k];
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, Bs[buf][16 * c + (lane & 15)][k], acc[0][c], 0, 0, 0);
                }
            }
            if (more) park(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    float *o = DIRECT ? part : part + (int64_t)blockIdx.y * M * N;
    const int64_t ldo = DIRECT ? ldc : N;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int64_t n = n0 + 16 * c + (lane & 15);
        if (n < N) {
            const float bv = (DIRECT && bias) ? bias[n] : 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rt = wave + 4 * t;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = rt * 16 + 4 * (lane >> 4) + q;
                    if (m < M) o[(int64_t)m * ldo + n] = DIRECT ? apply_act(acc[t][c][q] + bv, act) : acc[t][c][q];
                }
            }
        }
    }
}

bool gemm_skinny_ok(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    return M >= 1 && M <= SK_MAXM && N >= 16 && K >= SK_KC && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && K <= 0x7fffffff &&
           (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
}

// partial products part[slice][m][n] (row-major, leading dimension N) for *n_slices <= max_slices slices of the K range
int gemm_skinny_partials(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t M, int64_t N, int64_t K, int max_slices, float *part,
                         int *n_slices, hipStream_t st) {
    // 32-column strips from 1 024 columns up (the recurrence products: 3 D and D columns of a D >= 1 024 GRU); 16-column strips keep the
    // narrow ones spread over the chip
    // 64-column strips for a D = 2 048 GRU (6 144 gate columns over K = 2 048, 2 048 back over K = 6 144: 45 -> 39 us, 42 -> 38); at D = 1 024
    // (3 072 x 1 024) they measured slower than 32-column ones (SCAN's forward 1.26 -> 1.33 ms)
    const int nc = (N >= 4096 || (N >= 2048 && K >= 4096)) ? 4 : (N >= 1024 ? 2 : 1);
    const int64_t col_tiles = ceil_div(N, (int64_t)(16 * nc));
    int64_t s = ceil_div((int64_t)(nc > 1 ? 512 : 384), col_tiles); // ~1.5 workgroups per CU (2 of the 46 / 55 KB ones)
    const int64_t by_k = K / (2 * SK_KC) > 0 ? K / (2 * SK_KC) : 1; // a slice is at least 128 k
    if (s > by_k) s = by_k;
    if (s > max_slices) s = max_slices;
    if (s < 1) s = 1;
    const int kslice = (int)(ceil_div(ceil_div(K, s), (int64_t)SK_KC) * SK_KC);
    const int ns = (int)ceil_div(K, (int64_t)kslice);
    const dim3 grid((unsigned)col_tiles, (unsigned)ns);
    if (nc == 4)
        hipLaunchKernelGGL((gemm_skinny_kernel<false, 4, 32>), grid, dim3(256), 0, st, A, lda, B, ldb, (int)M, N, (int)K, kslice, part, (const float *)nullptr, 0,
                           (int64_t)0);
    else if (nc == 2)
        hipLaunchKernelGGL((gemm_skinny_kernel<false, 2, 32>), grid, dim3(256), 0, st, A, lda, B, ldb, (int)M, N, (int)K, kslice, part, (const float *)nullptr, 0,
                           (int64_t)0);
    else
        hipLaunchKernelGGL((gemm_skinny_kernel<false, 1, SK_KC>), grid, dim3(256), 0, st, A, lda, B, ldb, (int)M, N, (int)K, kslice, part, (const float *)nullptr,
                           0, (int64_t)0);
    ITR_CHECK_LAUNCH("gemm_skinny");
    *n_slices = ns;
    return ITR_OK;
}

// C = act(A B^T + bias) for M <= 128 in one launch (no K slices, no scratch).  A product with few strips (a 512-wide decoder layer: 32
// workgroups living ~1 us per 64 k) is better served by itr_gemm_nt_splitk, which runs the slices above and one slice sum (21 -> 10 us).
// Tried on this kernel in round 6 and not kept: 32-column strips here (half the workgroups: 21 -> 27 us); three register stages of
// branch-free asm loads with counted waits (no change: the chunk time is not the L2 round trip; plain branch-free loads are sunk to
// their LDS write by hipcc and ran 2.5 x slower).
int gemm_skinny_direct(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                       int act, hipStream_t st) {
    const int kslice = (int)(ceil_div(K, (int64_t)SK_KC) * SK_KC);
    hipLaunchKernelGGL((gemm_skinny_kernel<true, 1, SK_KC>), dim3((unsigned)ceil_div(N, (int64_t)16), 1u), dim3(256), 0, st, A, lda, B, ldb, (int)M, N, (int)K, kslice, C,
                       bias, act, ldc);
    ITR_CHECK_LAUNCH("gemm_skinny(direct)");
    return ITR_OK;
}

}  // namespace itr
