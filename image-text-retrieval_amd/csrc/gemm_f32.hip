// fp32 "NT" GEMM on the gfx950 matrix cores:  C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]).
//
// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-equal to an fmaf chain)
// because parity with the reference's fp32 CPU path is the first gate (no xf32/TF32 on gfx950).
//
// Tile 128 x 128 x 32, 256 threads = 4 waves (2 x 2), each wave owns 64 x 64 = 2 x 2 MFMA
// tiles of 32 x 32.  Both operands are K-contiguous, so a K-chunk of a tile is staged in LDS as
// 8 "planes" of float4 (plane p = k/4):  lds[p][row ^ p] (float4 units).
//   * global -> LDS: 8 consecutive lanes read one row's 128 contiguous bytes (coalesced) and
//     write 8 different planes; the `row ^ p` swizzle puts the 8 lanes of a ds_write_b128 group
//     on 8 different 16-byte slots (conflict-free);
//   * LDS -> MFMA fragment: lane (i = l & 31, g = l >> 5) reads ONE float4 from plane 2q + g and
//     feeds it to 4 consecutive MFMA k-steps (A and B use the same k mapping); every
//     ds_read_b128 lane group sees 16 distinct rows mod 16 -> conflict-free.
// Double-buffered LDS, next chunk prefetched to registers while the current one is multiplied.
//
// `group` > 1 fuses a max over `group` consecutive rows into the epilogue
// (MultiViewMatching, Fusionmodule.py:674-692: S[i,c] = max_v img[i,v,:] . cap[c,:]); the M tile
// then covers (128 / group) * group rows so no group straddles two tiles.
#include "itr_common.h"

namespace itr {

constexpr int BM = 128, BN = 128, BK = 32, NPLANE = BK / 4;
constexpr int GEMM_THREADS = 256;

struct GemmArgs {
    const float *A, *B, *bias;
    float *C;
    int64_t lda, ldb, ldc;
    int64_t M, N, K;
    int act;
    int group;       // 1 = plain GEMM; >1 = max over `group` consecutive rows of A*B^T
    int rows_per_tile;  // BM, or (BM / group) * group
    // optional "squared difference" epilogue (SGRAF, Fusionmodule.py:426-427: (Context_img - cap_i)^2 with the
    // l2norm of the context folded in as a per-row scale):  C = (acc * rowscale[m] - Z[m, n])^2
    const float *rowscale;
    const float *Z;
    int64_t ldz;
    int accumulate;   // 1: C = act(C + A*B^T + bias): sums the taps of a dilated Conv1d (camera_.py:100-103); 2: C = act(Z + A*B^T + bias):
                      //    a residual connection without first copying the residual into C (Rs_GCN, vsrn_.py:64-67)
    int64_t ksplit;   // > 0: blockIdx.y owns K range [y * ksplit, (y + 1) * ksplit) and writes the raw partial product to
    float *part;      //      part + y * M * N  (row-major [M, N]); bias / activation are applied by the reduction kernel
    // optional SECOND problem of the same shape in the same launch (fast kernel only: the two directions of a bi-GRU time step,
    // towers.hip): tiles [ntile, 2 ntile) of the persistent tile order read A2 / B2 / bias2 and write C2
    const float *A2, *B2, *bias2;
    float *C2;
};

template <bool ALIGNED>
__device__ __forceinline__ float4 load4(const float *base, int64_t row, int64_t nrows, int64_t ld,
                                        int64_t k, int64_t K) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
        const float *p = base + row * ld + k;
        if (ALIGNED) {
            if (k < K) v = *reinterpret_cast<const float4 *>(p);
        } else {
            if (k + 0 < K) v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
            if (k + 3 < K) v.w = p[3];
        }
    }
    return v;
}

// Shared epilogue: bias / activation / accumulate / squared-difference / max over row groups.
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, f32x16 (&acc)[2][2], float4 (*lds_raw), int tid, int lane, int wm, int wn,
                                              int64_t m0, int64_t n0, int64_t m_end) {
    float4(*lds)[2][NPLANE][BM] = reinterpret_cast<float4(*)[2][NPLANE][BM]>(lds_raw);
    // ---- epilogue ---------------------------------------------------------------------
    if (g.group <= 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t col = n0 + wn * 64 + j * 32 + (lane & 31);
                if (col >= g.N) continue;
                const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < m_end) {
                        float v = acc[i][j][r];
                        if (g.Z && g.accumulate != 2) {
                            v = v * g.rowscale[row] - g.Z[row * g.ldz + col];
                            v = v * v;
                        } else {
                            if (g.accumulate == 1) v += g.C[row * g.ldc + col];
                            else if (g.accumulate == 2) v += g.Z[row * g.ldz + col];        // residual read from its own matrix
                            v = apply_act(v + bv, g.act);
                        }
                        g.C[row * g.ldc + col] = v;
                    }
                }
            }
    } else {
        // max over `group` consecutive rows: stage the 128 x 128 tile in LDS (reusing the
        // operand buffers: 64 KB needed, 64 KB available), then one thread per (group, col).
        __syncthreads();
        float *tile = reinterpret_cast<float *>(&lds[0][0][0][0]);  // [128][128]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    tile[row * BN + col] = acc[i][j][r];
                }
            }
        __syncthreads();
        const int ngroups = g.rows_per_tile / g.group;
        for (int idx = tid; idx < ngroups * BN; idx += GEMM_THREADS) {
            const int gi = idx / BN, col = idx % BN;
            const int64_t grow = m0 / g.group + gi;
            if (m0 + (int64_t)gi * g.group >= g.M || n0 + col >= g.N) continue;
            float mx = -INFINITY;
            for (int v = 0; v < g.group; ++v) mx = fmaxf(mx, tile[(gi * g.group + v) * BN + col]);
            g.C[grow * g.ldc + n0 + col] = mx;
        }
    }
}

// Fast path: K % 32 == 0, 16-byte aligned rows.  Same tiling; the K loop is branch-free and free of vector-ALU work
// (on gfx950 the fp32 MFMA shares the vector ALU, so address arithmetic in the loop costs matrix throughput):
//   * global loads: uniform 64-bit base (advanced by a scalar add per chunk) + fixed per-lane 32-bit byte offset,
//     issued through inline asm with our own s_waitcnt; rows past the edge re-read the last valid row (their outputs
//     are never stored), the tail iterations re-load the last chunk and park data nobody reads;
//   * chunk kc+1 (loaded one iteration earlier) is parked in the other LDS buffer and chunk kc+2 requested at the TOP of
//     iteration kc, one memory instruction behind each of the 64 MFMAs of chunk kc; one barrier per chunk, mid-way;
//   * MFMAs are issued component-major so that consecutive instructions never touch the same accumulator;
//   * the K loop of a tile is ONE generated asm statement (gemm_tile_asm.inc; round 2 -- before, the interleave was a request
//     to hipcc's scheduler through sched_group_barrier and the loads were separate asm statements with compiler-allocated
//     destinations, see scan_mainloop.inc for why that is fragile); the accumulators come back in fixed registers.
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_fast_kernel(GemmArgs g_in) {
    GemmArgs g = g_in;
    __shared__ float4 lds[2][2][NPLANE][BM];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t tiles_m = (g.M + BM - 1) / BM;
    const int64_t tiles_n = (g.N + BN - 1) / BN;
    const int64_t ntile1 = tiles_m * tiles_n;
    // K slices (round 6, the training tape's few-tile products): the persistent order runs over (slice, tile) units, a unit multiplies the
    // K range [slice ksplit, (slice + 1) ksplit) and stores the raw partial tile to part + slice M N; the caller's reduction adds them.
    const int64_t nsl = g_in.ksplit > 0 ? (g_in.K + g_in.ksplit - 1) / g_in.ksplit : 1;
    const int64_t ntile = g_in.A2 ? 2 * ntile1 : ntile1 * nsl;
    // PERSISTENT: the grid is at most 2 workgroups per CU (8 XCDs x 64); XCD x owns a contiguous range of the tile order
    // (consecutive tiles walk down a column of M tiles and share the B panel in that XCD's L2) and its resident
    // workgroups stride through it.  A short-K tile lives ~25 us, and a fresh workgroup launch costs ~10 us of an idle
    // slot (88 % slot occupancy measured on the SCAN kernel), which the loop removes.
    // (Requesting the next tile's first chunk before the epilogue was tried and lost 15 %: the epilogue's own bias
    // load makes hipcc wait for vmcnt(0), i.e. for the prefetch, before the first store.)
    const int64_t q_ = ntile / 8, r_ = ntile % 8;
    const int64_t xcd = blockIdx.x % 8, slot = blockIdx.x / 8, nslots = gridDim.x / 8;
    const int64_t t_begin = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_);
    const int64_t t_count = q_ + (xcd < r_ ? 1 : 0);
    const int ld_row = tid >> 3, ld_p = tid & 7;
    constexpr unsigned OPER_BYTES = NPLANE * BM * 16u;
    const int fi = lane & 31, fg = lane >> 5;
    // LDS byte addresses of the generated loop (buffer 1, operand B and the 32-row passes are immediates there)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)&lds[0][0][0][0];
    const unsigned ls0 = lds0 + (unsigned)(ld_p * BM + (ld_row ^ ld_p)) * 16u;
    auto faddr = [&](int q, int base, int oper) -> unsigned {      // plane p = 2q + fg, row (base + fi) ^ p == base + (fi ^ p)
        const int p = 2 * q + fg;
        return lds0 + (unsigned)oper * OPER_BYTES + (unsigned)(p * BM + base + (fi ^ p)) * 16u;
    };
    const unsigned fa0 = faddr(0, wm * 64, 0), fa1 = faddr(1, wm * 64, 0), fa2 = faddr(2, wm * 64, 0), fa3 = faddr(3, wm * 64, 0);
    const unsigned fb0 = faddr(0, wn * 64, 1), fb1 = faddr(1, wn * 64, 1), fb2 = faddr(2, wn * 64, 1), fb3 = faddr(3, wn * 64, 1);

  for (int64_t tt = slot; tt < t_count; tt += nslots) {
    int64_t bid = t_begin + tt;
    int64_t k0 = 0, klen = g_in.K;
    if (g_in.ksplit > 0) {        // a K slice of tile bid % ntile1 (uniform)
        const int64_t sl = bid / ntile1;
        bid -= sl * ntile1;
        k0 = sl * g_in.ksplit;
        klen = k0 + g_in.ksplit < g_in.K ? g_in.ksplit : g_in.K - k0;
        g.C = g_in.part + sl * g_in.M * g_in.N; g.ldc = g_in.N; g.bias = nullptr; g.act = 0; g.accumulate = 0; g.Z = nullptr;
    } else if (bid >= ntile1) {   // second problem of a paired launch (uniform: bid is a scalar)
        bid -= ntile1;
        g.A = g_in.A2; g.B = g_in.B2; g.bias = g_in.bias2; g.C = g_in.C2;
    } else {
        g.A = g_in.A; g.B = g_in.B; g.bias = g_in.bias; g.C = g_in.C;
    }
    const int nk = __builtin_amdgcn_readfirstlane((int)(klen / BK));
    const int64_t tm = bid % tiles_m, tn = bid / tiles_m;
    const int64_t m0 = tm * BM, n0 = tn * BN;
    const int64_t m_end = (m0 + BM < g.M) ? m0 + BM : g.M;

    f32x16 acc[2][2];
    const char *abase = reinterpret_cast<const char *>(g.A + m0 * g.lda + k0);
    const char *bbase = reinterpret_cast<const char *>(g.B + n0 * g.ldb + k0);
    unsigned oa0, oa1, oa2, oa3, ob0, ob1, ob2, ob3;    // per-lane byte offsets (host checked: 128 rows * ld * 4 < 2^32)
    {
        auto offa = [&](int s_) -> unsigned {
            int64_t r = m0 + ld_row + 32 * s_;
            r = (r < m_end ? r : m_end - 1) - m0;
            return (unsigned)r * (unsigned)g.lda * 4u + ld_p * 16u;
        };
        auto offb = [&](int s_) -> unsigned {
            int64_t r = n0 + ld_row + 32 * s_;
            r = (r < g.N ? r : g.N - 1) - n0;
            return (unsigned)r * (unsigned)g.ldb * 4u + ld_p * 16u;
        };
        oa0 = offa(0); oa1 = offa(1); oa2 = offa(2); oa3 = offa(3);
        ob0 = offb(0); ob1 = offb(1); ob2 = offb(2); ob3 = offb(3);
    }
    // The K loop of this tile: one generated asm statement (gemm_tile_asm.inc, tools/gen_gemm_tile.py) -- chunk kc+1 is parked
    // in the other LDS buffer and chunk kc+2 requested behind the MFMAs of chunk kc's first fragment set, one barrier mid-way,
    // the next fragments are read behind the second set; it ends with nothing in flight and a barrier.
#include "gemm_tile_asm.inc"
    // (the loop ended on a barrier: nobody reads LDS any more, the next tile's prologue may overwrite it)
    gemm_epilogue(g, acc, &lds[0][0][0][0], tid, lane, wm, wn, m0, n0, m_end);
  }
}

template <bool ALIGNED>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(GemmArgs g) {
    // [buffer][operand][plane][row] float4
    __shared__ float4 lds[2][2][NPLANE][BM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: consecutive blocks of one XCD (b % 8 == const) walk down a column
    // of M tiles so they share the same B panel in that XCD's L2.
    const int64_t tiles_m = (g.M + g.rows_per_tile - 1) / g.rows_per_tile;
    const int64_t tiles_n = (g.N + BN - 1) / BN;
    int64_t bid = blockIdx.x;
    const int64_t ntile = tiles_m * tiles_n;
    {
        const int64_t q = ntile / 8, r = ntile % 8;
        const int64_t xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective remap
    }
    const int64_t tm = bid % tiles_m, tn = bid / tiles_m;
    const int64_t m0 = tm * g.rows_per_tile, n0 = tn * BN;
    const int64_t m_end = (m0 + g.rows_per_tile < g.M) ? m0 + g.rows_per_tile : g.M;
    if (g.ksplit > 0) {   // split-K slice of a skinny GEMM: shift the operands, shorten K, redirect the output
        const int64_t koff = (int64_t)blockIdx.y * g.ksplit;
        g.A += koff;
        g.B += koff;
        g.K = (g.K - koff < g.ksplit) ? g.K - koff : g.ksplit;
        g.C = g.part + (int64_t)blockIdx.y * g.M * g.N;
        g.ldc = g.N;
        g.bias = nullptr;
        g.act = 0;
        g.accumulate = 0;
    }

    const int ld_row = tid >> 3;  // 0..31
    const int ld_p = tid & 7;     // plane

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    const int64_t nk = (g.K + BK - 1) / BK;

    auto gload = [&](int64_t kc) {
        const int64_t k = kc * BK + ld_p * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ra[s] = load4<ALIGNED>(g.A, m0 + ld_row + 32 * s, m_end, g.lda, k, g.K);
            rb[s] = load4<ALIGNED>(g.B, n0 + ld_row + 32 * s, g.N, g.ldb, k, g.K);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = ld_row + 32 * s;
            lds[buf][0][ld_p][row ^ ld_p] = ra[s];
            lds[buf][1][ld_p][row ^ ld_p] = rb[s];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();

    const int fi = lane & 31, fg = lane >> 5;
    for (int64_t kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) gload(kc + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = 2 * q + fg;
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = lds[buf][0][p][(wm * 64 + i * 32 + fi) ^ p];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = lds[buf][1][p][(wn * 64 + j * 32 + fi) ^ p];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kc + 1 < nk) {
            lstore(buf ^ 1);
            __syncthreads();
        }
    }

    gemm_epilogue(g, acc, &lds[0][0][0][0], tid, lane, wm, wn, m0, n0, m_end);
}

static int launch_gemm(const GemmArgs &g, hipStream_t st) {
    if (g.M == 0 || g.N == 0) return ITR_OK;
    const int64_t tiles_m = ceil_div(g.M, g.rows_per_tile), tiles_n = ceil_div(g.N, BN);
    const int64_t nblk = tiles_m * tiles_n * (g.A2 ? 2 : 1);
    if (nblk > 0x7fffffffLL) {
        set_error("gemm: grid too large (%lld tiles)", (long long)nblk);
        return ITR_ERR_UNSUPPORTED;
    }
    const bool aligned = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (g.K % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
    const bool fast = aligned && g.group <= 1 && (g.K % BK == 0) && g.K >= BK && g.rows_per_tile == BM &&
                      (uint64_t)g.lda * 4u * BM < (1ull << 32) && (uint64_t)g.ldb * 4u * BN < (1ull << 32);
    if (g.A2 && !(fast && (reinterpret_cast<uintptr_t>(g.A2) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.B2) & 15) == 0)) {
        set_error("gemm: a paired launch needs the fast kernel's shape (K %% 32 == 0, 16-byte aligned rows)");
        return ITR_ERR_UNSUPPORTED;
    }
    if (fast) {
        const int64_t per_xcd = ceil_div(nblk, 8);
        const unsigned grid = 8u * (unsigned)(per_xcd < 64 ? per_xcd : 64);     // <= 2 workgroups per CU, persistent
        hipLaunchKernelGGL(gemm_nt_fast_kernel, dim3(grid), dim3(GEMM_THREADS), 0, st, g);
    }
    else if (aligned)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, dim3((unsigned)nblk), dim3(GEMM_THREADS), 0, st, g);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, dim3((unsigned)nblk), dim3(GEMM_THREADS), 0, st, g);
    ITR_CHECK_LAUNCH("gemm_nt");
    return ITR_OK;
}


// ---- skinny GEMMs (the GRU recurrence of a training batch: M = 128 rows gives 24 tiles for 256 CUs) ------------------
// The K range is cut into `splits` slices (grid.y), each slice writes its raw partial tile to the caller's scratch and a
// second kernel adds the slices in a fixed order, the bias, the old C (accumulate) and the activation: deterministic.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, int splits, int64_t M, int64_t N,
                                                            const float *__restrict__ bias, float *__restrict__ C, int64_t ldc, int act,
                                                            int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * N) return;
    const int64_t m = idx / N, n = idx % N;
    float v = accumulate ? C[m * ldc + n] : 0.f;
    for (int s_ = 0; s_ < splits; ++s_) v += part[(int64_t)s_ * M * N + idx];
    if (bias) v += bias[n];
    C[m * ldc + n] = apply_act(v, act);
}

size_t gemm_splitk_scratch_bytes(int64_t M, int64_t N, int splits) { return (size_t)splits * (size_t)M * (size_t)N * 4; }

// Picks the number of K slices so that the launch has >= ~256 workgroups; 1 = not worth it.
int gemm_splitk_choice(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
    if (tiles >= 128 || K < 256) return 1;
    int s_ = (int)(256 / tiles);
    while (s_ > 1 && K / s_ < 128) --s_;
    return s_ < 1 ? 1 : (s_ > 16 ? 16 : s_);
}

// The training tape's choice (itr_gemm_nt_splitk): two workgroups of the tile kernel fit a CU (64 KB of LDS each) and one alone leaves the
// matrix pipe idle while it waits for LDS, so the launch aims at 2 x 256 resident workgroups and never at a round and a bit -- BERT's
// 2 048 x 768 products over K = 3 072 are 96 tiles: 5 slices = 480 workgroups (2 slices = 192 took 150 us for a 61 us product: round 6).
int gemm_splitk_choice_fill(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
    if (tiles >= 128 || tiles < 1 || K < 256) return 1;     // (128-255 tiles: slicing measured slower on CAMERA's 144-tile products)
    int s_ = (int)(512 / tiles);
    if (s_ > 8) s_ = 8;          // (CAMERA's 36-tile products over K = 10 240: 14 slices measured slower than 7 -- the slices' sum grows with them)
    while (s_ > 1 && K / s_ < 128) --s_;
    return s_ < 1 ? 1 : s_;
}

int gemm_nt_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M, int64_t N,
                   int64_t K, int act, int accumulate, int splits, float *scratch, hipStream_t st) {
    if (M == 0 || N == 0) return ITR_OK;
    if (splits <= 1) {
        GemmArgs g1{A, B, bias, C, lda, ldb, ldc, M, N, K, act, 1, BM, nullptr, nullptr, 0, accumulate, 0, nullptr};
        return launch_gemm(g1, st);
    }
    const int64_t ksplit = ceil_div(ceil_div(K, (int64_t)splits), (int64_t)BK) * BK;       // slices start on chunk boundaries
    const int ns = (int)ceil_div(K, ksplit);
    GemmArgs g{A, B, nullptr, C, lda, ldb, ldc, M, N, K, 0, 1, BM, nullptr, nullptr, 0, 0, ksplit, scratch};
    const int64_t nblk = ceil_div(M, BM) * ceil_div(N, BN);
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    if (aligned)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, dim3((unsigned)nblk, (unsigned)ns), dim3(GEMM_THREADS), 0, st, g);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, dim3((unsigned)nblk, (unsigned)ns), dim3(GEMM_THREADS), 0, st, g);
    ITR_CHECK_LAUNCH("gemm_nt(split-K)");
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div(M * N, 256)), dim3(256), 0, st, scratch, ns, M, N, bias, C, ldc, act,
                       accumulate);
    ITR_CHECK_LAUNCH("splitk_reduce");
    return ITR_OK;
}

// The same, slices on the persistent asm tile kernel where the shape admits it (K a multiple of 32, aligned rows): the training tape's
// entry point only (itr_gemm_nt_splitk) -- the evaluation's small-batch recurrence keeps the kernel it was validated bit for bit on.
// (Round 6: the generic kernel ran the peeled tails and BERT's 96-tile products at ~70 TFLOP/s where the asm loop does ~118.)
int gemm_nt_splitk_fast(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M, int64_t N,
                        int64_t K, int act, int splits, float *scratch, hipStream_t st) {
    if (M == 0 || N == 0) return ITR_OK;
    const int64_t ksplit = ceil_div(ceil_div(K, (int64_t)(splits > 1 ? splits : 1)), (int64_t)BK) * BK;
    const int ns = (int)ceil_div(K, ksplit);
    const bool ok = splits > 1 && ns > 1 && (lda % 4 == 0) && (ldb % 4 == 0) && (K % BK == 0) && K >= BK &&
                    ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) &&
                    (uint64_t)lda * 4u * BM < (1ull << 32) && (uint64_t)ldb * 4u * BN < (1ull << 32);
    if (!ok) return gemm_nt_splitk(A, lda, B, ldb, bias, C, ldc, M, N, K, act, 0, splits, scratch, st);
    GemmArgs g{A, B, nullptr, C, lda, ldb, ldc, M, N, K, 0, 1, BM, nullptr, nullptr, 0, 0, ksplit, scratch};
    const int64_t units = ceil_div(M, BM) * ceil_div(N, BN) * ns;
    const int64_t per_xcd = ceil_div(units, 8);
    hipLaunchKernelGGL(gemm_nt_fast_kernel, dim3(8u * (unsigned)(per_xcd < 64 ? per_xcd : 64)), dim3(GEMM_THREADS), 0, st, g);
    ITR_CHECK_LAUNCH("gemm_nt_fast(split-K)");
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div(M * N, 256)), dim3(256), 0, st, scratch, ns, M, N, bias, C, ldc, act, 0);
    ITR_CHECK_LAUNCH("splitk_reduce");
    return ITR_OK;
}

// Slices only: scratch[s][m][n] (s < *n_slices) holds the raw partial products; the CONSUMER adds them (in slice order) --
// the GRU gate kernels do, which saves one launch and one pass per time step.
bool gemm_skinny_ok(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t M, int64_t N, int64_t K);      // gemm_skinny.hip
int gemm_skinny_partials(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t M, int64_t N, int64_t K, int max_slices, float *part,
                         int *n_slices, hipStream_t st);
int gemm_nt_splitk_partials(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t M, int64_t N, int64_t K, int splits, float *scratch,
                            int *n_slices, hipStream_t st) {
    *n_slices = 0;
    if (M == 0 || N == 0) return ITR_OK;
    // M <= 128 rows (a training batch's recurrence): 16-column strips over all rows instead of one 128-row tile per 128 columns
    // (gemm_skinny.hip); the caller's scratch holds 16 slices of M x N, the consumer adds whatever number of slices comes back
    if (splits > 1 && !ITR_EXP_ENV("ITR_GEMM_NO_SKINNY") && gemm_skinny_ok(A, lda, B, ldb, M, N, K))
        return gemm_skinny_partials(A, lda, B, ldb, M, N, K, 16, scratch, n_slices, st);
    const int64_t ksplit = ceil_div(ceil_div(K, (int64_t)(splits > 1 ? splits : 1)), (int64_t)BK) * BK;
    const int ns = (int)ceil_div(K, ksplit);
    GemmArgs g{A, B, nullptr, scratch, lda, ldb, N, M, N, K, 0, 1, BM, nullptr, nullptr, 0, 0, ksplit, scratch};
    const int64_t nblk = ceil_div(M, BM) * ceil_div(N, BN);
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    if (aligned)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, dim3((unsigned)nblk, (unsigned)ns), dim3(GEMM_THREADS), 0, st, g);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, dim3((unsigned)nblk, (unsigned)ns), dim3(GEMM_THREADS), 0, st, g);
    ITR_CHECK_LAUNCH("gemm_nt(split-K partials)");
    *n_slices = ns;
    return ITR_OK;
}

bool gemm_nt_stream(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, int act, hipStream_t st, int *rc, int algo);      // gemm_stream.hip

static int gemm_nt_algo(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t st, int algo);
int gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
            int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t st) {
    return gemm_nt_algo(A, lda, B, ldb, bias, C, ldc, M, N, K, act, st, 0);
}
int gemm_skinny_direct(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                       int act, hipStream_t st);      // gemm_skinny.hip
static int gemm_nt_algo(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t st, int algo) {
    // algo 4 (the training tape's request for a batch of <= 128 rows: a decoder step, a per-caption vector layer): 16-column strips over
    // all rows -- N / 16 workgroups where the tile kernel has N / 128.  Never chosen by the library itself: a row's result would then
    // depend on how many rows the call has (another k order), which the sharded evaluation's bit-identity across partitions forbids.
    if (algo == 4) {
        if (N >= 64 && K >= 128 && gemm_skinny_ok(A, lda, B, ldb, M, N, K)) return gemm_skinny_direct(A, lda, B, ldb, bias, C, ldc, M, N, K, act, st);
        algo = 0;
    }
    // short K, many row tiles: the streaming kernel takes the whole 128-row tiles, the tile kernel the remaining rows
    int rc = ITR_OK;
    if (gemm_nt_stream(A, lda, B, ldb, bias, C, ldc, M, N, K, act, st, &rc, algo)) {
        if (rc != ITR_OK) return rc;
        const int64_t done = M / BM * BM;
        if (done == M) return ITR_OK;
        GemmArgs gt{A + done * lda, B, bias, C + done * ldc, lda, ldb, ldc, M - done, N, K, act, 1, BM, nullptr, nullptr, 0, 0, 0, nullptr};
        return launch_gemm(gt, st);
    }
    GemmArgs g{A, B, bias, C, lda, ldb, ldc, M, N, K, act, 1, BM, nullptr, nullptr, 0, 0, 0, nullptr};
    return launch_gemm(g, st);
}

// Two GEMMs of one shape in one launch (the forward and reverse direction of a bi-GRU time step): C = A B^T + bias and
// C2 = A2 B2^T + bias2.  Every output element is the same fmaf chain as in the single launch.  Only for shapes the fast kernel takes
// (gemm_pair_ok); the caller falls back to two launches otherwise.
bool gemm_pair_ok(int64_t lda, int64_t ldb, int64_t K) {
    return lda % 4 == 0 && ldb % 4 == 0 && K % BK == 0 && K >= BK && (uint64_t)lda * 4u * BM < (1ull << 32) && (uint64_t)ldb * 4u * BN < (1ull << 32);
}
int gemm_nt_pair(const float *A, const float *A2, int64_t lda, const float *B, const float *B2, int64_t ldb, const float *bias, const float *bias2,
                 float *C, float *C2, int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    GemmArgs g{A, B, bias, C, lda, ldb, ldc, M, N, K, 0, 1, BM, nullptr, nullptr, 0, 0, 0, nullptr, A2, B2, bias2, C2};
    return launch_gemm(g, st);
}

int gemm_nt_acc(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc,
                int64_t M, int64_t N, int64_t K, int act, hipStream_t st) {
    GemmArgs g{A, B, bias, C, lda, ldb, ldc, M, N, K, act, 1, BM, nullptr, nullptr, 0, 1, 0, nullptr};
    return launch_gemm(g, st);
}

int gemm_nt_sqdiff(const float *A, int64_t lda, const float *B, int64_t ldb, const float *rowscale, const float *Z,
                   int64_t ldz, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    GemmArgs g{A, B, nullptr, C, lda, ldb, ldc, M, N, K, 0, 1, BM, rowscale, Z, ldz, 0, 0, nullptr};
    return launch_gemm(g, st);
}

int gemm_nt_groupmax(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                     int64_t Mgroups, int group, int64_t N, int64_t K, hipStream_t st) {
    GemmArgs g{A, B, nullptr, C, lda, ldb, ldc, Mgroups * group, N, K, 0, group, (BM / group) * group, nullptr, nullptr, 0, 0, 0, nullptr};
    return launch_gemm(g, st);
}

}  // namespace itr

extern "C" int itr_gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                           float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                           itr_stream_t stream) {
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 0, "itr_gemm_nt: negative dimension");
    if (M == 0 || N == 0) return ITR_OK;   // empty result: empty tensors carry null pointers
    ITR_REQUIRE(A && B && C, "itr_gemm_nt: null pointer");
    // lda < K is allowed: overlapping A rows express a convolution over consecutive rows (SAEM conv head)
    ITR_REQUIRE(lda >= 1 && ldb >= K && ldc >= N, "itr_gemm_nt: leading dimension smaller than row");
    ITR_REQUIRE(act >= 0 && act <= 6, "itr_gemm_nt: unknown activation %d", act);
    return itr::gemm_nt(A, lda, B, ldb, bias, C, ldc, M, N, K, act, itr::as_stream(stream));
}

extern "C" int itr_gemm_nt_algo(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                                float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, int algo,
                                itr_stream_t stream) {
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 0, "itr_gemm_nt_algo: negative dimension");
    if (M == 0 || N == 0) return ITR_OK;
    ITR_REQUIRE(A && B && C, "itr_gemm_nt_algo: null pointer");
    ITR_REQUIRE(lda >= 1 && ldb >= K && ldc >= N, "itr_gemm_nt_algo: leading dimension smaller than row");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_gemm_nt_algo: unknown activation %d", act);
    ITR_REQUIRE(algo >= 0 && algo <= 4, "itr_gemm_nt_algo: algo must be 0 (auto), 1 (tile), 2 (stream, plain map), 3 (stream, XCD map) or 4 (skinny: <= 128 rows)");
    return itr::gemm_nt_algo(A, lda, B, ldb, bias, C, ldc, M, N, K, act, itr::as_stream(stream), algo);
}

extern "C" int itr_gemm_nt_acc(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                               int64_t ldc, int64_t M, int64_t N, int64_t K, int act, itr_stream_t stream) {
    ITR_REQUIRE(A && B && C, "itr_gemm_nt_acc: null pointer");
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 0 && lda >= 1 && ldb >= K && ldc >= N, "itr_gemm_nt_acc: bad shape");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_gemm_nt_acc: unknown activation %d", act);
    return itr::gemm_nt_acc(A, lda, B, ldb, bias, C, ldc, M, N, K, act, itr::as_stream(stream));
}

extern "C" int itr_gemm_nt_residual(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, const float *R, int64_t ldr,
                                    float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, itr_stream_t stream) {
    ITR_REQUIRE(A && B && R && C, "itr_gemm_nt_residual: null pointer");
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 0 && lda >= 1 && ldb >= K && ldc >= N && ldr >= N, "itr_gemm_nt_residual: bad shape");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_gemm_nt_residual: unknown activation %d", act);
    itr::GemmArgs g{A, B, bias, C, lda, ldb, ldc, M, N, K, act, 1, itr::BM, nullptr, R, ldr, 2, 0, nullptr};
    return itr::launch_gemm(g, itr::as_stream(stream));
}

extern "C" int itr_cosine_scores(const float *im, const float *s, float *S, int64_t Ni, int64_t Nc, int D,
                                 int64_t ldS, itr_stream_t stream) {
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && D > 0 && ldS >= Nc, "itr_cosine_scores: bad shape");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    ITR_REQUIRE(im && s && S, "itr_cosine_scores: null pointer");
    return itr::gemm_nt(im, D, s, D, nullptr, S, ldS, Ni, Nc, D, 0, itr::as_stream(stream));
}

extern "C" int itr_mvm_scores(const float *imgs, const float *caps, float *S, int64_t Ni, int64_t Nc, int k,
                              int D, int64_t ldS, itr_stream_t stream) {
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && D > 0 && ldS >= Nc, "itr_mvm_scores: bad shape");
    if (Ni == 0 || Nc == 0) return ITR_OK;
    ITR_REQUIRE(imgs && caps && S, "itr_mvm_scores: null pointer");
    ITR_REQUIRE(k >= 1 && k <= itr::BM, "itr_mvm_scores: number of views must be in [1, %d]", itr::BM);
    return itr::gemm_nt_groupmax(imgs, D, caps, D, S, ldS, Ni, k, Nc, D, itr::as_stream(stream));
}

// Training tape only (a row's bits depend on the slice count, hence on M and N): the product with its K range cut into slices when the
// output has too few 128 x 128 tiles to fill the chip -- CAMERA's dilated convolutions as GEMMs are 4 608 x 128 outputs over K = 6 144 /
// 10 240: 36 tiles, 0.4 ms each on 40 workgroups.  Slices are added in a fixed order (deterministic).
extern "C" size_t itr_gemm_nt_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    if (M <= 128 && N >= 16 && K >= 128) return itr::gemm_splitk_scratch_bytes(M, N, 16);      // the skinny kernel's slices (at most 16)
    const int s_ = itr::gemm_splitk_choice_fill(M, N, K);
    return s_ > 1 ? itr::gemm_splitk_scratch_bytes(M, N, s_) : 0;
}

extern "C" int itr_gemm_nt_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M,
                                  int64_t N, int64_t K, int act, void *workspace, size_t workspace_bytes, itr_stream_t stream) {
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 0, "itr_gemm_nt_splitk: negative dimension");
    if (M == 0 || N == 0) return ITR_OK;
    ITR_REQUIRE(A && B && C, "itr_gemm_nt_splitk: null pointer");
    ITR_REQUIRE(lda >= K && ldb >= K && ldc >= N, "itr_gemm_nt_splitk: leading dimension smaller than row");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_gemm_nt_splitk: unknown activation %d", act);
    // <= 128 rows (a decoder step's layers: 16-96 strips of 16 columns, a workgroup's life ~1 us per 64 k): the skinny kernel over K slices
    // -- 128-384 short workgroups -- and one pass that adds the slices, the bias and the activation (21 -> ~10 us on 128 x 512 x 512)
    if (M <= 128 && N >= 16 && K >= 128 && itr::gemm_skinny_ok(A, lda, B, ldb, M, N, K) && workspace &&
        workspace_bytes >= itr::gemm_splitk_scratch_bytes(M, N, 16)) {
        int ns = 0;
        const int rc = itr::gemm_skinny_partials(A, lda, B, ldb, M, N, K, 16, static_cast<float *>(workspace), &ns, itr::as_stream(stream));
        if (rc != ITR_OK) return rc;
        hipLaunchKernelGGL(itr::splitk_reduce_kernel, dim3((unsigned)itr::ceil_div(M * N, (int64_t)256)), dim3(256), 0, itr::as_stream(stream),
                           static_cast<const float *>(workspace), ns, M, N, bias, C, ldc, act, 0);
        ITR_CHECK_LAUNCH("splitk_reduce");
        return ITR_OK;
    }
    const int s_ = itr::gemm_splitk_choice_fill(M, N, K);
    if (s_ <= 1) return itr::gemm_nt(A, lda, B, ldb, bias, C, ldc, M, N, K, act, itr::as_stream(stream));
    ITR_REQUIRE(workspace && workspace_bytes >= itr::gemm_splitk_scratch_bytes(M, N, s_), "itr_gemm_nt_splitk: workspace too small (itr_gemm_nt_splitk_workspace_bytes)");
    return itr::gemm_nt_splitk_fast(A, lda, B, ldb, bias, C, ldc, M, N, K, act, s_, static_cast<float *>(workspace), itr::as_stream(stream));
}
