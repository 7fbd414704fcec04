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
    int accumulate;   // C = act(C + A*B^T + bias): sums the taps of a dilated Conv1d (camera_.py:100-103)
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
                        if (g.Z) {
                            v = v * g.rowscale[row] - g.Z[row * g.ldz + col];
                            v = v * v;
                        } else {
                            if (g.accumulate) v += g.C[row * g.ldc + col];
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

static int launch_gemm(const GemmArgs &g, hipStream_t st) {
    if (g.M == 0 || g.N == 0) return ITR_OK;
    const int64_t tiles_m = ceil_div(g.M, g.rows_per_tile), tiles_n = ceil_div(g.N, BN);
    const int64_t nblk = tiles_m * tiles_n;
    if (nblk > 0x7fffffffLL) {
        set_error("gemm: grid too large (%lld tiles)", (long long)nblk);
        return ITR_ERR_UNSUPPORTED;
    }
    const bool aligned = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (g.K % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
    if (aligned)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, dim3((unsigned)nblk), dim3(GEMM_THREADS), 0, st, g);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, dim3((unsigned)nblk), dim3(GEMM_THREADS), 0, st, g);
    ITR_CHECK_LAUNCH("gemm_nt");
    return ITR_OK;
}

int gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
            int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t st) {
    GemmArgs g{A, B, bias, C, lda, ldb, ldc, M, N, K, act, 1, BM, nullptr, nullptr, 0, 0};
    return launch_gemm(g, st);
}

int gemm_nt_acc(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc,
                int64_t M, int64_t N, int64_t K, int act, hipStream_t st) {
    GemmArgs g{A, B, bias, C, lda, ldb, ldc, M, N, K, act, 1, BM, nullptr, nullptr, 0, 1};
    return launch_gemm(g, st);
}

int gemm_nt_sqdiff(const float *A, int64_t lda, const float *B, int64_t ldb, const float *rowscale, const float *Z,
                   int64_t ldz, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    GemmArgs g{A, B, nullptr, C, lda, ldb, ldc, M, N, K, 0, 1, BM, rowscale, Z, ldz, 0};
    return launch_gemm(g, st);
}

int gemm_nt_groupmax(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                     int64_t Mgroups, int group, int64_t N, int64_t K, hipStream_t st) {
    GemmArgs g{A, B, nullptr, C, lda, ldb, ldc, Mgroups * group, N, K, 0, group, (BM / group) * group, nullptr, nullptr, 0, 0};
    return launch_gemm(g, st);
}

}  // namespace itr

extern "C" int itr_gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                           float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                           itr_stream_t stream) {
    ITR_REQUIRE(A && B && C, "itr_gemm_nt: null pointer");
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 0, "itr_gemm_nt: negative dimension");
    // lda < K is allowed: overlapping A rows express a convolution over consecutive rows (SAEM conv head)
    ITR_REQUIRE(lda >= 1 && ldb >= K && ldc >= N, "itr_gemm_nt: leading dimension smaller than row");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_gemm_nt: unknown activation %d", act);
    return itr::gemm_nt(A, lda, B, ldb, bias, C, ldc, M, N, K, act, itr::as_stream(stream));
}

extern "C" int itr_gemm_nt_acc(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                               int64_t ldc, int64_t M, int64_t N, int64_t K, int act, itr_stream_t stream) {
    ITR_REQUIRE(A && B && C, "itr_gemm_nt_acc: null pointer");
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 0 && lda >= 1 && ldb >= K && ldc >= N, "itr_gemm_nt_acc: bad shape");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_gemm_nt_acc: unknown activation %d", act);
    return itr::gemm_nt_acc(A, lda, B, ldb, bias, C, ldc, M, N, K, act, itr::as_stream(stream));
}

extern "C" int itr_cosine_scores(const float *im, const float *s, float *S, int64_t Ni, int64_t Nc, int D,
                                 int64_t ldS, itr_stream_t stream) {
    ITR_REQUIRE(im && s && S, "itr_cosine_scores: null pointer");
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && D > 0 && ldS >= Nc, "itr_cosine_scores: bad shape");
    return itr::gemm_nt(im, D, s, D, nullptr, S, ldS, Ni, Nc, D, 0, itr::as_stream(stream));
}

extern "C" int itr_mvm_scores(const float *imgs, const float *caps, float *S, int64_t Ni, int64_t Nc, int k,
                              int D, int64_t ldS, itr_stream_t stream) {
    ITR_REQUIRE(imgs && caps && S, "itr_mvm_scores: null pointer");
    ITR_REQUIRE(Ni >= 0 && Nc >= 0 && D > 0 && ldS >= Nc, "itr_mvm_scores: bad shape");
    ITR_REQUIRE(k >= 1 && k <= itr::BM, "itr_mvm_scores: number of views must be in [1, %d]", itr::BM);
    return itr::gemm_nt_groupmax(imgs, D, caps, D, S, ldS, Ni, k, Nc, D, itr::as_stream(stream));
}
