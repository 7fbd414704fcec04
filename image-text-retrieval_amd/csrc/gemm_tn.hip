// fp32 "TN" GEMM with a split reduction on the gfx950 matrix cores:  C[P,Q] (+)= A[R,P]^T * B[R,Q].
//
// This is the weight gradient of every dense layer of the training step (dW = dY^T X, Models.py:139-144 -> loss.backward()):
// the REDUCTION runs over the rows (R = batch x regions / words / graph nodes: 4 608 ... 300 000), the output is the small weight
// matrix (32 x 32 ... 2 048 x 2 048).  Both operands are row-major with the reduced index as the ROW, so neither is K-contiguous:
// the NT kernel (gemm_f32.hip) needed two transposed copies, and its grid follows the OUTPUT tiles -- a 32 x 32 gradient over
// 295 000 rows was ONE workgroup (15.8 ms, CAMERA's gate layers; profiles/r06/train/before).  Here
//   * a chunk of 16 rows of each operand tile is staged in LDS exactly as it lies in memory ([row][column], coalesced float4 loads);
//     v_mfma_f32_16x16x4_f32 takes A[i][k] / B[k][j] with lane = 16 k + i -- i.e. four 16-float row segments per operand read,
//     which the row stride T + 16 spreads over all 64 banks;
//   * blockIdx.y owns a slice of the rows and writes a raw partial product; gemm_tn_reduce_kernel adds the slices in slice order
//     (deterministic) and stores or accumulates.  The slice count is chosen so that tiles x slices ~ 4 workgroups per CU.
// Exact fp32 (the MFMA is an fmaf chain); only the order of the row sum differs from the NT form.
#include "itr_common.h"

namespace itr {

constexpr int TN_RK = 16;
constexpr int TN_THREADS = 256;
constexpr int TN_MAXSLICES = 512;

template <int WM, bool ALIGNED>
__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
                                                             float *__restrict__ out, int64_t ldo, int64_t slice_stride, int64_t R, int P, int Q,
                                                             int64_t rows_per_slice, int64_t batch_a, int64_t batch_b, int64_t batch_o,
                                                             float *__restrict__ colsum) {
    constexpr int T = 32 * WM;           // tile extent in both output dimensions: 2 x 2 waves of WM x WM MFMA tiles
    constexpr int LD = T + 16;           // LDS row stride: the four k rows of an operand read land on four different 16-bank groups
    constexpr int NV = 4 * T;            // float4 per operand and chunk
    constexpr int NL = (NV + TN_THREADS - 1) / TN_THREADS;
    __shared__ __attribute__((aligned(16))) float As[2][TN_RK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][TN_RK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    A += (int64_t)blockIdx.z * batch_a;          // batched form (one problem per blockIdx.z, a single slice each)
    B += (int64_t)blockIdx.z * batch_b;
    out += (int64_t)blockIdx.z * batch_o;
    const int tiles_q = (Q + T - 1) / T;
    const int p0 = (int)(blockIdx.x / tiles_q) * T, q0 = (int)(blockIdx.x % tiles_q) * T;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_slice;
    const int64_t r_end = r_begin + rows_per_slice < R ? r_begin + rows_per_slice : R;

    f32x4 acc[WM][WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // optional bias gradient: the column sums of A (db = sum_r dY[r, :]) ride along in the workgroups of the first column tile -- a thread
    // always loads the same four columns (256 % (T / 4) == 0), so it sums what it loads and the workgroup folds the 16 row slots at the end
    const bool do_cs = colsum != nullptr && q0 == 0;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ra[NL], rb[NL];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int idx = tid + l * TN_THREADS;
            ra[l] = make_float4(0.f, 0.f, 0.f, 0.f);
            rb[l] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NV) {
                const int row = idx / (T / 4), c4 = (idx % (T / 4)) * 4;
                const int64_t r = r0 + row;
                if (r < r_end) {
                    const float *pa = A + r * lda + p0 + c4, *pb = B + r * ldb + q0 + c4;
                    if (ALIGNED) {
                        if (p0 + c4 < P) ra[l] = *reinterpret_cast<const float4 *>(pa);
                        if (q0 + c4 < Q) rb[l] = *reinterpret_cast<const float4 *>(pb);
                    } else {
                        if (p0 + c4 + 0 < P) ra[l].x = pa[0];
                        if (p0 + c4 + 1 < P) ra[l].y = pa[1];
                        if (p0 + c4 + 2 < P) ra[l].z = pa[2];
                        if (p0 + c4 + 3 < P) ra[l].w = pa[3];
                        if (q0 + c4 + 0 < Q) rb[l].x = pb[0];
                        if (q0 + c4 + 1 < Q) rb[l].y = pb[1];
                        if (q0 + c4 + 2 < Q) rb[l].z = pb[2];
                        if (q0 + c4 + 3 < Q) rb[l].w = pb[3];
                    }
                }
            }
            if (do_cs) { cs.x += ra[l].x; cs.y += ra[l].y; cs.z += ra[l].z; cs.w += ra[l].w; }
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int idx = tid + l * TN_THREADS;
            if (idx < NV) {
                const int row = idx / (T / 4), c4 = (idx % (T / 4)) * 4;
                *reinterpret_cast<float4 *>(&As[buf][row][c4]) = ra[l];
                *reinterpret_cast<float4 *>(&Bs[buf][row][c4]) = rb[l];
            }
        }
    };

    if (r_begin < r_end) {
        fetch(r_begin);
        park(0);
        __syncthreads();
        int buf = 0;
        for (int64_t r0 = r_begin; r0 < r_end; r0 += TN_RK) {
            const bool more = r0 + TN_RK < r_end;
            if (more) fetch(r0 + TN_RK);
#pragma unroll
            for (int kk = 0; kk < TN_RK / 4; ++kk) {
                const int kr = kk * 4 + (lane >> 4);
                float a[WM], b[WM];
#pragma unroll
                for (int i = 0; i < WM; ++i) a[i] = As[buf][kr][wp * 16 * WM + i * 16 + (lane & 15)];
#pragma unroll
                for (int j = 0; j < WM; ++j) b[j] = Bs[buf][kr][wq * 16 * WM + j * 16 + (lane & 15)];
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (more) park(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    if (do_cs) {          // (the last __syncthreads of the loop has passed: the operand buffers are free)
        float *red = &As[0][0][0];
        *reinterpret_cast<float4 *>(red + tid * 4) = cs;
        __syncthreads();
        if (tid < T && p0 + tid < P) {
            float t = 0.f;
            for (int j = tid >> 2; j < TN_THREADS; j += T / 4) t += red[j * 4 + (tid & 3)];
            colsum[(int64_t)blockIdx.y * P + p0 + tid] = t;
        }
    }
    float *o = out + (int64_t)blockIdx.y * slice_stride;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WM; ++j) {
            const int q = q0 + wq * 16 * WM + j * 16 + (lane & 15);
            if (q >= Q) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = p0 + wp * 16 * WM + i * 16 + 4 * (lane >> 4) + r;
                if (p < P) o[(int64_t)p * ldo + q] = acc[i][j][r];
            }
        }
}

// C[p, q] = (accumulate ? C : 0) + the sum over the slices, in a fixed order (four interleaved running sums, 16 loads in flight: with one
// load per loop trip a 512-slice sum was 512 L2 round trips per thread -- up to 80 us per call, 1.25 ms per SGRAF step in round 6).
// Blocks past the P x Q elements sum the column-sum slices cs [nsl][P] into colsum (one launch for both).
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float *__restrict__ part, int nsl, int P, int Q, float *__restrict__ C, int64_t ldc,
                                                             int accumulate, const float *__restrict__ cs, float *__restrict__ colsum, int main_blocks) {
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t n = (int64_t)P * Q;
    const float *src = part;
    float *dst;
    float s0 = 0.f;
    if ((int)blockIdx.x >= main_blocks) {
        e -= (int64_t)main_blocks * 256;
        n = P;
        if (e >= n) return;
        src = cs;
        dst = colsum + e;
    } else {
        if (e >= n) return;
        dst = C + (e / Q) * ldc + e % Q;
        if (accumulate) s0 = *dst;
    }
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 16 <= nsl; k += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = src[(int64_t)(k + i) * n + e];
#pragma unroll
        for (int i = 0; i < 16; i += 4) { s0 += v[i]; s1 += v[i + 1]; s2 += v[i + 2]; s3 += v[i + 3]; }
    }
    for (; k < nsl; ++k) s0 += src[(int64_t)k * n + e];
    *dst = (s0 + s1) + (s2 + s3);
}

// the smallest tile that covers the smaller output dimension in ONE tile where it can (<= 128): a 36-row output (d regions = P^T dC per
// image) on 32-wide tiles read the other operand -- 872 MB -- twice
static int tn_tile(int P, int Q) {
    const int m = P < Q ? P : Q;
    return m > 64 ? 128 : (m > 32 ? 64 : 32);
}
static void tn_plan(int64_t R, int P, int Q, int *T, int *nsl, int64_t *rows_per_slice) {
    *T = tn_tile(P, Q);
    const int64_t tiles = ceil_div(P, *T) * ceil_div(Q, *T);
#ifndef ITR_TN_TARGET
#define ITR_TN_TARGET 1024
#endif
    int64_t s = ceil_div((int64_t)ITR_TN_TARGET, tiles);       // ~4 workgroups per CU
    const int64_t by_rows = ceil_div(R, (int64_t)256);          // a slice is at least 256 rows (16 chunks)
    if (s > by_rows) s = by_rows;
    if (s > TN_MAXSLICES) s = TN_MAXSLICES;
    if (s < 1) s = 1;
    int64_t rps = ceil_div(ceil_div(R, s), (int64_t)TN_RK) * TN_RK;
    if (rps < TN_RK) rps = TN_RK;
    *rows_per_slice = rps;
    *nsl = (int)(R > 0 ? ceil_div(R, rps) : 1);
}

size_t gemm_tn_workspace_bytes(int64_t R, int P, int Q) {
    int T, nsl;
    int64_t rps;
    tn_plan(R, P, Q, &T, &nsl, &rps);
    return (size_t)nsl * P * ((size_t)Q + 1) * sizeof(float);      // partial products + partial column sums
}

template <int WM>
static void tn_launch(bool aligned, dim3 grid, hipStream_t st, const float *A, int64_t lda, const float *B, int64_t ldb, float *out, int64_t ldo,
                      int64_t slice_stride, int64_t R, int P, int Q, int64_t rps, int64_t ba = 0, int64_t bb = 0, int64_t bo = 0, float *cs = nullptr) {
    if (aligned)
        hipLaunchKernelGGL((gemm_tn_kernel<WM, true>), grid, dim3(TN_THREADS), 0, st, A, lda, B, ldb, out, ldo, slice_stride, R, P, Q, rps, ba, bb, bo, cs);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<WM, false>), grid, dim3(TN_THREADS), 0, st, A, lda, B, ldb, out, ldo, slice_stride, R, P, Q, rps, ba, bb, bo, cs);
}

int gemm_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, int64_t R, int P, int Q, int accumulate,
            float *colsum_a, void *workspace, size_t workspace_bytes, hipStream_t st) {
    int T, nsl;
    int64_t rps;
    tn_plan(R, P, Q, &T, &nsl, &rps);
    const bool aligned = lda % 4 == 0 && ldb % 4 == 0 && P % 4 == 0 && Q % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(B) & 15) == 0;
    const bool direct = nsl == 1 && !accumulate;
    if (!direct && workspace_bytes < (size_t)nsl * P * ((size_t)Q + 1) * sizeof(float)) {
        set_error("gemm_tn: workspace of %zu bytes, %zu needed (itr_gemm_tn_workspace_bytes)", workspace_bytes, (size_t)nsl * P * ((size_t)Q + 1) * sizeof(float));
        return ITR_ERR_BADARG;
    }
    float *out = direct ? C : static_cast<float *>(workspace);
    float *cs = !colsum_a ? nullptr : (direct ? colsum_a : static_cast<float *>(workspace) + (size_t)nsl * P * Q);
    const int64_t ldo = direct ? ldc : Q;
    const dim3 grid((unsigned)(ceil_div(P, T) * ceil_div(Q, T)), (unsigned)nsl);
    if (T == 128) tn_launch<4>(aligned, grid, st, A, lda, B, ldb, out, ldo, (int64_t)P * Q, R, P, Q, rps, 0, 0, 0, cs);
    else if (T == 64) tn_launch<2>(aligned, grid, st, A, lda, B, ldb, out, ldo, (int64_t)P * Q, R, P, Q, rps, 0, 0, 0, cs);
    else tn_launch<1>(aligned, grid, st, A, lda, B, ldb, out, ldo, (int64_t)P * Q, R, P, Q, rps, 0, 0, 0, cs);
    ITR_CHECK_LAUNCH("gemm_tn");
    if (!direct) {
        const int main_blocks = (int)ceil_div((int64_t)P * Q, (int64_t)256), cs_blocks = cs ? (int)ceil_div((int64_t)P, (int64_t)256) : 0;
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)(main_blocks + cs_blocks)), dim3(256), 0, st, (const float *)out, nsl, P, Q, C, ldc, accumulate,
                           (const float *)cs, colsum_a, main_blocks);
        ITR_CHECK_LAUNCH("gemm_tn_reduce");
    }
    return ITR_OK;
}

// `batch` independent problems of one shape in one launch, each reduced by a single workgroup row (no slices): C_z = A_z^T B_z.
int gemm_tn_batched(const float *A, int64_t lda, int64_t batch_a, const float *B, int64_t ldb, int64_t batch_b, float *C, int64_t ldc, int64_t batch_c,
                    int64_t R, int P, int Q, int batch, hipStream_t st) {
    const int T = tn_tile(P, Q);
    const bool aligned = lda % 4 == 0 && ldb % 4 == 0 && P % 4 == 0 && Q % 4 == 0 && batch_a % 4 == 0 && batch_b % 4 == 0 &&
                         (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
    const int64_t rps = R > 0 ? ceil_div(R, (int64_t)TN_RK) * TN_RK : TN_RK;
    const dim3 grid((unsigned)(ceil_div(P, T) * ceil_div(Q, T)), 1, (unsigned)batch);
    if (T == 128) tn_launch<4>(aligned, grid, st, A, lda, B, ldb, C, ldc, 0, R, P, Q, rps, batch_a, batch_b, batch_c);
    else if (T == 64) tn_launch<2>(aligned, grid, st, A, lda, B, ldb, C, ldc, 0, R, P, Q, rps, batch_a, batch_b, batch_c);
    else tn_launch<1>(aligned, grid, st, A, lda, B, ldb, C, ldc, 0, R, P, Q, rps, batch_a, batch_b, batch_c);
    ITR_CHECK_LAUNCH("gemm_tn_batched");
    return ITR_OK;
}

}  // namespace itr

extern "C" int itr_gemm_tn_batched(const float *A, int64_t lda, int64_t batch_a, const float *B, int64_t ldb, int64_t batch_b, float *C, int64_t ldc,
                                   int64_t batch_c, int64_t R, int64_t P, int64_t Q, int64_t batch, itr_stream_t stream) {
    ITR_REQUIRE(R >= 0 && P >= 0 && Q >= 0 && P <= 0x3fffffff && Q <= 0x3fffffff && batch >= 0 && batch <= 65535, "itr_gemm_tn_batched: bad shape");
    if (P == 0 || Q == 0 || batch == 0) return ITR_OK;
    ITR_REQUIRE(C && (R == 0 || (A && B)), "itr_gemm_tn_batched: null pointer");
    ITR_REQUIRE(lda >= P && ldb >= Q && ldc >= Q, "itr_gemm_tn_batched: leading dimension smaller than row");
    return itr::gemm_tn_batched(A, lda, batch_a, B, ldb, batch_b, C, ldc, batch_c, R, (int)P, (int)Q, (int)batch, itr::as_stream(stream));
}

extern "C" size_t itr_gemm_tn_workspace_bytes(int64_t R, int64_t P, int64_t Q) {
    if (R < 0 || P < 1 || Q < 1 || P > 0x3fffffff || Q > 0x3fffffff) return 0;
    return itr::gemm_tn_workspace_bytes(R, (int)P, (int)Q);
}

extern "C" int itr_gemm_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, int64_t R, int64_t P, int64_t Q,
                           int accumulate, float *colsum_a, void *workspace, size_t workspace_bytes, itr_stream_t stream) {
    ITR_REQUIRE(R >= 0 && P >= 0 && Q >= 0 && P <= 0x3fffffff && Q <= 0x3fffffff, "itr_gemm_tn: bad shape");
    if (P == 0 || Q == 0) return ITR_OK;
    ITR_REQUIRE(C && (R == 0 || (A && B)), "itr_gemm_tn: null pointer");
    ITR_REQUIRE(lda >= P && ldb >= Q && ldc >= Q, "itr_gemm_tn: leading dimension smaller than row");
    ITR_REQUIRE(P * Q <= (int64_t)0x7fffffff * 256, "itr_gemm_tn: output too large");
    return itr::gemm_tn(A, lda, B, ldb, C, ldc, R, (int)P, (int)Q, accumulate ? 1 : 0, colsum_a, workspace, workspace_bytes, itr::as_stream(stream));
}
