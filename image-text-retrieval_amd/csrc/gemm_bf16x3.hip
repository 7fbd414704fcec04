// Split-bf16 GEMM study (STUDY_SPLIT_PRECISION.md): C[M,N] = A[M,K] B[N,K]^T (+ bias) on the bf16 matrix core with fp32 operands
// split into bf16 planes,  x = hi + lo + O(2^-17 |x|):
//   TERMS = 3   hi.hi + hi.lo + lo.hi   ("bf16x3": dropped terms ~ 2^-17 per product, fp32 accumulation)
//   TERMS = 1   hi.hi                   (plain bf16 inputs, fp32 accumulation)
//   TERMS = 4   "fp16x3": fp16 planes of x' = s x (s a power of two from the tensor's absmax, so hi stays normal -- the matrix core
//               flushes fp16 subnormals), lo stored scaled by 2^11, hi.hi in one accumulator set and hi.lo' + lo'.hi in a second
//               one, folded back with exact powers of two in the epilogue: 11 + 11 mantissa bits, error at the fp32 rounding level
// v_mfma_f32_32x32x16_bf16 runs 16x the fp32 MFMA rate, so bf16x3 has a ceiling of 2.5 PF / 3 = 833 TFLOP/s of
// fp32-equivalent work against 157 for v_mfma_f32_32x32x2_f32.  NOT wired into any default path: the product GEMM
// (gemm_f32.hip) is exact fp32; this kernel is reported separately and opt-in.
//
// Operand format: per row and 32-wide K chunk the 64 B of hi are followed by the 64 B of lo ([rows][K / 32][hi | lo][32] bf16),
// so one chunk of one row is one full 128-byte line (separate whole-row planes made every load use half a line; the vector
// L1 path then limits -- found on the SCAN loop, STUDY_SPLIT_PRECISION.md).
// Tiling: 128 x 128 per workgroup, 4 waves as 2 x 2, each wave 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs);
// K chunks of 32 staged through LDS (row stride 80 B: 16 consecutive rows hit 16 different 16-byte bank groups), the next
// chunk's global loads are in flight in registers while the current one is multiplied.
#include "itr_common.h"

namespace itr {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;   // (arrays of HIP uint4 that live across blocks go to scratch)

constexpr int GB_T = 128, GB_BK = 32, GB_ROWB = 80, GB_PLANE = GB_T * GB_ROWB;   // 10 240 B per operand plane

__global__ __launch_bounds__(256) void split_bf16_kernel(const float *__restrict__ x, uint16_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // K % 32 == 0, so chunks never straddle rows
    if (i >= n) return;
    auto rne = [](float v) -> uint32_t {       // fp32 -> bf16 bits, round to nearest even (finite inputs)
        const uint32_t u = __float_as_uint(v);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    const float v = x[i];
    const uint32_t h = rne(v);
    const int64_t o = (i >> 5) * 64 + (i & 31);
    out[o] = (uint16_t)h;
    out[o + 32] = (uint16_t)rne(v - __uint_as_float(h << 16));
}

// ---- fp16x3 operand preparation: absmax -> power-of-two scale -> fp16 hi | scaled fp16 lo planes
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    __shared__ float red[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup (16 k same-address atomics cost more than the split itself); non-negative floats order like their bits
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
__device__ __forceinline__ float f16_scale(unsigned absmax_bits) {
    const float m = __uint_as_float(absmax_bits);
    if (!(m > 0.f) || !(m < 3.0e38f)) return 1.f;
    return ldexpf(1.f, 14 - ilogbf(m));                                    // m * s in [2^14, 2^15): far from 65 504, tiny values stay normal
}
__global__ __launch_bounds__(256) void split_f16_kernel(const float *__restrict__ x, uint16_t *__restrict__ out, int64_t n,
                                                        const unsigned *__restrict__ absmax, float *__restrict__ inv_scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float s = f16_scale(*absmax);
    if (i == 0) *inv_scale = 1.f / s;
    if (i >= n) return;
    const float xs = x[i] * s;
    const _Float16 hh = (_Float16)xs;
    const _Float16 ll = (_Float16)((xs - (float)hh) * 2048.f);
    const int64_t o = (i >> 5) * 64 + (i & 31);
    out[o] = __builtin_bit_cast(uint16_t, hh);
    out[o + 32] = __builtin_bit_cast(uint16_t, ll);
}

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int TERMS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B,
                                                           const float *__restrict__ bias, float *__restrict__ C, int64_t ldc,
                                                           int64_t M, int64_t N, int K, int64_t lda, int64_t ldb, int act,
                                                           int tiles_m, int tiles_n, const float *__restrict__ inv_sa,
                                                           const float *__restrict__ inv_sb) {
    constexpr bool F16 = TERMS == 4, SPLIT = TERMS >= 3;
    __shared__ __attribute__((aligned(16))) char lds[4 * GB_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // block -> tile: every XCD (block b runs on XCD b % 8) walks a contiguous range of a grouped order (16 row tiles share
    // one column tile's B panel back to back), so its L2 keeps the panels it is re-reading
    const int nb = tiles_m * tiles_n;
    int pid = blockIdx.x;
    {
        const int per = (nb + 7) / 8;
        const int p2 = (pid & 7) * per + (pid >> 3);
        pid = p2 < nb && (nb & 7) == 0 ? p2 : pid;
    }
    constexpr int GM = 16;
    const int group = pid / (GM * tiles_n), first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (pid % (GM * tiles_n)) % gsz, tn = (pid % (GM * tiles_n)) / gsz;
    const int64_t m0 = (int64_t)tm * GB_T, n0 = (int64_t)tn * GB_T;

    // global -> register staging: thread = (row r, 16-byte granule gq of the 64-byte chunk row); rows r and r + 64
    // TERMS == 3: 8 lanes cover the whole 128-byte line of one row chunk (granule p: hi for p < 4, lo otherwise), 32 rows per
    // pass, 4 passes per operand.  TERMS == 1: only the hi half is needed -- 4 lanes per row, 64 rows per pass, 2 passes.
    constexpr int LPR = SPLIT ? 8 : 4, PASS = SPLIT ? 4 : 2, RPP = 256 / LPR;
    const int r = tid / LPR, p = tid % LPR;
    int64_t ga[PASS], gb[PASS];
    unsigned la[PASS];
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
        ga[i] = min(m0 + r + RPP * i, M - 1) * lda + p * 8;
        gb[i] = min(n0 + r + RPP * i, N - 1) * ldb + p * 8;
        la[i] = (unsigned)(p >> 2) * GB_PLANE + (unsigned)(r + RPP * i) * GB_ROWB + (p & 3) * 16;
    }
    u32x4 sa[PASS], sb[PASS];
#define GB_LOAD(kc)                                                                                  \
    {                                                                                                \
        const int64_t ko = (int64_t)(kc) * (2 * GB_BK);                                              \
        _Pragma("unroll") for (int i = 0; i < PASS; ++i) {                                           \
            sa[i] = *reinterpret_cast<const u32x4 *>(A + ga[i] + ko);                                \
            sb[i] = *reinterpret_cast<const u32x4 *>(B + gb[i] + ko);                                \
        }                                                                                            \
    }
    f32x16 acc[2][2], acx[2][2];             // acx: the scaled cross terms of fp16x3 (unused otherwise)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[i][j][v] = 0.f; acx[i][j][v] = 0.f; }

    // fragment addresses: lane -> (row lane & 31, k group lane >> 5) of a 32-row tile; planes Ah | Al | Bh | Bl
    const unsigned fa = (unsigned)(wm * 64 + (lane & 31)) * GB_ROWB + (lane >> 5) * 16;
    const unsigned fb = (unsigned)(wn * 64 + (lane & 31)) * GB_ROWB + (lane >> 5) * 16;
    const int nk = K / GB_BK;
    GB_LOAD(0)
    for (int kc = 0; kc < nk; ++kc) {
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            *reinterpret_cast<u32x4 *>(lds + la[i]) = sa[i];
            *reinterpret_cast<u32x4 *>(lds + 2 * GB_PLANE + la[i]) = sb[i];
        }
        __syncthreads();
        if (kc + 1 < nk) GB_LOAD(kc + 1)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], bh[2], al[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8 *>(lds + fa + i * 32 * GB_ROWB + ks * 32);
                bh[i] = *reinterpret_cast<const bf16x8 *>(lds + 2 * GB_PLANE + fb + i * 32 * GB_ROWB + ks * 32);
                if (SPLIT) {
                    al[i] = *reinterpret_cast<const bf16x8 *>(lds + GB_PLANE + fa + i * 32 * GB_ROWB + ks * 32);
                    bl[i] = *reinterpret_cast<const bf16x8 *>(lds + 3 * GB_PLANE + fb + i * 32 * GB_ROWB + ks * 32);
                }
            }
#define GB_H(v) __builtin_bit_cast(f16x8, (v))
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (F16) {
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(GB_H(al[i]), GB_H(bh[j]), acx[i][j], 0, 0, 0);
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(GB_H(ah[i]), GB_H(bl[j]), acx[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(GB_H(ah[i]), GB_H(bh[j]), acc[i][j], 0, 0, 0);
                    } else {
                        if (TERMS == 3) {       // small terms first
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                }
#undef GB_H
        }
        __syncthreads();
    }
#undef GB_LOAD
    // C/D layout of the 32 x 32 tile: col = lane & 31, row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.f;
            const float unscale = F16 ? *inv_sa * *inv_sb : 1.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t row = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                const float r = F16 ? (acc[i][j][v] + acx[i][j][v] * (1.f / 2048.f)) * unscale : acc[i][j][v];
                if (row < M) C[row * ldc + col] = apply_act(r + bv, act);
            }
        }
}

}  // namespace itr

extern "C" int itr_split_bf16(const float *x, uint16_t *out, int64_t rows, int64_t K, itr_stream_t stream) {
    ITR_REQUIRE(rows >= 0 && K >= 32 && K % 32 == 0, "itr_split_bf16: K must be a positive multiple of 32");
    if (rows == 0) return ITR_OK;
    ITR_REQUIRE(x && out, "itr_split_bf16: null pointer");
    ITR_REQUIRE(itr::ceil_div(rows * K, (int64_t)256) <= 0x7fffffff, "itr_split_bf16: too many elements for one call");
    hipLaunchKernelGGL(itr::split_bf16_kernel, dim3((unsigned)itr::ceil_div(rows * K, (int64_t)256)), dim3(256), 0, itr::as_stream(stream),
                       x, out, rows * K);
    ITR_CHECK_LAUNCH("split_bf16");
    return ITR_OK;
}

extern "C" int itr_gemm_nt_bf16(const uint16_t *A, int64_t lda, const uint16_t *B, int64_t ldb, const float *bias, float *C,
                                int64_t ldc, int64_t M, int64_t N, int64_t K, int act, int terms, itr_stream_t stream) {
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 32 && K % 32 == 0 && K <= 0x3fffffff, "itr_gemm_nt_bf16: K must be a positive multiple of 32");
    ITR_REQUIRE(terms == 1 || terms == 3, "itr_gemm_nt_bf16: terms must be 1 (bf16) or 3 (bf16x3)");
    if (M == 0 || N == 0) return ITR_OK;
    ITR_REQUIRE(A && B && C, "itr_gemm_nt_bf16: null pointer");
    ITR_REQUIRE(ldc >= N && lda >= 64 && ldb >= 64 && lda % 64 == 0 && ldb % 64 == 0,
                "itr_gemm_nt_bf16: ldc >= N; lda / ldb count interleaved bf16 elements (2 x the fp32 stride), multiples of 64");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_gemm_nt_bf16: unknown activation");
    ITR_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0, "itr_gemm_nt_bf16: operands must be 16-byte aligned");
    const int64_t tm = itr::ceil_div(M, (int64_t)itr::GB_T), tn = itr::ceil_div(N, (int64_t)itr::GB_T);
    ITR_REQUIRE(tm * tn <= 0x7fffffff, "itr_gemm_nt_bf16: too many tiles");
    const dim3 grid((unsigned)(tm * tn));
    if (terms == 3)
        hipLaunchKernelGGL(itr::gemm_bf16_kernel<3>, grid, dim3(256), 0, itr::as_stream(stream), A, B, bias, C, ldc, M, N, (int)K, lda, ldb,
                           act, (int)tm, (int)tn, (const float *)nullptr, (const float *)nullptr);
    else
        hipLaunchKernelGGL(itr::gemm_bf16_kernel<1>, grid, dim3(256), 0, itr::as_stream(stream), A, B, bias, C, ldc, M, N, (int)K, lda, ldb,
                           act, (int)tm, (int)tn, (const float *)nullptr, (const float *)nullptr);
    ITR_CHECK_LAUNCH("gemm_bf16");
    return ITR_OK;
}

extern "C" int itr_split_f16(const float *x, uint16_t *out, float *scale_state, int64_t rows, int64_t K, itr_stream_t stream) {
    ITR_REQUIRE(rows >= 0 && K >= 32 && K % 32 == 0, "itr_split_f16: K must be a positive multiple of 32");
    ITR_REQUIRE(scale_state, "itr_split_f16: scale_state (2 floats: absmax bits, 1 / scale) missing");
    if (rows == 0) return ITR_OK;
    ITR_REQUIRE(x && out, "itr_split_f16: null pointer");
    const int64_t n = rows * K;
    ITR_REQUIRE(itr::ceil_div(n, (int64_t)256) <= 0x7fffffff, "itr_split_f16: too many elements for one call");
    ITR_CHECK_HIP(hipMemsetAsync(scale_state, 0, 4, itr::as_stream(stream)));
    const int64_t gb = itr::ceil_div(n, (int64_t)256);
    hipLaunchKernelGGL(itr::absmax_kernel, dim3((unsigned)(gb < 1024 ? gb : 1024)), dim3(256), 0, itr::as_stream(stream), x, n,
                       reinterpret_cast<unsigned *>(scale_state));
    hipLaunchKernelGGL(itr::split_f16_kernel, dim3((unsigned)gb), dim3(256), 0, itr::as_stream(stream), x, out, n,
                       reinterpret_cast<const unsigned *>(scale_state), scale_state + 1);
    ITR_CHECK_LAUNCH("split_f16");
    return ITR_OK;
}

extern "C" int itr_gemm_nt_f16x3(const uint16_t *A, int64_t lda, const float *scale_state_a, const uint16_t *B, int64_t ldb,
                                 const float *scale_state_b, const float *bias, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                                 itr_stream_t stream) {
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 32 && K % 32 == 0 && K <= 0x3fffffff, "itr_gemm_nt_f16x3: K must be a positive multiple of 32");
    if (M == 0 || N == 0) return ITR_OK;
    ITR_REQUIRE(A && B && C && scale_state_a && scale_state_b, "itr_gemm_nt_f16x3: null pointer");
    ITR_REQUIRE(ldc >= N && lda >= 64 && ldb >= 64 && lda % 64 == 0 && ldb % 64 == 0,
                "itr_gemm_nt_f16x3: ldc >= N; lda / ldb count interleaved fp16 elements (2 x the fp32 stride), multiples of 64");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_gemm_nt_f16x3: unknown activation");
    ITR_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0, "itr_gemm_nt_f16x3: operands must be 16-byte aligned");
    const int64_t tm = itr::ceil_div(M, (int64_t)itr::GB_T), tn = itr::ceil_div(N, (int64_t)itr::GB_T);
    ITR_REQUIRE(tm * tn <= 0x7fffffff, "itr_gemm_nt_f16x3: too many tiles");
    hipLaunchKernelGGL(itr::gemm_bf16_kernel<4>, dim3((unsigned)(tm * tn)), dim3(256), 0, itr::as_stream(stream), A, B, bias, C, ldc, M, N, (int)K,
                       lda, ldb, act, (int)tm, (int)tn, scale_state_a + 1, scale_state_b + 1);
    ITR_CHECK_LAUNCH("gemm_f16x3");
    return ITR_OK;
}
