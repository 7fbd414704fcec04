// Split-bf16 GEMM study: C[M,N] = A[M,K] B[N,K]^T (+ bias) on the bf16 matrix core with fp32 operands
// split into bf16 planes,  x = hi + lo + O(2^-17 |x|):
//   TERMS = 3   hi.hi + hi.lo + lo.hi   ("bf16x3": dropped terms ~ 2^-17 per product, fp32 accumulation)
//   TERMS = 1   hi.hi                   (plain bf16 inputs, fp32 accumulation)
// v_mfma_f32_32x32x16_bf16 runs 16x the fp32 MFMA rate, so bf16x3 has a ceiling of 2.5 PF / 3 = 833 TFLOP/s of
// fp32-equivalent work against 157 for v_mfma_f32_32x32x2_f32.  NOT wired into any default path: the product GEMM
// (gemm_f32.hip) is exact fp32; this kernel is reported separately and opt-in.
//
// Tiling: 128 x 128 per workgroup, 4 waves as 2 x 2, each wave 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs);
// K chunks of 32 staged through LDS (row stride 80 B: 16 consecutive rows hit 16 different 16-byte bank groups), the next
// chunk's global loads are in flight in registers while the current one is multiplied.
#include "itr_common.h"

namespace itr {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int GB_T = 128, GB_BK = 32, GB_ROWB = 80, GB_PLANE = GB_T * GB_ROWB;   // 10 240 B per operand plane

__global__ __launch_bounds__(256) void split_bf16_kernel(const float *__restrict__ x, uint16_t *__restrict__ hi,
                                                         uint16_t *__restrict__ lo, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    auto rne = [](float v) -> uint32_t {       // fp32 -> bf16 bits, round to nearest even (finite inputs)
        const uint32_t u = __float_as_uint(v);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    const float v = x[i];
    const uint32_t h = rne(v);
    hi[i] = (uint16_t)h;
    if (lo) lo[i] = (uint16_t)rne(v - __uint_as_float(h << 16));
}

template <int TERMS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const uint16_t *__restrict__ Ah, const uint16_t *__restrict__ Al,
                                                           const uint16_t *__restrict__ Bh, const uint16_t *__restrict__ Bl,
                                                           const float *__restrict__ bias, float *__restrict__ C, int64_t ldc,
                                                           int64_t M, int64_t N, int K, int64_t lda, int64_t ldb, int act,
                                                           int tiles_m, int tiles_n) {
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
    const int r = tid >> 2, gq = tid & 3;
    const int64_t ra0 = min(m0 + r, M - 1) * lda + gq * 8, ra1 = min(m0 + r + 64, M - 1) * lda + gq * 8;
    const int64_t rb0 = min(n0 + r, N - 1) * ldb + gq * 8, rb1 = min(n0 + r + 64, N - 1) * ldb + gq * 8;
    const unsigned ls0 = (unsigned)r * GB_ROWB + gq * 16, ls1 = (unsigned)(r + 64) * GB_ROWB + gq * 16;
    uint4 sah0, sah1, sbh0, sbh1, sal0, sal1, sbl0, sbl1;
#define GB_LOAD(kc)                                                                                  \
    {                                                                                                \
        const int64_t ko = (int64_t)(kc) * GB_BK;                                                    \
        sah0 = *reinterpret_cast<const uint4 *>(Ah + ra0 + ko);                                      \
        sah1 = *reinterpret_cast<const uint4 *>(Ah + ra1 + ko);                                      \
        sbh0 = *reinterpret_cast<const uint4 *>(Bh + rb0 + ko);                                      \
        sbh1 = *reinterpret_cast<const uint4 *>(Bh + rb1 + ko);                                      \
        if (TERMS == 3) {                                                                            \
            sal0 = *reinterpret_cast<const uint4 *>(Al + ra0 + ko);                                  \
            sal1 = *reinterpret_cast<const uint4 *>(Al + ra1 + ko);                                  \
            sbl0 = *reinterpret_cast<const uint4 *>(Bl + rb0 + ko);                                  \
            sbl1 = *reinterpret_cast<const uint4 *>(Bl + rb1 + ko);                                  \
        }                                                                                            \
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    // fragment addresses: lane -> (row lane & 31, k group lane >> 5) of a 32-row tile; planes Ah | Al | Bh | Bl
    const unsigned fa = (unsigned)(wm * 64 + (lane & 31)) * GB_ROWB + (lane >> 5) * 16;
    const unsigned fb = (unsigned)(wn * 64 + (lane & 31)) * GB_ROWB + (lane >> 5) * 16;
    const int nk = K / GB_BK;
    GB_LOAD(0)
    for (int kc = 0; kc < nk; ++kc) {
        *reinterpret_cast<uint4 *>(lds + ls0) = sah0;
        *reinterpret_cast<uint4 *>(lds + ls1) = sah1;
        *reinterpret_cast<uint4 *>(lds + 2 * GB_PLANE + ls0) = sbh0;
        *reinterpret_cast<uint4 *>(lds + 2 * GB_PLANE + ls1) = sbh1;
        if (TERMS == 3) {
            *reinterpret_cast<uint4 *>(lds + GB_PLANE + ls0) = sal0;
            *reinterpret_cast<uint4 *>(lds + GB_PLANE + ls1) = sal1;
            *reinterpret_cast<uint4 *>(lds + 3 * GB_PLANE + ls0) = sbl0;
            *reinterpret_cast<uint4 *>(lds + 3 * GB_PLANE + ls1) = sbl1;
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
                if (TERMS == 3) {
                    al[i] = *reinterpret_cast<const bf16x8 *>(lds + GB_PLANE + fa + i * 32 * GB_ROWB + ks * 32);
                    bl[i] = *reinterpret_cast<const bf16x8 *>(lds + 3 * GB_PLANE + fb + i * 32 * GB_ROWB + ks * 32);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (TERMS == 3) {       // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
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
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t row = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                if (row < M) C[row * ldc + col] = apply_act(acc[i][j][v] + bv, act);
            }
        }
}

}  // namespace itr

extern "C" int itr_split_bf16(const float *x, uint16_t *hi, uint16_t *lo, int64_t n, itr_stream_t stream) {
    ITR_REQUIRE(n >= 0, "itr_split_bf16: bad size");
    if (n == 0) return ITR_OK;
    ITR_REQUIRE(x && hi, "itr_split_bf16: null pointer");
    ITR_REQUIRE(itr::ceil_div(n, (int64_t)256) <= 0x7fffffff, "itr_split_bf16: too many elements for one call");
    hipLaunchKernelGGL(itr::split_bf16_kernel, dim3((unsigned)itr::ceil_div(n, (int64_t)256)), dim3(256), 0, itr::as_stream(stream), x,
                       hi, lo, n);
    ITR_CHECK_LAUNCH("split_bf16");
    return ITR_OK;
}

extern "C" int itr_gemm_nt_bf16(const uint16_t *Ah, const uint16_t *Al, int64_t lda, const uint16_t *Bh, const uint16_t *Bl,
                                int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                                int terms, itr_stream_t stream) {
    ITR_REQUIRE(M >= 0 && N >= 0 && K >= 32 && K % 32 == 0 && K <= 0x7fffffff, "itr_gemm_nt_bf16: K must be a positive multiple of 32");
    ITR_REQUIRE(terms == 1 || terms == 3, "itr_gemm_nt_bf16: terms must be 1 (bf16) or 3 (bf16x3)");
    if (M == 0 || N == 0) return ITR_OK;
    ITR_REQUIRE(Ah && Bh && C && (terms == 1 || (Al && Bl)), "itr_gemm_nt_bf16: null pointer");
    ITR_REQUIRE(ldc >= N && lda >= 8 && ldb >= 8 && lda % 8 == 0 && ldb % 8 == 0, "itr_gemm_nt_bf16: ldc >= N, lda / ldb multiples of 8");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_gemm_nt_bf16: unknown activation");
    const uintptr_t al = reinterpret_cast<uintptr_t>(Ah) | reinterpret_cast<uintptr_t>(Bh) | reinterpret_cast<uintptr_t>(Al) |
                         reinterpret_cast<uintptr_t>(Bl);
    ITR_REQUIRE((al & 15) == 0, "itr_gemm_nt_bf16: operands must be 16-byte aligned");
    const int64_t tm = itr::ceil_div(M, (int64_t)itr::GB_T), tn = itr::ceil_div(N, (int64_t)itr::GB_T);
    ITR_REQUIRE(tm * tn <= 0x7fffffff, "itr_gemm_nt_bf16: too many tiles");
    const dim3 grid((unsigned)(tm * tn));
    if (terms == 3)
        hipLaunchKernelGGL(itr::gemm_bf16_kernel<3>, grid, dim3(256), 0, itr::as_stream(stream), Ah, Al, Bh, Bl, bias, C, ldc, M, N,
                           (int)K, lda, ldb, act, (int)tm, (int)tn);
    else
        hipLaunchKernelGGL(itr::gemm_bf16_kernel<1>, grid, dim3(256), 0, itr::as_stream(stream), Ah, Al, Bh, Bl, bias, C, ldc, M, N,
                           (int)K, lda, ldb, act, (int)tm, (int)tn);
    ITR_CHECK_LAUNCH("gemm_bf16");
    return ITR_OK;
}
