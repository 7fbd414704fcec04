// Definitions shared by the SCAN (scan_xattn.hip) and SGRAF (sgraf.hip) kernels.
#pragma once
#include "itr_common.h"

namespace itr {

constexpr int SC_R = 36;                 // regions per image (precomp bottom-up features)
constexpr int SC_IMGS = 4;               // images per workgroup
constexpr int SC_MT = SC_IMGS * SC_R;    // 144 rows = 9 x 16
constexpr int SC_MTILES = SC_MT / 16;    // 9
constexpr int SC_NT = ITR_SCAN_NT;       // 64 word columns = 4 waves x 16
constexpr int SC_ROWS = SC_MT + SC_NT + 16;  // 208 staged rows per K chunk + 16 dump rows (224 = 7*256/8)
constexpr int SC_BK = 32, SC_PLANES = 8;
constexpr int SC_THREADS = 256;
constexpr int SC_MAXCAP = 16;            // captions per column tile (planner guarantees it)
constexpr int SC_LDT = SC_MT + 4;        // 148: row stride of the parked block, stored TRANSPOSED [col][row]

// Per column tile: which captions it holds and where (built on the device by scan_pack_kernel).
struct alignas(16) ScanTileMeta {
    int32_t ncap;
    int32_t cap_id[SC_MAXCAP];
    int32_t cap_start[SC_MAXCAP + 1];
    int8_t col_cap[SC_NT];       // caption slot of each column, -1 = padding
    int32_t far;                 // 1: some caption spans three or more 16-column blocks (its Gram block reaches beyond the neighbouring blocks)
    int32_t pad_[64 - 2 - SC_MAXCAP - (SC_MAXCAP + 1) - SC_NT / 4];
};
static_assert(sizeof(ScanTileMeta) == 256, "one 256-byte record per tile");

// LeakyReLU(0.1) = max(v, 0.1 v).  The maximum is ONE v_max_f32: fmaxf() makes hipcc canonicalise its operands first (an extra
// v_max_f32 v, v per element, for signalling NaNs), and in the epilogues every vector instruction waits for an issue slot next
// to the co-resident workgroup's MFMA stream.
__device__ __forceinline__ float leaky(float v) {
    float r;
    const float t = 0.1f * v;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(t));
    return r;
}
// exp via v_exp_f32 (2^x): 2 VALU instructions instead of ~12; relative error ~|x| * 1e-7, far inside the
// parity budget for the softmax / LogSumExp arguments here (|x| <= ~10).
__device__ __forceinline__ float fast_exp(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896341f); }
// single-instruction log / sqrt / reciprocal (v_log_f32, v_sqrt_f32, v_rcp_f32: 1 ulp) for the epilogue, where every
// instruction of the wave waits for a slot next to the other workgroup's MFMA stream; the libm forms expand to 10-20
// instructions each.  Relative error ~1e-7, far inside the 2e-5 parity budget of the scores.
__device__ __forceinline__ float fast_log(float v) { return __builtin_amdgcn_logf(v) * 0.693147180559945309f; }
__device__ __forceinline__ float fast_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }

// v[l] (+ | max) v[l ^ 16] and v[l] (+ | max) v[l ^ 32] in registers (gfx950 v_permlane16_swap / v_permlane32_swap): two vector
// instructions instead of a ds_bpermute round trip through the LDS crossbar, which an epilogue wave shares with the co-resident
// workgroup's main loop.  The operation is commutative, so the result is bit-identical to `v + __shfl_xor(v, 16 | 32)`.
__device__ __forceinline__ float xor16_add(float v) {
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float xor32_add(float v) {
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

#define SC_TICK(slot)                                                                              \
    if (g.dbg_cycles && tid == 0) {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                              \
        atomicAdd(&g.dbg_cycles[slot], now_ - tick_);                                              \
        tick_ = now_;                                                                              \
    }


}  // namespace itr
