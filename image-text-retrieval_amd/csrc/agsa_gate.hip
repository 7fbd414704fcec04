// The query / key gate of CAMERA's gated self-attention (GatedQueryAttLayer.forward, camera_.py:36-44), fused:
//     G = fc_q(q) * fc_k(k)            per (position, head) row of d_k values
//     M = sigmoid(fc_g(G))             2 d_k values
//     q' = q * M[:d_k]   k' = k * M[d_k:]
// As three Linear layers with d_k = 32 inputs this was three GEMMs with N = 32 / 64 and K = 32 on 128 x 128 tiles (75 % of the
// matrix work on padding, one pipeline fill per tile) plus three elementwise kernels: 6 % of the CAMERA evaluation step for ~6
// GB of traffic.  Here one wave takes 16 rows at a time and never leaves its registers:
//   Q'^T = W_q q^T    A = W_q fragments (kept in registers for the whole kernel), B = the row fragments as they come from memory
//                     (lane (fi, fg): 16 bytes at feature 4 fg + 16 kk of row fi).  Accumulator j of out-tile mt =
//                     Q'[row fi][feature 16 mt + 4 fg + j].
//   G^T  = Q'^T * K'^T elementwise, and already the B operand of the next product (k slot fg <-> feature 16 mt + 4 fg + j)
//   M^T  = W_g G^T    accumulator j of out-tile ot = M[row fi][16 ot + 4 fg + j]: the SAME (row, four features) the lane loaded
//                     from q (ot < d_k / 16) and k, so the gated rows are two elementwise products and 16-byte stores.
// 64 MFMAs (16x16x4) per 16 rows at d_k = 32; the kernel is bound by its 4 x rows x d_k x 4 bytes of traffic.
#include "itr_common.h"

namespace itr {

template <int DK>
__global__ __launch_bounds__(256) void agsa_gate_kernel(const float *__restrict__ q, const float *__restrict__ k, int64_t rows,
                                                        const float *__restrict__ Wq, const float *__restrict__ bq,
                                                        const float *__restrict__ Wk, const float *__restrict__ bk,
                                                        const float *__restrict__ Wg, const float *__restrict__ bg,
                                                        float *__restrict__ qo, float *__restrict__ ko) {
    constexpr int NT = DK / 16;
    const int lane = threadIdx.x & 63;
    const int fi = lane & 15, fg = lane >> 4;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (rows + 15) / 16;
    float4 wq[NT][NT], wk[NT][NT], wg[2 * NT][NT], biq[NT], bik[NT], big[2 * NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
            wq[mt][kk] = *reinterpret_cast<const float4 *>(Wq + (16 * mt + fi) * DK + 16 * kk + 4 * fg);
            wk[mt][kk] = *reinterpret_cast<const float4 *>(Wk + (16 * mt + fi) * DK + 16 * kk + 4 * fg);
        }
        biq[mt] = *reinterpret_cast<const float4 *>(bq + 16 * mt + 4 * fg);
        bik[mt] = *reinterpret_cast<const float4 *>(bk + 16 * mt + 4 * fg);
    }
#pragma unroll
    for (int ot = 0; ot < 2 * NT; ++ot) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) wg[ot][mt] = *reinterpret_cast<const float4 *>(Wg + (16 * ot + fi) * DK + 16 * mt + 4 * fg);
        big[ot] = *reinterpret_cast<const float4 *>(bg + 16 * ot + 4 * fg);
    }
    for (int64_t t = wave0; t < ntiles; t += nwaves) {
        const int64_t row = t * 16 + fi;
        const int64_t rl = row < rows ? row : rows - 1;          // a ragged last tile re-reads the last row, its results are not stored
        float4 qf[NT], kf[NT];
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
            qf[kk] = *reinterpret_cast<const float4 *>(q + rl * DK + 16 * kk + 4 * fg);
            kf[kk] = *reinterpret_cast<const float4 *>(k + rl * DK + 16 * kk + 4 * fg);
        }
        f32x4 g[NT];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            f32x4 aq = f32x4{biq[mt].x, biq[mt].y, biq[mt].z, biq[mt].w}, ak = f32x4{bik[mt].x, bik[mt].y, bik[mt].z, bik[mt].w};
#pragma unroll
            for (int kk = 0; kk < NT; ++kk) {
                aq = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[mt][kk].x, qf[kk].x, aq, 0, 0, 0);
                ak = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[mt][kk].x, kf[kk].x, ak, 0, 0, 0);
                aq = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[mt][kk].y, qf[kk].y, aq, 0, 0, 0);
                ak = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[mt][kk].y, kf[kk].y, ak, 0, 0, 0);
                aq = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[mt][kk].z, qf[kk].z, aq, 0, 0, 0);
                ak = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[mt][kk].z, kf[kk].z, ak, 0, 0, 0);
                aq = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[mt][kk].w, qf[kk].w, aq, 0, 0, 0);
                ak = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[mt][kk].w, kf[kk].w, ak, 0, 0, 0);
            }
            g[mt] = f32x4{aq[0] * ak[0], aq[1] * ak[1], aq[2] * ak[2], aq[3] * ak[3]};
        }
#pragma unroll
        for (int ot = 0; ot < 2 * NT; ++ot) {
            f32x4 m = f32x4{big[ot].x, big[ot].y, big[ot].z, big[ot].w};
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                m = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[ot][mt].x, g[mt][0], m, 0, 0, 0);
                m = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[ot][mt].y, g[mt][1], m, 0, 0, 0);
                m = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[ot][mt].z, g[mt][2], m, 0, 0, 0);
                m = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[ot][mt].w, g[mt][3], m, 0, 0, 0);
            }
            const float4 src = ot < NT ? qf[ot] : kf[ot - NT];
            const float4 res = float4{src.x * apply_act(m[0], 3), src.y * apply_act(m[1], 3), src.z * apply_act(m[2], 3), src.w * apply_act(m[3], 3)};
            if (row < rows) {
                float *dst = (ot < NT ? qo : ko) + row * DK + 16 * (ot < NT ? ot : ot - NT) + 4 * fg;
                *reinterpret_cast<float4 *>(dst) = res;
            }
        }
    }
}

}  // namespace itr

extern "C" int itr_agsa_gate(const float *q, const float *k, int64_t rows, int dk, const float *Wq, const float *bq, const float *Wk,
                             const float *bk, const float *Wg, const float *bg, float *q_out, float *k_out, itr_stream_t stream) {
    ITR_REQUIRE(q && k && Wq && bq && Wk && bk && Wg && bg && q_out && k_out, "itr_agsa_gate: null pointer");
    ITR_REQUIRE(rows >= 0, "itr_agsa_gate: bad row count");
    ITR_UNSUPPORTED(dk != 16 && dk != 32, "itr_agsa_gate: head size 16 or 32 (got %d); compose the three Linear layers for other sizes", dk);
    const uintptr_t al = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(Wq) | reinterpret_cast<uintptr_t>(bq) |
                         reinterpret_cast<uintptr_t>(Wk) | reinterpret_cast<uintptr_t>(bk) | reinterpret_cast<uintptr_t>(Wg) | reinterpret_cast<uintptr_t>(bg) |
                         reinterpret_cast<uintptr_t>(q_out) | reinterpret_cast<uintptr_t>(k_out);
    ITR_REQUIRE((al & 15) == 0, "itr_agsa_gate: operands must be 16-byte aligned");
    if (rows == 0) return ITR_OK;
    const int64_t ntiles = (rows + 15) / 16;
    const int64_t want = (ntiles + 3) / 4;
    const unsigned grid = (unsigned)(want < 256 * 8 ? want : 256 * 8);          // persistent: the weight fragments are loaded once per wave
    hipStream_t st = itr::as_stream(stream);
    if (dk == 32) hipLaunchKernelGGL(itr::agsa_gate_kernel<32>, dim3(grid), dim3(256), 0, st, q, k, rows, Wq, bq, Wk, bk, Wg, bg, q_out, k_out);
    else hipLaunchKernelGGL(itr::agsa_gate_kernel<16>, dim3(grid), dim3(256), 0, st, q, k, rows, Wq, bq, Wk, bk, Wg, bg, q_out, k_out);
    ITR_CHECK_LAUNCH("agsa_gate");
    return ITR_OK;
}
