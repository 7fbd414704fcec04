// Bidirectional hinge ranking loss with hardest-negative mining
// (ContrastiveLoss.forward, itr/modalmodule/Objectives.py:93-115; TripletLoss.forward :492-517):
//     cost_s [i,j] = [m + S_ij - S_ii]_+   (caption retrieval, reduced over j)
//     cost_im[i,j] = [m + S_ij - S_jj]_+   (image retrieval,   reduced over i)
//     diagonal zeroed; loss = sum_i red_j cost_s + sum_j red_i cost_im, red = max | sum.
// B is a training batch (128): latency-bound.  One workgroup per row AND per column computes
// its reduction with wave shuffles (+arg for the backward); a single-wave kernel then sums the
// 2B partials in a fixed order, so the loss is deterministic (no float atomics).
// Backward: dS is written row by row from the saved arg-max indices -- +-1 at <= 4B entries for
// the max form -- again without atomics.
#include "itr_common.h"

namespace itr {

constexpr int HINGE_THREADS = 256;

struct ValIdx {
    float v;
    int i;
};

__device__ __forceinline__ ValIdx better(ValIdx a, ValIdx b) {
    // larger value wins; on equal values the LOWER index wins (torch.max returns the first)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// blockIdx.x < B : row i = blockIdx.x (cost_s) ; else column j = blockIdx.x - B (cost_im)
__global__ __launch_bounds__(HINGE_THREADS) void hinge_fwd_kernel(const float *__restrict__ S, int B, int64_t ldS,
                                                                  float margin, int max_violation,
                                                                  float *__restrict__ cost,  // [2B]
                                                                  int32_t *__restrict__ row_arg,
                                                                  int32_t *__restrict__ col_arg) {
    __shared__ float s_v[HINGE_THREADS / 64];
    __shared__ int s_i[HINGE_THREADS / 64];
    const bool is_row = blockIdx.x < (unsigned)B;
    const int q = is_row ? blockIdx.x : blockIdx.x - B;
    const float diag = S[(int64_t)q * ldS + q];
    float sum = 0.f;
    ValIdx best{0.f, 0x7fffffff};  // masked diagonal contributes cost 0
    for (int k = threadIdx.x; k < B; k += HINGE_THREADS) {
        const float s = is_row ? S[(int64_t)q * ldS + k] : S[(int64_t)k * ldS + q];
        float c = fmaxf(margin + s - diag, 0.f);
        if (k == q) c = 0.f;
        sum += c;
        best = better(best, ValIdx{c, k});
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    sum = wave_sum(sum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ValIdx other{__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)};
        best = better(best, other);
    }
    if (lane == 0) {
        s_v[wave] = max_violation ? best.v : sum;
        s_i[wave] = best.i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        ValIdx b{0.f, 0x7fffffff};
        for (int w = 0; w < HINGE_THREADS / 64; ++w) {
            total += s_v[w];
            b = better(b, ValIdx{s_v[w], s_i[w]});
        }
        cost[blockIdx.x] = max_violation ? b.v : total;
        int32_t *arg = is_row ? row_arg : col_arg;
        if (arg) arg[q] = b.i;
    }
}

__global__ void hinge_sum_kernel(const float *__restrict__ cost, int n, float *__restrict__ loss) {
    // fixed-order summation: lane-strided partials, then a wave tree
    float s = 0.f;
    for (int k = threadIdx.x; k < n; k += 64) s += cost[k];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s;
}

// one workgroup per row i of dS
__global__ __launch_bounds__(HINGE_THREADS) void hinge_bwd_kernel(const float *__restrict__ S, int B, int64_t ldS,
                                                                  float margin, int max_violation,
                                                                  const int32_t *__restrict__ row_arg,
                                                                  const int32_t *__restrict__ col_arg,
                                                                  const float *__restrict__ grad_loss,
                                                                  float *__restrict__ dS, int64_t lddS) {
    __shared__ int s_cnt[HINGE_THREADS / 64];
    const int i = blockIdx.x;
    const float g = grad_loss[0];
    const float dii = S[(int64_t)i * ldS + i];
    int diag_cnt = 0;  // number of active hinges that have S_ii as the positive
    if (max_violation) {
        const int ja = row_arg[i];
        const int ia = col_arg[i];
        // an arg equal to i (or out of range) means "no violating negative": cost 0, no gradient
        const bool row_act = ja != i && ja >= 0 && ja < B && (margin + S[(int64_t)i * ldS + ja] - dii > 0.f);
        const bool col_act = ia != i && ia >= 0 && ia < B && (margin + S[(int64_t)ia * ldS + i] - dii > 0.f);
        for (int j = threadIdx.x; j < B; j += HINGE_THREADS) {
            if (j == i) continue;
            float v = 0.f;
            if (row_act && j == ja) v += g;
            // column j's hardest image is i ?
            if (col_arg[j] == i) {
                const float c = margin + S[(int64_t)i * ldS + j] - S[(int64_t)j * ldS + j];
                if (c > 0.f) v += g;
            }
            dS[(int64_t)i * lddS + j] = v;
        }
        if (threadIdx.x == 0) dS[(int64_t)i * lddS + i] = -g * (float)((int)row_act + (int)col_act);
        return;
    }
    for (int j = threadIdx.x; j < B; j += HINGE_THREADS) {
        if (j == i) continue;
        const float sij = S[(int64_t)i * ldS + j];
        const int a = (margin + sij - dii > 0.f);                          // cost_s[i,j] active
        const int b = (margin + sij - S[(int64_t)j * ldS + j] > 0.f);      // cost_im[i,j] active
        const int c = (margin + S[(int64_t)j * ldS + i] - dii > 0.f);      // cost_im[j,i] active (column i)
        dS[(int64_t)i * lddS + j] = g * (float)(a + b);
        diag_cnt += a + c;
    }
    diag_cnt = wave_sum_i(diag_cnt);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = diag_cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < HINGE_THREADS / 64; ++w) t += s_cnt[w];
        dS[(int64_t)i * lddS + i] = -g * (float)t;
    }
}

}  // namespace itr

extern "C" int itr_hinge_maxviol_fwd(const float *S, int B, int64_t ldS, float margin, int max_violation,
                                     float *loss, int32_t *row_arg, int32_t *col_arg, float *cost_ws,
                                     itr_stream_t stream) {
    ITR_REQUIRE(S && loss && cost_ws, "itr_hinge_maxviol_fwd: null pointer");
    ITR_REQUIRE(B >= 1 && ldS >= B, "itr_hinge_maxviol_fwd: bad shape B=%d ldS=%lld", B, (long long)ldS);
    ITR_REQUIRE(!max_violation || (row_arg && col_arg),
                "itr_hinge_maxviol_fwd: row_arg/col_arg are required with max_violation");
    hipStream_t st = itr::as_stream(stream);
    hipLaunchKernelGGL(itr::hinge_fwd_kernel, dim3(2 * B), dim3(itr::HINGE_THREADS), 0, st, S, B, ldS, margin,
                       max_violation, cost_ws, row_arg, col_arg);
    ITR_CHECK_LAUNCH("hinge_fwd");
    hipLaunchKernelGGL(itr::hinge_sum_kernel, dim3(1), dim3(64), 0, st, cost_ws, 2 * B, loss);
    ITR_CHECK_LAUNCH("hinge_sum");
    return ITR_OK;
}

extern "C" int itr_hinge_maxviol_bwd(const float *S, int B, int64_t ldS, float margin, int max_violation,
                                     const int32_t *row_arg, const int32_t *col_arg, const float *grad_loss,
                                     float *dS, int64_t lddS, itr_stream_t stream) {
    ITR_REQUIRE(S && grad_loss && dS, "itr_hinge_maxviol_bwd: null pointer");
    ITR_REQUIRE(B >= 1 && ldS >= B && lddS >= B, "itr_hinge_maxviol_bwd: bad shape");
    ITR_REQUIRE(!max_violation || (row_arg && col_arg),
                "itr_hinge_maxviol_bwd: row_arg/col_arg are required with max_violation");
    hipLaunchKernelGGL(itr::hinge_bwd_kernel, dim3(B), dim3(itr::HINGE_THREADS), 0, itr::as_stream(stream), S, B,
                       ldS, margin, max_violation, row_arg, col_arg, grad_loss, dS, lddS);
    ITR_CHECK_LAUNCH("hinge_bwd");
    return ITR_OK;
}
