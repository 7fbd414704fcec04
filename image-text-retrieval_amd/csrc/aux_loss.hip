// Training-only auxiliary losses of SAEM and CAMERA -- latency-bound: a batch is 64 .. 128 rows.
//   itr_angular_fwd / _bwd     AngularLoss.angular_loss on the three Gram-type products it consists of (Objectives.py:238-290)
//   itr_diversity_fwd / _bwd   DiversityRegularization of the summarisation matrices (Objectives.py:521-542)
#include "itr_common.h"

namespace itr {

// x[i][j] = c1 * (M1[i][j] + M2[i][j]) - c2 * Q[i][i]  for j != i     (M1 = anchors others^T, M2 = positives others^T,
// Q = anchors positives^T; c1 = 4 angle_bound, c2 = 2 (1 + angle_bound)).  One wave per row.
//   max_violation:  row = log(1 + exp(max_j x))                     stat[i] = max_j x, arg[i] = first j reaching it
//   otherwise:      row = t + log(exp(-t) + sum_j exp(x - t)), t = max_j x;   stat[i] = t, den[i] = exp(-t) + sum
__global__ __launch_bounds__(256) void angular_rows_kernel(const float *__restrict__ M1, const float *__restrict__ M2,
                                                           const float *__restrict__ Q, int n, float c1, float c2, int max_violation,
                                                           float *__restrict__ row, float *__restrict__ stat, float *__restrict__ den,
                                                           int32_t *__restrict__ arg) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float d = c2 * Q[(int64_t)i * n + i];
    const float *m1 = M1 + (int64_t)i * n, *m2 = M2 + (int64_t)i * n;
    float mx = -INFINITY;
    int am = -1;
    for (int j = lane; j < n; j += 64)
        if (j != i) {
            const float x = c1 * (m1[j] + m2[j]) - d;
            if (x > mx) { mx = x; am = j; }
        }
    const float t = wave_max(mx);
    if (max_violation) {
        // first index reaching the maximum (torch.max's choice on a row)
        int cand = (mx == t && am >= 0) ? am : 0x7fffffff;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(cand, o, 64); cand = other < cand ? other : cand; }
        if (lane == 0) {
            row[i] = logf(1.f + expf(t));
            stat[i] = t;
            arg[i] = cand;
        }
        return;
    }
    float s = 0.f;
    for (int j = lane; j < n; j += 64)
        if (j != i) s += expf(c1 * (m1[j] + m2[j]) - d - t);
    s = wave_sum(s) + expf(-t);
    if (lane == 0) {
        row[i] = t + logf(s);
        stat[i] = t;
        den[i] = s;
    }
}

// fixed-order sum of the rows (one block) -> loss[0] = scale * sum
__global__ __launch_bounds__(256) void sum_rows_kernel(const float *__restrict__ row, int64_t n, float scale, float *__restrict__ loss) {
    __shared__ float part[256];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = part[0] * scale;
}

// dM[i][j] = g * c1 * w_ij (the gradient of BOTH M1 and M2), dQ[i][i] = -g * c2 * sum_j w_ij, zero elsewhere;
// w = sigmoid(max) one-hot at arg (max_violation) or exp(x - t) / den / n.
__global__ __launch_bounds__(256) void angular_bwd_kernel(const float *__restrict__ M1, const float *__restrict__ M2,
                                                          const float *__restrict__ Q, int n, float c1, float c2, int max_violation,
                                                          const float *__restrict__ stat, const float *__restrict__ den,
                                                          const int32_t *__restrict__ arg, const float *__restrict__ gloss,
                                                          float *__restrict__ dM, float *__restrict__ dQ) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float g = gloss[0];
    float *dm = dM + (int64_t)i * n, *dq = dQ + (int64_t)i * n;
    if (max_violation) {
        const float t = stat[i];
        const float w = 1.f / (1.f + expf(-t));                  // d log(1 + e^t) / dt
        const int a = arg[i];
        for (int j = lane; j < n; j += 64) {
            dm[j] = (j == a) ? g * c1 * w : 0.f;
            dq[j] = (j == i) ? -g * c2 * w : 0.f;
        }
        return;
    }
    const float d = c2 * Q[(int64_t)i * n + i];
    const float *m1 = M1 + (int64_t)i * n, *m2 = M2 + (int64_t)i * n;
    const float t = stat[i], inv = 1.f / (den[i] * (float)n);
    float tot = 0.f;
    for (int j = lane; j < n; j += 64) {
        float w = 0.f;
        if (j != i) w = expf(c1 * (m1[j] + m2[j]) - d - t) * inv;
        dm[j] = g * c1 * w;
        tot += w;
    }
    tot = wave_sum(tot);
    for (int j = lane; j < n; j += 64) dq[j] = (j == i) ? -g * c2 * tot : 0.f;
}

// One block per image: S [R, K] -> column norms, Sn = S / max(norm, eps) (F.normalize(dim=1)), G = Sn^T Sn, sum (G - I)^2.
// LDS: Sn [R*K] + G [K*K] + norms [K].
template <bool BWD>
__global__ __launch_bounds__(256) void diversity_kernel(const float *__restrict__ S, int R, int K, float eps, float *__restrict__ part,
                                                        const float *__restrict__ gloss, float *__restrict__ dS) {
    extern __shared__ float lds[];
    float *sn = lds, *G = lds + R * K, *nrm = G + K * K;
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const float *s = S + (int64_t)blockIdx.x * R * K;
    for (int k = tid; k < K; k += 256) {
        float q = 0.f;
        for (int r = 0; r < R; ++r) { const float v = s[r * K + k]; q += v * v; }
        nrm[k] = fmaxf(sqrtf(q), eps);
    }
    __syncthreads();
    for (int e = tid; e < R * K; e += 256) sn[e] = s[e] / nrm[e % K];
    __syncthreads();
    float loc = 0.f;
    for (int e = tid; e < K * K; e += 256) {
        const int a = e / K, b = e - a * K;
        float q = 0.f;
        for (int r = 0; r < R; ++r) q += sn[r * K + a] * sn[r * K + b];
        q -= (a == b) ? 1.f : 0.f;
        G[e] = q;
        loc += q * q;
    }
    if (!BWD) {
        red[tid] = loc;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) part[blockIdx.x] = red[0];
        return;
    }
    __syncthreads();
    // dSn = Sn (dG + dG^T) = 4 Sn (G - I)   (G symmetric), then through the column normalisation
    const float g = gloss[0];
    float *dsn = dS + (int64_t)blockIdx.x * R * K;                 // first dSn in place, then dS
    for (int e = tid; e < R * K; e += 256) {
        const int r = e / K, k = e - r * K;
        float q = 0.f;
        for (int b = 0; b < K; ++b) q += sn[r * K + b] * G[b * K + k];
        dsn[e] = 4.f * g * q;
    }
    __syncthreads();
    // columns whose norm was clamped to eps have a constant divisor: no projection term
    for (int k = tid; k < K; k += 256) {
        float q = 0.f, raw = 0.f;
        for (int r = 0; r < R; ++r) { q += sn[r * K + k] * dsn[r * K + k]; const float v = s[r * K + k]; raw += v * v; }
        G[k] = (sqrtf(raw) > eps) ? q : 0.f;                      // reuse G[0 .. K) as the per-column dot
    }
    __syncthreads();
    for (int e = tid; e < R * K; e += 256) {
        const int k = e % K;
        dsn[e] = (dsn[e] - sn[e] * G[k]) / nrm[k];
    }
}

}  // namespace itr

extern "C" int itr_angular_fwd(const float *M1, const float *M2, const float *Q, int n, float angle_bound, int max_violation, float *loss,
                               float *row, float *stat, float *den, int32_t *arg, itr_stream_t stream) {
    ITR_REQUIRE(n >= 2, "itr_angular_fwd: needs at least two rows (every anchor needs a negative)");
    ITR_REQUIRE(M1 && M2 && Q && loss && row && stat && den && arg, "itr_angular_fwd: null pointer");
    const float c1 = 4.f * angle_bound, c2 = 2.f * (1.f + angle_bound);
    hipLaunchKernelGGL(itr::angular_rows_kernel, dim3((unsigned)itr::ceil_div(n, 4)), dim3(256), 0, itr::as_stream(stream), M1, M2, Q, n, c1, c2,
                       max_violation, row, stat, den, arg);
    hipLaunchKernelGGL(itr::sum_rows_kernel, dim3(1), dim3(256), 0, itr::as_stream(stream), (const float *)row, (int64_t)n,
                       max_violation ? 1.f : 1.f / (float)n, loss);
    ITR_CHECK_LAUNCH("angular_fwd");
    return ITR_OK;
}

extern "C" int itr_angular_bwd(const float *M1, const float *M2, const float *Q, int n, float angle_bound, int max_violation,
                               const float *stat, const float *den, const int32_t *arg, const float *grad_loss, float *dM, float *dQ,
                               itr_stream_t stream) {
    ITR_REQUIRE(n >= 2, "itr_angular_bwd: needs at least two rows");
    ITR_REQUIRE(M1 && M2 && Q && stat && den && arg && grad_loss && dM && dQ, "itr_angular_bwd: null pointer");
    const float c1 = 4.f * angle_bound, c2 = 2.f * (1.f + angle_bound);
    hipLaunchKernelGGL(itr::angular_bwd_kernel, dim3((unsigned)itr::ceil_div(n, 4)), dim3(256), 0, itr::as_stream(stream), M1, M2, Q, n, c1, c2,
                       max_violation, stat, den, arg, grad_loss, dM, dQ);
    ITR_CHECK_LAUNCH("angular_bwd");
    return ITR_OK;
}

static bool diversity_fits(int R, int K) { return R >= 1 && K >= 1 && ((size_t)R * K + (size_t)K * K + K) * 4 <= 60 * 1024; }

extern "C" int itr_diversity_fwd(const float *smry, int64_t B, int R, int K, float *part, float *loss, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && diversity_fits(R, K), "itr_diversity_fwd: bad shape (at most 65535 images, R*K + K*K + K <= 15360)");
    ITR_REQUIRE(loss && (B == 0 || (smry && part)), "itr_diversity_fwd: null pointer");
    const size_t lds = ((size_t)R * K + (size_t)K * K + K) * 4;
    if (B > 0)
        hipLaunchKernelGGL(itr::diversity_kernel<false>, dim3((unsigned)B), dim3(256), lds, itr::as_stream(stream), smry, R, K, 1e-12f, part,
                           (const float *)nullptr, (float *)nullptr);
    hipLaunchKernelGGL(itr::sum_rows_kernel, dim3(1), dim3(256), 0, itr::as_stream(stream), (const float *)part, B, 1.f, loss);
    ITR_CHECK_LAUNCH("diversity_fwd");
    return ITR_OK;
}

extern "C" int itr_diversity_bwd(const float *smry, int64_t B, int R, int K, const float *grad_loss, float *d_smry, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && diversity_fits(R, K), "itr_diversity_bwd: bad shape (at most 65535 images, R*K + K*K + K <= 15360)");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(smry && grad_loss && d_smry, "itr_diversity_bwd: null pointer");
    const size_t lds = ((size_t)R * K + (size_t)K * K + K) * 4;
    hipLaunchKernelGGL(itr::diversity_kernel<true>, dim3((unsigned)B), dim3(256), lds, itr::as_stream(stream), smry, R, K, 1e-12f, (float *)nullptr,
                       grad_loss, d_smry);
    ITR_CHECK_LAUNCH("diversity_bwd");
    return ITR_OK;
}
