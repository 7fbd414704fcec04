// Training-step primitives of VSRN's captioning branch (itr/modalmodule/Fusionmodule.py:10-367, Objectives.py:138-158): the decoder
// is a per-step loop (attention over the 36 encoder outputs -> GRU cell -> vocabulary projection -> log-softmax / NLL), so the pieces
// are single-step kernels; the dense layers are gemm_nt_kernel.
//   itr_gru_cell_fwd / _bwd     one nn.GRU step from the two projections gi = W_ih x + b_ih, gh = W_hh h + b_hh (gate order r, z, n):
//                               r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h
//   itr_nll_logsoftmax_fwd/_bwd F.log_softmax + nn.NLLLoss(reduce=False) * mask per row: loss[b] = -mask[b] log_softmax(logits[b])[target[b]]
#include "itr_common.h"

namespace itr {

__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float *__restrict__ gi, const float *__restrict__ gh, const float *__restrict__ h,
                                                           float *__restrict__ hn, float *__restrict__ gates, int64_t B, int H) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int64_t b = i / H;
    const int j = (int)(i - b * H);
    const float *a = gi + b * 3 * H, *c = gh + b * 3 * H;
    const float r = 1.f / (1.f + expf(-(a[j] + c[j])));
    const float z = 1.f / (1.f + expf(-(a[H + j] + c[H + j])));
    const float n = tanhf(a[2 * H + j] + r * c[2 * H + j]);
    hn[i] = (1.f - z) * n + z * h[i];
    gates[b * 3 * H + j] = r;
    gates[b * 3 * H + H + j] = z;
    gates[b * 3 * H + 2 * H + j] = n;
}

__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float *__restrict__ dhn, const float *__restrict__ gates,
                                                           const float *__restrict__ gh, const float *__restrict__ h, float *__restrict__ dgi,
                                                           float *__restrict__ dgh, float *__restrict__ dh, int64_t B, int H) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int64_t b = i / H;
    const int j = (int)(i - b * H);
    const float r = gates[b * 3 * H + j], z = gates[b * 3 * H + H + j], n = gates[b * 3 * H + 2 * H + j];
    const float g = dhn[i];
    const float dn_pre = g * (1.f - z) * (1.f - n * n);
    const float dz_pre = g * (h[i] - n) * z * (1.f - z);
    const float dr_pre = dn_pre * gh[b * 3 * H + 2 * H + j] * r * (1.f - r);
    dgi[b * 3 * H + j] = dr_pre; dgi[b * 3 * H + H + j] = dz_pre; dgi[b * 3 * H + 2 * H + j] = dn_pre;
    dgh[b * 3 * H + j] = dr_pre; dgh[b * 3 * H + H + j] = dz_pre; dgh[b * 3 * H + 2 * H + j] = dn_pre * r;
    dh[i] = g * z;
}

// one workgroup per row
__global__ __launch_bounds__(256) void nll_logsoftmax_fwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ target,
                                                                 const float *__restrict__ mask, float *__restrict__ loss, float *__restrict__ lse,
                                                                 int V) {
    __shared__ float red[4];
    const int64_t b = blockIdx.x;
    const float *x = logits + b * V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, x[v]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) s += expf(x[v] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l = m + logf(red[0] + red[1] + red[2] + red[3]);
        lse[b] = l;
        const int64_t t = target[b];
        loss[b] = (t >= 0 && t < V) ? -(x[t] - l) * mask[b] : 0.f;
    }
}
__global__ __launch_bounds__(256) void nll_logsoftmax_bwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ target,
                                                                 const float *__restrict__ mask, const float *__restrict__ lse,
                                                                 const float *__restrict__ dloss, float *__restrict__ dlogits, int V) {
    const int64_t b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float g = dloss[b] * mask[b];
    dlogits[b * V + v] = g * (expf(logits[b * V + v] - lse[b]) - (v == target[b] ? 1.f : 0.f));
}

// y[b, n, :] = act(x[b, n, :] + v[b, :]): the decoder attention's first layer, linear1(cat(enc_out, h)) = enc_out W_e^T + h W_h^T + b
// (Fusionmodule.py:136-140), with the encoder half computed ONCE per caption batch instead of once per decoder step (it does not
// depend on the step); backward from the output: d pre = dy act'(y), dx = d pre, dv[b] = sum_n d pre[b, n].
__global__ __launch_bounds__(256) void add_bcast_mid_act_kernel(const float *__restrict__ x, const float *__restrict__ v, float *__restrict__ y,
                                                                int N, int H, int act) {
    const int64_t b = blockIdx.y;
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    const float vv = v[b * H + h];
    for (int n = 0; n < N; ++n) {
        const int64_t o = (b * N + n) * (int64_t)H + h;
        y[o] = apply_act(x[o] + vv, act);
    }
}
__global__ __launch_bounds__(256) void add_bcast_mid_act_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy, float *__restrict__ dx,
                                                                    float *__restrict__ dv, int N, int H, int act) {
    const int64_t b = blockIdx.y;
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
        const int64_t o = (b * N + n) * (int64_t)H + h;
        const float yv = y[o];
        const float d = act == 1 ? (yv > 0.f ? 1.f : 0.f) : act == 2 ? 1.f - yv * yv : act == 3 ? yv * (1.f - yv) : 1.f;
        const float g = dy[o] * d;
        dx[o] = g;
        acc += g;
    }
    dv[b * H + h] = acc;
}

// The decoder attention's scores in one pass (Attention.forward, Fusionmodule.py:136-141: linear2(tanh(linear1(cat(enc, hidden))))):
//   e[b, n] = sum_h w[h] tanh(x[b, n, h] + v[b, h])        x = the encoder half of linear1 (+ bias), v = the hidden half, w = linear2's row.
// One wave per (b, n) row.  As three launches (broadcast add + tanh, a 4 608 x 1 "GEMM" over K slices, the slice sum) the tanh tensor
// crossed HBM twice per decoder step and the backward ran a K = 1 product on the unaligned tile kernel (round 6: 43 + 74 us per step).
__global__ __launch_bounds__(256) void addattn_score_fwd_kernel(const float *__restrict__ x, const float *__restrict__ v, const float *__restrict__ w,
                                                                float *__restrict__ e, int64_t rows, int N, int H) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *xr = x + row * H, *vr = v + (row / N) * H;
    float acc = 0.f;
    if ((H & 3) == 0) {
        for (int h4 = lane; h4 < H / 4; h4 += 64) {
            const float4 xv = reinterpret_cast<const float4 *>(xr)[h4], vv = reinterpret_cast<const float4 *>(vr)[h4], ww = reinterpret_cast<const float4 *>(w)[h4];
            acc += ww.x * apply_act(xv.x + vv.x, 2) + ww.y * apply_act(xv.y + vv.y, 2) + ww.z * apply_act(xv.z + vv.z, 2) + ww.w * apply_act(xv.w + vv.w, 2);
        }
    } else {
        for (int h = lane; h < H; h += 64) acc += w[h] * apply_act(xr[h] + vr[h], 2);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) e[row] = acc;
}
// Backward from de[b, n] (the tanh is recomputed):  g = de w[h] (1 - p^2);  dx[b, n, h] = g;  dv[b, h] = sum_n g;
// dw_part[b, h] = sum_n de[b, n] p  (the caller sums dw_part over b: itr_colsum).
__global__ __launch_bounds__(256) void addattn_score_bwd_kernel(const float *__restrict__ x, const float *__restrict__ v, const float *__restrict__ w,
                                                                const float *__restrict__ de, float *__restrict__ dx, float *__restrict__ dv,
                                                                float *__restrict__ dw_part, int N, int H) {
    const int64_t b = blockIdx.y;
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    const float vv = v[b * H + h], wv = w[h];
    float accv = 0.f, accw = 0.f;
    for (int n = 0; n < N; ++n) {
        const int64_t o = (b * N + n) * (int64_t)H + h;
        const float pv = apply_act(x[o] + vv, 2), d = de[b * N + n];
        const float g = d * wv * (1.f - pv * pv);
        dx[o] = g;
        accv += g;
        accw += d * pv;
    }
    dv[b * H + h] = accv;
    dw_part[b * H + h] = accw;
}

}  // namespace itr

extern "C" int itr_gru_cell_fwd(const float *gi, const float *gh, const float *h, float *h_next, float *gates, int64_t B, int H,
                                itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && H >= 1, "itr_gru_cell_fwd: bad shape");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(gi && gh && h && h_next && gates, "itr_gru_cell_fwd: null pointer");
    hipLaunchKernelGGL(itr::gru_cell_fwd_kernel, dim3((unsigned)itr::ceil_div(B * H, (int64_t)256)), dim3(256), 0, itr::as_stream(stream), gi, gh, h,
                       h_next, gates, B, H);
    ITR_CHECK_LAUNCH("gru_cell_fwd");
    return ITR_OK;
}

extern "C" int itr_gru_cell_bwd(const float *dh_next, const float *gates, const float *gh, const float *h, float *dgi, float *dgh, float *dh,
                                int64_t B, int H, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && H >= 1, "itr_gru_cell_bwd: bad shape");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(dh_next && gates && gh && h && dgi && dgh && dh, "itr_gru_cell_bwd: null pointer");
    hipLaunchKernelGGL(itr::gru_cell_bwd_kernel, dim3((unsigned)itr::ceil_div(B * H, (int64_t)256)), dim3(256), 0, itr::as_stream(stream), dh_next,
                       gates, gh, h, dgi, dgh, dh, B, H);
    ITR_CHECK_LAUNCH("gru_cell_bwd");
    return ITR_OK;
}

extern "C" int itr_nll_logsoftmax_fwd(const float *logits, const int64_t *target, const float *mask, float *loss, float *lse, int64_t B, int V,
                                      itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 0x7fffffff && V >= 1, "itr_nll_logsoftmax_fwd: bad shape");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(logits && target && mask && loss && lse, "itr_nll_logsoftmax_fwd: null pointer");
    hipLaunchKernelGGL(itr::nll_logsoftmax_fwd_kernel, dim3((unsigned)B), dim3(256), 0, itr::as_stream(stream), logits, target, mask, loss, lse, V);
    ITR_CHECK_LAUNCH("nll_logsoftmax_fwd");
    return ITR_OK;
}

extern "C" int itr_nll_logsoftmax_bwd(const float *logits, const int64_t *target, const float *mask, const float *lse, const float *dloss,
                                      float *dlogits, int64_t B, int V, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && V >= 1, "itr_nll_logsoftmax_bwd: bad shape (at most 65535 rows)");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(logits && target && mask && lse && dloss && dlogits, "itr_nll_logsoftmax_bwd: null pointer");
    hipLaunchKernelGGL(itr::nll_logsoftmax_bwd_kernel, dim3((unsigned)itr::ceil_div(V, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), logits,
                       target, mask, lse, dloss, dlogits, V);
    ITR_CHECK_LAUNCH("nll_logsoftmax_bwd");
    return ITR_OK;
}

extern "C" int itr_add_bcast_mid_act(const float *x, const float *v, float *y, int64_t B, int N, int H, int act, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && N >= 1 && H >= 1 && act >= 0 && act <= 3, "itr_add_bcast_mid_act: bad shape or activation (0 none, 1 relu, 2 tanh, 3 sigmoid)");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(x && v && y, "itr_add_bcast_mid_act: null pointer");
    hipLaunchKernelGGL(itr::add_bcast_mid_act_kernel, dim3((unsigned)itr::ceil_div(H, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), x, v, y, N, H,
                       act);
    ITR_CHECK_LAUNCH("add_bcast_mid_act");
    return ITR_OK;
}

extern "C" int itr_add_bcast_mid_act_bwd(const float *y, const float *dy, float *dx, float *dv, int64_t B, int N, int H, int act, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && N >= 1 && H >= 1 && act >= 0 && act <= 3, "itr_add_bcast_mid_act_bwd: bad shape or activation");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(y && dy && dx && dv, "itr_add_bcast_mid_act_bwd: null pointer");
    hipLaunchKernelGGL(itr::add_bcast_mid_act_bwd_kernel, dim3((unsigned)itr::ceil_div(H, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), y, dy, dx,
                       dv, N, H, act);
    ITR_CHECK_LAUNCH("add_bcast_mid_act_bwd");
    return ITR_OK;
}

extern "C" int itr_addattn_score(const float *x, const float *v, const float *w, float *e, int64_t B, int N, int H, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && N >= 1 && H >= 1 && B * N <= 0x7fffffffLL * 4, "itr_addattn_score: bad shape");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(x && v && w && e, "itr_addattn_score: null pointer");
    ITR_REQUIRE(H % 4 != 0 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(w)) & 15) == 0,
                "itr_addattn_score: rows of a multiple of 4 floats must start on 16 bytes");
    hipLaunchKernelGGL(itr::addattn_score_fwd_kernel, dim3((unsigned)itr::ceil_div(B * N, (int64_t)4)), dim3(256), 0, itr::as_stream(stream), x, v, w, e,
                       B * N, N, H);
    ITR_CHECK_LAUNCH("addattn_score");
    return ITR_OK;
}

extern "C" int itr_addattn_score_bwd(const float *x, const float *v, const float *w, const float *de, float *dx, float *dv, float *dw_part, int64_t B,
                                     int N, int H, itr_stream_t stream) {
    ITR_REQUIRE(B >= 0 && B <= 65535 && N >= 1 && H >= 1, "itr_addattn_score_bwd: bad shape");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(x && v && w && de && dx && dv && dw_part, "itr_addattn_score_bwd: null pointer");
    hipLaunchKernelGGL(itr::addattn_score_bwd_kernel, dim3((unsigned)itr::ceil_div(H, 256), (unsigned)B), dim3(256), 0, itr::as_stream(stream), x, v, w, de,
                       dx, dv, dw_part, N, H);
    ITR_CHECK_LAUNCH("addattn_score_bwd");
    return ITR_OK;
}
