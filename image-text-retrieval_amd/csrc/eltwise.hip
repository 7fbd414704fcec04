// Small HBM-bound helpers of the CAMERA towers (itr/modalmodule/camera_.py, ImgEncoder.py:355-433,
// TextEncoder.py:162-197): gating products, eval-mode BatchNorm / residual epilogues, the box position
// features and the multi-view summarisation.
#include "itr_common.h"

namespace itr {

// out[r, c] = a[r, c] * b[r * ldb + c]
__global__ __launch_bounds__(256) void mul_rows_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t ldb,
                                                       float *__restrict__ out, int64_t n, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    out[i] = a[i] * b[r * ldb + c];
}

// out = act(x * scale[c] + shift[c]) (+ residual): eval-mode BatchNorm1d folded to an affine map per column
__global__ __launch_bounds__(256) void affine_cols_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, const float *__restrict__ res,
                                                          float *__restrict__ out, int64_t n, int C, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    float v = x[i];
    if (scale) v = v * scale[c] + shift[c];
    v = apply_act(v, act);
    if (res) v += res[i];
    out[i] = v;
}

// absoluteEncode (camera_.py:118-128 / ImgEncoder.py:404-414): boxes (x1,y1,x2,y2), wh (W,H) ->
// (x/W, y/H, w/W, h/H, w/h, w*h/(W*H))
__global__ void posenc_kernel(const float *__restrict__ boxes, const float *__restrict__ wh, int R, int64_t n,
                              float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / R;
    const float x = boxes[i * 4], y = boxes[i * 4 + 1], w = boxes[i * 4 + 2] - x, h = boxes[i * 4 + 3] - y;
    const float W = wh[b * 2], H = wh[b * 2 + 1];
    float *o = out + i * 6;
    o[0] = x / W; o[1] = y / H; o[2] = w / W; o[3] = h / H; o[4] = w / h; o[5] = (w * h) / (W * H);
}

// Multi-view summarisation (ImgEncoder.py:385-389): L = softmax over regions of smry[b, :, v];
// out[b, v, :] = F.normalize( sum_r L[r, v] * X[b, r, :] ).  One workgroup per (image, view).
__global__ __launch_bounds__(256) void summarize_kernel(const float *__restrict__ smry, const float *__restrict__ X, int R, int k,
                                                        int D, float *__restrict__ out) {
    __shared__ float L[64];
    __shared__ float red[4];
    const int64_t b = blockIdx.x / k;
    const int v = blockIdx.x % k;
    if (threadIdx.x == 0) {
        float mx = -INFINITY;
        for (int r = 0; r < R; ++r) mx = fmaxf(mx, smry[(b * R + r) * k + v]);
        float den = 0.f;
        for (int r = 0; r < R; ++r) { L[r] = expf(smry[(b * R + r) * k + v] - mx); den += L[r]; }
        for (int r = 0; r < R; ++r) L[r] /= den;
    }
    __syncthreads();
    float ss = 0.f;
    float *o = out + (b * k + v) * (int64_t)D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += L[r] * X[(b * R + r) * (int64_t)D + d];
        o[d] = s;
        ss += s * s;
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);   // F.normalize eps
    for (int d = threadIdx.x; d < D; d += 256) o[d] /= nrm;
}

}  // namespace itr

extern "C" int itr_mul_rows(const float *a, const float *b, int64_t ldb, float *out, int64_t R, int C, itr_stream_t stream) {
    ITR_REQUIRE(a && b && out && R >= 0 && C >= 1 && ldb >= C, "itr_mul_rows: bad argument");
    const int64_t n = R * C;
    if (n == 0) return ITR_OK;
    hipLaunchKernelGGL(itr::mul_rows_kernel, dim3((unsigned)itr::ceil_div(n, 256)), dim3(256), 0, itr::as_stream(stream), a, b, ldb, out, n, C);
    ITR_CHECK_LAUNCH("mul_rows");
    return ITR_OK;
}

extern "C" int itr_affine_cols(const float *x, const float *scale, const float *shift, const float *residual, float *out,
                               int64_t R, int C, int act, itr_stream_t stream) {
    ITR_REQUIRE(x && out && R >= 0 && C >= 1 && (!scale == !shift), "itr_affine_cols: bad argument");
    ITR_REQUIRE(act >= 0 && act <= 5, "itr_affine_cols: unknown activation %d", act);
    const int64_t n = R * C;
    if (n == 0) return ITR_OK;
    hipLaunchKernelGGL(itr::affine_cols_kernel, dim3((unsigned)itr::ceil_div(n, 256)), dim3(256), 0, itr::as_stream(stream), x, scale, shift,
                       residual, out, n, C, act);
    ITR_CHECK_LAUNCH("affine_cols");
    return ITR_OK;
}

extern "C" int itr_camera_posenc(const float *boxes, const float *imgs_wh, float *out, int64_t B, int R, itr_stream_t stream) {
    ITR_REQUIRE(boxes && imgs_wh && out && B >= 0 && R >= 1, "itr_camera_posenc: bad argument");
    const int64_t n = B * R;
    if (n == 0) return ITR_OK;
    hipLaunchKernelGGL(itr::posenc_kernel, dim3((unsigned)itr::ceil_div(n, 256)), dim3(256), 0, itr::as_stream(stream), boxes, imgs_wh, R, n, out);
    ITR_CHECK_LAUNCH("posenc");
    return ITR_OK;
}

extern "C" int itr_camera_summarize(const float *smry, const float *X, float *out, int64_t B, int R, int k, int D,
                                    itr_stream_t stream) {
    ITR_REQUIRE(smry && X && out && B >= 0 && R >= 1 && R <= 64 && k >= 1 && D >= 1, "itr_camera_summarize: bad argument");
    ITR_REQUIRE(B * k < 0x7fffffffLL, "itr_camera_summarize: grid too large");
    if (B == 0) return ITR_OK;
    hipLaunchKernelGGL(itr::summarize_kernel, dim3((unsigned)(B * k)), dim3(256), 0, itr::as_stream(stream), smry, X, R, k, D, out);
    ITR_CHECK_LAUNCH("summarize");
    return ITR_OK;
}
