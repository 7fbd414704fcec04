// Row normalisation (itr/modalmodule/utils.py:4-15): y = x / (norm(x) + eps), eps AFTER the
// sqrt -- this is not F.normalize, and parity depends on it.  HBM-bound: one wave per row,
// float4 loads, wave-shuffle reduction; rows up to 4096 floats stay in registers (one read).
#include "itr_common.h"

namespace itr {

constexpr int NORM_WAVES = 4;
constexpr int NORM_MAXV = 16;  // float4 per lane kept in registers -> dim <= 64*4*16 = 4096

// kind: 0 l2 (+eps after sqrt), 1 l1 (+eps), 2 F.normalize (x / max(||x||, eps)),
//       3 plain x / ||x||  (pdist_cos, Objectives.py:318-319: no eps -> 0/0 = NaN kept)
// Returns the denominator d; the output is the true division x / d (matches torch.div).
__device__ __forceinline__ float finish_denominator(float s, float eps, int kind) {
    if (kind == 1) return s + eps;
    const float n = sqrtf(s);
    if (kind == 0) return n + eps;
    if (kind == 2) return fmaxf(n, eps);
    return n;
}

template <bool VEC>
__global__ __launch_bounds__(NORM_WAVES * 64) void norm_rows_kernel(const float *__restrict__ x,
                                                                   float *__restrict__ y, int64_t rows,
                                                                   int dim, float eps, int kind,
                                                                   int take_abs) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * NORM_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * dim;
    float *yr = y + row * dim;
    if (VEC) {
        const int nv = dim >> 2;
        float4 v[NORM_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                v[i] = reinterpret_cast<const float4 *>(xr)[c];
                if (kind == 1)
                    s += fabsf(v[i].x) + fabsf(v[i].y) + fabsf(v[i].z) + fabsf(v[i].w);
                else
                    s += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
            }
        }
        s = wave_sum(s);
        const float d = finish_denominator(s, eps, kind);
#pragma unroll
        for (int i = 0; i < NORM_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 o = make_float4(v[i].x / d, v[i].y / d, v[i].z / d, v[i].w / d);
                if (take_abs) o = make_float4(fabsf(o.x), fabsf(o.y), fabsf(o.z), fabsf(o.w));
                reinterpret_cast<float4 *>(yr)[c] = o;
            }
        }
    } else {
        float s = 0.f;
        for (int c = lane; c < dim; c += 64) {
            const float t = xr[c];
            s += (kind == 1) ? fabsf(t) : t * t;
        }
        s = wave_sum(s);
        const float d = finish_denominator(s, eps, kind);
        for (int c = lane; c < dim; c += 64) {
            float o = xr[c] / d;
            yr[c] = take_abs ? fabsf(o) : o;
        }
    }
}

int norm_rows(const float *x, float *y, int64_t rows, int dim, float eps, int kind, int take_abs,
              hipStream_t st) {
    if (rows == 0) return ITR_OK;
    const int64_t nblk = ceil_div(rows, NORM_WAVES);
    const bool vec = (dim % 4 == 0) && (dim <= 64 * 4 * NORM_MAXV) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(norm_rows_kernel<true>, dim3((unsigned)nblk), dim3(NORM_WAVES * 64), 0, st, x, y,
                           rows, dim, eps, kind, take_abs);
    else
        hipLaunchKernelGGL(norm_rows_kernel<false>, dim3((unsigned)nblk), dim3(NORM_WAVES * 64), 0, st, x, y,
                           rows, dim, eps, kind, take_abs);
    ITR_CHECK_LAUNCH("norm_rows");
    return ITR_OK;
}

}  // namespace itr

extern "C" int itr_l2norm_rows(const float *x, float *y, int64_t rows, int dim, float eps, int kind,
                               int take_abs, itr_stream_t stream) {
    ITR_REQUIRE(rows >= 0 && dim > 0, "itr_l2norm_rows: bad shape rows=%lld dim=%d", (long long)rows, dim);
    if (rows == 0) return ITR_OK;
    ITR_REQUIRE(x && y, "itr_l2norm_rows: null pointer");
    ITR_REQUIRE(kind >= 0 && kind <= 3, "itr_l2norm_rows: unknown kind %d", kind);
    ITR_REQUIRE(rows < (int64_t)4 * 0x7fffffff, "itr_l2norm_rows: too many rows");
    return itr::norm_rows(x, y, rows, dim, eps, kind, take_abs, itr::as_stream(stream));
}

// ---- mean over the middle axis: y[b, :] = mean_r x[b, r, :]  (torch.mean(x, 1) on the path:
// Fusionmodule.py:412 img_ave, :422 cap_ave; TextEncoder.py:191; ImgEncoder.py:348).  HBM-bound.
namespace itr {
__global__ __launch_bounds__(256) void mean_mid_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t B,
                                                       int R, int F) {
    const int64_t b = blockIdx.y;
    const int f = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (f >= F) return;
    const float *p = x + (b * R) * (int64_t)F + f;
    if (f + 3 < F && (F & 3) == 0) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < R; ++r) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (int64_t)r * F);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float inv = (float)R;
        *reinterpret_cast<float4 *>(y + b * F + f) = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
    } else {
        for (int u = 0; u < 4 && f + u < F; ++u) {
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += p[(int64_t)r * F + u];
            y[b * F + f + u] = s / (float)R;
        }
    }
}
}  // namespace itr

extern "C" int itr_mean_mid(const float *x, float *y, int64_t B, int R, int F, itr_stream_t stream) {
    ITR_REQUIRE(x && y, "itr_mean_mid: null pointer");
    ITR_REQUIRE(B >= 0 && R >= 1 && F >= 1 && B <= 65535 * (int64_t)1024, "itr_mean_mid: bad shape");
    ITR_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "itr_mean_mid: unaligned");
    if (B == 0) return ITR_OK;
    ITR_REQUIRE(B <= 65535, "itr_mean_mid: at most 65535 groups per call");
    dim3 grid((unsigned)itr::ceil_div(itr::ceil_div(F, 4), 256), (unsigned)B);
    hipLaunchKernelGGL(itr::mean_mid_kernel, grid, dim3(256), 0, itr::as_stream(stream), x, y, B, R, F);
    ITR_CHECK_LAUNCH("mean_mid");
    return ITR_OK;
}
