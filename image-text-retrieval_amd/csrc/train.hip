// Training-step primitives (Models.py train_emb: forward -> loss -> backward -> clip_grad_norm_ -> Adam), HBM-bound
// elementwise / reduction kernels; all dense contractions of the backward pass go through gemm_nt_kernel on
// transposed operands (itr_transpose2d):   dW = dY^T X,  dX = dY W.
//
//   itr_l2norm_fwd_save / itr_l2norm_bwd   z = x / (||x|| + eps) with the row norms kept for the backward
//                                          (utils.py:10-15;  dx = dz / (n + eps) - z (z . dz) / n)
//   itr_transpose2d                        out[c, r] = in[r, c]           (LDS-tiled, coalesced both ways)
//   itr_colsum                             out[c] = sum_r x[r, c]         (bias gradients; fixed summation order)
//   itr_embed_scatter_add                  dE[token[r], :] += dx[r, :]    (nn.Embedding backward)
//   itr_sq_sum                             partial sums of squares        (clip_grad_norm_, Models.py:223-224)
//   itr_adam_step                          torch.optim.Adam update (no weight decay, no amsgrad), gradient pre-scaled
#include "itr_common.h"

namespace itr {

// ---------------------------------------------------------------- l2norm with saved norms
__global__ __launch_bounds__(256) void l2norm_fwd_save_kernel(const float *__restrict__ x, float *__restrict__ z,
                                                              float *__restrict__ nrm, int64_t rows, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * dim;
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) s += xr[c] * xr[c];
    s = wave_sum(s);
    const float n = sqrtf(s);
    const float d = n + eps;
    for (int c = lane; c < dim; c += 64) z[row * dim + c] = xr[c] / d;
    if (lane == 0) nrm[row] = n;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *__restrict__ dz, const float *__restrict__ z,
                                                         const float *__restrict__ nrm, float *__restrict__ dx, int64_t rows,
                                                         int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *gr = dz + row * dim, *zr = z + row * dim;
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) s += gr[c] * zr[c];
    s = wave_sum(s);
    const float n = nrm[row];
    const float inv = 1.f / (n + eps);
    // d/dx [x / (n + eps)] = I / (n + eps) - x x^T / (n (n + eps)^2);  a zero row (n = 0) has the plain 1 / eps slope
    const float k = n > 0.f ? s / n : 0.f;
    for (int c = lane; c < dim; c += 64) dx[row * dim + c] = gr[c] * inv - zr[c] * k;
}

// ---------------------------------------------------------------- transpose
__global__ __launch_bounds__(256) void transpose2d_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t rows,
                                                          int64_t cols) {
    __shared__ float t[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        if (r0 + i < rows && c0 + tx < cols) t[i][tx] = in[(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < cols && r0 + tx < rows) out[(c0 + i) * rows + r0 + tx] = t[tx][i];
}

// ---------------------------------------------------------------- column sums (two deterministic passes)
// Rows per partial: 32 up to 2 048 rows, else rows / 64 rounded up to a multiple of 32 -- at most 64 partials, each summed with 16
// loads in flight.  (Round 6: with 256 rows per partial and one load per loop trip a 128 x 512 sum was two workgroups waiting for 128
// L2 round trips each, 30 us; the sum stays a fixed order of additions for a given shape.)
static int64_t cs_rows_per_part(int64_t rows) {
    if (rows <= 2048) return 32;
    return ceil_div(ceil_div(rows, (int64_t)64), (int64_t)32) * 32;
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ x, float *__restrict__ part, int64_t rows,
                                                             int64_t cols, int64_t rpp) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * rpp;
    const int64_t r1 = r0 + rpp < rows ? r0 + rpp : rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t r = r0;
    for (; r + 16 <= r1; r += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = x[(r + i) * cols + c];
#pragma unroll
        for (int i = 0; i < 16; i += 4) { s0 += v[i]; s1 += v[i + 1]; s2 += v[i + 2]; s3 += v[i + 3]; }
    }
    for (; r < r1; ++r) s0 += x[r * cols + c];
    part[(int64_t)blockIdx.y * cols + c] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ part, float *__restrict__ out, int64_t nparts,
                                                           int64_t cols, int accumulate) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s0 = accumulate ? out[c] : 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t p = 0;
    for (; p + 16 <= nparts; p += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = part[(p + i) * cols + c];
#pragma unroll
        for (int i = 0; i < 16; i += 4) { s0 += v[i]; s1 += v[i + 1]; s2 += v[i + 2]; s3 += v[i + 3]; }
    }
    for (; p < nparts; ++p) s0 += part[p * cols + c];
    out[c] = (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------- embedding backward
__global__ __launch_bounds__(128) void embed_scatter_add_kernel(const int64_t *__restrict__ tokens, const float *__restrict__ dx,
                                                                int64_t n_tok, int64_t V, int E, float *__restrict__ dE) {
    const int64_t row = blockIdx.x;
    const int64_t id = tokens[row];
    if (id < 0 || id >= V) return;
    for (int k = threadIdx.x; k < E; k += 128) atomicAdd(dE + id * E + k, dx[row * E + k]);
}

// ---------------------------------------------------------------- optimizer
__global__ __launch_bounds__(256) void sq_sum_kernel(const float *__restrict__ g, int64_t n, float *__restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// p, m, v updated in place.  torch.optim.Adam (single-tensor path):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// `gscale` carries clip_grad_norm_'s coefficient min(1, max_norm / (total_norm + 1e-6)).
__global__ __launch_bounds__(256) void adam_step_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                        float *__restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                        float bc1, float bc2_sqrt, const float *__restrict__ gscale_dev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gs = gscale_dev ? gscale_dev[0] : 1.f;
    const float gi = g[i] * gs;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
}

// ---- all tensors of an optimizer step in ONE launch each (round 6: a step of CAMERA / SGRAF / VSRN has 51-67 parameter tensors = twice
// that many launches of a few microseconds).  `tab`: one record per tensor (device copy of a host table the caller rebuilds per step --
// gradients are new allocations every step); `blk_tensor[b]` = the tensor workgroup b works on, `blk_first[t]` = its first workgroup.
// Same arithmetic per element / per partial as the single-tensor kernels (the partials land in the same layout: bit-identical norm).
struct OptTensor {
    float *p;
    const float *g;
    float *m, *v;
    int64_t n;
    int32_t first_blk, nblk;        // sq-sum: workgroups [first_blk, first_blk + nblk) stride over the tensor
};
__global__ __launch_bounds__(256) void sq_sum_multi_kernel(const OptTensor *__restrict__ tab, const int32_t *__restrict__ blk_tensor,
                                                           float *__restrict__ part) {
    __shared__ float red[4];
    const OptTensor t = tab[blk_tensor[blockIdx.x]];
    const int64_t local = (int64_t)blockIdx.x - t.first_blk;
    float s = 0.f;
    for (int64_t i = local * 256 + threadIdx.x; i < t.n; i += (int64_t)t.nblk * 256) s += t.g[i] * t.g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void adam_step_multi_kernel(const OptTensor *__restrict__ tab, const int32_t *__restrict__ blk_tensor,
                                                              const int32_t *__restrict__ blk_first, float lr, float b1, float b2, float eps,
                                                              float bc1, float bc2_sqrt, const float *__restrict__ gscale_dev) {
    const int ti = blk_tensor[blockIdx.x];
    const OptTensor t = tab[ti];
    const int64_t i = ((int64_t)blockIdx.x - blk_first[ti]) * 256 + threadIdx.x;
    if (i >= t.n) return;
    const float gs = gscale_dev ? gscale_dev[0] : 1.f;
    const float gi = t.g[i] * gs;
    const float mi = b1 * t.m[i] + (1.f - b1) * gi;
    const float vi = b2 * t.v[i] + (1.f - b2) * gi * gi;
    t.m[i] = mi;
    t.v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    t.p[i] = t.p[i] - (lr / bc1) * (mi / denom);
}

// total_norm = sqrt(sum of all partials); coef = min(1, max_norm / (total_norm + 1e-6))   (torch clip_grad_norm_)
__global__ __launch_bounds__(64) void clip_coef_kernel(const float *__restrict__ part, int64_t nparts, float max_norm, float *__restrict__ out2) {
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int64_t i = lane; i < nparts; i += 64) s += (double)part[i];        // fixed lane-strided order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane != 0) return;
    const float total = (float)sqrt(s);
    out2[1] = total;
    const float c = max_norm / (total + 1e-6f);
    out2[0] = (max_norm > 0.f && c < 1.f) ? c : 1.f;
}

}  // namespace itr

using namespace itr;

extern "C" int itr_l2norm_fwd_save(const float *x, float *z, float *norms, int64_t rows, int dim, float eps, itr_stream_t stream) {
    ITR_REQUIRE(x && z && norms, "itr_l2norm_fwd_save: null pointer");
    ITR_REQUIRE(rows >= 0 && dim > 0, "itr_l2norm_fwd_save: bad shape");
    if (rows == 0) return ITR_OK;
    hipLaunchKernelGGL(l2norm_fwd_save_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, as_stream(stream), x, z, norms, rows, dim, eps);
    ITR_CHECK_LAUNCH("l2norm_fwd_save");
    return ITR_OK;
}

extern "C" int itr_l2norm_bwd(const float *dz, const float *z, const float *norms, float *dx, int64_t rows, int dim, float eps,
                              itr_stream_t stream) {
    ITR_REQUIRE(dz && z && norms && dx, "itr_l2norm_bwd: null pointer");
    ITR_REQUIRE(rows >= 0 && dim > 0, "itr_l2norm_bwd: bad shape");
    if (rows == 0) return ITR_OK;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, as_stream(stream), dz, z, norms, dx, rows, dim, eps);
    ITR_CHECK_LAUNCH("l2norm_bwd");
    return ITR_OK;
}

extern "C" int itr_transpose2d(const float *in, float *out, int64_t rows, int64_t cols, itr_stream_t stream) {
    ITR_REQUIRE(in && out && in != out, "itr_transpose2d: null or aliased pointer");
    ITR_REQUIRE(rows >= 0 && cols >= 0, "itr_transpose2d: bad shape");
    if (rows == 0 || cols == 0) return ITR_OK;
    ITR_UNSUPPORTED(ceil_div(rows, 64) > 65535, "itr_transpose2d: more than 4M rows; transpose in slabs");
    hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)ceil_div(cols, 64), (unsigned)ceil_div(rows, 64)), dim3(256), 0, as_stream(stream),
                       in, out, rows, cols);
    ITR_CHECK_LAUNCH("transpose2d");
    return ITR_OK;
}

extern "C" size_t itr_colsum_workspace_bytes(int64_t rows, int64_t cols) {
    const int64_t r = rows > 0 ? rows : 1;
    return (size_t)ceil_div(r, cs_rows_per_part(r)) * (size_t)(cols > 0 ? cols : 1) * 4 + 256;
}

extern "C" int itr_colsum(const float *x, float *out, int64_t rows, int64_t cols, int accumulate, void *workspace, size_t workspace_bytes,
                          itr_stream_t stream) {
    ITR_REQUIRE(x && out && workspace, "itr_colsum: null pointer");
    ITR_REQUIRE(rows >= 0 && cols > 0, "itr_colsum: bad shape");
    ITR_REQUIRE(workspace_bytes >= itr_colsum_workspace_bytes(rows, cols), "itr_colsum: workspace too small");
    const int64_t rpp = cs_rows_per_part(rows > 0 ? rows : 1);
    const int64_t nparts = ceil_div(rows, rpp);
    float *part = static_cast<float *>(workspace);
    hipStream_t st = as_stream(stream);
    if (nparts > 0) {
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)ceil_div(cols, 256), (unsigned)nparts), dim3(256), 0, st, x, part, rows, cols, rpp);
        ITR_CHECK_LAUNCH("colsum_partial");
    }
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div(cols, 256)), dim3(256), 0, st, part, out, nparts, cols, accumulate);
    ITR_CHECK_LAUNCH("colsum_final");
    return ITR_OK;
}

extern "C" int itr_embed_scatter_add(const int64_t *tokens, const float *dx, int64_t n_tok, int64_t V, int E, float *dE, itr_stream_t stream) {
    ITR_REQUIRE(tokens && dx && dE, "itr_embed_scatter_add: null pointer");
    ITR_REQUIRE(n_tok >= 0 && V > 0 && E > 0, "itr_embed_scatter_add: bad shape");
    if (n_tok == 0) return ITR_OK;
    hipLaunchKernelGGL(embed_scatter_add_kernel, dim3((unsigned)n_tok), dim3(128), 0, as_stream(stream), tokens, dx, n_tok, V, E, dE);
    ITR_CHECK_LAUNCH("embed_scatter_add");
    return ITR_OK;
}

extern "C" int itr_sq_sum_blocks(int64_t n) {
    const int64_t b = ceil_div(n > 0 ? n : 1, 256 * 8);
    return (int)(b < 1024 ? b : 1024);
}

extern "C" int itr_sq_sum(const float *g, int64_t n, float *partials, itr_stream_t stream) {
    ITR_REQUIRE(g && partials, "itr_sq_sum: null pointer");
    ITR_REQUIRE(n >= 0, "itr_sq_sum: bad size");
    hipLaunchKernelGGL(sq_sum_kernel, dim3((unsigned)itr_sq_sum_blocks(n)), dim3(256), 0, as_stream(stream), g, n, partials);
    ITR_CHECK_LAUNCH("sq_sum");
    return ITR_OK;
}

extern "C" int itr_clip_coef(const float *partials, int64_t nparts, float max_norm, float *coef_and_norm, itr_stream_t stream) {
    ITR_REQUIRE(partials && coef_and_norm, "itr_clip_coef: null pointer");
    ITR_REQUIRE(nparts >= 0, "itr_clip_coef: bad size");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, as_stream(stream), partials, nparts, max_norm, coef_and_norm);
    ITR_CHECK_LAUNCH("clip_coef");
    return ITR_OK;
}

extern "C" int itr_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps,
                             int64_t step, const float *grad_scale_dev, itr_stream_t stream) {
    ITR_REQUIRE(p && g && m && v, "itr_adam_step: null pointer");
    ITR_REQUIRE(n >= 0 && step >= 1, "itr_adam_step: bad size / step");
    if (n == 0) return ITR_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1, beta2,
                       eps, (float)bc1, (float)sqrt(bc2), grad_scale_dev);
    ITR_CHECK_LAUNCH("adam_step");
    return ITR_OK;
}

/* One record per tensor, 48 bytes: p, g, m, v (device pointers), n (int64), first_blk, nblk (int32: the tensor's range of sq-sum
 * workgroups, nblk = itr_sq_sum_blocks(n)). */
extern "C" int itr_sq_sum_multi(const void *table_dev, const int32_t *blk_tensor_dev, int64_t n_blocks, float *partials, itr_stream_t stream) {
    ITR_REQUIRE(n_blocks >= 0 && n_blocks <= 0x7fffffff, "itr_sq_sum_multi: bad size");
    if (n_blocks == 0) return ITR_OK;
    ITR_REQUIRE(table_dev && blk_tensor_dev && partials, "itr_sq_sum_multi: null pointer");
    static_assert(sizeof(OptTensor) == 48, "OptTensor is the 48-byte record the Python side packs");
    hipLaunchKernelGGL(sq_sum_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, as_stream(stream), static_cast<const OptTensor *>(table_dev),
                       blk_tensor_dev, partials);
    ITR_CHECK_LAUNCH("sq_sum_multi");
    return ITR_OK;
}

extern "C" int itr_adam_step_multi(const void *table_dev, const int32_t *blk_tensor_dev, const int32_t *blk_first_dev, int64_t n_blocks, float lr,
                                   float beta1, float beta2, float eps, int64_t step, const float *grad_scale_dev, itr_stream_t stream) {
    ITR_REQUIRE(n_blocks >= 0 && n_blocks <= 0x7fffffff && step >= 1, "itr_adam_step_multi: bad size / step");
    if (n_blocks == 0) return ITR_OK;
    ITR_REQUIRE(table_dev && blk_tensor_dev && blk_first_dev, "itr_adam_step_multi: null pointer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_step_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, as_stream(stream), static_cast<const OptTensor *>(table_dev),
                       blk_tensor_dev, blk_first_dev, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale_dev);
    ITR_CHECK_LAUNCH("adam_step_multi");
    return ITR_OK;
}
