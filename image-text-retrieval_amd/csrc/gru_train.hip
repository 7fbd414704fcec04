// EncoderText (bi)GRU, training forward (keeps the gate activations) and backward through time
// (TextEncoder.py:38-70 under autograd; torch.nn.GRU equations, gate order r, z, n):
//     r = s(gi_r + gh_r)   z = s(gi_z + gh_z)   n = tanh(gi_n + r * gh_n)   h' = (1 - z) n + z h
// with gi = W_ih x + b_ih (all tokens, one GEMM) and gh = W_hh h + b_hh (one GEMM per step on the active prefix).
//
// Same packed layout as itr_gru_fwd (towers.hip): caption b owns token rows tok_off[b] .. +len[b], captions sorted
// by length descending, so the captions alive at step t are the prefix [0, n_t).
//
// save (per direction d, 5 planes of [n_tok, D] floats, indexed by TOKEN ROW):  h | r | z | n | gh_n
//
// Backward, per direction, t = Lmax-1 .. 0:
//     dh   = d_out[row] * (bi ? 1/2 : 1) + carry[b]
//     dn   = dh (1 - z)(1 - n^2);   dz = dh (h_prev - n) z (1 - z);   dr = dn gh_n r (1 - r)
//     dgi[row] = (dr, dz, dn)       dgh[row] = (dr, dz, dn r)          carry[b] = dh z  (+)= dgh W_hh   (MFMA GEMM)
// then, over ALL tokens at once (MFMA GEMMs on transposed operands):
//     dW_hh = dgh^T H_prev    db_hh = colsum(dgh)    dW_ih = dgi^T X    db_ih = colsum(dgi)    dX (+)= dgi W_ih
// and dE[token] += dX (atomic scatter; the only non-deterministic summation order of the step).
#include <stdlib.h>
#include "itr_common.h"
#include "side_stream.h"

namespace itr {

int gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc,
            int64_t M, int64_t N, int64_t K, int act, hipStream_t st);
int gemm_nt_acc(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc,
                int64_t M, int64_t N, int64_t K, int act, hipStream_t st);
// skinny GEMMs of the recurrence (M = batch): split-K with a deterministic reduction (gemm_f32.hip)
int gemm_splitk_choice(int64_t M, int64_t N, int64_t K);
size_t gemm_splitk_scratch_bytes(int64_t M, int64_t N, int splits);
int gemm_nt_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M, int64_t N,
                   int64_t K, int act, int accumulate, int splits, float *scratch, hipStream_t st);
int gemm_nt_splitk_partials(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t M, int64_t N, int64_t K, int splits, float *scratch,
                            int *n_slices, hipStream_t st);

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

__global__ void embed_gather_train_kernel(const int64_t *__restrict__ tokens, int64_t n_tok, const float *__restrict__ embed, int64_t V,
                                          int E, float *__restrict__ x, int *__restrict__ bad) {
    const int64_t row = blockIdx.x;
    int64_t id = tokens[row];
    if (id < 0 || id >= V) {
        if (threadIdx.x == 0) atomicExch(bad, 1);
        id = 0;
    }
    for (int k = threadIdx.x; k < E; k += blockDim.x) x[row * E + k] = embed[id * E + k];
}

struct GruSave {
    float *h, *r, *z, *n, *ghn;
};
static GruSave save_planes(void *save, int dir, int64_t n_tok, int D) {
    float *p = static_cast<float *>(save) + (size_t)dir * 5 * n_tok * D;
    const size_t pl = (size_t)n_tok * D;
    return GruSave{p, p + pl, p + 2 * pl, p + 3 * pl, p + 4 * pl};
}

// forward step; mode 0: out[row] = h', mode 1: out[row] = (out[row] + h') / 2
// ns > 0: `gh` holds ns split-K slices [ns][n_act][3D] of h W_hh^T (no bias): they are summed here, in slice order, + b_hh.
__global__ __launch_bounds__(256) void gru_gate_train_kernel(const float *__restrict__ gi, const float *__restrict__ gh, float *__restrict__ h,
                                                             float *__restrict__ out, GruSave sv, const int64_t *__restrict__ tok_off,
                                                             const int32_t *__restrict__ len, int t, int reverse, int mode, int D,
                                                             int64_t n_act, int ns, const float *__restrict__ b_hh) {
    const int64_t b = blockIdx.x;
    const int j = blockIdx.y * blockDim.x + threadIdx.x;
    if (b >= n_act || j >= D) return;
    const int64_t row = tok_off[b] + (reverse ? (len[b] - 1 - t) : t);
    const float *gir = gi + row * 3 * D;
    const float *ghr = gh + b * 3 * D;
    float g0 = ghr[j], g1 = ghr[D + j], g2 = ghr[2 * D + j];
    if (ns > 0) {
        const int64_t ss = n_act * 3 * (int64_t)D;
        for (int s_ = 1; s_ < ns; ++s_) { g0 += ghr[s_ * ss + j]; g1 += ghr[s_ * ss + D + j]; g2 += ghr[s_ * ss + 2 * D + j]; }
        g0 += b_hh[j]; g1 += b_hh[D + j]; g2 += b_hh[2 * D + j];
    }
    const float r = sigm(gir[j] + g0);
    const float z = sigm(gir[D + j] + g1);
    const float ghn = g2;
    const float n = tanhf(gir[2 * D + j] + r * ghn);
    const float hp = h[b * D + j];
    const float hn = (1.f - z) * n + z * hp;
    h[b * D + j] = hn;
    const int64_t o = row * D + j;
    sv.h[o] = hn; sv.r[o] = r; sv.z[o] = z; sv.n[o] = n; sv.ghn[o] = ghn;
    out[o] = mode ? (out[o] + hn) / 2.f : hn;
}

__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(const float *__restrict__ d_out, float dscale, GruSave sv, float *__restrict__ carry,
                                                           float *__restrict__ dgi, float *__restrict__ dgh, float *__restrict__ dgh_step,
                                                           float *__restrict__ hprev_all, const int64_t *__restrict__ tok_off,
                                                           const int32_t *__restrict__ len, int t, int reverse, int D, int64_t n_act,
                                                           const float *__restrict__ part, int ns, int64_t prev_n) {
    const int64_t b = blockIdx.x;
    const int j = blockIdx.y * blockDim.x + threadIdx.x;
    if (b >= n_act || j >= D) return;
    const int pos = reverse ? (len[b] - 1 - t) : t;
    const int64_t row = tok_off[b] + pos;
    const int64_t o = row * D + j;
    const float hp = (t > 0) ? sv.h[(row + (reverse ? 1 : -1)) * D + j] : 0.f;   // the state this step started from
    const float r = sv.r[o], z = sv.z[o], n = sv.n[o], ghn = sv.ghn[o];
    float cin = carry[b * D + j];
    if (ns > 0 && b < prev_n) {          // dgh W_hh of the step processed just before (time t + 1), still in split-K slices
        const int64_t ss = prev_n * (int64_t)D;
        for (int s_ = 0; s_ < ns; ++s_) cin += part[s_ * ss + b * D + j];
    }
    const float dh = d_out[o] * dscale + cin;
    const float dn = dh * (1.f - z) * (1.f - n * n);
    const float dz = dh * (hp - n) * z * (1.f - z);
    const float dr = dn * ghn * r * (1.f - r);
    float *gi_ = dgi + row * 3 * D, *gh_ = dgh + row * 3 * D, *gs_ = dgh_step + b * 3 * D;
    gi_[j] = dr; gi_[D + j] = dz; gi_[2 * D + j] = dn;
    const float dnr = dn * r;
    gh_[j] = dr; gh_[D + j] = dz; gh_[2 * D + j] = dnr;
    gs_[j] = dr; gs_[D + j] = dz; gs_[2 * D + j] = dnr;
    hprev_all[o] = hp;
    carry[b * D + j] = dh * z;
}

// out[c, r] = in[r, c] with out rows padded to `ldo` >= rows floats; the pad columns [rows, ldo) are written as zeros so
// that the padded length can serve as a K % 32 == 0 contraction dimension of the branch-free GEMM.
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t rows, int64_t cols,
                                                        int64_t ldo) {
    __shared__ float t[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) t[i][tx] = (r0 + i < rows && c0 + tx < cols) ? in[(r0 + i) * cols + c0 + tx] : 0.f;
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < cols && r0 + tx < ldo) out[(c0 + i) * ldo + r0 + tx] = t[tx][i];
}
static int transpose(const float *in, float *out, int64_t rows, int64_t cols, int64_t ldo, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div(cols, 64), (unsigned)ceil_div(ldo, 64)), dim3(256), 0, st, in, out, rows, cols,
                       ldo);
    ITR_CHECK_LAUNCH("gru transpose");
    return ITR_OK;
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

static int check_lengths(const int32_t *len_host, int64_t B, int64_t n_tok, const char *who) {
    int64_t total = 0;
    for (int64_t b = 0; b < B; ++b) {
        ITR_REQUIRE(len_host[b] >= 1, "%s: caption %lld has length %d", who, (long long)b, len_host[b]);
        ITR_REQUIRE(b == 0 || len_host[b] <= len_host[b - 1], "%s: captions must be sorted by length, descending", who);
        total += len_host[b];
    }
    ITR_REQUIRE(total == n_tok, "%s: sum(len) = %lld != n_tok = %lld", who, (long long)total, (long long)n_tok);
    return ITR_OK;
}

}  // namespace itr

using namespace itr;

extern "C" int itr_gather_rows(const int64_t *idx, int64_t n, const float *table, int64_t V, int E, float *out, int *bad_flag,
                               itr_stream_t stream) {
    ITR_REQUIRE(idx && table && out && bad_flag, "itr_gather_rows: null pointer");
    ITR_REQUIRE(n >= 0 && V > 0 && E > 0, "itr_gather_rows: bad shape");
    if (n == 0) return ITR_OK;
    hipLaunchKernelGGL(embed_gather_train_kernel, dim3((unsigned)n), dim3(128), 0, as_stream(stream), idx, n, table, V, E, out, bad_flag);
    ITR_CHECK_LAUNCH("gather_rows");
    return ITR_OK;
}

extern "C" size_t itr_gru_train_save_bytes(int64_t n_tok, int D, int bidirectional) {
    return (size_t)(bidirectional ? 2 : 1) * 5 * (size_t)n_tok * D * 4;
}

namespace itr {
// Workspace of the training forward / backward (one layout for both).  The two directions of a bi-GRU run SIDE BY SIDE (the reverse
// one on the device's side stream, side_stream.h): a time step of one direction is a strict chain of small dependent launches that
// leaves most of the chip idle, the other direction fills it.  So every buffer a recurrence writes exists once per direction.
struct GruTrainWs {
    float *x, *dx, *out_rev;          // shared: embedded tokens; input gradient; the reverse direction's outputs before the average
    int *bad;
    struct Dir {
        float *gi, *gh, *h;           // forward: gate pre-activations of all tokens; [B, 3D] recurrence product; [B, D] state
        float *hprev, *carry, *dgh_step, *whhT, *wihT, *skbuf;
        void *tn;                     // gemm_tn partials of the two weight gradients
        size_t tn_bytes;
    } d[2];                           // backward: gi / gh double as dgi / dgh
};
size_t gemm_tn_workspace_bytes(int64_t R, int P, int Q);
int gemm_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, int64_t R, int P, int Q, int accumulate,
            float *colsum_a, void *workspace, size_t workspace_bytes, hipStream_t st);

static size_t gru_train_ws_carve(void *base, int64_t n_tok, int64_t B, int E, int D, GruTrainWs *w) {
    const size_t nt = (size_t)n_tok, d3 = (size_t)3 * D;
    char *p = static_cast<char *>(base);
    auto take = [&](size_t bytes) { char *q = p; p += al256(bytes); return q; };
    GruTrainWs t;
    t.x = (float *)take(nt * E * 4);
    t.dx = (float *)take(nt * E * 4);
    t.out_rev = (float *)take(nt * D * 4);
    t.bad = (int *)take(512);
    const size_t tnb = gemm_tn_workspace_bytes(n_tok, 3 * D, D > E ? D : E);
    for (int k = 0; k < 2; ++k) {
        t.d[k].gi = (float *)take(nt * d3 * 4);
        t.d[k].gh = (float *)take(nt * d3 * 4);
        t.d[k].h = (float *)take((size_t)B * D * 4);
        t.d[k].hprev = (float *)take(nt * D * 4);
        t.d[k].carry = (float *)take((size_t)B * D * 4);
        t.d[k].dgh_step = (float *)take((size_t)B * d3 * 4);
        t.d[k].whhT = (float *)take(d3 * D * 4);
        t.d[k].wihT = (float *)take(d3 * E * 4);
        t.d[k].skbuf = (float *)take(gemm_splitk_scratch_bytes(B, 3 * D, 16));
        t.d[k].tn = take(tnb);
        t.d[k].tn_bytes = tnb;
    }
    if (w) *w = t;
    return (size_t)(p - static_cast<char *>(base));
}

}  // namespace itr
using namespace itr;

extern "C" size_t itr_gru_train_workspace_bytes(int64_t n_tok, int64_t B, int E, int D) {
    return gru_train_ws_carve(nullptr, n_tok > 0 ? n_tok : 1, B > 0 ? B : 1, E, D, nullptr);
}

__global__ __launch_bounds__(256) void gru_avg_kernel(float *__restrict__ out, const float *__restrict__ other, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (out[i] + other[i]) / 2.f;
}

extern "C" int itr_gru_fwd_train(const int64_t *tokens, const int64_t *tok_off, const int32_t *len_dev, const int32_t *len_host, int64_t B,
                                 int64_t n_tok, const float *embed, int64_t V, int E, int D, const float *w_ih, const float *w_hh,
                                 const float *b_ih, const float *b_hh, const float *w_ih_rev, const float *w_hh_rev, const float *b_ih_rev,
                                 const float *b_hh_rev, float *out, void *save, size_t save_bytes, void *workspace, size_t workspace_bytes,
                                 itr_stream_t stream) {
    ITR_REQUIRE(tokens && tok_off && len_dev && len_host && embed && w_ih && w_hh && b_ih && b_hh && out && save && workspace,
                "itr_gru_fwd_train: null pointer");
    ITR_REQUIRE(B >= 1 && n_tok >= 1 && E > 0 && D > 0 && V > 0, "itr_gru_fwd_train: bad shape");
    const bool bi = w_ih_rev != nullptr;
    ITR_REQUIRE(!bi || (w_hh_rev && b_ih_rev && b_hh_rev), "itr_gru_fwd_train: incomplete reverse-direction weights");
    ITR_REQUIRE(save_bytes >= itr_gru_train_save_bytes(n_tok, D, bi), "itr_gru_fwd_train: save buffer too small");
    ITR_REQUIRE(workspace_bytes >= itr_gru_train_workspace_bytes(n_tok, B, E, D), "itr_gru_fwd_train: workspace too small");
    int rc = check_lengths(len_host, B, n_tok, "itr_gru_fwd_train");
    if (rc != ITR_OK) return rc;
    hipStream_t st = as_stream(stream);
    GruTrainWs w;
    gru_train_ws_carve(workspace, n_tok, B, E, D, &w);
    const int splits_h = gemm_splitk_choice(B, 3 * D, D);
    const bool fuse = ITR_EXP_ENV("ITR_GRU_REDUCE_KERNEL") == nullptr;   // tools/ A/B switch: separate reduction kernel
    const int Lmax = len_host[0];
    ITR_CHECK_HIP(hipMemsetAsync(w.bad, 0, sizeof(int), st));
    hipLaunchKernelGGL(embed_gather_train_kernel, dim3((unsigned)n_tok), dim3(128), 0, st, tokens, n_tok, embed, V, E, w.x, w.bad);
    ITR_CHECK_LAUNCH("embed_gather(train)");
    SideStream side_obj;
    SideStream *side = (bi && side_stream(side_obj)) ? &side_obj : nullptr;
    if (side) {
        if (hipEventRecord(side->fork, st) == hipSuccess && hipStreamWaitEvent(side->st, side->fork, 0) == hipSuccess) side->forked = true;
        else { (void)hipGetLastError(); side = nullptr; }
    }
    auto run_dir = [&](int dir, hipStream_t sd) -> int {
        const float *wi = dir ? w_ih_rev : w_ih, *wh = dir ? w_hh_rev : w_hh;
        const float *bi_ = dir ? b_ih_rev : b_ih, *bh = dir ? b_hh_rev : b_hh;
        GruTrainWs::Dir &d = w.d[dir];
        GruSave sv = save_planes(save, dir, n_tok, D);
        int rc2 = gemm_nt(w.x, E, wi, E, bi_, d.gi, 3 * D, n_tok, 3 * D, E, 0, sd);
        if (rc2 != ITR_OK) return rc2;
        ITR_CHECK_HIP(hipMemsetAsync(d.h, 0, (size_t)B * D * 4, sd));
        int64_t n_act = B;
        for (int t = 0; t < Lmax; ++t) {
            while (n_act > 0 && len_host[n_act - 1] <= t) --n_act;
            int ns = 0;
            if (splits_h > 1 && fuse) rc2 = gemm_nt_splitk_partials(d.h, D, wh, D, n_act, 3 * D, D, splits_h, d.skbuf, &ns, sd);
            else rc2 = gemm_nt_splitk(d.h, D, wh, D, bh, d.gh, 3 * D, n_act, 3 * D, D, 0, 0, splits_h, d.skbuf, sd);
            if (rc2 != ITR_OK) return rc2;
            // the reverse direction writes its own output plane; the average with the forward direction follows the join
            hipLaunchKernelGGL(gru_gate_train_kernel, dim3((unsigned)n_act, (unsigned)ceil_div(D, 256)), dim3(256), 0, sd, d.gi,
                               ns > 0 ? d.skbuf : d.gh, d.h, dir ? w.out_rev : out, sv, tok_off, len_dev, t, dir, 0, D, n_act, ns, bh);
            ITR_CHECK_LAUNCH("gru_gate(train)");
        }
        return ITR_OK;
    };
    rc = ITR_OK;
    if (bi) rc = run_dir(1, side ? side->st : st);
    if (rc == ITR_OK) rc = run_dir(0, st);
    if (side) side->join_into(st);        // also on the error paths: nothing stays queued on the side stream behind our back
    if (rc != ITR_OK) return rc;
    if (bi) {
        const int64_t n = n_tok * (int64_t)D;
        hipLaunchKernelGGL(gru_avg_kernel, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, st, out, (const float *)w.out_rev, n);
        ITR_CHECK_LAUNCH("gru_avg");
    }
    return ITR_OK;
}

extern "C" int itr_gru_bwd(const int64_t *tokens, const int64_t *tok_off, const int32_t *len_dev, const int32_t *len_host, int64_t B,
                           int64_t n_tok, const float *embed, int64_t V, int E, int D, const float *w_ih, const float *w_hh,
                           const float *w_ih_rev, const float *w_hh_rev, const void *save, const float *d_out, float *d_embed, float *d_w_ih,
                           float *d_w_hh, float *d_b_ih, float *d_b_hh, float *d_w_ih_rev, float *d_w_hh_rev, float *d_b_ih_rev,
                           float *d_b_hh_rev, void *workspace, size_t workspace_bytes, itr_stream_t stream) {
    ITR_REQUIRE(tokens && tok_off && len_dev && len_host && embed && w_ih && w_hh && save && d_out && d_embed && d_w_ih && d_w_hh && d_b_ih &&
                    d_b_hh && workspace, "itr_gru_bwd: null pointer");
    ITR_REQUIRE(B >= 1 && n_tok >= 1 && E > 0 && D > 0 && V > 0, "itr_gru_bwd: bad shape");
    const bool bi = w_ih_rev != nullptr;
    ITR_REQUIRE(!bi || (w_hh_rev && d_w_ih_rev && d_w_hh_rev && d_b_ih_rev && d_b_hh_rev), "itr_gru_bwd: incomplete reverse-direction buffers");
    ITR_REQUIRE(workspace_bytes >= itr_gru_train_workspace_bytes(n_tok, B, E, D), "itr_gru_bwd: workspace too small");
    int rc = check_lengths(len_host, B, n_tok, "itr_gru_bwd");
    if (rc != ITR_OK) return rc;
    hipStream_t st = as_stream(stream);
    GruTrainWs w;
    gru_train_ws_carve(workspace, n_tok, B, E, D, &w);
    const int splits_c = gemm_splitk_choice(B, D, 3 * D);
    const bool fuse = ITR_EXP_ENV("ITR_GRU_REDUCE_KERNEL") == nullptr;
    const int Lmax = len_host[0];

    ITR_CHECK_HIP(hipMemsetAsync(w.bad, 0, sizeof(int), st));
    hipLaunchKernelGGL(embed_gather_train_kernel, dim3((unsigned)n_tok), dim3(128), 0, st, tokens, n_tok, embed, V, E, w.x, w.bad);
    ITR_CHECK_LAUNCH("embed_gather(bwd)");
    SideStream side_obj;
    SideStream *side = (bi && side_stream(side_obj)) ? &side_obj : nullptr;
    if (side) {
        if (hipEventRecord(side->fork, st) == hipSuccess && hipStreamWaitEvent(side->st, side->fork, 0) == hipSuccess) side->forked = true;
        else { (void)hipGetLastError(); side = nullptr; }
    }
#define GB_TRY(e) { int rc2 = (e); if (rc2 != ITR_OK) return rc2; }
    // BPTT + the weight gradients of one direction on stream sd; everything it writes is the direction's own
    auto run_dir = [&](int dir, hipStream_t sd) -> int {
        const float *wi = dir ? w_ih_rev : w_ih, *wh = dir ? w_hh_rev : w_hh;
        float *dwi = dir ? d_w_ih_rev : d_w_ih, *dwh = dir ? d_w_hh_rev : d_w_hh;
        float *dbi = dir ? d_b_ih_rev : d_b_ih, *dbh = dir ? d_b_hh_rev : d_b_hh;
        GruTrainWs::Dir &d = w.d[dir];
        float *dgi = d.gi, *dgh = d.gh;
        GruSave sv = save_planes(const_cast<void *>(save), dir, n_tok, D);
        GB_TRY(transpose(wh, d.whhT, 3 * D, D, 3 * D, sd));    // [3D, D] -> [D, 3D]:  carry += dgh_step . W_hh  ==  gemm_nt(dgh_step, W_hh^T)
        GB_TRY(transpose(wi, d.wihT, 3 * D, E, 3 * D, sd));    // [3D, E] -> [E, 3D]
        ITR_CHECK_HIP(hipMemsetAsync(d.carry, 0, (size_t)B * D * 4, sd));
        int ns_pending = 0;
        int64_t prev_n = 0;
        for (int t = Lmax - 1; t >= 0; --t) {
            int64_t n_act = 0;
            while (n_act < B && len_host[n_act] > t) ++n_act;
            hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3((unsigned)n_act, (unsigned)ceil_div(D, 256)), dim3(256), 0, sd, d_out, bi ? 0.5f : 1.f,
                               sv, d.carry, dgi, dgh, d.dgh_step, d.hprev, tok_off, len_dev, t, dir, D, n_act, d.skbuf, ns_pending, prev_n);
            ITR_CHECK_LAUNCH("gru_gate_bwd");
            ns_pending = 0;
            if (t > 0) {
                if (splits_c > 1 && fuse) {   // slices of dgh W_hh stay in scratch: the next gate kernel adds them to carry
                    GB_TRY(gemm_nt_splitk_partials(d.dgh_step, 3 * D, d.whhT, 3 * D, n_act, D, 3 * D, splits_c, d.skbuf, &ns_pending, sd));
                    prev_n = n_act;
                } else {
                    GB_TRY(gemm_nt_splitk(d.dgh_step, 3 * D, d.whhT, 3 * D, nullptr, d.carry, D, n_act, D, 3 * D, 0, 1, splits_c, d.skbuf, sd));
                }
            }
        }
        // weight gradients over all tokens: dW = dg^T A on the split-row TN GEMM, the bias gradient (column sums of dg) from the same pass
        GB_TRY(gemm_tn(dgh, 3 * D, d.hprev, D, dwh, D, n_tok, 3 * D, D, 0, dbh, d.tn, d.tn_bytes, sd));
        GB_TRY(gemm_tn(dgi, 3 * D, w.x, E, dwi, E, n_tok, 3 * D, E, 0, dbi, d.tn, d.tn_bytes, sd));
        return ITR_OK;
    };
    rc = ITR_OK;
    if (bi) rc = run_dir(1, side ? side->st : st);
    if (rc == ITR_OK) rc = run_dir(0, st);
    if (side) side->join_into(st);
    if (rc != ITR_OK) return rc;
    // input gradient: dx = dgi_fwd . W_ih + dgi_rev . W_ih_rev  (after the join: both directions' dgi are complete)
    GB_TRY(gemm_nt(w.d[0].gi, 3 * D, w.d[0].wihT, 3 * D, nullptr, w.dx, E, n_tok, E, 3 * D, 0, st));
    if (bi) GB_TRY(gemm_nt_acc(w.d[1].gi, 3 * D, w.d[1].wihT, 3 * D, nullptr, w.dx, E, n_tok, E, 3 * D, 0, st));
#undef GB_TRY
    return itr_embed_scatter_add(tokens, w.dx, n_tok, V, E, d_embed, stream);
}
