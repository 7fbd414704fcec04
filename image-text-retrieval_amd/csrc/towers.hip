// Image / text towers of the GRU model family (VSE++, SCAN, SGRAF):
//   itr_proj_l2norm : EncoderImagePrecomp.forward (itr/modalmodule/ImgEncoder.py:133-147)
//   itr_gru_fwd     : EncoderText.forward         (itr/modalmodule/TextEncoder.py:38-70)
//
// GRU layout: captions are PACKED (caption b owns token rows tok_off[b] .. +len[b]) and sorted by
// length descending, so the set of captions still running at step t is the prefix [0, n_t) --
// exactly torch's pack_padded_sequence batch_sizes.  The input projection of ALL tokens is hoisted
// into one MFMA GEMM [n_tok, E] x [E, 3D]; each time step is then one MFMA GEMM on the active
// prefix (h[0:n_t] x W_hh^T) plus one fused gate kernel (sigmoid / tanh / state update / output
// write, direction average and last-step gather folded in).
#include <stdlib.h>
#include "itr_common.h"
#include <mutex>
#define ITR_SIDE_STREAM_IMPL
#include "side_stream.h"
#include <mutex>

namespace itr {

int gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
            int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t st);
// the two directions of a bi-GRU time step in one launch (gemm_f32.hip)
bool gemm_pair_ok(int64_t lda, int64_t ldb, int64_t K);
int gemm_nt_pair(const float *A, const float *A2, int64_t lda, const float *B, const float *B2, int64_t ldb, const float *bias, const float *bias2,
                 float *C, float *C2, int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t st);
int norm_rows(const float *x, float *y, int64_t rows, int dim, float eps, int kind, int take_abs,
              hipStream_t st);
// skinny GEMMs of the recurrence when the batch is small (the reference-shaped encode_data path feeds 128 captions at
// a time): split-K with a deterministic reduction (gemm_f32.hip)
int gemm_splitk_choice(int64_t M, int64_t N, int64_t K);
size_t gemm_splitk_scratch_bytes(int64_t M, int64_t N, int splits);
int gemm_nt_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M, int64_t N,
                   int64_t K, int act, int accumulate, int splits, float *scratch, hipStream_t st);

// x rows are padded with zeros to Ep = E rounded up to 32 columns (300 -> 320) so that the input projection runs on
// the branch-free GEMM path (K % 32 == 0); the matching zero-padded copy of W_ih is made by pad_cols_kernel.
__global__ void embed_gather_kernel(const int64_t *__restrict__ tokens, int64_t n_tok, const float *__restrict__ embed,
                                    int64_t V, int E, int Ep, float *__restrict__ x) {
    const int64_t row = blockIdx.x;
    int64_t id = tokens[row];
    if (id < 0 || id >= V) id = 0;      // memory safety only: the CALLER validates the ids (itr_hip.h; nn.Embedding would raise IndexError)
    const float *src = embed + id * E;
    float *dst = x + row * Ep;
    for (int k = threadIdx.x; k < Ep; k += blockDim.x) dst[k] = k < E ? src[k] : 0.f;
}

__global__ void pad_cols_kernel(const float *__restrict__ in, int64_t rows, int cols, int cols_p, float *__restrict__ out) {
    const int64_t r = blockIdx.x;
    for (int k = threadIdx.x; k < cols_p; k += blockDim.x) out[r * cols_p + k] = k < cols ? in[r * cols + k] : 0.f;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// One GRU time step for the active prefix [0, n_act).  Gate order (r, z, n) (torch.nn.GRU):
//   r = s(gi_r + gh_r); z = s(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h' = (1 - z) * n + z * h
// mode 0: out[row] = h'            (forward direction, or uni-directional)
// mode 1: out[row] = (out[row] + h') / 2   (reverse direction of a bi-GRU, TextEncoder.py:54-55)
// mode 2: only the state h is updated (last-state callers: the sequence is never read)
// VEC = 4: one thread per four consecutive hidden units, 16-byte accesses (D % 4 == 0); VEC = 1: any D.
// gridDim.z == 2: both directions of a bi-GRU in one launch -- z = 1 is the reverse direction, its gi / gh / h are `dir_stride`
// floats behind the forward direction's (the two halves of the workspace are carved identically) and its output is out2.
template <int VEC>
__global__ __launch_bounds__(256) void gru_gate_kernel(const float *__restrict__ gi, const float *__restrict__ gh,
                                                       float *__restrict__ h, float *__restrict__ out,
                                                       const int64_t *__restrict__ tok_off,
                                                       const int32_t *__restrict__ len, int t, int reverse,
                                                       int mode, int D, int64_t n_act, int64_t dir_stride, float *__restrict__ out2,
                                                       const int64_t *__restrict__ tok_ids, int64_t V, int cap_stride, int cap_off,
                                                       int64_t gh_stride) {
    // (cap_stride > 1: this launch is ONE of `cap_stride` interleaved caption chains of a time step -- workgroup x is caption
    // x * cap_stride + cap_off; n_act counts the chain's captions.  All buffers stay in caption order.)
    if ((int64_t)blockIdx.x >= n_act) return;
    const int64_t b = (int64_t)blockIdx.x * cap_stride + cap_off;
    const int j = (blockIdx.y * blockDim.x + threadIdx.x) * VEC;
    if (j >= D) return;
    if (blockIdx.z) {
        reverse = 1;
        gi += dir_stride; gh += dir_stride; h += dir_stride; out = out2;
    }
    const int64_t row = tok_off[b] + (reverse ? (len[b] - 1 - t) : t);
    // gi is [n_tok, 3D] (one row per token) or, with tok_ids, the VOCABULARY table [V, 3D] (see itr_gru_fwd): the row of the token's id
    int64_t girow = row;
    if (tok_ids) {
        const int64_t id = tok_ids[row];
        girow = (id < 0 || id >= V) ? 0 : id;      // (nn.Embedding would raise: the Python layer checks the range; here memory stays safe)
    }
    const float *gir = gi + girow * 3 * D + j;
    // gh_stride = 3 D: the recurrence GEMM's row of caption b.  gh_stride = 0: the FIRST step of a direction -- h = 0, so
    // W_hh h + b_hh is b_hh itself for every caption (a GEMM of zero rows sums to +0 exactly): no GEMM is launched and gh = b_hh.
    const float *ghr = gh + b * gh_stride + j;
    float ir[VEC], iz[VEC], in[VEC], hr[VEC], hz[VEC], hn_[VEC], hp[VEC], ov[VEC];
    float *hrow = h + b * D + j, *o = out + row * D + j;
    auto ld = [](float (&d)[VEC], const float *p) {
        if constexpr (VEC == 4) { const float4 v = *reinterpret_cast<const float4 *>(p); d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
        else d[0] = p[0];
    };
    auto st = [](float *p, const float (&d)[VEC]) {
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(p) = float4{d[0], d[1], d[2], d[3]};
        else p[0] = d[0];
    };
    ld(ir, gir); ld(iz, gir + D); ld(in, gir + 2 * D);
    ld(hr, ghr); ld(hz, ghr + D); ld(hn_, ghr + 2 * D);
    ld(hp, hrow);
    if (mode) ld(ov, o);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const float hn = gru_cell(ir[k], iz[k], in[k], hr[k], hz[k], hn_[k], hp[k]);
        hp[k] = hn;
        ov[k] = mode ? (ov[k] + hn) / 2.f : hn;
    }
    st(hrow, hp);
    if (mode != 2) st(o, ov);
}

// Last-state output of a bi-GRU (method_name in {VSE++, VSRN}: TextEncoder.py:57-60 gathers position len - 1 of (fwd + bwd) / 2).
// At position len - 1 the BACKWARD direction has seen exactly one token -- it starts there, from h = 0 -- so its whole recurrence is
// one cell update per caption with gh = W_hh 0 + b_hh = b_hh: no recurrence GEMM, and its input projection is needed for the
// captions' last tokens only.  Same arithmetic as the general path (gru_cell on the same operands): bit-identical results.
__global__ void gather_last_rows_kernel(const float *__restrict__ x, const int64_t *__restrict__ tok_off, const int32_t *__restrict__ len,
                                        int Ep, float *__restrict__ x_last) {
    const int64_t b = blockIdx.x;
    const float *src = x + (tok_off[b] + len[b] - 1) * Ep;
    for (int k = threadIdx.x; k < Ep; k += blockDim.x) x_last[b * Ep + k] = src[k];
}
// the same from the padded embedding TABLE [V, Ep] (the vocabulary-table form of the input projection, itr_gru_fwd)
__global__ void gather_last_rows_tab_kernel(const float *__restrict__ xtab, const int64_t *__restrict__ tokens, int64_t V,
                                            const int64_t *__restrict__ tok_off, const int32_t *__restrict__ len, int Ep,
                                            float *__restrict__ x_last) {
    const int64_t b = blockIdx.x;
    int64_t id = tokens[tok_off[b] + len[b] - 1];
    if (id < 0 || id >= V) id = 0;
    const float *src = xtab + id * Ep;
    for (int k = threadIdx.x; k < Ep; k += blockDim.x) x_last[b * Ep + k] = src[k];
}
// out_last[b] = (h_fwd[b] + gru_cell(gi_last[b], b_hh, h = 0)) / 2
__global__ __launch_bounds__(256) void gru_last_state_bi_kernel(const float *__restrict__ gi_last, const float *__restrict__ b_hh,
                                                                const float *__restrict__ h_fwd, int D, float *__restrict__ out_last) {
    const int64_t b = blockIdx.x;
    for (int j = threadIdx.x; j < D; j += blockDim.x) {
        const float *g = gi_last + b * 3 * D;
        const float hb = gru_cell(g[j], g[D + j], g[2 * D + j], b_hh[j], b_hh[D + j], b_hh[2 * D + j], 0.f);
        out_last[b * D + j] = (h_fwd[b * D + j] + hb) / 2.f;
    }
}

__global__ void gather_last_kernel(const float *__restrict__ out, const int64_t *__restrict__ tok_off,
                                   const int32_t *__restrict__ len, int D, float *__restrict__ out_last) {
    const int64_t b = blockIdx.x;
    const float *src = out + (tok_off[b] + len[b] - 1) * D;
    for (int k = threadIdx.x; k < D; k += blockDim.x) out_last[b * D + k] = src[k];
}

// out = (out + other) / 2: the direction average of a bi-GRU (TextEncoder.py:54-55) when the two directions ran side by side
__global__ __launch_bounds__(256) void avg2_kernel(float *__restrict__ out, const float *__restrict__ other, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = reinterpret_cast<float4 *>(out)[i];
    const float4 b = reinterpret_cast<const float4 *>(other)[i];
    a.x = (a.x + b.x) / 2.f; a.y = (a.y + b.y) / 2.f; a.z = (a.z + b.z) / 2.f; a.w = (a.w + b.w) / 2.f;
    reinterpret_cast<float4 *>(out)[i] = a;
}
// the same for an element count that is not a multiple of 4 (embed dims that are not: no 16-byte rows to rely on)
__global__ __launch_bounds__(256) void avg2_scalar_kernel(float *__restrict__ out, const float *__restrict__ other, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (out[i] + other[i]) / 2.f;
}

// (SideStream / side_stream: side_stream.h)
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

struct GruWs {
    float *x, *gi, *gh, *h, *out_tmp, *wpad, *skbuf;
};

// split-K scratch only for batches that need it (<= 1024 captions: 16 slices x B x 3D floats <= 200 MB)
static inline size_t gru_splitk_bytes(int64_t B, int D) { return B <= 1024 ? gemm_splitk_scratch_bytes(B, 3 * D, 16) : 0; }

static inline int pad32(int E) { return (E + 31) / 32 * 32; }
static inline int64_t pad128(int64_t B) { return (B + 127) / 128 * 128; }

static GruWs carve(void *ws, int64_t n_tok, int64_t B, int E, int D) {
    char *p = static_cast<char *>(ws);
    GruWs w;
    w.x = reinterpret_cast<float *>(p); p += al256((size_t)n_tok * pad32(E) * 4);
    w.gi = reinterpret_cast<float *>(p); p += al256((size_t)n_tok * 3 * D * 4);
    w.gh = reinterpret_cast<float *>(p); p += al256((size_t)pad128(B) * 3 * D * 4);      // (rows padded to whole 128-row GEMM tiles: see the recurrence)
    w.h = reinterpret_cast<float *>(p); p += al256((size_t)pad128(B) * D * 4);
    w.out_tmp = reinterpret_cast<float *>(p); p += al256((size_t)n_tok * D * 4);
    w.wpad = reinterpret_cast<float *>(p); p += al256((size_t)3 * D * pad32(E) * 4);
    w.skbuf = reinterpret_cast<float *>(p); p += al256(gru_splitk_bytes(B, D));
    return w;
}

}  // namespace itr

extern "C" int itr_proj_l2norm(const float *x, const float *W, const float *b, float *out, int64_t rows, int F,
                               int D, int no_imgnorm, int use_abs, itr_stream_t stream) {
    ITR_REQUIRE(x && W && out, "itr_proj_l2norm: null pointer");
    ITR_REQUIRE(rows >= 0 && F > 0 && D > 0, "itr_proj_l2norm: bad shape");
    hipStream_t st = itr::as_stream(stream);
    int rc = itr::gemm_nt(x, F, W, F, b, out, D, rows, D, F, 0, st);
    if (rc != ITR_OK) return rc;
    if (!no_imgnorm) return itr::norm_rows(out, out, rows, D, 1e-8f, 0, use_abs, st);
    if (use_abs) {
        // abs without normalisation: kind 2 with eps = +inf would rescale; do it as l1 of nothing.
        itr::set_error("itr_proj_l2norm: use_abs without normalisation is not supported");
        return ITR_ERR_UNSUPPORTED;
    }
    return ITR_OK;
}

static size_t gru_ws_one(int64_t n_tok, int64_t B, int E, int D) {
    using itr::al256;
    return al256((size_t)n_tok * itr::pad32(E) * 4) + al256((size_t)n_tok * 3 * D * 4) + al256((size_t)itr::pad128(B) * 3 * D * 4) +
           al256((size_t)itr::pad128(B) * D * 4) + al256((size_t)n_tok * D * 4) + al256((size_t)3 * D * itr::pad32(E) * 4) +
           al256(itr::gru_splitk_bytes(B, D)) + 256;
}

extern "C" size_t itr_gru_workspace_bytes(int64_t n_tok, int64_t B, int E, int D, int bidirectional) {
    // a bi-GRU runs its two directions side by side (second HIP stream): each needs its own gate pre-activations and state
    return gru_ws_one(n_tok, B, E, D) * (bidirectional ? 2 : 1);
}

extern "C" int itr_gru_fwd(const int64_t *tokens, const int64_t *tok_off, const int32_t *len_dev,
                           const int32_t *len_host, int64_t B, int64_t n_tok, const float *embed, int64_t V, int E,
                           int D, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                           const float *w_ih_rev, const float *w_hh_rev, const float *b_ih_rev,
                           const float *b_hh_rev, int no_txtnorm, int use_abs, int gather_last, float *out,
                           float *out_last, void *workspace, size_t workspace_bytes, itr_stream_t stream) {
    using namespace itr;
    const bool batch_invariant = (gather_last & ITR_GRU_BATCH_INVARIANT) != 0;
    const bool want_paired = (gather_last & ITR_GRU_PAIRED_DIRECTIONS) != 0, want_input_after_fork = (gather_last & ITR_GRU_INPUT_AFTER_FORK) != 0;
    const bool want_per_token = (gather_last & ITR_GRU_PER_TOKEN_INPUT) != 0;
    const int chains_arg = (gather_last >> 5) & 7;      // ITR_GRU_CHAINS(n): 0 = the library's choice
    const bool want_first_gemm = (gather_last & ITR_GRU_FIRST_STEP_GEMM) != 0;
    gather_last &= ITR_GRU_GATHER_LAST;
    ITR_REQUIRE(tokens && tok_off && len_dev && len_host && embed && w_ih && w_hh && b_ih && b_hh && workspace,
                "itr_gru_fwd: null pointer");
    ITR_REQUIRE(B >= 1 && n_tok >= 1 && E > 0 && D > 0 && V > 0, "itr_gru_fwd: bad shape");
    const bool bi = w_ih_rev != nullptr;
    ITR_REQUIRE(!bi || (w_hh_rev && b_ih_rev && b_hh_rev), "itr_gru_fwd: incomplete reverse-direction weights");
    ITR_REQUIRE(gather_last ? out_last != nullptr : out != nullptr, "itr_gru_fwd: output pointer missing");
    ITR_REQUIRE(workspace_bytes >= itr_gru_workspace_bytes(n_tok, B, E, D, bi), "itr_gru_fwd: workspace too small");
    ITR_REQUIRE(use_abs == 0 || no_txtnorm == 0, "itr_gru_fwd: use_abs without l2norm is not supported");
    int64_t total = 0;
    for (int64_t b = 0; b < B; ++b) {
        ITR_REQUIRE(len_host[b] >= 1, "itr_gru_fwd: caption %lld has length %d", (long long)b, len_host[b]);
        ITR_REQUIRE(b == 0 || len_host[b] <= len_host[b - 1],
                    "itr_gru_fwd: captions must be sorted by length, descending (pack_padded_sequence)");
        total += len_host[b];
    }
    ITR_REQUIRE(total == n_tok, "itr_gru_fwd: sum(len) = %lld != n_tok = %lld", (long long)total, (long long)n_tok);
    hipStream_t st = as_stream(stream);
    GruWs w = carve(workspace, n_tok, B, E, D);
    float *seq = out ? out : w.out_tmp;
    const int Lmax = len_host[0];

    const int Ep = pad32(E);
    // (split-K sums a dot product in slices: the result depends on the batch size through the slice count -- never with
    // ITR_GRU_BATCH_INVARIANT; the plain kernels all run the same fmaf chain per output element whatever M is)
    const int splits_h = (B <= 1024 && !batch_invariant && !ITR_EXP_ENV("ITR_GRU_NO_SPLITK")) ? gemm_splitk_choice(B, 3 * D, D) : 1;   // env: A/B switch for tools/
    // The input projection of a token is a function of its ID alone: gi[token] = W_ih emb[id] + b_ih.  When the call has more tokens
    // than the vocabulary has words (an evaluation: 325 623 tokens over 11 353 words at 5k x 25k; Flickr30k 1k x 5k: 64 952 over 8 481)
    // the projection runs ONCE PER WORD -- a [V, Ep] x [Ep, 3D] GEMM into the head of the gi region -- and the gate kernel reads the
    // row of the token's id.  Same GEMM kernels, same fmaf chain per output element: bit-identical to the per-token projection
    // (tests/test_kernels_gpu.py: test_gru_vocabulary_table_is_bit_identical).  Round 5: the per-token GEMM was 8 ms of the 5k x 25k
    // step and 0.9 of VSE++'s 5.9 ms, the table GEMM is 0.3 / 0.2 ms; the 2 x 4 GB of per-token pre-activations are neither written
    // nor streamed back (the tables are 2 x 140 MB and stay in the Infinity Cache).  flag ITR_GRU_PER_TOKEN_INPUT: the per-token
    // form, kept as the cross-check.
    const bool table = 2 * V <= n_tok && !want_per_token;
    const int64_t x_rows = table ? V : n_tok;
    const int64_t *gi_ids = table ? tokens : nullptr;
    if (table) {
        hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)V), dim3(128), 0, st, embed, V, E, Ep, w.x);
        ITR_CHECK_LAUNCH("embed table");
    } else {
        hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)n_tok), dim3(128), 0, st, tokens, n_tok, embed, V, E, Ep, w.x);
        ITR_CHECK_LAUNCH("embed_gather");
    }

    GruWs w2 = w;
    if (bi) {
        w2 = carve(static_cast<char *>(workspace) + gru_ws_one(n_tok, B, E, D), n_tok, B, E, D);
        w2.x = w.x;                      // the gathered embeddings are shared (read-only)
    }
    // ---- last-state output (VSE++ / VSRN): the forward recurrence without sequence stores; of the backward direction only its first
    // step (see gru_last_state_bi_kernel).  Round 5: half of the recurrence of BASELINE config [1] was work nothing reads.
    if (gather_last && !want_paired && !want_input_after_fork) {
        auto project = [&](const float *xin, int64_t rows, const float *wi, const float *bi_, float *wpad, float *dst) -> int {
            const float *wi_use = wi;
            if (Ep != E) {
                hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)(3 * D)), dim3(128), 0, st, wi, (int64_t)3 * D, E, Ep, wpad);
                ITR_CHECK_LAUNCH("pad_cols");
                wi_use = wpad;
            }
            return gemm_nt(xin, Ep, wi_use, Ep, bi_, dst, 3 * D, rows, 3 * D, Ep, 0, st);
        };
        int rc = project(w.x, x_rows, w_ih, b_ih, w.wpad, w.gi);
        if (rc != ITR_OK) return rc;
        ITR_CHECK_HIP(hipMemsetAsync(w.h, 0, (size_t)B * D * 4, st));
        if (bi) {      // the backward direction's one step: projection of the B last tokens (the second workspace half's unused x region)
            float *x_last = reinterpret_cast<float *>(static_cast<char *>(workspace) + gru_ws_one(n_tok, B, E, D));
            if (table)
                hipLaunchKernelGGL(gather_last_rows_tab_kernel, dim3((unsigned)B), dim3(128), 0, st, w.x, tokens, V, tok_off, len_dev, Ep, x_last);
            else
                hipLaunchKernelGGL(gather_last_rows_kernel, dim3((unsigned)B), dim3(128), 0, st, w.x, tok_off, len_dev, Ep, x_last);
            ITR_CHECK_LAUNCH("gather_last_rows");
            rc = project(x_last, B, w_ih_rev, b_ih_rev, w2.wpad, w2.gi);
            if (rc != ITR_OK) return rc;
        }
        // The recurrence of a caption depends on no other caption: the batch is cut into `nch` INTERLEAVED chains (caption c belongs to
        // chain c mod nch: equal length mixes), each with its own GEMM -> gates -> GEMM sequence on its own stream.  A time step of ONE
        // chain is a strict sequence of dependent launches -- the chip drains after every GEMM (332 us per 5 000-row step in the VSE++
        // trace against 256 us for the same GEMM issued back to back) -- while the chains fill each other's drains, as the two
        // directions of the word-level models do.  Rows of h / gh stay in caption order (the GEMM's lda / ldc = nch rows); every
        // output element is the same fmaf chain: bit-identical to one chain (test_gru_chains_are_bit_identical).
        const int nch = splits_h > 1 ? 1 : (chains_arg > 0 ? (chains_arg < GRU_MAX_CHAINS ? chains_arg : GRU_MAX_CHAINS) : (B >= 4096 ? 2 : 1));
        SideStream chain_side[GRU_MAX_CHAINS - 1];
        int n_side = 0;
        for (int k = 0; k + 1 < nch; ++k) {
            if (!side_stream(chain_side[k], k)) break;
            ++n_side;
        }
        // fork: a chain whose event cannot be recorded or waited on is not used (the recurrence runs on the chains that did fork; with
        // none, on the caller's stream alone) -- never a return with chains 0 .. k - 1 forked and not joined
        {
            int forked = 0;
            for (int k = 0; k < n_side; ++k) {
                if (hipEventRecord(chain_side[k].fork, st) != hipSuccess || hipStreamWaitEvent(chain_side[k].st, chain_side[k].fork, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                chain_side[k].forked = true;
                ++forked;
            }
            n_side = forked;
        }
        const int nchains = n_side + 1;
        auto run_chains = [&]() -> int {
            int64_t n_act = B;
            for (int t = 0; t < Lmax; ++t) {
                while (n_act > 0 && len_host[n_act - 1] <= t) --n_act;
                for (int p = 0; p < nchains; ++p) {
                    const int64_t n_p = n_act > p ? (n_act - p + nchains - 1) / nchains : 0;      // captions c < n_act with c mod nchains == p
                    if (n_p == 0) continue;
                    hipStream_t sp = p ? chain_side[p - 1].st : st;
                    const bool first = t == 0 && !want_first_gemm;      // h = 0: W_hh h + b_hh = b_hh, no GEMM
                    if (!first) {
                        int rc2 = (splits_h > 1) ? gemm_nt_splitk(w.h, D, w_hh, D, b_hh, w.gh, 3 * D, n_act, 3 * D, D, 0, 0, splits_h, w.skbuf, sp)
                                                 : gemm_nt(w.h + (size_t)p * D, (int64_t)nchains * D, w_hh, D, b_hh, w.gh + (size_t)p * 3 * D,
                                                           (int64_t)nchains * 3 * D, n_p, 3 * D, D, 0, sp);
                        if (rc2 != ITR_OK) return rc2;
                    }
                    const float *gh_use = first ? b_hh : w.gh;
                    const int64_t gh_stride = first ? 0 : (int64_t)3 * D;
                    if (D % 4 == 0)
                        hipLaunchKernelGGL(gru_gate_kernel<4>, dim3((unsigned)n_p, (unsigned)ceil_div(D, 1024)), dim3(256), 0, sp, w.gi, gh_use, w.h,
                                           w.out_tmp, tok_off, len_dev, t, 0, 2, D, n_p, (int64_t)0, (float *)nullptr, gi_ids, V, nchains, p, gh_stride);
                    else
                        hipLaunchKernelGGL(gru_gate_kernel<1>, dim3((unsigned)n_p, (unsigned)ceil_div(D, 256)), dim3(256), 0, sp, w.gi, gh_use, w.h,
                                           w.out_tmp, tok_off, len_dev, t, 0, 2, D, n_p, (int64_t)0, (float *)nullptr, gi_ids, V, nchains, p, gh_stride);
                    ITR_CHECK_LAUNCH("gru_gate (state only)");
                }
            }
            return ITR_OK;
        };
        rc = run_chains();
        for (int k = 0; k < n_side; ++k) chain_side[k].join_into(st);      // also on the error path: nothing stays queued behind our back
        if (rc != ITR_OK) return rc;
        // a caption's row of h is not touched after its last step: it IS the state at position len - 1
        if (bi) {
            hipLaunchKernelGGL(gru_last_state_bi_kernel, dim3((unsigned)B), dim3(256), 0, st, w2.gi, b_hh_rev, w.h, D, out_last);
            ITR_CHECK_LAUNCH("gru_last_state_bi");
        } else {
            ITR_CHECK_HIP(hipMemcpyAsync(out_last, w.h, (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));
        }
        if (!no_txtnorm) {
            rc = norm_rows(out_last, out_last, B, D, 1e-8f, 0, use_abs, st);
            if (rc != ITR_OK) return rc;
        }
        return ITR_OK;
    }
    // Both input projections first, on the caller's stream (each fills the chip by itself), THEN the fork: the two recurrences start
    // together and their short last steps (a few hundred active captions: less than one round of tiles) overlap each other instead
    // of the reverse direction's tail running alone (flag ITR_GRU_INPUT_AFTER_FORK: the round-2 order, kept as a cross-check).
    const bool input_first = bi && !want_input_after_fork;
    auto input_projection = [&](int dir, hipStream_t sd) -> int {
        const GruWs &ww = dir ? w2 : w;
        const float *wi_use = dir ? w_ih_rev : w_ih;
        if (Ep != E) {   // zero-padded copy of W_ih: K = Ep is a multiple of 32 (zeros add nothing)
            hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)(3 * D)), dim3(128), 0, sd, wi_use, (int64_t)3 * D, E, Ep, ww.wpad);
            ITR_CHECK_LAUNCH("pad_cols");
            wi_use = ww.wpad;
        }
        int rc = gemm_nt(w.x, Ep, wi_use, Ep, dir ? b_ih_rev : b_ih, ww.gi, 3 * D, x_rows, 3 * D, Ep, 0, sd);
        if (rc != ITR_OK) return rc;
        ITR_CHECK_HIP(hipMemsetAsync(ww.h, 0, (size_t)pad128(B) * D * 4, sd));
        return ITR_OK;
    };
    if (input_first)
        for (int dir = 0; dir < 2; ++dir) {
            const int rc = input_projection(dir, st);
            if (rc != ITR_OK) return rc;
        }
    // flag ITR_GRU_PAIRED_DIRECTIONS (an experiment that lost: 10.26 against 9.92 ms on VSE++ 1k x 5k, same box): ONE GEMM launch and
    // ONE gate launch per time step for both directions.  The active prefix of step t is the same for both, so the pair is a GEMM
    // of twice the tiles; but GEMM -> gates -> GEMM is then a strict chain, while two streams let one direction's gate kernel run
    // in the other direction's last round of tiles.  Bit-identical results either way (same fmaf chain per element).
    const bool paired = input_first && splits_h == 1 && D % 4 == 0 && gemm_pair_ok(D, D, D) && want_paired;
    if (paired) {
        const int64_t dir_stride = (int64_t)(gru_ws_one(n_tok, B, E, D) / 4);
        int64_t n_act = B;
        for (int t = 0; t < Lmax; ++t) {
            while (n_act > 0 && len_host[n_act - 1] <= t) --n_act;
            int rc = gemm_nt_pair(w.h, w2.h, D, w_hh, w_hh_rev, D, b_hh, b_hh_rev, w.gh, w2.gh, 3 * D, n_act, 3 * D, D, st);
            if (rc != ITR_OK) return rc;
            hipLaunchKernelGGL(gru_gate_kernel<4>, dim3((unsigned)n_act, (unsigned)ceil_div(D, 1024), 2u), dim3(256), 0, st, w.gi, w.gh, w.h,
                               seq, tok_off, len_dev, t, 0, 0, D, n_act, dir_stride, w2.out_tmp, gi_ids, V, 1, 0, (int64_t)3 * D);
            ITR_CHECK_LAUNCH("gru_gate (both directions)");
        }
    }
    // the reverse direction's recurrence on a second stream with its own buffers (ITR_GRU_NO_OVERLAP=1: one after the other, for A/B timing)
    SideStream side_obj;
    SideStream *side = (bi && !paired && !ITR_EXP_ENV("ITR_GRU_NO_OVERLAP") && side_stream(side_obj)) ? &side_obj : nullptr;
    if (side) {
        ITR_CHECK_HIP(hipEventRecord(side->fork, st));
        ITR_CHECK_HIP(hipStreamWaitEvent(side->st, side->fork, 0));
        side->forked = true;
    }
    // (Round 4 measured the gates in the recurrence GEMM's epilogue -- one launch per time step, bit-identical -- and removed it again:
    // VSE++ 1k x 5k 10.13 ms against 10.06 for GEMM + gate kernel, and the larger argument block cost the plain GEMM 0.4 %;
    // profiles/r04/NOTES.md 4b, commit "GRU: gates in the recurrence GEMM's epilogue".)
    auto directions = [&]() -> int {
        for (int dir = 0; dir < (paired ? 0 : bi ? 2 : 1); ++dir) {
            const float *wh = dir ? w_hh_rev : w_hh, *bh = dir ? b_hh_rev : b_hh;
            const GruWs &ww = dir ? w2 : w;
            hipStream_t sd = (dir && side) ? side->st : st;
            float *dst = dir ? w2.out_tmp : seq;             // reverse direction: its own sequence buffer, averaged in below
            int rc = ITR_OK;
            if (!input_first && (rc = input_projection(dir, sd)) != ITR_OK) return rc;
            int64_t n_act = B;
            for (int t = 0; t < Lmax; ++t) {
                while (n_act > 0 && len_host[n_act - 1] <= t) --n_act;
                // (t = 0: h = 0, the recurrence product is b_hh for every caption -- the largest GEMM of the direction is not launched)
                const bool first = t == 0 && !want_first_gemm;
                if (!first) {
                    // The active prefix rounded UP to whole 128-row tiles (the rows exist: h / gh are padded, rows past n_act hold finished
                    // captions or zeros and nobody reads their products): the streaming GEMM then takes every row and the second launch for
                    // the M mod 128 remainder (24 workgroups, ~60 us, 32 per 5k x 25k step) is gone.  A row's result does not depend on M.
                    const int64_t m_gemm = n_act >= 5000 ? pad128(n_act) : n_act;
                    rc = (splits_h > 1) ? gemm_nt_splitk(ww.h, D, wh, D, bh, ww.gh, 3 * D, n_act, 3 * D, D, 0, 0, splits_h, ww.skbuf, sd)
                                        : gemm_nt(ww.h, D, wh, D, bh, ww.gh, 3 * D, m_gemm, 3 * D, D, 0, sd);
                    if (rc != ITR_OK) return rc;
                }
                const float *gh_use = first ? bh : ww.gh;
                const int64_t gh_stride = first ? 0 : (int64_t)3 * D;
                if (D % 4 == 0)
                    hipLaunchKernelGGL(gru_gate_kernel<4>, dim3((unsigned)n_act, (unsigned)ceil_div(D, 1024)), dim3(256), 0, sd, ww.gi, gh_use, ww.h,
                                       dst, tok_off, len_dev, t, dir, 0, D, n_act, (int64_t)0, (float *)nullptr, gi_ids, V, 1, 0, gh_stride);
                else
                    hipLaunchKernelGGL(gru_gate_kernel<1>, dim3((unsigned)n_act, (unsigned)ceil_div(D, 256)), dim3(256), 0, sd, ww.gi, gh_use, ww.h,
                                       dst, tok_off, len_dev, t, dir, 0, D, n_act, (int64_t)0, (float *)nullptr, gi_ids, V, 1, 0, gh_stride);
                ITR_CHECK_LAUNCH("gru_gate");
            }
        }
        return ITR_OK;
    };
    const int rc_dirs = directions();
    if (side) side->join_into(st);        // also on the error paths: nothing stays queued on the side stream behind our back
    if (rc_dirs != ITR_OK) return rc_dirs;
    if (bi) {
        const int64_t n = n_tok * (int64_t)D;
        if (n % 4 == 0)
            hipLaunchKernelGGL(avg2_kernel, dim3((unsigned)ceil_div(n / 4, 256)), dim3(256), 0, st, seq, w2.out_tmp, n / 4);
        else
            hipLaunchKernelGGL(avg2_scalar_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, seq, w2.out_tmp, n);
        ITR_CHECK_LAUNCH("gru avg2");
    }
    float *final_ = seq;
    int64_t final_rows = n_tok;
    if (gather_last) {
        hipLaunchKernelGGL(gather_last_kernel, dim3((unsigned)B), dim3(256), 0, st, seq, tok_off, len_dev, D, out_last);
        ITR_CHECK_LAUNCH("gather_last");
        final_ = out_last;
        final_rows = B;
    }
    if (!no_txtnorm) {
        int rc = norm_rows(final_, final_, final_rows, D, 1e-8f, 0, use_abs, st);
        if (rc != ITR_OK) return rc;
    }
    return ITR_OK;
}
