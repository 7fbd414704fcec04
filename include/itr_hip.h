/*
 * itr_hip.h -- C ABI of libitr_hip.so: the MI355X (gfx950) encode -> score -> loss / rank
 * hot path of WangFei-2019/Image-text-Retrieval.
 *
 * The reference is 100 % Python and has no FFI layer; its seams for this path are the
 * Python callables listed in SURVEY.md section 8(b).  Each entry point below replaces the
 * arithmetic of one of them (reference file:line cited per function); the Python package
 * `itr_amd` keeps the reference's call signatures on top of this ABI (see INTEGRATION.md for
 * the ctypes binding a maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major data unless the name ends in
 *     `_host`; fp32 unless stated; the caller owns all memory (no allocation inside);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call only
 *     enqueues work on it and returns;
 *   - return value: ITR_OK (0) or a negative ITR_ERR_* code; `itr_last_error()` returns a
 *     thread-local message for the last failing call on this thread;
 *   - no exceptions cross the ABI, no global mutable state besides the error string.
 */
#ifndef ITR_HIP_H
#define ITR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITR_OK 0
#define ITR_ERR_BADARG (-1)      /* -> Python ValueError */
#define ITR_ERR_UNSUPPORTED (-2) /* -> Python NotImplementedError */
#define ITR_ERR_HIP (-3)         /* -> Python RuntimeError */

typedef void *itr_stream_t;

const char *itr_last_error(void);
/* ABI version, bumped on any signature change. */
int itr_abi_version(void);

/* ---- a1: l2norm / l1norm  (itr/modalmodule/utils.py:4-15) ----------------------------
 * y[r,:] = x[r,:] / (norm(x[r,:]) + eps), eps added AFTER the sqrt.  kind: 0 = L2, 1 = L1,
 * 2 = F.normalize semantics x / max(||x||_2, eps) (TextEncoder.py:151, ImgEncoder.py:349),
 * 3 = plain x / ||x||_2 without eps (pdist_cos, Objectives.py:318-319; 0/0 stays NaN).
 * take_abs != 0 additionally applies |.| (use_abs, ImgEncoder.py:144).  In place allowed. */
int itr_l2norm_rows(const float *x, float *y, int64_t rows, int dim, float eps, int kind,
                    int take_abs, itr_stream_t stream);

/* y[b,:] = mean_r x[b,r,:] for x [B,R,F]  (torch.mean(x, 1): Fusionmodule.py:412,422; TextEncoder.py:191;
 * ImgEncoder.py:348; the VSE++ region pooling of SURVEY Q3). */
int itr_mean_mid(const float *x, float *y, int64_t B, int R, int F, itr_stream_t stream);

/* ---- measure='order' (config.py:74): order_sim, Objectives.py:24-30 ----------------------
 * out[i, c] = -sqrt( sum_d max(0, s[c, d] - im[i, d])^2 ), out [Ni, Nc] row-major.  D a multiple of 4. */
int itr_order_scores(const float *im, const float *s, float *out, int64_t Ni, int64_t Nc, int D, itr_stream_t stream);
/* Gradient of itr_order_scores for train_emb: S = its output, dS [Ni, Nc] -> d_im [Ni, D], d_s [Nc, D].  A pair with
 * S == 0 (no violated dimension) carries no gradient (torch propagates sqrt'(0) * 0 = NaN there). */
int itr_order_bwd(const float *im, const float *s, const float *S, const float *dS, float *d_im, float *d_s, int64_t Ni,
                  int64_t Nc, int D, itr_stream_t stream);
/* SAEM with measure='order' uses the euclidean distance `pdist` as its "similarity" (Objectives.py:54-56, :297-307):
 * S <- sqrt(n1[i] - 2 S[i, c] + n2[c] + 1e-4) in place on S = x1 x2^T (itr_gemm_nt), n1 / n2 = squared row norms
 * (itr_row_sqnorm: out[r] = sum_d x[r, d]^2). */
int itr_row_sqnorm(const float *x, float *out, int64_t rows, int D, itr_stream_t stream);
int itr_pdist_finish(float *S, const float *n1, const float *n2, int64_t Ni, int64_t Nc, itr_stream_t stream);

/* ---- VSRN region-relationship reasoning: Rs_GCN.forward between its 1x1 convolutions
 * (itr/modalmodule/vsrn_.py:50-71).  tpg [n_img*N, ld]: per region row the three convolution outputs side by side,
 * theta at columns [0, D), phi at [D, 2D), g at [2D, 3D) (one GEMM with the stacked weight).  For every image
 *   y[n, :] = sum_m (theta[n,:] . phi[m,:] / N) g[m, :]           (R = theta_v phi_v^T; R_div_C = R / N; y = R_div_C g_v)
 * y [n_img*N, ldy].  N <= 36 regions, D and ld multiples of 4. */
int itr_gcn_relation(const float *tpg, int64_t ld, float *y, int64_t ldy, int64_t n_img, int N, int D,
                     itr_stream_t stream);

/* ---- generic fp32 MFMA GEMM used by the towers ----------------------------------------
 * C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]);  lda/ldb/ldc are row strides in elements.
 * bias may be NULL.  lda < K is allowed (overlapping A rows = a convolution over consecutive rows).  act: 0 none, 1 relu, 2 tanh, 3 sigmoid, 4 gelu(erf), 5 leaky_relu(0.1),
 * 6 NaN -> 0 (itr_gemm_nt only: pdist_cos' `res[res != res] = 0`, Objectives.py:321, without a second pass over the matrix).
 * Exact fp32 (v_mfma_f32_32x32x2_f32).  Replaces nn.Linear / torch.mm call sites on the path
 * (ImgEncoder.py:137, Objectives.py:21, Fusionmodule.py:427-431 ...). */
int itr_gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                itr_stream_t stream);
/* The same product with the kernel chosen by the CALLER (cross-checks and timing; results are bit-identical for every choice):
 * algo 0 = itr_gemm_nt's own selection rule, 1 = the 128 x 128 tile kernel, 2 / 3 = the streaming kernel (plain / XCD-aware tile
 * map) for every shape it admits (N % 128 == 0, K % 64 == 0, K >= 128, 16-byte rows, act in {none, relu, gelu}; other shapes
 * fall through to the tile kernel).  algo 4 = the skinny kernel for M <= 128 rows (N >= 64, K >= 128, 16-byte rows; csrc/gemm_skinny.hip:
 * 16-column strips over all rows, N / 16 workgroups): NOT bit-identical to the others (another k order) and therefore never the
 * library's own choice -- the training tape asks for it for decoder steps and per-caption vector layers. */
int itr_gemm_nt_algo(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                     float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, int algo,
                     itr_stream_t stream);

/* ---- split-bf16 GEMM (study, opt-in; SURVEY.md 8d "bf16-in / fp32-acc variant reported separately") -------------
 * itr_split_bf16: x [rows, K] fp32 -> out [rows][K / 32][hi (32 bf16) | lo (32 bf16)], hi = bf16(x), lo = bf16(x - hi), both
 * round-to-nearest-even: x = hi + lo + O(2^-17 |x|).  K a multiple of 32; one 32-wide chunk of one row is one 128-byte line.
 * itr_gemm_nt_bf16: C[M,N] = act(A B^T + bias) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation from such operands;
 * terms = 3: hi.hi + hi.lo + lo.hi ("bf16x3"), terms = 1: hi.hi (plain bf16).  lda / ldb count interleaved bf16 elements
 * (2 x the fp32 row stride; lda < 2 K = overlapping rows as in itr_gemm_nt).  No default path calls these: itr_gemm_nt
 * (exact fp32) is the product GEMM. */
int itr_split_bf16(const float *x, uint16_t *out, int64_t rows, int64_t K, itr_stream_t stream);
int itr_gemm_nt_bf16(const uint16_t *A, int64_t lda, const uint16_t *B, int64_t ldb, const float *bias, float *C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int act, int terms, itr_stream_t stream);

/* "fp16x3" variant of the study GEMM: fp16 planes (11 + 11 mantissa bits) of x' = s x with s the power of two that puts the
 * tensor's absmax in [2^14, 2^15) (the matrix core flushes fp16 subnormals, so hi must stay normal), lo scaled by 2^11, cross
 * terms in a second accumulator set, everything folded back with exact powers of two: error at the fp32 rounding level for
 * ANY finite operands.  itr_split_f16 computes absmax and the planes; scale_state (2 floats, device) receives the absmax
 * bits and 1 / s and is handed to the GEMM.  Same operand layout as itr_split_bf16. */
int itr_split_f16(const float *x, uint16_t *out, float *scale_state, int64_t rows, int64_t K, itr_stream_t stream);
int itr_gemm_nt_f16x3(const uint16_t *A, int64_t lda, const float *scale_state_a, const uint16_t *B, int64_t ldb,
                      const float *scale_state_b, const float *bias, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                      itr_stream_t stream);

/* ---- a2: EncoderImagePrecomp.forward (itr/modalmodule/ImgEncoder.py:133-147) -----------
 * out[rows, D] = l2norm(x[rows, F] * W[D, F]^T + b[D]) [abs]; rows = n_img * n_regions.
 * no_imgnorm / use_abs as in the reference constructor. */
int itr_proj_l2norm(const float *x, const float *W, const float *b, float *out, int64_t rows,
                    int F, int D, int no_imgnorm, int use_abs, itr_stream_t stream);

/* ---- a3: EncoderText.forward (itr/modalmodule/TextEncoder.py:38-70) --------------------
 * Embedding -> (bi)GRU with packed-sequence semantics -> (fwd+bwd)/2 -> [l2norm] [abs].
 * Captions are PACKED: caption c owns rows [tok_off[c], tok_off[c]+len[c]) of `tokens`
 * (int64 ids, device) and of `out` (fp32 [n_tok, D]).  Captions must be sorted by length
 * DESCENDING (collate_fn does this, data_loader.py:146).  Host arrays: len_host[B].
 * w_ih [3D,E], w_hh [3D,D], b_ih/b_hh [3D] with torch gate order (r,z,n); *_rev = NULL for a
 * uni-directional GRU.  gather_last is a flag word: bit 0 (ITR_GRU_GATHER_LAST) writes only the len-1 state of every
 * caption to out_last[B,D] (method_name in {VSE++,VSRN}, TextEncoder.py:57-60) -- `out` may be NULL then; bit 1
 * (ITR_GRU_BATCH_INVARIANT) makes a caption's result independent of the batch it is encoded in, bit for bit (the
 * recurrence GEMM of a small batch is otherwise split along K, which sums in another order): the sharded evaluation
 * sets it so that every partition of the caption axis gives the single-process rank vectors.
 * workspace: itr_gru_workspace_bytes(n_tok, B, E, D, bi) bytes.
 * Token ids must lie in [0, V): the CALLER validates them (nn.Embedding raises IndexError; itr_amd.ops.gru_encode does the same before
 * the call).  The library only keeps memory safe -- an id outside the range reads embedding row 0 -- and raises no flag: the call
 * enqueues work and returns, it cannot report what a kernel finds. */
#define ITR_GRU_GATHER_LAST 1
#define ITR_GRU_BATCH_INVARIANT 2
/* launch-order variants of the bi-GRU kept as cross-checks (bit-identical results; both measured slower than the default):
 * bit 2: ONE recurrence GEMM and ONE gate launch per time step for both directions; bit 3: each direction's input projection
 * on its own stream after the fork (the round-2 order). */
#define ITR_GRU_PAIRED_DIRECTIONS 4
#define ITR_GRU_INPUT_AFTER_FORK 8
/* The input projection W_ih emb[id] + b_ih depends on the token's id alone: a call with at least twice as many tokens as the
 * vocabulary has words (an evaluation) projects the V embedding rows once and the gate kernel reads the row of the token's id
 * (bit-identical: the same GEMM kernels on the same rows).  bit 4 forces the per-token projection (the cross-check). */
#define ITR_GRU_PER_TOKEN_INPUT 16
/* Last-state output: the batch runs as n interleaved caption chains (caption c in chain c mod n), each a GEMM -> gates sequence
 * on its own stream, so that one chain's launches fill the drains of the others (bit-identical for every n).  bits 5-7: n = 1..4;
 * 0 = the library's choice (2 chains from 4 096 captions on). */
#define ITR_GRU_CHAINS(n) (((n) & 7) << 5)
/* The first time step of a direction starts from h = 0: W_hh h + b_hh is b_hh itself (a product of zero rows sums to +0 exactly), so
 * the step's recurrence GEMM -- the largest of the direction, all captions are still running -- is not launched.  bit 8 launches it
 * anyway (the cross-check; bit-identical). */
#define ITR_GRU_FIRST_STEP_GEMM 256
size_t itr_gru_workspace_bytes(int64_t n_tok, int64_t B, int E, int D, int bidirectional);
int itr_gru_fwd(const int64_t *tokens, const int64_t *tok_off, const int32_t *len_dev,
                const int32_t *len_host, int64_t B, int64_t n_tok, const float *embed, int64_t V,
                int E, int D, const float *w_ih, const float *w_hh, const float *b_ih,
                const float *b_hh, const float *w_ih_rev, const float *w_hh_rev,
                const float *b_ih_rev, const float *b_hh_rev, int no_txtnorm, int use_abs,
                int gather_last, float *out, float *out_last, void *workspace,
                size_t workspace_bytes, itr_stream_t stream);

/* ---- a4: cosine_sim (itr/modalmodule/Objectives.py:18-21) ------------------------------
 * S[Ni,Nc] = im[Ni,D] * s[Nc,D]^T.  Also serves pdist_cos (a9, :310-323) after
 * itr_l2norm_rows(kind=3) and MultiViewMatching (a8) via itr_mvm_scores. */
int itr_cosine_scores(const float *im, const float *s, float *S, int64_t Ni, int64_t Nc, int D,
                      int64_t ldS, itr_stream_t stream);

/* ---- a8: MultiViewMatching.forward (itr/modalmodule/Fusionmodule.py:674-692) ------------
 * S[i,c] = max_v imgs[i,v,:] . caps[c,:]   (imgs [Ni,k,D], caps [Nc,D]). */
int itr_mvm_scores(const float *imgs, const float *caps, float *S, int64_t Ni, int64_t Nc, int k,
                   int D, int64_t ldS, itr_stream_t stream);

/* ---- a5/a10: ContrastiveLoss / TripletLoss (Objectives.py:76-115, :492-517) ------------
 * loss = sum_i red_{j!=i}[m + S_ij - S_ii]_+ + sum_j red_{i!=j}[m + S_ij - S_jj]_+,
 * red = max if max_violation else sum.  fwd writes loss[0] and, for the backward, the arg of
 * the row / column maxima (row_arg[B], col_arg[B]; unused for the sum form, may be NULL).
 * bwd writes dS[B,B] = grad_loss[0] * dloss/dS. */
int itr_hinge_maxviol_fwd(const float *S, int B, int64_t ldS, float margin, int max_violation,
                          float *loss, int32_t *row_arg, int32_t *col_arg, float *cost_ws /* [2B] */,
                          itr_stream_t stream);
int itr_hinge_maxviol_bwd(const float *S, int B, int64_t ldS, float margin, int max_violation,
                          const int32_t *row_arg, const int32_t *col_arg, const float *grad_loss,
                          float *dS, int64_t lddS, itr_stream_t stream);

/* ---- a6: xattn_score_t2i / xattn_score_i2t (Objectives.py:329-476) ---------------------
 * img [Ni,R,D]; words [n_rows,D] with caption c owning rows cap_off[c] .. cap_off[c]+len[c]-1
 * (padded (Nc,L,D) input: cap_off[c] = c*L).  S[i,c] written with row stride ldS.
 * mode: 0 t2i, 1 i2t.  norm: 0 clipped_l2norm, 1 l2norm, 2 softmax, 3 no_norm, 4 clipped,
 * 5 l1norm, 6 clipped_l1norm.  agg: 0 LogSumExp, 1 Max, 2 Sum, 3 Mean.
 * Three steps:
 *  1. itr_scan_plan_tiles (pure CPU): whole captions are bin-packed (tiles filled exactly where the lengths allow it) into column
 *     tiles of <= ITR_SCAN_NT words: cap_order[Nc] lists the caption ids tile by tile and
 *     tile_begin[n_tiles+1] indexes it; the caller copies both to the device.
 *  2. itr_scan_prepare: once per (image block, caption set, mode) -- re-packs the word embeddings tile
 *     by tile into the workspace and precomputes region Gram matrices V_i V_i^T + word norms (t2i) or
 *     region norms + caption Gram matrices E_c E_c^T (i2t).
 *  3. itr_scan_xattn_scores: the fused MFMA kernel; needs only `img` and the prepared workspace and may
 *     be called repeatedly (other norm / agg / lambda) without preparing again.
 * workspace: itr_scan_workspace_bytes(Ni, R, n_rows, Nc, n_tiles, D) bytes. */
#define ITR_SCAN_NT 64
int itr_scan_plan_tiles(const int32_t *len_host, int64_t Nc, int nt, int32_t *tile_begin_host,
                        int32_t *cap_order_host, int64_t *n_tiles);
size_t itr_scan_workspace_bytes(int64_t Ni, int R, int64_t n_rows, int64_t Nc, int64_t n_tiles, int D);
int itr_scan_prepare(const float *img, const float *words, const int64_t *cap_off,
                     const int32_t *cap_len, const int32_t *tile_begin_dev,
                     const int32_t *cap_order_dev, int64_t n_tiles, int64_t Ni, int64_t Nc,
                     int64_t n_rows, int R, int D, int mode, void *workspace, size_t workspace_bytes,
                     itr_stream_t stream);
int itr_scan_xattn_scores(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows,
                          int R, int D, int mode, int norm, int agg, float lambda_softmax,
                          float lambda_lse, float *S, int64_t ldS, void *workspace,
                          size_t workspace_bytes, itr_stream_t stream);

/* Opt-in study variant (SURVEY.md 8d "bf16-in / fp32-acc variant reported separately"): the same scores
 * with the region x word dot products on the bf16 matrix core from split operands, x = hi + lo, hi.hi + hi.lo + lo.hi,
 * fp32 accumulation ("bf16x3", ~1e-6 of the fp32 result on the unit-norm operands of this path); the epilogue is the
 * fp32 one.  bf16_workspace: itr_scan_bf16_workspace_bytes(Ni, R, n_tiles, D) bytes of scratch for the split planes.
 * split_format 0: bf16 planes (8 + 8 mantissa bits).  split_format 1 ("fp16x3"): fp16 planes (11 + 11 bits, lo scaled by 2^11,
 * cross terms in a second accumulator): ~1e-7 of the fp32 result, needs |x| <= 65504 (unit-norm rows on this path). */
size_t itr_scan_bf16_workspace_bytes(int64_t Ni, int R, int64_t n_tiles, int D);
int itr_scan_xattn_scores_bf16x3(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows, int R, int D,
                                 int mode, int norm, int agg, float lambda_softmax, float lambda_lse, float *S,
                                 int64_t ldS, void *workspace, size_t workspace_bytes, void *bf16_workspace,
                                 size_t bf16_workspace_bytes, int split_format, itr_stream_t stream);

/* ---- a11: BERT building blocks (itr/modalmodule/bert.py:113-358); the dense layers use itr_gemm_nt (act 4 =
 * erf-GELU :29-34, act 2 = tanh pooler :299-302).
 * itr_bert_embed_ln : out[b,t,:] = LN(word[ids] + pos[t] + type[type_ids]) (:127-157); type_ids may be NULL (zeros).
 * itr_add_layernorm : out = LN(x + residual) with the TF-style epsilon inside the sqrt (:113-126); residual may be NULL.
 * itr_mha_small     : per (sequence, head): softmax(Q K^T * scale + (1 - mask) * -10000) V for L <= 64 (:185-207);
 *                     q/k/v are [B*L, ld*] views (heads contiguous, head h at column h*dk); mask [B*L] of 0/1 or NULL.
 * itr_relu_maxpool  : out[b,c] = max_{t<valid} relu(x[b,t,c])  (SAEM conv head, TextEncoder.py:148-149). */
int itr_bert_embed_ln(const int64_t *ids, const int64_t *type_ids, const float *word_emb, const float *pos_emb,
                      const float *type_emb, const float *gamma, const float *beta, float *out, int64_t B,
                      int L, int H, int64_t vocab, int max_pos, int type_vocab, float eps, itr_stream_t stream);
int itr_add_layernorm(const float *x, const float *residual, const float *gamma, const float *beta, float *out,
                      int64_t rows, int H, float eps, itr_stream_t stream);
int itr_mha_small(const float *q, const float *k, const float *v, int64_t ldq, int64_t ldk, int64_t ldv,
                  const float *mask, float *out, int64_t ldo, int64_t B, int L, int heads, int dk, float scale,
                  itr_stream_t stream);
int itr_relu_maxpool(const float *x, float *out, int64_t ldo, int64_t B, int L, int C, int valid,
                     itr_stream_t stream);

/* ---- a13: CAMERA helpers (itr/modalmodule/camera_.py; ImgEncoder.py:355-433; TextEncoder.py:162-197) ----
 * itr_gemm_nt_acc     : C = act(C + A B^T + bias)  -- sums the taps of the dilated Conv1d summarisation (:100-103)
 * itr_mul_rows        : out[r,c] = a[r,c] * b[r*ldb + c]   (position gating :75, query/key gates :38-40)
 * itr_affine_cols     : out = act(x*scale[c] + shift[c]) + residual  (eval BatchNorm1d folded; scale/shift/residual may be NULL)
 * itr_agsa_gate       : the query / key gate of GatedQueryAttLayer.forward (:36-44) in one kernel: rows = positions x heads, q / k
 *                       [rows, dk] (dk = 16 or 32; other head sizes: compose itr_gemm_nt + itr_mul_rows),  G = fc_q(q) * fc_k(k),
 *                       M = sigmoid(fc_g(G)) [rows, 2 dk],  q_out = q * M[:, :dk],  k_out = k * M[:, dk:]  (outputs may alias inputs)
 * itr_camera_posenc   : absoluteEncode(boxes, imgs_wh) -> [B*R, 6]   (:118-128)
 * itr_camera_summarize: softmax over regions of smry[B,R,k], L^T X, F.normalize -> [B,k,D]  (ImgEncoder.py:385-389) */
int itr_gemm_nt_acc(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                    int64_t ldc, int64_t M, int64_t N, int64_t K, int act, itr_stream_t stream);
/* C = act(R + A B^T + bias): itr_gemm_nt_acc with the summand read from its own matrix (Rs_GCN's `W(y) + v`, vsrn_.py:64-67: no
 * copy of v into the output first).  Same arithmetic order as itr_gemm_nt_acc on a copy: bit-identical. */
int itr_gemm_nt_residual(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, const float *R, int64_t ldr,
                         float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, itr_stream_t stream);
int itr_mul_rows(const float *a, const float *b, int64_t ldb, float *out, int64_t R, int C, itr_stream_t stream);
int itr_affine_cols(const float *x, const float *scale, const float *shift, const float *residual, float *out,
                    int64_t R, int C, int act, itr_stream_t stream);
int itr_agsa_gate(const float *q, const float *k, int64_t rows, int dk, const float *Wq, const float *bq, const float *Wk,
                  const float *bk, const float *Wg, const float *bg, float *q_out, float *k_out, itr_stream_t stream);
int itr_camera_posenc(const float *boxes, const float *imgs_wh, float *out, int64_t B, int R, itr_stream_t stream);
int itr_camera_summarize(const float *smry, const float *X, float *out, int64_t B, int R, int k, int D,
                         itr_stream_t stream);

/* ---- a7: EncoderSimilarity.forward (SGRAF; itr/modalmodule/Fusionmodule.py:373-664), eval mode ----
 * img [Ni,36,D] (l2-normalised regions), words [n_rows,D] with the caption layout of the SCAN entry points
 * (cap_off / cap_len / tile plan from itr_scan_plan_tiles on the word lengths).  module: 0 = SAF, 1 = SGR.
 * weights: device pointers named after the reference's state_dict (row-major [out,in] like nn.Linear).
 * S[i,c] in (0,1).  max_len = longest caption (<= 63).  workspace: itr_sgraf_workspace_bytes(...).
 * node_group_*: optional plan for SGR's fused graph steps (GraphReasoning x sgr_step, Fusionmodule.py:564-597, in one
 * workgroup per group of captions): itr_sgr_plan_node_groups (pure CPU) bins whole captions by the node count of their
 * graph (cap_len[c] + 1) into groups of <= 16 captions and <= 64 node rows (small_rows = 64: one workgroup per CU, the
 * default) or, for captions of <= 31 words, <= 32 node rows (small_rows = 32: two workgroups per CU; same scores); pass
 * its group_begin / group_order arrays (device copies) and the group count.  Any plan with <= 16 captions and <= 64 node
 * rows per group is accepted; a group that breaks these bounds is not scored and its captions' columns of S are filled
 * with NaN.  NULL / 0: the steps run one kernel chain per step (also taken for sim_dim != 256). */
typedef struct {
    /* v_global_w (VisualSA :464-507) */
    const float *v_loc_w, *v_loc_b, *v_loc_bn_w, *v_loc_bn_b, *v_loc_bn_mean, *v_loc_bn_var; /* Linear[D,D], BN(36) */
    const float *v_glo_w, *v_glo_b, *v_glo_bn_w, *v_glo_bn_b, *v_glo_bn_mean, *v_glo_bn_var; /* Linear[D,D], BN(D)  */
    const float *v_com_w, *v_com_b;                                                          /* Linear[1,D]         */
    /* t_global_w (TextSA :519-559) */
    const float *t_loc_w, *t_loc_b, *t_glo_w, *t_glo_b, *t_com_w, *t_com_b;
    /* sim_tranloc_w / sim_tranglo_w [S,D], sim_eval_w [1,S] (:391-395) */
    const float *loc_w, *loc_b, *glo_w, *glo_b, *eval_w, *eval_b;
    /* SAF_module.attn_sim_w [1,S], SAF_module.bn (1 channel) (:608-611) */
    const float *saf_w, *saf_b, *saf_bn_w, *saf_bn_b, *saf_bn_mean, *saf_bn_var;
    /* SGR_module.sgr{k}.graph_query_w / graph_key_w / sim_graph_w [S,S] (:573-576) */
    const float *sgr_q_w[8], *sgr_q_b[8], *sgr_k_w[8], *sgr_k_b[8], *sgr_g_w[8], *sgr_g_b[8];
} itr_sgraf_weights;
int itr_sgr_plan_node_groups(const int32_t *cap_len_host, int64_t Nc, int small_rows /* 64 or 32 */,
                             int32_t *group_begin_host, int32_t *group_order_host, int64_t *n_groups);
/* image_block: images per block of the pair stage -- 0 = ITR_SGRAF_DEFAULT_IMAGE_BLOCK, else a multiple of 4 in [4, 64]; never more than
 * the call's image count rounded up to 4.  The workspace grows with it (5k x 25k: 64 -> SAF 38 GB, SGR 42 GB; 16 -> 13 / 14 GB) and
 * the scores do not depend on it: the caller picks it for the memory it has -- itr_sgraf_pick_image_block returns the largest of
 * 64 / 32 / 16 / 8 / 4 whose workspace fits max_workspace_bytes (and that size in *workspace_bytes), or ITR_ERR_UNSUPPORTED when even a
 * 4-image block does not fit (the reference's own loop is O(tile) in memory: Fusionmodule.py:406-451 scores one caption at a time).
 * flags: ITR_SGRAF_UNFUSED_STEPS = SGR's graph steps as the step-by-step kernel chain even when a node-group plan is given (the form
 * the fused kernel is cross-checked against; also what runs -- and what the workspace must be sized for -- when NO plan is passed);
 * ITR_SGRAF_NON_PERSISTENT = fused SGR with one workgroup per (image, group) instead of the persistent walk.  Same scores. */
#define ITR_SGRAF_DEFAULT_IMAGE_BLOCK 64
#define ITR_SGRAF_UNFUSED_STEPS 1
#define ITR_SGRAF_NON_PERSISTENT 2
size_t itr_sgraf_workspace_bytes(int64_t Ni, int64_t Nc, int64_t n_rows, int64_t n_tiles, int D, int S,
                                 int module, int image_block, int flags);
int itr_sgraf_pick_image_block(int64_t Ni, int64_t Nc, int64_t n_rows, int64_t n_tiles, int D, int S, int module, int flags,
                               size_t max_workspace_bytes, size_t *workspace_bytes /* host, may be NULL */);
int itr_sgraf_scores(const float *img, const float *words, const int64_t *cap_off, const int32_t *cap_len,
                     const int32_t *tile_begin_dev, const int32_t *cap_order_dev, int64_t n_tiles,
                     int64_t Ni, int64_t Nc, int64_t n_rows, int max_len, int R, int D, int S, int module,
                     int sgr_step, const itr_sgraf_weights *w, const int32_t *node_group_begin_dev,
                     const int32_t *node_group_order_dev, int64_t n_node_groups, int image_block, int flags, float *S_out,
                     int64_t ldS, void *workspace, size_t workspace_bytes, itr_stream_t stream);

/* diagnostics (tools/): occupancy of the SCAN kernel as reported by the HIP runtime */
int itr_debug_scan_occupancy(int *blocks_per_cu, int *lds_bytes);
/* diagnostics (bench.py's sustained-clock probe): one INSTRUMENTED launch of the SCAN kernel with the arguments of
 * itr_scan_xattn_scores.  `scratch` [Ni, ld_scratch >= Nc + 64] is not a score matrix afterwards: its first 64 bytes hold eight
 * uint64 counters summed over all workgroups -- [0..6] shader-clock cycles (s_memtime) per phase, [7] 100 MHz ticks
 * (s_memrealtime); the rest is undefined.  sustained clock [MHz] = 100 * sum(c[0..6]) / c[7]. */
int itr_debug_scan_clock_probe(const float *img, int64_t n_tiles, int64_t Ni, int64_t Nc, int64_t n_rows, int R, int D, int mode,
                               int norm, int agg, float lambda_softmax, float lambda_lse, float *scratch, int64_t ld_scratch,
                               void *workspace, size_t workspace_bytes, itr_stream_t stream);

/* ---- a17: i2t / t2i ranker (itr/metricmodule/evaluation.py:156-222) --------------------
 * Sort-free: rank(query, gt) = #{k: S_k > S_gt} + #{k > gt: S_k == S_gt}; i2t takes the min
 * over the im_div GT captions im_div*i .. im_div*i+im_div-1; t2i's GT image is j / im_div.
 * top1 = argmax (highest index on ties).  S is the LOCAL row block [n_rows_local, Nc] holding
 * global image rows row0 .. row0+n_rows_local-1 (row0 = 0 for a single GPU).
 *   i2t_rank/top1 [n_rows_local]  -- complete (rows are local).
 *   t2i needs the GT score of every caption: pass s_gt[Nc] (gathered over ranks); the kernel
 *   ACCUMULATES partial counts into t2i_rank[Nc] (int32, zero it first) and max-reduces
 *   t2i_best[Nc] (uint64 key = ordered(score) << 32 | row; zero it first), so a sum /
 *   max all-reduce over ranks completes them.  itr_rank_gather_gt fills s_gt for the GT
 *   rows this rank owns (others left untouched; buffer pre-filled with -inf + max-reduce).
 * ONE pass over S serves both directions (every element is read once: 4 bytes per pair), between a small preparation and a
 * small finishing kernel.  workspace: itr_rank_workspace_bytes(n_rows_local, Nc) bytes, 8-byte aligned (per-row GT keys and running
 * top-1 keys, the GT scores when the call gathers them itself).
 * s_gt may be NULL when the block holds EVERY ground-truth row (row0 = 0 and n_rows_local * im_div >= Nc: the single-GPU call):
 * the scores are then read from S itself -- no itr_rank_gather_gt launch.  flags: ITR_RANK_INIT_COLUMNS = zero t2i_rank / t2i_best
 * inside the call (single-call use; leave it out when several row blocks -- or several ranks -- accumulate into them).
 * Order of scores: -0.0 == +0.0; NaN sorts as the LARGEST value (np.argsort's order: NaN last ascending, first after
 * [::-1]), ties -- also among NaN / +inf -- go to the higher index; the float64 entry points use the same rule. */
#define ITR_RANK_INIT_COLUMNS 1
int itr_rank_gather_gt(const float *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                       int im_div, float *s_gt, itr_stream_t stream);
size_t itr_rank_workspace_bytes(int64_t n_rows_local, int64_t Nc);
int itr_rank_counts(const float *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                    int im_div, const float *s_gt, int32_t *i2t_rank, int32_t *i2t_top1,
                    int32_t *t2i_rank, uint64_t *t2i_best, int flags, void *workspace, size_t workspace_bytes, itr_stream_t stream);
/* float64 matrices.  The reference ranks the float64 array cal_sims returns (evaluation.py:169, :209), and
 * evalrank_ensemble averages two models' matrices in float64 before ranking (evaluation.py:380, :398): such
 * scores are not fp32-representable, so they are counted in float64 -- same counts, same tie rule, index-exact.
 * t2i_best_key[Nc] (zero it first) is max-reduced to the 64-bit ordered key of every column's best score;
 * itr_rank_t2i_top1_f64 then max-reduces t2i_top1[Nc] (int32, pre-filled with -1) to the highest global row
 * holding that key (second pass over S; with several ranks: max all-reduce the keys between the two calls and
 * t2i_top1 after). */
int itr_rank_gather_gt_f64(const double *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                           int im_div, double *s_gt, itr_stream_t stream);
int itr_rank_counts_f64(const double *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                        int im_div, const double *s_gt, int32_t *i2t_rank, int32_t *i2t_top1,
                        int32_t *t2i_rank, uint64_t *t2i_best_key, itr_stream_t stream);
int itr_rank_t2i_top1_f64(const double *S, int64_t ldS, int64_t row0, int64_t n_rows_local, int64_t Nc,
                          const uint64_t *t2i_best_key, int32_t *t2i_top1, itr_stream_t stream);
/* host-side summary (evaluation.py:181-185): out5 = r1, r5, r10, medr, meanr. */
int itr_recall_from_ranks(const int32_t *ranks_host, int64_t n, double *out5);

/* ---- a14: the training step  model.train_emb (itr/modalmodule/Models.py:198-225, :115-145): forward -> loss ->
 * backward -> clip_grad_norm_(2.0) -> Adam.  Backward contractions are itr_gemm_nt on transposed operands;
 * these are the pieces around them. ------------------------------------------------------------------------- */
/* z = x / (||x|| + eps) keeping the row norms (utils.py:10-15), and its backward
 *   dx = dz / (n + eps) - z (z . dz) / n. */
int itr_l2norm_fwd_save(const float *x, float *z, float *norms, int64_t rows, int dim, float eps, itr_stream_t stream);
int itr_l2norm_bwd(const float *dz, const float *z, const float *norms, float *dx, int64_t rows, int dim, float eps,
                   itr_stream_t stream);
/* out[c, r] = in[r, c] (row-major [rows, cols] -> [cols, rows]). */
int itr_transpose2d(const float *in, float *out, int64_t rows, int64_t cols, itr_stream_t stream);
/* out[c] (+)= sum_r x[r, c], fixed summation order (bias gradients). */
size_t itr_colsum_workspace_bytes(int64_t rows, int64_t cols);
int itr_colsum(const float *x, float *out, int64_t rows, int64_t cols, int accumulate, void *workspace,
               size_t workspace_bytes, itr_stream_t stream);
/* C = act(A B^T + bias) with the K range cut into slices when the output has too few 128 x 128 tiles to fill the chip (M N small, K long:
 * CAMERA's dilated convolutions as GEMMs); slices added in a fixed order.  TRAINING TAPE ONLY: a row's bits depend on the slice count,
 * hence on the shape of the call -- the evaluation keeps itr_gemm_nt.  Falls back to itr_gemm_nt when no split is chosen.  The slice
 * count aims at two resident workgroups per CU (512 / tiles, at most 8, slices of >= 128 k).  M <= 128 rows (ABI 30: a decoder step's
 * layers) take the 16-column-strip kernel over K slices and one pass that adds the slices, the bias and the activation; the workspace
 * for that is 16 M N floats. */
size_t itr_gemm_nt_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
int itr_gemm_nt_splitk(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M,
                       int64_t N, int64_t K, int act, void *workspace, size_t workspace_bytes, itr_stream_t stream);
/* Weight gradient of a dense layer, dW = dY^T X:  C[P, Q] (+)= A[R, P]^T B[R, Q] (row-major, the REDUCED index is the row of
 * both operands: no transposed copies).  The rows are split into slices whose partial products (workspace) are added in slice
 * order: deterministic.  accumulate != 0 adds to C.  colsum_a (may be NULL) receives sum_r A[r, :] -- the bias gradient, from the
 * operand tiles the product loads anyway.  Exact fp32 MFMA (csrc/gemm_tn.hip). */
size_t itr_gemm_tn_workspace_bytes(int64_t R, int64_t P, int64_t Q);
int itr_gemm_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, int64_t R, int64_t P, int64_t Q,
                int accumulate, float *colsum_a, void *workspace, size_t workspace_bytes, itr_stream_t stream);
/* `batch` problems of one shape in one launch (operand z at A + z batch_a ...), each C_z = A_z^T B_z reduced in one pass. */
int itr_gemm_tn_batched(const float *A, int64_t lda, int64_t batch_a, const float *B, int64_t ldb, int64_t batch_b, float *C, int64_t ldc,
                        int64_t batch_c, int64_t R, int64_t P, int64_t Q, int64_t batch, itr_stream_t stream);

/* ---- a7 / a14: SGRAF.train_emb's similarity module on ALL pairs of the batch at once (EncoderSimilarity.forward in training mode and its
 * backward: itr/modalmodule/Fusionmodule.py:406-451, SCAN_attention :632-664, GraphReasoning :579-586, AttentionFiltration :613-618,
 * TextSA :549-564; caller Models.py:518-546).  Ragged IMAGE-MAJOR matrices over B images and C captions with T words in total
 * (cap_off[C + 1] = int32 word offsets, device):
 *   local rows b T + t;   node rows b (T + C) + cap_off[c] + c + j (j = 0 the global alignment, 1 .. W_c the words);   pair rows b C + c.
 * The dense layers between these stages are itr_gemm_nt / itr_gemm_tn; csrc/sgraf_train.hip holds the stages themselves:
 *   attn    P[(b,t), r] = softmax_r(smooth * l2norm_w(LeakyReLU_0.1(A[(b,r), t])))      A [B R, ldA] = regions . words^T
 *   ctx     X[(b,t), :] = (l2norm(sum_r P r img[b,r,:]) - words[t,:])^2,  cnorm = the context norms;  backward: d ctx [B T, D], d words
 *           (summed over the images; workspace = itr_sgt_ctx_bwd_workspace_bytes), then d P = itr_sgt_dp and d regions = itr_gemm_tn_batched
 *   pair_sqdiff  X[(b,c), :] = (img_glo[b] - cap_glo[c])^2
 *   nodes   scatter of the global / local alignment rows into the node matrix (backward != 0: the inverse copy)
 *   graph   E = softmax_r(q_v . k_r), Z_v = sum_r E[v,r] x_r per pair; E saved at b e_off[C] + e_off[c] (e_off = prefix sums of n_c^2)
 *   segbn   BatchNorm1d(1) with the batch statistics of ONE caption's B (W_c + 1) attention logits (the reference calls it per caption)
 *   saf_pool  sigmoid -> l1norm over the nodes of a pair -> weighted node sum
 *   seg_mean / seg_smry  caption mean; softmax(logit over the words of a caption) . words      (TextSA on packed captions)
 * Limits (ITR_ERR_UNSUPPORTED beyond; the caller then takes the grouped path): D % 4 == 0, D <= 2048, R <= 64, S % 4 == 0, LDS per pair. */
int itr_sgt_attn_fwd(const float *A, int64_t ldA, const int32_t *cap_off, int B, int C, int T, int R, int Wmax, float smooth, float eps,
                     float *P, itr_stream_t stream);
int itr_sgt_attn_bwd(const float *A, int64_t ldA, const float *P, const float *dP, const int32_t *cap_off, int B, int C, int T, int R,
                     int Wmax, float smooth, float eps, float *dA, itr_stream_t stream);
int itr_sgt_ctx_fwd(const float *P, const float *img, const float *words, int B, int T, int R, int D, float eps, float *X, float *cnorm,
                    itr_stream_t stream);
size_t itr_sgt_ctx_bwd_workspace_bytes(int B, int T, int D);
int itr_sgt_ctx_bwd(const float *P, const float *img, const float *words, const float *cnorm, const float *dX, int B, int T, int R, int D,
                    float eps, float *dctx, float *dwords, void *workspace, size_t workspace_bytes, itr_stream_t stream);
int itr_sgt_dp(const float *dctx, const float *img, int B, int T, int R, int D, float *dP, itr_stream_t stream);
int itr_sgt_pair_sqdiff_fwd(const float *img_glo, const float *cap_glo, int B, int C, int D, float *X, itr_stream_t stream);
int itr_sgt_pair_sqdiff_bwd(const float *img_glo, const float *cap_glo, const float *dX, int B, int C, int D, float *dimg_glo,
                            float *dcap_glo, itr_stream_t stream);
int itr_sgt_nodes(float *glo, float *loc, float *nodes, const int32_t *cap_off, const int32_t *node_cap, int B, int C, int T, int S,
                  int backward, itr_stream_t stream);
/* row0_only != 0: only node 0 of every pair is a query (the last reasoning step, whose output the reference reads at [:, 0, :]):
 * q / Z / dq are pair rows [B C, S], E one row of n weights per pair at its node offset [B (T + C)]. */
int itr_sgt_graph_fwd(const float *q, const float *k, const float *x, const int32_t *cap_off, const int32_t *e_off, int B, int C, int T, int S,
                      int nmax, int row0_only, float *E, float *Z, itr_stream_t stream);
int itr_sgt_graph_bwd(const float *q, const float *k, const float *x, const float *E, const float *dZ, const int32_t *cap_off,
                      const int32_t *e_off, int B, int C, int T, int S, int nmax, int row0_only, float *dq, float *dk, float *dx,
                      itr_stream_t stream);
int itr_sgt_segbn_fwd(const float *a, const int32_t *cap_off, int B, int C, int T, const float *gamma, const float *beta, float eps, float *y,
                      float *mean, float *var, float *invstd, itr_stream_t stream);
int itr_sgt_segbn_bwd(const float *dy, const float *a, const int32_t *cap_off, int B, int C, int T, const float *gamma, const float *mean,
                      const float *invstd, float *da, float *dgamma_c, float *dbeta_c, itr_stream_t stream);
int itr_sgt_saf_pool_fwd(const float *y, const float *nodes, const int32_t *cap_off, int B, int C, int T, int S, int nmax, float eps, float *out,
                         itr_stream_t stream);
int itr_sgt_saf_pool_bwd(const float *y, const float *nodes, const float *dout, const int32_t *cap_off, int B, int C, int T, int S, int nmax,
                         float eps, float *dy, float *dnodes, itr_stream_t stream);
/* spread == 0: out[c] = sum (mean != 0: mean) of the caption's rows of in [T, D]; spread != 0: out[t] = in[caption of t] (/ W_c if mean) */
int itr_sgt_seg_mean(const float *in, const int32_t *cap_off, int C, int D, float *out, int spread, int mean, itr_stream_t stream);
int itr_sgt_seg_smry_fwd(const float *logit, const float *words, const int32_t *cap_off, int C, int D, int Wmax, float *p, float *out,
                         itr_stream_t stream);
int itr_sgt_seg_smry_bwd(const float *p, const float *words, const float *dout, const int32_t *cap_off, int C, int D, int Wmax, float *dlogit,
                         float *dwords, itr_stream_t stream);
/* nn.Embedding backward: dE[tokens[r], :] += dx[r, :] (atomic adds). */
int itr_embed_scatter_add(const int64_t *tokens, const float *dx, int64_t n_tok, int64_t V, int E, float *dE,
                          itr_stream_t stream);
/* out[r, :] = table[idx[r], :]  (nn.Embedding forward; the last-valid-step gather of TextEncoder.py:57-60).  Indices
 * outside [0, V) read row 0 and set *bad_flag (device int) to 1. */
int itr_gather_rows(const int64_t *idx, int64_t n, const float *table, int64_t V, int E, float *out, int *bad_flag,
                    itr_stream_t stream);
/* clip_grad_norm_ (Models.py:223-224): itr_sq_sum writes itr_sq_sum_blocks(n) partial sums of squares of one gradient
 * tensor; itr_clip_coef turns all partials of all tensors into coef_and_norm[0] = min(1, max_norm / (||g|| + 1e-6)),
 * coef_and_norm[1] = ||g|| (device floats). */
int itr_sq_sum_blocks(int64_t n);
int itr_sq_sum(const float *g, int64_t n, float *partials, itr_stream_t stream);
int itr_clip_coef(const float *partials, int64_t nparts, float max_norm, float *coef_and_norm, itr_stream_t stream);
/* torch.optim.Adam step on one tensor (no weight decay / amsgrad); the gradient is multiplied by *grad_scale_dev
 * (device float, may be NULL) first; `step` is the 1-based update count. */
int itr_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps,
                  int64_t step, const float *grad_scale_dev, itr_stream_t stream);

/* The same two for ALL parameter tensors of a step in one launch each (a step has up to 67 tensors).  table_dev: one 48-byte record per
 * tensor {float *p; const float *g; float *m, *v; int64 n; int32 first_blk, nblk} (device copy of a host table; [first_blk, first_blk +
 * nblk) = the tensor's itr_sq_sum_blocks(n) workgroups of itr_sq_sum_multi, partials in the same layout as per-tensor itr_sq_sum calls:
 * the norm is bit-identical); blk_tensor[b] = tensor of workgroup b.  itr_adam_step_multi has its own maps: one workgroup per 256
 * elements, blk_first[t] = first workgroup of tensor t. */
int itr_sq_sum_multi(const void *table_dev, const int32_t *blk_tensor_dev, int64_t n_blocks, float *partials, itr_stream_t stream);
int itr_adam_step_multi(const void *table_dev, const int32_t *blk_tensor_dev, const int32_t *blk_first_dev, int64_t n_blocks, float lr,
                        float beta1, float beta2, float eps, int64_t step, const float *grad_scale_dev, itr_stream_t stream);

/* ---- transformer towers under autograd (SAEM: TransformerMapping / BertMapping, ImgEncoder.py:324-350, TextEncoder.py:75-152,
 * bert.py:113-300); the dense layers are itr_gemm_nt on transposed operands as for every other backward pass ----------------
 * nn.Dropout: y = x * keep / (1 - p), keep(i) from a counter-based hash of (seed, offset + i).  Stateless: the backward pass
 * is the same call on dy.  (The random stream is not torch's; only the distribution is the reference's.) */
int itr_dropout(const float *x, float *y, int64_t n, float p, uint64_t seed, uint64_t offset, itr_stream_t stream);
/* BERTLayerNorm(x + residual) (bert.py:113-126; residual may be NULL) keeping z = x + residual, the row means and
 * 1 / sqrt(var + eps); backward: dz (for x and for the residual) and t = dy * xhat (dgamma = column sums of t, dbeta of dy). */
int itr_add_ln_fwd(const float *x, const float *residual, const float *gamma, const float *beta, float *z, float *out,
                   float *mean, float *rstd, int64_t rows, int H, float eps, itr_stream_t stream);
int itr_ln_bwd(const float *dy, const float *z, const float *mean, const float *rstd, const float *gamma, float *dz, float *t,
               int64_t rows, int H, itr_stream_t stream);
/* gelu (bert.py:104-110): dy == NULL -> out = gelu(x);  else out = dy * gelu'(x). */
int itr_gelu(const float *x, const float *dy, float *out, int64_t n, itr_stream_t stream);
/* BERTSelfAttention (bert.py:175-215) for short sequences (L <= 64, head size <= 64): per (sequence, head)
 *   P = softmax(Q K^T * scale + (1 - mask01) * -10000);  ctx = dropout(P) V.
 * q / k / v: row b * L + i, column head * dk + d, row stride ld (the three thirds of one fused QKV GEMM output).
 * P [B, heads, L, L] keeps the probabilities for the backward pass, which returns dq / dk / dv with row stride ldg. */
int itr_mha_train_fwd(const float *q, const float *k, const float *v, int64_t ld, const float *mask01, int64_t B, int L,
                      int heads, int dk, float scale, float p_drop, uint64_t seed, float *P, float *ctx, int64_t ldc,
                      itr_stream_t stream);
int itr_mha_train_bwd(const float *q, const float *k, const float *v, int64_t ld, const float *mask01, int64_t B, int L,
                      int heads, int dk, float scale, float p_drop, uint64_t seed, const float *P, const float *dctx,
                      int64_t ldc, float *dq, float *dk_out, float *dv, int64_t ldg, itr_stream_t stream);
/* F.relu + F.max_pool1d over all positions (TextEncoder.py:122-124): x [B, npos, C] -> out [B, C], arg [B, C] (first position of
 * the maximum, -1 when it is 0); backward scatters dy to the arg-max position. */
int itr_relu_maxpool_arg(const float *x, int64_t B, int npos, int C, float *out, int32_t *arg, itr_stream_t stream);
int itr_relu_maxpool_bwd(const float *dy, const int32_t *arg, int64_t B, int npos, int C, float *dx, itr_stream_t stream);
/* backward of torch.mean(x, 1): dx[b, r, :] = scale * dy[b, :]. */
int itr_bcast_mid(const float *dy, float *dx, int64_t B, int R, int F, float scale, itr_stream_t stream);

/* ---- a18: training-only auxiliary losses (csrc/aux_loss.hip) ---------------------------------------------------------------
 * AngularLoss.angular_loss (Objectives.py:262-290) on its three n x n products M1 = anchors others^T, M2 = positives others^T,
 * Q = anchors positives^T (row-major, ld = n):  x[i][j] = 4 ab (M1 + M2)[i][j] - 2 (1 + ab) Q[i][i] over the j != i;
 *   max_violation:  loss = sum_i log(1 + exp(max_j x[i][j]))                 (stat = max, arg = its first column)
 *   otherwise:      loss = mean_i log(1 + sum_j exp(x[i][j])), evaluated as t + log(exp(-t) + sum exp(x - t))   (stat = t, den)
 * row / stat / den / arg are [n] scratch the backward reads; loss and grad_loss are device scalars.  The backward writes the
 * gradient of M1 (== that of M2) to dM and that of Q (its diagonal) to dQ, both dense [n, n]. */
int itr_angular_fwd(const float *M1, const float *M2, const float *Q, int n, float angle_bound, int max_violation, float *loss,
                    float *row, float *stat, float *den, int32_t *arg, itr_stream_t stream);
int itr_angular_bwd(const float *M1, const float *M2, const float *Q, int n, float angle_bound, int max_violation, const float *stat,
                    const float *den, const int32_t *arg, const float *grad_loss, float *dM, float *dQ, itr_stream_t stream);
/* DiversityRegularization (Objectives.py:521-542): smry [B, R, K] -> sum_b || Sn_b^T Sn_b - I ||_F^2 with Sn = F.normalize(S, dim=1)
 * (columns over the R regions, eps 1e-12).  part: [B] scratch; loss / grad_loss device scalars; R*K + K*K + K <= 15360. */
int itr_diversity_fwd(const float *smry, int64_t B, int R, int K, float *part, float *loss, itr_stream_t stream);
int itr_diversity_bwd(const float *smry, int64_t B, int R, int K, const float *grad_loss, float *d_smry, itr_stream_t stream);

/* ---- CAMERA towers under autograd (camera_.py:14-114, ImgEncoder.py:355-389, TextEncoder.py:162-192, Fusionmodule.py:674-692):
 * the pieces between the dense layers ---------------------------------------------------------------------------------------
 * out = a * b (backward: two more calls). */
int itr_ew_mul(const float *a, const float *b, float *out, int64_t n, itr_stream_t stream);
/* dx = dy * act'(.) from the OUTPUT y of the activation: act 1 relu, 2 tanh, 3 sigmoid, 4 LeakyReLU(0.1). */
int itr_act_bwd(const float *y, const float *dy, float *dx, int64_t n, int act, itr_stream_t stream);
/* GatedQueryAttLayer gate (camera_.py:41-44): q' = q * M[:, :dk], k' = k * M[:, dk:] on [rows, dk] operands, M [rows, 2 dk]. */
int itr_gate_apply(const float *q, const float *k, const float *M, float *qo, float *ko, int64_t rows, int dk, itr_stream_t stream);
int itr_gate_apply_bwd(const float *q, const float *k, const float *M, const float *dqo, const float *dko, float *dq, float *dk_out,
                       float *dM, int64_t rows, int dk, itr_stream_t stream);
/* nn.BatchNorm1d in training mode on x [N, C]: batch mean / biased variance per column; keeps mean and 1 / sqrt(var + eps).
 * scratch: itr_bn_train_scratch_bytes(N, C) bytes (partial sums of the row slices; fixed summation order). */
size_t itr_bn_train_scratch_bytes(int64_t N, int C);
int itr_bn_train_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *invstd, int64_t N, int C,
                     float eps, void *scratch, itr_stream_t stream);
int itr_bn_train_bwd(const float *dy, const float *x, const float *mean, const float *invstd, const float *gamma, float *dx,
                     float *dgamma, float *dbeta, int64_t N, int C, void *scratch, itr_stream_t stream);
/* utils.l2norm with its default dim=1 on [B, R, D] (normalises ACROSS the R regions, ImgEncoder.py:378,384). */
int itr_l2norm_mid_fwd(const float *x, float *z, float *norms, int64_t B, int R, int D, float eps, itr_stream_t stream);
int itr_l2norm_mid_bwd(const float *dz, const float *z, const float *norms, float *dx, int64_t B, int R, int D, float eps,
                       itr_stream_t stream);
/* Multi-view summarisation (ImgEncoder.py:386-387): L = softmax(smry [B, R, K], dim=1); out [B, K, D] = L^T x.  scratch: B*R*K floats.
 * R, K <= 96.  The same contraction is SGRAF's SCAN_attention weighting (K = words, Fusionmodule.py:649-661) and the edge
 * aggregation of its graph reasoning (R = K = nodes, :581-584). */
int itr_smry_fwd(const float *smry, const float *x, float *L, float *out, int64_t B, int R, int K, int D, itr_stream_t stream);
int itr_smry_bwd(const float *x, const float *L, const float *dout, float *dx, float *dsmry, float *scratch, int64_t B, int R, int K,
                 int D, itr_stream_t stream);
/* C[b] (M x N) = op(A[b]) op(B[b]) for small dense row-major operands, op = identity / transpose (torch.bmm in GraphReasoning,
 * Fusionmodule.py:582-584, and its backward). */
int itr_bmm_small(const float *A, const float *B, float *C, int64_t batch, int M, int N, int K, int trans_a, int trans_b,
                  itr_stream_t stream);
/* MultiViewMatching (Fusionmodule.py:674-692) on top of the [Ni * k, Nc] view scores: S[i, c] = max_v T[i * k + v, c]. */
int itr_groupmax_fwd(const float *T, int64_t Ni, int k, int64_t Nc, float *S, int32_t *arg, itr_stream_t stream);
int itr_groupmax_bwd(const float *dS, const int32_t *arg, int64_t Ni, int k, int64_t Nc, float *dT, itr_stream_t stream);

/* ---- VSRN captioning branch under autograd (Fusionmodule.py:10-367, Objectives.py:138-158): single decoder steps ---------------
 * One nn.GRU step from gi = W_ih x + b_ih and gh = W_hh h + b_hh ([B, 3H], gate order r, z, n); gates [B, 3H] keeps r, z, n. */
int itr_gru_cell_fwd(const float *gi, const float *gh, const float *h, float *h_next, float *gates, int64_t B, int H,
                     itr_stream_t stream);
int itr_gru_cell_bwd(const float *dh_next, const float *gates, const float *gh, const float *h, float *dgi, float *dgh, float *dh,
                     int64_t B, int H, itr_stream_t stream);
/* y[b, n, :] = act(x[b, n, :] + v[b, :]) and its backward from the output (dx = dy act'(y), dv[b] = sum_n dx[b, n]); act 0 none, 1 relu,
 * 2 tanh, 3 sigmoid.  The decoder attention's linear1(cat(enc_out, h)) (Fusionmodule.py:136-140) = enc_out W_e^T + b (once per batch)
 * + h W_h^T (per step), added here. */
int itr_add_bcast_mid_act(const float *x, const float *v, float *y, int64_t B, int N, int H, int act, itr_stream_t stream);
int itr_add_bcast_mid_act_bwd(const float *y, const float *dy, float *dx, float *dv, int64_t B, int N, int H, int act, itr_stream_t stream);
/* The decoder attention's scores in one pass (ABI 30; Attention.forward, Fusionmodule.py:136-141: linear2(tanh(linear1(cat(enc, hidden))))):
 *   e[b, n] = sum_h w[h] tanh(x[b, n, h] + v[b, h])     x [B, N, H] = the encoder half of linear1 (+ bias), v [B, H] = the hidden half, w [H].
 * Backward from de [B, N], the tanh recomputed: dx[b, n, h] = de w[h] (1 - p^2), dv[b, h] = sum_n dx, dw_part[b, h] = sum_n de[b, n] p
 * (the caller sums dw_part [B, H] over b: itr_colsum).  B <= 65535 in the backward; rows of H % 4 == 0 floats must be 16-byte aligned. */
int itr_addattn_score(const float *x, const float *v, const float *w, float *e, int64_t B, int N, int H, itr_stream_t stream);
int itr_addattn_score_bwd(const float *x, const float *v, const float *w, const float *de, float *dx, float *dv, float *dw_part, int64_t B, int N,
                          int H, itr_stream_t stream);
/* loss[b] = -mask[b] * log_softmax(logits[b, :])[target[b]]  (F.log_softmax + NLLLoss(reduce=False) * mask); lse keeps the row
 * log-sum-exp for the backward pass dlogits = dloss[b] mask[b] (softmax - onehot). */
int itr_nll_logsoftmax_fwd(const float *logits, const int64_t *target, const float *mask, float *loss, float *lse, int64_t B, int V,
                           itr_stream_t stream);
int itr_nll_logsoftmax_bwd(const float *logits, const int64_t *target, const float *mask, const float *lse, const float *dloss,
                           float *dlogits, int64_t B, int V, itr_stream_t stream);

/* EncoderText (bi)GRU under autograd (TextEncoder.py:38-70): training forward that keeps the gate activations, and
 * backpropagation through time.  Same packed layout / sorting contract as itr_gru_fwd.  `out` [n_tok, D] is the RAW
 * sequence output ((fwd + bwd) / 2 for a bi-GRU); l2norm / last-step gather are separate differentiable steps.
 * Gradients: d_embed [V, E] is ACCUMULATED into (zero it first); d_w_* / d_b_* are overwritten.  Token ids: validated by the caller,
 * as for itr_gru_fwd.  The two directions of a bi-GRU run side by side (the reverse one on a per-device side stream, joined before
 * the call returns to the caller's stream order). */
size_t itr_gru_train_save_bytes(int64_t n_tok, int D, int bidirectional);
size_t itr_gru_train_workspace_bytes(int64_t n_tok, int64_t B, int E, int D);
int itr_gru_fwd_train(const int64_t *tokens, const int64_t *tok_off, const int32_t *len_dev, const int32_t *len_host,
                      int64_t B, int64_t n_tok, const float *embed, int64_t V, int E, int D, const float *w_ih,
                      const float *w_hh, const float *b_ih, const float *b_hh, const float *w_ih_rev,
                      const float *w_hh_rev, const float *b_ih_rev, const float *b_hh_rev, float *out, void *save,
                      size_t save_bytes, void *workspace, size_t workspace_bytes, itr_stream_t stream);
int itr_gru_bwd(const int64_t *tokens, const int64_t *tok_off, const int32_t *len_dev, const int32_t *len_host, int64_t B,
                int64_t n_tok, const float *embed, int64_t V, int E, int D, const float *w_ih, const float *w_hh,
                const float *w_ih_rev, const float *w_hh_rev, const void *save, const float *d_out, float *d_embed,
                float *d_w_ih, float *d_w_hh, float *d_b_ih, float *d_b_hh, float *d_w_ih_rev, float *d_w_hh_rev,
                float *d_b_ih_rev, float *d_b_hh_rev, void *workspace, size_t workspace_bytes, itr_stream_t stream);

/* SCAN t2i similarity of a TRAINING batch (xattn_score_t2i, Objectives.py:329-372, under autograd).
 *   A [Bi*R, ldA] = V E^T  raw dot products (itr_gemm_nt), G [Bi, R, R] = V_i V_i^T and enorm[n_tok] = ||e_w||
 *   (itr_scan_train_prepare);  captions packed: cap_off[Bc], cap_len[Bc] (any order), at most 96 words.
 *   R regions per image: 36 in every reference configuration (compile-time loop bounds); any other 1 <= R <= 100 runs the
 *   same kernels with run-time bounds (the evaluation entry point's path for images that are not 36 regions).  The backward
 *   holds one more R x W block in LDS: R > 36 together with max_len > 64 returns ITR_ERR_UNSUPPORTED.
 *   norm in {0 clipped_l2norm, 1 l2norm, 2 softmax, 3 no_norm, 4 clipped, 5 l1norm, 6 clipped_l1norm};
 *   agg in {0 LogSumExp, 1 Max, 2 Sum, 3 Mean}.
 * itr_scan_train_bwd (dS [Bi, Bc]) writes dA [Bi*R, ldA], per-pair Gram gradients dG_pairs [Bi, Bc, R, R] and
 * d_enorm_pairs [Bi, ldA]; the caller finishes with GEMMs  dV = dA E,  dE = dA^T V  and itr_scan_train_finish, which
 * ADDS (sum_c dG + its transpose) V_i to dV and  colsum_i(d_enorm_pairs) e / ||e||  (d_enorm [n_tok]) to dE; it sums the per-pair
 * Gram gradients IN PLACE (into the slot of caption 0 of every image: dG_pairs is scratch of the backward pass, not an output). */
int itr_scan_train_prepare(const float *V, const float *E, int64_t Bi, int64_t n_tok, int R, int D, float *G,
                           float *enorm, itr_stream_t stream);
int itr_scan_train_fwd(const float *A, int64_t ldA, const float *G, const float *enorm, const int64_t *cap_off,
                       const int32_t *cap_len, int64_t Bi, int64_t Bc, int64_t n_tok, int R, int D, int max_len, int norm,
                       int agg, float lambda_softmax, float lambda_lse, float *S, itr_stream_t stream);
int itr_scan_train_bwd(const float *A, int64_t ldA, const float *G, const float *enorm, const int64_t *cap_off,
                       const int32_t *cap_len, int64_t Bi, int64_t Bc, int64_t n_tok, int R, int D, int max_len, int norm,
                       int agg, float lambda_softmax, float lambda_lse, const float *dS, float *dA, float *dG_pairs,
                       float *d_enorm_pairs, itr_stream_t stream);
int itr_scan_train_finish(float *dG_pairs, int64_t Bi, int64_t Bc, const float *V, const float *E,
                          const float *enorm, const float *d_enorm, int64_t n_tok, int R, int D, float *dV, float *dE,
                          itr_stream_t stream);

/* SCAN i2t similarity of a TRAINING batch (xattn_score_i2t, Objectives.py:376-417, under autograd): regions attend over
 * the words of a caption.  A as above; H = packed caption Gram matrices (caption c: W_c x W_c floats at H + h_off[c],
 * h_total = sum W_c^2) and vnorm[Bi*R] = ||v_r|| (itr_scan_train_i2t_prepare; R as above, 1..100).  itr_scan_train_i2t_bwd writes dA,
 * per-pair dH partials dH_pairs [Bi, h_total] and d||v|| partials d_vnorm_pairs [Bc, Bi*R]; the caller sums them over
 * images / captions (itr_colsum), runs  dV = dA E,  dE = dA^T V  and itr_scan_train_i2t_finish, which ADDS
 * (dH_c + dH_c^T) E_c to dE and  d||v|| v / ||v||  to dV. */
int itr_scan_train_i2t_prepare(const float *V, const float *E, const int64_t *cap_off, const int32_t *cap_len,
                               const int64_t *h_off, int64_t Bi, int64_t Bc, int R, int D, float *H, float *vnorm,
                               itr_stream_t stream);
int itr_scan_train_i2t_fwd(const float *A, int64_t ldA, const float *H, const int64_t *h_off, const float *vnorm,
                           const int64_t *cap_off, const int32_t *cap_len, int64_t Bi, int64_t Bc, int64_t n_tok, int R,
                           int D, int max_len, int norm, int agg, float lambda_softmax, float lambda_lse, float *S,
                           itr_stream_t stream);
int itr_scan_train_i2t_bwd(const float *A, int64_t ldA, const float *H, const int64_t *h_off, int64_t h_total,
                           const float *vnorm, const int64_t *cap_off, const int32_t *cap_len, int64_t Bi, int64_t Bc,
                           int64_t n_tok, int R, int D, int max_len, int norm, int agg, float lambda_softmax,
                           float lambda_lse, const float *dS, float *dA, float *dH_pairs, float *d_vnorm_pairs,
                           itr_stream_t stream);
int itr_scan_train_i2t_finish(const float *dH, const int64_t *h_off, const int64_t *cap_off, const int32_t *cap_len,
                              int64_t Bc, const float *E, const float *V, const float *vnorm, const float *d_vnorm,
                              int64_t Bi, int R, int D, float *dV, float *dE, itr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ITR_HIP_H */
