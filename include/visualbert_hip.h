/* visualbert_hip.h -- C ABI of libvisualbert_hip.so (gfx950 / MI355X).
 *
 * The reference (uclanlp/visualbert) is 100 % Python and has NO FFI / plugin / operator registry
 * (SURVEY.md section 8b); its only native-op seam is the import-time class substitution
 *   try: from apex.normalization.fused_layer_norm import FusedLayerNorm as BertLayerNorm
 *   (visualbert/pytorch_pretrained_bert/modeling.py:158-160)
 * and the optimizer swap under fp16 (visualbert/models/model_wrapper.py:118-134).  This header is
 * therefore the ABI a maintainer would bind (ctypes) to replace the ATen op
 * sequences listed next to each entry point.  Paths below are relative to
 * /root/reference/visualbert/pytorch_pretrained_bert/ unless they start with models/.
 *
 * Conventions
 *   - plain pointers + sizes; all pointers are DEVICE pointers unless named host_*; no torch types.
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (no sync, no
 *     allocation); the library owns no persistent device memory.
 *   - return 0 on success, negative VB_ERR_* otherwise; nothing throws across the ABI.
 *   - dtype: VB_F32 or VB_BF16 selects the storage type "T" of activations / GEMM operands;
 *     statistics, losses, parameters' master copies and parameter gradients are always fp32.
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 *   - re-entrant per stream: entry points keep no process-wide tuning state.  Launch options are attached to a STREAM
 *     (vb_stream_set_opts) and read by the calls enqueued on that stream only; the RCCL communicator is an object the
 *     caller owns (vb_comm_*); the optional launch timing is attached to a stream as well (vb_stream_profile).  Knobs that
 *     change results or exist only for kernel analysis are NOT in this header: they live in visualbert_hip_dev.h and exist
 *     only in libvisualbert_hip_dev.so (built with -DVB_DEV_KNOBS).
 */
#ifndef VISUALBERT_HIP_H
#define VISUALBERT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VB_F32 = 0, VB_BF16 = 1,
       /* GEMM entry points (vb_gemm, vb_wgrad_grouped, vb_bert_layer_*) and attention (vb_attn_*, which splits its fp32
        * tensors on chip): "bf16x3" split-operand mode.  Activations,
        * gradients and epilogue operands are fp32; each GEMM operand is handed over SPLIT (vb_split_bf16): a row of leading
        * dimension ld holds a bf16 hi plane in columns [0, ld/2) and a bf16 lo plane in [ld/2, ld), x = hi + lo to ~2^-17
        * relative.  The kernels form hi.hi + lo.hi + hi.lo on the bf16 matrix pipe with fp32 accumulation (the lo.lo term,
        * ~2^-18 relative, is dropped): fp32-class results (BERT-base logits within ~1e-5 of the fp32 reference, the
        * north-star asks 1e-3) at three bf16 MFMA passes instead of the ~16x slower fp32-input MFMA. */
       VB_BF16X3 = 2 };
enum { VB_KCONTIG = 0, VB_KSTRIDED = 1 };
enum { VB_ACT_NONE = 0, VB_ACT_GELU = 1, VB_ACT_TANH = 2, VB_ACT_GELU_GRAD = 3,
       VB_ACT_GELU_SAVE_GRAD = 4, VB_ACT_MUL_AUX = 5 };

/* library / build identification: returns a static string such as "visualbert_hip gfx950 r1" */
const char* vb_version(void);

/* ------------------------------------------------------------------------------------------------
 * Per-stream launch options (all zero = defaults).  They select among kernels that compute the SAME result; they
 * never change numerics beyond summation order.  Calls on other streams are unaffected (tests/test_kernels.py::
 * test_stream_options_do_not_leak_across_streams).
 *   persistent_workgroups: workgroups launched by the persistent GEMM kernels; 0 = one per compute unit.  A
 *                          data-parallel caller lowers it while RCCL kernels are resident (parallel.py).
 *   nt_kernel: pins the K-contiguous x K-contiguous bf16 GEMM kernel to one the dispatcher could have chosen itself (for
 *              reproducible summation order, or an A/B run); 0 = chosen from the shape; 14 / 22 / 24 / 42 = two-barrier 64x128 (four LDS stages) / 128x128 (two / four) / 256x128
 *              tiles; 81 = persistent 256x256 tile; 90 = 256x128 tiles, two workgroups per compute unit; 1 = the generic
 *              register-staged kernel.  Anything else is VB_ERR_ARG: experiment arms and the vendor-library yardstick exist in
 *              the developer library only (include/visualbert_hip_dev.h).  This library owns no device memory.
 *   attn_two_pass: 1 = two-pass attention backward even where the one-pass kernel applies.
 *   reserved: must be 0 (VB_ERR_ARG otherwise).
 * vb_stream_set_opts(stream, NULL) forgets the stream's entry (call it before destroying a stream).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vb_stream_opts {
    int persistent_workgroups;
    int nt_kernel;
    int attn_two_pass;
    int reserved;
} vb_stream_opts;
int vb_stream_set_opts(void* stream, const vb_stream_opts* opts);
int vb_stream_get_opts(void* stream, vb_stream_opts* out);

/* ------------------------------------------------------------------------------------------------
 * Caller-owned scratch for in-launch reductions, per stream (round 6).  Small GEMMs -- per-GPU batches of 8-16, the regime of the
 * reference's own configs (visualbert/configs/vqa/coco-pre-train.json:17 over 8 GPUs, models/train.py:146) -- have fewer output tiles
 * than the chip has compute units; vb_gemm then cuts the reduction (K) of a long-K problem into slices that run on different
 * compute units, each slice leaves its fp32 partial tile in this buffer, and the LAST slice to arrive (a device-scope ticket) sums the
 * slices IN SLICE ORDER (deterministic: the result does not depend on which slice came last) and runs the ordinary epilogue.
 * The library never allocates device memory: without a registered buffer that form is simply not chosen.
 *   scratch: device memory, 256-byte aligned, >= VB_SCRATCH_MIN_BYTES; the first 16 KB are arrival counters (zeroed HERE, on `stream`;
 *            every kernel leaves them zero), the rest holds the partial tiles.  The buffer belongs to the stream: launches on ONE stream
 *            are ordered, two streams must not share a buffer.  It must stay alive until the stream's work has drained;
 *   vb_stream_set_scratch(stream, NULL, 0) forgets the entry.
 * ---------------------------------------------------------------------------------------------- */
#define VB_SCRATCH_MIN_BYTES (1 << 20)
int vb_stream_set_scratch(void* stream, void* scratch, int64_t bytes);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue.   C[M,N] = epi( alpha * sum_k Aop[m,k] * Bop[n,k] )
 *   a_layout / b_layout: VB_KCONTIG  -> operand stored [rows][K]  (ld = row pitch)
 *                        VB_KSTRIDED -> operand stored [K][rows]  (ld = k pitch)
 *   epi: + bias[n] (fp32, may be NULL) -> act -> + addend[m,n] (T, may be NULL) -> (+= C if accumulate)
 *   act: VB_ACT_GELU also writes the pre-activation (T) to aux_out when non-NULL;
 *        VB_ACT_GELU_GRAD multiplies by gelu'(aux_in[m,n]) (T);  VB_ACT_TANH applies tanh.
 *        VB_ACT_GELU_SAVE_GRAD applies GELU and writes gelu'(pre-activation) (T) to aux_out (required);
 *        VB_ACT_MUL_AUX multiplies by aux_in[m,n] (T).  The pair is how an encoder layer runs: the forward
 *        FFN-in GEMM has erf and exp in registers anyway, so it saves the derivative and the backward dgrad
 *        epilogue is one multiply instead of 128 erf+exp per lane per tile (242 -> ~150 us at B=128).
 *   colsum_out (fp32 [N], may be NULL): += column sums of the values written to C (the bias gradient of the
 *   Linear whose output gradient this GEMM produces) -- saves a separate pass over C.
 *   out_dtype: dtype (T) or VB_F32.  alpha_dev (may be NULL) is an optional fp32 DEVICE scalar that
 *   multiplies alpha (used to carry an upstream loss-gradient scalar without a host sync).
 *   Requirements: lda, ldb multiples of 8; A, B 16-byte aligned; a K-contiguous operand must be
 *   readable up to round_up(K, 8) per row and a K-strided one up to round_up(rows, 8) per k.
 * Replaces: nn.Linear forward/backward at modeling.py:232-234, 271, 303, 316, 383-385, 398, 419, 451,
 *           1220 and the GELU of :56-61 fused behind :303 / :398.
 * ---------------------------------------------------------------------------------------------- */
/*   dtype VB_BF16X3: A [M, lda] and B [N, ldb] are split operands (see the enum), both K-contiguous, K % 64 == 0,
 *   K <= lda / 2, K <= ldb / 2, lda and ldb multiples of 16; bias / addend / aux are fp32.  out_dtype VB_F32: C fp32;
 *   out_dtype VB_BF16X3: C is written as a split operand itself (bf16 [M, ldc], N % 8 == 0, N <= ldc / 2, ldc % 16 == 0, no
 *   accumulate) -- a GEMM that feeds only GEMMs (FFN-in -> FFN-out) skips the fp32 round trip and the split pass. */
int vb_gemm(int dtype, int out_dtype, int a_layout, int b_layout,
            const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
            int M, int N, int K, float alpha, const float* alpha_dev, const float* bias,
            const void* addend, int64_t ld_addend, int act,
            const void* aux_in, void* aux_out, int64_t ld_aux, int accumulate,
            float* colsum_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BertLayerNorm (+ fused residual add and dropouts), forward and backward.
 *   z = dropout_in(x) + resid ;  y = dropout_out( gamma * (z - mean) / sqrt(var + eps) + beta )
 * x, resid (may be NULL), z_out (may be NULL), y: T [M,H].  mean, rstd: fp32 [M] (may be NULL).
 * Dropout masks are regenerated from (seed, stream id, element index) -- never stored.
 * Backward: dz (T, gradient w.r.t. z == w.r.t. resid), dx (T, = dz o mask_in/(1-p_in); may alias dz
 * or be NULL when p_in == 0), dgamma/dbeta/dbias (fp32 [H], ACCUMULATED; any may be NULL; dbias is
 * the column sum of dx = bias gradient of the Linear that produced x).
 * Replaces: modeling.py:171-175 (BertLayerNorm), :272-273, :317-318 (dropout + residual + LN),
 *           :1255-1256 (embedding LN + dropout), :400 (MLM transform LN).
 * ---------------------------------------------------------------------------------------------- */
int vb_ln_fwd(int dtype, const void* x, const void* resid, void* z_out, void* y, float* mean, float* rstd,
              const float* gamma, const float* beta, int M, int H, float eps,
              float p_in, uint32_t stream_in, float p_out, uint32_t stream_out, uint64_t seed, void* stream);
int vb_ln_bwd(int dtype, const void* dy, const void* z, const float* mean, const float* rstd,
              const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias,
              int M, int H, float p_in, uint32_t stream_in, float p_out, uint32_t stream_out,
              uint64_t seed, float* ws, void* stream);
/* ws (optional, vb_ln_bwd_ws_bytes(M,H) bytes): per-block partial column sums -> two-stage reduction with no
 * global atomics; NULL falls back to fp32 atomics on dgamma/dbeta/dbias. */
int64_t vb_ln_bwd_ws_bytes(int M, int H);
/* The THREE-tensor form vb_bert_layer_fwd / _bwd run (H <= 768, no output dropout): the forward tests gamma / beta in the kernel --
 * |beta| <= 2 |gamma| on every channel -- and, if so, does NOT write z_out (three [M,H] streams per launch instead of four) and
 * sets *rebuild = 1; the backward, handed the forward's output y, beta and the same device int, then takes
 * x-hat = (y - beta) / gamma instead of (z - mean) rstd.  *rebuild = 0: z_out was written and the backward reads it (both must
 * always be passed).  gamma / beta must be unchanged between the two calls.  Same arithmetic contract as vb_ln_fwd / vb_ln_bwd
 * otherwise (modeling.py:171-175, :272-273, :317-318). */
int vb_ln_fwd_rb(int dtype, const void* x, const void* resid, void* z_out, void* y, float* mean, float* rstd,
                 const float* gamma, const float* beta, int M, int H, float eps,
                 float p_in, uint32_t stream_in, uint64_t seed, int* rebuild, void* stream);
int vb_ln_bwd_rb(int dtype, const void* dy, const void* z, const float* mean, const float* rstd,
                 const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, float* dbias,
                 int M, int H, float p_in, uint32_t stream_in, uint64_t seed, float* ws,
                 const void* y, const float* beta, const int* rebuild, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BertEmbeddingsWithVisualEmbedding gather-add.
 *   z[b, s<T]  = word[ids[b,s]] + pos[s] + type[type_ids[b,s]]
 *   z[b, T+r]  = vis_proj[b,r] + pos_vis[0] + type_vis[visual_type[b,r]] [+ pos_align[b,r]]   (visual position id is 0)
 * Tables are the fp32 master parameters; vis_proj is the projection GEMM's output (T [B*R,H]); pos_align
 * (fp32 [B*R,H], NULL when image_text_alignment is None) is vb_align_pos_fwd's output.
 * Backward scatters dz into the fp32 table gradients (ACCUMULATED) and copies the visual rows to
 * d_vis_proj (T [B*R,H]).   Replaces: modeling.py:1213-1253 and its autograd.
 * ---------------------------------------------------------------------------------------------- */
int vb_embed_fwd(int dtype, const int64_t* input_ids, const int64_t* token_type_ids, const int64_t* visual_type,
                 const void* vis_proj, const float* word, const float* pos, const float* type,
                 const float* pos_vis, const float* type_vis, const float* pos_align, void* z,
                 int B, int T, int R, int H, int V, int type_vocab, int max_pos, void* stream);
int vb_embed_bwd(int dtype, const void* dz, const int64_t* input_ids, const int64_t* token_type_ids,
                 const int64_t* visual_type, float* d_word, float* d_pos, float* d_type,
                 float* d_pos_vis, float* d_type_vis, void* d_vis_proj,
                 int B, int T, int R, int H, int V, int type_vocab, int max_pos, void* stream);

/* output_attention_weights (modeling.py:241-261, :259-260): the softmax(QK^T/sqrt(d) + mask) probabilities the training
 * kernels never materialise, written as fp32 [B, nh, S, S] from the packed qkv (T [B*S, 3H]) -- forward only, any S. */
int vb_attn_probs(int dtype, const void* qkv, const float* mask_add, float* probs, int B, int S, int nh, int head_dim,
                  void* stream);

/* image_text_alignment branch (modeling.py:1223-1245): alignment int64 [B, Ra, A] holds, per region, the text
 * positions of the words it is aligned with, -1 padded; only the first R <= Ra regions of a sample are used
 * (:1241-1243).  fwd: out[b*R+r] = mean of pos[alignment[b,r,a]] over the entries != -1 (zeros when there is none).
 * bwd: d_pos[alignment[b,r,a]] += dz[b, T+r] / count, dz being the T [B*(T+R), H] gradient of the summed embedding. */
int vb_align_pos_fwd(const int64_t* alignment, const float* pos, float* out, int B, int R, int Ra, int A,
                     int H, int max_pos, void* stream);
int vb_align_pos_bwd(int dtype, const void* dz, const int64_t* alignment, float* d_pos, int B, int T, int R,
                     int Ra, int A, int H, int max_pos, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-head self-attention (head size 64).  qkv: T [B*S, 3H] (Q | K | V column blocks),
 * mask_add: fp32 [B,S] additive key mask ((1 - mask) * -10000, modeling.py:1293-1294),
 * ctx: T [B*S, H], lse: fp32 [B,nh,S] row log-sum-exp (saved for backward),
 * keepbits: uint64 [B*nh * vb_attn_keepbits_words(S)] dropout keep-bits (only touched when p_drop > 0).
 * Backward writes dqkv (T [B*S,3H]) completely; dsum_ws is an fp32 scratch of vb_attn_bwd_ws_floats(B, S, nh) elements.
 * Sequence lengths up to 512 keys (max_position_embeddings, modeling.py:83): a (sample, head)'s K / V stay in LDS while they fit
 * (forward: 256 keys fp32 / 512 bf16; dQ pass: 192 fp32 / 416 bf16), longer sequences stream the keys in chunks of 128
 * (online softmax forward; two-sweep dQ pass) -- same results, same keep-bit layout.
 * ctx_fwd (optional): the forward output ctx -- with it, bf16 and S <= 192 the backward runs as ONE kernel (D = rowsum(P o dP)
 * taken as dO . ctx, scores and probabilities computed once); without it, or for longer sequences / fp32, as two passes
 * (dQ, then dK/dV).  dqkv_bias (optional, fp32 [3H]): += column sums of dqkv over the B*S tokens, i.e. the gradient of
 * the packed q | k | v bias (modeling.py:232-234) -- from the one-pass kernel's fp32 accumulators through per-sample
 * partial sums (no pass over dqkv), else by one column-sum pass.
 * dtype VB_BF16X3 (these four entry points and the two cross-attention ones): every tensor is fp32 exactly as for VB_F32 -- same
 * shapes, pitches, sequence limits and kernel forms -- but each MFMA operand is split once (while staging, or from registers for
 * the probabilities) into hi = bf16(x), lo = bf16(x - hi) and each product is the three bf16 MFMAs hi*hi + lo*hi + hi*lo:
 * results within ~1e-5 of the fp32 kernels (tests/test_kernels.py ATTN_X3_*): 3 bf16 MFMAs of K = 32 where the fp32 form issues
 * 8 fp32-input MFMAs of K = 4.
 * Replaces: BertSelfAttention.forward modeling.py:236-256 and its autograd.
 * ---------------------------------------------------------------------------------------------- */
int64_t vb_attn_keepbits_words(int S);
int vb_attn_fwd(int dtype, const void* qkv, const float* mask_add, void* ctx, float* lse, uint64_t* keepbits,
                int B, int S, int nh, int head_dim, float p_drop, uint64_t seed, uint32_t stream_id, void* stream);
int64_t vb_attn_bwd_ws_floats(int B, int S, int nh);
int vb_attn_bwd(int dtype, const void* qkv, const float* mask_add, const void* dctx, const float* lse,
                const uint64_t* keepbits, float* dsum_ws, void* dqkv, const void* ctx_fwd, float* dqkv_bias,
                int B, int S, int nh, int head_dim, float p_drop, uint64_t seed, uint32_t stream_id, void* stream);

/* Cross-attention with the same kernels (forward; two-pass backward): queries from one tensor, keys / values from another,
 * with their own sequence lengths and row pitches (elements; multiples of 8).  q: T [B*Sq, >= nh*64 columns from the pointer],
 * k, v: T [B*Sk, ...]; mask_add: fp32 [B, Sk] over the KEYS; ctx: T [B*Sq, ldctx]; lse / dsum_ws: fp32 [B, nh, Sq];
 * keepbits: uint64 [B*nh * vb_attn_cross_keepbits_words(Sq, Sk)].  Backward writes dq [B*Sq], dk, dv [B*Sk] completely.
 * Replaces: BertAttention.forward with context != hidden_states in the sibling LXRT model
 * (unsupervised_visualbert/src/lxrt/modeling.py:347-411, used by BertCrossattLayer :427-436 / LXRTXLayer :660-712). */
int64_t vb_attn_cross_keepbits_words(int Sq, int Sk);
int vb_attn_cross_fwd(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                      const float* mask_add, void* ctx, int64_t ldctx, float* lse, uint64_t* keepbits,
                      int B, int Sq, int Sk, int nh, int head_dim, float p_drop, uint64_t seed, uint32_t stream_id, void* stream);
int vb_attn_cross_bwd(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                      const float* mask_add, const void* dctx, int64_t lddctx, const float* lse, const uint64_t* keepbits,
                      float* dsum_ws, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                      int B, int Sq, int Sk, int nh, int head_dim, float p_drop, uint64_t seed, uint32_t stream_id, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Losses.  logits are fp32 with leading dimension ld_logits (pad columns are ignored).
 * vb_ce_fwd_bwd: CrossEntropyLoss(ignore_index) mean over counted rows -> loss[0]; acc2 is an fp32[66]
 *   scratch {sum, count, 64 partial sums}; dlogits (T, may be NULL) receives d loss / d logits for an upstream gradient
 *   of 1, INCLUDING zeroed pad columns up to ld_dlogits and all-zero rows for ignored labels.
 *   Replaces modeling.py:1471-1477 (masked LM, image-text match), :1563-1565 (NLVR2) and autograd.
 * vb_kldiv_fwd_bwd: KLDivLoss(batchmean)(log_softmax(logits), target) -> loss[0]; score[0] (may be NULL)
 *   = mean VQA score of compute_score_with_logits; dlogits fp32 (may be NULL).
 *   Replaces modeling.py:1517-1523 and :1697-1711.
 * ---------------------------------------------------------------------------------------------- */
int vb_ce_fwd_bwd(int dtype, const float* logits, int64_t ld_logits, const int64_t* labels, int ignore_index,
                  float* acc2, float* loss, void* dlogits, int64_t ld_dlogits, int M, int V, void* stream);
/* Same loss; the gradient is written COMPACTLY: row r of dlogits_compact (T [n_rows_padded, ld_dlogits]) is the
 * gradient of source row rows[r] (the rows whose label is counted, in any order), rows r >= n_rows are zero padding.
 * Rows with ignored labels have an all-zero gradient, so every backward GEMM of the MLM decoder (dgrad, wgrad, bias
 * gradient) can run over the ~12 % masked rows only -- exactly (pytorch_pretrained_bert/modeling.py:1471-1474). */
int vb_ce_fwd_bwd_rows(int dtype, const float* logits, int64_t ld_logits, const int64_t* labels, int ignore_index,
                       const int64_t* rows, int n_rows, int n_rows_padded, float* acc2, float* loss,
                       void* dlogits_compact, int64_t ld_dlogits, int M, int V, void* stream);
int vb_kldiv_fwd_bwd(const float* logits, int64_t ld_logits, const float* target, int64_t ld_target,
                     float* loss, float* score, float* dlogits, int64_t ld_dlogits, int M, int V, void* stream);

/* Heads with a tiny output width (seq_relationship 768->2, modeling.py:451; NLVR2 768->2, :1558).
 * x: T [M,K]; W: fp32 master [N,K]; y / dy: fp32 [M,N].  Backward: dx (T, overwritten, may be NULL),
 * dW / db fp32 ACCUMULATED (may be NULL); scale_dev: optional fp32 device scalar (upstream gradient). */
int vb_small_linear_fwd(int dtype, const void* x, int64_t ldx, const float* W, const float* bias, float* y,
                        int M, int N, int K, void* stream);
int vb_small_linear_bwd(int dtype, const float* dy, const void* x, int64_t ldx, const float* W,
                        void* dx, int64_t lddx, float* dW, float* db, const float* scale_dev,
                        int M, int N, int K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-tensor BertAdam over a flat fp32 arena (params / grads / exp_avg / exp_avg_sq share one
 * layout).  tensor_table: int64[n_tensors][4] = {arena offset, numel, bf16-shadow offset or -1,
 * flags (bit0 optimise, bit1 weight decay)}; chunk_table: int64[n_chunks][4] = {tensor id, arena
 * offset, length, number of chunks of this tensor}, the chunks of one tensor adjacent.  norm2_ws: fp32[n_tensors + n_chunks] scratch (per-tensor squared
 * norms, summed in table order -- bit-reproducible, so data-parallel replicas stay identical); step_counters:
 * int32[n_tensors] (state).
 * touched: optional fp32[n_tensors] on the device; tensor t takes NO step (no moments, no weight decay, no counter
 * increment) when touched[t] == 0, its gradient norm is 0 and it has never taken a step (step_counters[t] == 0) -- what
 * `p.grad is None` means to the reference's loop (optimization.py:254-255; after a parameter's first backward its .grad is
 * a tensor for good, zeroed by zero_grad(), and the reference steps it every time).  The decision is taken on the device, every step, from data that is identical on every
 * data-parallel rank once the flags have been all-reduced like the gradients (parallel.py).
 * schedule: 0 none, 1 warmup_linear(warmup, t_total).  bf16_shadow may be NULL.
 * Replaces BertAdam.step, optimization.py:239-304 (+ WarmupLinearSchedule :164-173).
 * ---------------------------------------------------------------------------------------------- */
int vb_bert_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                      void* bf16_shadow, const int64_t* chunk_table, int n_chunks,
                      const int64_t* tensor_table, int n_tensors, const float* touched, float* norm2_ws, int* step_counters,
                      float lr, float b1, float b2, float eps, float weight_decay,
                      float max_grad_norm, float warmup, float t_total, int schedule, void* stream);
int vb_refresh_bf16_shadow(const float* params, void* bf16_shadow, const int64_t* chunk_table,
                           int n_chunks, const int64_t* tensor_table, void* stream);
/* W^T copies of the bf16 GEMM weights (so dgrad dx = dy W is a K-contiguous x K-contiguous GEMM):
 * tensor_table5: int64[n][5] = {src offset in bf16_shadow, R, C, dst offset in bf16_shadow_t, dst ld};
 * tile_table3: int64[n_tiles][3] = {tensor, 64-row tile, 64-col tile}. Pad columns of the destination are
 * never written (allocate it zeroed). */
int vb_refresh_transposed_shadow(const void* bf16_shadow, void* bf16_shadow_t, const int64_t* tensor_table5,
                                 const int64_t* tile_table3, int n_tiles, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input preparation (integer work, bit-exact): image_mask[b,r] = r < image_dim[b] (models/model.py:262-268,
 * or image_mask given), attention_mask = cat(input_mask, image_mask) (modeling.py:1417), additive mask
 * (1 - m) * -10000 (:1293-1294), LM labels extended with -1 over visual slots (:1419-1426).
 * ---------------------------------------------------------------------------------------------- */
int vb_prepare_inputs(const int64_t* input_mask, const int64_t* image_dim, const int64_t* image_mask,
                      const int64_t* masked_lm_labels, int64_t* attention_mask, float* mask_add,
                      int64_t* labels_ext, int B, int T, int R, void* stream);
/* zero `bytes` bytes (multiple of 16, 16-byte aligned): the flat gradient arena, once per step (replaces the
 * zero_grad loop over parameters of pytorch_pretrained_bert/optimization.py's callers) */
int vb_zero(void* dst, int64_t bytes, void* stream);
/* elementwise dtype conversion (VB_F32 / VB_BF16 in any combination) */
int vb_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream);
/* Split operands of the VB_BF16X3 GEMM mode.  src: fp32 [rows, cols] (ld_src); dst: bf16 [rows, ld_dst], ld_dst a multiple
 * of 16 with cols <= ld_dst / 2:  dst[r, c] = hi = bf16(src[r, c]),  dst[r, ld_dst/2 + c] = bf16(src[r, c] - hi); the columns
 * cols .. ld_dst/2 - 1 of both planes are written as zeros (so a GEMM may reduce over whole K tiles of a padded operand).
 * vb_split_bf16_t writes the TRANSPOSE: dst[c, r] and dst[c, ld_dst/2 + r] for src[r, c] (dst: bf16 [cols, ld_dst],
 * rows <= ld_dst / 2, pad columns zeroed) -- the W^T operands of the dgrad GEMMs. */
int vb_split_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols, void* stream);
int vb_split_bf16_t(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols, void* stream);
/* nn.Dropout on a contiguous T[n] (modeling.py:1495, 1509, 1557: the [B, H] state in front of the multichoice / VQA / NLVR2
 * heads): y = keep ? x / (1 - p) : 0, keep-bits from (seed, stream_id, element index).  Its own backward: call it on dy with
 * the same seed and stream_id.  x == y (in place) is allowed. */
int vb_dropout(int dtype, const void* x, void* y, int64_t n, float p, uint64_t seed, uint32_t stream_id, void* stream);
/* VQA head gather (modeling.py:1503-1505): out[b] = x[b, input_mask[b].sum() - 2]; index_out int64[B] */
int vb_gather_rows(int dtype, const void* x, const int64_t* input_mask, void* out, int64_t* index_out,
                   int B, int S, int T, int H, void* stream);
int vb_scatter_rows(int dtype, const void* dout, const int64_t* index, void* dx, int B, int S, int H, void* stream);

/* Flickr30k grounding head (modeling.py:1568-1598).
 * vb_gather_index_rows: batched_index_select (:1713-1716), out[b*E+e] = x[b, index[b,e]] with the -1 padding read as
 *   position 0 (:1574); vb_scatter_index_rows is its adjoint, dx[b,s] = addend[b,s] + sum_{e: index[b,e]==s} dsel[b*E+e]
 *   (addend may be NULL; dx is written completely).
 * vb_flickr_scores_fwd: FlickrAttention.forward (:1624-1648) after its two projections -- q: T [B*E, ldq] (entity
 *   queries), k: T [B*S, ldk] (keys of EVERY position; the regions are rows T..S-1 of a sample):
 *   scores[b,e,r] = q[b,e].k[b,T+r] / sqrt(d) + (1 - image_mask[b,r]) * -10000 (fp32 [B*E, R]); stats fp32[3] =
 *   {entities whose best-scoring region has a non-zero label, sum of the labels, entities with position != -1},
 *   i.e. compute_score_with_logits_flickr (:1650-1673, recall 1) and entities_num (:1570).
 * vb_flickr_scores_bwd: dq (T [B*E, ldq]) and dk (T [B*S, ldk], text rows zeroed) from dscores (fp32 [B*E, R]),
 *   scaled by alpha * scale_dev[0] (scale_dev may be NULL). */
int vb_gather_index_rows(int dtype, const void* x, const int64_t* index, void* out, int B, int S, int E, int H,
                         void* stream);
int vb_scatter_index_rows(int dtype, const void* dsel, const int64_t* index, const void* addend, void* dx,
                          int B, int S, int E, int H, void* stream);
int vb_flickr_scores_fwd(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const int64_t* image_mask,
                         const float* label, const int64_t* position, float* scores, float* stats,
                         int B, int E, int R, int S, int T, int d, void* stream);
int vb_flickr_scores_bwd(int dtype, const float* dscores, const void* q, int64_t ldq, const void* k, int64_t ldk,
                         void* dq, void* dk, const float* scale_dev, float alpha,
                         int B, int E, int R, int S, int T, int d, void* stream);

/* bias gradients: out[n] += scale * sum_m x[m,n]  (x: T [M,N], out fp32, scale_dev optional device scalar) */
int vb_colsum(int dtype, const void* x, int64_t ld, float* out, const float* scale_dev, int M, int N, void* stream);
/* dx = dy * act'(aux): act = VB_ACT_GELU (aux = pre-activation, modeling.py:56-61) or VB_ACT_TANH
 * (aux = tanh output, BertPooler modeling.py:385).  Contiguous T[n]. */
int vb_act_bwd(int dtype, const void* dy, const void* aux, void* dx, int64_t n, int act, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One whole BertLayer per call (modeling.py:331-341): host-side sequencing of the entry points above
 * (7 launches forward, 15 backward) so the caller pays one FFI call per layer and direction.
 *   weights[VB_LW_COUNT]: matrices in T ([3H,H] packed q|k|v, [H,H], [I,H], [H,I]); biases and LayerNorm
 *                         parameters fp32.   grads[VB_LW_COUNT]: fp32 accumulation targets, same order.
 *   saved  : vb_bert_layer_saved_bytes() bytes written by forward, read by backward (qkv, ctx, lse,
 *            keep-bits, pre-LN sums + statistics, attention output, FFN pre-activation and activation).
 *            The layout is the library's business (opaque to the caller).  VB_BF16X3: tensors that only GEMMs
 *            read -- the context and the FFN activation here; dfo, dao, dqkv, d(pre-activation) in the scratch
 *            during backward -- exist ONLY as their bf16 hi | lo images, written by the kernels that produce them
 *   scratch: vb_bert_layer_scratch_bytes() bytes of temporaries, reusable by every layer on one stream
 *   h_in/h_out/d_out/d_in: T [B*S, H];  mask_add: fp32 [B,S].  Dropout sites use stream ids sid..sid+4.
 *   The backward is handed the forward's h_out again (required when H <= 768): a LayerNorm forward does not write its pre-LN sum
 *   when the backward can rebuild x-hat = (y - beta) / gamma from the output it reads anyway (three tensors per launch instead of
 *   four).  The kernels decide that themselves from the parameters (|beta| <= 2 |gamma| on every channel -- the rounding of y then
 *   costs x-hat no more than the rounding of a saved sum would) and record it in `saved`; gamma / beta must not change between a
 *   layer's forward and its backward (as for every weight the backward reads).
 * ---------------------------------------------------------------------------------------------- */
enum { VB_LW_QKV_W = 0, VB_LW_QKV_B, VB_LW_AO_W, VB_LW_AO_B, VB_LW_LN1_G, VB_LW_LN1_B,
       VB_LW_FI_W, VB_LW_FI_B, VB_LW_FO_W, VB_LW_FO_B, VB_LW_LN2_G, VB_LW_LN2_B, VB_LW_COUNT };
/* optional transposed weights for backward (weights_t[4], entries may be NULL -> K-strided read of weights[]):
 * W^T as T [in, ld >= out] for QKV, attention-out, FFN-in, FFN-out; ld_t[4] their leading dimensions */
enum { VB_LWT_QKV = 0, VB_LWT_AO, VB_LWT_FI, VB_LWT_FO, VB_LWT_COUNT };
int64_t vb_bert_layer_saved_bytes(int dtype, int B, int S, int H, int I, int nh, float p_attn);
int64_t vb_bert_layer_scratch_bytes(int dtype, int B, int S, int H, int I, int nh);
int vb_bert_layer_fwd(int dtype, const void* h_in, const float* mask_add, void* h_out,
                      void* saved, void* scratch, const void* const* weights,
                      int B, int S, int H, int I, int nh, float p_hidden, float p_attn, float eps,
                      uint64_t seed, uint32_t sid, void* stream);
int vb_bert_layer_bwd(int dtype, const void* h_in, const void* h_out, const float* mask_add, const void* d_out, void* d_in,
                      const void* saved, void* scratch, const void* const* weights, void* const* grads,
                      const void* const* weights_t, const int64_t* ld_t,
                      int B, int S, int H, int I, int nh, float p_hidden, float p_attn,
                      uint64_t seed, uint32_t sid, void* stream);

/* Optional per-stream timing of the GEMM launches (what bench.py's roofline object is measured with, over its timed region):
 * vb_stream_profile(stream, 1) makes every later vb_gemm / vb_wgrad_grouped launch ENQUEUED ON THAT STREAM record a HIP
 * event pair around itself (other streams are unaffected; nothing is recorded by default);
 * vb_stream_profile_read(stream, ...) (after synchronising the stream) returns per-launch {milliseconds, algorithmic FLOPs
 * 2MNK, key}; key bits: 8 = fp32 operands (else bf16), 4 = fp32 output, 2 = A K-strided, 1 = B K-strided, 16 = 256x256-tile
 * kernel, 64 = two-workgroup 256x128 kernel, 256 = split-operand (bf16x3) mode, 1024 = the K range of each tile was split over
 * several workgroups (vb_stream_set_scratch); 39 (= 32|4|2|1) is the one-workgroup-per-dW-tile form of vb_wgrad_grouped.
 * vb_stream_profile(stream, 0) stops and frees the events. */
int vb_stream_profile(void* stream, int enable);
int64_t vb_stream_profile_read(void* stream, double* ms, double* flops, int* key, int64_t max_records);

/* Weight gradients of a group of Linears that saw the same tokens (the four of an encoder layer), one launch:
 *   dw[i][n_out[i], n_in[i]] (fp32, ld_dw[i]) += alpha * dy[i]^T x[i],   dy[i]: [tokens, n_out[i]] (T, ld_dy[i]),
 *   x[i]: [tokens, n_in[i]] (T, ld_x[i]).  n <= 8.  alpha_dev: optional fp32 device scalar multiplying alpha.
 * bf16 with leading dimensions that are multiples of 8 (>= round_up(features, 8)) takes one of two routes:
 *   - up to 128 token tiles of 64 (tokens <= 8192; fewer than 64 tokens included), every n_in[i] % 8 == 0, ld_dw[i] % 4 == 0,
 *     16-byte-aligned dw[i] and no two dw ranges overlapping: ONE launch with one workgroup per dW tile (128x128, or 256x128
 *     when that many tiles still fit the compute units), each walking all the tokens (ragged last tile zero-filled in
 *     registers) and finishing with a plain 16-byte read-modify-write of its dW tile - no atomics, no second launch;
 *   - otherwise the whole 64-token tiles as ONE persistent kernel (operands copied as stored, fragments gathered by
 *     transposing LDS reads, token slices added with fp32 atomics) and the ragged tokens % 64 rows of all problems as one
 *     more launch (one after the other if two problems' dw ranges overlap).
 * Anything else falls back to n vb_gemm calls.
 * Replaces: the dW = dy^T x half of autograd for nn.Linear at pytorch_pretrained_bert/modeling.py:232-234 (Q,K,V),
 * :271, :303, :316. */
int vb_wgrad_grouped(int dtype, int n, const void* const* dy, const int64_t* ld_dy, const void* const* x,
                     const int64_t* ld_x, void* const* dw, const int64_t* ld_dw, const int* n_out, const int* n_in,
                     int tokens, float alpha, const float* alpha_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient all-reduce over RCCL (xGMI), one communicator per process / GPU.
 * Replaces nn.DataParallel's per-step broadcast + reduce-add (models/model_wrapper.py:75, 146; train.py:146): every rank
 * owns a replica, and the flat fp32 gradient arena is averaged bucket by bucket while backward is still running.
 *   vb_comm_unique_id: rank 0 fills VB_COMM_ID_BYTES bytes (ncclGetUniqueId); the caller ships them to the other ranks by
 *                      any host channel (the Python side uses torch.distributed's store).
 *   vb_comm_init:      collective over all ranks (ncclCommInitRank on the CURRENT device); *comm receives an opaque handle.
 *   vb_allreduce_bucket: in-place sum (average != 0: mean over ranks, ncclAvg) of `count` elements of `dtype`
 *                      (VB_F32 / VB_BF16), enqueued on `stream`; no host synchronisation.
 *   vb_comm_destroy:   frees the communicator.
 * librccl is opened at the first vb_comm_* call (the copy already mapped into the process, e.g. PyTorch's, else
 * librccl.so.1 from the loader path); a single-GPU user never needs it.  Errors: VB_ERR_UNSUPPORTED when RCCL cannot be
 * loaded, VB_ERR_LAUNCH when an RCCL call fails.
 * ---------------------------------------------------------------------------------------------- */
#define VB_COMM_ID_BYTES 128
int vb_comm_unique_id(void* host_id);
int vb_comm_init(const void* host_id, int rank, int nranks, void** comm);
int vb_comm_nranks(void* comm);
int vb_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, int average, void* stream);
int vb_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
