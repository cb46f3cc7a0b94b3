/*
 * kemr.h -- C ABI of libkemr.so, the MI355X (gfx950) CLIP retrieval hot path.
 *
 * The reference (REEVALUATE/knowledge_enhanced_multimodal_retrieval) is 100 % Python and has no
 * native plugin/FFI boundary of its own: its boundary for this path is a set of Python call
 * signatures.  This header is the C ABI one level below those signatures;
 * every entry point names the reference call it replaces.  The Python host side
 * (knowledge_enhanced_multimodal_retrieval_amd/, and the drop-in `src.clip.*` mirror) binds these
 * symbols with ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain C: opaque handle, raw pointers, sizes; no torch / C++ types cross the ABI.
 *  - every function returns KEMR_OK (0) or a negative kemr_status; kemr_last_error() returns a
 *    thread-local message for the last failure.  Nothing throws across the ABI.
 *  - "dev" pointers are device (HBM) pointers owned by the caller; the library never allocates on
 *    the hot path.  Scratch comes from the caller through a workspace sized by *_workspace_bytes().
 *    The library owns only the model handle and its packed weights (allocated in finalize).
 *  - all launches are asynchronous on the hipStream_t passed as `void* stream` (0 = default stream);
 *    no entry point synchronises the device except kemr_model_finalize().
 *  - no HIP call happens at library load time (fork-safe for DataLoader workers,
 *    reference: src/clip/eval/evaluator_baseline.py:83-92).
 */
#ifndef KEMR_H_
#define KEMR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: kemr_sim_workspace_bytes takes kdim, panels are allocated with ceil256(rows) rows (kemr_panel_build zero-fills up to
 *    256), kemr_model_set_option / kemr_model_get_option, the tools' switches live in kemr_debug.h (kemr_debug_set).
 * 3: kemr_encode_text_packed / kemr_text_packed_workspace_bytes, option "last_block_pooled_row"; kemr_workspace_bytes grows by the
 *    last block's pooled-row area (callers that size their workspace with it need no change).
 * 4: the same entry points with changed defaults and semantics: option "residual_stream_24bit" defaults to 1; the fp8 precisions apply
 *    to the vision tower only and scale the e4m3 A operand per channel; kemr_preprocess_u8_batch accepts 0 x 0 items (an undecodable
 *    image: zeros after normalisation); kemr_debug_set refuses the experiment kernels the library was built without. */
#define KEMR_ABI_VERSION 4

typedef enum kemr_status {
    KEMR_OK = 0,
    KEMR_ERR_INVALID = -1,      /* bad argument / shape the kernels do not support */
    KEMR_ERR_STATE = -2,        /* call order (e.g. encode before finalize, missing tensor) */
    KEMR_ERR_HIP = -3,          /* a HIP runtime call failed; message carries hipGetErrorString */
    KEMR_ERR_WORKSPACE = -4,    /* workspace too small / misaligned */
    KEMR_ERR_NOMEM = -5
} kemr_status;

typedef enum kemr_dtype { KEMR_F32 = 0, KEMR_BF16 = 1, KEMR_I32 = 2, KEMR_FP8 = 3 /* OCP e4m3fn */ } kemr_dtype;

/* compute precision of the encoder GEMMs (activations + weights); accumulation is always fp32 */
/* KEMR_PREC_BF16:       the default.  bf16 GEMM / attention operands, fp32 accumulation, fp32 residual stream (closest to the
 *                       reference's .float() model); the out-proj / fc2 updates are stored as bf16 and added in fp32 by the
 *                       LayerNorms (option "residual_fusion" = 2 adds them in the GEMM epilogues without rounding them).
 * KEMR_PREC_BF16_RES16: as above with the residual stream stored as bf16 between layers (what fp16/bf16 CLIP inference
 *                       does everywhere); LayerNorm statistics and the residual add stay fp32.  8 instead of 12 bytes
 *                       of HBM traffic per residual element and LayerNorm, and 16 instead of 18 workspace bytes.
 *                       For calls of more than 512 token rows the residual add runs inside the out-proj / fc2 GEMM epilogues
 *                       unless the environment holds KEMR_RESADD=0 (x is then rounded once per layer instead of twice, -3.7 %).
 * KEMR_PREC_FP8:        BASELINE config 5.  The QKV GEMMs of the VISION tower (22 % of a gallery item's FLOPs) run on fp8 e4m3
 *                       operands with the block-scaled MFMA (K = 128 per instruction, twice the bf16 rate): ln_1 writes its
 *                       output as e4m3 -- its gain moved out of the values by per-channel power-of-two scales that finalize folds
 *                       into gamma / beta and into the weight columns, so that a trained tower's gains of 30-100 do not saturate
 *                       at +-448 (round 4) --, the weights are quantised per output channel at finalize, the scale is applied
 *                       to the fp32 accumulators.  The text tower (one pooled row behind a causal softmax: e4m3 q / k / v cost
 *                       it 1 - cos 1.4e-3 .. 5e-3, over the path's 1e-3 bar) and everything else stay as KEMR_PREC_BF16.
 *                       Recall@10 stays within 0.2 points of bf16 (tests/test_encoder_gpu.py).  Vision width: multiple of 128.
 * KEMR_PREC_FP8_MLP:    the fc1 GEMMs as well (56 % of the FLOPs in fp8, +25 % encode throughput); on the synthetic
 *                       near-duplicate retrieval test this costs about one point of Recall@10 where bf16 is below 90 %
 *                       (the MLP update goes straight into the residual stream, the QKV error is averaged by the softmax),
 *                       so it is outside config 5's bar and opt-in.
 * KEMR_PREC_FP8_RES16:  KEMR_PREC_FP8 with the bf16 residual stream of KEMR_PREC_BF16_RES16 (the two savings add). */
typedef enum kemr_precision { KEMR_PREC_BF16 = 1, KEMR_PREC_BF16_RES16 = 2, KEMR_PREC_FP8 = 3, KEMR_PREC_FP8_MLP = 4, KEMR_PREC_FP8_RES16 = 5 } kemr_precision;

typedef enum kemr_tower { KEMR_TOWER_VISION = 0, KEMR_TOWER_TEXT = 1 } kemr_tower;

/* Architecture numbers of an OpenAI-CLIP style model (heads are width/64, MLP is 4*width).
 * ViT-L/14: {768,224,14,1024,24,768,12,49408,77}; ViT-B/32: {512,224,32,768,12,512,12,49408,77}. */
typedef struct kemr_cfg {
    int32_t embed_dim;    /* joint embedding size D */
    int32_t image_size;   /* input resolution (square) */
    int32_t patch;        /* patch size, image_size % patch == 0 */
    int32_t v_width;      /* vision width, multiple of 256 */
    int32_t v_layers;
    int32_t t_width;      /* text width, multiple of 256 */
    int32_t t_layers;
    int32_t vocab;
    int32_t ctx;          /* text context length (77) */
} kemr_cfg;

typedef struct kemr_model kemr_model;

const char* kemr_last_error(void);
int kemr_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Model lifecycle.  Replaces `clip.load(name, device)` + `.float()` + `load_state_dict(sd,
 * strict=True)` (reference: src/clip/model/clip_model.py:41-64).
 *   create -> load_tensor (once per state-dict entry, OpenAI-CLIP names, fp32 host memory)
 *          -> finalize (checks every required tensor is present = "strict", packs bf16 weights
 *             into device memory: q-scale folded into W_q, conv1 flattened/padded to a GEMM panel)
 *   load_tensor may be called again after finalize (fine-tuned checkpoint): call finalize again.
 * ------------------------------------------------------------------------------------------- */
int kemr_model_create(const kemr_cfg* cfg, kemr_model** out);
int kemr_model_load_tensor(kemr_model* m, const char* name, const void* host_ptr, int dtype /*kemr_dtype, F32 only*/,
                           const int64_t* shape, int rank);
int kemr_model_finalize(kemr_model* m, int precision /*kemr_precision*/);
int kemr_model_destroy(kemr_model* m);
/* Per-model options (any time after create; take effect on the next encode call of this model, no other model is touched):
 *   "residual_fusion"  calls of more than 512 token rows add the residual inside the out-proj / fc2 GEMM epilogues instead of
 *                      storing bf16 updates that the LayerNorms apply: 0 never; 1 (default; KEMR_RESADD in the environment at
 *                      create overrides) for bf16 residual streams (x is then rounded twice per layer instead of once); 2 for
 *                      fp32 residual streams as well (x += A.W^T + b in fp32, no update is rounded at all; slower).
 *   "last_block_pooled_row"  1 (default; KEMR_LAST_BLOCK_FULL=1 in the environment at create makes it 0): one row per item leaves
 *                      a tower -- the class token's (`ln_post(x[:, 0, :]) @ proj`) or the end-of-text token's -- so the LAST
 *                      block computes K and V for every row but the query, attention output, out-proj, ln_2 and MLP for that
 *                      row alone (2 instead of 12 W^2 of GEMM work per token row in that block).  The pooled rows see the same
 *                      arithmetic in smaller launches.  Applies with store-only epilogues and fc1 on bf16 (not KEMR_PREC_FP8_MLP); 0 = every row
 *                      through every block, as the reference computes it.
 *   "residual_stream_24bit"  (set BEFORE kemr_model_finalize; default 1 since round 4, KEMR_STREAM24=0 in the environment at create
 *                      makes it 0): the fp32 residual stream of KEMR_PREC_BF16 / FP8 / FP8_MLP is stored as 24-bit floats -- the
 *                      upper three bytes of the fp32, rounded to nearest: 15 mantissa bits, 128 x finer than bf16 -- 3 instead of
 *                      4 bytes per element in the HBM-bound LayerNorm passes (+2 % on the ViT-L/14 step); every statistic and add
 *                      stays fp32 arithmetic.  The default because it meets every bar the 4-byte stream meets (1 - cos against the
 *                      fp32 oracle on plain and heavy-tailed weights, the fixed 0.2-point Recall@10 bar on ViT-B/32 and ViT-L/14,
 *                      the end-to-end margin rule: DESIGN.md section 2); 0 = the 4-byte stream (python: precision "bf16"). */
int kemr_model_set_option(kemr_model* m, const char* key, int value);
int kemr_model_get_option(const kemr_model* m, const char* key, int* value);
/* number of required tensor names; name i via kemr_model_tensor_name (for strict-load diagnostics) */
int kemr_model_num_tensors(const kemr_model* m);
const char* kemr_model_tensor_name(const kemr_model* m, int i);

/* ---------------------------------------------------------------------------------------------
 * Encoders.  Replace `model.encode_image(images)` / `model.encode_text(tokens)` and the following
 * `x / x.norm(dim=-1, keepdim=True)` (reference: src/clip/eval/evaluator_baseline.py:107-108,
 * 113-114, 119-120; src/clip/eval/evaluator.py:121-122, 127-128, 133-134;
 * src/clip/model/fusion_model.py:287-303).
 *   pixels_dev : fp32 [B,3,S,S] contiguous NCHW, already mean/std normalised
 *   ids_dev    : int32 [B,ctx]; pooled at the first position of the row maximum (EOT)
 *   out_dev    : fp32 [B, embed_dim]; L2-normalised when normalize != 0
 *   workspace  : >= kemr_workspace_bytes(m, tower, B) bytes, 256-byte aligned, contents don't matter
 * ------------------------------------------------------------------------------------------- */
size_t kemr_workspace_bytes(const kemr_model* m, int tower /*kemr_tower*/, int batch);
int kemr_encode_image(kemr_model* m, const float* pixels_dev, int batch, float* out_dev, int normalize,
                      void* workspace_dev, size_t workspace_bytes, void* stream);
int kemr_encode_text(kemr_model* m, const int32_t* ids_dev, int batch, float* out_dev, int normalize,
                     void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same text embeddings from the rows that can reach them.  The text transformer's mask is causal and the pooled row is the
 * end-of-text token's (reference: `x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection` behind the masked
 * transformer in the CLIP model `clip.load` returns; call sites as above), so positions behind that token cannot influence the
 * output: text i is computed on its first lens[i] positions only, all texts packed one behind the other (`rows` = sum of lens
 * token rows instead of B * ctx in every launch).  With lens[i] >= argmax_i + 1 the result is kemr_encode_text's up to the fp32
 * summation order of the GEMM kernel a launch of that many rows is routed to -- what another batch size changes as well.
 *   lens_dev : int32 [B] on the device, each in 1 .. ctx (clamped on the device)
 *   rows     : HOST integer, the sum of lens (the tokenizer's side knows it; B <= rows <= B * ctx is checked, and the device
 *              clamps the prefix sums into it, so no argument can make a kernel index outside the workspace)
 *   workspace: >= kemr_text_packed_workspace_bytes(m, rows, B) bytes
 * A length shorter than argmax_i + 1 pools text i's last computed row (memory-safe, not the reference's embedding). */
size_t kemr_text_packed_workspace_bytes(const kemr_model* m, int rows, int batch);
int kemr_encode_text_packed(kemr_model* m, const int32_t* ids_dev, const int32_t* lens_dev, int rows, int batch, float* out_dev,
                            int normalize, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Similarity + ranking.  Replaces `S = Q @ C.T`, the weighted T2I/T2T sum, and the two full
 * `np.argsort(-S)` passes of Recall@K / MRR (reference: src/clip/eval/metrics.py:13-76, 102,
 * 145-148; src/clip/eval/fusion.py:6-20) without materialising the Q x N matrix.
 *
 * Operands are "panels": bf16 [rows, kdim] row-major with kdim % 64 == 0, built by
 * kemr_panel_build from fp32 embeddings.  A panel concatenates `nparts` embedding sets along k
 * (fused T2I+T2T scoring = one contraction over [w_i*I ; w_t*T], metrics.py:148) and, with
 * terms == 3, stores the bf16 split hi/lo so that the contraction reproduces fp32 products
 * (query panel [hi | lo | hi], gallery panel [hi | hi | lo]); terms == 1 is plain bf16.
 *   kdim = nparts * terms * ceil64(D)
 * A panel buffer must be allocated with its row count rounded up to a multiple of 256
 * (kemr_panel_build zero-fills the pad rows; the kernels read whole 128- and 256-row tiles).
 * ------------------------------------------------------------------------------------------- */
typedef enum kemr_panel_side { KEMR_SIDE_QUERY = 0, KEMR_SIDE_GALLERY = 1 } kemr_panel_side;

int64_t kemr_panel_kdim(int d, int nparts, int terms);
/* parts[p] : fp32 [rows, d] dev pointers (nparts of them, host array of pointers);
 * part_scale[p] : scalar weight applied to part p (NULL = 1); row_scale[p] : optional fp32 [rows]
 * dev pointer with a per-row factor for part p (per-query gate, fusion_model.py:136-196), or NULL. */
int kemr_panel_build(const float* const* parts_dev, const float* part_scale, const float* const* row_scale_dev,
                     int nparts, int rows, int d, int terms, int side /*kemr_panel_side*/,
                     void* panel_dev /* bf16 [rows, kdim] */, void* stream);

/* Fused score + per-query top-k + rank of a ground-truth candidate.
 *   q_panel [nq,kdim], g_panel [ng,kdim]
 *   gallery_offset : global id of gallery row 0 (sharded galleries); ids written are global
 *   k <= 32; top_scores/top_idx [nq,k] sorted (score desc, id asc), padded with -inf / -1;
 *   k == 0 = rank only (top_* may be NULL, the ground truth is then required); without a bonus list this is the fast
 *            256 x 256-tile pass on the encoder GEMM's main loop (same scores, bit for bit)
 *   gt_idx  : optional int32 [nq] GLOBAL candidate id of each query's ground truth
 *   gt_score: optional fp32 [nq]; score of (query, gt) as produced by kemr_pair_scores
 *   ahead   : optional int32 [nq]; += #{j in this gallery : s_ij > s_gt or (s_ij == s_gt and id_j < gt)}
 *             (rank = 1 + sum of `ahead` over shards; caller zeroes it)
 *   bonus_*: optional sparse additive bonuses (SPARQL hits, eval/fusion.py:22-206) in CSR over
 *            queries: bonus_rowptr int32 [nq+1], bonus_col int32 (GLOBAL ids, ascending within a row),
 *            bonus_val fp32.  Weighted fusion's alpha is folded into the query panel (part_scale).
 *   Galleries of >= 8192 rows with >= 256 queries and no bonus list take their scores from the encoder GEMM's main loop as
 *   well: thresholds from the block maxima of a strided sample of the gallery (about a tenth of the rows at k = 10), one 256 x 256-tile pass that appends the
 *   candidates above a query's threshold to its lists (the rank count rides along), one selection pass.  Same scores, same
 *   order rule, same outputs bit for bit; lists that overflow re-run the slower kernel on the device (no host round trip).
 *   workspace >= kemr_sim_workspace_bytes(nq, ng, kdim, k), 256-byte aligned */
size_t kemr_sim_workspace_bytes(int nq, int ng, int64_t kdim, int k);
int kemr_sim_topk(const void* q_panel_dev, int nq, const void* g_panel_dev, int ng, int64_t kdim,
                  int64_t gallery_offset, int k, float* top_scores_dev, int32_t* top_idx_dev,
                  const int32_t* gt_idx_dev, const float* gt_score_dev, int32_t* ahead_dev,
                  const int32_t* bonus_rowptr_dev, const int32_t* bonus_col_dev, const float* bonus_val_dev,
                  void* workspace_dev, size_t workspace_bytes, void* stream);

/* score of listed (query row, LOCAL gallery row) pairs with exactly the arithmetic of kemr_sim_topk
 * (same MFMA k-order), so that `s_ij > s_gt` comparisons are self-consistent. out fp32 [npairs]. */
int kemr_pair_scores(const void* q_panel_dev, const void* g_panel_dev, int64_t kdim,
                     const int32_t* q_rows_dev, const int32_t* g_rows_dev, int npairs,
                     float* out_dev, void* stream);

/* Merge `nlists` sorted top-k lists per query (shards / column chunks) into one.
 * in_scores/in_idx [nq, nlists, k] -> out [nq, k]; same order rule as kemr_sim_topk. */
int kemr_topk_merge(const float* in_scores_dev, const int32_t* in_idx_dev, int nq, int nlists, int k,
                    float* out_scores_dev, int32_t* out_idx_dev, void* stream);

/* Dense score tile (for the learned fusion heads that are not GEMM epilogues and for debugging):
 * out fp32 [nq, ng] row-major = q_panel . g_panel^T.  Reference: fusion_model.py:324-325. */
int kemr_scores_dense(const void* q_panel_dev, int nq, const void* g_panel_dev, int ng, int64_t kdim,
                      float* out_dev, int64_t ld_out, void* stream);

/* Rank an already materialised score matrix (fp32 [nq, ld], device): for every row the number of
 * candidates ranked ahead of column gt_idx[row] (written, not accumulated) and/or the sorted top-k.
 * Replaces the np.argsort passes when the caller hands over a matrix: compute_recall_at_k,
 * compute_mrr_and_mean_rank (metrics.py:13-76), compute_retrieval_metrics_fusion (metrics.py:165-185),
 * evaluate_retrieval (eval/fusion.py:6-20), evaluator_fusion.py:126.  Either output pair may be NULL. */
int kemr_rank_dense(const float* scores_dev, int nq, int ng, int64_t ld, const int32_t* gt_idx_dev,
                    int32_t* ahead_dev, int k, float* top_scores_dev, int32_t* top_idx_dev, void* stream);

/* Gate of the learned gated fusion heads in eval mode (reference src/clip/model/fusion_model.py: SimpleGatedFusion /
 * SimpleGatedFusionWithBias `sigmoid((q * w).sum(1) + b)`, GatedFusionHead `Linear(d,128) -> ReLU -> Linear(128,1) -> Sigmoid`):
 * out[r] = sigmoid(sum_c act(x[r,c] + pre_bias[c]) * w[c] + bias), act = ReLU when relu != 0 else identity; x fp32 [rows, cols]
 * (the queries, or their Linear(d,128) image from the dense kernels), pre_bias fp32 [cols] or NULL, w fp32 [cols]. */
int kemr_gate_rows(const float* x_dev, int rows, int cols, const float* pre_bias_dev, const float* w_dev, float bias, int relu,
                   float* out_dev, void* stream);

/* Learned "linear" fusion head in eval mode (reference src/clip/model/fusion_model.py:25-48):
 * out[i] = w1 . relu(W0 . [t2i[i], t2t[i]] + b0) + b1 over n score pairs (W0 fp32 [hidden,2], b0/w1 fp32 [hidden]). */
int kemr_linear_head(const float* t2i_dev, const float* t2t_dev, int64_t n, const float* w0_dev, const float* b0_dev,
                     const float* w1_dev, float b1, int hidden, float* out_dev, void* stream);

/* Pair stage of the learned "cross_attention" fusion head in eval mode (reference src/clip/model/fusion_model.py:51-133).
 * Inputs are per-query / per-candidate quantities the host precomputes with the dense kernels (see fusion_model.py of the
 * build): st_x fp32 [heads][n_c][n_q] scaled attention scores (x = image / target key), p_x fp32 [n_c][heads][hid1] =
 * W1.Wo[:,head].V_x, c0 = W1.bo + b1 [hid1], w2t = W2^T [hid1][hid2], b2 [hid2], w3 [hid2], b3.
 * out_t fp32 [n_c][n_q] = 0.5 * tanh(MLP(softmax-mixed context))  (transposed score matrix). */
int kemr_cross_attention_pairs(const float* st_i_dev, const float* st_t_dev, const float* p_i_dev, const float* p_t_dev,
                               const float* c0_dev, const float* w2t_dev, const float* b2_dev, const float* w3_dev, float b3,
                               int heads, int n_q, int n_c, int hid1, int hid2, float* out_t_dev, void* stream);

/* Optional per-kernel-class timing with hipEvents recorded on the launch stream (bench.py's roofline line).
 * Classes: 0 GEMM, 1 LayerNorm, 2 attention, 3 embed/tail, 4 similarity tile kernel.  Not thread-safe;
 * profile_end synchronises the device.  Off by default: no events are recorded on the normal path. */
#define KEMR_PROF_NCLASS 5
int kemr_profile_begin(int max_launches);
int kemr_profile_end(double* ms_per_class, int64_t* launches_per_class, int nclass);

/* ---------------------------------------------------------------------------------------------
 * Building blocks exposed for per-kernel parity tests (tests/ call these through the ABI).
 * ------------------------------------------------------------------------------------------- */
typedef enum kemr_epilogue {
    KEMR_EPI_BIAS_BF16 = 0,        /* C_bf16 = A.W^T + bias                                  */
    KEMR_EPI_BIAS_QGELU_BF16 = 1,  /* C_bf16 = quickgelu(A.W^T + bias)                       */
    KEMR_EPI_BIAS_RESID_F32 = 2,   /* X_f32 += A.W^T + bias   (in place on the residual; from 128 output tiles of 256 x 256 and more
                                      than 512 rows up the persistent kernel: C with ceil256(m) rows) */
    KEMR_EPI_BIAS_RESADD_BF16 = 4  /* X_bf16 = bf16(bf16(A.W^T + bias) + X_bf16), in place; persistent 256 x 256 kernel only:
                                      N % 256 == 0, m > 512, C with ceil256(m) rows                              */
} kemr_epilogue;
/* A bf16 [m_alloc, k] and C [m_alloc, n] with m_alloc = m rounded up to 256 rows (pad rows of A are read; pad rows of C
 * may be written by the bf16 epilogues), W bf16 [n, k], bias fp32 [n] */
/* ---- image preprocessing on the GPU ------------------------------------------------
 * Replaces the per-sample host transform the reference gets from clip.load (src/clip/datasets/clip_dataset.py:110-125):
 * uint8 HWC RGB [height, width, 3] on the device -> Resize(n_px, bicubic, shorter side; Pillow's antialiased two-pass
 * filter with 8-bit intermediates, bit-exact) -> CenterCrop(n_px) -> /255 -> (x - mean) / std -> fp32 CHW [3, n_px, n_px].
 * The workspace holds the coefficient tables (uploaded by the call) and the horizontally filtered rows. */
size_t kemr_preprocess_workspace_bytes(int height, int width, int n_px);
int kemr_preprocess_u8(const unsigned char* img_dev, int height, int width, int n_px, float* out_dev,
                       void* workspace_dev, size_t workspace_bytes, void* stream);
/* A whole loader batch in ONE launch pair: `batch` uint8 HWC images of any sizes packed back to back in packed_dev
 * (image b starts at byte offsets[b] and has heights[b] x widths[b] pixels; the three arrays are HOST arrays) ->
 * out_dev fp32 [batch, 3, n_px, n_px].  Same arithmetic, bit for bit, as kemr_preprocess_u8 per image.
 * An item with heights[b] == widths[b] == 0 is one the caller could not decode: it has no bytes in packed_dev and its
 * output is 0.0f everywhere, i.e. zeros AFTER normalisation -- the tensor the reference's dataset substitutes
 * (src/clip/datasets/clip_dataset.py:120-125, torch.zeros(3, 224, 224)).  Any other non-positive size is KEMR_ERR_INVALID. */
size_t kemr_preprocess_batch_workspace_bytes(const int32_t* heights, const int32_t* widths, int batch, int n_px);
int kemr_preprocess_u8_batch(const unsigned char* packed_dev, const int64_t* offsets, const int32_t* heights,
                             const int32_t* widths, int batch, int n_px, float* out_dev, void* workspace_dev,
                             size_t workspace_bytes, void* stream);

int kemr_op_gemm(const void* a_dev, const void* w_dev, const float* bias_dev, void* c_dev,
                 int m, int n, int k, int epilogue, void* stream);
int kemr_op_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev, void* y_dev,
                      int rows, int width, int out_dtype /*KEMR_BF16|KEMR_F32*/, void* stream);
/* fp8 operands: a e4m3 [ceil256(m), k] and w e4m3 [n, k] (bytes), wscale fp32 [n] multiplies the accumulators per output
 * channel before the bias; c bf16 [ceil256(m), n]; n % 256 == 0, k % 128 == 0, k >= 256; epilogue BIAS_BF16 | BIAS_QGELU_BF16 */
int kemr_op_gemm_fp8(const void* a_dev, const void* w_dev, const float* wscale_dev, const float* bias_dev, void* c_dev,
                     int m, int n, int k, int epilogue, void* stream);
/* the host-side fp32 -> e4m3 conversion used when weights are packed (round to nearest even, saturating at +-448); no GPU */
int kemr_op_e4m3_host(const float* in, unsigned char* out, long long n);
/* fused residual form used inside the towers: x_f32 += delta_bf16 (written back), y_bf16 = LayerNorm(x) */
int kemr_op_layernorm_resid(float* x_dev, const void* delta_dev, const float* gamma_dev, const float* beta_dev,
                            void* y_dev, int rows, int width, void* stream);
/* general form: rows of x_dtype (KEMR_F32|KEMR_BF16); y = LayerNorm(x [+ delta [+ delta2]]) (deltas bf16, may be NULL);
 * writeback != 0 stores the sum back into x (required with delta2); without deltas x is only read, unless y_dev == x_dev */
int kemr_op_layernorm_rows(void* x_dev, int x_dtype, const void* delta_dev, const void* delta2_dev, int writeback,
                           const float* gamma_dev, const float* beta_dev, void* y_dev, int rows, int width, int out_dtype,
                           void* stream);
/* qkv bf16 [batch*t, 3*width] (q pre-scaled by 1/8) -> out bf16 [batch*t, width] */
int kemr_op_attention(const void* qkv_dev, void* out_dev, int batch, int t, int width, int causal, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KEMR_H_ */
