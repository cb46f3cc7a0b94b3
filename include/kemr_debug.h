/*
 * kemr_debug.h -- experiment switches and diagnostics of libkemr.so for tools/ and tests/.
 *
 * NOT part of the product ABI (include/kemr.h): everything here is process-wide, not thread-safe, and may change a caller's
 * timing or route -- never its results, which tests/ check on every route.  Per-model behaviour that does change numerics
 * (the residual fusion) is a model option in kemr.h (kemr_model_set_option), not a switch here.
 */
#ifndef KEMR_DEBUG_H_
#define KEMR_DEBUG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One key per switch; a call touches that switch only.
 * PRODUCT library (the default build): a switch only ROUTES between kernels the product needs anyway.  The experiment kernels of
 * rounds 1-3 (earlier GEMM generations, the staggered 256x256 GEMM, the long-interval K loop, the stamped / store-dropping GEMM
 * instantiations, the attention variants) are compiled only by `python -m knowledge_enhanced_multimodal_retrieval_amd.build
 * --ab-variants`; without them every value marked [A/B] below is refused with KEMR_ERR_INVALID.  "ab_variants" (read-only) says
 * which library this is.
 *   "gemm_variant"  0 = automatic (default), 1 = 128x128x64 / 4 waves, 2 = 256x256x64 / 8 waves in lockstep,
 *                   7 = persistent 8 waves with one K-tile pipeline across tiles (the automatic choice from 128 tiles up),
 *                   8 = skinny-M split-K (automatic up to 512 rows); [A/B] 3 = 256x256x64 staggered, 4, 5, 6, 9 = earlier
 *                   persistent generations
 *   "gemm_flags"    [A/B] timing experiments of the persistent GEMM's diagnostic instantiation: 1 = drop the C stores, 4 = plain
 *                   instead of non-temporal stores, 64 (+ 32) = cycle stamps per barrier interval, 128 = whole-kernel clock
 *   "gemm_order"    tile order of the persistent GEMM: 0 = N fastest, else log2(column-group width) + 1 (default 3)
 *   "gemm_grid"     cap on the persistent GEMM's grid = the CUs it may take (0 = one workgroup per CU, the default); results do not depend on it
 *   "gemm_conc"     both wave halves' epilogues in one barrier interval: 0 never, 1 always, 2 = QuickGELU epilogue only (default)
 *   "gemm_kl"       K loop of the persistent GEMM: 0 = eight 256-cycle barrier intervals per K-tile (the product loop),
 *                   [A/B] 1 = four of 512
 *   "attn_v"        attention kernel at T = 257: 0 = the product kernel (16-query tiles, 4 waves); [A/B, csrc/attention_ab.hip]
 *                   1 = 32-query tiles on v_mfma_f32_32x32x16_bf16, 2 = eight waves per workgroup with the keys in two halves
 *                   (online softmax), 3 = as 2, one persistent workgroup per CU, the next head's K / V by LDS-DMA while this one
 *                   is computed, 4 = two query tiles per pass sharing the K / V fragments (half the LDS bytes per tile)
 *   "attn_xcd"      attention: 1 = XCD x computes the images = x (mod 8), the heads of an image next to each other in time
 *                   (default), 0 = grid order
 *   "attn_waves"    [A/B] waves per workgroup at T = 257: 0 = default (4 for attn_v 0, 16 for attn_v 3), 6 (attn_v 0), 8 (attn_v 3);
 *                   2 (attn_v 0) = the build with s_memtime stamps, which writes 8 counters per wave BEHIND the output
 *                   (tools/prof_attention.py allocates the room; nothing else may select it)
 *   "sim_lists"     kemr_sim_topk: 0 = never the candidate-list route (nor the fast rank pass for bonus lists), 1 = where it pays
 *                   (default: from 2 048 gallery rows, 256 queries and 1.2e10 multiply-adds up), 3 = wherever it fits, 2 = as 3 and
 *                   the exact fallback forced to run after the lists
 *   "ln_nt"         LayerNorm cache hints: 3 = deltas and residual rows loaded, the ln_1 write-back of x stored non-temporally (the product
 *                   kernel); [A/B] 0 = plain (rounds 1-3), 1 = deltas only, 2 = deltas + row loads
 *   "attn_waves"    also: [A/B] 7 = the vision attention kernel with plain K / V loads (rounds 1-3), 5 = Q rows non-temporal as well,
 *                   8 (with attn_v 0) = TIMING ONLY, wrong results: the ragged 17th query tile (one valid row) is not computed
 *   "ab_variants"   read-only: 1 = the library holds the [A/B] kernels */
int kemr_debug_set(const char* key, int value);
int kemr_debug_get(const char* key, int* value);

/* flag / longest list / capacity / chunks / sampled rows / records per query left in `workspace` by the last kemr_sim_topk of
 * these sizes on the candidate-list route (all zero when the route does not apply; synchronises the device) */
int kemr_debug_sim_lists(const void* workspace_dev, int nq, int ng, int64_t kdim, int k, int32_t* out6);

/* cycle sums the persistent GEMM's diagnostic instantiation left behind ("gemm_flags" 64): 16 words per workgroup = the barrier
 * intervals of the K loop, K-loop tail, epilogue, tiles, K-tiles per tile.  Synchronises the device. */
int kemr_debug_gemm_stamps(unsigned* host_out, int n_words);

#ifdef __cplusplus
}
#endif
#endif /* KEMR_DEBUG_H_ */
