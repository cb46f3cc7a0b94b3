// Shared declarations for libkemr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/kemr.h"

namespace kemr {

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment: 8 bf16 = 4 VGPRs
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // MFMA 16x16 C/D fragment

// ---- error plumbing (thread-local message, never throws) -----------------------------------
void set_error(const char* fmt, ...);
#define KEMR_FAIL(code, ...) do { ::kemr::set_error(__VA_ARGS__); return (code); } while (0)
#define KEMR_CHECK_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    ::kemr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    return KEMR_ERR_HIP; } } while (0)
#define KEMR_CHECK_LAUNCH(what) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    ::kemr::set_error("launch of %s failed: %s (%s:%d)", what, hipGetErrorString(e_), __FILE__, __LINE__); \
    return KEMR_ERR_HIP; } } while (0)
#define KEMR_TRY(expr) do { int r_ = (expr); if (r_ != KEMR_OK) return r_; } while (0)

// ---- host helpers ---------------------------------------------------------------------------
static inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u; __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);                 // round to nearest even
}
static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
// fp32 -> OCP fp8 e4m3fn (1-4-3, bias 7, max 448, no inf), round to nearest even, saturating; NaN -> 0x7f
static inline uint8_t f32_to_e4m3_host(float f) {
    if (f != f) return 0x7f;
    const uint8_t sign = f < 0 ? 0x80 : 0;
    float a = f < 0 ? -f : f;
    if (a >= 448.f) return sign | 0x7e;
    if (a < 0.0009765625f) return sign;                     // below half of the smallest subnormal (2^-9): zero
    int e;
    (void)__builtin_frexpf(a, &e);                          // a = m * 2^e, m in [0.5, 1)  ->  floor(log2 a) = e - 1
    e -= 1;
    if (e < -6) {                                           // subnormal: multiples of 2^-9
        const float q = a * 512.f;
        int m = (int)__builtin_nearbyintf(q);               // default rounding mode = nearest even
        return sign | (uint8_t)m;                           // m == 8 is the smallest normal 0x08
    }
    int m = (int)__builtin_nearbyintf(__builtin_ldexpf(a, 3 - e));      // in [8, 16]
    if (m == 16) { m = 8; e += 1; }
    if (e > 8 || (e == 8 && m > 14)) return sign | 0x7e;
    return sign | (uint8_t)(((e + 7) << 3) | (m - 8));
}

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// Residual-stream rows as 24-bit floats (internal dtype code KEMR_F24): sign, 8 exponent bits, 15 mantissa bits = the upper three bytes
// of the fp32, rounded to nearest.  A ROW of W elements is stored as its W bf16 upper halves followed by its W third bytes (3 W
// bytes, so `f24_t* + row * W` is the row's address like for any other element type); 2^-16 relative per rounding, 128 x finer than
// bf16, for 3 instead of 4 bytes per element in the LayerNorm passes, which are HBM-bound.
constexpr int KEMR_F24 = 24;
struct __attribute__((packed)) f24_t { uint8_t b[3]; };
__device__ __forceinline__ float f24_to_f32(bf16_t hi, uint8_t lo) { return __uint_as_float(((uint32_t)hi << 16) | ((uint32_t)lo << 8)); }
__device__ __forceinline__ uint32_t f32_to_f24_bits(float f) { return __float_as_uint(f) + 0x80u; }      // use bits 31..8
// RNE; inputs on this path are finite (a NaN would come out as NaN-or-inf, MI355X_MICROARCH "Correctness boundaries")
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// two fp32 -> packed bf16x2 (RNE) in ONE instruction: hipcc lowers the vector cast to v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 kemr_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float kemr_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const kemr_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, kemr_bf16x2_t));
}
// QuickGELU x * sigmoid(1.702 x) with v_exp_f32 / v_rcp_f32 (exp2 with the constant folded; rcp is 1 ulp)
__device__ __forceinline__ float quick_gelu(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
}
// two at a time: the scale, the + 1 and the final product as packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32, same
// IEEE results as the scalar forms), the two transcendentals per element stay
typedef float __attribute__((ext_vector_type(2))) f32x2_t;
__device__ __forceinline__ f32x2_t quick_gelu2(f32x2_t v) {
    const f32x2_t t = v * (-1.702f * 1.4426950408889634f);
    f32x2_t e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2_t d = e + 1.0f;
    f32x2_t r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    return v * r;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- optional per-kernel-class event timing (api.hip); used by bench.py for the roofline line --------
enum ProfClass : int { PROF_GEMM = 0, PROF_LAYERNORM = 1, PROF_ATTENTION = 2, PROF_OTHER = 3, PROF_SIM = 4, PROF_NCLASS = 5 };
struct ProfScope {            // records a start/stop hipEvent pair on `s` when profiling is enabled
    ProfScope(int cls, hipStream_t s);
    ~ProfScope();
    int slot; hipStream_t stream;
};

// ---- GEMM launcher (gemm.hip) -----------------------------------------------------------------
enum GemmEpi : int {
    EPI_BIAS_BF16 = KEMR_EPI_BIAS_BF16,
    EPI_BIAS_QGELU_BF16 = KEMR_EPI_BIAS_QGELU_BF16,
    EPI_BIAS_RESID_F32 = KEMR_EPI_BIAS_RESID_F32,
    EPI_PATCH_F32 = 3,
    EPI_BIAS_RESADD_BF16 = KEMR_EPI_BIAS_RESADD_BF16,   // X_bf16 = bf16(bf16(A.W^T + bias) + X_bf16), in place (gemm256u only)     // X_f32[remap(m)] = acc + pos[1 + m % tokens_per_img]   (patch embedding)
};
struct GemmParams {
    const bf16_t* A;     // [m_alloc, lda] bf16, K contiguous
    const bf16_t* W;     // [N, ldw] bf16, K contiguous (PyTorch Linear layout)
    const float* bias;   // [N] or nullptr
    const float* wscale; // fp8 operands only: [N] per-output-channel scale applied to the accumulators before the bias
    void* C;             // bf16 [m_alloc, ldc] or fp32 [*, ldc] depending on the epilogue
    const float* pos;    // EPI_PATCH_F32: positional embedding [1 + patches, N]
    int M, N, K;         // M = valid rows (stores are guarded), N % 128 == 0, K % 64 == 0
    int lda, ldw, ldc;
    int patches;         // EPI_PATCH_F32: patches per image (row remap m -> m + m / patches + 1)
    int dbg;             // timing experiments only (tools/): 1 = skip the epilogue stores, 2 = drain vmcnt at every wait (gemm256p/q); gemm256u: 4 = plain instead of nt stores, 32 = half the stores
    int c_rows_padded;   // C has ceil256(M) writable rows (lets the persistent kernel store without row masks)
    int order;           // gemm256u tile order: 0 = N fastest, else log2(column-group width) + 1 (see gemm256u.hip)
    unsigned* stamps;    // gemm256u DBG instantiation with dbg & 64: per-workgroup cycle sums (tools/)
    // gemm256u SIM instantiation (rank-only similarity pass): A = query panel [M = nq], W = gallery panel [N = ceil256(ng)]
    const int32_t* sim_gt;    // [nq] GLOBAL candidate id of each query's ground truth
    const float* sim_sgt;     // [nq] its score (kemr_pair_scores arithmetic)
    int32_t* sim_ahead;       // [nq] += candidates of this gallery ranked ahead of it
    int sim_ng, sim_gbase;    // valid gallery rows; global id of gallery row 0
    int sim_nchunks, sim_tpc; // gallery chunks per query tile, gallery tiles per chunk
    int sim_xcd;              // 1: the chunks of a query tile are dealt to one XCD (gemm256u.hip, SIM prologue)
    // SIM == 2 (top-k candidates on top of the rank count): a lane whose 16 candidates of a query hold a score >= the query's
    // threshold appends all 16 scores as one record to the (query, chunk) list
    const float* simk_taud;   // [nq] next float below the threshold (s > taud  <=>  s >= threshold)
    float* simk_scores;       // [nq][nchunks][cap][16]
    int32_t* simk_base;       // [nq][nchunks][cap] global id of a record's first candidate (element j: + (j >> 2) * 16 + (j & 3))
    int32_t* simk_count;      // [nq][nchunks] records appended (clamped to cap)
    int32_t* simk_flag;       // |= 1 when a list overflowed: the caller's exact fallback runs
    int simk_cap;
};
extern int g_gemm_dbg;
extern int g_gemm_order;
extern int g_sim_lists;     // sim.hip: candidate-list route of kemr_sim_topk (kemr_debug_set "sim_lists")
extern int g_attn_xcd;      // attention.hip (tools)
extern int g_attn_waves;    // attention.hip (tools)
extern int g_attn_v;        // attention.hip: 0 = the product kernel; 1..4 = attention_ab.hip (A/B builds only)
extern int g_gemm_kl;       // gemm256u: 0 = eight barrier intervals per K-tile (the product loop), 1 = the long-interval K loop (A/B builds only)
extern int g_ln_nt;         // layernorm.hip: cache-hint level of the residual forms, 3 = the product kernel; 0 / 1 / 2 in A/B builds only
extern int g_gemm_grid;     // tools: cap on the persistent GEMM's grid (0 = one workgroup per CU)
extern int g_gemm_conc;     // gemm256u: both wave halves run their epilogues in the same barrier interval (0 never, 1 always, 2 = QuickGELU epilogue only)
int gemm_read_stamps(unsigned* host_out, int n_words);
int launch_gemm256u_simgmax(const bf16_t* q_panel, int nq, const bf16_t* g_panel, int ng, int kdim, int stride, float* out,
                            hipStream_t stream, bool* used);
struct SimkPlan { int nchunks, tpc, cap; size_t scores_bytes, base_bytes, count_bytes; };
int gemm256u_simk_plan(int nq, int ng, int kdim, double hits_per_query, double spread, SimkPlan* plan, bool* ok);
int launch_gemm256u_simk(const bf16_t* q_panel, int nq, const bf16_t* g_panel, int ng, int kdim, long long gallery_offset,
                         const int32_t* gt_idx, const float* gt_score, int32_t* ahead, const float* taud, const SimkPlan& plan,
                         float* rec_scores, int32_t* rec_base, int32_t* rec_count, int32_t* flag, hipStream_t stream);
int launch_gemm(const GemmParams& p, int epi, hipStream_t stream);      // picks the tile variant
int launch_gemm256(const GemmParams& p, int epi, hipStream_t stream);   // gemm256.hip: 256x256x64, 8 waves, counted vmcnt
int launch_gemm256p(const GemmParams& p, int epi, hipStream_t stream);  // gemm256p.hip: persistent, async epilogue (bf16-store epilogues)
int launch_gemm256q(const GemmParams& p, int epi, hipStream_t stream);  // gemm256q.hip: persistent, 2 long phases per K-tile
int launch_gemm256u(const GemmParams& p, int epi, hipStream_t stream);  // gemm256u.hip: gemm256p's K loop, one K-tile pipeline across tiles
bool gemm256u_fits(const GemmParams& p, int elem_size, int c_elem_size = 2);
// rank-only similarity pass on the persistent GEMM's K loop (k == 0, no bonus); false in *used when the shape does not fit it
int launch_gemm256u_simrank(const bf16_t* q_panel, int nq, const bf16_t* g_panel, int ng, int kdim, long long gallery_offset,
                            const int32_t* gt_idx, const float* gt_score, int32_t* ahead, hipStream_t stream, bool* used);               // gemm256u.hip: inside its tile table / 32-bit offsets
int launch_gemm256u_fp8(const GemmParams& p, int epi, hipStream_t stream);  // gemm256u.hip: fp8 e4m3 operands (A, W in bytes), bf16 C
int launch_gemm256r(const GemmParams& p, int epi, hipStream_t stream);  // gemm256r.hip: persistent, 4 waves x 128x128, register-staged operands
int launch_gemm_skinny(const GemmParams& p, int epi, hipStream_t stream);   // gemm_skinny.hip: M <= 512 rows (online queries), split-K over 8 waves
int launch_gemm256w(const GemmParams& p, int epi, hipStream_t stream);  // gemm256w.hip: persistent, 4 waves x 128x128 (AGPR accumulators)
extern int g_gemm_variant;   // 0 auto, 1 = 128x128 (gemm.hip), 2 / 3 = 256x256 lockstep / staggered, 4 = persistent 256x256 (bf16 epilogues)

// ---- other launchers --------------------------------------------------------------------------
// delta != nullptr: x += delta (bf16 [rows, width], the previous GEMM's output) is applied first and written back
// x_dtype (KEMR_F32 / KEMR_BF16) is the storage type of the residual rows
// delta2 (needs delta and writeback) is added as well; writeback == 0 leaves x as it is and normalises x + delta
int launch_layernorm(void* x, int x_dtype, const bf16_t* delta, const bf16_t* delta2, int writeback, const float* gamma,
                     const float* beta, void* y, int rows, int width, int out_dtype, hipStream_t stream);
int launch_attention(const bf16_t* qkv, bf16_t* out, int batch, int t, int width, int causal, hipStream_t stream);
// causal, items of lengths 1 .. max_t packed one behind the other: item b = rows row_start[b] .. row_start[b + 1] - 1 (device ints)
int launch_attention_packed(const bf16_t* qkv, bf16_t* out, const int* row_start, int batch, int max_t, int width, hipStream_t stream);
int launch_im2col(const float* pixels, bf16_t* patches, int batch, int image_size, int patch, int kpad, hipStream_t stream);
int launch_cls_rows(float* x, const float* class_emb, const float* pos, int batch, int tokens, int width, hipStream_t stream);
// the pooled row of every item (class token: ids == nullptr; else first argmax of the token ids, inside row_start's rows when packed)
int launch_pool_index(const int32_t* ids, const int* row_start, int batch, int tokens, int* pool_idx, hipStream_t stream);
// x[pool_idx[b]] -> xc[b], h[pool_idx[b]] -> hc[b]: compact copies of the pooled rows
int launch_gather_pooled(const void* x, int x_dtype, const void* h, int h_dtype /* KEMR_BF16 | KEMR_FP8 */, const int* pool_idx, int batch,
                         int width, void* xc, void* hc, hipStream_t stream);
// attention of the pooled row alone: q [items, width] compact, k / v from the call's qkv buffer, out [items, width] compact
int launch_attention_pooled(const bf16_t* q, const bf16_t* qkv, bf16_t* out, const int* pool_idx, const int* row_start, int items,
                            int tokens, int width, int causal, hipStream_t stream);
// exclusive prefix sums of the (clamped) text lengths: the packed-row layout of the text tower
int launch_row_starts(const int32_t* lens, int batch, int max_len, int rows, int* row_start, hipStream_t stream);
// row_start (optional, device, batch + 1 ints): packed rows -- text i contributes only its first row_start[i + 1] - row_start[i]
// positions, `rows` in total
int launch_text_embed(const int32_t* ids, const float* tok_emb, const float* pos, void* x, int x_dtype, int batch, int ctx,
                      int width, int vocab, hipStream_t stream, const int* row_start = nullptr, int rows = 0);
// pooled row -> LayerNorm -> @ proj [width, d] -> optional L2 normalise.  ids == nullptr: row = b * tokens (CLS)
// delta, delta2 (optional): the last block's pending residual updates, added to the pooled row
int launch_tail(const void* x, int x_dtype, const bf16_t* delta, const bf16_t* delta2, const int32_t* ids, int batch, int tokens, int width, const float* gamma,
                const float* beta, const float* proj, int d, int normalize, float* out, hipStream_t stream, const int* row_start = nullptr);

}  // namespace kemr
