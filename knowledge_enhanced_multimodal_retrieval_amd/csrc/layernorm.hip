// LayerNorm over the residual stream: fp32 (or, KEMR_PREC_BF16_RES16, bf16) rows in, bf16 (GEMM operand) or fp32
// (ln_pre, in place) out.  HBM-bound: one wave per row, the whole row lives in registers (16 B / lane loads for fp32
// rows, 8 B for bf16), mean and variance by wave shuffles in fp32 (two-pass, like torch's fp32 LayerNorm), 8-byte
// bf16 stores.  Algorithmic bytes per row: width * (4|2 read + 2 written) [+ 8 * width once for gamma/beta, L2-resident].
#include "common.h"

namespace kemr {

// MODE: the residual updates of the preceding GEMMs are fused here.  Those GEMMs stored their outputs (bias included) as
// bf16 "deltas" (store-only epilogues); this kernel adds them while it reads x anyway.
//   MODE 1 (ln_1 of a block):  x += d1 (+ d2), written back, y = LN(x)    4 + 2 (+ 2) bytes read, 4 + 2 written per element
//   MODE 2 (ln_2 of a block):  y = LN(x + d1), x NOT written back         4 + 2 read, 2 written: the attention delta d1
//                              stays pending and is added for good, together with the MLP delta, by the next ln_1
// (one x write per block instead of two; the sums are formed in the same order as two separate updates, so an fp32
// stream holds bit-identical values).
// (W = the row's width in elements; only the 24-bit rows need it: their third bytes sit behind the W upper halves)
__device__ __forceinline__ float4 load_row4(const float* r, int i, int) { return ((const float4*)r)[i]; }
__device__ __forceinline__ float4 load_row4(const f24_t* r, int i, int W) {
    const uint2 h = ((const uint2*)r)[i];
    const uint32_t l = ((const uint32_t*)((const uint8_t*)r + 2 * (size_t)W))[i];
    return make_float4(f24_to_f32((bf16_t)(h.x & 0xffff), (uint8_t)l), f24_to_f32((bf16_t)(h.x >> 16), (uint8_t)(l >> 8)),
                       f24_to_f32((bf16_t)(h.y & 0xffff), (uint8_t)(l >> 16)), f24_to_f32((bf16_t)(h.y >> 16), (uint8_t)(l >> 24)));
}
// A/B (round 4): the residual rows by non-temporal loads too
typedef unsigned u32x2_nt __attribute__((ext_vector_type(2)));
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_row4_nt(const float* r, int i, int) {
    const f32x4_nt v = __builtin_nontemporal_load((const f32x4_nt*)r + i);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 load_row4_nt(const f24_t* r, int i, int W) {
    const u32x2_nt h = __builtin_nontemporal_load((const u32x2_nt*)r + i);
    const uint32_t l = __builtin_nontemporal_load((const uint32_t*)((const uint8_t*)r + 2 * (size_t)W) + i);
    return make_float4(f24_to_f32((bf16_t)(h.x & 0xffff), (uint8_t)l), f24_to_f32((bf16_t)(h.x >> 16), (uint8_t)(l >> 8)),
                       f24_to_f32((bf16_t)(h.y & 0xffff), (uint8_t)(l >> 16)), f24_to_f32((bf16_t)(h.y >> 16), (uint8_t)(l >> 24)));
}
__device__ __forceinline__ float4 load_row4_nt(const bf16_t* r, int i, int) {
    const u32x2_nt d = __builtin_nontemporal_load((const u32x2_nt*)r + i);
    return make_float4(bf16_to_f32((bf16_t)(d.x & 0xffff)), bf16_to_f32((bf16_t)(d.x >> 16)),
                       bf16_to_f32((bf16_t)(d.y & 0xffff)), bf16_to_f32((bf16_t)(d.y >> 16)));
}
__device__ __forceinline__ void store_row4_nt(float* r, int i, float4 v, int) {
    f32x4_nt t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (f32x4_nt*)r + i);
}
__device__ __forceinline__ void store_row4_nt(f24_t* r, int i, float4 v, int W) {
    const uint32_t a = f32_to_f24_bits(v.x), b = f32_to_f24_bits(v.y), c = f32_to_f24_bits(v.z), d = f32_to_f24_bits(v.w);
    u32x2_nt h = {(a >> 16) | (b & 0xffff0000u), (c >> 16) | (d & 0xffff0000u)};
    __builtin_nontemporal_store(h, (u32x2_nt*)r + i);
    __builtin_nontemporal_store(((a >> 8) & 0xff) | (b & 0xff00) | ((c << 8) & 0xff0000) | ((d << 16) & 0xff000000u), (uint32_t*)((uint8_t*)r + 2 * (size_t)W) + i);
}
__device__ __forceinline__ void store_row4_nt(bf16_t* r, int i, float4 v, int) {
    u32x2_nt pk = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    __builtin_nontemporal_store(pk, (u32x2_nt*)r + i);
}
// the deltas (out-proj / fc2 outputs) are read exactly once, by the LayerNorm that applies them: a non-temporal load
__device__ __forceinline__ float4 load_delta4(const bf16_t* r, int i) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ d = __builtin_nontemporal_load((const u32x2_*)r + i);
    return make_float4(bf16_to_f32((bf16_t)(d.x & 0xffff)), bf16_to_f32((bf16_t)(d.x >> 16)),
                       bf16_to_f32((bf16_t)(d.y & 0xffff)), bf16_to_f32((bf16_t)(d.y >> 16)));
}
__device__ __forceinline__ float4 load_row4(const bf16_t* r, int i, int) {
    const uint2 d = ((const uint2*)r)[i];
    return make_float4(bf16_to_f32((bf16_t)(d.x & 0xffff)), bf16_to_f32((bf16_t)(d.x >> 16)),
                       bf16_to_f32((bf16_t)(d.y & 0xffff)), bf16_to_f32((bf16_t)(d.y >> 16)));
}
__device__ __forceinline__ void store_row4(float* r, int i, float4 v, int) { ((float4*)r)[i] = v; }
__device__ __forceinline__ void store_row4(f24_t* r, int i, float4 v, int W) {
    const uint32_t a = f32_to_f24_bits(v.x), b = f32_to_f24_bits(v.y), c = f32_to_f24_bits(v.z), d = f32_to_f24_bits(v.w);
    uint2 h;
    h.x = (a >> 16) | (b & 0xffff0000u);
    h.y = (c >> 16) | (d & 0xffff0000u);
    ((uint2*)r)[i] = h;
    ((uint32_t*)((uint8_t*)r + 2 * (size_t)W))[i] = ((a >> 8) & 0xff) | (b & 0xff00) | ((c << 8) & 0xff0000) | ((d << 16) & 0xff000000u);
}
// (a row that was just stored as 24-bit floats is read back rounded: the statistics of ln_1 use the fp32 sum, like for a bf16 stream)
__device__ __forceinline__ void store_row4(bf16_t* r, int i, float4 v, int) {
    uint2 pk;
    pk.x = pack_bf16x2(v.x, v.y);
    pk.y = pack_bf16x2(v.z, v.w);
    ((uint2*)r)[i] = pk;
}

struct fp8_t { uint8_t v; };          // output tag: OCP e4m3 bytes (the A operand of the fp8 GEMMs), saturating.  The per-channel scale of the
                                      // operand lives in gamma / beta (api.hip finalize divides them by 2^ceil(log2 max(|gamma|, |beta|))
                                      // and multiplies the weight columns): the values written here are |z * gamma' + beta'| <= ~45
__device__ __forceinline__ void store_row4(fp8_t* r, int i, float4 v, int) {
    const float lim = 448.f;
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.x, -lim, lim), __builtin_amdgcn_fmed3f(v.y, -lim, lim), 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.z, -lim, lim), __builtin_amdgcn_fmed3f(v.w, -lim, lim), pk, true);
    ((int*)r)[i] = pk;
}

// NTD (round 4; 3 = the product kernel): cache hints for bytes that are touched exactly once here.  The deltas are dead after this read; the
// residual row is next read a millisecond and 1.5 GB of traffic later; as ordinary loads / stores they pushed what IS reused soon -- the
// GEMM operand h this kernel writes, the q | k | v rows the next kernels produce -- out of L2 / Infinity Cache.  Measured in the chain,
// interleaved A/B rounds of 40 steps on one box, bit-identical results (debug switch ln_nt in A/B builds; profiles/r04_cache_hints.txt):
//   0 plain loads and stores (rounds 1-3)        18 298 items/s   LayerNorm 5.59 ms   GEMM 31.75   attention 4.15
//   1 deltas non-temporal                        18 360           5.44                31.70        4.18
//   2 + residual rows loaded non-temporally      18 439           5.42                31.57        4.18
//   3 + the ln_1 write-back of x non-temporal    + 0.2 % over 2   5.48                31.35        4.04
template <int NV, typename XT, typename OutT, int MODE, int NTD = 3>   // width = NV * 256
__global__ __launch_bounds__(256) void layernorm_kernel(XT* x, const bf16_t* __restrict__ delta, const bf16_t* __restrict__ delta2,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        OutT* y, int rows, float eps) {
    constexpr int W = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    XT* xr = x + (size_t)row * W;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = NTD >= 2 ? load_row4_nt(xr, i * 64 + lane, W) : load_row4(xr, i * 64 + lane, W);
        if constexpr (MODE != 0) {
            const float4 d = NTD >= 1 ? load_delta4(delta + (size_t)row * W, i * 64 + lane) : load_row4(delta + (size_t)row * W, i * 64 + lane, W);
            v[i].x += d.x; v[i].y += d.y; v[i].z += d.z; v[i].w += d.w;
            if (MODE == 1 && delta2) {               // wave-uniform
                const float4 e = NTD >= 1 ? load_delta4(delta2 + (size_t)row * W, i * 64 + lane) : load_row4(delta2 + (size_t)row * W, i * 64 + lane, W);
                v[i].x += e.x; v[i].y += e.y; v[i].z += e.z; v[i].w += e.w;
            }
            if constexpr (MODE == 1) {                 // a bf16 stream rounds here; the statistics use the fp32 sum
                if constexpr (NTD >= 3) store_row4_nt(xr, i * 64 + lane, v[i], W);
                else store_row4(xr, i * 64 + lane, v[i], W);
            }
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / W);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / W) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 g = ((const float4*)gamma)[i * 64 + lane];
        const float4 b = ((const float4*)beta)[i * 64 + lane];
        float4 o;
        o.x = v[i].x * rstd * g.x + b.x;
        o.y = v[i].y * rstd * g.y + b.y;
        o.z = v[i].z * rstd * g.z + b.z;
        o.w = v[i].w * rstd * g.w + b.w;
        store_row4(y + (size_t)row * W, i * 64 + lane, o, W);
    }
}

int g_ln_nt = 3;          // A/B builds: 0 / 1 / 2 = the earlier hint levels (see NTD above)

template <int NV, typename XT>
static int launch_nv(XT* x, const bf16_t* d1, const bf16_t* d2, int writeback, const float* g, const float* b, void* y, int rows,
                     int out_dtype, hipStream_t s) {
    const int blocks = (rows + 3) / 4;
    ProfScope prof(PROF_LAYERNORM, s);
    if (d1 && writeback && out_dtype == KEMR_FP8)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, fp8_t, 1>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (fp8_t*)y, rows, 1e-5f);
    else if (d1 && out_dtype == KEMR_FP8)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, fp8_t, 2>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (fp8_t*)y, rows, 1e-5f);
    else if (out_dtype == KEMR_FP8)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, fp8_t, 0>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (fp8_t*)y, rows, 1e-5f);
#ifdef KEMR_AB_VARIANTS
    else if (d1 && writeback && g_ln_nt == 0)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, bf16_t, 1, 0>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (bf16_t*)y, rows, 1e-5f);
    else if (d1 && g_ln_nt == 0)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, bf16_t, 2, 0>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (bf16_t*)y, rows, 1e-5f);
    else if (d1 && writeback && g_ln_nt == 1)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, bf16_t, 1, 1>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (bf16_t*)y, rows, 1e-5f);
    else if (d1 && g_ln_nt == 1)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, bf16_t, 2, 1>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (bf16_t*)y, rows, 1e-5f);
    else if (d1 && writeback && g_ln_nt == 2)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, bf16_t, 1, 2>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (bf16_t*)y, rows, 1e-5f);
#endif
    else if (d1 && writeback)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, bf16_t, 1>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (bf16_t*)y, rows, 1e-5f);
    else if (d1)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, bf16_t, 2>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (bf16_t*)y, rows, 1e-5f);
    else if (out_dtype == KEMR_BF16)
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, bf16_t, 0>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (bf16_t*)y, rows, 1e-5f);
    else if (out_dtype == KEMR_F24) {                  // ln_pre into a 24-bit stream: fp32 rows in only
        if constexpr (sizeof(XT) == 4)
            hipLaunchKernelGGL((layernorm_kernel<NV, XT, f24_t, 0>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (f24_t*)y, rows, 1e-5f);
        else
            KEMR_FAIL(KEMR_ERR_INVALID, "layernorm: 24-bit output rows need fp32 input rows");
    } else
        hipLaunchKernelGGL((layernorm_kernel<NV, XT, float, 0>), dim3(blocks), dim3(256), 0, s, x, d1, d2, g, b, (float*)y, rows, 1e-5f);
    KEMR_CHECK_LAUNCH("layernorm_kernel");
    return KEMR_OK;
}

template <typename XT>
static int launch_xt(XT* x, const bf16_t* d1, const bf16_t* d2, int writeback, const float* gamma, const float* beta, void* y,
                     int rows, int width, int out_dtype, hipStream_t stream) {
    switch (width) {
        case 256:  return launch_nv<1>(x, d1, d2, writeback, gamma, beta, y, rows, out_dtype, stream);
        case 512:  return launch_nv<2>(x, d1, d2, writeback, gamma, beta, y, rows, out_dtype, stream);
        case 768:  return launch_nv<3>(x, d1, d2, writeback, gamma, beta, y, rows, out_dtype, stream);
        case 1024: return launch_nv<4>(x, d1, d2, writeback, gamma, beta, y, rows, out_dtype, stream);
        case 1280: return launch_nv<5>(x, d1, d2, writeback, gamma, beta, y, rows, out_dtype, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "layernorm: width %d not in {256,512,768,1024,1280}", width);
}

// x_dtype: KEMR_F32 or KEMR_BF16 rows.  delta != nullptr: LN(x + delta [+ delta2]); with `writeback` the sum replaces x
// (delta2 needs writeback); the output is bf16 then.
int launch_layernorm(void* x, int x_dtype, const bf16_t* delta, const bf16_t* delta2, int writeback, const float* gamma,
                     const float* beta, void* y, int rows, int width, int out_dtype, hipStream_t stream) {
    if (rows <= 0) return KEMR_OK;
    if (out_dtype != KEMR_BF16 && out_dtype != KEMR_F32 && out_dtype != KEMR_FP8 && out_dtype != KEMR_F24) KEMR_FAIL(KEMR_ERR_INVALID, "layernorm: bad out dtype %d", out_dtype);
    if (x_dtype != KEMR_BF16 && x_dtype != KEMR_F32 && x_dtype != KEMR_F24) KEMR_FAIL(KEMR_ERR_INVALID, "layernorm: bad row dtype %d", x_dtype);
    if (delta && (out_dtype == KEMR_F32 || out_dtype == KEMR_F24)) KEMR_FAIL(KEMR_ERR_INVALID, "layernorm: the residual forms write bf16 or fp8");
    if (delta2 && !(delta && writeback)) KEMR_FAIL(KEMR_ERR_INVALID, "layernorm: a second delta needs the first one and writeback");
    if (x_dtype == KEMR_BF16) return launch_xt((bf16_t*)x, delta, delta2, writeback, gamma, beta, y, rows, width, out_dtype, stream);
    if (x_dtype == KEMR_F24) return launch_xt((f24_t*)x, delta, delta2, writeback, gamma, beta, y, rows, width, out_dtype, stream);
    return launch_xt((float*)x, delta, delta2, writeback, gamma, beta, y, rows, width, out_dtype, stream);
}

}  // namespace kemr
