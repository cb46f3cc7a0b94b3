// Multi-head self-attention for the CLIP towers (head dim 64; T = 257 / 197 / 50 non-causal, T = 77 causal).
//
// One workgroup of NW waves per (image or text, head): NW = 4, or 5 for the 77-token text tower (5 query tiles: one per wave
// instead of 2 / 1 / 1 / 1; with the key tiles behind the causal diagonal skipped: 27.7 -> 22.7 us per launch at B = 255).
// Tried and dropped for T = 257 (17 query tiles = 5 / 4 / 4 / 4 per wave; round 2, same device): 6 waves per workgroup
// (3 / 3 / 3 / 3 / 3 / 2) 157 -> 173 us (still there as NW = 6 for tools); the single-query 17th tile split over the keys
// of three waves with an LDS merge 157 -> 161 us.  Two workgroups share a CU and interleave; the fifth tile of wave 0 is not
// what the launch waits for.  The whole K and V of that head (T <= 288 keys,
// 2 x 36 KB) sit in LDS; every wave owns 16-query tiles and, because T is short, keeps the full score
// row in registers -- plain softmax, no online rescaling.  MFMA v_mfma_f32_16x16x32_bf16 throughout:
//   S^T tile = K_tile . Q^T      (A = K rows from LDS, B = Q rows straight from HBM)  -> a lane holds
//              4 consecutive keys of ONE query column, so row max / sum are 2 shuffles (xor 16, 32);
//   O^T      = V^T . P^T         (A = V^T fragment by ds_read_b64_tr_b16 from the row-major V image,
//                                 B = P^T fragment = the score registers, converted to bf16 in place).
// The k-slot order of the second product is permuted (slot 8*lq+j <-> key 32u + 16*(j>>2) + 4*lq + (j&3))
// identically for both operands, which is what lets P feed the MFMA without any lane movement.
// LDS images: K rows of 128 B with chunk ^= (row>>1)&7 (conflict-free ds_read_b128), V rows of 128 B
// with 32-byte-chunk ^= (row>>1)&3 (conflict-free transposed reads).  Pad keys are zero-filled and masked.
// The 1/sqrt(64) scale is folded into W_q / b_q when the weights are packed (exact: a power of two).
#include "common.h"

namespace kemr {

__device__ __forceinline__ bf16x4 lds_read_tr16(const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
}

// TC > 0: the sequence length is a compile-time constant (257 / 77: the shapes that matter), so every pad-key mask and
// tile-skip test folds away; TC == 0 keeps T a run-time value (other models, tests).  With a run-time T the uniform
// conditions of the 18 unrolled tiles overflowed the SGPR file (150+ v_readlane/v_writelane spills per query tile).
// Which (head, image) a workgroup computes.  xbatch == 0: blockIdx.x = head, blockIdx.y = image.  xbatch = the batch size: the
// grid is (heads, batch rounded up to 8) and the workgroups -- dealt to the eight XCDs round-robin by their linear id -- are
// renumbered so that XCD x computes the images = x (mod 8), all heads of an image next to each other in time: the 128-byte
// slices the sixteen heads take out of one token's 6 KB q|k|v row then come through ONE L2 at about the same time instead of
// through eight (round 3: 155 -> 150 us at B = 255 x 16 heads, bit-identical results; the default, debug switch attn_xcd).
__device__ __forceinline__ bool attn_item(int xbatch, int& h, int& b) {
    h = blockIdx.x; b = blockIdx.y;
    if (xbatch == 0) return true;
    const int id = blockIdx.y * gridDim.x + blockIdx.x, j = id >> 3;
    b = (j / (int)gridDim.x) * 8 + (id & 7);
    h = j % (int)gridDim.x;
    return b < xbatch;
}

// NTLOAD (round 4, the default): the q | k | v rows are read ONCE per launch (K / V by the one workgroup of their head, a Q row by one wave),
// so they are loaded non-temporally and do not displace the attention output -- which the out-proj GEMM reads next -- from L2 / Infinity
// Cache: vision launch 147.2 -> 144.6 us alone, 4.50 -> 4.43 ms per step in the chain, bit-identical (tools/bench_attention_ab.py 0:0,0:7).
// Wave priority by phase (compile-time A/B, -DKEMR_ATTN_PRIO=n with tools/ab_build_flag.sh; 0 = none, the product): 1 = priority 1 for the
// two MFMA phases (S^T, PV) of a query tile, 2 = priority 1 for the softmax phase between them.  Two workgroups share a CU, so a SIMD holds two
// waves of different (image, head) items in unrelated phases.
#ifndef KEMR_ATTN_PRIO
#define KEMR_ATTN_PRIO 0
#endif
__device__ __forceinline__ void attn_prio(int mfma_phase) {
    if constexpr (KEMR_ATTN_PRIO == 1) { if (mfma_phase) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    if constexpr (KEMR_ATTN_PRIO == 2) { if (mfma_phase) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
}

template <bool NT>
__device__ __forceinline__ bf16x8 load_q8(const bf16_t* p) {
    if constexpr (NT) return __builtin_nontemporal_load((const bf16x8*)p);
    else return *(const bf16x8*)p;
}

template <int NT32, bool CAUSAL, int TC, int NW = 4, bool STAMP = false, int NTLOAD = 1, bool SKIPTAIL = false>   // keys padded to NT32 * 32; NW waves per workgroup; NTLOAD: 0 plain loads, 1 = K / V non-temporal, 2 = Q too
__global__ __launch_bounds__(NW * 64, 2) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           int T_rt, int width, int xbatch, const int* __restrict__ row_start) {
    constexpr int TP = NT32 * 32;
    constexpr int NT16 = NT32 * 2;
    constexpr int NTH = NW * 64;
    constexpr int NCH = (TP * 8 + NTH - 1) / NTH;      // 16-byte chunks of K (and of V) per thread
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + TP * 128;

    int h, b;
    if (!attn_item(xbatch, h, b)) return;
    // row_start (run-time T only): items of different lengths packed one behind the other -- item b owns the rows row_start[b] ..
    // row_start[b + 1] - 1 of qkv and of out (the text tower computes a caption only up to its end-of-text token)
    int T_ = TC > 0 ? TC : T_rt;
    size_t row0 = (size_t)b * T_;
    if (TC == 0 && row_start) {
        const int r0 = row_start[b];
        T_ = row_start[b + 1] - r0;
        row0 = r0;
    }
    const int T = T_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = 3 * width;
    const bf16_t* base = qkv + row0 * ld + h * 64;
    const int lrow = lane & 15, lq = lane >> 4;
    const int nqt = SKIPTAIL ? T >> 4 : (T + 15) >> 4;      // SKIPTAIL (A/B timing only, WRONG results): the ragged last query tile is not computed

    unsigned long long st_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long now;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            st_t[k] += now - st_prev;
            st_prev = now;
        }
    };
    if constexpr (STAMP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
    // first query tile of this wave: issue its loads before the K/V staging so that their latency overlaps it
    bf16x8 qn[2];
    {
        const int q0 = wid * 16 + lrow;
        const int qc = q0 < T ? q0 : T - 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qn[kk] = load_q8<(NTLOAD >= 2)>(base + (size_t)qc * ld + kk * 32 + lq * 8);
    }
    // stage K and V: ALL global loads of both matrices in flight together (one latency, not two), then the swizzled
    // LDS writes.  (PMC: with K-then-V staging the waves sat 56 % of their life in s_waitcnt.)
    {
        uint4 kv[NCH], vv[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * NTH;
            const int row = idx >> 3, c = idx & 7;
            // branch-free: pad rows load the last valid row and are zeroed by a select at the LDS write (a conditional
            // load, or a select right here, makes hipcc wait for the loads in the middle of the batch)
            const int rc = row < T ? row : T - 1;
            if constexpr (NTLOAD >= 1) {
                typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                const u32x4_ a_ = __builtin_nontemporal_load((const u32x4_*)(base + (size_t)rc * ld + width + c * 8));
                const u32x4_ b_ = __builtin_nontemporal_load((const u32x4_*)(base + (size_t)rc * ld + 2 * width + c * 8));
                kv[i] = make_uint4(a_.x, a_.y, a_.z, a_.w);
                vv[i] = make_uint4(b_.x, b_.y, b_.z, b_.w);
            } else {
                kv[i] = *(const uint4*)(base + (size_t)rc * ld + width + c * 8);
                vv[i] = *(const uint4*)(base + (size_t)rc * ld + 2 * width + c * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * NTH;
            const int row = idx >> 3, c = idx & 7;
            if (idx < TP * 8) {
                const unsigned keep = row < T ? 0xffffffffu : 0u;       // component-wise: a struct select went to scratch
                uint4 a = kv[i], b2 = vv[i];
                a.x &= keep; a.y &= keep; a.z &= keep; a.w &= keep;
                b2.x &= keep; b2.y &= keep; b2.z &= keep; b2.w &= keep;
                *(uint4*)(sK + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = a;
                *(uint4*)(sV + row * 128 + ((c ^ (((row >> 1) & 3) << 1)) << 4)) = b2;
            }
        }
    }
    __syncthreads();
    stamp(0);                                           // staging (loads, LDS writes, barrier)

    for (int qt = wid; qt < nqt; qt += NW) {           // wave-uniform trip count: EXEC stays full for the tr reads
        const int q = qt * 16 + lrow;
        bf16x8 qf[2] = {qn[0], qn[1]};
        if (qt + NW < nqt) {                           // prefetch the next query tile of this wave
            const int q2 = q + NW * 16;
            const int qc = q2 < T ? q2 : T - 1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) qn[kk] = load_q8<(NTLOAD >= 2)>(base + (size_t)qc * ld + kk * 32 + lq * 8);
        }

        // S^T tiles in groups of G key tiles, K fragments double-buffered in registers: the reads of group g+1 are in
        // flight while the MFMAs of group g issue (hipcc otherwise emits read-wait-MFMA per tile on ONE register set
        // and exposes the LDS latency 18 times per query tile)
        constexpr int G = (NT16 % 3 == 0) ? 3 : 2;
        constexpr int NG = NT16 / G;
        f32x4 s[NT16];
        bf16x8 kfr[2][G][2];
        auto load_group = [&](int g, bf16x8 (&dst)[G][2]) {
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    dst[j][kk] = *(const bf16x8*)(sK + ((g * G + j) * 16 + lrow) * 128 + (((kk * 4 + lq) ^ (lrow >> 1)) << 4));
        };
        load_group(0, kfr[0]);
        attn_prio(1);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG && (g + 1) * G * 16 < T) load_group(g + 1, kfr[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int t = g * G + j;
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (t * 16 < T && (!CAUSAL || t <= qt)) {      // tiles made only of pad keys -- or, causal, of keys behind the query tile -- are skipped (uniform); the mask below covers them
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[g & 1][j][0], qf[0], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[g & 1][j][1], qf[1], s[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        attn_prio(0);
        stamp(1);                                       // S^T MFMAs + K reads (+ the wait for this tile's Q)
        // s[t][r] = S[query lrow][key t*16 + lq*4 + r]
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            const bool partial = (t + 1) * 16 > T || CAUSAL;   // only the boundary tiles (or causal) need a mask
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (partial) {
                    const int key = t * 16 + lq * 4 + r;
                    const bool ok = key < T && (!CAUSAL || key <= q);
                    s[t][r] = ok ? s[t][r] : -INFINITY;
                }
                mx = fmaxf(mx, s[t][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mxl = mx * LOG2E;
        stamp(2);                                       // mask + row max + exchange
        // exp(s - max) = exp2(s * log2e - max * log2e); masked keys -> 0.  The multiply-add and the row sum run two elements per
        // instruction (v_pk_fma_f32 / v_pk_add_f32): the softmax is issue-bound, the MFMAs hide behind it
        f32x2_t sum2 = {0.f, 0.f};
        const f32x2_t l2 = {LOG2E, LOG2E}, nm = {-mxl, -mxl};
        auto exp_tile = [&](int t) {
            f32x2_t a = f32x2_t{s[t][0], s[t][1]} * l2 + nm, c = f32x2_t{s[t][2], s[t][3]} * l2 + nm;
            a.x = __builtin_amdgcn_exp2f(a.x); a.y = __builtin_amdgcn_exp2f(a.y);
            c.x = __builtin_amdgcn_exp2f(c.x); c.y = __builtin_amdgcn_exp2f(c.y);
            s[t][0] = a.x; s[t][1] = a.y; s[t][2] = c.x; s[t][3] = c.y;
            sum2 += a;
            sum2 += c;
        };
#pragma unroll
        for (int t = 0; t < NT16; ++t) exp_tile(t);

        stamp(3);                                       // exponentials + row sum
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // O^T += V^T . P^T per 32-key block, V fragments double-buffered the same way
        bf16x8 vfr[2][4];
        auto load_v = [&](int u, bf16x8 (&dst)[4]) {
            const int ra = u * 32 + lq * 4 + (lrow >> 2);       // rows 32u + 4lq .. +3 (first half of the k slots)
            const int rb = ra + 16;                             // rows 32u + 16 + 4lq .. +3
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int c = dt * 2 + ((lrow & 3) >> 1);
                const bf16x4 va = lds_read_tr16(sV + ra * 128 + ((c ^ (((ra >> 1) & 3) << 1)) << 4) + (lrow & 1) * 8);
                const bf16x4 vb = lds_read_tr16(sV + rb * 128 + ((c ^ (((rb >> 1) & 3) << 1)) << 4) + (lrow & 1) * 8);
                dst[dt][0] = va[0]; dst[dt][1] = va[1]; dst[dt][2] = va[2]; dst[dt][3] = va[3];
                dst[dt][4] = vb[0]; dst[dt][5] = vb[1]; dst[dt][6] = vb[2]; dst[dt][7] = vb[3];
            }
        };
        load_v(0, vfr[0]);
        attn_prio(1);
        // a block that is all pad keys, or (causal) all behind the diagonal, has P = 0 and is skipped (uniform); the live blocks are a prefix
        auto live = [&](int u) { return u < NT32 && u * 32 < T && !(CAUSAL && u * 32 > qt * 16 + 15); };
#pragma unroll
        for (int u = 0; u < NT32; ++u) {
            if (!live(u)) continue;
            if (u + 1 < NT32 && (u + 1) * 32 < T) load_v(u + 1, vfr[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            union { bf16x8 v; uint32_t w[4]; } pf;
            pf.w[0] = pack_bf16x2(s[2 * u][0], s[2 * u][1]);
            pf.w[1] = pack_bf16x2(s[2 * u][2], s[2 * u][3]);
            pf.w[2] = pack_bf16x2(s[2 * u + 1][0], s[2 * u + 1][1]);
            pf.w[3] = pack_bf16x2(s[2 * u + 1][2], s[2 * u + 1][3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[u & 1][dt], pf.v, o[dt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        attn_prio(0);
        stamp(4);                                       // PV MFMAs + V reads + P packing
        float sum = sum2.x + sum2.y;
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        // o[dt][r] = O[query lrow][d = dt*16 + lq*4 + r]
        if (q < T) {
            const float inv = 1.0f / sum;
            bf16_t* dst = out + (row0 + q) * width + h * 64 + lq * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 pk;
                pk.x = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
                pk.y = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
                *(uint2*)(dst + dt * 16) = pk;
            }
        }
        stamp(5);                                       // normalise + store
    }
    if constexpr (STAMP) {                              // behind the output: 8 counters of wave-cycles (tools/prof_attention.py stamps)
        unsigned long long* dst = (unsigned long long*)(out + (size_t)xbatch * T * width) + (((size_t)b * (width >> 6) + h) * NW + wid) * 8;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dst[k] += st_t[k];
            dst[6] += 1;
        }
    }
}

// The attention experiments of rounds 2-4 (32-query tiles on the 32x32x16 MFMA, eight waves, persistent LDS-DMA staging, two tiles
// sharing fragments, persistent workgroups with the next item's K / V prefetched into registers, six waves, the s_memtime-stamped
// build) live in attention_ab.hip / behind KEMR_AB_VARIANTS: they are built only
// by `build.py --ab-variants`; the product library holds ONE attention kernel per shape and kemr_debug_set refuses the others.
int g_attn_v = 0;          // A/B builds only (attention_ab.hip), T = 257: 0 = the product kernel, 1..5 = the experiments

int g_attn_xcd = 1;        // 1 = the images dealt to the XCDs (attn_item above; default), 0 = grid order (tools)
int g_attn_waves = 0;      // tools: 0 = the default choice below, else waves per workgroup for the 257-token shape (4 or 6)

#ifdef KEMR_AB_VARIANTS
int launch_attention_ab(int v, int waves, const bf16_t* qkv, bf16_t* out, int batch, int width, int xbatch, hipStream_t stream);
#endif

template <int NT32>
static int launch_nt(const bf16_t* qkv, bf16_t* out, int batch, int t, int width, int causal, hipStream_t stream, const int* row_start = nullptr) {
    constexpr int smem = NT32 * 32 * 128 * 2;
    const dim3 grid(width / 64, batch);
    ProfScope prof(PROF_ATTENTION, stream);
    void (*kern)(const bf16_t*, bf16_t*, int, int, int, const int*);
    const int xbatch = g_attn_xcd ? batch : 0;
    const dim3 xgrid(width / 64, xbatch ? (batch + 7) / 8 * 8 : batch);
    int threads = 256;
    if (causal) {
        if (NT32 == 3 && t == 77 && !row_start) { kern = attention_kernel<NT32, true, NT32 == 3 ? 77 : 0, NT32 == 3 ? 5 : 4>; threads = 320; }
        else kern = attention_kernel<NT32, true, 0>;
    } else if (NT32 == 9 && t == 257) {
#ifdef KEMR_AB_VARIANTS
        if (g_attn_v != 0) return launch_attention_ab(g_attn_v, g_attn_waves, qkv, out, batch, width, xbatch, stream);
        if (g_attn_waves == 6) { kern = attention_kernel<NT32, false, NT32 == 9 ? 257 : 0, NT32 == 9 ? 6 : 4>; threads = 384; }
        else if (g_attn_waves == 7) kern = attention_kernel<NT32, false, NT32 == 9 ? 257 : 0, 4, false, NT32 == 9 ? 0 : 1>;      // A/B: the plain (temporal) loads of rounds 1-3
        else if (g_attn_waves == 8) kern = attention_kernel<NT32, false, NT32 == 9 ? 257 : 0, 4, false, 1, NT32 == 9>;          // A/B timing only: without the 17th (one-row) query tile
        else if (g_attn_waves == 5) kern = attention_kernel<NT32, false, NT32 == 9 ? 257 : 0, 4, false, NT32 == 9 ? 2 : 1>;      // A/B: Q rows non-temporal as well
        else if (g_attn_waves == 2 && xbatch) kern = attention_kernel<NT32, false, NT32 == 9 ? 257 : 0, 4, NT32 == 9>;   // stamps: the caller's
                                                                                                                         // `out` has room behind it
        else
#else
        if (g_attn_v != 0 || g_attn_waves != 0)
            KEMR_FAIL(KEMR_ERR_INVALID, "attention: attn_v = %d / attn_waves = %d select A/B kernels this library was built without (build.py --ab-variants)", g_attn_v, g_attn_waves);
#endif
        kern = attention_kernel<NT32, false, NT32 == 9 ? 257 : 0>;
    } else {
        kern = attention_kernel<NT32, false, 0>;
    }
    KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipLaunchKernelGGL(kern, xgrid, dim3(threads), smem, stream, qkv, out, t, width, xbatch, row_start);
    KEMR_CHECK_LAUNCH("attention_kernel");
    return KEMR_OK;
}

int launch_attention(const bf16_t* qkv, bf16_t* out, int batch, int t, int width, int causal, hipStream_t stream) {
    if (batch <= 0) return KEMR_OK;
    if (width % 64 != 0 || t <= 0) KEMR_FAIL(KEMR_ERR_INVALID, "attention: bad shape t=%d width=%d", t, width);
    if (batch > 65528) KEMR_FAIL(KEMR_ERR_INVALID, "attention: batch %d > 65528 (grid.y, rounded up to a multiple of 8)", batch);
    const int nt32 = (t + 31) / 32;
    switch (nt32) {
        case 1: return launch_nt<1>(qkv, out, batch, t, width, causal, stream);
        case 2: return launch_nt<2>(qkv, out, batch, t, width, causal, stream);
        case 3: return launch_nt<3>(qkv, out, batch, t, width, causal, stream);
        case 4: return launch_nt<4>(qkv, out, batch, t, width, causal, stream);
        case 5: return launch_nt<5>(qkv, out, batch, t, width, causal, stream);
        case 6: return launch_nt<6>(qkv, out, batch, t, width, causal, stream);
        case 7: return launch_nt<7>(qkv, out, batch, t, width, causal, stream);
        case 8: return launch_nt<8>(qkv, out, batch, t, width, causal, stream);
        case 9: return launch_nt<9>(qkv, out, batch, t, width, causal, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "attention: sequence length %d > 288 not supported", t);
}

// ---- the attention of the pooled row alone (last block of a tower) ----------------------------------------------------------------
// Only one row per item leaves a tower (the class token / the end-of-text token), so in the LAST block only that row's query is
// needed: one wave per (item, head) -- scores of the one query against the item's keys (eight lanes per key row, eight keys per
// pass; bf16 products summed in fp32 like the MFMA does), softmax in fp32 with P rounded to bf16 and the row sum taken before the rounding (as in the
// tile kernels), then out[d] = sum_j p_j v_j[d] with lane = (pair of d, key parity).  q: [items, width] compact rows (already scaled:
// the 1/sqrt(64) lives in the packed q weights); k, v: the qkv buffer of the whole call (ld 3 * width); out: [items, width] compact.
// Item b's keys are the rows key0[b] .. key0[b] + nkeys[b] - 1, taken from pool_idx / row_start:
//   vision (causal = 0): all `tokens` rows from b * tokens;  text: the rows from the text's first up to the pooled one.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void attention_pooled_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ qkv,
                                                               bf16_t* __restrict__ out, const int* __restrict__ pool_idx,
                                                               const int* __restrict__ row_start, int items, int tokens, int width,
                                                               int causal) {
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int MAXK = 320;
    __shared__ float sp[4][MAXK];
    const int heads = width >> 6, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wv;
    if (item >= items * heads) return;                  // (no barrier below: a wave works on its own)
    const int b = item / heads, h = item - b * heads;
    const int r0 = row_start ? row_start[b] : b * tokens;
    int nk = causal ? pool_idx[b] - r0 + 1 : tokens;
    nk = nk < 1 ? 1 : (nk > MAXK ? MAXK : nk);
    const size_t ld = 3 * (size_t)width;
    // scores: eight lanes share a key row (lane & 7 = its 16-byte chunk: one 128-byte row per load instruction and key), eight keys
    // per pass; the lane's eight products are summed in fp32, then the eight lanes of the key by shuffles
    const int ck = lane & 7, kq = lane >> 3;
    float qf[8];
    {
        const uint4 v = ((const uint4*)(q + (size_t)b * width + h * 64))[ck];
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            qf[2 * k] = bf16_to_f32((bf16_t)(w4[k] & 0xffff));
            qf[2 * k + 1] = bf16_to_f32((bf16_t)(w4[k] >> 16));
        }
    }
    const bf16_t* kbase = qkv + (size_t)r0 * ld + width + h * 64 + ck * 8;
    const int npass = (nk + 7) >> 3;
    for (int pss = 0; pss < npass; ++pss) {
        const int j = pss * 8 + kq;
        const int jc = j < nk ? j : nk - 1;
        const uint4 v = *(const uint4*)(kbase + (size_t)jc * ld);
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = fmaf(qf[2 * k], bf16_to_f32((bf16_t)(w4[k] & 0xffff)), s);
            s = fmaf(qf[2 * k + 1], bf16_to_f32((bf16_t)(w4[k] >> 16)), s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (ck == 0 && j < nk) sp[wv][j] = s;
    }
    // softmax over the wave's LDS row, written and read by this wave only: the hardware keeps a wave's LDS accesses in order, the
    // fence + wave barrier keep the COMPILER from moving a lane's loads above another lane's stores it cannot prove to alias
    wave_lds_fence();
    float sc[MAXK / 64];
    float mx = -INFINITY;
#pragma unroll
    for (int pss = 0; pss < MAXK / 64; ++pss) {
        const int j = pss * 64 + lane;
        sc[pss] = j < nk ? sp[wv][j] : -INFINITY;
        mx = fmaxf(mx, sc[pss]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int pss = 0; pss < MAXK / 64; ++pss) {
        const int j = pss * 64 + lane;
        const float p = __builtin_amdgcn_exp2f((sc[pss] - mx) * LOG2E);         // -inf -> 0 beyond the keys
        sum += p;
        if (j < MAXK) sp[wv][j] = bf16_to_f32(f32_to_bf16(p));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    wave_lds_fence();                                    // the bf16-rounded P row: stores of all lanes before any lane's loads
    const int dp = lane & 31, par = lane >> 5;
    const bf16_t* vp = qkv + (size_t)r0 * ld + 2 * width + h * 64 + dp * 2;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
    for (int j = par; j < nk; j += 2) {
        const uint32_t v = *(const uint32_t*)(vp + (size_t)j * ld);
        const float p = sp[wv][j];
        a0 = fmaf(p, bf16_to_f32((bf16_t)(v & 0xffff)), a0);
        a1 = fmaf(p, bf16_to_f32((bf16_t)(v >> 16)), a1);
    }
    a0 += __shfl_xor(a0, 32);
    a1 += __shfl_xor(a1, 32);
    if (par == 0) {
        const float inv = 1.0f / sum;
        *(uint32_t*)(out + (size_t)b * width + h * 64 + dp * 2) = pack_bf16x2(a0 * inv, a1 * inv);
    }
}

int launch_attention_pooled(const bf16_t* q, const bf16_t* qkv, bf16_t* out, const int* pool_idx, const int* row_start, int items,
                            int tokens, int width, int causal, hipStream_t stream) {
    if (items <= 0) return KEMR_OK;
    if (width % 64 != 0 || tokens <= 0 || tokens > 320) KEMR_FAIL(KEMR_ERR_INVALID, "attention (pooled row): bad shape t=%d width=%d", tokens, width);
    if (causal && !pool_idx) KEMR_FAIL(KEMR_ERR_INVALID, "attention (pooled row): the causal form needs the pooled positions");
    ProfScope prof(PROF_ATTENTION, stream);
    const long waves = (long)items * (width / 64);
    hipLaunchKernelGGL(attention_pooled_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, q, qkv, out, pool_idx, row_start,
                       items, tokens, width, causal);
    KEMR_CHECK_LAUNCH("attention_pooled_kernel");
    return KEMR_OK;
}

// Causal attention over items of different lengths packed one behind the other: item b = rows row_start[b] .. row_start[b + 1] - 1
// (device array of batch + 1 ints), every length in 1 .. max_t.
int launch_attention_packed(const bf16_t* qkv, bf16_t* out, const int* row_start, int batch, int max_t, int width, hipStream_t stream) {
    if (batch <= 0) return KEMR_OK;
    if (!row_start) KEMR_FAIL(KEMR_ERR_INVALID, "attention: packed rows need row_start");
    if (width % 64 != 0 || max_t <= 0) KEMR_FAIL(KEMR_ERR_INVALID, "attention: bad shape t=%d width=%d", max_t, width);
    if (batch > 65528) KEMR_FAIL(KEMR_ERR_INVALID, "attention: batch %d > 65528 (grid.y, rounded up to a multiple of 8)", batch);
    switch ((max_t + 31) / 32) {
        case 1: return launch_nt<1>(qkv, out, batch, max_t, width, 1, stream, row_start);
        case 2: return launch_nt<2>(qkv, out, batch, max_t, width, 1, stream, row_start);
        case 3: return launch_nt<3>(qkv, out, batch, max_t, width, 1, stream, row_start);
        case 4: return launch_nt<4>(qkv, out, batch, max_t, width, 1, stream, row_start);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "attention: packed rows support lengths up to 128, got %d", max_t);
}

}  // namespace kemr
