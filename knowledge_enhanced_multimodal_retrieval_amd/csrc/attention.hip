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
template <int NT32, bool CAUSAL, int TC, int NW = 4>   // keys padded to NT32 * 32; NW waves per workgroup
__global__ __launch_bounds__(NW * 64, 2) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           int T_rt, int width) {
    const int T = TC > 0 ? TC : T_rt;
    constexpr int TP = NT32 * 32;
    constexpr int NT16 = NT32 * 2;
    constexpr int NTH = NW * 64;
    constexpr int NCH = (TP * 8 + NTH - 1) / NTH;      // 16-byte chunks of K (and of V) per thread
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + TP * 128;

    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = 3 * width;
    const bf16_t* base = qkv + (size_t)b * T * ld + h * 64;
    const int lrow = lane & 15, lq = lane >> 4;
    const int nqt = (T + 15) >> 4;

    // first query tile of this wave: issue its loads before the K/V staging so that their latency overlaps it
    bf16x8 qn[2];
    {
        const int q0 = wid * 16 + lrow;
        const int qc = q0 < T ? q0 : T - 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qn[kk] = *(const bf16x8*)(base + (size_t)qc * ld + kk * 32 + lq * 8);
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
            kv[i] = *(const uint4*)(base + (size_t)rc * ld + width + c * 8);
            vv[i] = *(const uint4*)(base + (size_t)rc * ld + 2 * width + c * 8);

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

    for (int qt = wid; qt < nqt; qt += NW) {           // wave-uniform trip count: EXEC stays full for the tr reads
        const int q = qt * 16 + lrow;
        bf16x8 qf[2] = {qn[0], qn[1]};
        if (qt + NW < nqt) {                           // prefetch the next query tile of this wave
            const int q2 = q + NW * 16;
            const int qc = q2 < T ? q2 : T - 1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) qn[kk] = *(const bf16x8*)(base + (size_t)qc * ld + kk * 32 + lq * 8);
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
        // exp(s - max) = exp2(s * log2e - max * log2e); masked keys -> 0.  The multiply-add and the row sum run two elements per
        // instruction (v_pk_fma_f32 / v_pk_add_f32): the softmax is issue-bound, the MFMAs hide behind it
        f32x2_t sum2 = {0.f, 0.f};
        const f32x2_t l2 = {LOG2E, LOG2E}, nm = {-mxl, -mxl};
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            f32x2_t a = f32x2_t{s[t][0], s[t][1]} * l2 + nm, c = f32x2_t{s[t][2], s[t][3]} * l2 + nm;
            a.x = __builtin_amdgcn_exp2f(a.x); a.y = __builtin_amdgcn_exp2f(a.y);
            c.x = __builtin_amdgcn_exp2f(c.x); c.y = __builtin_amdgcn_exp2f(c.y);
            s[t][0] = a.x; s[t][1] = a.y; s[t][2] = c.x; s[t][3] = c.y;
            sum2 += a;
            sum2 += c;
        }
        float sum = sum2.x + sum2.y;
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);

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
#pragma unroll
        for (int u = 0; u < NT32; ++u) {
            if (u * 32 >= T || (CAUSAL && u * 32 > qt * 16 + 15)) continue;      // all-pad key block, or all behind the diagonal: P = 0 (uniform)
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
        // o[dt][r] = O[query lrow][d = dt*16 + lq*4 + r]
        if (q < T) {
            const float inv = 1.0f / sum;
            bf16_t* dst = out + ((size_t)b * T + q) * width + h * 64 + lq * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 pk;
                pk.x = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
                pk.y = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
                *(uint2*)(dst + dt * 16) = pk;
            }
        }
    }
}

int g_attn_waves = 0;      // tools: 0 = the default choice below, else waves per workgroup for the 257-token shape (4 or 6)

template <int NT32>
static int launch_nt(const bf16_t* qkv, bf16_t* out, int batch, int t, int width, int causal, hipStream_t stream) {
    constexpr int smem = NT32 * 32 * 128 * 2;
    const dim3 grid(width / 64, batch);
    ProfScope prof(PROF_ATTENTION, stream);
    void (*kern)(const bf16_t*, bf16_t*, int, int);
    int threads = 256;
    if (causal) {
        if (NT32 == 3 && t == 77) { kern = attention_kernel<NT32, true, NT32 == 3 ? 77 : 0, NT32 == 3 ? 5 : 4>; threads = 320; }
        else kern = attention_kernel<NT32, true, 0>;
    } else if (NT32 == 9 && t == 257) {
        if (g_attn_waves == 6) { kern = attention_kernel<NT32, false, NT32 == 9 ? 257 : 0, NT32 == 9 ? 6 : 4>; threads = 384; }
        else kern = attention_kernel<NT32, false, NT32 == 9 ? 257 : 0>;
    } else {
        kern = attention_kernel<NT32, false, 0>;
    }
    KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipLaunchKernelGGL(kern, grid, dim3(threads), smem, stream, qkv, out, t, width);
    KEMR_CHECK_LAUNCH("attention_kernel");
    return KEMR_OK;
}

int launch_attention(const bf16_t* qkv, bf16_t* out, int batch, int t, int width, int causal, hipStream_t stream) {
    if (batch <= 0) return KEMR_OK;
    if (width % 64 != 0 || t <= 0) KEMR_FAIL(KEMR_ERR_INVALID, "attention: bad shape t=%d width=%d", t, width);
    if (batch > 65535) KEMR_FAIL(KEMR_ERR_INVALID, "attention: batch %d > 65535", batch);
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

}  // namespace kemr
