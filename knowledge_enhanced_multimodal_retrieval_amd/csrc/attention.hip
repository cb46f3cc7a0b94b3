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

// ---- T = 257 (ViT-L/14 and ViT-B/16 vision towers): 32-query tiles on v_mfma_f32_32x32x16_bf16 (round 3) ---------------------------
// Round 2's PMC pass of the 16-query kernel above put a wave at one third issuing, one third parked at s_waitcnt and one third
// issue-stalled, MFMA pipe 20 % busy: per 16 queries it issues 72 MFMAs, 36 + 72 LDS reads and ~1 100 cycles of softmax VALU, with
// every phase waiting on the one before.  A 32 x 32 tile halves the MFMA and LDS-read instructions per query (one K / V fragment
// serves 32 queries), an MFMA holds the issue port for 8 of its 32 cycles instead of 8 of 16, and 257 queries are 8 full tiles --
// two per wave, balanced -- plus ONE query (the 16-query kernel: 17 tiles as 5 / 4 / 4 / 4).
//   S^T block (32 keys x 32 queries) = K_blk . Q^T: A = K rows (ds_read_b128: lane (key l & 31, half l >> 5) holds d = 16 ks + 8 half
//   .. + 7), B = Q rows straight from HBM in the same shape; a lane ends with 16 keys of ONE query per block (C layout: key =
//   (reg & 3) + 8 (reg >> 2) + 4 half), so row max / sum are the in-lane reduction and one exchange with lane ^ 32.
//   O^T (64 x 32) += V^T . P^T per 16-key step: the score registers 8 st .. 8 st + 7, packed to bf16, ARE the B fragment (k slot
//   8 half + j <-> key 16 st + 8 (j >> 2) + 4 half + (j & 3)); the A fragment takes the same keys from the row-major V image with two
//   ds_read_b64_tr_b16 (rows 16 st + 4 half + 0..3 and + 8).
// LDS images: K as above (chunk ^= (row >> 1) & 7: conflict-free for both MFMA shapes); V rows of 128 B with the 32-byte chunk index
// ^= ((row >> 1) & 1) << 1 | ((row >> 2) & 1), which makes the half-wave patterns of BOTH shapes conflict-free (this one reads 4 rows
// x 64 B per half wave, the 16 x 16 one 8 rows x 32 B).
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int v_swz32(int row) { return (((row >> 1) & 1) << 1) | ((row >> 2) & 1); }

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attention32_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int width) {
    constexpr int T = 257, NKB = 9, TP = NKB * 32, NTH = NW * 64, NCH = (TP * 8 + NTH - 1) / NTH, NQT = (T + 31) / 32;
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + TP * 128;
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = 3 * width;
    const bf16_t* base = qkv + (size_t)b * T * ld + h * 64;
    const int r32 = lane & 31, hh = lane >> 5;

    // first query tile of this wave: its loads go out before the K / V staging so that their latency overlaps it
    bf16x8 qn[4];
    {
        const int q0 = wid * 32 + r32;
        const int qc = q0 < T ? q0 : T - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qn[ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 16 + hh * 8);
    }
    {
        uint4 kv[NCH], vv[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * NTH;
            const int row = idx >> 3, c = idx & 7;
            const int rc = row < T ? row : T - 1;
            kv[i] = *(const uint4*)(base + (size_t)rc * ld + width + c * 8);
            vv[i] = *(const uint4*)(base + (size_t)rc * ld + 2 * width + c * 8);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * NTH;
            const int row = idx >> 3, c = idx & 7;
            if (idx < TP * 8) {
                const unsigned keep = row < T ? 0xffffffffu : 0u;
                uint4 a = kv[i], b2 = vv[i];
                a.x &= keep; a.y &= keep; a.z &= keep; a.w &= keep;
                b2.x &= keep; b2.y &= keep; b2.z &= keep; b2.w &= keep;
                *(uint4*)(sK + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = a;
                *(uint4*)(sV + row * 128 + ((((c >> 1) ^ v_swz32(row)) << 5) | ((c & 1) << 4))) = b2;
            }
        }
    }
    __syncthreads();

    // per-lane LDS offsets that do not depend on the block: K row r32, V rows 4 hh + (i >> 2) and columns of the lane's 16-lane group
    const int kswz = (r32 >> 1) & 7;
    const int vi = lane & 15, vg = (lane >> 4) & 1;
    for (int qt = wid; qt < NQT; qt += NW) {           // wave-uniform trip count: EXEC stays full for the tr reads
        const int q = qt * 32 + r32;
        bf16x8 qf[4] = {qn[0], qn[1], qn[2], qn[3]};
        if (qt + NW < NQT) {
            const int q2 = q + NW * 32;
            const int qc = q2 < T ? q2 : T - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qn[ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 16 + hh * 8);
        }
        // ---- S^T: nine blocks of 32 keys, K fragments of the next block in flight while this block's four MFMAs issue
        f32x16 s[NKB];
        bf16x8 kfr[2][4];
        auto load_k = [&](int kb, bf16x8 (&dst)[4]) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                dst[ks] = *(const bf16x8*)(sK + (kb * 32 + r32) * 128 + (((ks * 2 + hh) ^ kswz) << 4));
        };
        load_k(0, kfr[0]);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb + 1 < NKB) load_k(kb + 1, kfr[(kb + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb & 1][ks], qf[ks], s[kb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // s[kb][r] = S[query r32][key kb*32 + (r & 3) + 8 (r >> 2) + 4 hh]; of the last block only key 256 exists
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kb == NKB - 1) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    s[kb][r] = key < T ? s[kb][r] : -INFINITY;
                }
                mx = fmaxf(mx, s[kb][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mxl = mx * LOG2E;
        f32x2_t sum2 = {0.f, 0.f};
        const f32x2_t l2 = {LOG2E, LOG2E}, nm = {-mxl, -mxl};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2_t a = f32x2_t{s[kb][r], s[kb][r + 1]} * l2 + nm;
                a.x = __builtin_amdgcn_exp2f(a.x);
                a.y = __builtin_amdgcn_exp2f(a.y);
                s[kb][r] = a.x;
                s[kb][r + 1] = a.y;
                sum2 += a;
            }
        float sum = sum2.x + sum2.y;
        sum += __shfl_xor(sum, 32);

        // ---- O^T += V^T . P^T: per block two 16-key steps x two 32-row halves of d
        f32x16 o[2];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dh][r] = 0.f;
        auto load_v = [&](int kb, int st, bf16x8 (&dst)[2]) {
            const int ra = kb * 32 + 16 * st + 4 * hh + (vi >> 2), rb = ra + 8;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                const int c32 = dh * 2 + vg;
                const bf16x4 va = lds_read_tr16(sV + ra * 128 + ((c32 ^ v_swz32(ra)) << 5) + (vi & 3) * 8);
                const bf16x4 vb = lds_read_tr16(sV + rb * 128 + ((c32 ^ v_swz32(rb)) << 5) + (vi & 3) * 8);
                dst[dh][0] = va[0]; dst[dh][1] = va[1]; dst[dh][2] = va[2]; dst[dh][3] = va[3];
                dst[dh][4] = vb[0]; dst[dh][5] = vb[1]; dst[dh][6] = vb[2]; dst[dh][7] = vb[3];
            }
        };
        constexpr int NST = 2 * NKB - 1;                // the last block's second step holds pad keys only
        bf16x8 vfr[2][2];
        load_v(0, 0, vfr[0]);
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int kb = u >> 1, st = u & 1;
            if (u + 1 < NST) load_v((u + 1) >> 1, (u + 1) & 1, vfr[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            union { bf16x8 v; uint32_t w[4]; } pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) pf.w[j] = pack_bf16x2(s[kb][8 * st + 2 * j], s[kb][8 * st + 2 * j + 1]);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
                o[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[u & 1][dh], pf.v, o[dh], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // o[dh][r] = O[query r32][d = 32 dh + (r & 3) + 8 (r >> 2) + 4 hh]
        if (q < T) {
            const float inv = 1.0f / sum;
            bf16_t* dst = out + ((size_t)b * T + q) * width + h * 64 + 4 * hh;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    uint2 pk;
                    pk.x = pack_bf16x2(o[dh][4 * rg] * inv, o[dh][4 * rg + 1] * inv);
                    pk.y = pack_bf16x2(o[dh][4 * rg + 2] * inv, o[dh][4 * rg + 3] * inv);
                    *(uint2*)(dst + 32 * dh + 8 * rg) = pk;
                }
        }
    }
}

// Measured (round 3, B = 255 x 16 heads, same device; profiles/r03_v2_attention_pmc.txt): 172 us against the 16-query kernel's 165.
// Per launch it issues 11 % fewer VALU and 47 % fewer LDS instructions (23.9 M / 4.2 M against 26.8 M / 7.9 M) at the same MFMA
// work, yet its waves spend 43 % of their cycles issue-stalled (SQ_WAIT_INST_ANY) where the 16-query kernel's spend 32 %: with 1.4
// waves per SIMD on average neither hides the MFMA -> VALU -> MFMA dependence of a tile (S^T, softmax, PV), and the longer 32 x 32
// chains expose more of it.  What is missing is a second tile in flight per wave, not fewer instructions.  Kept behind the
// debug switch attn_v = 1 with its tests; the default stays the 16-query kernel.  (Also tried on the 16-query kernel and dropped: the
// fences between its phases removed, mask + row maximum of key group g - 1 behind the MFMAs of group g, each 32-key block's
// exponentials in the PV loop in front of the MFMAs that consume them -- hipcc hoists the exponentials, the kernel reaches 256
// VGPRs with 28 bytes of scratch and takes 199-208 us against 178-184 on the same device.  A second tile in flight needs the
// schedule written by hand, as the GEMM's is.)
int g_attn_v = 0;          // tools: 0 = the 16-query-tile kernel (default), 1 = 32-query tiles on the 32x32x16 MFMA at T = 257

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
    } else if (NT32 == 9 && t == 257 && g_attn_v == 1) {
        auto k32 = attention32_kernel<4>;
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        hipLaunchKernelGGL(k32, grid, dim3(256), smem, stream, qkv, out, width);
        KEMR_CHECK_LAUNCH("attention32_kernel");
        return KEMR_OK;
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
