// bf16 GEMM, 256 x 256 x 64 tile, 8 waves (2 M x 4 N, 128 x 64 per wave), LDS-DMA staging with COUNTED vmcnt
// across raw s_barriers (cdna_hip_programming.md section 5: "256^2 8-phase template", T3+T4), gfx950 only.
//
// LDS (128 KiB, one array): two K-tile buffers, each [A0 | A1 | B0 | B1] half-tiles of 128 rows x 64 bf16 (16 KiB),
// rows of 128 B with the same 16-byte-chunk XOR swizzle as gemm.hip (source-side for the DMA, read-side for ds_read).
// A wave reads only A half `wr` and B half `wc >> 1`.
//
// Per K-tile t (buffer t & 1) four phases, one C quadrant (64 x 32, 16 MFMAs of 16x16x32) each:
//   P1  reads A(rows 0-63) + B(cols 0-31)   stages A0(t+1)   MFMA quadrant (0,0)
//   P2  reads            B(cols 32-63)      stages A1(t+1)   MFMA quadrant (0,1)
//   P3  reads A(rows 64-127)                stages B0(t+2)   MFMA quadrant (1,1)
//   P4  (B(cols 0-31) kept in registers)    stages B1(t+2)   MFMA quadrant (1,0)   then s_waitcnt vmcnt(4)
// one s_barrier ends every phase.  Slot reuse: the B halves of tile t are last read at the start of P2, so P3/P4
// refill them for tile t+2; the A halves of tile t-1 were last read in its P3, so P1/P2 refill them for tile t+1.
// The wait at the end of P4 leaves the two youngest half-tiles (B0/B1 of t+2, 4 LDS-DMA ops per lane) in flight
// and guarantees A(t+1) and B(t+1) have landed before the barrier that precedes their first ds_read.
#include "common.h"

namespace kemr {

namespace {

constexpr int TBUF = 65536;      // bytes per K-tile buffer
constexpr int THALF = 16384;     // bytes per half-tile

__device__ __forceinline__ void glds16b(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int MH, int NH>
__device__ __forceinline__ void quadrant(f32x4 (&acc)[8][4], const bf16x8 (&af)[4][2], const bf16x8 (&wf)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[MH * 4 + mi][NH * 2 + ni] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][kk], af[mi][kk], acc[MH * 4 + mi][NH * 2 + ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);     // keep the MFMA cluster in front of the phase's closing s_barrier
}

// Same quadrant with the MFMAs as volatile asm: hipcc treats the builtin as a pure function and sinks part of a
// cluster below the following s_barrier, which defeats the staggered schedule; volatile asm keeps program order.
// Operands come from ds_reads (the compiler still inserts the lgkmcnt wait in front of the first asm use); the
// accumulate chain needs no wait states; the epilogue pads the MFMA -> VALU read hazard itself (s_nop after the loop).
template <int MH, int NH>
__device__ __forceinline__ void quadrant_pinned(f32x4 (&acc)[8][4], const bf16x8 (&af)[4][2], const bf16x8 (&wf)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
    asm volatile("s_nop 1" ::: "memory");      // any compiler VALU write just above -> first asm MFMA operand read
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                             : "+v"(acc[MH * 4 + mi][NH * 2 + ni]) : "v"(wf[ni][kk]), "v"(af[mi][kk]));
    __builtin_amdgcn_s_setprio(0);
}

}  // namespace

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_bf16_nt_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    const int tiles_n = p.N >> 8;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int row0 = tile_m * 256, col0 = tile_n * 256;

    // staging: this wave fills pieces 2*wid and 2*wid+1 (8 rows x 128 B each) of every half-tile
    const int srow = lane >> 3, schunk = lane & 7;
    const int r0 = wid * 16 + srow, r1 = r0 + 8;
    const bf16_t* a_src0 = p.A + (size_t)(row0 + r0) * p.lda + ((schunk ^ ((r0 >> 1) & 7)) << 3);
    const bf16_t* a_src1 = p.A + (size_t)(row0 + r1) * p.lda + ((schunk ^ ((r1 >> 1) & 7)) << 3);
    const bf16_t* w_src0 = p.W + (size_t)(col0 + r0) * p.ldw + ((schunk ^ ((r0 >> 1) & 7)) << 3);
    const bf16_t* w_src1 = p.W + (size_t)(col0 + r1) * p.ldw + ((schunk ^ ((r1 >> 1) & 7)) << 3);
    const size_t a_half = (size_t)128 * p.lda, w_half = (size_t)128 * p.ldw;
    char* const stage_base = smem + wid * 2048;

    auto stage_a = [&](int half, int tau) {
        char* dst = stage_base + (tau & 1) * TBUF + half * THALF;
        glds16b(a_src0 + half * a_half + tau * 64, dst);
        glds16b(a_src1 + half * a_half + tau * 64, dst + 1024);
    };
    auto stage_w = [&](int half, int tau) {
        char* dst = stage_base + (tau & 1) * TBUF + (2 + half) * THALF;
        glds16b(w_src0 + half * w_half + tau * 64, dst);
        glds16b(w_src1 + half * w_half + tau * 64, dst + 1024);
    };

    const int lrow = lane & 15, lq = lane >> 4;
    const int swz = lrow >> 1;
    const int co0 = ((0 + lq) ^ swz) << 4, co1 = ((4 + lq) ^ swz) << 4;
    const int a_off = wr * THALF + lrow * 128;
    const int b_off = 2 * THALF + (wc >> 1) * THALF + ((wc & 1) * 64 + lrow) * 128;

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K >> 6;
    stage_a(0, 0); stage_a(1, 0); stage_w(0, 0); stage_w(1, 0);
    if (nt > 1) {
        stage_w(0, 1); stage_w(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    bf16x8 af[4][2], w0[2][2], w1[2][2];
    for (int t = 0; t < nt; ++t) {
        const char* sa = smem + (t & 1) * TBUF + a_off;
        const char* sb = smem + (t & 1) * TBUF + b_off;
        // ---- P1
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            w0[ni][0] = *(const bf16x8*)(sb + ni * 2048 + co0);
            w0[ni][1] = *(const bf16x8*)(sb + ni * 2048 + co1);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            af[mi][0] = *(const bf16x8*)(sa + mi * 2048 + co0);
            af[mi][1] = *(const bf16x8*)(sa + mi * 2048 + co1);
        }
        if (t + 1 < nt) stage_a(0, t + 1);
        quadrant<0, 0>(acc, af, w0);
        __builtin_amdgcn_s_barrier();
        // ---- P2
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            w1[ni][0] = *(const bf16x8*)(sb + 4096 + ni * 2048 + co0);
            w1[ni][1] = *(const bf16x8*)(sb + 4096 + ni * 2048 + co1);
        }
        if (t + 1 < nt) stage_a(1, t + 1);
        quadrant<0, 1>(acc, af, w1);
        __builtin_amdgcn_s_barrier();
        // ---- P3
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            af[mi][0] = *(const bf16x8*)(sa + 8192 + mi * 2048 + co0);
            af[mi][1] = *(const bf16x8*)(sa + 8192 + mi * 2048 + co1);
        }
        if (t + 2 < nt) stage_w(0, t + 2);
        quadrant<1, 1>(acc, af, w1);
        __builtin_amdgcn_s_barrier();
        // ---- P4
        if (t + 2 < nt) {
            stage_w(1, t + 2);
            quadrant<1, 0>(acc, af, w0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            quadrant<1, 0>(acc, af, w0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }

    // Epilogue. acc[mi][ni][r] = C[row0 + wr*128 + mi*16 + lrow][col0 + wc*64 + ni*16 + lq*4 + r]
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = row0 + wr * 128 + mi * 16 + lrow;
        if (m >= p.M || (p.dbg & 1)) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = col0 + wc * 64 + ni * 16 + lq * 4;
            f32x4 v = acc[mi][ni];
            if (p.bias) {
                const float4 b = *(const float4*)(p.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if constexpr (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
            }
            if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16) {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
            } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
                float4* dst = (float4*)((float*)p.C + (size_t)m * p.ldc + n);
                float4 x = *dst;
                x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
                *dst = x;
            } else {
                const int img = m / p.patches, pi = m - img * p.patches;
                const float4 pe = *(const float4*)(p.pos + (size_t)(pi + 1) * p.N + n);
                float4 x;
                x.x = v[0] + pe.x; x.y = v[1] + pe.y; x.z = v[2] + pe.z; x.w = v[3] + pe.w;
                *(float4*)((float*)p.C + (size_t)(m + img + 1) * p.ldc + n) = x;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Staggered variant: two s_barriers per phase (R: issue ds_reads + LDS-DMA | M: MFMAs) and the wr == 1 half of the
// workgroup runs ONE barrier interval behind the wr == 0 half, so the two waves that share a SIMD (w and w + 4)
// alternate: one issues LDS reads while the other owns the matrix pipe (MI355X_MICROARCH "Two waves per SIMD").
// Interval numbering inside tile t (group 0 / group 1): R1 0/1, M1 1/2, R2 2/3, M2 3/4, R3 4/5, M3 5/6, R4 6/7, M4 7/8.
// Staging per tile t, each wave in its own R phase:  R1 B1(t+1) | R2 A0(t+1) | R3 A1(t+1) | R4 B0(t+2).
//   slot of B1(t+1): tile t-1's B1, last read R2(t-1) (done by interval 4 of t-1)            -> written from 8
//   slot of A0(t+1): tile t-1's A0, read only by group 0 in R3(t-1) (done by interval 5)      -> written from 10
//   slot of A1(t+1): tile t-1's A1, read only by group 1 in R3(t-1) (done by interval 6)      -> written from 12
//   slot of B0(t+2): tile t's   B0, last read R2(t) (group 1's wait is at the start of 4)     -> written from 6
// Visibility: every wave waits `vmcnt(2)` (only B0(t+2) may stay in flight) before the barrier that closes
// interval 7 -- group 0 at the end of M4, group 1 at the end of R4 -- and tile t+1 is first read in interval 8.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256s_bf16_nt_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    const int tiles_n = p.N >> 8;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int row0 = tile_m * 256, col0 = tile_n * 256;

    const int srow = lane >> 3, schunk = lane & 7;
    const int r0 = wid * 16 + srow, r1 = r0 + 8;
    const bf16_t* a_src0 = p.A + (size_t)(row0 + r0) * p.lda + ((schunk ^ ((r0 >> 1) & 7)) << 3);
    const bf16_t* a_src1 = p.A + (size_t)(row0 + r1) * p.lda + ((schunk ^ ((r1 >> 1) & 7)) << 3);
    const bf16_t* w_src0 = p.W + (size_t)(col0 + r0) * p.ldw + ((schunk ^ ((r0 >> 1) & 7)) << 3);
    const bf16_t* w_src1 = p.W + (size_t)(col0 + r1) * p.ldw + ((schunk ^ ((r1 >> 1) & 7)) << 3);
    const size_t a_half = (size_t)128 * p.lda, w_half = (size_t)128 * p.ldw;
    char* const stage_base = smem + wid * 2048;

    auto stage_a = [&](int half, int tau) {
        char* dst = stage_base + (tau & 1) * TBUF + half * THALF;
        glds16b(a_src0 + half * a_half + tau * 64, dst);
        glds16b(a_src1 + half * a_half + tau * 64, dst + 1024);
    };
    auto stage_w = [&](int half, int tau) {
        char* dst = stage_base + (tau & 1) * TBUF + (2 + half) * THALF;
        glds16b(w_src0 + half * w_half + tau * 64, dst);
        glds16b(w_src1 + half * w_half + tau * 64, dst + 1024);
    };

    const int lrow = lane & 15, lq = lane >> 4;
    const int swz = lrow >> 1;
    const int co0 = ((0 + lq) ^ swz) << 4, co1 = ((4 + lq) ^ swz) << 4;
    const int a_off = wr * THALF + lrow * 128;
    const int b_off = 2 * THALF + (wc >> 1) * THALF + ((wc & 1) * 64 + lrow) * 128;

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+v"(acc[mi][ni]));     // zero materialised here, far from the asm MFMAs (see gemm256p.hip)
        }

    const int nt = p.K >> 6;
    stage_a(0, 0); stage_a(1, 0); stage_w(0, 0); stage_w(1, 0);
    if (nt > 1) {
        stage_w(0, 1);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one interval behind

    bf16x8 af[4][2], w0[2][2], w1[2][2];
    for (int t = 0; t < nt; ++t) {
        const char* sa = smem + (t & 1) * TBUF + a_off;
        const char* sb = smem + (t & 1) * TBUF + b_off;
        const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
        // ---- R1 / M1
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            w0[ni][0] = *(const bf16x8*)(sb + ni * 2048 + co0);
            w0[ni][1] = *(const bf16x8*)(sb + ni * 2048 + co1);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            af[mi][0] = *(const bf16x8*)(sa + mi * 2048 + co0);
            af[mi][1] = *(const bf16x8*)(sa + mi * 2048 + co1);
        }
        if (more1) stage_w(1, t + 1);
        __builtin_amdgcn_s_barrier();
        quadrant_pinned<0, 0>(acc, af, w0);
        __builtin_amdgcn_s_barrier();
        // ---- R2 / M2
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            w1[ni][0] = *(const bf16x8*)(sb + 4096 + ni * 2048 + co0);
            w1[ni][1] = *(const bf16x8*)(sb + 4096 + ni * 2048 + co1);
        }
        if (more1) stage_a(0, t + 1);
        __builtin_amdgcn_s_barrier();
        quadrant_pinned<0, 1>(acc, af, w1);
        __builtin_amdgcn_s_barrier();
        // ---- R3 / M3
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            af[mi][0] = *(const bf16x8*)(sa + 8192 + mi * 2048 + co0);
            af[mi][1] = *(const bf16x8*)(sa + 8192 + mi * 2048 + co1);
        }
        if (more1) stage_a(1, t + 1);
        __builtin_amdgcn_s_barrier();
        quadrant_pinned<1, 1>(acc, af, w1);
        __builtin_amdgcn_s_barrier();
        // ---- R4 / M4
        if (more2) stage_w(0, t + 2);
        if (wr == 1) {                                   // group 1 closes interval 7 here
            if (more2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        quadrant_pinned<1, 0>(acc, af, w0);
        if (wr == 0) {                                   // group 0 closes interval 7 here
            if (more2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();          // balance the extra barrier group 1 took at the start
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA result -> VALU read hazard of the asm MFMAs (>= 12 states)

#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = row0 + wr * 128 + mi * 16 + lrow;
        if (m >= p.M || (p.dbg & 1)) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = col0 + wc * 64 + ni * 16 + lq * 4;
            f32x4 v = acc[mi][ni];
            if (p.bias) {
                const float4 b = *(const float4*)(p.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if constexpr (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
            }
            if constexpr (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16) {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
            } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
                float4* dst = (float4*)((float*)p.C + (size_t)m * p.ldc + n);
                float4 x = *dst;
                x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
                *dst = x;
            } else {
                const int img = m / p.patches, pi = m - img * p.patches;
                const float4 pe = *(const float4*)(p.pos + (size_t)(pi + 1) * p.N + n);
                float4 x;
                x.x = v[0] + pe.x; x.y = v[1] + pe.y; x.z = v[2] + pe.z; x.w = v[3] + pe.w;
                *(float4*)((float*)p.C + (size_t)(m + img + 1) * p.ldc + n) = x;
            }
        }
    }
}

template <int EPI>
static int launch256(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 2 * TBUF;
    auto kern_lock = gemm256_bf16_nt_kernel<EPI>;      // the product kernel (variant 2): one barrier per phase, waves in lockstep
#ifdef KEMR_AB_VARIANTS
    auto kern_stag = gemm256s_bf16_nt_kernel<EPI>;     // variant 3: staggered wave halves (measured 1-9 % slower; A/B builds only)
#endif
    static bool attr_done = false;
    if (!attr_done) {
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern_lock, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
#ifdef KEMR_AB_VARIANTS
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern_stag, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
#endif
        attr_done = true;
    }
    const int tiles = ((p.M + 255) / 256) * (p.N / 256);
    ProfScope prof(PROF_GEMM, stream);
#ifdef KEMR_AB_VARIANTS
    if (g_gemm_variant == 3) hipLaunchKernelGGL(kern_stag, dim3(tiles), dim3(512), smem, stream, p);
    else
#endif
    { GemmParams q = p; q.dbg = g_gemm_dbg; hipLaunchKernelGGL(kern_lock, dim3(tiles), dim3(512), smem, stream, q); }
    KEMR_CHECK_LAUNCH("gemm256_bf16_nt_kernel");
    return KEMR_OK;
}

int launch_gemm256(const GemmParams& p, int epi, hipStream_t stream) {
    switch (epi) {
        case EPI_BIAS_BF16:       return launch256<EPI_BIAS_BF16>(p, stream);
        case EPI_BIAS_QGELU_BF16: return launch256<EPI_BIAS_QGELU_BF16>(p, stream);
        case EPI_BIAS_RESID_F32:  return launch256<EPI_BIAS_RESID_F32>(p, stream);
        case EPI_PATCH_F32:       return launch256<EPI_PATCH_F32>(p, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "gemm256: unknown epilogue %d", epi);
}

}  // namespace kemr
