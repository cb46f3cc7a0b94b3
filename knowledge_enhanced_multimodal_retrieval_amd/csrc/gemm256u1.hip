// Persistent bf16 GEMM for the bf16-output epilogues, 8 waves x (128 x 64), the staggered K loop of gemm256p.hip with ONE
// K-tile pipeline that runs across output tiles ("uniform"): the LDS-DMA pieces staged during K-tile g belong to K-tiles
// g+1 (W1, A0, A1) and g+2 (W0) of the workgroup's whole tile sequence, so the first K-tiles of the next output tile
// are already in LDS when this tile's last MFMA retires, and a tile switch is just: epilogue (16 stores per lane), zero
// the accumulators, carry on.  gemm256p.hip paid a prologue per tile (issue 16 pieces, wait for K-tile 0: ~2.6 us x 12-16
// tiles per workgroup on the encoder shapes).
//
// vmcnt (loads, LDS-DMA and stores retire in order): the wait that closes K-tile g needs K-tile g+1 landed and may leave
// W0(g+2) (2 pieces per lane) in flight: vmcnt(2), or vmcnt(0) when nothing was staged behind it.  The 16 stores of a
// tile sit between W0(g+2) and W1(g+2) in that order, so they get one K-tile of time before a wait covers them.
// The next tile's 256 bias floats ride with its first W0 piece (wave 0) into the other half of a 2 x 1 KiB LDS area;
// the epilogue reads bias and staging area with asm LDS ops (no memory operand, so hipcc adds no vmcnt(0) for them).
#include "common.h"

namespace kemr {

__device__ unsigned g_gemm_stamp_buf1[1024 * 16];

namespace {

constexpr int PBUF = 65536;      // bytes per K-tile buffer
constexpr int PHALF = 16384;     // bytes per half-tile
constexpr int PEPI = 131072;     // offset of the epilogue area (8 waves x 2 KiB)
constexpr int PBIAS = PEPI + 16384;   // 2 x 1 KiB: fp32 bias of the current / next tile's 256 columns
constexpr int PSCALE = PBIAS + 2048;  // fp8 operands: 2 x 1 KiB per-output-channel weight scales of the current / next tile
constexpr int PSMEM = PSCALE + 2048;

__device__ __forceinline__ void glds16u1(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Epilogue LDS traffic as inline asm: hipcc (SIInsertWaitcnts) guards every LDS access that carries a memory operand
// with `s_waitcnt vmcnt(0)` while an LDS-DMA is outstanding, which would drain the next tile's prefetch.  The wave-
// private epilogue area is never a DMA target, so no such wait is needed; asm LDS ops carry no memory operand.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void lds_write_b64(unsigned addr, u32x2 v) {
    // the trailing s_nop keeps hipcc from overwriting the data registers while the LDS unit still reads them (observed:
    // a packed VALU op right behind the asm store corrupted the second data dword; cdna guide 5.7 item 1, "Stores")
    asm volatile("ds_write_b64 %0, %1\n\ts_nop 2" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
    u32x4 d;
    asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory");
    return d;
}

template <int MH, int NH>
__device__ __forceinline__ void quad(f32x4 (&acc)[8][4], const bf16x8 (&af)[4][2], const bf16x8 (&wf)[2][2]) {
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

// fp8 (OCP e4m3) operands: one block-scaled MFMA covers K = 128 (a lane holds 32 consecutive k bytes of its row), at twice
// the bf16 MFMA's cycles, i.e. twice the FLOP rate; the E8M0 block scales are all 2^0 (0x7f), the real scales (one per
// output channel of W) are applied in the epilogue.  A K-tile is still 128 bytes per row: staging, LDS image and swizzle
// are the bf16 kernel's, a fragment is two adjacent 16-byte chunks instead of one.
typedef __attribute__((ext_vector_type(8))) int fp8x32;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int MH, int NH>
__device__ __forceinline__ void quad8(f32x4 (&acc)[8][4], const fp8x32 (&af)[4], const fp8x32 (&wf)[2], int one) {
    __builtin_amdgcn_s_setprio(1);
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                         : "+v"(acc[MH * 4 + mi][NH * 2 + ni]) : "v"(wf[ni]), "v"(af[mi]), "v"(one));
    __builtin_amdgcn_s_setprio(0);
}

}  // namespace

template <int EPI, bool FP8, bool AFIRST, bool DBG = false>
__global__ __launch_bounds__(512, 2) void gemm256u1_bf16_nt_kernel(const GemmParams p) {
    constexpr int ES = FP8 ? 1 : 2;          // operand element size
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    const int tiles_n = p.N >> 8;
    const int ntiles = ((p.M + 255) >> 8) * tiles_n;
    const int full = (ntiles / (int)gridDim.x) * (int)gridDim.x;     // tiles inside complete rounds
    auto tile_of = [&](int idx, int& row0, int& col0) {              // XCD-contiguous order, as gemm256p.hip
        int L = idx;
        if (idx < full && (gridDim.x & 7) == 0) {
            const int rnd = idx / (int)gridDim.x, b = idx - rnd * (int)gridDim.x;
            L = rnd * (int)gridDim.x + (b & 7) * ((int)gridDim.x >> 3) + (b >> 3);
        }
        const int tm = L / tiles_n;
        row0 = tm << 8;
        col0 = (L - tm * tiles_n) << 8;
    };
    const int nt = FP8 ? p.K >> 7 : p.K >> 6;      // K-tile = 128 bytes per row
    // Order of the three pieces of K-tile g+1 staged during K-tile g.  Long K (fc2: A is the 0.5 GB MLP hidden, streamed
    // from HBM, 4 tiles per A panel): the A halves first, a full K-tile ahead of their use, then W1 (measured, sustained:
    // 470 -> 443 us at K = 4096).  Short K (A panels shared by 12-16 column tiles, mostly L2 hits): W1, A0, A1 spread over
    // the first three intervals is 1 % faster.
    // (a template parameter: a run-time branch at the three staging sites of this loop costs several per cent)
    constexpr bool afirst = AFIRST;

    // staging addresses = wave-uniform K-tile base (SGPRs) + a per-lane 32-bit byte offset that never changes
    const int srow = lane >> 3, schunk = lane & 7;
    const int r0 = wid * 16 + srow, r1 = r0 + 8;
    const unsigned a_lane0 = (unsigned)(r0 * p.lda * ES + ((schunk ^ ((r0 >> 1) & 7)) << 4));
    const unsigned a_lane1 = (unsigned)(r1 * p.lda * ES + ((schunk ^ ((r1 >> 1) & 7)) << 4));
    const unsigned w_lane0 = (unsigned)(r0 * p.ldw * ES + ((schunk ^ ((r0 >> 1) & 7)) << 4));
    const unsigned w_lane1 = (unsigned)(r1 * p.ldw * ES + ((schunk ^ ((r1 >> 1) & 7)) << 4));
    const size_t a_half = (size_t)128 * p.lda * ES, w_half = (size_t)128 * p.ldw * ES;      // bytes between the two half-tiles
    char* const stage_base = smem + wid * 2048;

    // a K-tile of the workgroup's tile sequence (all wave-uniform)
    struct Cur { const char* a; const char* w; int idx, tau, par, seq, col0; bool valid; };
    auto cur_set = [&](Cur& c) {
        int row0, col0;
        tile_of(c.idx, row0, col0);
        c.a = (const char*)p.A + (size_t)row0 * p.lda * ES;
        c.w = (const char*)p.W + (size_t)col0 * p.ldw * ES;
        c.col0 = col0;
    };
    auto cur_next = [&](Cur& c) {
        c.par ^= 1;
        if (++c.tau == nt) {
            c.tau = 0;
            c.seq++;
            c.idx += gridDim.x;
            c.valid = c.idx < ntiles;
            if (c.valid) cur_set(c);
        }
    };
    auto stage_a = [&](int half, const Cur& c) {
        char* dst = stage_base + c.par * PBUF + half * PHALF;
        const char* src = c.a + half * a_half + c.tau * 128;
        glds16u1(src + a_lane0, dst);
        glds16u1(src + a_lane1, dst + 1024);
    };
    auto stage_w = [&](int half, const Cur& c) {
        char* dst = stage_base + c.par * PBUF + (2 + half) * PHALF;
        const char* src = c.w + half * w_half + c.tau * 128;
        glds16u1(src + w_lane0, dst);
        glds16u1(src + w_lane1, dst + 1024);
    };
    auto stage_w0 = [&](const Cur& c) {       // first piece of a K-tile; with K-tile 0 of a tile: that tile's bias (wave 0)
        if (c.tau == 0 && wid == 0 && p.bias) glds16u1(p.bias + c.col0 + lane * 4, smem + PBIAS + (c.seq & 1) * 1024);
        if (FP8 && c.tau == 0 && wid == 0) glds16u1(p.wscale + c.col0 + lane * 4, smem + PSCALE + (c.seq & 1) * 1024);
        stage_w(0, c);
    };

    const int lrow = lane & 15, lq = lane >> 4;
    const int swz = lrow >> 1;
    // bf16: fragments of the two 32-wide k steps (chunks lq and 4 + lq); fp8: the two halves of ONE 32-byte fragment
    const int co0 = ((FP8 ? 2 * lq : lq) ^ swz) << 4, co1 = ((FP8 ? 2 * lq + 1 : 4 + lq) ^ swz) << 4;
    const int a_off = wr * PHALF + lrow * 128;
    const int b_off = 2 * PHALF + (wc >> 1) * PHALF + ((wc & 1) * 64 + lrow) * 128;

    if (!p.bias && tid < 128) *(float4*)(smem + PBIAS + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    // prologue: K-tile 0 complete and W0 of K-tile 1 (nt >= 2, so both belong to the first tile)
    Cur c1;
    c1.idx = blockIdx.x; c1.tau = 0; c1.par = 0; c1.seq = 0; c1.valid = true;
    cur_set(c1);
    stage_w0(c1); stage_w(1, c1); stage_a(0, c1); stage_a(1, c1);
    cur_next(c1);                        // c1 = K-tile 1: its W1, A0, A1 are staged during K-tile 0
    stage_w0(c1);
    Cur c2 = c1;
    cur_next(c2);                        // c2 = K-tile 2: its W0 is staged at the end of K-tile 0
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // wr == 1 half runs one interval behind for the whole launch

    unsigned long long clk0 = 0, rt0 = 0;
    if constexpr (DBG) {
        if ((p.dbg & 128) && p.stamps && wid == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0), "=s"(rt0) :: "memory");
    }
    int gpar = 0, seq = 0;
    bf16x8 af[4][2], w0[2][2], w1[2][2];          // bf16 operands
    fp8x32 af8[4], w08[2], w18[2];                 // fp8 operands (only one set is live, by FP8)
    int one = 0x7f7f7f7f;                          // E8M0 block scales 2^0
    asm volatile("" : "+v"(one));
    auto ld_w = [&](bf16x8 (&w)[2][2], fp8x32 (&w8)[2], const char* q) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if constexpr (FP8) {
                w8[ni].lo = *(const i32x4*)(q + ni * 2048 + co0);
                w8[ni].hi = *(const i32x4*)(q + ni * 2048 + co1);
            } else {
                w[ni][0] = *(const bf16x8*)(q + ni * 2048 + co0);
                w[ni][1] = *(const bf16x8*)(q + ni * 2048 + co1);
            }
        }
    };
    auto ld_a = [&](const char* q) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            if constexpr (FP8) {
                af8[mi].lo = *(const i32x4*)(q + mi * 2048 + co0);
                af8[mi].hi = *(const i32x4*)(q + mi * 2048 + co1);
            } else {
                af[mi][0] = *(const bf16x8*)(q + mi * 2048 + co0);
                af[mi][1] = *(const bf16x8*)(q + mi * 2048 + co1);
            }
        }
    };
    for (int idx = blockIdx.x; idx < ntiles; idx += gridDim.x, ++seq) {
        f32x4 acc[8][4];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+v"(acc[mi][ni]));    // materialise the zeros here (see gemm256p.hip)
            }

        for (int t = 0; t < nt; ++t, gpar ^= 1) {
            const char* sa = smem + gpar * PBUF + a_off;
            const char* sb = smem + gpar * PBUF + b_off;
            ld_w(w0, w08, sb);
            ld_a(sa);
            if (c1.valid) { if (afirst) { stage_a(0, c1); stage_a(1, c1); } else stage_w(1, c1); }
            __builtin_amdgcn_s_barrier();
            if constexpr (FP8) quad8<0, 0>(acc, af8, w08, one); else quad<0, 0>(acc, af, w0);
            __builtin_amdgcn_s_barrier();
            ld_w(w1, w18, sb + 4096);
            if (c1.valid) { if (afirst) stage_w(1, c1); else stage_a(0, c1); }
            __builtin_amdgcn_s_barrier();
            if constexpr (FP8) quad8<0, 1>(acc, af8, w18, one); else quad<0, 1>(acc, af, w1);
            __builtin_amdgcn_s_barrier();
            ld_a(sa + 8192);
            if (c1.valid && !afirst) stage_a(1, c1);
            __builtin_amdgcn_s_barrier();
            if constexpr (FP8) quad8<1, 1>(acc, af8, w18, one); else quad<1, 1>(acc, af, w1);
            __builtin_amdgcn_s_barrier();
            const bool v1 = c1.valid, v2 = c2.valid;
            if (v2) stage_w0(c2);
            auto close_tile = [&]() {           // K-tile g+1 landed (everything older too, the last tile's stores included)
                if (v2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (v1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            };
            if (wr == 1) close_tile();
            __builtin_amdgcn_s_barrier();
            if constexpr (FP8) quad8<1, 0>(acc, af8, w08, one); else quad<1, 0>(acc, af, w0);
            if (wr == 0) close_tile();
            __builtin_amdgcn_s_barrier();
            c1 = c2;
            cur_next(c2);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMA result -> VALU read (>= 12 wait states)

        // ---- epilogue (lane constants behind an opaque copy of `lane`: recomputed here, not kept across the K loop)
        int el = lane;
        asm volatile("" : "+v"(el));
        const int erow = el & 15, eq = el >> 4;
        const int er = el >> 3, ec = el & 7;                              // read-back: row er (+8), chunk ec
        const unsigned epi = lds_addr(smem + PEPI) + wid * 2048;          // wave-private: 16 rows x 128 B, chunk ^= row & 7
        const unsigned c_lane = (unsigned)(er * p.ldc + ec * 8) * 2u;
        const unsigned epi_w = epi + erow * 128 + (((eq >> 1) ^ (erow & 7)) << 4) + (eq & 1) * 8;
        const unsigned epi_r0 = epi + er * 128 + ((ec ^ er) << 4);        // rows er and er + 8: (er + 8) & 7 == er
        const unsigned epi_r1 = epi_r0 + 1024;
        const unsigned bias_r = lds_addr(smem + PBIAS) + (seq & 1) * 1024 + (wc * 64 + eq * 4) * 4;
        int row0, col0;
        tile_of(idx, row0, col0);
        char* const c_tile = (char*)p.C + ((size_t)(row0 + wr * 128) * p.ldc + col0 + wc * 64) * 2;
        u32x4 bias[4], wsc[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bias[ni] = lds_read_b128(bias_r + ni * 64);
        if constexpr (FP8) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wsc[ni] = lds_read_b128(bias_r + (PSCALE - PBIAS) + ni * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wsc[0]), "+v"(wsc[1]), "+v"(wsc[2]), "+v"(wsc[3]) :: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]) :: "memory");
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {                 // 8 passes of 16 rows through the wave's private LDS area
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 v = acc[mi][ni];
                if constexpr (FP8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], __uint_as_float(wsc[ni][r]), __uint_as_float(bias[ni][r]));
                } else {
                    v[0] += __uint_as_float(bias[ni][0]); v[1] += __uint_as_float(bias[ni][1]);
                    v[2] += __uint_as_float(bias[ni][2]); v[3] += __uint_as_float(bias[ni][3]);
                }
                if constexpr (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                }
                u32x2 o;
                o[0] = pack_bf16x2(v[0], v[1]);
                o[1] = pack_bf16x2(v[2], v[3]);
                lds_write_b64(epi_w ^ (ni * 32), o);      // chunk (ni*2 + (eq>>1)) ^ (erow & 7): ni only flips bits 5-6
            }
            u32x4 d0 = lds_read_b128(epi_r0), d1 = lds_read_b128(epi_r1);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1) :: "memory");
            // Non-temporal stores: C is not read again by this kernel, and written as ordinary write-back lines the 0.1-0.5 GB
            // of a launch evicts the A / W panels the K loops live on from L2 (measured on the encoder shapes: plain stores
            // +27 % time on QKV, +17 % on fc1 over no stores at all; `nt` stores +0 % / +6 %; `sc1` write-through +13 %;
            // the same stores aimed at an L2-resident 8 MiB cost nothing, so the instruction issue is not the price).
            // Inside the encoder chain (bench.py, same device): all four block GEMMs with plain stores 13 490 items/s,
            // all non-temporal 14 065; every one of the four contributes, the LayerNorm reading the deltas included.
            // Inline asm because __builtin_nontemporal_store did not produce this encoding; 2 stores per lane and pass,
            // the vmcnt bookkeeping in the header counts them.
            // the timing-experiment flags exist only in the DBG instantiation (tools/: any non-zero flag selects it)
            if (!DBG || (!(p.dbg & 1) && !((p.dbg & 32) && mi >= 4))) {          // dbg 1 / 32: TIMING ONLY, all / half the stores dropped
                char* q0 = c_tile + (size_t)(mi * 16) * p.ldc * 2 + c_lane;
                char* q1 = c_tile + (size_t)(mi * 16 + 8) * p.ldc * 2 + c_lane;
                if (DBG && (p.dbg & 4))
                    asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %2, %3, off\n\ts_nop 1"
                                 :: "v"(q0), "v"(d0), "v"(q1), "v"(d1) : "memory");        // dbg 4: plain stores, for A/B
                else
                    asm volatile("global_store_dwordx4 %0, %1, off nt\n\tglobal_store_dwordx4 %2, %3, off nt\n\ts_nop 1"
                                 :: "v"(q0), "v"(d0), "v"(q1), "v"(d1) : "memory");
            }
        }
    }
    if constexpr (DBG) {
        if ((p.dbg & 128) && p.stamps && wid == 0 && lane == 0) {
            unsigned long long clk1, rt1;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1), "=s"(rt1) :: "memory");
            p.stamps[blockIdx.x * 16 + 12] = (unsigned)(clk1 - clk0);
            p.stamps[blockIdx.x * 16 + 13] = (unsigned)(rt1 - rt0);
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();      // pairs with the extra barrier the wr == 1 half took at the start
}

template <int EPI, bool FP8, bool AFIRST, bool DBG>
static int launch256u1_a(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm256u1_bf16_nt_kernel<EPI, FP8, AFIRST, DBG>;
    static bool attr_done = false;
    static int num_cu = 0;
    if (!attr_done) {
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PSMEM));
        int dev = 0;
        KEMR_CHECK_HIP(hipGetDevice(&dev));
        KEMR_CHECK_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_done = true;
    }
    const int tiles = ((p.M + 255) / 256) * (p.N / 256);
    const int grid = tiles < num_cu ? tiles : num_cu;
    GemmParams q = p;
    q.dbg = g_gemm_dbg;
    q.stamps = nullptr;
    if (DBG && (g_gemm_dbg & 128)) KEMR_CHECK_HIP(hipGetSymbolAddress((void**)&q.stamps, HIP_SYMBOL(g_gemm_stamp_buf1)));
    ProfScope prof(PROF_GEMM, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), PSMEM, stream, q);
    KEMR_CHECK_LAUNCH("gemm256u1_bf16_nt_kernel");
    return KEMR_OK;
}

template <int EPI, bool FP8>
static int launch256u1(const GemmParams& p, hipStream_t stream) {
    const int nt = FP8 ? p.K >> 7 : p.K >> 6;
    if (g_gemm_dbg && !FP8) return nt >= 32 ? launch256u1_a<EPI, false, true, true>(p, stream) : launch256u1_a<EPI, false, false, true>(p, stream);
    return nt >= 32 ? launch256u1_a<EPI, FP8, true, false>(p, stream) : launch256u1_a<EPI, FP8, false, false>(p, stream);
}

int gemm_read_stamps1(unsigned* host_out, int n_words) {
    KEMR_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamp_buf1), (size_t)n_words * 4, 0, hipMemcpyDeviceToHost));
    return KEMR_OK;
}

// C must have ceil256(M) rows: rows in [M, ceil256(M)) are written (with values computed from A's pad rows).
int launch_gemm256u1(const GemmParams& p, int epi, hipStream_t stream) {
    switch (epi) {
        case EPI_BIAS_BF16:       return launch256u1<EPI_BIAS_BF16, false>(p, stream);
        case EPI_BIAS_QGELU_BF16: return launch256u1<EPI_BIAS_QGELU_BF16, false>(p, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "gemm256u: epilogue %d is not a bf16-store epilogue", epi);
}

// fp8 e4m3 operands: A [ceil256(M), lda] and W [N, ldw] in bytes, K % 128 == 0, K >= 256, N % 256 == 0; p.wscale[N] scales the
// accumulators per output channel before the bias; C is bf16 as above.
int launch_gemm256u1_fp8(const GemmParams& p, int epi, hipStream_t stream) {
    if (p.M <= 0) return KEMR_OK;
    if (p.N % 256 != 0 || p.K % 128 != 0 || p.K < 256) KEMR_FAIL(KEMR_ERR_INVALID, "gemm fp8: need N %% 256 == 0, K %% 128 == 0, K >= 256 (got N=%d K=%d)", p.N, p.K);
    if ((p.lda % 16) || (p.ldw % 16) || (p.ldc % 4)) KEMR_FAIL(KEMR_ERR_INVALID, "gemm fp8: leading dimensions must keep 16-byte alignment");
    if (!p.wscale || !p.c_rows_padded) KEMR_FAIL(KEMR_ERR_INVALID, "gemm fp8: needs weight scales and a row-padded C");
    switch (epi) {
        case EPI_BIAS_BF16:       return launch256u1<EPI_BIAS_BF16, true>(p, stream);
        case EPI_BIAS_QGELU_BF16: return launch256u1<EPI_BIAS_QGELU_BF16, true>(p, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "gemm fp8: epilogue %d is not a bf16-store epilogue", epi);
}

}  // namespace kemr
