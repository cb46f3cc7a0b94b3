// Persistent bf16 GEMM, "wide wave" layout: one 256-thread workgroup per CU walks 256 x 256 output tiles, each of the
// 4 waves (one per SIMD, 512 registers) owns a 128 x 128 quadrant: 256 accumulator registers (AGPRs) per lane.
// Against the 8-wave layout of gemm256p.hip (128 x 64 per wave) this cuts the LDS fragment traffic per K-tile from
// 192 KiB to 128 KiB per CU -- with the 64 KiB of LDS-DMA writes the 8-wave kernel keeps the LDS as busy as the matrix
// cores -- and leaves one instruction stream per SIMD, scheduled by hand below.
//
// Pipeline (global K-tile counter g per workgroup, running ACROSS output tiles; buffer = g & 1, 64 KiB each):
//   S(g):   s_waitcnt vmcnt(N) [K-tile g landed] ; barrier A ; 16 ds_read_b128 (k 0..31 fragments)
//   phase 1: 64 MFMA on k 0..31; the 16 reads of the k 32..63 fragments ride in the first 16 MFMA slots;
//            after slot 23: lgkmcnt(0) ; barrier B  [every wave has the whole K-tile in registers: buffer g & 1 is free]
//            then the 16 LDS-DMA of K-tile g + 2 (possibly the NEXT output tile's) ride in the following MFMA slots
//   phase 2: 64 MFMA on k 32..63
// so a K-tile's loads are in flight for ~1.6 K-tile times, and a tile switch needs no prologue: the next tile's K-tiles
// 0 and 1 are staged during this tile's last two K-tiles, the epilogue stores (32 per lane) drain behind the next tile's
// first two K-tiles.  vmcnt bookkeeping (loads, LDS-DMA and stores retire in order): at S(g) the ops younger than K-tile g
// are K-tile g+1 (16, if any) and -- for the first two K-tiles after a tile switch -- the 32 stores: N in {0,16,32,48}.
// Every lane issues exactly 32 stores per tile (C must have ceil256(M) rows).  MFMAs are volatile asm (hipcc would sink
// builtins across the barriers); accumulators are pinned to AGPRs through the "a" constraint.
#include "common.h"

namespace kemr {

namespace {

constexpr int WBUF = 65536;            // bytes per K-tile buffer: A rows 0..255 (32 KiB) | W rows 0..255 (32 KiB)
constexpr int WHALF = 32768;
constexpr int WEPI = 131072;           // 4 waves x 4 KiB epilogue staging (16 rows x 256 B)
constexpr int WBIAS = WEPI + 16384;    // 2 x 1 KiB fp32 bias of the current / next tile
constexpr int WSMEM = WBIAS + 2048;

__device__ __forceinline__ void glds16w(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ unsigned lds_addr_w(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// asm LDS accesses carry no memory operand: hipcc does not guard them with vmcnt(0) while LDS-DMA is in flight
__device__ __forceinline__ void lds_write_b64w(unsigned addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1\n\ts_nop 2" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 lds_read_b128w(unsigned addr) {
    u32x4 d;
    asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory");
    return d;
}
__device__ __forceinline__ void mfma_acc(f32x4& acc, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}

}  // namespace

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm256w_bf16_nt_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;

    const int tiles_n = p.N >> 8;
    const int ntiles = ((p.M + 255) >> 8) * tiles_n;
    const int full = (ntiles / (int)gridDim.x) * (int)gridDim.x;
    auto tile_of = [&](int idx, int& row0, int& col0) {      // same XCD-contiguous order as gemm256p.hip
        int L = idx;
        if (idx < full && (gridDim.x & 7) == 0) {
            const int rnd = idx / (int)gridDim.x, b = idx - rnd * (int)gridDim.x;
            L = rnd * (int)gridDim.x + (b & 7) * ((int)gridDim.x >> 3) + (b >> 3);
        }
        const int tm = L / tiles_n;
        row0 = tm << 8;
        col0 = (L - tm * tiles_n) << 8;
    };
    const int nt = p.K >> 6;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g_total = my_tiles * nt;

    // ---- staging: wave w moves rows w*64 .. w*64+63 of the A and of the W tile, 8 rows (1 KiB) per LDS-DMA.
    // LDS row = 128 B = 8 chunks of 16 B; physical chunk c of row r holds logical chunk c ^ ((r >> 1) & 7).
    // With r = w*64 + i*8 + (lane >> 3) the key is (lane >> 4) for even i and (lane >> 4) + 4 for odd i.
    const int srow = lane >> 3, schunk = lane & 7;
    const unsigned a_le = (unsigned)(srow * p.lda + ((schunk ^ (srow >> 1)) << 3)) * 2u;
    const unsigned a_lo = (unsigned)(srow * p.lda + ((schunk ^ ((srow >> 1) + 4)) << 3)) * 2u;
    const unsigned w_le = (unsigned)(srow * p.ldw + ((schunk ^ (srow >> 1)) << 3)) * 2u;
    const unsigned w_lo = (unsigned)(srow * p.ldw + ((schunk ^ ((srow >> 1) + 4)) << 3)) * 2u;
    const size_t a_step = (size_t)8 * p.lda * 2, w_step = (size_t)8 * p.ldw * 2;
    char* const stage_base = smem + wid * 64 * 128;

    int s_idx = blockIdx.x, s_tau = 0, s_par = 0, s_seq = 0, s_col0 = 0;   // staging cursor (wave-uniform)
    const char *s_a = nullptr, *s_w = nullptr;
    auto s_set = [&]() {
        int r0, c0;
        tile_of(s_idx, r0, c0);
        s_a = (const char*)p.A + ((size_t)r0 + wid * 64) * p.lda * 2;
        s_w = (const char*)p.W + ((size_t)c0 + wid * 64) * p.ldw * 2;
        s_col0 = c0;
    };
    auto stage_one = [&](int j) {       // j = 0..7: A rows, 8..15: W rows (compile-time after unrolling)
        const int i = j & 7;
        char* dst = stage_base + s_par * WBUF + (j < 8 ? 0 : WHALF) + i * 1024;
        if (j < 8) glds16w(s_a + i * a_step + s_tau * 128 + ((i & 1) ? a_lo : a_le), dst);
        else glds16w(s_w + i * w_step + s_tau * 128 + ((i & 1) ? w_lo : w_le), dst);
    };
    auto stage_bias = [&]() {           // with K-tile 0 of a tile: its 256 bias floats (wave 0)
        if (s_tau == 0 && wid == 0 && p.bias) glds16w(p.bias + s_col0 + lane * 4, smem + WBIAS + (s_seq & 1) * 1024);
    };
    auto stage_advance = [&]() {
        s_par ^= 1;
        if (++s_tau == nt) {
            s_tau = 0;
            s_seq++;
            s_idx += gridDim.x;
            if (s_idx < ntiles) s_set();
        }
    };

    // ---- fragment addressing (MFMA 16x16x32: lane (lrow, lq) reads 8 consecutive k of row lrow at k = 32*kk + 8*lq)
    const int lrow = lane & 15, lq = lane >> 4;
    const int swz = lrow >> 1;
    const int co0 = ((0 + lq) ^ swz) << 4, co1 = ((4 + lq) ^ swz) << 4;
    const int a_off = (wr * 128 + lrow) * 128;
    const int b_off = WHALF + (wc * 128 + lrow) * 128;

    if (!p.bias && tid < 128) *(float4*)(smem + WBIAS + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    // prologue: K-tiles 0 and 1 of the first tile (g_total >= 2 because nt >= 2)
    s_set();
    stage_bias();
#pragma unroll
    for (int j = 0; j < 16; ++j) stage_one(j);
    stage_advance();
#pragma unroll
    for (int j = 0; j < 16; ++j) stage_one(j);
    stage_advance();
    int g_staged = 2;

    int g = 0, par = 0, seq = 0;
    bool had_stores = false;
    bf16x8 a0[8], b0[8], a1[8], b1[8];

    for (int idx = blockIdx.x; idx < ntiles; idx += gridDim.x, ++seq) {
        f32x4 acc[8][8];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+a"(acc[mi][ni]));     // materialise the zeros here, not in front of the first MFMA
            }

        for (int t = 0; t < nt; ++t, ++g, par ^= 1) {
            // ---- S(g)
            const bool nxt = g + 1 < g_total, st = had_stores && t < 2;
            if (nxt) {
                if (st) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else {
                if (st) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            const char* sa = smem + par * WBUF + a_off;
            const char* sb = smem + par * WBUF + b_off;
            a0[0] = *(const bf16x8*)(sa + co0);
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) b0[ni] = *(const bf16x8*)(sb + ni * 2048 + co0);
#pragma unroll
            for (int mi = 1; mi < 8; ++mi) a0[mi] = *(const bf16x8*)(sa + mi * 2048 + co0);
            const bool do_stage = g_staged < g_total;      // K-tile g + 2

            asm volatile("s_nop 1" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
            // ---- phase 1
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                mfma_acc(acc[s >> 3][s & 7], b0[s & 7], a0[s >> 3]);
                if (s == 0) a1[0] = *(const bf16x8*)(sa + co1);
                else if (s <= 8) b1[s - 1] = *(const bf16x8*)(sb + (s - 1) * 2048 + co1);
                else if (s <= 15) a1[s - 8] = *(const bf16x8*)(sa + (s - 8) * 2048 + co1);
                if (s == 23) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();           // B: buffer `par` is free
                    if (do_stage) stage_bias();
                }
                if (s >= 24 && s < 56 && (s & 1) == 0 && do_stage) stage_one((s - 24) >> 1);
            }
            if (do_stage) { stage_advance(); ++g_staged; }
            // ---- phase 2
#pragma unroll
            for (int s = 0; s < 64; ++s) mfma_acc(acc[s >> 3][s & 7], b1[s & 7], a1[s >> 3]);
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMA result -> accvgpr read

        // ---- epilogue: 8 passes of 16 rows x 128 columns through the wave's private LDS area (4 KiB = 16 rows x 256 B,
        // 16-byte chunk index ^= row).  The lane constants are derived behind an opaque copy of `lane` so that hipcc
        // recomputes them here instead of keeping ~10 registers alive across the K loop (it spilled them to scratch,
        // and scratch reloads count in vmcnt).
        int el = lane;
        asm volatile("" : "+v"(el));
        const int erow = el & 15, eq = el >> 4;                           // MFMA layout: row (m), 4-column group
        const int er = el >> 4, ec = el & 15;                             // read-back: row er (+4i), chunk ec
        const unsigned epi_base = lds_addr_w(smem + WEPI) + wid * 4096;
        const unsigned c_lane = (unsigned)(er * p.ldc + ec * 8) * 2u;
        const unsigned epi_w = epi_base + erow * 256 + ((((eq >> 1) ^ erow) & 15) << 4) + (eq & 1) * 8;
        const unsigned epi_r = epi_base + er * 256 + ((ec ^ er) << 4);
        const unsigned bias_r = lds_addr_w(smem + WBIAS) + (wc * 128 + eq * 4) * 4;
        int row0, col0;
        tile_of(idx, row0, col0);
        char* const c_tile = (char*)p.C + ((size_t)(row0 + wr * 128) * p.ldc + col0 + wc * 128) * 2;
        u32x4 bias[8];
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) bias[ni] = lds_read_b128w(bias_r + (seq & 1) * 1024 + ni * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                // explicit AGPR reads, in pass order: left to itself hipcc hoists all 256 of them and spills
                f32x4 v;
                asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\t"
                             "v_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                             : "a"(acc[mi][ni][0]), "a"(acc[mi][ni][1]), "a"(acc[mi][ni][2]), "a"(acc[mi][ni][3]));
                v[0] += __uint_as_float(bias[ni][0]); v[1] += __uint_as_float(bias[ni][1]);
                v[2] += __uint_as_float(bias[ni][2]); v[3] += __uint_as_float(bias[ni][3]);
                if constexpr (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                }
                u32x2 o;
                o[0] = pack_bf16x2(v[0], v[1]);
                o[1] = pack_bf16x2(v[2], v[3]);
                lds_write_b64w(epi_w ^ (ni * 32), o);
            }
            u32x4 d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = lds_read_b128w((epi_r ^ (i << 6)) + i * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) :: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4* dst = (u32x4*)(c_tile + (size_t)(mi * 16 + i * 4) * p.ldc * 2 + c_lane);
                if (p.dbg & 1) {}                       // timing experiment: no stores (vmcnt counts then over-wait; harmless)
                else __builtin_nontemporal_store(d[i], dst);
            }
        }
        had_stores = !(p.dbg & 1);
    }
}

template <int EPI>
static int launch256w(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm256w_bf16_nt_kernel<EPI>;
    static bool attr_done = false;
    static int num_cu = 0;
    if (!attr_done) {
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WSMEM));
        int dev = 0;
        KEMR_CHECK_HIP(hipGetDevice(&dev));
        KEMR_CHECK_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_done = true;
    }
    if (p.K < 128) KEMR_FAIL(KEMR_ERR_INVALID, "gemm256w: K must be >= 128");
    const int tiles = ((p.M + 255) / 256) * (p.N / 256);
    const int grid = tiles < num_cu ? tiles : num_cu;
    GemmParams q = p;
    q.dbg = g_gemm_dbg;
    ProfScope prof(PROF_GEMM, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), WSMEM, stream, q);
    KEMR_CHECK_LAUNCH("gemm256w_bf16_nt_kernel");
    return KEMR_OK;
}

// C must have ceil256(M) rows: rows in [M, ceil256(M)) are written (with values computed from A's pad rows).
int launch_gemm256w(const GemmParams& p, int epi, hipStream_t stream) {
    switch (epi) {
        case EPI_BIAS_BF16:       return launch256w<EPI_BIAS_BF16>(p, stream);
        case EPI_BIAS_QGELU_BF16: return launch256w<EPI_BIAS_QGELU_BF16>(p, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "gemm256w: epilogue %d is not a bf16-store epilogue", epi);
}

}  // namespace kemr
