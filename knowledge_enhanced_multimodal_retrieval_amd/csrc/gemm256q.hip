// Persistent bf16 GEMM, variant "q": same tile hand-over and LDS-staged asynchronous epilogue as gemm256p.hip, but the K
// loop has TWO long phases per K-tile instead of four (4 barriers instead of 8, 32-MFMA clusters of 512 cycles):
//   RA  16 ds_read_b128: B(cols 0-31), B(cols 32-63), A(rows 0-63); stage tile t+1: own A half (4 LDS-DMA) + B0 + B1
//   MA  quadrants (0,0) and (0,1)
//   RB   8 ds_read_b128: A(rows 64-127)
//   MB  quadrants (1,1) and (1,0)
// The wr == 1 half of the workgroup runs one barrier interval behind the wr == 0 half (intervals g0/g1: RA 0/1, MA 1/2,
// RB 2/3, MB 3/4), so a SIMD's two waves alternate a 512-cycle MFMA cluster with the partner's LDS reads.
// Staging ownership: an A half is read only by its own wave group, so that group alone stages it (4 pieces per wave)
// right at its RA -- after its own last read of the slot (RB of the previous tile) in program order; the B halves of
// tile t-1 were last read at RA(t-1) (waits at the start of intervals 1/2) and are refilled from interval 4/5 on.
// Everything of tile t+1 is issued in interval 0/1 of tile t and waited for at the end of interval 3
// (g0 after MB, g1 after RB): three intervals (~1500 cycles) of flight time.
#include "common.h"

namespace kemr {

namespace {

constexpr int PBUF = 65536;      // bytes per K-tile buffer
constexpr int PHALF = 16384;     // bytes per half-tile
constexpr int PEPI = 131072;     // offset of the epilogue area (8 waves x 2 KiB)
constexpr int PBIAS = PEPI + 16384;   // 2 x 1 KiB: fp32 bias of the current / next tile's 256 columns
constexpr int PSMEM = PBIAS + 2048;

__device__ __forceinline__ void glds16q(const void* gsrc, void* lds_wave_base) {
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

}  // namespace

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256q_bf16_nt_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    const int tiles_n = p.N >> 8;
    const int ntiles = ((p.M + 255) >> 8) * tiles_n;
    const int full = (ntiles / (int)gridDim.x) * (int)gridDim.x;     // tiles inside complete rounds
    // logical tile of a work index: inside complete rounds every XCD (blocks b, b+8, ... by observed round-robin
    // placement; speed only) gets a contiguous run of tiles so that its L2 sees few A / W panels at a time
    auto tile_of = [&](int idx, int& row0, int& col0) {
        int L = idx;
        if (idx < full && (gridDim.x & 7) == 0) {
            const int rnd = idx / (int)gridDim.x, b = idx - rnd * (int)gridDim.x;
            L = rnd * (int)gridDim.x + (b & 7) * ((int)gridDim.x >> 3) + (b >> 3);
        }
        const int tm = L / tiles_n;
        row0 = tm << 8;
        col0 = (L - tm * tiles_n) << 8;
    };

    // staging addresses = wave-uniform tile base (SGPRs) + a per-lane 32-bit byte offset that never changes.
    // A: the wave stages pieces 4*(wid&3) .. +3 of ITS group's half; B: pieces 2*wid, 2*wid+1 of both halves.
    const int srow = lane >> 3, schunk = lane & 7;
    unsigned a_lane[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = ((wid & 3) * 4 + j) * 8 + srow;
        a_lane[j] = (unsigned)(r * p.lda + ((schunk ^ ((r >> 1) & 7)) << 3)) * 2u;
    }
    const int r0 = wid * 16 + srow, r1 = r0 + 8;
    const unsigned w_lane0 = (unsigned)(r0 * p.ldw + ((schunk ^ ((r0 >> 1) & 7)) << 3)) * 2u;
    const unsigned w_lane1 = (unsigned)(r1 * p.ldw + ((schunk ^ ((r1 >> 1) & 7)) << 3)) * 2u;
    const size_t a_half = (size_t)256 * p.lda, w_half = (size_t)256 * p.ldw;      // bytes between the two half-tiles

    const char *a_tile, *w_tile;                        // wave-uniform
    auto set_src = [&](int row0, int col0) {
        a_tile = (const char*)p.A + (size_t)row0 * p.lda * 2 + wr * a_half;      // this group's A half
        w_tile = (const char*)p.W + (size_t)col0 * p.ldw * 2;
    };
    auto stage_a_own = [&](int tau) {                   // 4 LDS-DMA: the whole A half of this wave group
        char* dst = smem + (tau & 1) * PBUF + wr * PHALF + (wid & 3) * 4096;
        const char* src = a_tile + tau * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16q(src + a_lane[j], dst + j * 1024);
    };
    auto stage_w = [&](int half, int tau) {
        char* dst = smem + wid * 2048 + (tau & 1) * PBUF + (2 + half) * PHALF;
        const char* src = w_tile + half * w_half + tau * 128;
        glds16q(src + w_lane0, dst);
        glds16q(src + w_lane1, dst + 1024);
    };
    auto stage_tile = [&](int tau) { stage_a_own(tau); stage_w(0, tau); stage_w(1, tau); };   // 8 LDS-DMA
    auto prologue = [&](int col0_, int parity) {   // K-tiles 0 and 1 complete: 16 LDS-DMA per lane (nt >= 2)
        if (wid == 0 && p.bias) glds16q(p.bias + col0_ + lane * 4, smem + PBIAS + parity * 1024);
        stage_tile(0);
        stage_tile(1);
    };

    const int lrow = lane & 15, lq = lane >> 4;
    const int swz = lrow >> 1;
    const int co0 = ((0 + lq) ^ swz) << 4, co1 = ((4 + lq) ^ swz) << 4;
    const int a_off = wr * PHALF + lrow * 128;
    const int b_off = 2 * PHALF + (wc >> 1) * PHALF + ((wc & 1) * 64 + lrow) * 128;
    const int nt = p.K >> 6;

    // epilogue addressing (wave-private 2 KiB: 16 rows x 128 B, 16-byte chunk ^= row & 7)
    char* const epi = smem + PEPI + wid * 2048;
    const int er = lane >> 3, ec = lane & 7;                          // read-back: row er (+8i), chunk ec
    const unsigned c_lane = (unsigned)(er * p.ldc + ec * 8) * 2u;     // per-lane byte offset inside the C tile
    const unsigned epi_w = lds_addr(epi) + lrow * 128 + (((lq >> 1) ^ (lrow & 7)) << 4) + (lq & 1) * 8;
    const unsigned epi_r0 = lds_addr(epi) + er * 128 + ((ec ^ er) << 4);               // rows er and er + 8:
    const unsigned epi_r1 = epi_r0 + 1024;                                           // (er + 8) & 7 == er

    if (!p.bias && tid < 128) *(float4*)(smem + PBIAS + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    int row0, col0;
    tile_of(blockIdx.x, row0, col0);
    set_src(row0, col0);
    prologue(col0, 0);
    bool first = true;
    int parity = 0;

    for (int idx = blockIdx.x; idx < ntiles; idx += gridDim.x) {
        f32x4 acc[8][4];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                // materialise the zero HERE: hipcc cannot see that the asm below is an MFMA and otherwise sinks the
                // v_mov right in front of the first use, inside the VALU-write -> MFMA-SrcC hazard window (observed:
                // accumulator elements 2,3 of one fragment started from stale register contents)
                asm volatile("" : "+v"(acc[mi][ni]));
            }

        // K-tile 0 (and everything older) landed; K-tile 1 (8 DMA) and, after the first tile, the 16 stores may fly
        const bool had_stores = !first;
        if (first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        if (p.dbg & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // bisecting aid: drain everything
        first = false;
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();

        bf16x8 af[4][2], w0[2][2], w1[2][2];
        for (int t = 0; t < nt; ++t) {
            const char* sa = smem + (t & 1) * PBUF + a_off;
            const char* sb = smem + (t & 1) * PBUF + b_off;
            // ---- RA
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                w0[ni][0] = *(const bf16x8*)(sb + ni * 2048 + co0);
                w0[ni][1] = *(const bf16x8*)(sb + ni * 2048 + co1);
                w1[ni][0] = *(const bf16x8*)(sb + 4096 + ni * 2048 + co0);
                w1[ni][1] = *(const bf16x8*)(sb + 4096 + ni * 2048 + co1);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                af[mi][0] = *(const bf16x8*)(sa + mi * 2048 + co0);
                af[mi][1] = *(const bf16x8*)(sa + mi * 2048 + co1);
            }
            if (t > 0 && t + 1 < nt) stage_tile(t + 1);          // K-tile 1 came with the hand-over prologue
            __builtin_amdgcn_s_barrier();
            // ---- MA
            quad<0, 0>(acc, af, w0);
            quad<0, 1>(acc, af, w1);
            __builtin_amdgcn_s_barrier();
            // ---- RB
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                af[mi][0] = *(const bf16x8*)(sa + 8192 + mi * 2048 + co0);
                af[mi][1] = *(const bf16x8*)(sa + 8192 + mi * 2048 + co1);
            }
            auto close_tile = [&]() {      // tile t+1 landed; only the previous tile's stores may still fly (t == 0)
                if (p.dbg & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (t == 0 && had_stores) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            };
            if (wr == 1) close_tile();
            __builtin_amdgcn_s_barrier();
            // ---- MB
            quad<1, 1>(acc, af, w1);
            quad<1, 0>(acc, af, w0);
            if (wr == 0) close_tile();
            __builtin_amdgcn_s_barrier();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();      // both halves are past their last LDS read: K buffers are free
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMA result -> VALU read (>= 12 wait states)

        // ---- hand-over: the next tile's first loads, then this tile's stores (bias comes from LDS)
        const int ccol = col0 + wc * 64;
        const int crow = row0 + wr * 128;
        // bias of THIS tile out of LDS before the next tile's LDS-DMA is issued: hipcc guards a ds_read of a DMA-written
        // LDS range with vmcnt(0); here nothing is in flight yet, after the prologue it would drain the prefetch
        const float* sbias = (const float*)(smem + PBIAS + parity * 1024) + wc * 64 + lq * 4;
        float4 bias[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bias[ni] = *(const float4*)(sbias + ni * 16);
        const int nidx = idx + gridDim.x;
        if (nidx < ntiles) {
            tile_of(nidx, row0, col0);
            set_src(row0, col0);
            prologue(col0, parity ^ 1);
        }
        parity ^= 1;
        char* const c_tile = (char*)p.C + ((size_t)crow * p.ldc + ccol) * 2;      // wave-uniform; lanes add c_lane
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {                 // 8 passes of 16 rows through the wave's private LDS area
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 v = acc[mi][ni];
                v[0] += bias[ni].x; v[1] += bias[ni].y; v[2] += bias[ni].z; v[3] += bias[ni].w;
                if constexpr (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                }
                u32x2 o;
                o[0] = pack_bf16x2(v[0], v[1]);
                o[1] = pack_bf16x2(v[2], v[3]);
                lds_write_b64(epi_w ^ (ni * 32), o);      // chunk (ni*2 + (lq>>1)) ^ (lrow & 7): ni only flips bits 5-6
            }
            u32x4 d0 = lds_read_b128(epi_r0), d1 = lds_read_b128(epi_r1);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1) :: "memory");
            if (!(p.dbg & 1)) {
                *(u32x4*)(c_tile + (size_t)(mi * 16) * p.ldc * 2 + c_lane) = d0;
                *(u32x4*)(c_tile + (size_t)(mi * 16 + 8) * p.ldc * 2 + c_lane) = d1;
            }
        }
    }
}

template <int EPI>
static int launch256q(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm256q_bf16_nt_kernel<EPI>;
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
    ProfScope prof(PROF_GEMM, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), PSMEM, stream, q);
    KEMR_CHECK_LAUNCH("gemm256q_bf16_nt_kernel");
    return KEMR_OK;
}

// C must have ceil256(M) rows: rows in [M, ceil256(M)) are written (with values computed from A's pad rows).
int launch_gemm256q(const GemmParams& p, int epi, hipStream_t stream) {
    switch (epi) {
        case EPI_BIAS_BF16:       return launch256q<EPI_BIAS_BF16>(p, stream);
        case EPI_BIAS_QGELU_BF16: return launch256q<EPI_BIAS_QGELU_BF16>(p, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "gemm256p: epilogue %d is not a bf16-store epilogue", epi);
}

}  // namespace kemr
