// Persistent bf16 GEMM, 4 waves x (128 x 128) per 256 x 256 tile, REGISTER-staged operands: the answer to what gemm256w.hip
// measured.  One wave per SIMD (512 registers: 256 AGPR accumulators) has no partner wave to hide the 60-100 issue cycles of
// every LDS-DMA piece, so here the K-tile travels global -> VGPR (global_load_dwordx4, saddr form) -> LDS (ds_write_b128),
// the way the vendor's 256-thread kernel does it.  Because no LDS-DMA exists in the kernel, hipcc's own s_waitcnt
// insertion is exact for every load, LDS access and store: there is no hand-counted vmcnt here, only the order in which
// the instructions are laid between the volatile-asm MFMAs.
//
// Pipeline (global K-tile counter g per workgroup, running across output tiles; LDS buffer = g & 1, 64 KiB each;
// R = 64 VGPRs holding one K-tile slice of this wave: 8 x 16 B of A rows and 8 x 16 B of W rows):
//   top:      lgkmcnt(0) ; barrier            [every wave's ds_writes of K-tile g are in buffer g & 1;
//                                              every wave is done reading buffer (g+1) & 1]
//             16 ds_read_b128: k 0..31 fragments
//   phase 1:  64 MFMA (k 0..31); the 16 reads of the k 32..63 fragments ride in the first 16 slots
//   phase 2:  64 MFMA (k 32..63); slots 0..31 carry the 16 ds_write_b128 of R (K-tile g+1) into buffer (g+1) & 1,
//             slots 32..63 the 16 global loads of K-tile g+2 into R -- a K-tile's loads are in flight for a whole K-tile
// One barrier per K-tile.  The epilogue (accumulators -> bias / QuickGELU -> bf16 -> wave-private LDS transpose -> 16-byte
// non-temporal stores) runs between two K-tiles with R in flight.
#include "common.h"

namespace kemr {

namespace {

constexpr int RBUF = 65536;            // bytes per K-tile buffer: A rows 0..255 (32 KiB) | W rows 0..255 (32 KiB)
constexpr int RHALF = 32768;
constexpr int REPI = 131072;           // 4 waves x 4 KiB epilogue staging (16 rows x 256 B)
constexpr int RSMEM = REPI + 16384;

typedef __attribute__((ext_vector_type(2))) unsigned r_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned r_u32x4;

__device__ __forceinline__ void mfma_r(f32x4& acc, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}

}  // namespace

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm256r_bf16_nt_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;

    const int tiles_n = p.N >> 8;
    const int ntiles = ((p.M + 255) >> 8) * tiles_n;
    const int full = (ntiles / (int)gridDim.x) * (int)gridDim.x;
    auto tile_of = [&](int idx, int& row0, int& col0) {      // same XCD-contiguous order as gemm256u.hip
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

    // ---- staging: wave w moves rows w*64 .. w*64+63 of the A and of the W tile; load i covers rows w*64 + i*8 + (lane>>3),
    // 16-byte chunk (lane & 7) -> 128 contiguous bytes per row.  LDS row = 128 B; physical chunk c of row r holds logical
    // chunk c ^ ((r >> 1) & 7): the ds_write goes to chunk (lane & 7) ^ key, key = (lane >> 4) for even i, + 4 for odd i.
    const int srow = lane >> 3, schunk = lane & 7;
    const unsigned a_ln = (unsigned)(srow * p.lda + schunk * 8) * 2u;
    const unsigned w_ln = (unsigned)(srow * p.ldw + schunk * 8) * 2u;
    const size_t a_step = (size_t)8 * p.lda * 2, w_step = (size_t)8 * p.ldw * 2;
    const int wl_e = (wid * 64 + srow) * 128 + ((schunk ^ (srow >> 1)) << 4);              // LDS byte offset, even i (+ i*1024)
    const int wl_o = (wid * 64 + srow) * 128 + ((schunk ^ ((srow >> 1) + 4)) << 4);        // odd i

    int s_idx = blockIdx.x, s_tau = 0;                   // load cursor (wave-uniform): K-tile to load next
    const char *s_a = nullptr, *s_w = nullptr;
    auto s_set = [&]() {
        int r0, c0;
        tile_of(s_idx, r0, c0);
        s_a = (const char*)p.A + ((size_t)r0 + wid * 64) * p.lda * 2;
        s_w = (const char*)p.W + ((size_t)c0 + wid * 64) * p.ldw * 2;
    };
    auto s_advance = [&]() {
        if (++s_tau == nt) {
            s_tau = 0;
            s_idx += gridDim.x;
            if (s_idx < ntiles) s_set();
        }
    };
    r_u32x4 R[16];                                       // 0..7: A rows, 8..15: W rows
    auto load_one = [&](int j) {
        const int i = j & 7;
        if (j < 8) R[j] = *(const r_u32x4*)(s_a + i * a_step + s_tau * 128 + a_ln);
        else R[j] = *(const r_u32x4*)(s_w + i * w_step + s_tau * 128 + w_ln);
    };
    auto write_one = [&](int j, int buf) {
        const int i = j & 7;
        *(r_u32x4*)(smem + buf * RBUF + (j < 8 ? 0 : RHALF) + i * 1024 + ((i & 1) ? wl_o : wl_e)) = R[j];
    };

    // ---- fragment addressing (MFMA 16x16x32: lane (lrow, lq) reads 8 consecutive k of row lrow at k = 32*kk + 8*lq)
    const int lrow = lane & 15, lq = lane >> 4;
    const int swz = lrow >> 1;
    const int co0 = ((0 + lq) ^ swz) << 4, co1 = ((4 + lq) ^ swz) << 4;
    const int a_off = (wr * 128 + lrow) * 128;
    const int b_off = RHALF + (wc * 128 + lrow) * 128;

    // prologue: K-tile 0 -> R -> buffer 0; K-tile 1 -> R (in flight)      (g_total >= 2 because nt >= 2)
    s_set();
#pragma unroll
    for (int j = 0; j < 16; ++j) load_one(j);
    s_advance();
#pragma unroll
    for (int j = 0; j < 16; ++j) write_one(j, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) load_one(j);
    s_advance();
    int g_loaded = 2;

    int g = 0, par = 0;
    bf16x8 a0[8], b0[8], a1[8], b1[8];

    for (int idx = blockIdx.x; idx < ntiles; idx += gridDim.x) {
        f32x4 acc[8][8];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+a"(acc[mi][ni]));     // materialise the zeros here, not in front of the first MFMA
            }

        for (int t = 0; t < nt; ++t, ++g, par ^= 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* sa = smem + par * RBUF + a_off;
            const char* sb = smem + par * RBUF + b_off;
            a0[0] = *(const bf16x8*)(sa + co0);
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) b0[ni] = *(const bf16x8*)(sb + ni * 2048 + co0);
#pragma unroll
            for (int mi = 1; mi < 8; ++mi) a0[mi] = *(const bf16x8*)(sa + mi * 2048 + co0);
            const bool do_write = g + 1 < g_total;         // R holds K-tile g+1
            const bool do_load = g_loaded < g_total;       // K-tile g+2

            asm volatile("s_nop 1" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
            // ---- phase 1
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                mfma_r(acc[s >> 3][s & 7], b0[s & 7], a0[s >> 3]);
                if (s == 0) a1[0] = *(const bf16x8*)(sa + co1);
                else if (s <= 8) b1[s - 1] = *(const bf16x8*)(sb + (s - 1) * 2048 + co1);
                else if (s <= 15) a1[s - 8] = *(const bf16x8*)(sa + (s - 8) * 2048 + co1);
            }
            // ---- phase 2
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                mfma_r(acc[s >> 3][s & 7], b1[s & 7], a1[s >> 3]);
                if (s < 32 && (s & 1) == 0 && do_write) write_one(s >> 1, par ^ 1);
                if (s >= 32 && (s & 1) == 0 && do_load) load_one((s - 32) >> 1);
            }
            __builtin_amdgcn_s_setprio(0);
            if (do_load) { s_advance(); ++g_loaded; }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMA result -> accvgpr read

        // ---- epilogue: 8 passes of 16 rows x 128 columns through the wave's private LDS area (4 KiB = 16 rows x 256 B,
        // 16-byte chunk index ^= row).  Lane constants behind an opaque copy of `lane` (recomputed here, not kept live).
        int el = lane;
        asm volatile("" : "+v"(el));
        const int erow = el & 15, eq = el >> 4;                           // MFMA layout: row (m), 4-column group
        const int er = el >> 4, ec = el & 15;                             // read-back: row er (+4i), chunk ec
        char* const epi = smem + REPI + wid * 4096;
        const unsigned c_lane = (unsigned)(er * p.ldc + ec * 8) * 2u;
        const int epi_w = erow * 256 + ((((eq >> 1) ^ erow) & 15) << 4) + (eq & 1) * 8;
        const int epi_r = er * 256 + ((ec ^ er) << 4);
        int row0, col0;
        tile_of(idx, row0, col0);
        char* const c_tile = (char*)p.C + ((size_t)(row0 + wr * 128) * p.ldc + col0 + wc * 128) * 2;
        float4 bias[8];
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
            bias[ni] = p.bias ? *(const float4*)(p.bias + col0 + wc * 128 + ni * 16 + eq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
                v[0] += bias[ni].x; v[1] += bias[ni].y; v[2] += bias[ni].z; v[3] += bias[ni].w;
                if constexpr (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                }
                r_u32x2 o;
                o[0] = pack_bf16x2(v[0], v[1]);
                o[1] = pack_bf16x2(v[2], v[3]);
                *(r_u32x2*)(epi + (epi_w ^ (ni * 32))) = o;
            }
            // wave-private area: in-order LDS pipe, no barrier; the compiler waits for the writes before the reads' data is used
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const r_u32x4 d = *(const r_u32x4*)(epi + ((epi_r ^ (i << 6)) + i * 1024));
                __builtin_nontemporal_store(d, (r_u32x4*)(c_tile + (size_t)(mi * 16 + i * 4) * p.ldc * 2 + c_lane));
            }
        }
    }
}

template <int EPI>
static int launch256r(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm256r_bf16_nt_kernel<EPI>;
    static bool attr_done = false;
    static int num_cu = 0;
    if (!attr_done) {
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, RSMEM));
        int dev = 0;
        KEMR_CHECK_HIP(hipGetDevice(&dev));
        KEMR_CHECK_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_done = true;
    }
    if (p.K < 128) KEMR_FAIL(KEMR_ERR_INVALID, "gemm256r: K must be >= 128");
    const int tiles = ((p.M + 255) / 256) * (p.N / 256);
    const int grid = tiles < num_cu ? tiles : num_cu;
    ProfScope prof(PROF_GEMM, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), RSMEM, stream, p);
    KEMR_CHECK_LAUNCH("gemm256r_bf16_nt_kernel");
    return KEMR_OK;
}

// C must have ceil256(M) rows: rows in [M, ceil256(M)) are written (with values computed from A's pad rows).
int launch_gemm256r(const GemmParams& p, int epi, hipStream_t stream) {
    switch (epi) {
        case EPI_BIAS_BF16:       return launch256r<EPI_BIAS_BF16>(p, stream);
        case EPI_BIAS_QGELU_BF16: return launch256r<EPI_BIAS_QGELU_BF16>(p, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "gemm256r: epilogue %d is not a bf16-store epilogue", epi);
}

}  // namespace kemr
