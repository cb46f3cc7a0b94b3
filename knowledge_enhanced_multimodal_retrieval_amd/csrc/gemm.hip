// bf16 GEMM  C[M,N] = A[M,K] . W[N,K]^T  (+ fused epilogues) for the CLIP encoder towers, gfx950 only.
//
// Both operands are K-contiguous (activations row-major, weights in PyTorch Linear [out,in] layout), so
// the A and W tiles are staged identically: global -> LDS with `global_load_lds_dwordx4` (16 B / lane,
// no VGPR round trip), LDS rows of 64 bf16 (128 B) with a 16-byte-chunk XOR swizzle
//     physical_chunk = chunk ^ ((row >> 1) & 7)
// applied on the SOURCE address (the LDS image of an LDS-DMA is lane-linear) and again on the
// ds_read_b128 side, which makes every 16-lane ds_read_b128 group hit 16 distinct 16-byte slots.
// MFMA: v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand and the activation fragment
// as the B operand, i.e. each wave computes C^T tiles: a lane then owns 4 consecutive output columns of
// one output row and the epilogue stores 8 B (bf16) / 16 B (fp32) per lane.
//
// v1 structure ("minimum 2-phase" of cdna_hip_programming.md T3+T4): double-buffered LDS, the loads of
// K-tile t+1 are issued before the MFMAs of K-tile t, one vmcnt(0)+barrier per K-tile.
#include "common.h"
#include <cstdlib>

namespace kemr {

constexpr int BK = 64;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    // LDS destination = wave-uniform base + lane * 16 (hardware rule); gsrc is per lane.
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_nt_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;     // per-wave output tile
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;   // 1 KiB LDS-DMA pieces per wave
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WN, wc = wid % WN;

    // Tile id: blocks with equal blockIdx % 8 share an XCD (observed round-robin placement; speed only).
    // Give every XCD a contiguous range of tile ids so that neighbours (same A row panel) share its L2.
    const int tiles_n = p.N / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int row0 = tile_m * BM, col0 = tile_n * BN;

    const int srow = lane >> 3;                      // row inside an 8-row LDS-DMA piece
    const int schunk = lane & 7;                     // physical 16-byte chunk this lane fills
    const bf16_t* gA = p.A + (size_t)row0 * p.lda;
    const bf16_t* gW = p.W + (size_t)col0 * p.ldw;

    auto stage = [&](int buf, int kt) {
        char* sA = smem + buf * STAGE_BYTES;
        char* sB = sA + A_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
            const int piece = wid * A_INSTR + j;
            const int r = piece * 8 + srow;
            const int c = schunk ^ ((r >> 1) & 7);
            glds16(gA + (size_t)r * p.lda + k0 + c * 8, sA + piece * 1024);
        }
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) {
            const int piece = wid * B_INSTR + j;
            const int r = piece * 8 + srow;
            const int c = schunk ^ ((r >> 1) & 7);
            glds16(gW + (size_t)r * p.ldw + k0 + c * 8, sB + piece * 1024);
        }
    };

    const int lrow = lane & 15, lq = lane >> 4;
    const int swz = lrow >> 1;                        // ((row >> 1) & 7) for row = 16 * i + lrow
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nt) stage(cur ^ 1, kt + 1);
        const char* sA = smem + cur * STAGE_BYTES + (wr * WTM + lrow) * 128;
        const char* sB = smem + cur * STAGE_BYTES + A_BYTES + (wc * WTN + lrow) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + lq) ^ swz) << 4;
            bf16x8 af[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = *(const bf16x8*)(sA + mi * 16 * 128 + coff);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const bf16x8*)(sB + ni * 16 * 128 + coff);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // Epilogue.  acc[mi][ni][r] = C[row0 + wr*WTM + mi*16 + lrow][col0 + wc*WTN + ni*16 + lq*4 + r]
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = row0 + wr * WTM + mi * 16 + lrow;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = col0 + wc * WTN + ni * 16 + lq * 4;
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
            } else {  // EPI_PATCH_F32: token row = image * (patches + 1) + 1 + patch, plus positional embedding
                const int img = m / p.patches, pi = m - img * p.patches;
                const float4 pe = *(const float4*)(p.pos + (size_t)(pi + 1) * p.N + n);
                float4 x;
                x.x = v[0] + pe.x; x.y = v[1] + pe.y; x.z = v[2] + pe.z; x.w = v[3] + pe.w;
                *(float4*)((float*)p.C + (size_t)(m + img + 1) * p.ldc + n) = x;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI>
static int launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 2 * (BM + BN) * BK * 2;
    auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, EPI>;
    static bool attr_done = false;   // idempotent; a race only repeats the call
    if (!attr_done) {
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    ProfScope prof(PROF_GEMM, stream);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), smem, stream, p);
    KEMR_CHECK_LAUNCH("gemm_bf16_nt_kernel");
    return KEMR_OK;
}

int g_gemm_variant = 0;
int g_gemm_dbg = 0;
int g_gemm_order = 3;
int g_gemm_conc = 2;
int g_gemm_grid = 0;       // tools: > 0 caps the persistent GEMM's grid (workgroups = CUs it may take); 0 = every CU
// gemm256u K loop: 0 = eight 256-cycle barrier intervals per K-tile (round 2; the default), 1 = four of 512 (round 3 experiment:
// half the hand-overs, but the W pieces of a K-tile then have ONE interval of flight before the wait that needs them -- their
// region is read until two intervals before -- and the stall eats the gain: 402 -> 414 us on fc2, 106 -> 111 on out-proj, same
// device, bit-identical results; tools/bench_gemm_r3.py)
int g_gemm_kl = 0;

int launch_gemm(const GemmParams& p, int epi, hipStream_t stream) {
    if (p.M <= 0) return KEMR_OK;
    if (p.N % 128 != 0 || p.K % BK != 0 || p.K <= 0)
        KEMR_FAIL(KEMR_ERR_INVALID, "gemm: need N %% 128 == 0 and K %% 64 == 0 (got M=%d N=%d K=%d)", p.M, p.N, p.K);
    if ((p.lda % 8) || (p.ldw % 8) || (p.ldc % 4))
        KEMR_FAIL(KEMR_ERR_INVALID, "gemm: leading dimensions must keep 16-byte alignment");
    // 256x256 tiles (one workgroup per CU, deep LDS-DMA pipeline) once there is at least ~half a wave of them
    const bool can256 = p.N % 256 == 0 && p.K >= 128;
    const long tiles256 = (long)((p.M + 255) / 256) * (p.N / 256);
    const bool bf16_epi = epi == EPI_BIAS_BF16 || epi == EPI_BIAS_QGELU_BF16;
    if (epi == EPI_BIAS_RESADD_BF16) {        // read-modify-write of the bf16 residual stream: only the persistent kernel has it
        if (!(can256 && p.c_rows_padded && p.M > 512 && gemm256u_fits(p, 2)))
            KEMR_FAIL(KEMR_ERR_INVALID, "gemm: the residual-add epilogue needs N %% 256 == 0, more than 512 rows and a row-padded C (M=%d N=%d K=%d)", p.M, p.N, p.K);
        return launch_gemm256u(p, epi, stream);
    }
    // fp32 residual stream updated in place: the persistent kernel where it has whole rounds of work (the towers' out-proj / fc2
    // at more than 512 rows), the earlier kernels below otherwise (same arithmetic up to the position of the bias in the sum)
    if (epi == EPI_BIAS_RESID_F32 && can256 && p.c_rows_padded && p.M > 512 && (g_gemm_variant == 0 || g_gemm_variant == 7) &&
        (g_gemm_variant == 7 || tiles256 >= 128) && gemm256u_fits(p, 2, 4))
        return launch_gemm256u(p, epi, stream);
    // a handful of rows (one or a few online queries): 6-24 tiles would leave the chip idle; split K inside the workgroup
    if (bf16_epi && p.c_rows_padded && (g_gemm_variant == 8 || (g_gemm_variant == 0 && p.M <= 512))) return launch_gemm_skinny(p, epi, stream);
#ifdef KEMR_AB_VARIANTS      // earlier persistent generations, A/B timing from tools/ only (build.py --ab-variants)
    if (can256 && bf16_epi && p.c_rows_padded && g_gemm_variant == 5) return launch_gemm256q(p, epi, stream);
    if (can256 && bf16_epi && p.c_rows_padded && g_gemm_variant == 6) return launch_gemm256w(p, epi, stream);
    if (can256 && bf16_epi && p.c_rows_padded && g_gemm_variant == 9) return launch_gemm256r(p, epi, stream);
    if (can256 && bf16_epi && p.c_rows_padded && g_gemm_variant == 4) return launch_gemm256p(p, epi, stream);
#else
    if (g_gemm_variant == 4 || g_gemm_variant == 5 || g_gemm_variant == 6 || g_gemm_variant == 9)
        KEMR_FAIL(KEMR_ERR_INVALID, "gemm: variant %d is an A/B kernel that this library was built without (build.py --ab-variants)", g_gemm_variant);
#endif
    // A ragged last row tile that would cost the persistent kernel one more round over all its workgroups (64 images are
    // 64 x 256 + 64 token rows: 260 tiles of an N = 1024 GEMM on 256 CUs) goes to the skinny kernel instead: measured 4 406 ->
    // 5 320 images/s at 64 images per call (5 650 at 63).
    if (can256 && bf16_epi && p.c_rows_padded && g_gemm_variant == 0 && p.M > 512 && gemm256u_fits(p, 2)) {
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            KEMR_CHECK_HIP(hipGetDevice(&dev));
            KEMR_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        const int rem = p.M & 255, full = p.M - rem;
        const long tn = p.N / 256, tiles_full = (long)(full / 256) * tn;
        if (rem > 0 && tiles_full >= 128 && (tiles_full + tn + ncu - 1) / ncu > (tiles_full + ncu - 1) / ncu) {
            GemmParams a = p, b = p;
            a.M = full;
            b.M = rem;
            b.A = p.A + (size_t)full * p.lda;
            b.C = (bf16_t*)p.C + (size_t)full * p.ldc;
            KEMR_TRY(launch_gemm256u(a, epi, stream));
            return launch_gemm_skinny(b, epi, stream);
        }
    }
    if (can256 && bf16_epi && p.c_rows_padded && (g_gemm_variant == 7 || (g_gemm_variant == 0 && tiles256 >= 128)) && gemm256u_fits(p, 2))
        return launch_gemm256u(p, epi, stream);
    if (can256 && (g_gemm_variant >= 2 || (g_gemm_variant == 0 && tiles256 >= 128))) return launch_gemm256(p, epi, stream);
    switch (epi) {
        case EPI_BIAS_BF16:       return launch_cfg<128, 128, 2, 2, EPI_BIAS_BF16>(p, stream);
        case EPI_BIAS_QGELU_BF16: return launch_cfg<128, 128, 2, 2, EPI_BIAS_QGELU_BF16>(p, stream);
        case EPI_BIAS_RESID_F32:  return launch_cfg<128, 128, 2, 2, EPI_BIAS_RESID_F32>(p, stream);
        case EPI_PATCH_F32:       return launch_cfg<128, 128, 2, 2, EPI_PATCH_F32>(p, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "gemm: unknown epilogue %d", epi);
}

}  // namespace kemr
