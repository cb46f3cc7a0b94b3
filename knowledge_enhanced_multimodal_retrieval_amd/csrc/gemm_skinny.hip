// Skinny-M bf16 GEMM for the online path (one or a few queries: M = 77..512 rows): C[M, N] = epi(A[M, K] . W[N, K]^T + bias).
// The tiled kernels give such a launch 6-24 workgroups (N / 128 tiles) and 18-50 us; here one 512-thread workgroup owns a
// 16-column slice of W for up to 128 rows of A and its 8 waves split K (k-step s of 32 goes to wave s mod 8), so N = 768
// already gives 48 workgroups and every wave has only K / 256 dependent steps.  Operands go straight from global memory
// (L2: A is a few hundred KB and is re-read by every workgroup) into MFMA fragments, no LDS staging; the 8 partial
// accumulators meet in LDS (64 KiB), wave w finishes m-tile w.
#include "common.h"

namespace kemr {

template <int EPI>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const GemmParams p) {
    __shared__ f32x4 red[8][8][64];                      // [wave][m-tile][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 128;
    const int rows = p.M - m0 < 128 ? p.M - m0 : 128;
    const int mt = (rows + 15) >> 4;                     // m-tiles of this pass (A and C have ceil256(M) rows: pad rows are read)
    const int steps = p.K >> 5;

    const bf16_t* wp = p.W + (size_t)(n0 + lrow) * p.ldw + lq * 8;
    const bf16_t* ap = p.A + (size_t)(m0 + lrow) * p.lda + lq * 8;
    f32x4 acc[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // two fragment sets in flight: the loads of step s + 8 are issued before the MFMAs of step s
    bf16x8 wfa, wfb, afa[8], afb[8];
    auto load = [&](int s, bf16x8& wf, bf16x8 (&af)[8]) {
        wf = *(const bf16x8*)(wp + s * 32);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
            if (mi < mt) af[mi] = *(const bf16x8*)(ap + (size_t)mi * 16 * p.lda + s * 32);
    };
    auto compute = [&](const bf16x8& wf, const bf16x8 (&af)[8]) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
            if (mi < mt) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[mi], acc[mi], 0, 0, 0);
    };
    int s = wid;
    if (s < steps) {
        load(s, wfa, afa);
        for (;;) {
            const int s1 = s + 8;
            if (s1 < steps) load(s1, wfb, afb);
            compute(wfa, afa);
            if (s1 >= steps) break;
            s = s1 + 8;
            if (s < steps) load(s, wfa, afa);
            compute(wfb, afb);
            if (s >= steps) break;
        }
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) red[wid][mi][lane] = acc[mi];
    __syncthreads();

    // wave w finishes m-tile w: lane holds C[m0 + 16 w + lrow][n0 + 4 lq .. + 3]
    if (wid < mt) {
        f32x4 v = red[0][wid][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const f32x4 t = red[w][wid][lane];
            v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
        const int row = m0 + wid * 16 + lrow, col = n0 + lq * 4;
        if (p.bias) {
            const float4 b = *(const float4*)(p.bias + col);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if constexpr (EPI == EPI_BIAS_QGELU_BF16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
        }
        if (row < p.M) {
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)((bf16_t*)p.C + (size_t)row * p.ldc + col) = o;
        }
    }
}

// bf16-store epilogues, N % 16 == 0, K % 32 == 0; A needs ceil16(M) readable rows (the callers give ceil256)
int launch_gemm_skinny(const GemmParams& p, int epi, hipStream_t stream) {
    if (p.M <= 0) return KEMR_OK;
    if (p.N % 16 != 0 || p.K % 32 != 0 || p.K <= 0) KEMR_FAIL(KEMR_ERR_INVALID, "gemm skinny: need N %% 16 == 0 and K %% 32 == 0 (got N=%d K=%d)", p.N, p.K);
    const dim3 grid(p.N / 16, (p.M + 127) / 128);
    ProfScope prof(PROF_GEMM, stream);
    switch (epi) {
        case EPI_BIAS_BF16:       hipLaunchKernelGGL(gemm_skinny_kernel<EPI_BIAS_BF16>, grid, dim3(512), 0, stream, p); break;
        case EPI_BIAS_QGELU_BF16: hipLaunchKernelGGL(gemm_skinny_kernel<EPI_BIAS_QGELU_BF16>, grid, dim3(512), 0, stream, p); break;
        default: KEMR_FAIL(KEMR_ERR_INVALID, "gemm skinny: epilogue %d is not a bf16-store epilogue", epi);
    }
    KEMR_CHECK_LAUNCH("gemm_skinny_kernel");
    return KEMR_OK;
}

}  // namespace kemr
