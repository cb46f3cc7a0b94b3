// Fused cosine-similarity + per-query top-k + rank-of-ground-truth; the Q x N score matrix never leaves the CU.
//
// Operands are bf16 "panels" [rows, kdim] (kemr_panel_build): fused T2I+T2T scoring concatenates the weighted
// embedding sets along k (one contraction instead of two GEMMs + an N^2 axpy, reference metrics.py:145-148),
// and terms == 3 appends the bf16 residuals so that hi*hi + lo*hi + hi*lo reproduces the fp32 products.
//
// sim_kernel: a 256-thread workgroup owns 128 queries and walks gallery tiles of 128 rows:
//   1. K loop exactly like gemm.hip (LDS-DMA staging, chunk-XOR swizzle, v_mfma_f32_16x16x32_bf16; gallery
//      fragment = A operand, query fragment = B operand) -> a 128 x 128 fp32 score tile in accumulators;
//   2. the tile is dumped to LDS as S[query][candidate] (row stride 132 floats: conflict-free both ways);
//   3. two scanner threads per query stream their 64 candidates: add the sparse SPARQL bonus (CSR cursor),
//      count candidates ranked ahead of the ground truth, and keep a sorted top-KMAX list in registers
//      (threshold test first, the insertion runs only on a hit).
// Per (query, gallery chunk) partial lists go to the workspace and are merged by topk_merge_kernel, the same
// kernel that merges per-GPU shards.  Order rule everywhere: higher score first, then lower candidate id.
#include "common.h"
#include "../../include/kemr_debug.h"
#include <cmath>
#include <vector>

namespace kemr {

constexpr int SBK = 64;
constexpr int ST = 128;          // tile edge (queries and candidates)
constexpr int SLD = 132;         // LDS row stride of the score tile in floats

__device__ __forceinline__ void glds16s(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------ panel build
struct PanelArgs {
    const float* parts[4];
    const float* row_scale[4];
    float part_scale[4];
    int nparts, rows, rows_alloc, d, dpad, terms, side;
    long long kdim;
};

__global__ __launch_bounds__(256) void panel_build_kernel(const PanelArgs a, bf16_t* __restrict__ out, long long total) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int half = a.dpad >> 1;
    const int i = (int)(gid % half) * 2;
    const long long t = gid / half;
    const int p = (int)(t % a.nparts);
    const long long row = t / a.nparts;
    float v[2] = {0.f, 0.f};
    if (row < a.rows) {
        float sc = a.part_scale[p];
        if (a.row_scale[p]) sc *= a.row_scale[p][row];
        const float* src = a.parts[p] + (size_t)row * a.d;
        if (i < a.d) v[0] = src[i] * sc;
        if (i + 1 < a.d) v[1] = src[i + 1] * sc;
    }
    const bf16_t h0 = f32_to_bf16(v[0]), h1 = f32_to_bf16(v[1]);
    const uint32_t hi = (uint32_t)h0 | ((uint32_t)h1 << 16);
    bf16_t* dst = out + (size_t)row * a.kdim + (size_t)p * a.terms * a.dpad + i;
    if (a.terms == 1) {
        *(uint32_t*)dst = hi;
    } else {
        const uint32_t lo = pack_bf16x2(v[0] - bf16_to_f32(h0), v[1] - bf16_to_f32(h1));
        // query panel [hi | lo | hi], gallery panel [hi | hi | lo]  ->  hi.hi + lo.hi + hi.lo
        *(uint32_t*)dst = hi;
        *(uint32_t*)(dst + a.dpad) = a.side == KEMR_SIDE_QUERY ? lo : hi;
        *(uint32_t*)(dst + 2 * a.dpad) = a.side == KEMR_SIDE_QUERY ? hi : lo;
    }
}

// ------------------------------------------------------------------------------------------------ fused sim + top-k
struct SimParams {
    const bf16_t* Q;
    const bf16_t* G;
    int nq, ng, kdim;
    long long goff;
    int k;
    float* part_scores;      // [nq, nchunks, k]
    int32_t* part_idx;
    const int32_t* gt_idx;
    const float* gt_score;
    int32_t* ahead;
    const int32_t* brow;
    const int32_t* bcol;
    const float* bval;
    float* dense;
    long long ld_dense;
    int nchunks, tiles_per_chunk, g_tiles;
    const int32_t* run_if;   // non-null: the launch is the exact fallback of the candidate-list path and exits unless *run_if != 0
};

template <int KMAX>
__device__ __forceinline__ void topk_insert(float (&s)[KMAX], int (&id)[KMAX], float v, int idx) {
#pragma unroll
    for (int i = KMAX - 1; i > 0; --i) {
        const bool shift = v > s[i - 1];
        const bool here = !shift && v > s[i];
        s[i] = shift ? s[i - 1] : (here ? v : s[i]);
        id[i] = shift ? id[i - 1] : (here ? idx : id[i]);
    }
    if (v > s[0]) { s[0] = v; id[0] = idx; }
}

__device__ __forceinline__ bool ranks_before(float sa, int ia, float sb, int ib) {
    return sa > sb || (sa == sb && ia < ib);
}

template <int KMAX, bool DENSE>
__global__ __launch_bounds__(256) void sim_kernel(const SimParams p) {
    constexpr int TILE_BYTES = ST * SBK * 2;             // 16 KiB per operand tile
    constexpr int STAGE_BYTES = 2 * TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [128][132] fp32 score tile (also the list exchange area).  It ALIASES the two operand stages: they are dead between
    // the last K-tile's barrier and the next tile's first stage, which is behind the barrier that ends the scan.  66 KiB
    // of LDS instead of 130 lets two workgroups share a CU, so one scans while the other is in its MFMA loop.
    float* sS = (float*)smem;

    if (p.run_if && *p.run_if == 0) return;              // uniform: before any barrier
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wid >> 1, wcn = wid & 1;              // wave's query half / candidate half
    const int qt = blockIdx.x / p.nchunks, chunk = blockIdx.x % p.nchunks;
    const int tile_begin = chunk * p.tiles_per_chunk;
    const int tile_end = min(tile_begin + p.tiles_per_chunk, p.g_tiles);

    const int srow = lane >> 3, schunk = lane & 7;
    const bf16_t* gQ = p.Q + (size_t)qt * ST * p.kdim;
    const int lrow = lane & 15, lq = lane >> 4;
    const int swz = lrow >> 1;
    const int nt = p.kdim / SBK;

    // scanner state: thread (q_local, half) owns candidates [half*64, half*64+64) of every tile for one query
    const int q_local = tid & 127, half = tid >> 7;
    const int qrow = qt * ST + q_local;
    const bool q_valid = qrow < p.nq;
    float ls[KMAX];
    int li[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) { ls[i] = -INFINITY; li[i] = -1; }
    const bool want_topk = p.part_scores != nullptr;      // rank-only calls (k == 0) skip the list maintenance
    int gt = -1, cnt = 0;
    float sgt = 0.f;
    int bcur = 0, bend = 0, next_col = INT_MAX;
    if (!DENSE && q_valid) {
        if (p.gt_idx) { gt = p.gt_idx[qrow]; sgt = p.gt_score[qrow]; }
        if (p.brow) {
            bcur = p.brow[qrow]; bend = p.brow[qrow + 1];
            if (bcur < bend) next_col = p.bcol[bcur];
        }
    }

    for (int gtile = tile_begin; gtile < tile_end; ++gtile) {
        const bf16_t* gG = p.G + (size_t)gtile * ST * p.kdim;
        auto stage = [&](int buf, int kt) {
            char* sA = smem + buf * STAGE_BYTES;         // gallery tile
            char* sB = sA + TILE_BYTES;                  // query tile
            const int k0 = kt * SBK;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int piece = wid * 4 + j;
                const int r = piece * 8 + srow;
                const int c = schunk ^ ((r >> 1) & 7);
                glds16s(gG + (size_t)r * p.kdim + k0 + c * 8, sA + piece * 1024);
                glds16s(gQ + (size_t)r * p.kdim + k0 + c * 8, sB + piece * 1024);
            }
        };
        f32x4 acc[4][4];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) acc[ci][qi] = f32x4{0.f, 0.f, 0.f, 0.f};

        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nt) stage(cur ^ 1, kt + 1);
            const char* sA = smem + cur * STAGE_BYTES + (wcn * 64 + lrow) * 128;
            const char* sB = smem + cur * STAGE_BYTES + TILE_BYTES + (wq * 64 + lrow) * 128;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int coff = ((kk * 4 + lq) ^ swz) << 4;
                bf16x8 gf[4], qf[4];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) gf[ci] = *(const bf16x8*)(sA + ci * 16 * 128 + coff);
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) qf[qi] = *(const bf16x8*)(sB + qi * 16 * 128 + coff);
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int qi = 0; qi < 4; ++qi)
                        acc[ci][qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[ci], qf[qi], acc[ci][qi], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // acc[ci][qi][r] = S[query wq*64 + qi*16 + lrow][candidate wcn*64 + ci*16 + lq*4 + r]
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const f32x4 v = acc[ci][qi];
                *(float4*)(sS + (wq * 64 + qi * 16 + lrow) * SLD + wcn * 64 + ci * 16 + lq * 4) =
                    make_float4(v[0], v[1], v[2], v[3]);
            }
        __syncthreads();

        const int cbase = gtile * ST;
        if constexpr (DENSE) {
            for (int r = wid * 32; r < wid * 32 + 32; ++r) {
                const int qr = qt * ST + r;
                const int c = cbase + lane * 2;
                if (qr < p.nq) {
                    const float2 v = *(const float2*)(sS + r * SLD + lane * 2);
                    float* dst = p.dense + (size_t)qr * p.ld_dense + c;
                    if (c < p.ng) dst[0] = v.x;
                    if (c + 1 < p.ng) dst[1] = v.y;
                }
            }
        } else if (q_valid) {
            const float* rowp = sS + q_local * SLD + half * 64;
#pragma unroll 4
            for (int j4 = 0; j4 < 16; ++j4) {
                const float4 v4 = *(const float4*)(rowp + j4 * 4);
                const float ve[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int lid = cbase + half * 64 + j4 * 4 + e;
                    if (lid >= p.ng) continue;
                    const long long g64 = p.goff + lid;
                    const int gid = (int)g64;
                    float sc = ve[e];
                    if (gid >= next_col) {                    // rare: this candidate may carry a SPARQL bonus
                        while (bcur < bend && p.bcol[bcur] < gid) ++bcur;
                        while (bcur < bend && p.bcol[bcur] == gid) { sc += p.bval[bcur]; ++bcur; }
                        next_col = bcur < bend ? p.bcol[bcur] : INT_MAX;
                    }
                    if (gid != gt) cnt += ranks_before(sc, gid, sgt, gt) ? 1 : 0;
                    if (want_topk && sc > ls[KMAX - 1]) topk_insert<KMAX>(ls, li, sc, gid);
                }
            }
        }
        __syncthreads();          // the scan is done with sS before the next tile's operands are staged over it
    }

    if constexpr (!DENSE) {
        float* xs = sS;                                   // [128][2][KMAX] scores
        int* xi = (int*)(sS + ST * 2 * KMAX);             // [128][2][KMAX] ids
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            xs[(q_local * 2 + half) * KMAX + i] = ls[i];
            xi[(q_local * 2 + half) * KMAX + i] = li[i];
        }
        __syncthreads();
        if (q_valid) {
            if (p.ahead && gt >= 0 && cnt) atomicAdd(p.ahead + qrow, cnt);
            if (half == 0 && p.part_scores) {
                const float* as = xs + (q_local * 2) * KMAX;
                const float* bs = as + KMAX;
                const int* ai = xi + (q_local * 2) * KMAX;
                const int* bi = ai + KMAX;
                int ia = 0, ib = 0;
                float* os = p.part_scores + ((size_t)qrow * p.nchunks + chunk) * p.k;
                int32_t* oi = p.part_idx + ((size_t)qrow * p.nchunks + chunk) * p.k;
                for (int o = 0; o < p.k; ++o) {
                    const bool ta = ia < KMAX && ai[ia] >= 0, tb = ib < KMAX && bi[ib] >= 0;
                    float so = -INFINITY;
                    int io = -1;
                    if (ta && (!tb || ranks_before(as[ia], ai[ia], bs[ib], bi[ib]))) { so = as[ia]; io = ai[ia]; ++ia; }
                    else if (tb) { so = bs[ib]; io = bi[ib]; ++ib; }
                    os[o] = so;
                    oi[o] = io;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ list merge
// one wave per query; k rounds of "best element strictly after the previous pick" over nlists*k entries
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ in_s, const int32_t* __restrict__ in_i,
                                                         int nq, int total, int k, float* __restrict__ out_s,
                                                         int32_t* __restrict__ out_i, const int32_t* __restrict__ run_if) {
    if (run_if && *run_if == 0) return;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float* s = in_s + (size_t)q * total;
    const int32_t* ix = in_i + (size_t)q * total;
    float ps = INFINITY;
    int pi = -1;
    for (int o = 0; o < k; ++o) {
        float bs = -INFINITY;
        int bi = -1;
        for (int e = lane; e < total; e += 64) {
            const float es = s[e];
            const int ei = ix[e];
            if (ei < 0) continue;
            const bool after_prev = (o == 0) || ranks_before(ps, pi, es, ei);
            if (after_prev && (bi < 0 || ranks_before(es, ei, bs, bi))) { bs = es; bi = ei; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oi = __shfl_xor(bi, off);
            if (oi >= 0 && (bi < 0 || ranks_before(os, oi, bs, bi))) { bs = os; bi = oi; }
        }
        if (lane == 0) { out_s[(size_t)q * k + o] = bi >= 0 ? bs : -INFINITY; out_i[(size_t)q * k + o] = bi; }
        if (bi < 0) {                                     // exhausted: pad the rest
            for (int r = o + 1 + lane; r < k; r += 64) { out_s[(size_t)q * k + r] = -INFINITY; out_i[(size_t)q * k + r] = -1; }
            break;
        }
        ps = bs;
        pi = bi;
    }
}

// Few queries, many lists (the online path: 1 query x 336 gallery chunks x k): one 256-thread workgroup per query keeps
// its <= 16 entries per thread in registers, so the k selection rounds never go back to memory (the wave-per-query
// kernel above re-reads all entries every round: 142 us for one query over 43 000 items).  Same order rule, same output.
constexpr int MERGE_EPT = 16;
__global__ __launch_bounds__(256) void topk_merge_block_kernel(const float* __restrict__ in_s, const int32_t* __restrict__ in_i,
                                                               int total, int k, float* __restrict__ out_s,
                                                               int32_t* __restrict__ out_i) {
    __shared__ float ws[4];
    __shared__ int wi[4];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* s = in_s + (size_t)q * total;
    const int32_t* ix = in_i + (size_t)q * total;
    float es[MERGE_EPT];
    int ei[MERGE_EPT];
#pragma unroll
    for (int j = 0; j < MERGE_EPT; ++j) {
        const int e = tid + j * 256;
        es[j] = e < total ? s[e] : 0.f;
        ei[j] = e < total ? ix[e] : -1;
    }
    float ps = INFINITY;
    int pi = -1;
    for (int o = 0; o < k; ++o) {
        float bs = -INFINITY;
        int bi = -1;
#pragma unroll
        for (int j = 0; j < MERGE_EPT; ++j) {
            if (ei[j] < 0) continue;
            const bool after_prev = (o == 0) || ranks_before(ps, pi, es[j], ei[j]);
            if (after_prev && (bi < 0 || ranks_before(es[j], ei[j], bs, bi))) { bs = es[j]; bi = ei[j]; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oi = __shfl_xor(bi, off);
            if (oi >= 0 && (bi < 0 || ranks_before(os, oi, bs, bi))) { bs = os; bi = oi; }
        }
        __syncthreads();                                  // the previous round's ws / wi have been read
        if (lane == 0) { ws[wid] = bs; wi[wid] = bi; }
        __syncthreads();
        bs = ws[0]; bi = wi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (wi[w] >= 0 && (bi < 0 || ranks_before(ws[w], wi[w], bs, bi))) { bs = ws[w]; bi = wi[w]; }
        if (tid == 0) { out_s[(size_t)q * k + o] = bi >= 0 ? bs : -INFINITY; out_i[(size_t)q * k + o] = bi; }
        if (bi < 0) {                                     // exhausted (uniform): pad the rest
            for (int r = o + 1 + tid; r < k; r += 256) { out_s[(size_t)q * k + r] = -INFINITY; out_i[(size_t)q * k + r] = -1; }
            break;
        }
        ps = bs;
        pi = bi;
    }
}

static int launch_topk_merge(const float* in_s, const int32_t* in_i, int nq, int total, int k, float* out_s, int32_t* out_i,
                             hipStream_t stream, const int32_t* run_if = nullptr) {
    if (!run_if && nq <= 64 && total > 256 && total <= 256 * MERGE_EPT) {
        hipLaunchKernelGGL(topk_merge_block_kernel, dim3(nq), dim3(256), 0, stream, in_s, in_i, total, k, out_s, out_i);
        KEMR_CHECK_LAUNCH("topk_merge_block_kernel");
    } else {
        hipLaunchKernelGGL(topk_merge_kernel, dim3((nq + 3) / 4), dim3(256), 0, stream, in_s, in_i, nq, total, k, out_s, out_i, run_if);
        KEMR_CHECK_LAUNCH("topk_merge_kernel");
    }
    return KEMR_OK;
}

// ------------------------------------------------------------------------------------------------ pair scores
// 16 (query row, gallery row) pairs per wave; same operand roles and k order as sim_kernel, diagonal extracted
__global__ __launch_bounds__(64) void pair_scores_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ G,
                                                         int kdim, const int32_t* __restrict__ qrows,
                                                         const int32_t* __restrict__ grows, int npairs,
                                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63, lrow = lane & 15, lq = lane >> 4;
    const int pair = blockIdx.x * 16 + lrow;
    const int pc = pair < npairs ? pair : npairs - 1;
    const bf16_t* qp = Q + (size_t)qrows[pc] * kdim + lq * 8;
    const bf16_t* gp = G + (size_t)grows[pc] * kdim + lq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < kdim; k0 += 32) {
        const bf16x8 gf = *(const bf16x8*)(gp + k0);
        const bf16x8 qf = *(const bf16x8*)(qp + k0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, qf, acc, 0, 0, 0);
    }
    // D[row = candidate i][col = query j]; the pair's own score is D[i][i]: lane (lrow = i, lq = i >> 2), reg i & 3
    if (lq == (lrow >> 2) && pair < npairs) {
        const int r = lrow & 3;
        out[pair] = r == 0 ? acc[0] : (r == 1 ? acc[1] : (r == 2 ? acc[2] : acc[3]));
    }
}

}  // namespace kemr

// ================================================================================================ C ABI
using namespace kemr;

extern "C" int64_t kemr_panel_kdim(int d, int nparts, int terms) {
    if (d <= 0 || nparts <= 0 || (terms != 1 && terms != 3)) return -1;
    return (int64_t)nparts * terms * round_up(d, 64);
}

extern "C" int kemr_panel_build(const float* const* parts_dev, const float* part_scale, const float* const* row_scale_dev,
                                int nparts, int rows, int d, int terms, int side, void* panel_dev, void* stream) {
    if (!parts_dev || !panel_dev) KEMR_FAIL(KEMR_ERR_INVALID, "panel_build: null pointer");
    if (nparts < 1 || nparts > 4) KEMR_FAIL(KEMR_ERR_INVALID, "panel_build: nparts %d not in 1..4", nparts);
    if (terms != 1 && terms != 3) KEMR_FAIL(KEMR_ERR_INVALID, "panel_build: terms must be 1 or 3");
    if (rows < 0 || d <= 0) KEMR_FAIL(KEMR_ERR_INVALID, "panel_build: bad shape rows=%d d=%d", rows, d);
    if (side != KEMR_SIDE_QUERY && side != KEMR_SIDE_GALLERY) KEMR_FAIL(KEMR_ERR_INVALID, "panel_build: bad side");
    PanelArgs a;
    for (int p = 0; p < 4; ++p) {
        a.parts[p] = p < nparts ? parts_dev[p] : nullptr;
        a.row_scale[p] = (p < nparts && row_scale_dev) ? row_scale_dev[p] : nullptr;
        a.part_scale[p] = (p < nparts && part_scale) ? part_scale[p] : 1.0f;
        if (p < nparts && !a.parts[p]) KEMR_FAIL(KEMR_ERR_INVALID, "panel_build: part %d is null", p);
    }
    a.nparts = nparts; a.rows = rows; a.rows_alloc = (int)round_up(rows, 256); a.d = d; a.dpad = (int)round_up(d, 64);
    a.terms = terms; a.side = side; a.kdim = kemr_panel_kdim(d, nparts, terms);
    const long long total = (long long)a.rows_alloc * nparts * (a.dpad / 2);
    if (total == 0) return KEMR_OK;
    hipLaunchKernelGGL(panel_build_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                       (bf16_t*)panel_dev, total);
    KEMR_CHECK_LAUNCH("panel_build_kernel");
    return KEMR_OK;
}

// ------------------------------------------------------------------------------------------------ top-k by candidate lists
// Large galleries: the scores come out of the persistent GEMM's K loop (gemm256u.hip, SIM == 2) instead of sim_kernel.
//   1. thresholds: a strided SAMPLE of the gallery (m rows, from k: simk_layout; read in place with a row stride) goes through the same K loop, which keeps the best
//      score of every query within each block of 64 sampled rows (SIM == 3); the k-th largest of these block maxima is the
//      score of k distinct gallery items, hence a lower bound of the k-th score of the gallery: every member of the true
//      top-k scores >= it -- whatever the data, the lists below contain the answer.
//   2. the 256 x 256-tile pass over the whole gallery: a lane whose 16 candidates of a query hold a score >= the query's
//      threshold appends them as one record to the (query, chunk) list; the ground-truth rank count rides in the same pass.
//   3. one wave per query picks the k best of its lists' entries with the path's order rule (score desc, id asc).
// A list that overflows (thresholds far too low: e.g. thousands of equal scores) raises a flag, and the sim_kernel path
// then runs after all: its launches are always queued and exit at once while the flag is clear (no host round trip).
// one wave per query: k-th largest of its <= 128 block maxima (k rounds of "wave maximum, remove one instance")
__global__ __launch_bounds__(256) void simk_threshold_kernel(const float* __restrict__ gmax, int nq, int groups, int k,
                                                             float* __restrict__ taud, int32_t* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *flag = 0;
    if (q >= nq) return;
    float v0 = lane < groups ? gmax[(size_t)q * groups + lane] : -INFINITY;
    float v1 = lane + 64 < groups ? gmax[(size_t)q * groups + lane + 64] : -INFINITY;
    float m = -INFINITY;
    for (int o = 0; o < k; ++o) {
        m = fmaxf(v0, v1);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const unsigned long long owners = __builtin_amdgcn_ballot_w64(v0 == m || v1 == m);
        if (lane == __builtin_ctzll(owners)) {
            if (v0 == m) v0 = -INFINITY;
            else v1 = -INFINITY;
        }
    }
    if (lane == 0) {
        const unsigned u = __float_as_uint(m);
        // next float below the threshold (s > taud  <=>  s >= threshold); fewer than k blocks: -inf, everything is listed
        taud[q] = u == 0xff800000u ? -INFINITY : __uint_as_float((u << 1) == 0u ? 0x80000001u : ((u >> 31) ? u + 1u : u - 1u));
    }
}

constexpr int SIMK_SELECT = 512;        // entries >= threshold one query may bring to the selection (more: the flag)
__global__ __launch_bounds__(256) void simk_select_kernel(const float* __restrict__ rec_scores, const int32_t* __restrict__ rec_base,
                                                          const int32_t* __restrict__ rec_count, int nchunks, int cap,
                                                          const float* __restrict__ taud, int nq, int k, int n_end,
                                                          float* __restrict__ out_s, int32_t* __restrict__ out_i,
                                                          int32_t* __restrict__ flag) {
    __shared__ float ls[4][SIMK_SELECT];
    __shared__ int li[4][SIMK_SELECT];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;                                             // no workgroup barrier below
    const float td = taud[q];
    int n = 0;
    for (int c0 = 0; c0 < nchunks; c0 += 64) {
        const int mine = c0 + lane < nchunks ? rec_count[(size_t)q * nchunks + c0 + lane] : 0;      // 64 list lengths at once
        const int span = min(64, nchunks - c0);
        // The lists are walked in units of 16 records (a lane takes a quarter -- 4 scores -- of one record); the loads of the
        // next unit are in flight while the current one is compacted (the walk is a chain of dependent memory round trips).
        int ucc = -1, ur0 = 0, ucnt = 0;                              // wave-uniform cursor: chunk, first record, list length
        auto advance = [&]() {                                       // -> next unit, false when this group of lists is done
            ur0 += 16;
            while (ur0 >= ucnt) {
                if (++ucc >= span) return false;
                ucnt = __shfl(mine, ucc);
                ur0 = 0;
            }
            return true;
        };
        struct Loaded { float4 sc; int id0; bool ok; };
        auto load = [&]() {
            Loaded u;
            const int rec = ur0 + (lane >> 2), ni = lane & 3;
            const size_t rec0 = ((size_t)q * nchunks + c0 + ucc) * cap;
            u.ok = rec < ucnt;
            u.sc = u.ok ? ((const float4*)rec_scores)[(rec0 + rec) * 4 + ni] : make_float4(0.f, 0.f, 0.f, 0.f);
            u.id0 = u.ok ? rec_base[rec0 + rec] + ni * 16 : 0;
            return u;
        };
        bool have = advance();
        Loaded cur{};
        if (have) cur = load();
        while (have) {
            const bool more = advance();
            Loaded nxt{};
            if (more) nxt = load();
            const float v[4] = {cur.sc.x, cur.sc.y, cur.sc.z, cur.sc.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool keep = cur.ok && v[r] > td && cur.id0 + r < n_end;       // id >= n_end: the gallery panel's zero pad rows
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
                const int pos = n + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                if (keep && pos < SIMK_SELECT) { ls[w][pos] = v[r]; li[w][pos] = cur.id0 + r; }
                n += __builtin_popcountll(mask);
            }
            cur = nxt;
            have = more;
        }
    }
    if (n > SIMK_SELECT) {
        if (lane == 0) atomicOr(flag, 1);
        n = SIMK_SELECT;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float ps = INFINITY;
    int pi = -1;
    for (int o = 0; o < k; ++o) {
        float bs = -INFINITY;
        int bi = -1;
        for (int e = lane; e < n; e += 64) {
            const float es = ls[w][e];
            const int ei = li[w][e];
            const bool after_prev = (o == 0) || ranks_before(ps, pi, es, ei);
            if (after_prev && (bi < 0 || ranks_before(es, ei, bs, bi))) { bs = es; bi = ei; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oi = __shfl_xor(bi, off);
            if (oi >= 0 && (bi < 0 || ranks_before(os, oi, bs, bi))) { bs = os; bi = oi; }
        }
        if (lane == 0) { out_s[(size_t)q * k + o] = bi >= 0 ? bs : -INFINITY; out_i[(size_t)q * k + o] = bi; }
        if (bi < 0) {
            for (int r = o + 1 + lane; r < k; r += 64) { out_s[(size_t)q * k + r] = -INFINITY; out_i[(size_t)q * k + r] = -1; }
            break;
        }
        ps = bs;
        pi = bi;
    }
}

// Rank-only passes WITH a bonus list (the alpha sweep of the SPARQL score fusion, evaluator.py:164-218 -> eval/fusion.py:22-85: nine
// passes over the 43 000 x 43 000 problem per weight setting) take the fast rank-count pass on the RAW scores against the ground
// truth's fused score, and this kernel then corrects the count for the few candidates that carry a bonus: one wave per query walks
// its CSR row 16 entries at a time, recomputes those candidates' raw scores with the tile kernels' arithmetic (same MFMA, same
// operand roles and k order as pair_scores_kernel: bit-identical), and moves each from the side of the ground truth its raw score
// put it on to the side its fused score puts it on.  Entries of one candidate (equal columns) add up in list order, as in
// sim_kernel's scanner; the ground truth itself and candidates outside this gallery are skipped.
__global__ __launch_bounds__(64) void bonus_rank_fixup_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ G, int nq, int ng,
                                                              int kdim, long long goff, const int32_t* __restrict__ brow,
                                                              const int32_t* __restrict__ bcol, const float* __restrict__ bval,
                                                              const int32_t* __restrict__ gt_idx, const float* __restrict__ gt_score,
                                                              int32_t* __restrict__ ahead) {
    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63, lrow = lane & 15, lq = lane >> 4;
    const int e0 = brow[q], e1 = brow[q + 1];
    if (e0 >= e1) return;
    const float g = gt_score[q];
    const int gt = gt_idx[q];
    const bf16_t* qp = Q + (size_t)q * kdim + lq * 8;
    int delta = 0;
    for (int base = e0; base < e1; base += 16) {
        const int e = base + lrow;
        const int col = e < e1 ? bcol[e] : -1;
        const long long loc = (long long)col - goff;
        const bool inside = e < e1 && loc >= 0 && loc < ng;
        const bf16_t* gp = G + (size_t)(inside ? loc : 0) * kdim + lq * 8;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < kdim; k0 += 32) {
            const bf16x8 gf = *(const bf16x8*)(gp + k0);
            const bf16x8 qf = *(const bf16x8*)(qp + k0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, qf, acc, 0, 0, 0);
        }
        // D[candidate i][query column]: every column is this query; lanes with lrow == 0 hold candidates 4 lq .. 4 lq + 3
        if (lrow == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ee = base + lq * 4 + r;
                if (ee >= e1) continue;
                const int c = bcol[ee];
                const long long lc = (long long)c - goff;
                if (lc < 0 || lc >= ng || c == gt) continue;
                if (ee > e0 && bcol[ee - 1] == c) continue;              // not the first entry of this candidate's run
                float fused = acc[r];
                for (int t = ee; t < e1 && bcol[t] == c; ++t) fused += bval[t];
                const float raw = acc[r];
                const int before = (raw > g || (raw == g && c < gt)) ? 1 : 0;
                const int after = (fused > g || (fused == g && c < gt)) ? 1 : 0;
                delta += after - before;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) delta += __shfl_xor(delta, off);
    if (lane == 0 && delta != 0) ahead[q] += delta;
}

namespace kemr { int g_sim_lists = 1; }

static int sim_chunks(int nq, int ng, int* tiles_per_chunk) {
    const int q_tiles = (nq + ST - 1) / ST, g_tiles = (ng + ST - 1) / ST;
    int want = (768 + q_tiles - 1) / q_tiles;           // aim for >= 3 workgroups per CU
    if (want > g_tiles) want = g_tiles;
    if (want < 1) want = 1;
    const int tpc = (g_tiles + want - 1) / want;
    *tiles_per_chunk = tpc;
    return (g_tiles + tpc - 1) / tpc;
}

static size_t sim_lists_bytes(int nq, int ng, int k) {
    int tpc;
    const int nchunks = sim_chunks(nq, ng, &tpc);
    return (size_t)round_up((int64_t)nq * nchunks * k * 8, 256);
}

// workspace of the candidate-list path (behind the sim_kernel path's own lists, which its fallback needs)
struct SimkLayout {
    bool on = false;
    int m = 0, stride = 1;         // sampled gallery rows: 0, stride, 2 stride, ...
    SimkPlan plan{};
    size_t off_gmax = 0, off_taud = 0, off_flag = 0, off_count = 0, off_base = 0, off_scores = 0, bytes = 0;
};

static SimkLayout simk_layout(int nq, int ng, int64_t kdim, int k) {
    SimkLayout L;
    L.bytes = sim_lists_bytes(nq, ng, k);
    // from 2 048 gallery rows up (round 2: 8 192): an 8-way shard of the 43 000 gallery has 5 375 rows (BASELINE configs[3])
    if (k < 1 || nq < 256 || ng < 2048 || kdim % 64 != 0 || kdim < 128 || kdim > 65536) return L;
    // ... where it pays: the route is six launches (sample pass, thresholds, list pass, selection, two fallback launches that exit
    // at once) against sim_kernel's two, and runs at about twice sim_kernel's rate.  Measured (round 3, same device, query panel
    // build included): 1 024 x 5 375 x 768 0.089 ms against 0.086 -- a tie; 1 024 x 43 000 0.152 against 0.361; break-even near
    // 1.2e10 multiply-adds.  Below it sim_kernel keeps the call (debug switch sim_lists >= 2: the lists wherever they fit, tests).
    if (g_sim_lists == 1 && (double)nq * ng * (double)kdim < 1.2e10) return L;
    // Sampled rows m: the entries >= threshold a query brings to the selection number about 1.15 k ng / m (the k-th block
    // maximum sits a little below the k-th item of the sample; measured 1.1x), Gamma(k)-distributed around that mean: the
    // smaller k, the longer the tail (k = 1: exponential).  spread(k) ~ the 1 - 1e-8 quantile over the mean (20.5 / 6.3 / 4.1 /
    // 2.4 for k = 1 / 5 / 10 / 32).  m is chosen so that mean x spread fits the selection's SIMK_SELECT entries (k = 10 at
    // ng = 43 000: 4 096 rows, 121 entries; k = 32: 7 680 rows; k = 1: 2 048 rows); where the largest sample cannot do that the
    // route is off.  The per-(query, chunk) lists are sized with the same spread.
    const double spread = 1.0 + 5.5 / sqrt((double)k) + 14.0 / k;
    const double mean_max = SIMK_SELECT / spread;
    int m = (int)round_up((int64_t)ceil(1.15 * k * (double)ng / mean_max), 256);
    const int m_min = (int)round_up(64 * (int64_t)k, 256);                     // k block maxima need k blocks of 64 rows
    m = m < 512 ? 512 : m;
    m = m < m_min ? m_min : m;
    if (m > 8192 || 2 * m > ng) return L;            // at most half the gallery as the sample (the sample pass costs m / ng of the list pass)
    L.stride = ng / m;                               // rows 0, stride, ... (m - 1) stride: distinct, all < ng
    bool ok = false;
    if (gemm256u_simk_plan(nq, ng, (int)kdim, 1.15 * k * ng / m, spread, &L.plan, &ok) != KEMR_OK || !ok) return L;
    size_t at = L.bytes;
    auto take = [&](size_t b) { const size_t o = at; at += (size_t)round_up((int64_t)b, 256); return o; };
    L.off_gmax = take((size_t)nq * (m / 64) * 4);
    L.off_taud = take((size_t)nq * 4);
    L.off_flag = take(4);
    L.off_count = take(L.plan.count_bytes);
    L.off_base = take(L.plan.base_bytes);
    L.off_scores = take(L.plan.scores_bytes);
    if (at > ((size_t)8 << 30)) return L;          // lists beyond 8 GiB: not this path
    L.on = true; L.m = m; L.bytes = at;
    return L;
}

extern "C" size_t kemr_sim_workspace_bytes(int nq, int ng, int64_t kdim, int k) {
    if (nq <= 0 || ng <= 0 || k <= 0) return 0;     // k == 0: rank only, no partial lists
    return simk_layout(nq, ng, kdim, k).bytes;
}

template <int KMAX, bool DENSE>
static int launch_sim(const SimParams& p, hipStream_t stream) {
    constexpr int stages = 2 * 2 * ST * SBK * 2, tile = ST * SLD * 4;
    constexpr int smem = stages > tile ? stages : tile;      // the score tile aliases the stages
    auto kern = sim_kernel<KMAX, DENSE>;
    static int attr_dev = -1;                 // per instantiation; one process drives one device at a time
    int dev = 0;
    KEMR_CHECK_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_dev = dev;
    }
    const int q_tiles = (p.nq + ST - 1) / ST;
    ProfScope prof(PROF_SIM, stream);
    hipLaunchKernelGGL(kern, dim3(q_tiles * p.nchunks), dim3(256), smem, stream, p);
    KEMR_CHECK_LAUNCH("sim_kernel");
    return KEMR_OK;
}

static int check_panels(const void* q, int nq, const void* g, int ng, int64_t kdim) {
    if (!q || !g) KEMR_FAIL(KEMR_ERR_INVALID, "sim: null panel");
    if (nq <= 0 || ng <= 0) KEMR_FAIL(KEMR_ERR_INVALID, "sim: empty operand (nq=%d ng=%d)", nq, ng);
    if (kdim <= 0 || kdim % 64 != 0 || kdim > (1 << 20)) KEMR_FAIL(KEMR_ERR_INVALID, "sim: kdim %lld must be a positive multiple of 64", (long long)kdim);
    return KEMR_OK;
}

extern "C" int kemr_sim_topk(const void* q_panel_dev, int nq, const void* g_panel_dev, int ng, int64_t kdim,
                             int64_t gallery_offset, int k, float* top_scores_dev, int32_t* top_idx_dev,
                             const int32_t* gt_idx_dev, const float* gt_score_dev, int32_t* ahead_dev,
                             const int32_t* bonus_rowptr_dev, const int32_t* bonus_col_dev, const float* bonus_val_dev,
                             void* workspace_dev, size_t workspace_bytes, void* stream) {
    KEMR_TRY(check_panels(q_panel_dev, nq, g_panel_dev, ng, kdim));
    if (k < 0 || k > 32) KEMR_FAIL(KEMR_ERR_INVALID, "sim_topk: k=%d not in 0..32", k);
    if (k > 0 && (!top_scores_dev || !top_idx_dev)) KEMR_FAIL(KEMR_ERR_INVALID, "sim_topk: null output");
    if (k == 0 && !gt_idx_dev) KEMR_FAIL(KEMR_ERR_INVALID, "sim_topk: k == 0 (rank only) needs a ground truth");
    if ((gt_idx_dev != nullptr) != (gt_score_dev != nullptr) || (gt_idx_dev != nullptr) != (ahead_dev != nullptr))
        KEMR_FAIL(KEMR_ERR_INVALID, "sim_topk: gt_idx, gt_score and ahead must be given together");
    if ((bonus_rowptr_dev != nullptr) != (bonus_col_dev != nullptr) || (bonus_rowptr_dev != nullptr) != (bonus_val_dev != nullptr))
        KEMR_FAIL(KEMR_ERR_INVALID, "sim_topk: bonus CSR arrays must be given together");
    if (gallery_offset < 0 || gallery_offset + ng > 0x7fffffffLL) KEMR_FAIL(KEMR_ERR_INVALID, "sim_topk: candidate ids exceed int32");
    const bool lists_ok = !bonus_rowptr_dev && k > 0;
    const SimkLayout L = lists_ok ? simk_layout(nq, ng, kdim, k) : SimkLayout{};
    const size_t need = k > 0 ? (L.on ? L.bytes : sim_lists_bytes(nq, ng, k)) : 0;
    if (k > 0 && (!workspace_dev || workspace_bytes < need)) KEMR_FAIL(KEMR_ERR_WORKSPACE, "sim_topk: workspace %zu < %zu bytes", workspace_bytes, need);
    if (k > 0 && (uintptr_t)workspace_dev % 256) KEMR_FAIL(KEMR_ERR_WORKSPACE, "sim_topk: workspace must be 256-byte aligned");
    SimParams p{};
    p.Q = (const bf16_t*)q_panel_dev; p.G = (const bf16_t*)g_panel_dev; p.nq = nq; p.ng = ng; p.kdim = (int)kdim;
    p.goff = gallery_offset; p.k = k;
    p.nchunks = sim_chunks(nq, ng, &p.tiles_per_chunk);
    p.g_tiles = (ng + ST - 1) / ST;
    p.part_scores = k > 0 ? (float*)workspace_dev : nullptr;
    p.part_idx = k > 0 ? (int32_t*)((char*)workspace_dev + (size_t)nq * p.nchunks * k * 4) : nullptr;
    p.gt_idx = gt_idx_dev; p.gt_score = gt_score_dev; p.ahead = ahead_dev;
    p.brow = bonus_rowptr_dev; p.bcol = bonus_col_dev; p.bval = bonus_val_dev;
    hipStream_t s = (hipStream_t)stream;
    if (k == 0 && (!bonus_rowptr_dev || g_sim_lists)) {        // ranks only: the 256 x 256-tile pass on the persistent GEMM's K loop (gemm256u.hip, SIM)
        bool used = false;
        KEMR_TRY(launch_gemm256u_simrank(p.Q, nq, p.G, ng, (int)kdim, gallery_offset, gt_idx_dev, gt_score_dev, ahead_dev, s, &used));
        if (used) {
            if (bonus_rowptr_dev) {            // + the candidates that carry a bonus change sides where their fused score says so
                hipLaunchKernelGGL(bonus_rank_fixup_kernel, dim3(nq), dim3(64), 0, s, p.Q, p.G, nq, ng, (int)kdim, (long long)gallery_offset,
                                   bonus_rowptr_dev, bonus_col_dev, bonus_val_dev, gt_idx_dev, gt_score_dev, ahead_dev);
                KEMR_CHECK_LAUNCH("bonus_rank_fixup_kernel");
            }
            return KEMR_OK;
        }
    }
    if (L.on && g_sim_lists) {                // top-k (and ranks) through candidate lists: see above
        char* ws = (char*)workspace_dev;
        float* gmax = (float*)(ws + L.off_gmax);
        float* taud = (float*)(ws + L.off_taud);
        int32_t* flag = (int32_t*)(ws + L.off_flag);
        bool used = false;
        KEMR_TRY(launch_gemm256u_simgmax(p.Q, nq, p.G, L.m, (int)kdim, L.stride, gmax, s, &used));
        if (!used) KEMR_FAIL(KEMR_ERR_STATE, "sim_topk: the sample pass does not fit the kernel that the plan accepted");
        hipLaunchKernelGGL(simk_threshold_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, gmax, nq, L.m / 64, k, taud, flag);
        KEMR_CHECK_LAUNCH("simk_threshold_kernel");
        KEMR_TRY(launch_gemm256u_simk(p.Q, nq, p.G, ng, (int)kdim, gallery_offset, gt_idx_dev, gt_score_dev, ahead_dev, taud, L.plan,
                                      (float*)(ws + L.off_scores), (int32_t*)(ws + L.off_base), (int32_t*)(ws + L.off_count), flag, s));
        hipLaunchKernelGGL(simk_select_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, (const float*)(ws + L.off_scores),
                           (const int32_t*)(ws + L.off_base), (const int32_t*)(ws + L.off_count), L.plan.nchunks, L.plan.cap, taud, nq, k,
                           (int)(gallery_offset + ng), top_scores_dev, top_idx_dev, flag);
        KEMR_CHECK_LAUNCH("simk_select_kernel");
        // the exact fallback, queued unconditionally: both kernels exit at once unless a list overflowed.  The ranks are
        // already complete (the count does not depend on the lists), so the fallback runs without a ground truth.
        p.gt_idx = nullptr; p.gt_score = nullptr; p.ahead = nullptr;
        p.run_if = g_sim_lists == 2 ? nullptr : flag;          // 2: the fallback forced to run after the lists (tests); 3: lists wherever they fit
        if (k <= 10) KEMR_TRY((launch_sim<10, false>(p, s)));
        else KEMR_TRY((launch_sim<32, false>(p, s)));
        return launch_topk_merge(p.part_scores, p.part_idx, nq, p.nchunks * k, k, top_scores_dev, top_idx_dev, s, p.run_if);
    }
    if (k <= 10) KEMR_TRY((launch_sim<10, false>(p, s)));
    else KEMR_TRY((launch_sim<32, false>(p, s)));
    if (k == 0) return KEMR_OK;
    return launch_topk_merge(p.part_scores, p.part_idx, nq, p.nchunks * k, k, top_scores_dev, top_idx_dev, s);
}

// tools: what the candidate-list route left in a workspace after kemr_sim_topk with the same sizes (synchronises the device):
// out[0] = overflow flag, out[1] = longest list, out[2] = list capacity, out[3] = chunks, out[4] = sampled rows,
// out[5] = sum of the list lengths / nq (records per query, rounded down)
extern "C" int kemr_debug_sim_lists(const void* workspace_dev, int nq, int ng, int64_t kdim, int k, int32_t* out6) {
    if (!workspace_dev || !out6) KEMR_FAIL(KEMR_ERR_INVALID, "debug_sim_lists: null argument");
    const SimkLayout L = simk_layout(nq, ng, kdim, k);
    for (int i = 0; i < 6; ++i) out6[i] = 0;
    if (!L.on) return KEMR_OK;
    KEMR_CHECK_HIP(hipDeviceSynchronize());
    const char* ws = (const char*)workspace_dev;
    KEMR_CHECK_HIP(hipMemcpy(&out6[0], ws + L.off_flag, 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> counts((size_t)nq * L.plan.nchunks);
    KEMR_CHECK_HIP(hipMemcpy(counts.data(), ws + L.off_count, counts.size() * 4, hipMemcpyDeviceToHost));
    long long sum = 0;
    for (int32_t c : counts) { sum += c; if (c > out6[1]) out6[1] = c; }
    out6[2] = L.plan.cap; out6[3] = L.plan.nchunks; out6[4] = L.m; out6[5] = (int32_t)(sum / nq);
    return KEMR_OK;
}

// tools / tests: 0 = always sim_kernel, 1 = candidate lists where they apply (default), 2 = lists AND the fallback forced to
// run after them (its result overwrites theirs: exercises the overflow route)
extern "C" int kemr_scores_dense(const void* q_panel_dev, int nq, const void* g_panel_dev, int ng, int64_t kdim,
                                 float* out_dev, int64_t ld_out, void* stream) {
    KEMR_TRY(check_panels(q_panel_dev, nq, g_panel_dev, ng, kdim));
    if (!out_dev || ld_out < ng) KEMR_FAIL(KEMR_ERR_INVALID, "scores_dense: bad output (ld=%lld)", (long long)ld_out);
    SimParams p{};
    p.Q = (const bf16_t*)q_panel_dev; p.G = (const bf16_t*)g_panel_dev; p.nq = nq; p.ng = ng; p.kdim = (int)kdim;
    p.k = 1; p.dense = out_dev; p.ld_dense = ld_out;
    p.nchunks = sim_chunks(nq, ng, &p.tiles_per_chunk);
    p.g_tiles = (ng + ST - 1) / ST;
    return launch_sim<10, true>(p, (hipStream_t)stream);
}

extern "C" int kemr_pair_scores(const void* q_panel_dev, const void* g_panel_dev, int64_t kdim, const int32_t* q_rows_dev,
                                const int32_t* g_rows_dev, int npairs, float* out_dev, void* stream) {
    if (npairs == 0) return KEMR_OK;
    if (!q_panel_dev || !g_panel_dev || !q_rows_dev || !g_rows_dev || !out_dev || npairs < 0)
        KEMR_FAIL(KEMR_ERR_INVALID, "pair_scores: bad argument");
    if (kdim <= 0 || kdim % 64 != 0) KEMR_FAIL(KEMR_ERR_INVALID, "pair_scores: kdim must be a positive multiple of 64");
    hipLaunchKernelGGL(pair_scores_kernel, dim3((npairs + 15) / 16), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)q_panel_dev,
                       (const bf16_t*)g_panel_dev, (int)kdim, q_rows_dev, g_rows_dev, npairs, out_dev);
    KEMR_CHECK_LAUNCH("pair_scores_kernel");
    return KEMR_OK;
}

extern "C" int kemr_topk_merge(const float* in_scores_dev, const int32_t* in_idx_dev, int nq, int nlists, int k,
                               float* out_scores_dev, int32_t* out_idx_dev, void* stream) {
    if (nq == 0) return KEMR_OK;
    if (!in_scores_dev || !in_idx_dev || !out_scores_dev || !out_idx_dev || nq < 0 || nlists < 1 || k < 1)
        KEMR_FAIL(KEMR_ERR_INVALID, "topk_merge: bad argument");
    return launch_topk_merge(in_scores_dev, in_idx_dev, nq, nlists * k, k, out_scores_dev, out_idx_dev, (hipStream_t)stream);
}
