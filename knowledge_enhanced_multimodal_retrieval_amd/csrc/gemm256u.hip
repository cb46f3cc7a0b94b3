// Persistent bf16 / fp8 GEMM for the bf16-output epilogues: one 512-thread workgroup (8 waves x (128 x 64)) per CU walks
// 256 x 256 output tiles with ONE K-tile pipeline that runs across output tiles: the LDS-DMA pieces staged during K-tile g
// belong to K-tiles g+1 (A0, A1, W1) and g+2 (W0) of the workgroup's whole tile sequence, so the first K-tiles of the next
// output tile are already in LDS when this tile's last MFMA retires, and a tile switch is just: epilogue (16 stores per
// lane), carry on (the first MFMA of every accumulator takes the bias as its C operand: no accumulator initialisation).
//
// Round 2, from in-kernel stamps (DBG instantiation, tools/bench_gemm_r2.py).  The two wave halves (wr = 0 / 1) run one
// barrier interval apart for the whole launch, so a SIMD alternates one wave's 16-MFMA cluster (256 cycles) with its
// partner's load interval, and an interval lasts as long as the longer of the two.  What a load interval costs is the
// number of instructions the wave has to issue in it (about 4 cycles each) plus 60-100 cycles per LDS-DMA piece: round 1's
// first interval (12 ds_read_b128, 2 pieces and the ~60 scalar instructions of the K-tile bookkeeping) took ~470 cycles,
// its 4- and 8-read intervals ~270-330.  Now:
//  * clusters are split along k instead of along the W columns, so no interval has more than 8 reads:
//      L1: W(k 0-31) + A rows 0-63 (k 0-31), stage A0(g+1)      M1: acc[rows 0-63]   += . (k 0-31)
//      L2: W(k 32-63) + A rows 0-63 (k 32-63), stage A1(g+1)    M2: acc[rows 0-63]   += . (k 32-63)
//      L3: A rows 64-127 (both k halves), stage W1(g+1)         M3: acc[rows 64-127] += . (k 0-31)
//      L4: stage W0(g+2); wait for K-tile g+1                   M4: acc[rows 64-127] += . (k 32-63) + bookkeeping
//    (every accumulator still sums k 0-31 before k 32-63 of each K-tile: bit-identical to round 1); the piece with the
//    shortest flight time (3 intervals) is a W half, which the tile order keeps L2-resident, instead of an A half;
//  * the K-tile bookkeeping is two pointer streams (A: K-tile g+1, W: K-tile g+2) advanced by 128 bytes, issued two or
//    three scalar instructions at a time BETWEEN the MFMAs of M4; a stream computes a tile's coordinates once, when the W
//    stream enters it, and hands them down (W -> A -> epilogue).
// fp8 operands keep the round-1 clusters (one K = 128 MFMA per K-tile has no k halves to split) on the same staging plan.
//
// Round 4 (A/Bs in profiles/r04_gemm_prio.txt, stamps in profiles/r04_gemm_stamps.txt): what the wave in its load interval issues comes out of
// its SIMD partner's MFMA cluster, so the K loop's load intervals now hold no vector-ALU instruction and no priority change -- no s_setprio
// flips (KEMR_GEMM_PRIO), the stream offsets in the scalar base of the LDS-DMA pieces (KEMR_GEMM_SBASE), the LDS-buffer flips as asm inside
// MFMA gaps (KEMR_GEMM_FLIPASM), the streams' wrap arithmetic only in the K-tiles that can wrap (KEMR_GEMM_MID) -- and the next tile's first A
// pieces are staged in front of the store epilogue so that no wait of its first K-tile covers the stores (KEMR_GEMM_PRESTAGE): GEMM class
// 0.5255 -> 0.545-0.553 of the bf16 peak, MFMA pipe busy 66.9 -> 72.6 %, results unchanged.  What remains is the tile boundary (every workgroup of
// the chip stores its tile at the same moment: 9 % of a K = 1024 launch) and the clock (1.75-1.90 GHz under the kernel).
//
// LDS regions and who reads them: a wave reads W from the half that holds its 64 columns (in L1 and L2) and A from its own
// row half (L1-L3); a region may be re-staged once BOTH wave halves are past their last read of it and have waited for the
// data (the lgkmcnt(0) in front of the next cluster) and a barrier lies in between: W0 of this K-tile's buffer from the
// wr == 0 half's L4 on; the other buffer (K-tile g+1) was last read a K-tile ago.
//
// vmcnt (loads, LDS-DMA and stores retire in order): the wait that closes K-tile g needs K-tile g+1 landed and may leave
// W0(g+2) (2 pieces per lane) in flight: vmcnt(2), or vmcnt(0) when nothing was staged behind it.  The 16 stores of a
// tile sit between W1(g+2) and A0(g+2) in that order; since round 4 (PRE, at the K-tile schedule) the store epilogue issues
// the next tile's A(1) pieces in front of them and the first K-tile's waits leave the stores in flight, so they get a K-tile
// and a half before a wait covers them.
// The next tile's 256 bias floats ride with its first W0 piece (wave 0) into the other half of a 2 x 1 KiB LDS area.
//
// Epilogue: 8 passes of 16 rows through LDS (acc -> act -> bf16 -> ds_write_b64, chunk-XOR swizzled -> ds_read_b128 -> one
// full 128-byte line per 8 lanes, non-temporal), double-buffered: the two wave halves never run their epilogues at the
// same time (they are separated by the barriers on either side), so a wave alternates between its own 2 KiB area and its
// partner's (wid ^ 4) and the LDS round trip of pass mi + 1 hides behind the stores of pass mi.  (Stamps: 3 300 cycles per
// wave half and tile for the single-buffered form, 4 400 for a register-only form with v_permlane16_swap and 64-byte row
// segments per store instruction - the full-line form is the one to keep.)
// CONC (the default for the QuickGELU epilogue only): both halves in the SAME barrier interval, each wave on its own area
// single-buffered; the next pass's QuickGELU arithmetic covers the LDS round trip (fc1: 11.1 k -> 9.2 k cycles per tile for
// both halves; +-0 on the plain epilogue).
//
// SIM = 1 / 2 / 3 (launch_gemm256u_simrank / _simk / _simgmax, called from sim.hip): the same K loop as the scoring half of the
// retrieval path -- A = query panel, W = gallery panel, C never written; the epilogue is a register scan of the accumulators
// (rank count / rank count + candidate lists / block maxima of a gallery sample); see the comments at the template and at
// the scan.  One workgroup = one 256-query tile x one chunk of gallery tiles; both halves scan in the same barrier interval.
#include "common.h"
#include <type_traits>
// (hipcc misses odr-uses inside asm operands of generic lambdas: the hooks below name their captures, which it then calls unused)
#pragma clang diagnostic ignored "-Wunused-lambda-capture"

namespace kemr {

namespace {

constexpr int PBUF = 65536;      // bytes per K-tile buffer
constexpr int PHALF = 16384;     // bytes per half-tile
constexpr int PEPI = 131072;     // offset of the epilogue area (8 waves x 2 KiB)
constexpr int PBIAS = PEPI + 16384;   // 2 x 1 KiB: fp32 bias of the current / next tile's 256 columns
constexpr int PSCALE = PBIAS + 2048;  // fp8 operands: 2 x 1 KiB per-output-channel weight scales of the current / next tile
constexpr int PSMEM = PSCALE + 2048;

// Epilogue / bias LDS traffic as inline asm: hipcc (SIInsertWaitcnts) guards every LDS access that carries a memory operand
// with `s_waitcnt vmcnt(0)` while an LDS-DMA is outstanding, which would drain the prefetch.  These areas are never a DMA
// target of an in-flight piece when they are read, so no such wait is needed; asm LDS ops carry no memory operand.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ u32x4 lds_read_b128(unsigned addr) {
    u32x4 d;
    asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory");
    return d;
}

template <int I> struct HookAt { static constexpr int value = I; };

// Wave priority around the MFMA clusters (compile-time, -DKEMR_GEMM_PRIO=n for tools/ab_build_flag.sh; a run-time switch inside the K loop
// costs 5-8 %).  1 = NO priority changes: the product kernel since round 4.  0 = priority 1 for the cluster, 0 for the load interval
// (rounds 1-3: the computing wave then starves its SIMD partner's load interval of issue slots exactly where the LDS-DMA pieces, at
// 60-100 issue cycles each, are the long pole); 2 = static, the second-dispatched wave half (waves 4-7) at priority 1 for the whole
// launch (cdna guide T5, static form); 3 = the inverse of 0 (the LOAD interval at priority 1); 4 = static for waves 0-3.
// Same box, interleaved builds, bench.py --steps 40 (profiles/r04_gemm_prio.txt): mode 0 GEMM class 31.82 ms per step (0.5255 of peak),
// mode 2 31.24 (0.5355), mode 1 31.14 (0.5371): +1.55 % on the step.  Results do not depend on the mode.
#ifndef KEMR_GEMM_PRESTAGE
#define KEMR_GEMM_PRESTAGE 1   // the next tile's K-tile-1 A pieces are staged at the START of the store epilogue (see the K-tile schedule); 0 = in its first K-tile (rounds 1-3)
#endif
#ifndef KEMR_GEMM_STORE_MIX
#define KEMR_GEMM_STORE_MIX 0      // C stores: 0 = all non-temporal (the product), 3 = every other 16-row pass as plain write-back stores, 4 = all plain (A/B)
#endif
#ifndef KEMR_GEMM_FLIPASM
#define KEMR_GEMM_FLIPASM 1
#endif
#ifndef KEMR_GEMM_SBASE
#define KEMR_GEMM_SBASE 1
#endif
#ifndef KEMR_GEMM_MID
#define KEMR_GEMM_MID 1        // K-tiles that cannot wrap a stream advance it by plain adds (0: the wrap arithmetic everywhere, rounds 2-3)
#endif
#ifndef KEMR_GEMM_PRIO
#define KEMR_GEMM_PRIO 1
#endif
__device__ __forceinline__ void cluster_prio(int v) {
    if constexpr (KEMR_GEMM_PRIO == 0) { if (v) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    if constexpr (KEMR_GEMM_PRIO == 3) { if (v) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
}

// 16 MFMAs: rows MH*64 .. +63 of the wave's tile x its 64 columns x one 32-wide k step.  FIRST: the accumulators' first
// product of a tile, C = the bias quad of the column group (a lane's acc[mi][ni] covers columns ni*16 + lq*4 .. +3 for every
// mi).  hook(HookAt<i>) runs after the i-th MFMA: a few scalar instructions issue for free while the matrix pipe is busy.
template <int MH, bool FIRST, class Hook>
__device__ __forceinline__ void cluster(f32x4 (&acc)[8][4], const bf16x8 (&af)[4], const bf16x8 (&wf)[4], const u32x4 (&b4)[4],
                                        Hook&& hook) {
    cluster_prio(1);
    asm volatile("s_nop 1" ::: "memory");      // any compiler VALU write just above -> first asm MFMA operand read
    auto step = [&](auto mi_c, auto ni_c) {
        constexpr int mi = decltype(mi_c)::value, ni = decltype(ni_c)::value;
        if constexpr (FIRST)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3"
                         : "=&v"(acc[MH * 4 + mi][ni]) : "v"(wf[ni]), "v"(af[mi]), "v"(b4[ni]));
        else
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                         : "+v"(acc[MH * 4 + mi][ni]) : "v"(wf[ni]), "v"(af[mi]));
        hook(HookAt<mi * 4 + ni>{});
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    step(I0{}, I0{}); step(I0{}, I1{}); step(I0{}, I2{}); step(I0{}, I3{});
    step(I1{}, I0{}); step(I1{}, I1{}); step(I1{}, I2{}); step(I1{}, I3{});
    step(I2{}, I0{}); step(I2{}, I1{}); step(I2{}, I2{}); step(I2{}, I3{});
    step(I3{}, I0{}); step(I3{}, I1{}); step(I3{}, I2{}); step(I3{}, I3{});
    cluster_prio(0);
    // the fragment reads the hooks issued (asm ds_read: hipcc does not count them) have returned before the wave arrives at the
    // barrier behind this cluster: that orders them in front of any re-staging of the region and in front of their use
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// 32 MFMAs: all 128 rows of the wave's tile x its 64 columns x one 32-wide k step (the long-interval K loop, KL): row tile
// mi's four MFMAs are issued together, so that af[mi] is dead behind the fourth and the hook of that gap may re-read it IN PLACE
// for the next cluster (the read's data returns >= 64 cycles later, the MFMA reads its operands in its first passes).
// ENDWAIT: what the wave waits for behind the cluster's last MFMA, in front of the barrier.
template <bool FIRST, class Hook>
__device__ __forceinline__ void cluster32(f32x4 (&acc)[8][4], const bf16x8 (&af)[8], const bf16x8 (&wf)[4], const u32x4 (&b4)[4],
                                          Hook&& hook) {
    cluster_prio(1);
    asm volatile("s_nop 1" ::: "memory");      // any compiler VALU write just above -> first asm MFMA operand read
    auto step = [&](auto mi_c, auto ni_c) {
        constexpr int mi = decltype(mi_c)::value, ni = decltype(ni_c)::value;
        if constexpr (FIRST)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3"
                         : "=&v"(acc[mi][ni]) : "v"(wf[ni]), "v"(af[mi]), "v"(b4[ni]));
        else
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                         : "+v"(acc[mi][ni]) : "v"(wf[ni]), "v"(af[mi]));
        hook(HookAt<mi * 4 + ni>{});
    };
    auto row = [&](auto mi_c) {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        step(mi_c, I0{}); step(mi_c, I1{}); step(mi_c, I2{}); step(mi_c, I3{});
    };
    row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{});
    row(std::integral_constant<int, 3>{}); row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{});
    row(std::integral_constant<int, 6>{}); row(std::integral_constant<int, 7>{});
    cluster_prio(0);
}

// fp8 (OCP e4m3) operands: one block-scaled MFMA covers K = 128 (a lane holds 32 consecutive k bytes of its row), at twice
// the bf16 MFMA's cycles, i.e. twice the FLOP rate; the E8M0 block scales are all 2^0 (0x7f), the real scales (one per
// output channel of W) are applied in the epilogue.  A K-tile is still 128 bytes per row: staging, LDS image and swizzle
// are the bf16 kernel's, a fragment is two adjacent 16-byte chunks instead of one.  8 MFMAs (of 32 cycles) per quadrant.
typedef __attribute__((ext_vector_type(8))) int fp8x32;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int MH, int NH, bool FIRST, class Hook>
__device__ __forceinline__ void quad8(f32x4 (&acc)[8][4], const fp8x32 (&af)[4], const fp8x32 (&wf)[2], int one, Hook&& hook) {
    cluster_prio(1);
    asm volatile("s_nop 1" ::: "memory");
    auto step = [&](auto mi_c, auto ni_c) {
        constexpr int mi = decltype(mi_c)::value, ni = decltype(ni_c)::value;
        if constexpr (FIRST)
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]"
                         : "=&v"(acc[MH * 4 + mi][NH * 2 + ni]) : "v"(wf[ni]), "v"(af[mi]), "v"(one));
        else
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                         : "+v"(acc[MH * 4 + mi][NH * 2 + ni]) : "v"(wf[ni]), "v"(af[mi]), "v"(one));
        hook(HookAt<2 * (mi * 2 + ni)>{});
        hook(HookAt<2 * (mi * 2 + ni) + 1>{});
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    step(I0{}, I0{}); step(I0{}, I1{}); step(I1{}, I0{}); step(I1{}, I1{});
    step(I2{}, I0{}); step(I2{}, I1{}); step(I3{}, I0{}); step(I3{}, I1{});
    cluster_prio(0);
}

}  // namespace

__device__ unsigned g_gemm_stamp_buf[1024 * 16];      // DBG stamps: [workgroup][16]

constexpr int GEMM256U_MAX_TILES_PER_WG = 62;         // the tile table is one lane per tile (+ 2 dummies behind the last)

// SIM: the same K loop as the scoring half of the path: A = query panel, W = gallery panel (both bf16 [rows, kdim], kdim
// contiguous: kemr_panel_build), C is never written; the epilogue counts, per query, the candidates of the tile that rank
// ahead of the query's ground truth (reference metrics.py:13-76: Recall@K / MRR need nothing else).  One workgroup = one
// 256-query tile x one chunk of gallery tiles (blockIdx = q_tile * nchunks + chunk); same MFMA operand roles and k order as
// sim_kernel / pair_scores_kernel (sim.hip), so the scores are theirs bit for bit.
template <int EPI, bool FP8, bool DBG = false, int SIM = 0, bool CONC = false, bool KL = false>
__global__ __launch_bounds__(512, 2) void gemm256u_bf16_nt_kernel(const GemmParams p) {
    constexpr int ES = FP8 ? 1 : 2;          // operand element size
    constexpr int CS = EPI == EPI_BIAS_RESID_F32 ? 4 : 2;      // C element size (the fp32 residual stream, else bf16)
    // LONGK (round 3 experiment, debug switch gemm_kl = 1; NOT the default): two 32-MFMA clusters per K-tile and wave instead of
    // four 16-MFMA ones, i.e. four barrier intervals of 512 MFMA cycles per K-tile instead of eight of 256, on the reading of round
    // 2's stamps (an interval takes 300-320 cycles for 256 of MFMA issue) that 50-60 cycles go to the hand-over whatever the
    // interval's length.  Measured (tools/bench_gemm_r3.py, same device, bit-identical results): 3 % SLOWER on every shape (fc2
    // 402 -> 414 us, out-proj 106 -> 111, 4096^3 96.4 -> 99.6).  Stamps: the 32-MFMA intervals take 660-740 cycles -- the cost is
    // per MFMA, not per hand-over (19-20 cycles per MFMA in either loop) -- and the interval that ends with the wait for K-tile
    // g+1 takes 880: the W regions of a buffer are read until two intervals before they are needed again, so their pieces have
    // ONE interval of flight here (three in the 8-interval loop).  Schedule per K-tile g (wave half wr = 1 one interval behind):
    //   LA: stage W(g+1) -> buffer g+1 [+ bias]       MA: acc += . (k 0-31), hooks read W / A (k 32-63) of buffer g, W stream;
    //                                                     then vmcnt(0): K-tile g+1 has landed (A from LB(g-1), W from LA(g))
    //   LB: stage A(g+2) -> buffer g (own half)        MB: acc += . (k 32-63), hooks read W / A (k 0-31) of buffer g+1, A stream
    // Who reads what, when (slots of one interval, half 0: LA 4g, MA 4g+1, LB 4g+2, MB 4g+3; half 1 one later): buffer g is read in
    // MB(g-1) and MA(g), i.e. slots 4g-1 .. 4g+2.  Its A region (private to a half) is free behind the half's own MA(g): LB(g)
    // re-stages it.  Its W regions are free from slot 4g+3 on: LA(g+1) (slots 4g+4 / 4g+5) re-stages them with K-tile g+2, which
    // MB(g+1) (slots 4g+7 / 4g+8) reads behind every wave's vmcnt(0) at the end of its MA(g+1) (slots 4g+5 / 4g+6) and a barrier.
    // A fragments are re-read in place (cluster32), W fragments alternate between two register sets: 64 VGPRs as before.
    // Every accumulator still sums k 0-31 before k 32-63 of each K-tile: bit-identical to the round-2 loop.
    constexpr bool LONGK = KL && !FP8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    if constexpr (KEMR_GEMM_PRIO == 2) { if (wid >= 4) __builtin_amdgcn_s_setprio(1); }      // (wid is wave-uniform: readfirstlane above)
    if constexpr (KEMR_GEMM_PRIO == 4) { if (wid < 4) __builtin_amdgcn_s_setprio(1); }

    const int tiles_n = p.N >> 8;
    const int tiles_m = (p.M + 255) >> 8;
    const int ntiles = tiles_m * tiles_n;
    const int G = (int)gridDim.x;
    const int full = (ntiles / G) * G;                               // tiles inside complete rounds
    // Tile order.  Blocks with equal blockIdx % 8 share an XCD (observed round-robin placement; speed only), and every XCD
    // has its own 4 MiB L2.  Within a round the G / 8 workgroups of an XCD take G / 8 consecutive positions of the order
    // below; a position is mapped to a tile so that these form a block of (G / 8 / cw) row tiles x cw column tiles, and an
    // XCD's consecutive rounds stay in the same column group: its cw W panels stay L2-resident while the A panels stream
    // (p.order = log2(cw) + 1; 0 = the round-1 order, N fastest: an XCD then cycles through ALL of W every round).
    const int lgcw = p.order - 1;
    auto tile_of = [&](int idx, int& row0, int& col0) {
        int L = idx;
        const int per = G >> 3;
        if (idx < full && (G & 7) == 0) {
            const int rnd = idx / G, b = idx - rnd * G;
            L = rnd * G + (b & 7) * per + (b >> 3);
        }
        int tm, tn;
        if (lgcw >= 0 && (G & 7) == 0 && per >= (1 << lgcw) && (per & ((1 << lgcw) - 1)) == 0) {
            const int cw = 1 << lgcw, rh = per >> lgcw;
            const int RM = (tiles_m / rh) * rh, CN = tiles_n & ~(cw - 1);
            const int nsup = RM * CN;
            if (L < nsup) {
                const int beta = L / per, within = L - beta * per;
                const int nrb = RM / rh;
                const int cg = beta / nrb, rb = beta - cg * nrb;
                tm = rb * rh + (within >> lgcw);
                tn = (cg << lgcw) + (within & (cw - 1));
            } else {
                int r = L - nsup;
                const int wdt = tiles_n - CN, strip = wdt * RM;      // right strip (rows < RM, columns >= CN), then the bottom rows
                if (r < strip) { tm = r / wdt; tn = CN + (r - tm * wdt); }
                else { r -= strip; tm = RM + r / tiles_n; tn = r - (tm - RM) * tiles_n; }
            }
        } else {
            tm = L / tiles_n;
            tn = L - tm * tiles_n;
        }
        row0 = tm << 8;
        col0 = tn << 8;
    };
    const int nt = FP8 ? p.K >> 7 : p.K >> 6;      // K-tile = 128 bytes per row; nt >= 2

    // ---- the workgroup's tile table: lane s = its s-th tile (tile index blockIdx + s * G), computed ONCE with vector
    // arithmetic; the scalar side fetches an entry with v_readlane when a stream enters a tile, so the K loop has no tile
    // arithmetic and no branch.  Lanes behind the last tile hold tile (0, 0): the pointer streams run two K-tiles past the
    // end and stage (valid, unused) data instead of being switched off.
    int ntl = (ntiles - (int)blockIdx.x + G - 1) / G;                // tiles of this workgroup, <= GEMM256U_MAX_TILES_PER_WG (host)
    int t_row = 0, t_col = 0;
    int sim_tb = 0;                                                  // SIM: first gallery tile of this workgroup's chunk
    int sim_chunk = 0;
    if constexpr (SIM != 0) {
        // (query tile, gallery chunk) of this workgroup.  Round 3's FETCH_SIZE pass of the list pass read 12 GB through the fabric
        // per launch at Q = N = 43 000 for 0.13 GB of operands: a workgroup re-reads its 393 KB query tile for every gallery tile,
        // and with blockIdx = tile * chunks + chunk the 32 workgroups an XCD holds at a time own ~11 different query tiles -- 12 MB
        // of A against a 4 MiB L2, so every re-read goes to the Infinity Cache.  Now the workgroups of ONE query tile (its chunks)
        // are dealt to ONE XCD (blocks with equal blockIdx % 8 share an XCD: observed placement, speed only), back to back, so
        // that they start together, walk their gallery tiles in step and share every re-read of the query tile in that XCD's
        // L2; the ~10 workgroups of an XCD that sit on the same chunk share its gallery tiles as before.  The grid is padded to
        // whole groups of 8 query tiles; a workgroup of a tile that does not exist returns before its first barrier.
        // (Few query tiles -- fewer than 64, not a multiple of 8 -- keep blockIdx = tile * chunks + chunk: Q = 1 024 is 4 tiles, which
        // this dealing would put on 4 of the 8 XCDs: 0.15 -> 0.21 ms.)
        int qt, chunk;
        if (p.sim_xcd) {
            const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
            const int jq = j / p.sim_nchunks;
            chunk = j - jq * p.sim_nchunks;
            qt = jq * 8 + xcd;
            if ((qt << 8) >= p.M) return;
        } else {
            qt = (int)blockIdx.x / p.sim_nchunks;
            chunk = (int)blockIdx.x - qt * p.sim_nchunks;
        }
        sim_chunk = chunk;
        sim_tb = chunk * p.sim_tpc;
        const int te = min(sim_tb + p.sim_tpc, tiles_n);
        ntl = te - sim_tb;
        t_row = qt << 8;
        t_col = (sim_tb + (lane < ntl ? lane : 0)) << 8;
    } else {
        const int tidx = (int)blockIdx.x + lane * G;
        if (tidx < ntiles) tile_of(tidx, t_row, t_col);
    }
    const int t_row_u = __builtin_amdgcn_readfirstlane(t_row);                  // SIM: the workgroup's query tile (the same in every lane)
    const unsigned v_aoff = (unsigned)t_row * (unsigned)(p.lda * ES);          // byte offsets (< 4 GiB: host check)
    const unsigned v_woff = (unsigned)t_col * (unsigned)(p.ldw * ES);
    const unsigned v_coff = ((unsigned)t_row * (unsigned)p.ldc + (unsigned)t_col) * (unsigned)CS;
    const int v_bcol = t_col;

    // staging addresses = wave-uniform K-tile pointer (SGPRs) + a per-lane 32-bit byte offset that never changes
    const int srow = lane >> 3, schunk = lane & 7;
    const int r0 = wid * 16 + srow, r1 = r0 + 8;                    // W: every wave stages 16 rows of each 128-row half
    // A: a wave half stages ITS OWN 128-row region (the only one it reads), 32 rows per wave in four 8-row pieces, so that the
    // wait for the A pieces only has to be agreed inside the half (see the K-tile schedule below)
    const int ar = wr * 128 + (wid & 3) * 32 + srow;
    unsigned a_lane[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = ar + j * 8;
        a_lane[j] = (unsigned)(r * p.lda * ES + ((schunk ^ ((r >> 1) & 7)) << 4));
    }
    const unsigned w_lane0 = (unsigned)(r0 * p.ldw * ES + ((schunk ^ ((r0 >> 1) & 7)) << 4));
    const unsigned w_lane1 = (unsigned)(r1 * p.ldw * ES + ((schunk ^ ((r1 >> 1) & 7)) << 4));

    // ---- two streams run ahead of the computing K-tile g of the workgroup's K-tile sequence: the A stream at K-tile g+1
    // (staged in L1 / L2), the W stream at g+2 (staged in L3 / L4).  A stream is a 32-bit byte offset into its matrix (tile
    // offset from the table + 128 bytes per K-tile), t* = K-tiles left in the stream's tile including the one it points at,
    // seq* = the tile's position in the table.
    int tw = nt, seqw = 0, ta = nt, seqa = 0;
    unsigned ow = __builtin_amdgcn_readlane(v_woff, 0), oa = __builtin_amdgcn_readlane(v_aoff, 0);
    const unsigned w_half = 128u * (unsigned)(p.ldw * ES);     // bytes between the two W half-tiles
    unsigned ow1 = ow + w_half;
    // one advance, branch-free, in five steps (inside the K loop one step per MFMA gap, see the hooks):
    //   1 --t; wrapped = t == 0; t = wrapped ? nt : t; seq += wrapped     (SCC carries `wrapped` into the s_addc)
    //   2 tile offset of table entry seq   3 offset + 128   4 pick by t == nt (true only right after a wrap)   5 second half
#define KEMR_STREAM_STEP1(T, SEQ) asm volatile("s_sub_i32 %0, %0, 1\n\ts_cmp_eq_u32 %0, 0\n\ts_cselect_b32 %0, %2, %0\n\ts_addc_u32 %1, %1, 0" \
                                               : "+s"(T), "+s"(SEQ) : "s"(nt) : "scc")
    unsigned toff_a = 0, toff_w = 0, inc_a = 0, inc_w = 0;
    auto advance_w = [&]() {
        KEMR_STREAM_STEP1(tw, seqw);
        toff_w = __builtin_amdgcn_readlane(v_woff, seqw);
        inc_w = ow + 128;
        ow = tw == nt ? toff_w : inc_w;
        ow1 = ow + w_half;
    };
    auto advance_a = [&]() {
        KEMR_STREAM_STEP1(ta, seqa);
        toff_a = __builtin_amdgcn_readlane(v_aoff, seqa);
        inc_a = oa + 128;
        oa = ta == nt ? toff_a : inc_a;
    };
    // LDS-DMA of one 16-row x 128-byte piece per wave: SGPR matrix base + per-lane 32-bit offset -> LDS at M0 + lane * 16.
    // All LDS-DMA of the kernel goes through this statement (M0 is written in the statement that uses it; the compiler has no
    // LDS-DMA of its own here whose M0 it could move across).
#define KEMR_GLDS(VOFF, SBASE, LDSADDR) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" \
                                                     :: "v"(VOFF), "s"(SBASE), "s"(LDSADDR) : "memory")
    const unsigned stage_lds = lds_addr(smem) + wid * 2048;        // W pieces: + region; A pieces use a_dst from the buffer base
    const unsigned buf_lds = lds_addr(smem);
    const unsigned a_dst = (unsigned)(wr * PHALF + (wid & 3) * 4096);      // the wave's 32 rows inside its half's region
    // KEMR_GEMM_SBASE (round 4): the stream offset goes into the SCALAR base (two SALU per call) instead of into each piece's lane
    // offset (one VALU per piece): the load interval then issues no vector-ALU instruction beside its partner's MFMA cluster.
    auto stage_a = [&](int pair, unsigned off, unsigned buf) {             // pair 0 / 1: rows 0-15 / 16-31 of the wave's 32
        if constexpr (KEMR_GEMM_SBASE) {
            const char* const sb = (const char*)p.A + off;
            KEMR_GLDS(a_lane[2 * pair], sb, buf + a_dst + pair * 2048);
            KEMR_GLDS(a_lane[2 * pair + 1], sb, buf + a_dst + pair * 2048 + 1024);
        } else {
            const unsigned v0 = a_lane[2 * pair] + off, v1 = a_lane[2 * pair + 1] + off;
            KEMR_GLDS(v0, p.A, buf + a_dst + pair * 2048);
            KEMR_GLDS(v1, p.A, buf + a_dst + pair * 2048 + 1024);
        }
    };
    auto stage_w = [&](unsigned off, unsigned dst) {
        if constexpr (KEMR_GEMM_SBASE) {
            const char* const sb = (const char*)p.W + off;
            KEMR_GLDS(w_lane0, sb, dst);
            KEMR_GLDS(w_lane1, sb, dst + 1024);
        } else {
            const unsigned v0 = w_lane0 + off, v1 = w_lane1 + off;
            KEMR_GLDS(v0, p.W, dst);
            KEMR_GLDS(v1, p.W, dst + 1024);
        }
    };
    // With the W1 half of K-tile 0 of a tile: that tile's bias (wave 0).  The bias rides with the LAST piece in front of a
    // wait-free stretch, so that nobody waits for it straight after issuing it.
    auto stage_bias = [&]() {
        if constexpr (SIM != 0) return;
        if (__builtin_expect(tw == nt, 0)) {
            if (wid == 0) {
                const unsigned boff = ((unsigned)__builtin_amdgcn_readlane(v_bcol, seqw) + lane * 4) * 4u;
                const unsigned dst = lds_addr(smem) + PBIAS + (seqw & 1) * 1024;
                if (p.bias) KEMR_GLDS(boff, p.bias, dst);
                if (FP8) KEMR_GLDS(boff, p.wscale, dst + (PSCALE - PBIAS));
            }
        }
    };

    const int lrow = lane & 15, lq = lane >> 4;
    const int swz = lrow >> 1;
    // bf16: fragments of the two 32-wide k steps (chunks lq and 4 + lq); fp8: the two halves of ONE 32-byte fragment
    const int co0 = ((FP8 ? 2 * lq : lq) ^ swz) << 4, co1 = ((FP8 ? 2 * lq + 1 : 4 + lq) ^ swz) << 4;
    const int a_off = wr * PHALF + lrow * 128;
    const int b_off = 2 * PHALF + (wc >> 1) * PHALF + ((wc & 1) * 64 + lrow) * 128;

    if (!p.bias && tid < 128) *(float4*)(smem + PBIAS + tid * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (SIM == 2) {
        // per-query constants and the list counters of the workgroup's 256 queries live in the (otherwise unused) epilogue
        // area: {ground-truth score, next float below it, ground-truth id, next float below the top-k threshold}, counter
        if (tid < 256) {
            const int q = t_row_u + tid;
            const bool ok = q < p.M;
            const float g = (ok && p.sim_sgt) ? p.sim_sgt[q] : INFINITY;
            const unsigned u = __float_as_uint(g);
            const float gd = __uint_as_float((u << 1) == 0u ? 0x80000001u : ((u >> 31) ? u + 1u : u - 1u));
            const int gi = (ok && p.sim_gt) ? p.sim_gt[q] : -1;
            *(float4*)(smem + PEPI + tid * 16) = make_float4(g, gd, __int_as_float(gi), ok ? p.simk_taud[q] : INFINITY);
            *(int*)(smem + PEPI + 4096 + tid * 4) = 0;
        }
    }
    // prologue: K-tile 0 complete; in flight behind it (nt >= 2, so all of it belongs to the first tile): both W halves of
    // K-tile 1, or, LONGK, the wave's A pieces of K-tile 1 (the W stream then stays at K-tile 1, the A stream goes on to 2)
    stage_w(ow, stage_lds + 2 * PHALF);
    stage_w(ow1, stage_lds + 3 * PHALF);
    stage_bias();
    stage_a(0, oa, buf_lds);
    stage_a(1, oa, buf_lds);
    advance_w();                               // W stream -> K-tile 1
    advance_a();                               // A stream -> K-tile 1
    if constexpr (LONGK) {
        stage_a(0, oa, buf_lds + PBUF);
        stage_a(1, oa, buf_lds + PBUF);
        advance_a();                           // A stream -> K-tile 2
    } else {
        stage_w(ow, stage_lds + PBUF + 2 * PHALF);
        stage_w(ow1, stage_lds + PBUF + 3 * PHALF);
        advance_w();                           // W stream -> K-tile 2
    }
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      // lgkmcnt: the zero-filled bias area (no-bias launches)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // wr == 1 half runs one interval behind for the whole launch

    int gpar = 0;
    bf16x8 ak0[4], ak1[4], wk0[4], wk1[4];        // bf16 operands: A rows (current 64-row half) / W columns x k step
    bf16x8 ak[8];                                  // LONGK: A rows of all 128 rows, one k step, re-read in place
    fp8x32 af8[4], w08[2], w18[2];                 // fp8 operands (only one set is live, by FP8)
    int one = 0x7f7f7f7f;                          // E8M0 block scales 2^0
    asm volatile("" : "+v"(one));
    auto ld_w8 = [&](fp8x32 (&w8)[2], const char* q) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            w8[ni].lo = *(const i32x4*)(q + ni * 2048 + co0);
            w8[ni].hi = *(const i32x4*)(q + ni * 2048 + co1);
        }
    };
    auto ld_a8 = [&](const char* q) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            af8[mi].lo = *(const i32x4*)(q + mi * 2048 + co0);
            af8[mi].hi = *(const i32x4*)(q + mi * 2048 + co1);
        }
    };
    // bf16 fragment reads as asm, issued from the hooks between MFMAs; LDS byte addresses of the wave's fragment rows in the
    // buffer the NEXT reads come from (dynamic LDS starts at address 0 and every base is below 64 KiB: ^ PBUF switches buffer)
    unsigned vb0 = lds_addr(smem) + b_off + co0, vb1 = lds_addr(smem) + b_off + co1;
    unsigned va0 = lds_addr(smem) + a_off + co0, va1 = lds_addr(smem) + a_off + co1;
#define KEMR_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF) : "memory")
    if constexpr (!FP8) {                          // fragments of the first cluster of K-tile 0
#pragma unroll
        for (int i = 0; i < 4; ++i) { KEMR_DSR(wk0[i], vb0, i * 2048); }
        if constexpr (LONGK) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { KEMR_DSR(ak[i], va0, i * 2048); }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { KEMR_DSR(ak0[i], va0, i * 2048); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // in-kernel stamps (DBG instantiation; wave 0 only): dbg & 64 = cycles per barrier interval of the K loop + epilogue,
    // + dbg & 32 = own work before three of the barriers, dbg & 128 alone = whole-kernel cycles and 100 MHz ticks
    unsigned st_prev = 0, st_slot[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // + dbg & 8: the stamps of wave 4 (the trailing half) instead of wave 0; + dbg & 16: the K-loop slots count only the FIRST K-tile of every tile
    const int st_wave = (DBG && (p.dbg & 8)) ? 4 : 0;
    bool st_count = true;
    const bool stamping = DBG && (p.dbg & 64) && wid == st_wave && p.stamps;
    auto stamp = [&](int slot) {
        if constexpr (DBG) {
            if (stamping) {
                unsigned long long tt;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
                const unsigned now = (unsigned)tt;
                if (st_count) st_slot[slot] += now - st_prev;
                st_prev = now;
            }
        }
    };
    auto stamp_pre = [&](int slot) {           // time since the last interval stamp, st_prev untouched
        if constexpr (DBG) {
            if (stamping && (p.dbg & 32)) {
                unsigned long long tt;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
                if (st_count) st_slot[slot] += (unsigned)tt - st_prev;
            }
        }
    };
    unsigned long long clk0 = 0, rt0 = 0;
    const bool clocking = DBG && (p.dbg & (64 | 128)) && wid == st_wave && p.stamps;
    if constexpr (DBG) {
        if (clocking) {
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0), "=s"(rt0) :: "memory");
            st_prev = (unsigned)clk0;
        }
    }

    f32x4 acc[8][4];
    u32x4 b4[4];                                   // bf16 operands: the tile's bias quads = C of the first MFMA per accumulator
    // Hooks: what a wave issues in the gap behind the I-th MFMA of a cluster.  An MFMA holds the SIMD's vector issue for half
    // of its 16 cycles, so ONE cheap instruction per gap is free and everything beyond it costs its issue time (stamps: a
    // cluster with 25 extra instructions bunched in its first gaps took 420 cycles instead of 256).  Per K-tile there are 64
    // gaps for 24 fragment reads and ~25 instructions of stream bookkeeping: at most one action per gap, A stream in M2 (its
    // offsets were last used in L2), W stream in M4 (last used in L4), behind that cluster's reads.
    // A pin in front of a step orders it behind the MFMA in front of it, the one behind it in front of the next MFMA.
    auto step_a = [&](auto at) {
        constexpr int J = decltype(at)::value;
        if constexpr (J == 0) { KEMR_STREAM_STEP1(ta, seqa); }
        if constexpr (J == 1) { asm volatile("" : "+s"(seqa)); toff_a = __builtin_amdgcn_readlane(v_aoff, seqa); asm volatile("" : "+s"(toff_a)); }
        if constexpr (J == 2) { asm volatile("" : "+s"(oa)); inc_a = oa + 128; asm volatile("" : "+s"(inc_a)); }
        if constexpr (J == 3) { asm volatile("" : "+s"(inc_a)); oa = ta == nt ? toff_a : inc_a; asm volatile("" : "+s"(oa)); }
    };
    auto step_w = [&](auto at) {
        constexpr int J = decltype(at)::value;
        if constexpr (J == 0) { KEMR_STREAM_STEP1(tw, seqw); }
        if constexpr (J == 1) { asm volatile("" : "+s"(seqw)); toff_w = __builtin_amdgcn_readlane(v_woff, seqw); asm volatile("" : "+s"(toff_w)); }
        if constexpr (J == 2) { asm volatile("" : "+s"(ow)); inc_w = ow + 128; asm volatile("" : "+s"(inc_w)); }
        if constexpr (J == 3) { asm volatile("" : "+s"(inc_w)); ow = tw == nt ? toff_w : inc_w; asm volatile("" : "+s"(ow)); }
        if constexpr (J == 4) { asm volatile("" : "+s"(ow)); ow1 = ow + w_half; asm volatile("" : "+s"(ow1)); }
    };
    // The same advances for a K-tile in which neither stream can reach the end of its tile (1 <= t < nt - 3, see the tile loop):
    // a count-down and an add per offset, 5 scalar instructions per K-tile instead of 17 and no v_readlane.
    auto step_a_mid = [&ta, &oa](auto at) {
        constexpr int J = decltype(at)::value;
        if constexpr (J == 0) { asm volatile("s_sub_i32 %0, %0, 1" : "+s"(ta) :: "scc"); }
        if constexpr (J == 1) { asm volatile("s_add_u32 %0, %0, 0x80" : "+s"(oa) :: "scc"); }
    };
    auto step_w_mid = [&tw, &ow, &ow1](auto at) {
        constexpr int J = decltype(at)::value;
        if constexpr (J == 0) { asm volatile("s_sub_i32 %0, %0, 1" : "+s"(tw) :: "scc"); }
        if constexpr (J == 1) { asm volatile("s_add_u32 %0, %0, 0x80" : "+s"(ow) :: "scc"); }
        if constexpr (J == 2) { asm volatile("s_add_u32 %0, %0, 0x80" : "+s"(ow1) :: "scc"); }
    };
    auto nohook = [](auto) {};
    // every cluster reads the NEXT cluster's fragments (their registers were last used two clusters ago)
    auto hook_m1 = [&wk1, &ak1, &vb1, &va1](auto at) {                  // M1 (lo rows, k 0-31) -> M2's fragments: W and lo rows, k 32-63
        constexpr int I = decltype(at)::value;
        if constexpr (I < 4) { KEMR_DSR(wk1[I], vb1, I * 2048); }
        else if constexpr (I < 8) { KEMR_DSR(ak1[I - 4], va1, (I - 4) * 2048); }
    };
    auto hook_m2 = [&ak0, &va0, &step_a](auto at) {                     // M2 -> M3's fragments: hi rows, k 0-31; A stream
        constexpr int I = decltype(at)::value;
        if constexpr (I < 4) { KEMR_DSR(ak0[I], va0, 8192 + I * 2048); }
        else if constexpr (I >= 5 && I <= 13 && (I & 1)) step_a(HookAt<(I - 5) / 2>{});
    };
    auto hook_m3 = [&ak1, &vb0, &vb1, &va0, &va1](auto at) {            // M3 -> M4's fragments: hi rows, k 32-63; then the bases switch buffer
        constexpr int I = decltype(at)::value;
        // (the flips as asm: as C++ the compiler hoists two of them into the wave's LOAD interval, where a vector-ALU instruction costs
        // the partner's MFMA cluster issue slots)
        if constexpr (I < 4) { KEMR_DSR(ak1[I], va1, 8192 + I * 2048); }
        else if constexpr (I == 6) { if constexpr (KEMR_GEMM_FLIPASM) asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(vb0)); else { vb0 ^= PBUF; asm volatile("" : "+v"(vb0)); } }
        else if constexpr (I == 8) { if constexpr (KEMR_GEMM_FLIPASM) asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(vb1)); else { vb1 ^= PBUF; asm volatile("" : "+v"(vb1)); } }
        else if constexpr (I == 10) { if constexpr (KEMR_GEMM_FLIPASM) asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(va0)); else { va0 ^= PBUF; asm volatile("" : "+v"(va0)); } }
        else if constexpr (I == 12) { if constexpr (KEMR_GEMM_FLIPASM) asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(va1)); else { va1 ^= PBUF; asm volatile("" : "+v"(va1)); } }
        static_assert(PBUF == 0x10000, "the asm flips above");
    };
    // M4 -> the next K-tile's M1 fragments: W and lo rows, k 0-31; W stream (both of its offsets were last used in L4)
    auto hook_m4 = [&wk0, &ak0, &vb0, &va0, &gpar, &step_w](auto at) {
        constexpr int I = decltype(at)::value;
        if constexpr (I < 4) { KEMR_DSR(wk0[I], vb0, I * 2048); }
        else if constexpr (I < 8) { KEMR_DSR(ak0[I - 4], va0, (I - 4) * 2048); }
        else if constexpr (I == 8) step_w(HookAt<0>{});
        else if constexpr (I >= 10 && I <= 13) step_w(HookAt<I - 9>{});
        else if constexpr (I == 15) { asm volatile("" : "+s"(gpar)); gpar ^= 1; asm volatile("" : "+s"(gpar)); }
    };
    auto hook_m2_mid = [&ak0, &va0, &step_a_mid](auto at) {
        constexpr int I = decltype(at)::value;
        if constexpr (I < 4) { KEMR_DSR(ak0[I], va0, 8192 + I * 2048); }
        else if constexpr (I == 7) step_a_mid(HookAt<0>{});
        else if constexpr (I == 11) step_a_mid(HookAt<1>{});
    };
    auto hook_m4_mid = [&wk0, &ak0, &vb0, &va0, &gpar, &step_w_mid](auto at) {
        constexpr int I = decltype(at)::value;
        if constexpr (I < 4) { KEMR_DSR(wk0[I], vb0, I * 2048); }
        else if constexpr (I < 8) { KEMR_DSR(ak0[I - 4], va0, (I - 4) * 2048); }
        else if constexpr (I == 9) step_w_mid(HookAt<0>{});
        else if constexpr (I == 11) step_w_mid(HookAt<1>{});
        else if constexpr (I == 13) step_w_mid(HookAt<2>{});
        else if constexpr (I == 15) { asm volatile("" : "+s"(gpar)); gpar ^= 1; asm volatile("" : "+s"(gpar)); }
    };
    // LONGK.  MA (k 0-31 of buffer g): reads the k 32-63 fragments of the same buffer -- W into the other register set in the first
    // gaps, A row tile mi in place in the gap behind its fourth MFMA --, advances the W stream (its offsets were last used in LA)
    // and then points the k 0-31 bases at the next buffer.  MB (k 32-63): the same with the k 0-31 fragments of buffer g+1, the A
    // stream (last used in LB) and the k 32-63 bases; gpar flips last.  At most one action per gap.
    auto hook_la = [&wk1, &ak, &vb0, &vb1, &va0, &va1, &step_w](auto at) {
        constexpr int I = decltype(at)::value;
        if constexpr ((I & 3) == 3) { KEMR_DSR(ak[I >> 2], va1, (I >> 2) * 2048); }
        else if constexpr (I < 3) { KEMR_DSR(wk1[I], vb1, I * 2048); }
        else if constexpr (I == 4) { KEMR_DSR(wk1[3], vb1, 3 * 2048); }
        else if constexpr (I == 6) step_w(HookAt<0>{});
        else if constexpr (I == 9) step_w(HookAt<1>{});
        else if constexpr (I == 10) step_w(HookAt<2>{});
        else if constexpr (I == 13) step_w(HookAt<3>{});
        else if constexpr (I == 14) step_w(HookAt<4>{});
        else if constexpr (I == 17) { vb0 ^= PBUF; asm volatile("" : "+v"(vb0)); }
        else if constexpr (I == 18) { va0 ^= PBUF; asm volatile("" : "+v"(va0)); }
    };
    auto hook_lb = [&wk0, &ak, &vb0, &vb1, &va0, &va1, &gpar, &step_a](auto at) {
        constexpr int I = decltype(at)::value;
        if constexpr ((I & 3) == 3) { KEMR_DSR(ak[I >> 2], va0, (I >> 2) * 2048); }
        else if constexpr (I < 3) { KEMR_DSR(wk0[I], vb0, I * 2048); }
        else if constexpr (I == 4) { KEMR_DSR(wk0[3], vb0, 3 * 2048); }
        else if constexpr (I == 6) step_a(HookAt<0>{});
        else if constexpr (I == 9) step_a(HookAt<1>{});
        else if constexpr (I == 10) step_a(HookAt<2>{});
        else if constexpr (I == 13) step_a(HookAt<3>{});
        else if constexpr (I == 17) { vb1 ^= PBUF; asm volatile("" : "+v"(vb1)); }
        else if constexpr (I == 18) { va1 ^= PBUF; asm volatile("" : "+v"(va1)); }
        else if constexpr (I == 21) { asm volatile("" : "+s"(gpar)); gpar ^= 1; asm volatile("" : "+s"(gpar)); }
    };
    // fp8: the reads stay in the load intervals; the streams use the same clusters
    auto hook8_m2 = [&step_a](auto at) { constexpr int I = decltype(at)::value; if constexpr (I >= 1 && I <= 9 && (I & 1)) step_a(HookAt<(I - 1) / 2>{}); };
    auto hook8_m4 = [&gpar, &step_w](auto at) {
        constexpr int I = decltype(at)::value;
        if constexpr (I >= 1 && I <= 9 && (I & 1)) step_w(HookAt<(I - 1) / 2>{});
        else if constexpr (I == 11) { asm volatile("" : "+s"(gpar)); gpar ^= 1; asm volatile("" : "+s"(gpar)); }
    };
    auto hook8_m2_mid = [&step_a_mid](auto at) { constexpr int I = decltype(at)::value; if constexpr (I == 3) step_a_mid(HookAt<0>{}); else if constexpr (I == 7) step_a_mid(HookAt<1>{}); };
    auto hook8_m4_mid = [&gpar, &step_w_mid](auto at) {
        constexpr int I = decltype(at)::value;
        if constexpr (I == 3) step_w_mid(HookAt<0>{});
        else if constexpr (I == 5) step_w_mid(HookAt<1>{});
        else if constexpr (I == 7) step_w_mid(HookAt<2>{});
        else if constexpr (I == 11) { asm volatile("" : "+s"(gpar)); gpar ^= 1; asm volatile("" : "+s"(gpar)); }
    };

    // One K-tile.  LDS-DMA per interval: L1 A0(g+1), L2 A1(g+1), L3 W0(g+2), L4 W1(g+2) [+ bias]; the wait that needs K-tile
    // g+1 complete sits at the end of M3 (in front of the barrier in front of M4, whose hooks read K-tile g+1) and leaves
    // W0(g+2) in flight.
    // PRE (round 4): vmcnt retires in order, so a wait for a piece that was issued BEHIND a tile's 16 stores also waits for the stores'
    // acknowledgements -- and every workgroup of the chip stores its tile at the same moment (stamps: the first K-tile's vmcnt(4) cost
    // +150 .. +840 cycles per tile).  The store epilogue therefore stages the A pieces of the next tile's K-tile 1 (normally L1 / L2 of
    // its first K-tile; their LDS region, the last K-tile's A half, is free behind that K-tile's M3) BEFORE its stores, the first K-tile
    // skips them, and its two waits let the 16 stores stay in flight: W(1) is followed by 4 A pieces + 16 stores + W0(2) = 22
    // operations, A(1) by 16 stores + W(2) = 20.  The next wait that covers the stores is K-tile 1's, fourteen intervals behind them.
    constexpr bool PRE_OK = KEMR_GEMM_PRESTAGE && !LONGK && SIM == 0 && (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16);
    bool pre = false;                              // this tile's first K-tile finds its A(1) pieces staged
    // MID: a K-tile in which no stream wraps (no bias piece either: that rides with a tile's first W piece).
    auto ktile = [&](auto first_c, auto mid_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr bool MID = decltype(mid_c)::value;
        if constexpr (DBG) st_count = FIRST || !(p.dbg & 16);
        const unsigned buf_this = stage_lds + gpar * PBUF;
        const unsigned abuf_next = buf_lds + (gpar ^ 1) * PBUF;
        if constexpr (FP8) {
            const char* sa = smem + gpar * PBUF + a_off;
            const char* sb = smem + gpar * PBUF + b_off;
            ld_w8(w08, sb);
            ld_a8(sa);
            if (!(FIRST && PRE_OK && pre)) stage_a(0, oa, abuf_next);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w08[0]), "+v"(w08[1]), "+v"(af8[0]), "+v"(af8[1]), "+v"(af8[2]), "+v"(af8[3]) :: "memory");
            __builtin_amdgcn_s_barrier();
            stamp(0);
            quad8<0, 0, FIRST>(acc, af8, w08, one, nohook);
            __builtin_amdgcn_s_barrier();
            stamp(1);
            ld_w8(w18, sb + 4096);
            if (!(FIRST && PRE_OK && pre)) stage_a(1, oa, abuf_next);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w18[0]), "+v"(w18[1]) :: "memory");     // the reads are done in front of the barrier: W may be re-staged behind the next one
            __builtin_amdgcn_s_barrier();
            stamp(2);
            if constexpr (MID) quad8<0, 1, FIRST>(acc, af8, w18, one, hook8_m2_mid);
            else quad8<0, 1, FIRST>(acc, af8, w18, one, hook8_m2);
            __builtin_amdgcn_s_barrier();
            stamp(3);
            ld_a8(sa + 8192);
            stage_w(ow, buf_this + 2 * PHALF);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af8[0]), "+v"(af8[1]), "+v"(af8[2]), "+v"(af8[3]) :: "memory");
            __builtin_amdgcn_s_barrier();
            stamp(4);
            quad8<1, 1, FIRST>(acc, af8, w18, one, nohook);
            if (FIRST && PRE_OK && pre) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // W(g+1) landed (and everything older: a tile's stores); all waves agree on it at this barrier
            __builtin_amdgcn_s_barrier();
            stamp(5);
            stage_w(ow1, buf_this + 3 * PHALF);
            if (FIRST && PRE_OK && pre) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // the half's own A(g+1) pieces landed; W(g+2) stays in flight
            if constexpr (!MID) stage_bias();
            __builtin_amdgcn_s_barrier();
            stamp(6);
            if constexpr (MID) quad8<1, 0, FIRST>(acc, af8, w08, one, hook8_m4_mid);
            else quad8<1, 0, FIRST>(acc, af8, w08, one, hook8_m4);
            __builtin_amdgcn_s_barrier();
            stamp(7);
        } else if constexpr (LONGK) {
            stage_w(ow, stage_lds + (gpar ^ 1) * PBUF + 2 * PHALF);
            stage_w(ow1, stage_lds + (gpar ^ 1) * PBUF + 3 * PHALF);
            stage_bias();
            stamp_pre(10);
            __builtin_amdgcn_s_barrier();
            stamp(0);
            cluster32<FIRST>(acc, ak, wk0, b4, hook_la);
            // K-tile g+1 has landed: this wave's A pieces from LB(g-1) and W pieces from LA(g) (and everything older: a tile's
            // stores); the fragment reads of the hooks have returned (orders them in front of LB's re-staging of the A region)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            stamp_pre(11);
            __builtin_amdgcn_s_barrier();
            stamp(1);
            stage_a(0, oa, buf_lds + gpar * PBUF);
            stage_a(1, oa, buf_lds + gpar * PBUF);
            __builtin_amdgcn_s_barrier();
            stamp(2);
            cluster32<false>(acc, ak, wk1, b4, hook_lb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stamp(3);
        } else {
            if (!(FIRST && PRE_OK && pre)) stage_a(0, oa, abuf_next);
            stamp_pre(10);
            __builtin_amdgcn_s_barrier();
            stamp(0);
            cluster<0, FIRST>(acc, ak0, wk0, b4, hook_m1);
            stamp_pre(11);
            __builtin_amdgcn_s_barrier();
            stamp(1);
            if (!(FIRST && PRE_OK && pre)) stage_a(1, oa, abuf_next);
            __builtin_amdgcn_s_barrier();
            stamp(2);
            if constexpr (MID) cluster<0, false>(acc, ak1, wk1, b4, hook_m2_mid);
            else cluster<0, false>(acc, ak1, wk1, b4, hook_m2);
            __builtin_amdgcn_s_barrier();
            stamp(3);
            stage_w(ow, buf_this + 2 * PHALF);
            __builtin_amdgcn_s_barrier();
            stamp(4);
            cluster<1, FIRST>(acc, ak0, wk0, b4, hook_m3);
            if (FIRST && PRE_OK && pre) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // W(g+1) landed (and everything older: a tile's stores); all waves agree on it at this barrier
            __builtin_amdgcn_s_barrier();
            stamp(5);
            stage_w(ow1, buf_this + 3 * PHALF);
            stamp_pre(12);
            if (FIRST && PRE_OK && pre) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // the half's own A(g+1) pieces landed; W(g+2) stays in flight
            stamp_pre(13);
            if constexpr (!MID) stage_bias();
            __builtin_amdgcn_s_barrier();
            stamp(6);
            if constexpr (MID) cluster<1, false>(acc, ak1, wk1, b4, hook_m4_mid);
            else cluster<1, false>(acc, ak1, wk1, b4, hook_m4);
            __builtin_amdgcn_s_barrier();
            stamp(7);
        }
    };

    // SIM: per-lane state of the 8 queries a lane holds (rows wr * 128 + mi * 16 + lrow of the query tile)
    float sgt[8], sgd[8];                          // ground-truth score, and the next float below it (s > sgd  <=>  s >= sgt)
    int gtid[8], cnt[8];
    int k_ovf = 0;
    if constexpr (SIM != 0) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) { b4[ni] = u32x4{0u, 0u, 0u, 0u}; asm volatile("" : "+v"(b4[ni])); }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) cnt[mi] = 0;
    }
    if constexpr (SIM == 1) {
        int il = lane;
        asm volatile("" : "+v"(il));
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int q = (t_row_u) + wr * 128 + mi * 16 + (il & 15);
            const bool ok = q < p.M;
            gtid[mi] = ok ? p.sim_gt[q] : -1;
            sgt[mi] = ok ? p.sim_sgt[q] : INFINITY;
            const unsigned u = __float_as_uint(sgt[mi]);
            // next float below (finite input, NaN excluded by the host): -0 / +0 -> the smallest negative number
            sgd[mi] = __uint_as_float((u << 1) == 0u ? 0x80000001u : ((u >> 31) ? u + 1u : u - 1u));
        }
    }
    for (int seq = 0; seq < ntl; ++seq) {
        if constexpr (!FP8 && SIM == 0) {
            // The tile's bias came in with its first W0 piece, which the wait + barrier that closed the previous K-tile (or the
            // prologue) cover; a lane's accumulators cover columns wc * 64 + ni * 16 + lq * 4 .. + 3.
            int il = lane;
            asm volatile("" : "+v"(il));
            const unsigned bias_r = lds_addr(smem + PBIAS) + (seq & 1) * 1024 + (wc * 64 + (il >> 4) * 4) * 4;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) b4[ni] = lds_read_b128(bias_r + ni * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]) :: "memory");
        }
        // The A stream reaches the end of its tile in K-tile nt - 2 and the W stream in K-tile nt - 3 (t = 0 as well when nt <= 3):
        // the K-tiles in between advance the streams without the wrap arithmetic.
        ktile(std::true_type{}, std::false_type{});
        if constexpr (KEMR_GEMM_MID && !LONGK) {
            const int t_mid = nt - 3;
            for (int t = 1; t < t_mid; ++t) ktile(std::false_type{}, std::true_type{});
            for (int t = t_mid < 1 ? 1 : t_mid; t < nt; ++t) ktile(std::false_type{}, std::false_type{});
        } else {
            for (int t = 1; t < nt; ++t) ktile(std::false_type{}, std::false_type{});
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // asm MFMA result -> VALU read (>= 12 wait states)
        st_count = true;

        if constexpr (SIM != 0) {
            // ---- scan: acc[mi][ni][r] = score(query wr*128 + mi*16 + lrow, candidate wc*64 + ni*16 + lq*4 + r of this gallery tile).
            // Order rule of the whole path: a candidate ranks ahead of the ground truth iff  s > sgt  or  (s == sgt and id < gt).
            // All of a lane's 16 candidates lie on one side of gt unless gt falls into the lane's 52-id window, so one threshold per
            // query does (sgd for "ids below gt": >= as >); the rare mixed window and a gallery's ragged last tile are recounted
            // element by element.
            // Both halves scan in the SAME barrier interval: the leading half sits out the interval of the trailing half's last
            // cluster, and the trailing half takes its next staging interval on its own afterwards (one extra barrier per tile
            // for each).  Scanning one after the other -- each against one 256-cycle cluster of the other half -- left the
            // SIMD to a single wave's dependent VALU chain at ~9 cycles per instruction (stamps, round 2); two waves interleave.
            if (wr == 0) __builtin_amdgcn_s_barrier();
            stamp(8);
            int sl = lane;
            asm volatile("" : "+v"(sl));
            if constexpr (SIM == 3) {
                // group maxima: the best score of every query among the 64 candidates of this wave's column block (a lane's 16,
                // then the 4 lanes that share the query row) -> simk_scores[query][tile * 4 + wc]
                const int groups = (p.N >> 8) * 4, grp = (sim_tb + seq) * 4 + wc;
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    float m = acc[mi][0][0];
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[mi][ni][r]);
                    m = fmaxf(m, __shfl_xor(m, 16));
                    m = fmaxf(m, __shfl_xor(m, 32));
                    const int q = t_row_u + wr * 128 + mi * 16 + (sl & 15);
                    if (sl < 16 && q < p.M) p.simk_scores[(size_t)q * groups + grp] = m;
                }
                if (wr == 1) __builtin_amdgcn_s_barrier();
                continue;
            }
            const int cb = p.sim_gbase + ((sim_tb + seq) << 8) + wc * 64 + (sl >> 4) * 4;       // global id of the lane's first candidate
            const bool ragged = ((sim_tb + seq + 1) << 8) > p.sim_ng;                            // wave-uniform: some candidates do not exist
            const int n_end = p.sim_gbase + p.sim_ng;
            unsigned hitmask = 0;
            int slot[8];
            const float4* const qs = (const float4*)(smem + PEPI) + wr * 128 + (sl & 15);      // SIM == 2: the lane's query rows, + mi * 16
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                float g_s, g_d, taud = 0.f;
                int g_i;
                if constexpr (SIM == 2) {
                    const float4 st = qs[mi * 16];
                    g_s = st.x; g_d = st.y; g_i = __float_as_int(st.z); taud = st.w;
                } else {
                    g_s = sgt[mi]; g_d = sgd[mi]; g_i = gtid[mi];
                }
                const int d = g_i - cb;
                const float thr = d > 51 ? g_d : g_s;
                const bool mixed = (unsigned)d <= 51u;
                int c = 0;
                if (SIM == 1 || p.sim_gt) {           // (uniform) a search without a ground truth has nothing to count
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) c += acc[mi][ni][r] > thr ? 1 : 0;
                if (__builtin_amdgcn_ballot_w64(mixed) != 0 || ragged) {
                    int ce = 0;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int e = ni * 16 + r;
                            const float sc = acc[mi][ni][r];
                            const bool ahead = (cb + e < n_end) && e != d && (sc > g_s || (sc == g_s && e < d));
                            ce += ahead ? 1 : 0;
                        }
                    c = (mixed || ragged) ? ce : c;
                }
                }
                cnt[mi] += c;
                if constexpr (SIM == 2) {
                    // top-k candidates: one test per (lane, query) on the maximum of the lane's 16 scores; a hit takes a slot
                    // of the query's list (LDS counter: the 4 x 4 lanes that share a query row race for it)
                    float m = acc[mi][0][0];
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[mi][ni][r]);
                    slot[mi] = 0;
                    if (m > taud) {
                        hitmask |= 1u << mi;
                        slot[mi] = atomicAdd((int*)(smem + PEPI + 4096) + wr * 128 + mi * 16 + (sl & 15), 1);
                    }
                }
            }
            if constexpr (SIM == 2) {
                if (__builtin_amdgcn_ballot_w64(hitmask != 0u) != 0) {
                    const int chunk = sim_chunk;
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        if ((hitmask >> mi) & 1u) {
                            const int q = t_row_u + wr * 128 + mi * 16 + (sl & 15);
                            if (slot[mi] < p.simk_cap) {
                                const size_t rec = ((size_t)q * p.sim_nchunks + chunk) * p.simk_cap + slot[mi];
                                f32x4* dst = (f32x4*)(p.simk_scores + rec * 16);
#pragma unroll
                                for (int ni = 0; ni < 4; ++ni) dst[ni] = acc[mi][ni];
                                p.simk_base[rec] = cb;
                            } else {
                                k_ovf = 1;
                            }
                        }
                    }
                }
            }
            if (wr == 1) __builtin_amdgcn_s_barrier();
        } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
            // ---- fp32 residual stream, updated in place in the ACCUMULATOR domain: x += A.W^T + bias (the bias is in the accumulators
            // since their first MFMA).  A lane's quad acc[mi][ni] is 16 contiguous bytes of row mi*16 + lrow (columns ni*16 + lq*4
            // .. +3), the 4 lanes of a row make a 64-byte segment and ni = 0 / 1 (2 / 3) the two halves of a 128-byte line: no LDS
            // transposition, no conversion -- per 16-row pass 4 loads, 16 adds, 4 stores (the bf16 store epilogue: 8 converts, 4 LDS
            // writes, 2 LDS reads, 2 stores and the LDS round trip; its residual-add form 48 more VALU).  The x quads come in
            // through a window of two passes; vmcnt retires loads and stores in issue order, so the wait in front of pass p's adds
            // counts what was issued behind its loads: the next pass's loads and the previous pass's stores.
            int el = lane;
            asm volatile("" : "+v"(el));
            const char* const ctile = (const char*)p.C + __builtin_amdgcn_readlane(v_coff, seq) + ((size_t)(wr * 128) * p.ldc + wc * 64) * 4;     // wave-uniform
            const unsigned step16 = (unsigned)p.ldc * 64u;                    // 16 rows in bytes
            unsigned xoff = (unsigned)((el & 15) * p.ldc + (el >> 4) * 4) * 4u, soff = xoff;
            stamp(8);
            f32x4 xa[4], xb[4];
    #define KEMR_XLOAD4(X)                                                                                                          \
            do {                                                                                                                    \
                asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:64\n\t"                       \
                             "global_load_dwordx4 %2, %4, %5 offset:128\n\tglobal_load_dwordx4 %3, %4, %5 offset:192"               \
                             : "=&v"(X[0]), "=&v"(X[1]), "=&v"(X[2]), "=&v"(X[3]) : "v"(xoff), "s"(ctile) : "memory");              \
                xoff += step16;                                                                                                     \
            } while (0)
    #define KEMR_XPASS(MI, X, VM)                                                                                                   \
            do {                                                                                                                    \
                asm volatile("s_waitcnt vmcnt(" #VM ")" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]) :: "memory");               \
                _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[MI][ni] += X[ni];                                              \
                asm volatile("global_store_dwordx4 %0, %1, %5 nt\n\tglobal_store_dwordx4 %0, %2, %5 offset:64 nt\n\t"              \
                             "global_store_dwordx4 %0, %3, %5 offset:128 nt\n\tglobal_store_dwordx4 %0, %4, %5 offset:192 nt\n\ts_nop 1" \
                             :: "v"(soff), "v"(acc[MI][0]), "v"(acc[MI][1]), "v"(acc[MI][2]), "v"(acc[MI][3]), "s"(ctile) : "memory"); \
                soff += step16;                                                                                                     \
            } while (0)
            // VMEM order: L0 L1 | S0 L2 | S1 L3 | S2 L4 | S3 L5 | S4 L6 | S5 L7 | S6 | S7   (L = 4 loads, S = 4 stores)
            KEMR_XLOAD4(xa); KEMR_XLOAD4(xb);
            KEMR_XPASS(0, xa, 4); KEMR_XLOAD4(xa);
            KEMR_XPASS(1, xb, 8); KEMR_XLOAD4(xb);
            KEMR_XPASS(2, xa, 8); KEMR_XLOAD4(xa);
            KEMR_XPASS(3, xb, 8); KEMR_XLOAD4(xb);
            KEMR_XPASS(4, xa, 8); KEMR_XLOAD4(xa);
            KEMR_XPASS(5, xb, 8); KEMR_XLOAD4(xb);
            KEMR_XPASS(6, xa, 8);
            KEMR_XPASS(7, xb, 4);
    #undef KEMR_XLOAD4
    #undef KEMR_XPASS
        } else {
            // ---- epilogue (lane constants behind an opaque copy of `lane`: recomputed here, not kept across the K loop)
            // CONC: both halves run their epilogues in the SAME barrier interval (as the similarity scans do): the leading half
            // sits out the interval of the trailing half's last cluster, the trailing half takes its next staging interval alone
            // afterwards.  Each wave then has only its own 2 KiB area (single-buffered passes); two waves per SIMD interleave.
            if (CONC && wr == 0) __builtin_amdgcn_s_barrier();
            if constexpr (PRE_OK) {                                              // (PRE above; the A stream points at K-tile 1 of the next tile since the last K-tile's M2)
                if (!(DBG && (p.dbg & 1))) {
                    const unsigned abuf_pre = buf_lds + (gpar ^ 1) * PBUF;
                    stage_a(0, oa, abuf_pre);
                    stage_a(1, oa, abuf_pre);
                    pre = true;
                }
            }
            int el = lane;
            asm volatile("" : "+v"(el));
            const int erow = el & 15, eq = el >> 4;
            const int er = el >> 3, ec = el & 7;                              // read-back: row er (+8), chunk ec
            // 16 rows x 128 B per area, chunk ^= row & 7.  Even passes use the area of wave (wid & 3), odd passes that of wave
            // (wid & 3) + 4 (8 KiB further): the wave's own and its partner's, which is idle (header)
            const unsigned epi0 = lds_addr(smem + PEPI) + (wid & 3) * 2048 + (CONC ? wr * 8192 : 0);
            const unsigned epi_w = epi0 + erow * 128 + (((eq >> 1) ^ (erow & 7)) << 4) + (eq & 1) * 8;
            const unsigned epi_r = epi0 + er * 128 + ((ec ^ er) << 4);        // rows er and er + 8: (er + 8) & 7 == er
            const unsigned ew0 = epi_w, ew1 = epi_w ^ 32, ew2 = epi_w ^ 64, ew3 = epi_w ^ 96;      // chunk (ni*2 + (eq>>1)) ^ (erow & 7): ni flips bits 5-6
            const char* const ctile = (const char*)p.C + __builtin_amdgcn_readlane(v_coff, seq) + ((size_t)(wr * 128) * p.ldc + wc * 64) * 2;     // wave-uniform
            unsigned voff = (unsigned)(er * p.ldc + ec * 8) * 2u;            // + 16 rows per pass
            const unsigned step8 = (unsigned)p.ldc * 16u;                     // 8 rows in bytes
            u32x4 bias[4], wsc[4];
            if constexpr (FP8) {
                const unsigned bias_r = lds_addr(smem + PBIAS) + (seq & 1) * 1024 + (wc * 64 + eq * 4) * 4;
    #pragma unroll
                for (int ni = 0; ni < 4; ++ni) bias[ni] = lds_read_b128(bias_r + ni * 64);
    #pragma unroll
                for (int ni = 0; ni < 4; ++ni) wsc[ni] = lds_read_b128(bias_r + (PSCALE - PBIAS) + ni * 64);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wsc[0]), "+v"(wsc[1]), "+v"(wsc[2]), "+v"(wsc[3]) :: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]) :: "memory");
            }
            stamp(8);
            u32x2 o[4];
            auto pack = [&](const f32x4 (&a4)[4]) {
    #pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    f32x4 v = a4[ni];
                    if constexpr (FP8) {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], __uint_as_float(wsc[ni][r]), __uint_as_float(bias[ni][r]));
                    }
                    if constexpr (EPI == EPI_BIAS_QGELU_BF16) {
                        const f32x2_t g0 = quick_gelu2(f32x2_t{v[0], v[1]}), g1 = quick_gelu2(f32x2_t{v[2], v[3]});
                        v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
                    }
                    o[ni][0] = pack_bf16x2(v[0], v[1]);
                    o[ni][1] = pack_bf16x2(v[2], v[3]);
                }
            };
            u32x4 dA0, dA1, dB0, dB1;           // read-back of the pass in flight in each of the two areas
            // EPI_BIAS_RESADD_BF16: the 16-byte chunks of the residual stream that the passes' stores will overwrite come in through
            // a rolling window of four passes (loaded at the offsets the stores use); vmcnt retires loads and stores in order, so the
            // wait in front of pass p's add counts what was issued behind its pair: the younger pairs and the older passes' stores.
            constexpr bool RES = EPI == EPI_BIAS_RESADD_BF16;
            u32x4 xr0[4], xr1[4];
            unsigned xoff = voff;
    #define KEMR_XLOAD(P)                                                                                                           \
            do {                                                                                                                    \
                if constexpr (RES) {                                                                                                \
                    asm volatile("global_load_dwordx4 %0, %2, %4\n\tglobal_load_dwordx4 %1, %3, %4"                                  \
                                 : "=&v"(xr0[(P) & 3]), "=&v"(xr1[(P) & 3]) : "v"(xoff), "v"(xoff + step8), "s"(ctile) : "memory"); \
                    xoff += 2 * step8;                                                                                              \
                }                                                                                                                   \
            } while (0)
            auto add8 = [](u32x4 d, u32x4 x) {          // bf16 x 8 + bf16 x 8 in fp32, rounded once (RNE)
                u32x4 r;
    #pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const f32x2_t a = {__uint_as_float(d[w] << 16), __uint_as_float(d[w] & 0xffff0000u)};
                    const f32x2_t b = {__uint_as_float(x[w] << 16), __uint_as_float(x[w] & 0xffff0000u)};
                    const f32x2_t c = a + b;
                    r[w] = pack_bf16x2(c.x, c.y);
                }
                return r;
            };
            // one pass through LDS: 4 writes of the packed quads, 2 reads of whole 16-byte chunks; area B is 8 KiB behind area A
    #define KEMR_LDS_PASS(D0, D1, OFF, OFF1)                                                                                        \
            asm volatile("ds_write_b64 %2, %6 offset:" #OFF "\n\tds_write_b64 %3, %7 offset:" #OFF "\n\tds_write_b64 %4, %8 offset:" #OFF "\n\t" \
                         "ds_write_b64 %5, %9 offset:" #OFF "\n\tds_read_b128 %0, %10 offset:" #OFF "\n\tds_read_b128 %1, %10 offset:" #OFF1 \
                         : "=&v"(D0), "=&v"(D1)                                                                                     \
                         : "v"(ew0), "v"(ew1), "v"(ew2), "v"(ew3), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(epi_r) : "memory")
            // wait for a pass's two reads (WAIT = LDS operations of the next pass issued behind them), then its 2 full-line stores.
            // Non-temporal: C is not read again by this kernel, and written as ordinary write-back lines the 0.1-0.5 GB of a
            // launch evicts the A / W panels the K loops live on from L2 (round 1, encoder shapes: plain stores +27 % time on QKV,
            // +17 % on fc1 over no stores at all; `nt` stores +0 % / +6 %; `sc1` write-through +13 %).  Exactly 16 stores per lane
            // and tile: the vmcnt bookkeeping in the header counts them.  dbg 1 / 4 (DBG instantiation, tools/): stores dropped / plain.
    #define KEMR_RES_ADD(D0, D1, P, VM)                                                                                             \
            do {                                                                                                                    \
                if constexpr (RES) {                                                                                                \
                    asm volatile("s_waitcnt vmcnt(" #VM ")" : "+v"(xr0[(P) & 3]), "+v"(xr1[(P) & 3]) :: "memory");                  \
                    D0 = add8(D0, xr0[(P) & 3]);                                                                                    \
                    D1 = add8(D1, xr1[(P) & 3]);                                                                                    \
                }                                                                                                                   \
            } while (0)
    #define KEMR_STORE_PASS(D0, D1, WAIT) KEMR_STORE_PASS_(D0, D1, WAIT, 0)
    #define KEMR_STORE_PASS_(D0, D1, WAIT, ODD)                                                                                     \
            do {                                                                                                                    \
                asm volatile("s_waitcnt lgkmcnt(" #WAIT ")" : "+v"(D0), "+v"(D1) :: "memory");                                      \
                const unsigned voff8 = voff + step8;                                                                                \
                if (!DBG || !(p.dbg & 1)) {                                                                                         \
                    if ((DBG && (p.dbg & 4)) || (KEMR_GEMM_STORE_MIX == 3 && (ODD)) || KEMR_GEMM_STORE_MIX == 4)                     \
                        asm volatile("global_store_dwordx4 %0, %1, %4\n\tglobal_store_dwordx4 %2, %3, %4\n\ts_nop 1"               \
                                     :: "v"(voff), "v"(D0), "v"(voff8), "v"(D1), "s"(ctile) : "memory");                            \
                    else                                                                                                            \
                        asm volatile("global_store_dwordx4 %0, %1, %4 nt\n\tglobal_store_dwordx4 %2, %3, %4 nt\n\ts_nop 1"         \
                                     :: "v"(voff), "v"(D0), "v"(voff8), "v"(D1), "s"(ctile) : "memory");                            \
                }                                                                                                                   \
                voff = voff8 + step8;                                                                                               \
            } while (0)
            if constexpr (CONC) {
                // one area: a pass's read-back must have landed before the next pass's writes (the wait of its stores covers it);
                // the next pass's conversion (and QuickGELU) is issued in front of that wait
                pack(acc[0]); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                pack(acc[1]); KEMR_STORE_PASS_(dA0, dA1, 0, 0); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                pack(acc[2]); KEMR_STORE_PASS_(dA0, dA1, 0, 1); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                pack(acc[3]); KEMR_STORE_PASS_(dA0, dA1, 0, 0); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                pack(acc[4]); KEMR_STORE_PASS_(dA0, dA1, 0, 1); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                pack(acc[5]); KEMR_STORE_PASS_(dA0, dA1, 0, 0); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                pack(acc[6]); KEMR_STORE_PASS_(dA0, dA1, 0, 1); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                pack(acc[7]); KEMR_STORE_PASS_(dA0, dA1, 0, 0); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                KEMR_STORE_PASS_(dA0, dA1, 0, 1);
            } else if constexpr (RES) {
                // VMEM order: L0 L1 L2 L3 | S0 L4 | S1 L5 | S2 L6 | S3 L7 | S4 | S5 | S6 | S7 (a pair = 2 loads, a pass = 2 stores)
                KEMR_XLOAD(0); KEMR_XLOAD(1); KEMR_XLOAD(2); KEMR_XLOAD(3);
                pack(acc[0]); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                pack(acc[1]); KEMR_LDS_PASS(dB0, dB1, 8192, 9216);
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(dA0), "+v"(dA1) :: "memory"); KEMR_RES_ADD(dA0, dA1, 0, 6);
                KEMR_STORE_PASS(dA0, dA1, 6); KEMR_XLOAD(4); pack(acc[2]); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(dB0), "+v"(dB1) :: "memory"); KEMR_RES_ADD(dB0, dB1, 1, 8);
                KEMR_STORE_PASS(dB0, dB1, 6); KEMR_XLOAD(5); pack(acc[3]); KEMR_LDS_PASS(dB0, dB1, 8192, 9216);
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(dA0), "+v"(dA1) :: "memory"); KEMR_RES_ADD(dA0, dA1, 2, 10);
                KEMR_STORE_PASS(dA0, dA1, 6); KEMR_XLOAD(6); pack(acc[4]); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(dB0), "+v"(dB1) :: "memory"); KEMR_RES_ADD(dB0, dB1, 3, 12);
                KEMR_STORE_PASS(dB0, dB1, 6); KEMR_XLOAD(7); pack(acc[5]); KEMR_LDS_PASS(dB0, dB1, 8192, 9216);
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(dA0), "+v"(dA1) :: "memory"); KEMR_RES_ADD(dA0, dA1, 4, 12);
                KEMR_STORE_PASS(dA0, dA1, 6); pack(acc[6]); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(dB0), "+v"(dB1) :: "memory"); KEMR_RES_ADD(dB0, dB1, 5, 10);
                KEMR_STORE_PASS(dB0, dB1, 6); pack(acc[7]); KEMR_LDS_PASS(dB0, dB1, 8192, 9216);
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(dA0), "+v"(dA1) :: "memory"); KEMR_RES_ADD(dA0, dA1, 6, 8);
                KEMR_STORE_PASS(dA0, dA1, 6);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dB0), "+v"(dB1) :: "memory"); KEMR_RES_ADD(dB0, dB1, 7, 6);
                KEMR_STORE_PASS(dB0, dB1, 0);
            } else {
            pack(acc[0]); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
            pack(acc[1]); KEMR_LDS_PASS(dB0, dB1, 8192, 9216);
            KEMR_STORE_PASS(dA0, dA1, 6); pack(acc[2]); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
            KEMR_STORE_PASS_(dB0, dB1, 6, 1); pack(acc[3]); KEMR_LDS_PASS(dB0, dB1, 8192, 9216);
            KEMR_STORE_PASS(dA0, dA1, 6); pack(acc[4]); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
            KEMR_STORE_PASS_(dB0, dB1, 6, 1); pack(acc[5]); KEMR_LDS_PASS(dB0, dB1, 8192, 9216);
            KEMR_STORE_PASS(dA0, dA1, 6); pack(acc[6]); KEMR_LDS_PASS(dA0, dA1, 0, 1024);
            KEMR_STORE_PASS_(dB0, dB1, 6, 1); pack(acc[7]); KEMR_LDS_PASS(dB0, dB1, 8192, 9216);
            KEMR_STORE_PASS(dA0, dA1, 6);
            KEMR_STORE_PASS_(dB0, dB1, 0, 1);
            }
            if (CONC && wr == 1) __builtin_amdgcn_s_barrier();
    #undef KEMR_LDS_PASS
    #undef KEMR_STORE_PASS
    #undef KEMR_STORE_PASS_
    #undef KEMR_RES_ADD
    #undef KEMR_XLOAD
        }
        stamp(9);
    }
#undef KEMR_DSR
#undef KEMR_GLDS
#undef KEMR_STREAM_STEP1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the streams' pieces behind the last K-tile land in LDS: before the exit
    if constexpr (SIM != 0) {
        int fl = lane;
        asm volatile("" : "+v"(fl));
        if (SIM == 1 || (SIM == 2 && p.sim_gt)) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int q = t_row_u + wr * 128 + mi * 16 + (fl & 15);
                if (q < p.M && cnt[mi]) atomicAdd(p.sim_ahead + q, cnt[mi]);
            }
        }
        if constexpr (SIM == 2) {
            if (k_ovf) atomicOr(p.simk_flag, 1);
        }
    }
    if constexpr (DBG) {
        if (clocking && lane == 0) {
#pragma unroll
            for (int i = 0; i < 14; ++i) p.stamps[blockIdx.x * 16 + i] = st_slot[i];
            p.stamps[blockIdx.x * 16 + 14] = (unsigned)ntl;       // tiles of this workgroup
            p.stamps[blockIdx.x * 16 + 15] = (unsigned)nt;
            if (!stamping) {
                unsigned long long clk1, rt1;
                asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1), "=s"(rt1) :: "memory");
                p.stamps[blockIdx.x * 16 + 12] = (unsigned)(clk1 - clk0);
                p.stamps[blockIdx.x * 16 + 13] = (unsigned)(rt1 - rt0);
            }
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();      // pairs with the extra barrier the wr == 1 half took at the start
    if constexpr (SIM == 2) {
        // both halves are level again: one more barrier and every wave's list appends are in the counters
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 256 && t_row_u + tid < p.M) {
            const int chunk = sim_chunk;
            const int n = *(volatile int*)(smem + PEPI + 4096 + tid * 4);
            p.simk_count[(size_t)(t_row_u + tid) * p.sim_nchunks + chunk] = min(n, p.simk_cap);
        }
    }
}

static int gemm256u_num_cu(int* out) {
    // one process drives one device at a time (torchrun design); re-read when the device changes
    static int cu_dev = -1, num_cu = 0;
    int dev = 0;
    KEMR_CHECK_HIP(hipGetDevice(&dev));
    if (cu_dev != dev) {
        KEMR_CHECK_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        cu_dev = dev;
    }
    *out = num_cu;
    return KEMR_OK;
}

// what the kernel's tile table and 32-bit tile offsets can hold (everything the encoders launch is far inside)
bool gemm256u_fits(const GemmParams& p, int elem_size, int c_elem_size) {
    int num_cu = 0;
    if (gemm256u_num_cu(&num_cu) != KEMR_OK || num_cu <= 0) return false;
    const long tiles = (long)((p.M + 255) / 256) * (p.N / 256);
    const long grid = tiles < num_cu ? tiles : num_cu;
    const long rows = (long)((p.M + 255) / 256) * 256;
    return (tiles + grid - 1) / grid <= GEMM256U_MAX_TILES_PER_WG && rows * p.lda * elem_size < (1L << 32) &&
           (long)p.N * p.ldw * elem_size < (1L << 32) && rows * p.ldc * c_elem_size < (1L << 32);
}

template <int EPI, bool FP8, bool DBG, bool CONC, bool KL = false>
static int launch256u_a(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm256u_bf16_nt_kernel<EPI, FP8, DBG, 0, CONC, KL>;
    static int attr_dev = -1;
    int dev = 0, num_cu = 0;
    KEMR_CHECK_HIP(hipGetDevice(&dev));
    KEMR_TRY(gemm256u_num_cu(&num_cu));
    if (attr_dev != dev) {
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PSMEM));
        attr_dev = dev;
    }
    if (!gemm256u_fits(p, FP8 ? 1 : 2, EPI == EPI_BIAS_RESID_F32 ? 4 : 2)) KEMR_FAIL(KEMR_ERR_INVALID, "gemm256u: M=%d N=%d K=%d is beyond the persistent kernel's tile table / 32-bit tile offsets", p.M, p.N, p.K);
    const int tiles = ((p.M + 255) / 256) * (p.N / 256);
    int grid = tiles < num_cu ? tiles : num_cu;
    if (g_gemm_grid > 0 && grid > g_gemm_grid && (tiles + g_gemm_grid - 1) / g_gemm_grid <= GEMM256U_MAX_TILES_PER_WG) grid = g_gemm_grid;
    GemmParams q = p;
    q.dbg = g_gemm_dbg;
    q.order = g_gemm_order;
    q.stamps = nullptr;
    if (DBG && (g_gemm_dbg & (64 | 128))) KEMR_CHECK_HIP(hipGetSymbolAddress((void**)&q.stamps, HIP_SYMBOL(g_gemm_stamp_buf)));
    ProfScope prof(PROF_GEMM, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), PSMEM, stream, q);
    KEMR_CHECK_LAUNCH("gemm256u_bf16_nt_kernel");
    return KEMR_OK;
}

template <int EPI, bool FP8>
static int launch256u(const GemmParams& p, hipStream_t stream) {
    // Epilogues of the two halves in one barrier interval: measured (round 2, same device, sustained) +1.8 % on fc1 + QuickGELU
    // (the VALU-heavy epilogue: 11.1 k -> 9.2 k cycles per tile for both halves), +-0 on the plain store epilogue (single-
    // buffered 3.3 k for both against 2.1 k + 2.9 k one after the other).  g_gemm_conc: 0 never, 1 always, 2 = where it pays.
    const bool conc = g_gemm_conc == 1 || (g_gemm_conc == 2 && EPI == EPI_BIAS_QGELU_BF16);
    if constexpr (FP8) {
        if (conc) return launch256u_a<EPI, true, false, true>(p, stream);
        return launch256u_a<EPI, true, false, false>(p, stream);
    } else {
#ifdef KEMR_AB_VARIANTS                            // tools/ only (build.py --ab-variants); kemr_debug_set refuses both switches otherwise
        if (g_gemm_dbg) {                         // the stamped / timing-experiment instantiation, for either K loop
            if (g_gemm_kl) return launch256u_a<EPI, false, true, false, true>(p, stream);
            if (conc) return launch256u_a<EPI, false, true, true>(p, stream);
            return launch256u_a<EPI, false, true, false>(p, stream);
        }
        if (g_gemm_kl) {                          // the long-interval K loop for A/B timing
            if (conc) return launch256u_a<EPI, false, false, true, true>(p, stream);
            return launch256u_a<EPI, false, false, false, true>(p, stream);
        }
#endif
        if (conc) return launch256u_a<EPI, false, false, true>(p, stream);
        return launch256u_a<EPI, false, false, false>(p, stream);
    }
}

// tools/ only: the stamp sums of the last DBG launch with flag 64 (per workgroup: 8 K-loop intervals, K-loop tail, epilogue,
// tiles, K-tiles per tile)
int gemm_read_stamps(unsigned* host_out, int n_words) {
    if (n_words < 0 || n_words > 1024 * 16) KEMR_FAIL(KEMR_ERR_INVALID, "gemm stamps: at most %d words", 1024 * 16);
    KEMR_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamp_buf), (size_t)n_words * 4, 0, hipMemcpyDeviceToHost));
    return KEMR_OK;
}

// chunks of the gallery per query tile: as few rounds of workgroups over the CUs as possible, counting half a tile of
// prologue per workgroup; a chunk is at most the tile table's size.  Returns 0 when the shape is outside the kernel.
static int sim_chunking(int nq, int ng, int kdim, long long gallery_offset, int* tpc_out) {
    int num_cu = 0;
    if (gemm256u_num_cu(&num_cu) != KEMR_OK) return 0;
    const int q_tiles = (nq + 255) / 256, g_tiles = (ng + 255) / 256;
    if (kdim % 64 != 0 || kdim < 128 || gallery_offset + ng > 0x7fffffffLL) return 0;
    if ((long)q_tiles * 256 * kdim * 2 >= (1L << 32) || (long)g_tiles * 256 * kdim * 2 >= (1L << 32)) return 0;
    int best_c = 0;
    double best = 1e30;
    for (int c = (g_tiles + GEMM256U_MAX_TILES_PER_WG - 1) / GEMM256U_MAX_TILES_PER_WG; c <= g_tiles && c <= 512; ++c) {
        const int tpc = (g_tiles + c - 1) / c, nch = (g_tiles + tpc - 1) / tpc;
        const long items = (long)q_tiles * nch;
        const double cost = (double)((items + num_cu - 1) / num_cu) * (tpc + 0.5);
        if (cost < best - 1e-9) { best = cost; best_c = nch; }
    }
    if (best_c <= 0) return 0;
    *tpc_out = (g_tiles + best_c - 1) / best_c;
    return (g_tiles + *tpc_out - 1) / *tpc_out;
}

template <int SIM, bool DBG, bool KL = false>
static int launch_sim_mode_a(const GemmParams& p, int q_tiles, hipStream_t stream) {
    auto kern = gemm256u_bf16_nt_kernel<EPI_BIAS_BF16, false, DBG, SIM, false, KL>;
    static int attr_dev = -1;
    int dev = 0;
    KEMR_CHECK_HIP(hipGetDevice(&dev));
    if (attr_dev != dev) {
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PSMEM));
        attr_dev = dev;
    }
    GemmParams q = p;
    if (DBG) {
        q.dbg = g_gemm_dbg;
        KEMR_CHECK_HIP(hipGetSymbolAddress((void**)&q.stamps, HIP_SYMBOL(g_gemm_stamp_buf)));
    }
    ProfScope prof(PROF_SIM, stream);
    q.sim_xcd = (q_tiles % 8 == 0 || q_tiles >= 64) ? 1 : 0;
    const int grid_q = q.sim_xcd ? ((q_tiles + 7) / 8) * 8 : q_tiles;      // XCD dealing: whole groups of 8 query tiles (one per XCD)
    hipLaunchKernelGGL(kern, dim3(grid_q * p.sim_nchunks), dim3(512), PSMEM, stream, q);
    KEMR_CHECK_LAUNCH("gemm256u_bf16_nt_kernel<sim>");
    return KEMR_OK;
}

template <int SIM>
static int launch_sim_mode(const GemmParams& p, int q_tiles, hipStream_t stream) {
#ifdef KEMR_AB_VARIANTS
    if constexpr (SIM != 3) {
        if (g_gemm_dbg & (64 | 128)) return launch_sim_mode_a<SIM, true>(p, q_tiles, stream);      // tools: stamped instantiation
    }
    if constexpr (SIM == 2) {
        if (g_gemm_kl) return launch_sim_mode_a<SIM, false, true>(p, q_tiles, stream);      // tools/: long-interval K loop, A/B timing
    }
#endif
    return launch_sim_mode_a<SIM, false>(p, q_tiles, stream);
}

// Rank-only similarity pass (kemr_sim_topk with k == 0 and no bonus list).  Panels must be allocated with their row count
// rounded up to 256 (include/kemr.h).  *used = false: the shape is outside this kernel (the caller falls back to sim_kernel).
int launch_gemm256u_simrank(const bf16_t* q_panel, int nq, const bf16_t* g_panel, int ng, int kdim, long long gallery_offset,
                            const int32_t* gt_idx, const float* gt_score, int32_t* ahead, hipStream_t stream, bool* used) {
    *used = false;
    int tpc = 0;
    const int nch = sim_chunking(nq, ng, kdim, gallery_offset, &tpc);
    if (nch <= 0) return KEMR_OK;
    const int q_tiles = (nq + 255) / 256, g_tiles = (ng + 255) / 256;
    GemmParams p{};
    p.A = q_panel; p.lda = kdim; p.W = g_panel; p.ldw = kdim; p.M = nq; p.N = g_tiles * 256; p.K = kdim;
    p.sim_gt = gt_idx; p.sim_sgt = gt_score; p.sim_ahead = ahead; p.sim_ng = ng; p.sim_gbase = (int)gallery_offset;
    p.sim_tpc = tpc; p.sim_nchunks = nch;
    KEMR_TRY(launch_sim_mode<1>(p, q_tiles, stream));
    *used = true;
    return KEMR_OK;
}

// Top-k candidate pass (SIM == 2).  hits_per_query: the expected number of (lane, query) records per query over the whole
// gallery, spread: how far above its mean a list may run (the caller derives both from how its thresholds were chosen: the
// count is Poisson around a Gamma(k)-distributed mean, whose upper tail is long -- over 10^5 lists a capacity of 3x the mean
// overflowed in every second call at k = 10, round 2).  A list holds spread x the mean per chunk + 16, rounded up to a power
// of two.  *ok = false: the shape is outside the kernel.
int gemm256u_simk_plan(int nq, int ng, int kdim, double hits_per_query, double spread, SimkPlan* plan, bool* ok) {
    *ok = false;
    int tpc = 0;
    const int nch = sim_chunking(nq, ng, kdim, 0, &tpc);
    if (nch <= 0) return KEMR_OK;
    int cap = 16;
    const double want = (spread < 3.0 ? 3.0 : spread) * hits_per_query / nch + 16.0;
    while (cap < want && cap < 4096) cap *= 2;
    plan->nchunks = nch; plan->tpc = tpc; plan->cap = cap;
    const size_t lists = (size_t)nq * nch;
    plan->scores_bytes = lists * cap * 64;
    plan->base_bytes = lists * cap * 4;
    plan->count_bytes = (lists * 4 + 255) / 256 * 256;
    *ok = true;
    return KEMR_OK;
}

int launch_gemm256u_simk(const bf16_t* q_panel, int nq, const bf16_t* g_panel, int ng, int kdim, long long gallery_offset,
                         const int32_t* gt_idx, const float* gt_score, int32_t* ahead, const float* taud, const SimkPlan& plan,
                         float* rec_scores, int32_t* rec_base, int32_t* rec_count, int32_t* flag, hipStream_t stream) {
    if (gallery_offset + ng > 0x7fffffffLL) KEMR_FAIL(KEMR_ERR_INVALID, "simk: candidate ids exceed int32");
    const int q_tiles = (nq + 255) / 256, g_tiles = (ng + 255) / 256;
    GemmParams p{};
    p.A = q_panel; p.lda = kdim; p.W = g_panel; p.ldw = kdim; p.M = nq; p.N = g_tiles * 256; p.K = kdim;
    p.sim_gt = gt_idx; p.sim_sgt = gt_score; p.sim_ahead = ahead; p.sim_ng = ng; p.sim_gbase = (int)gallery_offset;
    p.sim_tpc = plan.tpc; p.sim_nchunks = plan.nchunks;
    p.simk_taud = taud; p.simk_scores = rec_scores; p.simk_base = rec_base; p.simk_count = rec_count; p.simk_flag = flag;
    p.simk_cap = plan.cap;
    return launch_sim_mode<2>(p, q_tiles, stream);
}

// Group maxima (SIM == 3): out[nq][ng / 64] = the best score of every query within each block of 64 sampled gallery rows.
// The sample is rows 0, stride, 2 stride, ... of g_panel, read in place through the W operand's leading dimension (ng rows, a
// multiple of 256, all REAL: ng * stride <= the panel's valid rows) -- no copy of the sample, one launch fewer (round 3).
int launch_gemm256u_simgmax(const bf16_t* q_panel, int nq, const bf16_t* g_panel, int ng, int kdim, int stride, float* out,
                            hipStream_t stream, bool* used) {
    *used = false;
    if (ng % 256 != 0 || stride < 1) KEMR_FAIL(KEMR_ERR_INVALID, "simgmax: %d sampled rows are not whole tiles (stride %d)", ng, stride);
    if ((long)ng * stride * kdim * 2 >= (1L << 32)) return KEMR_OK;          // 32-bit W offsets
    int tpc = 0;
    const int nch = sim_chunking(nq, ng, kdim, 0, &tpc);
    if (nch <= 0) return KEMR_OK;
    GemmParams p{};
    p.A = q_panel; p.lda = kdim; p.W = g_panel; p.ldw = kdim * stride; p.M = nq; p.N = ng; p.K = kdim;
    p.sim_ng = ng; p.sim_tpc = tpc; p.sim_nchunks = nch;
    p.simk_scores = out;
    KEMR_TRY(launch_sim_mode<3>(p, (nq + 255) / 256, stream));
    *used = true;
    return KEMR_OK;
}

// C must have ceil256(M) rows: rows in [M, ceil256(M)) are written (with values computed from A's pad rows).
int launch_gemm256u(const GemmParams& p, int epi, hipStream_t stream) {
    switch (epi) {
        case EPI_BIAS_BF16:       return launch256u<EPI_BIAS_BF16, false>(p, stream);
        case EPI_BIAS_QGELU_BF16: return launch256u<EPI_BIAS_QGELU_BF16, false>(p, stream);
        case EPI_BIAS_RESADD_BF16: return launch256u_a<EPI_BIAS_RESADD_BF16, false, false, false>(p, stream);
        case EPI_BIAS_RESID_F32:
#ifdef KEMR_AB_VARIANTS
            if (g_gemm_kl) return launch256u_a<EPI_BIAS_RESID_F32, false, false, false, true>(p, stream);
#endif
            return launch256u_a<EPI_BIAS_RESID_F32, false, false, false>(p, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "gemm256u: epilogue %d is not one of this kernel's", epi);
}

// fp8 e4m3 operands: A [ceil256(M), lda] and W [N, ldw] in bytes, K % 128 == 0, K >= 256, N % 256 == 0; p.wscale[N] scales the
// accumulators per output channel before the bias; C is bf16 as above.
int launch_gemm256u_fp8(const GemmParams& p, int epi, hipStream_t stream) {
    if (p.M <= 0) return KEMR_OK;
    if (p.N % 256 != 0 || p.K % 128 != 0 || p.K < 256) KEMR_FAIL(KEMR_ERR_INVALID, "gemm fp8: need N %% 256 == 0, K %% 128 == 0, K >= 256 (got N=%d K=%d)", p.N, p.K);
    if ((p.lda % 16) || (p.ldw % 16) || (p.ldc % 4)) KEMR_FAIL(KEMR_ERR_INVALID, "gemm fp8: leading dimensions must keep 16-byte alignment");
    if (!p.wscale || !p.c_rows_padded) KEMR_FAIL(KEMR_ERR_INVALID, "gemm fp8: needs weight scales and a row-padded C");
    switch (epi) {
        case EPI_BIAS_BF16:       return launch256u<EPI_BIAS_BF16, true>(p, stream);
        case EPI_BIAS_QGELU_BF16: return launch256u<EPI_BIAS_QGELU_BF16, true>(p, stream);
    }
    KEMR_FAIL(KEMR_ERR_INVALID, "gemm fp8: epilogue %d is not a bf16-store epilogue", epi);
}

}  // namespace kemr
