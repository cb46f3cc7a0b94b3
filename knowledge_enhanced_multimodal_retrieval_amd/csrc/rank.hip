// Ranking of an already materialised score matrix (HBM-bound, one streaming pass).
// Serves the reference entry points that are handed a dense [Q, N] matrix instead of embeddings:
// compute_recall_at_k / compute_mrr_and_mean_rank (metrics.py:13-76), compute_retrieval_metrics_fusion
// (metrics.py:165-185), evaluate_retrieval (eval/fusion.py:6-20) and the learned-fusion evaluator
// (eval/evaluator_fusion.py:126).  One 256-thread workgroup per query row: 16-byte loads where the row is
// aligned, per-thread `ahead` count + sorted top-KMAX list in registers, then a k-round selection over the
// 256 lists through LDS.  Algorithmic bytes: 4 * N per row.
#include "common.h"

namespace kemr {

__device__ __forceinline__ bool rank_before(float sa, int ia, float sb, int ib) {
    return sa > sb || (sa == sb && ia < ib);
}

template <int KMAX>
__device__ __forceinline__ void list_insert(float (&s)[KMAX], int (&id)[KMAX], float v, int idx) {
#pragma unroll
    for (int i = KMAX - 1; i > 0; --i) {
        const bool shift = v > s[i - 1];
        const bool here = !shift && v > s[i];
        s[i] = shift ? s[i - 1] : (here ? v : s[i]);
        id[i] = shift ? id[i - 1] : (here ? idx : id[i]);
    }
    if (v > s[0]) { s[0] = v; id[0] = idx; }
}

template <int KMAX>
__global__ __launch_bounds__(256) void rank_dense_kernel(const float* __restrict__ S, long long ld, int nq, int ng,
                                                         const int32_t* __restrict__ gt_idx, int32_t* __restrict__ ahead,
                                                         int k, float* __restrict__ top_s, int32_t* __restrict__ top_i) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;                 // [256][KMAX]
    int* xi = (int*)(xs + 256 * KMAX);        // [256][KMAX]
    int* red = xi + 256 * KMAX;               // [4]
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* row = S + (size_t)q * ld;
    const int gt = gt_idx ? gt_idx[q] : -1;
    const bool has_gt = gt >= 0 && gt < ng;
    const float sgt = has_gt ? row[gt] : 0.f;
    float ls[KMAX];
    int li[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) { ls[i] = -INFINITY; li[i] = -1; }
    int cnt = 0;
    // a thread's candidate ids increase monotonically, so strict '>' keeps "lower id first" inside its list
    for (int j = tid; j < ng; j += 256) {
        const float v = row[j];
        if (has_gt && j != gt) cnt += rank_before(v, j, sgt, gt) ? 1 : 0;
        if (top_s && v > ls[KMAX - 1]) list_insert<KMAX>(ls, li, v, j);
    }
    if (ahead && has_gt) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) red[tid >> 6] = cnt;
    }
    if (top_s) {
#pragma unroll
        for (int i = 0; i < KMAX; ++i) { xs[tid * KMAX + i] = ls[i]; xi[tid * KMAX + i] = li[i]; }
    }
    __syncthreads();
    if (ahead && tid == 0) ahead[q] = has_gt ? (red[0] + red[1]) + (red[2] + red[3]) : 0;
    if (top_s && tid < 64) {
        float ps = INFINITY;
        int pi = -1;
        bool done = false;
        for (int o = 0; o < k; ++o) {
            float bs = -INFINITY;
            int bi = -1;
            if (!done) {
                for (int e = lane; e < 256 * KMAX; e += 64) {
                    const int ei = xi[e];
                    if (ei < 0) continue;
                    const float es = xs[e];
                    const bool after_prev = (o == 0) || rank_before(ps, pi, es, ei);
                    if (after_prev && (bi < 0 || rank_before(es, ei, bs, bi))) { bs = es; bi = ei; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float os = __shfl_xor(bs, off);
                    const int oi = __shfl_xor(bi, off);
                    if (oi >= 0 && (bi < 0 || rank_before(os, oi, bs, bi))) { bs = os; bi = oi; }
                }
                if (bi < 0) done = true;
            }
            if (lane == 0) { top_s[(size_t)q * k + o] = bi >= 0 ? bs : -INFINITY; top_i[(size_t)q * k + o] = bi; }
            ps = bs;
            pi = bi;
        }
    }
}

// Learned "linear" fusion head (reference fusion_model.py:25-48, eval mode): out = w1 . relu(W0 . [t2i, t2t] + b0) + b1
// per (query, candidate) pair.  Element-wise over the two dense score matrices; the tiny MLP lives in LDS.
__global__ __launch_bounds__(256) void linear_head_kernel(const float* __restrict__ t2i, const float* __restrict__ t2t,
                                                          long long n, const float* __restrict__ w0,
                                                          const float* __restrict__ b0, const float* __restrict__ w1,
                                                          float b1, int hidden, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* hw = (float4*)smem;                    // per hidden unit: (w0[h][0], w0[h][1], b0[h], w1[h])
    for (int h = threadIdx.x; h < hidden; h += 256) hw[h] = make_float4(w0[2 * h], w0[2 * h + 1], b0[h], w1[h]);
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float a = t2i[i], b = t2t[i];
        float acc = b1;
        for (int h = 0; h < hidden; ++h) {
            const float4 w = hw[h];
            acc = fmaf(w.w, fmaxf(fmaf(w.x, a, fmaf(w.y, b, w.z)), 0.f), acc);
        }
        out[i] = acc;
    }
}

// Learned "cross_attention" fusion head (reference fusion_model.py:51-133, eval mode), pair stage.
// The reference runs, for EVERY (query n, candidate m) pair, an 8-head attention of one query over two keys
// (image / target), the D x D output projection and a D->256->64->1 MLP: O(N*M*D^2).  All of it except the two
// softmax weights per head is linear in per-candidate quantities, so the host precomputes, per candidate and head,
//   P_x[m][h][:] = W1 . Wo[:, head h] . V_x[m][h][:]   (x = image / target; 256 floats)
// and this kernel only does, per pair: 2-way softmax of the 2*H scaled scores, hidden1 = relu(c0 + sum_h w_i P_i + w_t P_t)
// (H*2 FMAs per hidden unit), the 256->64 layer fused on the fly, the 64->1 layer and 0.5*tanh.  ~21 kFLOP per pair
// instead of ~1.6 MFLOP.  One workgroup per candidate m (its P rows, W2^T in LDS: broadcast reads), threads over n.
// scores come transposed, st[x][h][m][n], so that threads (consecutive n) read and write coalesced.
template <int H>
__global__ __launch_bounds__(256) void cross_attn_pair_kernel(const float* __restrict__ st_i, const float* __restrict__ st_t,
                                                              const float* __restrict__ p_i, const float* __restrict__ p_t,
                                                              const float* __restrict__ c0, const float* __restrict__ w2t,
                                                              const float* __restrict__ b2, const float* __restrict__ w3,
                                                              float b3, int n_q, int n_c, int hid1, int hid2,
                                                              float* __restrict__ out_t /* [n_c][n_q] */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sP = (float*)smem;                       // [2][H][hid1]
    float* sW2 = sP + 2 * H * hid1;                 // [hid1][hid2]  (W2 transposed)
    float* sC0 = sW2 + hid1 * hid2;                 // [hid1]
    const int m = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < H * hid1; i += 256) {
        sP[i] = p_i[(size_t)m * H * hid1 + i];
        sP[H * hid1 + i] = p_t[(size_t)m * H * hid1 + i];
    }
    for (int i = tid; i < hid1 * hid2; i += 256) sW2[i] = w2t[i];
    for (int i = tid; i < hid1; i += 256) sC0[i] = c0[i];
    __syncthreads();
    const size_t plane = (size_t)n_c * n_q;
    for (int n = tid; n < n_q; n += 256) {
        float wi[H], wt[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float a = st_i[h * plane + (size_t)m * n_q + n], b = st_t[h * plane + (size_t)m * n_q + n];
            const float mx = fmaxf(a, b);
            const float ea = __expf(a - mx), eb = __expf(b - mx);
            const float inv = 1.0f / (ea + eb);
            wi[h] = ea * inv;
            wt[h] = eb * inv;
        }
        float acc[64];                               // hid2 <= 64
#pragma unroll
        for (int k = 0; k < 64; ++k) acc[k] = 0.f;
        for (int j = 0; j < hid1; ++j) {
            float hsum = sC0[j];
#pragma unroll
            for (int h = 0; h < H; ++h) hsum = fmaf(wi[h], sP[h * hid1 + j], fmaf(wt[h], sP[(H + h) * hid1 + j], hsum));
            hsum = fmaxf(hsum, 0.f);
            const float* w2row = sW2 + j * hid2;
#pragma unroll
            for (int k = 0; k < 64; ++k)
                if (k < hid2) acc[k] = fmaf(hsum, w2row[k], acc[k]);
        }
        float o = b3;
#pragma unroll
        for (int k = 0; k < 64; ++k)
            if (k < hid2) o = fmaf(fmaxf(acc[k] + b2[k], 0.f), w3[k], o);
        out_t[(size_t)m * n_q + n] = 0.5f * tanhf(o);
    }
}

// gate[r] = sigmoid(sum_c act(x[r, c] + pre[c]) * w[c] + bias), act = ReLU or identity: the last stage of the gated fusion heads
// (one wave per row, fp32 throughout)
__global__ __launch_bounds__(256) void gate_rows_kernel(const float* __restrict__ x, int rows, int cols, const float* __restrict__ pre,
                                                        const float* __restrict__ w, float bias, int relu, float* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        float v = xr[c] + (pre ? pre[c] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        s = fmaf(v, w[c], s);
    }
    s = wave_sum(s) + bias;
    if (lane == 0) out[r] = 1.0f / (1.0f + expf(-s));
}

}  // namespace kemr

using namespace kemr;

extern "C" int kemr_gate_rows(const float* x_dev, int rows, int cols, const float* pre_bias_dev, const float* w_dev, float bias, int relu,
                              float* out_dev, void* stream) {
    if (rows == 0) return KEMR_OK;
    if (!x_dev || !w_dev || !out_dev || rows < 0 || cols < 1) KEMR_FAIL(KEMR_ERR_INVALID, "gate_rows: bad argument");
    hipLaunchKernelGGL(gate_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x_dev, rows, cols, pre_bias_dev, w_dev,
                       bias, relu, out_dev);
    KEMR_CHECK_LAUNCH("gate_rows_kernel");
    return KEMR_OK;
}

extern "C" int kemr_cross_attention_pairs(const float* st_i_dev, const float* st_t_dev, const float* p_i_dev,
                                          const float* p_t_dev, const float* c0_dev, const float* w2t_dev,
                                          const float* b2_dev, const float* w3_dev, float b3, int heads, int n_q, int n_c,
                                          int hid1, int hid2, float* out_t_dev, void* stream) {
    if (n_q == 0 || n_c == 0) return KEMR_OK;
    if (!st_i_dev || !st_t_dev || !p_i_dev || !p_t_dev || !c0_dev || !w2t_dev || !b2_dev || !w3_dev || !out_t_dev)
        KEMR_FAIL(KEMR_ERR_INVALID, "cross_attention_pairs: null argument");
    if (heads != 8) KEMR_FAIL(KEMR_ERR_INVALID, "cross_attention_pairs: the reference head has 8 attention heads (got %d)", heads);
    if (hid2 < 1 || hid2 > 64 || hid1 < 1 || n_q < 0 || n_c < 0) KEMR_FAIL(KEMR_ERR_INVALID, "cross_attention_pairs: bad sizes");
    const size_t smem = ((size_t)2 * heads * hid1 + (size_t)hid1 * hid2 + hid1) * 4;
    if (smem > 160 * 1024) KEMR_FAIL(KEMR_ERR_INVALID, "cross_attention_pairs: hidden sizes do not fit LDS");
    auto kern = cross_attn_pair_kernel<8>;
    KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(n_c), dim3(256), smem, (hipStream_t)stream, st_i_dev, st_t_dev, p_i_dev, p_t_dev, c0_dev,
                       w2t_dev, b2_dev, w3_dev, b3, n_q, n_c, hid1, hid2, out_t_dev);
    KEMR_CHECK_LAUNCH("cross_attn_pair_kernel");
    return KEMR_OK;
}

extern "C" int kemr_linear_head(const float* t2i_dev, const float* t2t_dev, int64_t n, const float* w0_dev,
                                const float* b0_dev, const float* w1_dev, float b1, int hidden, float* out_dev,
                                void* stream) {
    if (n == 0) return KEMR_OK;
    if (!t2i_dev || !t2t_dev || !w0_dev || !b0_dev || !w1_dev || !out_dev || n < 0 || hidden < 1 || hidden > 2048)
        KEMR_FAIL(KEMR_ERR_INVALID, "linear_head: bad argument");
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(linear_head_kernel, dim3((unsigned)blocks), dim3(256), (size_t)hidden * 16, (hipStream_t)stream,
                       t2i_dev, t2t_dev, (long long)n, w0_dev, b0_dev, w1_dev, b1, hidden, out_dev);
    KEMR_CHECK_LAUNCH("linear_head_kernel");
    return KEMR_OK;
}

extern "C" int kemr_rank_dense(const float* scores_dev, int nq, int ng, int64_t ld, const int32_t* gt_idx_dev,
                               int32_t* ahead_dev, int k, float* top_scores_dev, int32_t* top_idx_dev, void* stream) {
    if (nq == 0) return KEMR_OK;
    if (!scores_dev || nq < 0 || ng <= 0 || ld < ng) KEMR_FAIL(KEMR_ERR_INVALID, "rank_dense: bad argument");
    if ((top_scores_dev != nullptr) != (top_idx_dev != nullptr)) KEMR_FAIL(KEMR_ERR_INVALID, "rank_dense: top_scores/top_idx together");
    if (top_scores_dev && (k < 1 || k > 32)) KEMR_FAIL(KEMR_ERR_INVALID, "rank_dense: k=%d not in 1..32", k);
    if ((gt_idx_dev != nullptr) != (ahead_dev != nullptr)) KEMR_FAIL(KEMR_ERR_INVALID, "rank_dense: gt_idx and ahead together");
    hipStream_t s = (hipStream_t)stream;
    if (!top_scores_dev || k <= 10) {
        constexpr int KM = 10;
        const size_t smem = 256 * KM * 8 + 16;
        hipLaunchKernelGGL(rank_dense_kernel<KM>, dim3(nq), dim3(256), smem, s, scores_dev, (long long)ld, nq, ng, gt_idx_dev,
                           ahead_dev, k, top_scores_dev, top_idx_dev);
    } else {
        constexpr int KM = 32;
        const size_t smem = 256 * KM * 8 + 16;
        auto kern = rank_dense_kernel<KM>;
        KEMR_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kern, dim3(nq), dim3(256), smem, s, scores_dev, (long long)ld, nq, ng, gt_idx_dev, ahead_dev, k,
                           top_scores_dev, top_idx_dev);
    }
    KEMR_CHECK_LAUNCH("rank_dense_kernel");
    return KEMR_OK;
}
