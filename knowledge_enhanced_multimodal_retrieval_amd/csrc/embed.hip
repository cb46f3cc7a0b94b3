// Token assembly and pooling tail of the CLIP towers (HBM-bound element-wise work, fused where it is free).
//   im2col      : NCHW fp32 pixels -> bf16 patch rows [B*P, kpad] so that the patch-embedding conv is a GEMM
//                 (k = c*p*p + dy*p + dx, zero-padded to a multiple of 64); the GEMM epilogue adds the positional
//                 embedding and scatters to token rows (gemm.hip, EPI_PATCH_F32).
//   cls_rows    : token 0 of every image = class_embedding + positional_embedding[0].
//   text_embed  : token_embedding[ids] + positional_embedding.
//   tail        : pooled row (CLS, or the first arg-max of the ids = EOT) -> LayerNorm -> @ proj (fp32 weights)
//                 -> optional L2 normalisation, one workgroup per output row.
#include "common.h"

namespace kemr {

__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ px, bf16_t* __restrict__ out,
                                                     int image_size, int patch, int grid, int kpad, int kvalid,
                                                     long long total_pairs) {
    // one thread per pair of adjacent k (patch is even for every CLIP model; odd patch handled element-wise)
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_pairs) return;
    const int kp2 = kpad >> 1;
    const long long row = gid / kp2;
    const int k = (int)(gid - row * kp2) * 2;
    const int P = grid * grid;
    const int b = (int)(row / P), pi = (int)(row - (long long)b * P);
    const int py = pi / grid, pxi = pi - py * grid;
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int kk = k + e;
        if (kk < kvalid) {
            const int c = kk / (patch * patch), rem = kk - c * patch * patch;
            const int dy = rem / patch, dx = rem - dy * patch;
            v[e] = px[(((size_t)b * 3 + c) * image_size + (py * patch + dy)) * image_size + pxi * patch + dx];
        } else {
            v[e] = 0.f;
        }
    }
    *(uint32_t*)(out + (size_t)row * kpad + k) = pack_bf16x2(v[0], v[1]);
}

int launch_im2col(const float* pixels, bf16_t* patches, int batch, int image_size, int patch, int kpad, hipStream_t stream) {
    const int grid = image_size / patch;
    const long long pairs = (long long)batch * grid * grid * (kpad / 2);
    if (pairs <= 0) return KEMR_OK;
    const long long blocks = (pairs + 255) / 256;
    if (blocks > 0x7fffffffLL) KEMR_FAIL(KEMR_ERR_INVALID, "im2col: batch too large");
    ProfScope prof(PROF_OTHER, stream);
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pixels, patches, image_size, patch,
                       grid, kpad, 3 * patch * patch, pairs);
    KEMR_CHECK_LAUNCH("im2col_kernel");
    return KEMR_OK;
}

__global__ __launch_bounds__(256) void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, int tokens, int width) {
    const int b = blockIdx.x;
    float* dst = x + (size_t)b * tokens * width;
    for (int i = threadIdx.x; i < width; i += 256) dst[i] = cls[i] + pos[i];
}

int launch_cls_rows(float* x, const float* class_emb, const float* pos, int batch, int tokens, int width, hipStream_t stream) {
    if (batch <= 0) return KEMR_OK;
    ProfScope prof(PROF_OTHER, stream);
    hipLaunchKernelGGL(cls_rows_kernel, dim3(batch), dim3(256), 0, stream, x, class_emb, pos, tokens, width);
    KEMR_CHECK_LAUNCH("cls_rows_kernel");
    return KEMR_OK;
}

// row_start[i] = lens[0] + .. + lens[i - 1] for the packed text rows (one workgroup; batch <= 65535).  Whatever the caller passed,
// the result is safe to index with: every length is clamped into 1 .. max_len and the prefix sums into `rows` such that every text
// keeps at least one row (row_start[i] <= rows - (batch - i)); with honest arguments (rows = the sum) the clamps change nothing.
__global__ __launch_bounds__(1024) void row_starts_kernel(const int32_t* __restrict__ lens, int batch, int max_len, int rows,
                                                          int* __restrict__ row_start) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (batch + 1023) / 1024;
    const int i0 = tid * per, i1 = i0 + per < batch ? i0 + per : batch;
    int sum = 0;
    for (int i = i0; i < i1; ++i) { const int l = lens[i]; sum += l < 1 ? 1 : (l > max_len ? max_len : l); }
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                // inclusive scan of the per-thread sums
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;                          // exclusive
    if (tid == 0) row_start[0] = 0;
    for (int i = i0; i < i1; ++i) {
        const int l = lens[i];
        run += l < 1 ? 1 : (l > max_len ? max_len : l);
        const int cap = rows - (batch - (i + 1));
        row_start[i + 1] = run < cap ? run : cap;
    }
}

int launch_row_starts(const int32_t* lens, int batch, int max_len, int rows, int* row_start, hipStream_t stream) {
    if (batch <= 0) return KEMR_OK;
    if (batch > 65535) KEMR_FAIL(KEMR_ERR_INVALID, "row_starts: batch %d > 65535", batch);
    if (rows < batch || (int64_t)rows > (int64_t)batch * max_len) KEMR_FAIL(KEMR_ERR_INVALID, "row_starts: %d rows for %d texts of 1 .. %d positions", rows, batch, max_len);
    ProfScope prof(PROF_OTHER, stream);
    hipLaunchKernelGGL(row_starts_kernel, dim3(1), dim3(1024), 0, stream, lens, batch, max_len, rows, row_start);
    KEMR_CHECK_LAUNCH("row_starts_kernel");
    return KEMR_OK;
}

template <typename XT>
__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ ids, const float* __restrict__ tok,
                                                         const float* __restrict__ pos, XT* __restrict__ x,
                                                         int ctx, int width, int vocab, long long total4,
                                                         const int* __restrict__ row_start, int batch) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total4) return;
    const int w4 = width >> 2;
    const long long row = gid / w4;
    const int c4 = (int)(gid - row * w4);
    int t = (int)(row % ctx);
    long long src = row;
    if (row_start) {                  // packed rows: text i owns the rows row_start[i] .. row_start[i + 1] - 1 = its first positions
        int lo = 0, hi = batch - 1;   // the last i with row_start[i] <= row
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (row_start[mid] <= row) lo = mid; else hi = mid - 1;
        }
        t = (int)(row - row_start[lo]);
        t = t < ctx ? t : ctx - 1;    // (rows behind the last text, if the caller's row count exceeds the sum of the lengths)
        src = (long long)lo * ctx + t;
    }
    int id = ids[src];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // out-of-range ids are clamped (torch would raise)
    const float4 a = ((const float4*)(tok + (size_t)id * width))[c4];
    const float4 p = ((const float4*)(pos + (size_t)t * width))[c4];
    if constexpr (sizeof(XT) == 4) {
        ((float4*)(x + (size_t)row * width))[c4] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    } else if constexpr (sizeof(XT) == 3) {            // 24-bit rows: W upper halves, then W third bytes (common.h f24_t)
        const uint32_t e0 = f32_to_f24_bits(a.x + p.x), e1 = f32_to_f24_bits(a.y + p.y), e2 = f32_to_f24_bits(a.z + p.z), e3 = f32_to_f24_bits(a.w + p.w);
        uint8_t* r = (uint8_t*)(x + (size_t)row * width);
        uint2 h;
        h.x = (e0 >> 16) | (e1 & 0xffff0000u);
        h.y = (e2 >> 16) | (e3 & 0xffff0000u);
        ((uint2*)r)[c4] = h;
        ((uint32_t*)(r + 2 * (size_t)width))[c4] = ((e0 >> 8) & 0xff) | (e1 & 0xff00) | ((e2 << 8) & 0xff0000) | ((e3 << 16) & 0xff000000u);
    } else {
        uint2 pk;
        pk.x = pack_bf16x2(a.x + p.x, a.y + p.y);
        pk.y = pack_bf16x2(a.z + p.z, a.w + p.w);
        ((uint2*)(x + (size_t)row * width))[c4] = pk;
    }
}

int launch_text_embed(const int32_t* ids, const float* tok_emb, const float* pos, void* x, int x_dtype, int batch, int ctx,
                      int width, int vocab, hipStream_t stream, const int* row_start, int rows) {
    const long long total4 = (row_start ? (long long)rows : (long long)batch * ctx) * (width / 4);
    if (total4 <= 0) return KEMR_OK;
    ProfScope prof(PROF_OTHER, stream);
    const dim3 grid((unsigned)((total4 + 255) / 256));
    if (x_dtype == KEMR_BF16)
        hipLaunchKernelGGL(text_embed_kernel<bf16_t>, grid, dim3(256), 0, stream, ids, tok_emb, pos, (bf16_t*)x, ctx, width, vocab, total4, row_start, batch);
    else if (x_dtype == KEMR_F24)
        hipLaunchKernelGGL(text_embed_kernel<f24_t>, grid, dim3(256), 0, stream, ids, tok_emb, pos, (f24_t*)x, ctx, width, vocab, total4, row_start, batch);
    else
        hipLaunchKernelGGL(text_embed_kernel<float>, grid, dim3(256), 0, stream, ids, tok_emb, pos, (float*)x, ctx, width, vocab, total4, row_start, batch);
    KEMR_CHECK_LAUNCH("text_embed_kernel");
    return KEMR_OK;
}

// pool_idx[b] = the row of item b that leaves the tower: b * tokens (class token; ids == nullptr) or the first position of the
// row maximum of its token ids (end-of-text token, torch.argmax semantics) -- inside the rows row_start gives the text when the
// rows are packed (clamped to its last row, see kemr_encode_text_packed).  One wave per item.
__global__ __launch_bounds__(256) void pool_index_kernel(const int32_t* __restrict__ ids, const int* __restrict__ row_start, int batch,
                                                         int tokens, int* __restrict__ pool_idx) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= batch) return;
    int best_t = 0;
    if (ids) {
        int best_v = INT_MIN;
        best_t = INT_MAX;
        for (int t = lane; t < tokens; t += 64) {
            const int v = ids[(size_t)b * tokens + t];
            if (v > best_v) { best_v = v; best_t = t; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(best_v, o), ot = __shfl_xor(best_t, o);
            if (ov > best_v || (ov == best_v && ot < best_t)) { best_v = ov; best_t = ot; }
        }
    }
    if (lane == 0) {
        int row = b * tokens + best_t;
        if (row_start) {
            const int r0 = row_start[b], len = row_start[b + 1] - r0;
            row = r0 + (best_t < len ? best_t : len - 1);
        }
        pool_idx[b] = row;
    }
}

int launch_pool_index(const int32_t* ids, const int* row_start, int batch, int tokens, int* pool_idx, hipStream_t stream) {
    if (batch <= 0) return KEMR_OK;
    ProfScope prof(PROF_OTHER, stream);
    hipLaunchKernelGGL(pool_index_kernel, dim3((batch + 3) / 4), dim3(256), 0, stream, ids, row_start, batch, tokens, pool_idx);
    KEMR_CHECK_LAUNCH("pool_index_kernel");
    return KEMR_OK;
}

// The pooled rows of the residual stream x (fp32 or bf16) and of the LayerNorm output h (bf16), copied out into compact [batch, width]
// buffers: what the last block's query path works on.  One workgroup per item.
// (rows copied as bytes: x_bytes = 4, 3 (24-bit rows, W upper halves + W third bytes) or 2 per element; h_bytes = 2 or 1 (e4m3))
__global__ __launch_bounds__(256) void gather_pooled_kernel(const uint8_t* __restrict__ x, int x_bytes, const uint8_t* __restrict__ h, int h_bytes,
                                                            const int* __restrict__ pool_idx, int width, uint8_t* __restrict__ xc,
                                                            uint8_t* __restrict__ hc) {
    const int b = blockIdx.x;
    const size_t row = (size_t)pool_idx[b];
    const int xb = width * x_bytes, hb = width * h_bytes;              // multiples of 4: width % 4 == 0
    for (int i = threadIdx.x * 4; i < xb; i += 1024) *(uint32_t*)(xc + (size_t)b * xb + i) = *(const uint32_t*)(x + row * xb + i);
    for (int i = threadIdx.x * 4; i < hb; i += 1024) *(uint32_t*)(hc + (size_t)b * hb + i) = *(const uint32_t*)(h + row * hb + i);
}

int launch_gather_pooled(const void* x, int x_dtype, const void* h, int h_dtype, const int* pool_idx, int batch, int width, void* xc, void* hc,
                         hipStream_t stream) {
    if (batch <= 0) return KEMR_OK;
    if (width % 4) KEMR_FAIL(KEMR_ERR_INVALID, "gather: width %d must be a multiple of 4", width);
    ProfScope prof(PROF_OTHER, stream);
    const int xb = x_dtype == KEMR_BF16 ? 2 : (x_dtype == KEMR_F24 ? 3 : 4), hb = h_dtype == KEMR_FP8 ? 1 : 2;
    hipLaunchKernelGGL(gather_pooled_kernel, dim3(batch), dim3(256), 0, stream, (const uint8_t*)x, xb, (const uint8_t*)h, hb, pool_idx, width,
                       (uint8_t*)xc, (uint8_t*)hc);
    KEMR_CHECK_LAUNCH("gather_pooled_kernel");
    return KEMR_OK;
}

// block-wide sum over 256 threads; red must hold 4 floats; all threads get the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Pooling tail, part 1: pooled row -> LayerNorm -> 64 output columns of  y @ proj  per workgroup.
// grid (ceil(d / 64), batch): every workgroup re-normalises its row (width <= 1280 floats, trivial) so that no
// intermediate buffer is needed; wave w owns 16 columns, lane (g = lane >> 2, jq = lane & 3) accumulates columns
// 4*jq..4*jq+3 over the rows i = g (mod 16) with 16-byte loads of the fp32 projection, then 4 shuffle steps.
template <typename XT>
__global__ __launch_bounds__(256) void tail_proj_kernel(const XT* __restrict__ x, const bf16_t* __restrict__ delta,
                                                        const bf16_t* __restrict__ delta2, const int32_t* __restrict__ ids,
                                                        int tokens, int width, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ proj,
                                                        int d, float* __restrict__ out, const int* __restrict__ row_start) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* y = (float*)smem;          // [width] normalised row
    float* red = y + width;           // [4]
    int* pool = (int*)(red + 4);      // [1]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    if (tid < 64) {
        int best_t = 0;
        if (ids) {                    // first position of the row maximum (torch.argmax semantics)
            int best_v = INT_MIN;
            best_t = INT_MAX;
            for (int t = tid; t < tokens; t += 64) {
                const int v = ids[(size_t)b * tokens + t];
                if (v > best_v) { best_v = v; best_t = t; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int ov = __shfl_xor(best_v, o), ot = __shfl_xor(best_t, o);
                if (ov > best_v || (ov == best_v && ot < best_t)) { best_v = ov; best_t = ot; }
            }
        }
        if (tid == 0) *pool = best_t;
    }
    __syncthreads();
    // packed rows (text tower): the pooled position inside the rows this text owns (its length covers the end-of-text token when
    // the caller kept the contract; clamped so that a short length reads its last row instead of a neighbour's)
    size_t prow = (size_t)b * tokens + *pool;
    if (row_start) {
        const int r0 = row_start[b], len = row_start[b + 1] - r0;
        prow = (size_t)r0 + (*pool < len ? *pool : len - 1);
    }
    const XT* xr = x + prow * width;
    const bf16_t* dr = delta ? delta + prow * width : nullptr;
    const bf16_t* dr2 = delta2 ? delta2 + prow * width : nullptr;

    float s = 0.f;
    for (int i = tid; i < width; i += 256) {           // pooled row (+ the pending residual update) into LDS
        float v;
        if constexpr (sizeof(XT) == 4) v = xr[i];
        else if constexpr (sizeof(XT) == 3) v = f24_to_f32(((const bf16_t*)xr)[i], ((const uint8_t*)xr)[2 * (size_t)width + i]);
        else v = bf16_to_f32(xr[i]);
        if (dr) v += bf16_to_f32(dr[i]);       // same order as the fused LayerNorm updates: (x + d1) + d2
        if (dr2) v += bf16_to_f32(dr2[i]);
        y[i] = v;
        s += v;
    }
    const float mean = block_sum(s, red) / width;
    float q = 0.f;
    for (int i = tid; i < width; i += 256) { const float c = y[i] - mean; q += c * c; }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / width + 1e-5f);
    for (int i = tid; i < width; i += 256) y[i] = (y[i] - mean) * rstd * gamma[i] + beta[i];
    __syncthreads();

    const int j = blockIdx.x * 64 + wid * 16 + (lane & 3) * 4;     // first of this lane's 4 columns
    const int g = lane >> 2;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < d) {                       // d % 4 == 0 (checked by the launcher)
#pragma unroll 8
        for (int i = g; i < width; i += 16) {
            const float yi = y[i];
            const float4 p = *(const float4*)(proj + (size_t)i * d + j);
            acc.x = fmaf(yi, p.x, acc.x); acc.y = fmaf(yi, p.y, acc.y);
            acc.z = fmaf(yi, p.z, acc.z); acc.w = fmaf(yi, p.w, acc.w);
        }
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
        acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if (g == 0 && j < d) *(float4*)(out + (size_t)b * d + j) = acc;
}

// Pooling tail, part 2: x / ||x||_2 per row, in place (one wave per row; no eps, like the reference's `x / x.norm()`).
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ x, int rows, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* r = x + (size_t)row * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += r[i] * r[i];
    const float inv = 1.0f / sqrtf(wave_sum(s));
    for (int i = lane; i < d; i += 64) r[i] *= inv;
}

int launch_tail(const void* x, int x_dtype, const bf16_t* delta, const bf16_t* delta2, const int32_t* ids, int batch, int tokens, int width, const float* gamma,
                const float* beta, const float* proj, int d, int normalize, float* out, hipStream_t stream, const int* row_start) {
    if (batch <= 0) return KEMR_OK;
    if (d % 4 != 0 || d <= 0) KEMR_FAIL(KEMR_ERR_INVALID, "tail: embed_dim %d must be a positive multiple of 4", d);
    if (batch > 65535) KEMR_FAIL(KEMR_ERR_INVALID, "tail: batch %d > 65535", batch);
    const size_t smem = (size_t)width * 4 + 32;
    ProfScope prof(PROF_OTHER, stream);
    const dim3 grid((d + 63) / 64, batch);
    if (x_dtype == KEMR_BF16)
        hipLaunchKernelGGL(tail_proj_kernel<bf16_t>, grid, dim3(256), smem, stream, (const bf16_t*)x, delta, delta2, ids, tokens, width, gamma, beta, proj, d, out, row_start);
    else if (x_dtype == KEMR_F24)
        hipLaunchKernelGGL(tail_proj_kernel<f24_t>, grid, dim3(256), smem, stream, (const f24_t*)x, delta, delta2, ids, tokens, width, gamma, beta, proj, d, out, row_start);
    else
        hipLaunchKernelGGL(tail_proj_kernel<float>, grid, dim3(256), smem, stream, (const float*)x, delta, delta2, ids, tokens, width, gamma, beta, proj, d, out, row_start);
    KEMR_CHECK_LAUNCH("tail_proj_kernel");
    if (normalize) {
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3((batch + 3) / 4), dim3(256), 0, stream, out, batch, d);
        KEMR_CHECK_LAUNCH("l2norm_rows_kernel");
    }
    return KEMR_OK;
}

}  // namespace kemr
