// CLIP image preprocessing on the GPU (the reference applies the callable it gets from
// clip.load per sample on the host: /root/reference/src/clip/datasets/clip_dataset.py:110-125):
//   uint8 HWC RGB -> Resize(n_px, bicubic, shorter side) -> CenterCrop(n_px) -> / 255 -> (x - mean) / std -> fp32 CHW.
// The resize is Pillow's (Image.resize(..., BICUBIC), the dependency the reference's transform resolves to; libImaging
// Resample.c as shipped in Pillow 12.2.0): an antialiased separable filter (support 2 * max(scale, 1), a = -0.5 cubic),
// coefficients normalised in double and rounded to 22-bit fixed point, horizontal pass first, 8-bit clipped
// intermediates.  Integer work, so it is reproduced bit for bit: the host computes the same coefficient tables (double
// arithmetic in the same order) for the 2 x n_px output rows / columns that survive the crop, the kernels do the int32
// accumulation.  Only the input rows the vertical pass needs are filtered horizontally.
#include "common.h"

#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#pragma STDC FP_CONTRACT OFF      // the coefficient arithmetic must round like Pillow's plain C doubles: no fused multiply-adds

namespace kemr {

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

struct Axis {
    bool resample = false;           // false: in_size == out_size, the pass is a copy
    int ksize = 0, first = 0, count = 0;
    int in_lo = 0, in_hi = 0;        // input index range touched by outputs [first, first + count)
    std::vector<int32_t> kk;         // [count, ksize]
    std::vector<int32_t> bounds;     // [count, 2]: xmin, number of taps
};

// Resample.c precompute_coeffs + normalize_coeffs_8bpc, for the output indices [first, first + count) only
void precompute(int in_size, int out_size, int first, int count, Axis& ax) {
    ax.first = first; ax.count = count;
    ax.resample = in_size != out_size;
    if (!ax.resample) { ax.in_lo = first; ax.in_hi = first + count; return; }
    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    ax.ksize = (int)ceil(support) * 2 + 1;
    ax.kk.assign((size_t)count * ax.ksize, 0);
    ax.bounds.assign((size_t)count * 2, 0);
    ax.in_lo = in_size; ax.in_hi = 0;
    std::vector<double> k(ax.ksize);
    for (int i = 0; i < count; ++i) {
        const int xx = first + i;
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < xmax; ++x)
            ax.kk[(size_t)i * ax.ksize + x] = k[x] < 0 ? (int32_t)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int32_t)(0.5 + k[x] * (1 << PRECISION_BITS));
        ax.bounds[2 * i] = xmin; ax.bounds[2 * i + 1] = xmax;
        if (xmin < ax.in_lo) ax.in_lo = xmin;
        if (xmin + xmax > ax.in_hi) ax.in_hi = xmin + xmax;
    }
}

struct Plan {
    int nw = 0, nh = 0, left = 0, top = 0;
    Axis hx, vy;
    std::vector<int32_t> table;      // hx.kk | hx.bounds | vy.kk | vy.bounds, as uploaded
    size_t off_hb = 0, off_vk = 0, off_vb = 0, temp_off = 0, bytes = 0;
};

const Plan& plan_for(int height, int width, int n) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, Plan> cache;      // entries are never modified once inserted
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(height, width, n);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    Plan p;
    // Resize(n): shorter side -> n, the other side int(n * long / short) (torchvision / clip _transform)
    if (width <= height) { p.nw = n; p.nh = (int)((double)n * height / width); }
    else { p.nw = (int)((double)n * width / height); p.nh = n; }
    // CenterCrop(n): int(round((size - n) / 2.0)) with Python's round-half-to-even
    p.left = (int)nearbyint((p.nw - n) / 2.0);
    p.top = (int)nearbyint((p.nh - n) / 2.0);
    precompute(width, p.nw, p.left, n, p.hx);
    precompute(height, p.nh, p.top, n, p.vy);
    p.table = p.hx.kk;
    p.off_hb = p.table.size(); p.table.insert(p.table.end(), p.hx.bounds.begin(), p.hx.bounds.end());
    p.off_vk = p.table.size(); p.table.insert(p.table.end(), p.vy.kk.begin(), p.vy.kk.end());
    p.off_vb = p.table.size(); p.table.insert(p.table.end(), p.vy.bounds.begin(), p.vy.bounds.end());
    p.temp_off = (size_t)round_up((int64_t)p.table.size() * 4, 256);
    p.bytes = p.temp_off + (size_t)round_up((int64_t)(p.vy.in_hi - p.vy.in_lo) * n * 3, 256);
    return cache.emplace(key, std::move(p)).first->second;
}

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass over the input rows [row_lo, row_lo + rows): temp[r][xx][c], xx = cropped output column
__global__ __launch_bounds__(256) void preprocess_h_kernel(const uint8_t* __restrict__ img, int width, int row_lo, int rows, int n,
                                                           const int32_t* __restrict__ kk, const int32_t* __restrict__ bounds,
                                                           int ksize, int resample, int left, uint8_t* __restrict__ temp) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * n) return;
    const int r = gid / n, xx = gid - r * n;
    const uint8_t* src = img + ((size_t)(row_lo + r) * width) * 3;
    uint8_t* dst = temp + ((size_t)r * n + xx) * 3;
    if (!resample) {
        const uint8_t* s = src + (size_t)(left + xx) * 3;
        dst[0] = s[0]; dst[1] = s[1]; dst[2] = s[2];
        return;
    }
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int32_t* k = kk + (size_t)xx * ksize;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) {
        const uint8_t* s = src + (size_t)(xmin + x) * 3;
        const int c = k[x];
        s0 += s[0] * c; s1 += s[1] * c; s2 += s[2] * c;
    }
    dst[0] = (uint8_t)clip8(s0 >> PRECISION_BITS); dst[1] = (uint8_t)clip8(s1 >> PRECISION_BITS); dst[2] = (uint8_t)clip8(s2 >> PRECISION_BITS);
}

// vertical pass + ToTensor + Normalize: out[c][yy][xx] = (v / 255 - mean[c]) / std[c], fp32 operations in torch's order
__global__ __launch_bounds__(256) void preprocess_v_kernel(const uint8_t* __restrict__ temp, int row_lo, int n,
                                                           const int32_t* __restrict__ kk, const int32_t* __restrict__ bounds,
                                                           int ksize, int resample, int top, float m0, float m1, float m2,
                                                           float d0, float d1, float d2, float* __restrict__ out) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * n) return;
    const int yy = gid / n, xx = gid - yy * n;
    int v0, v1, v2;
    if (!resample) {
        const uint8_t* s = temp + ((size_t)(top + yy - row_lo) * n + xx) * 3;
        v0 = s[0]; v1 = s[1]; v2 = s[2];
    } else {
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const int32_t* k = kk + (size_t)yy * ksize;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < cnt; ++y) {
            const uint8_t* s = temp + ((size_t)(ymin + y - row_lo) * n + xx) * 3;
            const int c = k[y];
            s0 += s[0] * c; s1 += s[1] * c; s2 += s[2] * c;
        }
        v0 = clip8(s0 >> PRECISION_BITS); v1 = clip8(s1 >> PRECISION_BITS); v2 = clip8(s2 >> PRECISION_BITS);
    }
    const size_t plane = (size_t)n * n;
    out[gid] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.0f), m0), d0);
    out[plane + gid] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.0f), m1), d1);
    out[2 * plane + gid] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.0f), m2), d2);
}

// ---- one launch pair per BATCH: images of any sizes packed back to back, one descriptor per image ----------------------
struct PreImg {
    long long img_off;     // byte offset of the image in the packed uint8 buffer
    long long temp_off;    // byte offset of its horizontally filtered rows in the temp area
    int width, row_lo, rows, left, top;
    int ksize_h, resample_h, ksize_v, resample_v;
    int tab_hk, tab_hb, tab_vk, tab_vb;      // int32 offsets of its plan's tables in the table area
    int zero_out;          // 1: an item the caller could not decode (height = width = 0): its output is 0.0f AFTER normalisation,
                           // what the reference's dataset feeds the encoder then (clip_dataset.py:120-125: torch.zeros(3, 224, 224))
};
static_assert(sizeof(PreImg) == 72, "descriptor layout");

__global__ __launch_bounds__(256) void preprocess_h_batch_kernel(const uint8_t* __restrict__ packed, const PreImg* __restrict__ desc,
                                                                 const int32_t* __restrict__ tab, uint8_t* __restrict__ temp_all, int n) {
    const PreImg d = desc[blockIdx.y];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= d.rows * n) return;
    const int r = gid / n, xx = gid - r * n;
    const uint8_t* src = packed + d.img_off + ((size_t)(d.row_lo + r) * d.width) * 3;
    uint8_t* dst = temp_all + d.temp_off + ((size_t)r * n + xx) * 3;
    if (!d.resample_h) {
        const uint8_t* s = src + (size_t)(d.left + xx) * 3;
        dst[0] = s[0]; dst[1] = s[1]; dst[2] = s[2];
        return;
    }
    const int32_t* bounds = tab + d.tab_hb;
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int32_t* k = tab + d.tab_hk + (size_t)xx * d.ksize_h;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) {
        const uint8_t* s = src + (size_t)(xmin + x) * 3;
        const int c = k[x];
        s0 += s[0] * c; s1 += s[1] * c; s2 += s[2] * c;
    }
    dst[0] = (uint8_t)clip8(s0 >> PRECISION_BITS); dst[1] = (uint8_t)clip8(s1 >> PRECISION_BITS); dst[2] = (uint8_t)clip8(s2 >> PRECISION_BITS);
}

__global__ __launch_bounds__(256) void preprocess_v_batch_kernel(const PreImg* __restrict__ desc, const int32_t* __restrict__ tab,
                                                                 const uint8_t* __restrict__ temp_all, int n, float m0, float m1, float m2,
                                                                 float d0, float d1, float d2, float* __restrict__ out_all) {
    const PreImg d = desc[blockIdx.y];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * n) return;
    const int yy = gid / n, xx = gid - yy * n;
    const uint8_t* temp = temp_all + d.temp_off;
    const size_t plane = (size_t)n * n;
    float* out = out_all + (size_t)blockIdx.y * 3 * plane;
    if (d.zero_out) {
        out[gid] = 0.0f; out[plane + gid] = 0.0f; out[2 * plane + gid] = 0.0f;
        return;
    }
    int v0, v1, v2;
    if (!d.resample_v) {
        const uint8_t* s = temp + ((size_t)(d.top + yy - d.row_lo) * n + xx) * 3;
        v0 = s[0]; v1 = s[1]; v2 = s[2];
    } else {
        const int32_t* bounds = tab + d.tab_vb;
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const int32_t* k = tab + d.tab_vk + (size_t)yy * d.ksize_v;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < cnt; ++y) {
            const uint8_t* s = temp + ((size_t)(ymin + y - d.row_lo) * n + xx) * 3;
            const int c = k[y];
            s0 += s[0] * c; s1 += s[1] * c; s2 += s[2] * c;
        }
        v0 = clip8(s0 >> PRECISION_BITS); v1 = clip8(s1 >> PRECISION_BITS); v2 = clip8(s2 >> PRECISION_BITS);
    }
    out[gid] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.0f), m0), d0);
    out[plane + gid] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.0f), m1), d1);
    out[2 * plane + gid] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.0f), m2), d2);
}

// Pinned staging for the per-batch descriptor blob: a ring of slots, each guarded by an event recorded after its copy; a slot
// is reused only after that copy has finished (with 8 slots the wait only triggers when the host is 8 batches ahead).
int staging_slot(size_t bytes, char** host, hipEvent_t* done) {
    struct Slot { char* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
    static std::mutex mu;
    static Slot ring[8];
    static unsigned next = 0;
    std::lock_guard<std::mutex> lock(mu);
    Slot& sl = ring[next++ % 8];
    if (!sl.ev) KEMR_CHECK_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.used) KEMR_CHECK_HIP(hipEventSynchronize(sl.ev));
    if (sl.cap < bytes) {
        if (sl.host) KEMR_CHECK_HIP(hipHostFree(sl.host));
        sl.host = nullptr; sl.cap = 0;
        const size_t cap = (size_t)round_up((int64_t)bytes * 2, 65536);
        KEMR_CHECK_HIP(hipHostMalloc((void**)&sl.host, cap, hipHostMallocDefault));
        sl.cap = cap;
    }
    sl.used = true;
    *host = sl.host;
    *done = sl.ev;
    return KEMR_OK;
}

// host side of a batch: descriptors + the tables of the distinct sizes, and where everything lives in the workspace
struct BatchLayout {
    std::vector<PreImg> desc;
    std::vector<int32_t> tables;
    size_t tab_off = 0, temp_off = 0, bytes = 0;
    int max_rows = 0;
};

int layout_batch(const int32_t* heights, const int32_t* widths, const int64_t* offsets, int batch, int n, BatchLayout& L) {
    std::map<std::pair<int, int>, std::array<int, 4>> seen;       // (h, w) -> table offsets of its plan
    L.desc.resize(batch);
    size_t temp = 0;
    for (int b = 0; b < batch; ++b) {
        const int h = heights[b], w = widths[b];
        if (h == 0 && w == 0) {                  // undecodable item: no input bytes, zeros out (PreImg::zero_out)
            PreImg& d = L.desc[b];
            memset(&d, 0, sizeof(d));
            d.img_off = offsets ? offsets[b] : 0;
            d.temp_off = (long long)temp;
            d.zero_out = 1;
            continue;
        }
        if (h <= 0 || w <= 0 || h > 32768 || w > 32768) KEMR_FAIL(KEMR_ERR_INVALID, "preprocess batch: bad image size %d x %d (image %d)", h, w, b);
        const Plan& p = plan_for(h, w, n);
        auto it = seen.find({h, w});
        if (it == seen.end()) {
            const int base = (int)L.tables.size();
            L.tables.insert(L.tables.end(), p.table.begin(), p.table.end());
            it = seen.emplace(std::make_pair(h, w), std::array<int, 4>{base, base + (int)p.off_hb, base + (int)p.off_vk, base + (int)p.off_vb}).first;
        }
        PreImg& d = L.desc[b];
        d.img_off = offsets ? offsets[b] : 0;
        d.temp_off = (long long)temp;
        d.width = w; d.row_lo = p.vy.in_lo; d.rows = p.vy.in_hi - p.vy.in_lo; d.left = p.left; d.top = p.top;
        d.ksize_h = p.hx.ksize; d.resample_h = p.hx.resample ? 1 : 0; d.ksize_v = p.vy.ksize; d.resample_v = p.vy.resample ? 1 : 0;
        d.tab_hk = it->second[0]; d.tab_hb = it->second[1]; d.tab_vk = it->second[2]; d.tab_vb = it->second[3];
        d.zero_out = 0;
        temp += (size_t)round_up((int64_t)d.rows * n * 3, 16);
        if (d.rows > L.max_rows) L.max_rows = d.rows;
    }
    L.tab_off = (size_t)round_up((int64_t)batch * sizeof(PreImg), 256);
    L.temp_off = L.tab_off + (size_t)round_up((int64_t)L.tables.size() * 4, 256);
    L.bytes = L.temp_off + (size_t)round_up((int64_t)temp, 256);
    return KEMR_OK;
}

}  // namespace

}  // namespace kemr

using namespace kemr;

extern "C" size_t kemr_preprocess_batch_workspace_bytes(const int32_t* heights, const int32_t* widths, int batch, int n_px) {
    if (!heights || !widths || batch <= 0 || n_px <= 0 || n_px > 1024) return 0;
    BatchLayout L;
    if (layout_batch(heights, widths, nullptr, batch, n_px, L) != KEMR_OK) return 0;
    return L.bytes;
}

extern "C" int kemr_preprocess_u8_batch(const unsigned char* packed_dev, const int64_t* offsets, const int32_t* heights,
                                        const int32_t* widths, int batch, int n_px, float* out_dev, void* workspace_dev,
                                        size_t workspace_bytes, void* stream) {
    if (batch == 0) return KEMR_OK;
    if (!offsets || !heights || !widths || !out_dev || batch < 0) KEMR_FAIL(KEMR_ERR_INVALID, "preprocess batch: null argument");
    if (n_px <= 0 || n_px > 1024) KEMR_FAIL(KEMR_ERR_INVALID, "preprocess batch: bad output size %d", n_px);
    BatchLayout L;
    KEMR_TRY(layout_batch(heights, widths, offsets, batch, n_px, L));
    if (!packed_dev && L.max_rows > 0) KEMR_FAIL(KEMR_ERR_INVALID, "preprocess batch: null image buffer");
    if (!workspace_dev || workspace_bytes < L.bytes) KEMR_FAIL(KEMR_ERR_WORKSPACE, "preprocess batch: workspace too small: %zu < %zu bytes", workspace_bytes, L.bytes);
    if ((uintptr_t)workspace_dev % 256) KEMR_FAIL(KEMR_ERR_WORKSPACE, "preprocess batch: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace_dev;
    // descriptors + tables cross in ONE copy out of a pinned slot: a pageable source would make the "async" copy wait for
    // the stream, i.e. for the encoders queued before it, and stall the loader thread behind the GPU
    const size_t blob = L.tab_off + L.tables.size() * 4;
    char* host = nullptr;
    hipEvent_t done = nullptr;
    KEMR_TRY(staging_slot(blob, &host, &done));
    memcpy(host, L.desc.data(), L.desc.size() * sizeof(PreImg));
    if (!L.tables.empty()) memcpy(host + L.tab_off, L.tables.data(), L.tables.size() * 4);
    KEMR_CHECK_HIP(hipMemcpyAsync(ws, host, blob, hipMemcpyHostToDevice, s));
    KEMR_CHECK_HIP(hipEventRecord(done, s));
    const int n = n_px;
    ProfScope prof(PROF_OTHER, s);
    if (L.max_rows > 0) {                        // 0: every item of the batch is a zero_out item
        hipLaunchKernelGGL(preprocess_h_batch_kernel, dim3((unsigned)(((size_t)L.max_rows * n + 255) / 256), (unsigned)batch), dim3(256), 0, s,
                           packed_dev, (const PreImg*)ws, (const int32_t*)(ws + L.tab_off), (uint8_t*)(ws + L.temp_off), n);
        KEMR_CHECK_LAUNCH("preprocess_h_batch_kernel");
    }
    hipLaunchKernelGGL(preprocess_v_batch_kernel, dim3((unsigned)((n * n + 255) / 256), (unsigned)batch), dim3(256), 0, s,
                       (const PreImg*)ws, (const int32_t*)(ws + L.tab_off), (const uint8_t*)(ws + L.temp_off), n, 0.48145466f, 0.4578275f,
                       0.40821073f, 0.26862954f, 0.26130258f, 0.27577711f, out_dev);
    KEMR_CHECK_LAUNCH("preprocess_v_batch_kernel");
    return KEMR_OK;
}

extern "C" size_t kemr_preprocess_workspace_bytes(int height, int width, int n_px) {
    if (height <= 0 || width <= 0 || n_px <= 0 || height > 32768 || width > 32768 || n_px > 1024) return 0;
    return plan_for(height, width, n_px).bytes;
}

extern "C" int kemr_preprocess_u8(const unsigned char* img_dev, int height, int width, int n_px, float* out_dev,
                                  void* workspace_dev, size_t workspace_bytes, void* stream) {
    if (!img_dev || !out_dev) KEMR_FAIL(KEMR_ERR_INVALID, "preprocess: null argument");
    if (height <= 0 || width <= 0 || n_px <= 0 || height > 32768 || width > 32768 || n_px > 1024)
        KEMR_FAIL(KEMR_ERR_INVALID, "preprocess: bad sizes %d x %d -> %d", height, width, n_px);
    const Plan& p = plan_for(height, width, n_px);
    if (!workspace_dev || workspace_bytes < p.bytes) KEMR_FAIL(KEMR_ERR_WORKSPACE, "preprocess: workspace too small: %zu < %zu bytes", workspace_bytes, p.bytes);
    if ((uintptr_t)workspace_dev % 256) KEMR_FAIL(KEMR_ERR_WORKSPACE, "preprocess: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int n = n_px;
    int32_t* tab = (int32_t*)workspace_dev;
    uint8_t* temp = (uint8_t*)workspace_dev + p.temp_off;
    if (!p.table.empty()) KEMR_CHECK_HIP(hipMemcpyAsync(tab, p.table.data(), p.table.size() * 4, hipMemcpyHostToDevice, s));
    const int row_lo = p.vy.in_lo, rows = p.vy.in_hi - p.vy.in_lo;
    ProfScope prof(PROF_OTHER, s);
    hipLaunchKernelGGL(preprocess_h_kernel, dim3((unsigned)(((size_t)rows * n + 255) / 256)), dim3(256), 0, s, img_dev, width, row_lo, rows, n,
                       tab, tab + p.off_hb, p.hx.ksize, p.hx.resample ? 1 : 0, p.left, temp);
    KEMR_CHECK_LAUNCH("preprocess_h_kernel");
    // CLIP's normalisation constants
    hipLaunchKernelGGL(preprocess_v_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, s, temp, row_lo, n, tab + p.off_vk,
                       tab + p.off_vb, p.vy.ksize, p.vy.resample ? 1 : 0, p.top, 0.48145466f, 0.4578275f, 0.40821073f, 0.26862954f,
                       0.26130258f, 0.27577711f, out_dev);
    KEMR_CHECK_LAUNCH("preprocess_v_kernel");
    return KEMR_OK;
}
