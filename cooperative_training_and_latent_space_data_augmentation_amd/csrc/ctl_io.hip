// the data-facing ends of the hot path, kept on the GPU so that neither the validation loop nor the
// input pipeline forces a host round trip per batch.  All HBM-bound byte / integer work: grid-stride loops, integer atomics only
// (deterministic), no MFMA.
//   ctl_confusion_hist    runningScore._fast_hist (medseg/common_utils/metrics.py:18-23) accumulated on device
//   ctl_rescale_intensity min-max rescale per (n,c) plane (medseg/common_utils/basic_operations.py:232-245)
//   ctl_noise_clamp       clamp(x + 0.05*N(0,1), 0, 1) (medseg/train_adv_supervised_segmentation_triplet.py:185-187)
//   ctl_crop_or_pad       centre crop / zero pad of [n,h,w] arrays (medseg/common_utils/basic_operations.py:173-220)
#include "ctl_common.h"

#define IOB 256
#define S_ (hipStream_t) stream

static inline unsigned io_blocks(int64_t items, int cap = 2048) {
    int64_t b = ctl_cdiv64(items, IOB);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------ confusion matrix
__global__ __launch_bounds__(IOB) void confusion_hist_kernel(const int64_t* __restrict__ lt, const uint8_t* __restrict__ lp,
                                                              int64_t count, int n, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int sh[256];
    const int bins = n * n;
    for (int b = threadIdx.x; b < bins; b += IOB) sh[b] = 0u;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * IOB;
    for (int64_t i = (int64_t)blockIdx.x * IOB + threadIdx.x; i < count; i += stride) {
        const int64_t t = lt[i];
        const int p = lp[i];
        if (t >= 0 && t < n && p < n) atomicAdd(&sh[(int)t * n + p], 1u);      // labels outside [0, n) are ignored, as upstream
    }
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += IOB)
        if (sh[b]) atomicAdd(&hist[b], (unsigned long long)sh[b]);             // integer atomics: order-independent
}

extern "C" int ctl_confusion_hist(const int64_t* label_true, const uint8_t* label_pred, int64_t count, int32_t n_class,
                                  int64_t* hist, ctl_stream stream) {
    CTL_REQUIRE(label_true && label_pred && hist && count > 0 && n_class >= 1 && n_class <= 16, "confusion_hist: bad arguments");
    confusion_hist_kernel<<<dim3(io_blocks(count, 1024)), dim3(IOB), 0, S_>>>(label_true, label_pred, count, n_class,
                                                                             (unsigned long long*)hist);
    CTL_LAUNCH_CHECK("confusion_hist");
    return CTL_OK;
}

// ------------------------------------------------------------------------------------------------ min-max rescale
#define RS_BPP 64      // blocks per plane
__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* sm) {
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w * 2] = mn; sm[w * 2 + 1] = mx; }
    __syncthreads();
    mn = sm[0]; mx = sm[1];
    for (int i = 1; i < IOB / 64; ++i) { mn = fminf(mn, sm[i * 2]); mx = fmaxf(mx, sm[i * 2 + 1]); }
    __syncthreads();
}
__global__ __launch_bounds__(IOB) void minmax_partial_kernel(const float* __restrict__ x, int64_t plane_elems, float* __restrict__ partial) {
    __shared__ float sm[2 * IOB / 64];
    const float* xp = x + (int64_t)blockIdx.y * plane_elems;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * IOB + threadIdx.x; i < plane_elems; i += (int64_t)gridDim.x * IOB) {
        const float v = xp[i];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    block_minmax(mn, mx, sm);
    if (threadIdx.x == 0) { partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2] = mn; partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = mx; }
}
__global__ __launch_bounds__(IOB) void rescale_apply_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                             int64_t plane_elems, float new_min, float range, float eps,
                                                             float* __restrict__ out) {
#pragma clang fp contract(off)      // mul and add must round separately, as torch does (HIP's __fmul_rn/__fadd_rn wrappers are inlined
                                    // with their own 'contract' flags and would still fuse)
    __shared__ float sm[2 * IOB / 64];
    float mn = INFINITY, mx = -INFINITY;
    if (threadIdx.x < gridDim.x) {
        mn = partial[((int64_t)blockIdx.y * gridDim.x + threadIdx.x) * 2];
        mx = partial[((int64_t)blockIdx.y * gridDim.x + threadIdx.x) * 2 + 1];
    }
    block_minmax(mn, mx, sm);
    // (x - min) / (max - min + eps) * (new_max - new_min) + new_min with torch's operation order and one rounding per operation
    const float den = (mx - mn) + eps;
    const float* xp = x + (int64_t)blockIdx.y * plane_elems;
    float* op = out + (int64_t)blockIdx.y * plane_elems;
    for (int64_t i = (int64_t)blockIdx.x * IOB + threadIdx.x; i < plane_elems; i += (int64_t)gridDim.x * IOB)
        op[i] = ((xp[i] - mn) / den) * range + new_min;       // plain operators: the contract(off) pragma above governs them
}

extern "C" size_t ctl_rescale_intensity_ws_floats(int32_t planes) { return planes > 0 ? (size_t)planes * RS_BPP * 2 : 0; }
extern "C" int ctl_rescale_intensity(const float* x, float* out, float* workspace, int32_t planes, int64_t plane_elems,
                                     float new_min, float new_max, float eps, ctl_stream stream) {
    CTL_REQUIRE(x && out && workspace && planes > 0 && plane_elems > 0, "rescale_intensity: bad arguments");
    const dim3 grid(RS_BPP, (unsigned)planes), blk(IOB);
    minmax_partial_kernel<<<grid, blk, 0, S_>>>(x, plane_elems, workspace);
    rescale_apply_kernel<<<grid, blk, 0, S_>>>(x, workspace, plane_elems, new_min, new_max - new_min, eps, out);
    CTL_LAUNCH_CHECK("rescale_intensity");
    return CTL_OK;
}

// ------------------------------------------------------------------------------------------------ input noise
__device__ __forceinline__ uint64_t io_mix(uint64_t z) {       // splitmix64 finaliser (same generator family as ctl_mask.hip)
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(IOB) void noise_clamp_kernel(const float* __restrict__ x, const float* __restrict__ noise, uint64_t seed,
                                                           float sigma, float lo, float hi, float* __restrict__ out, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * IOB;
    for (int64_t i = (int64_t)blockIdx.x * IOB + threadIdx.x; i < count; i += stride) {
        float nz;
        if (noise) {
            nz = noise[i];
        } else {      // Box-Muller on two counter-hash uniforms: stateless, reproducible from (seed, index)
            const uint64_t h = io_mix(seed ^ io_mix((uint64_t)i));
            const float u1 = ((float)((h >> 40) + 1)) * (1.0f / 16777216.0f);        // (0, 1]
            const float u2 = (float)((h >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);   // [0, 1)
            nz = sigma * sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
        }
        out[i] = fminf(fmaxf(__fadd_rn(x[i], nz), lo), hi);
    }
}
extern "C" int ctl_noise_clamp(const float* x, const float* noise, uint64_t seed, float sigma, float lo, float hi, float* out,
                               int64_t count, ctl_stream stream) {
    CTL_REQUIRE(x && out && count > 0 && lo <= hi, "noise_clamp: bad arguments");
    noise_clamp_kernel<<<dim3(io_blocks(count)), dim3(IOB), 0, S_>>>(x, noise, seed, sigma, lo, hi, out, count);
    CTL_LAUNCH_CHECK("noise_clamp");
    return CTL_OK;
}

// ------------------------------------------------------------------------------------------------ crop or pad
template <typename T>
__global__ __launch_bounds__(IOB) void crop_or_pad_kernel(const T* __restrict__ src, T* __restrict__ dst, int n, int h, int w, int nh,
                                                           int nw, int hs, int ws) {
    const int64_t total = (int64_t)n * nh * nw;
    const int64_t stride = (int64_t)gridDim.x * IOB;
    for (int64_t i = (int64_t)blockIdx.x * IOB + threadIdx.x; i < total; i += stride) {
        const int x = (int)(i % nw);
        const int64_t r = i / nw;
        const int y = (int)(r % nh);
        const int64_t b = r / nh;
        const int sy = y + hs, sx = x + ws;           // hs, ws = floor((size - new) / 2): negative when padding
        T v = (T)0;
        if (sy >= 0 && sy < h && sx >= 0 && sx < w) v = src[(b * h + sy) * w + sx];
        dst[i] = v;
    }
}
static inline int floordiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
extern "C" int ctl_crop_or_pad(const void* src, void* dst, int32_t elem_bytes, int32_t n, int32_t h, int32_t w, int32_t new_h,
                               int32_t new_w, ctl_stream stream) {
    CTL_REQUIRE(src && dst && n > 0 && h > 0 && w > 0 && new_h > 0 && new_w > 0, "crop_or_pad: bad arguments");
    const int hs = floordiv2(h - new_h), ws = floordiv2(w - new_w);
    const dim3 grid(io_blocks((int64_t)n * new_h * new_w)), blk(IOB);
    if (elem_bytes == 4) crop_or_pad_kernel<uint32_t><<<grid, blk, 0, S_>>>((const uint32_t*)src, (uint32_t*)dst, n, h, w, new_h, new_w, hs, ws);
    else if (elem_bytes == 8) crop_or_pad_kernel<uint64_t><<<grid, blk, 0, S_>>>((const uint64_t*)src, (uint64_t*)dst, n, h, w, new_h, new_w, hs, ws);
    else if (elem_bytes == 1) crop_or_pad_kernel<uint8_t><<<grid, blk, 0, S_>>>((const uint8_t*)src, (uint8_t*)dst, n, h, w, new_h, new_w, hs, ws);
    else CTL_FAIL(CTL_EINVAL, "crop_or_pad: element size %d (1, 4 or 8 bytes)", elem_bytes);
    CTL_LAUNCH_CHECK("crop_or_pad");
    return CTL_OK;
}
