// Latent-space hard-example generator, device side (reference: medseg/models/model_util.py:224-249 channel-wise,
// 285-312 spatial-wise; F.dropout2d branch of model.py:333).
//
//   score  : signed mean of dL/dz over H*W (channel mode) or over C (spatial mode)        -- pass 1 over grad
//   select : entry i is masked  <=>  score_i > sort(score, desc)[k]  <=>  #{j : score_j >= score_i} <= k
//            (strict '>' of the reference, exact under ties; no sort: each block ranks only the entries it applies)
//   apply  : masked = code * mask (mask broadcast over H*W or over C)                     -- pass 1 over code, 1 write
//
// HBM traffic = read grad + read code + write masked = 3*N*C*H*W*4 bytes (+ the tiny score/mask vectors): the kernels
// are pure streams, 16 B per lane, coalesced NHWC.  At the configured size (16x128x16x16 = 2 MiB per tensor) everything
// sits in L2/Infinity Cache and the pair is launch-latency bound; bench.py sweeps sizes to show the HBM-bound regime.
#include <stdlib.h>

#include "ctl_common.h"

#define MB 256
// once-written / once-read streams of the HBM-sized problems: non-temporal policy (tuning: -DCTL_MASK_NT=0 none, 1 stores, 2 + loads)
#ifndef CTL_MASK_NT
#define CTL_MASK_NT 1      // measured (64x128x64x64): 86.5 us with plain stores, 67.2 with non-temporal stores, 70.0 with loads too
#endif
__device__ __forceinline__ void stream_store(f32x4* p, f32x4 v) {
    if (CTL_MASK_NT >= 1) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ f32x4 stream_load(const f32x4* p) {
    if (CTL_MASK_NT >= 2) return __builtin_nontemporal_load(p);
    return *p;
}
// pixels per block in the channel-mode score pass: 64 keeps the tiny configured problem (hw = 256) spread over 64 blocks;
// large problems use 512-pixel slabs (more bytes in flight per block, 8x fewer partial rows)
// (the split fixes the summation order of the channel scores: the three-launch path and the fused kernel share it, so their scores are
// bit-identical; >= ~1024 partial sums in flight where the problem allows)
static inline int score_split_pix(int n, int hw) {
    static const int forced = ctl_tune_int("CTL_MASK_SPLIT", 0);     // tuning hook
    if (forced > 0) return forced;
    int sp = 512;
    while (sp > 64 && (int64_t)n * ctl_cdiv(hw, sp) < 1024) sp >>= 1;
    return sp;
}

// ---- channel mode, pass 1: partial[n][split][c] = sum over the split's pixels of grad[n][pix][c]
__global__ __launch_bounds__(MB) void score_channel_partial_kernel(const f32x4* __restrict__ grad,
                                                                    float* __restrict__ partial, int hw, int cq,
                                                                    int splits, int split_pix) {
    __shared__ f32x4 sm[MB];
    const int n = blockIdx.y, sp = blockIdx.x;
    const int q = threadIdx.x % cq;              // MB % cq == 0 (checked on the host)
    const int prow = threadIdx.x / cq;           // pixel lane inside the block
    const int ppb = MB / cq;                     // pixels in flight per iteration
    const int p0 = sp * split_pix;
    const int p1 = min(hw, p0 + split_pix);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    // 8 loads in flight per thread (a one-load-per-iteration loop keeps 4 KiB per block in flight: latency-bound at ~4 TB/s); the
    // adds keep the order of the rolled loop, and the +0 of a lane past the split leaves a sum unchanged bit for bit
    for (int pix = p0 + prow; pix < p1; pix += 8 * ppb) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pp = pix + u * ppb;
            v[u] = pp < p1 ? stream_load(grad + ((int64_t)n * hw + pp) * cq + q) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < cq) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int k = threadIdx.x; k < MB; k += cq) { const f32x4 v = sm[k]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        reinterpret_cast<f32x4*>(partial)[((int64_t)n * splits + sp) * cq + threadIdx.x] = t;
    }
}
__global__ void score_channel_finalize_kernel(const float* __restrict__ partial, float* __restrict__ score, int c,
                                              int splits, float inv_count) {
    const int n = blockIdx.y;
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += partial[((int64_t)n * splits + sp) * c + ch];
    score[(int64_t)n * c + ch] = s * inv_count;
}

// ---- spatial mode: score[n][pix] = mean over c; a group of cq lanes (cq | 64) owns one pixel
__global__ __launch_bounds__(MB) void score_spatial_kernel(const f32x4* __restrict__ grad, float* __restrict__ score,
                                                            int64_t pixels, int cq, float inv_count) {
    const int ppb = MB / cq;
    const int q = threadIdx.x % cq, prow = threadIdx.x / cq;
    const int64_t stride = (int64_t)gridDim.x * ppb;
    const int64_t npad = ctl_cdiv64(pixels, stride) * stride;      // keep whole waves converged for the shuffles
    for (int64_t pix0 = (int64_t)blockIdx.x * ppb + prow; pix0 < npad; pix0 += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                      // 4 loads in flight per thread
            const int64_t pix = pix0 + u * stride;
            v[u] = pix < pixels ? grad[pix * cq + q] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t pix = pix0 + u * stride;
            float s = (v[u].x + v[u].y) + (v[u].z + v[u].w);
            for (int o = cq >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (q == 0 && pix < pixels) score[pix] = s * inv_count;
        }
    }
}

// ---- select + apply
// grid (slabs, n).  LDS holds the image's whole score row (L <= 8192 floats).  The block's first 8 code quads per thread are requested
// before anything else, so the score row, the ranking and k's load run under their latency.  `partial` (channel mode, optional):
// the split sums of score_channel_partial_kernel -- the block then finalises the score row itself, in the order of
// score_channel_finalize_kernel (bit-identical scores, one launch fewer); block 0 of an image publishes the row to `score_out`.
#define APF 8
template <int MODE>
__global__ __launch_bounds__(MB) void mask_apply_kernel(const f32x4* __restrict__ code, const float* __restrict__ score,
                                                         const float* __restrict__ partial, int splits, float inv_count,
                                                         float* __restrict__ score_out,
                                                         const float* __restrict__ soft_noise, int k_host,
                                                         const int* __restrict__ k_dev, f32x4* __restrict__ masked,
                                                         float* __restrict__ mask_out, int hw, int cq, int slab_pix) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c = cq * 4;
    const int L = (MODE == 0) ? c : hw;
    float* srow = sm;            // [L] scores
    float* mval = sm + L;        // [c] (channel) or [slab_pix] (spatial) mask values
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * slab_pix;
    const int p1 = min(hw, p0 + slab_pix);
    const int64_t base = ((int64_t)n * hw + p0) * cq;
    const int quads = (p1 - p0) * cq;
    f32x4 v[APF];
#pragma unroll
    for (int u = 0; u < APF; ++u) {
        const int e = threadIdx.x + u * MB;
        v[u] = e < quads ? stream_load(code + base + e) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int k = k_dev ? k_dev[0] : k_host;
    if (MODE == 0 && partial) {
        for (int i = threadIdx.x; i < L; i += MB) {
            float a = 0.f;
            for (int sp = 0; sp < splits; ++sp) a += partial[((int64_t)n * splits + sp) * c + i];
            a *= inv_count;
            srow[i] = a;
            if (score_out && blockIdx.x == 0) score_out[(int64_t)n * L + i] = a;
        }
    } else {
        for (int i = threadIdx.x; i < L; i += MB) srow[i] = score[(int64_t)n * L + i];
    }
    __syncthreads();
    k = k < 0 ? 0 : (k > L - 1 ? L - 1 : k);      // a device-side k is not range-checked on the host
    const int first = (MODE == 0) ? 0 : p0;
    const int count = (MODE == 0) ? c : (p1 - p0);
    constexpr int RK = 4;                                     // lanes per ranked entry (interleaved scan + shuffle sum)
    for (int e0 = 0; e0 < count; e0 += MB / RK) {
        const int e = e0 + threadIdx.x / RK, sub = threadIdx.x % RK;
        const bool ok = e < count;
        const int i = first + (ok ? e : 0);
        const float si = srow[i];
        int ge = 0;
#pragma unroll 8
        for (int j = sub; j < L; j += RK) ge += (srow[j] >= si) ? 1 : 0;
        ge += __shfl_xor(ge, 1); ge += __shfl_xor(ge, 2);
        if (ok && sub == 0) {
            float mv = 1.f;
            if (ge <= k) mv = soft_noise ? 0.5f * soft_noise[(int64_t)n * L + i] : 0.f;
            mval[e] = mv;
            if (MODE == 1 || blockIdx.x == 0) mask_out[(int64_t)n * L + i] = mv;
        }
    }
    __syncthreads();
    auto apply = [&](f32x4 t, int e) -> f32x4 {
        if (MODE == 0) {
            const f32x4 m = reinterpret_cast<const f32x4*>(mval)[e % cq];
            t.x *= m.x; t.y *= m.y; t.z *= m.z; t.w *= m.w;
        } else {
            const float m = mval[e / cq];
            t.x *= m; t.y *= m; t.z *= m; t.w *= m;
        }
        return t;
    };
#pragma unroll
    for (int u = 0; u < APF; ++u) {
        const int e = threadIdx.x + u * MB;
        if (e < quads) stream_store(masked + base + e, apply(v[u], e));
    }
    for (int e0 = APF * MB; e0 < quads; e0 += APF * MB) {       // slabs beyond 8 quads per thread: same 8-deep batches
#pragma unroll
        for (int u = 0; u < APF; ++u) {
            const int e = e0 + threadIdx.x + u * MB;
            v[u] = e < quads ? stream_load(code + base + e) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < APF; ++u) {
            const int e = e0 + threadIdx.x + u * MB;
            if (e < quads) stream_store(masked + base + e, apply(v[u], e));
        }
    }
}

// ---- long rows (L > 1024): threshold = sort(score, descending)[k] by a bitonic sort in LDS, one block per image; the apply
// pass is then a pure stream (mask = score > thr), exactly the reference's formulation (model_util.py:231-244).
__global__ __launch_bounds__(MB) void latent_threshold_kernel(const float* __restrict__ score, int L, int Lp2, int k_host,
                                                               const int* __restrict__ k_dev, float* __restrict__ thr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int n = blockIdx.x;
    int k = k_dev ? k_dev[0] : k_host;
    k = k < 0 ? 0 : (k > L - 1 ? L - 1 : k);      // a device-side k is not range-checked on the host (sm[] holds Lp2 >= L entries)
    for (int i = threadIdx.x; i < Lp2; i += MB) sm[i] = (i < L) ? score[(int64_t)n * L + i] : -INFINITY;
    __syncthreads();
    for (int size = 2; size <= Lp2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < (Lp2 >> 1); i += MB) {
                const int lo = 2 * i - (i & (stride - 1));         // index with bit `stride` cleared
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);               // overall order: descending
                const float a = sm[lo], b = sm[hi];
                if ((a < b) == desc) { sm[lo] = b; sm[hi] = a; }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) thr[n] = sm[k];
}

template <int MODE>
__global__ __launch_bounds__(MB) void mask_apply_thr_kernel(const f32x4* __restrict__ code, const float* __restrict__ score,
                                                             const float* __restrict__ soft_noise,
                                                             const float* __restrict__ thr, f32x4* __restrict__ masked,
                                                             float* __restrict__ mask_out, int hw, int cq, int slab_pix) {
    extern __shared__ __attribute__((aligned(16))) float mval[];    // [c] (channel) or [slab_pix] (spatial)
    const int c = cq * 4, n = blockIdx.y;
    const int L = (MODE == 0) ? c : hw;
    const float t = thr[n];
    const int p0 = blockIdx.x * slab_pix, p1 = min(hw, p0 + slab_pix);
    const int first = (MODE == 0) ? 0 : p0, count = (MODE == 0) ? c : (p1 - p0);
    for (int e = threadIdx.x; e < count; e += MB) {
        const int i = first + e;
        float mv = 1.f;
        if (score[(int64_t)n * L + i] > t) mv = soft_noise ? 0.5f * soft_noise[(int64_t)n * L + i] : 0.f;
        mval[e] = mv;
        if (MODE == 1 || blockIdx.x == 0) mask_out[(int64_t)n * L + i] = mv;
    }
    __syncthreads();
    const int64_t base = ((int64_t)n * hw + p0) * cq;
    const int quads = (p1 - p0) * cq;
    for (int e0 = 0; e0 < quads; e0 += APF * MB) {              // 8 loads in flight per thread
        f32x4 v[APF];
#pragma unroll
        for (int u = 0; u < APF; ++u) {
            const int e = e0 + threadIdx.x + u * MB;
            v[u] = e < quads ? code[base + e] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < APF; ++u) {
            const int e = e0 + threadIdx.x + u * MB;
            if (e >= quads) continue;
            f32x4 t = v[u];
            if (MODE == 0) {
                const f32x4 m = reinterpret_cast<const f32x4*>(mval)[e % cq];
                t.x *= m.x; t.y *= m.y; t.z *= m.z; t.w *= m.w;
            } else {
                const float m = mval[e / cq];
                t.x *= m; t.y *= m; t.z *= m; t.w *= m;
            }
            masked[base + e] = t;
        }
    }
}

// ---- ONE launch for the whole generator tail at latent-code sizes: one 1024-thread block per IMAGE, no
// inter-block communication at all.  The block requests its image's grad AND code up front (all loads in flight at once), builds
// the score row in LDS with exactly the summation order of the three-launch path (4 "virtual" 256-thread blocks each reduce one
// 64-pixel split the way score_channel_partial_kernel does, then the splits are added in order like score_channel_finalize_kernel:
// bit-identical scores), ranks (strict '>' of the reference, exact under ties) and stores code * mask from the registers it already holds.
// Why not a multi-block kernel with a grid barrier (built and measured on MI355X, profiles/README.md): the 8 XCDs' L2s are not
// coherent with each other, an agent-scope release is a write-back of the whole L2 per block (~60 us per launch at 128 blocks), and
// even a fence-free barrier (write-through stores + relaxed atomics) pays ~0.15 us per same-address atomic, i.e. ~25 us at 128
// blocks -- two kernel boundaries (~3 us each) are cheaper.  Hence: images up to 64 Ki elements take this kernel, larger problems
// the HBM-streaming three-launch path (72 % of 8 TB/s at 128 MiB), both behind ctl_latent_mask_fused.
#define IB 1024
#define IPF 16      // quads of grad / code a thread holds: 1024 * 16 * 4 = 64 Ki elements per image
template <int MODE, int CPF, int GPF>      // quads of code / of grad a thread holds at once (the host picks the smallest instantiation that fits)
__global__ __launch_bounds__(IB) void latent_mask_image_kernel(const f32x4* __restrict__ grad, const f32x4* __restrict__ code,
                                                                const float* __restrict__ soft_noise, int k_host,
                                                                const int* __restrict__ k_dev, f32x4* __restrict__ masked,
                                                                float* __restrict__ mask_out, float* __restrict__ score_out, int hw,
                                                                int cq, int split_pix, int splits, int S, int slab_pix, int n_img, int xcd_map) {
    // S blocks per image (small batches: more CUs at work): every block builds the WHOLE score row (the image's grad is re-read
    // from L2 by its S blocks) but holds and stores only its own slab of the code
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // Block -> (image, slab): the S blocks of an image sit on ONE XCD (block b runs on XCD b % 8), so that the image's grad -- which every one of
    // them reads whole -- is fetched from HBM once and served from that XCD's L2 to the others (round 3 spread them over S XCDs: FETCH_SIZE
    // counted 2.0x the algorithmic bytes).  XCD x hosts images x, x + 8, ...; the j-th block of an XCD is slab j % S of its (j / S)-th image.
    const int c = cq * 4, tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int img = xcd_map ? (xcd + 8 * (jx / S)) : (int)(blockIdx.x / S), sl = xcd_map ? (jx % S) : (int)(blockIdx.x - img * S);
    if (img >= n_img) return;      // (the grid is rounded up to whole XCD rounds)
    const int L = (MODE == 0) ? c : hw;
    float* srow = sm;                                                   // [L]
    float* mval = sm + L;                                               // [L]
    f32x4* red = reinterpret_cast<f32x4*>(mval + L);                    // [IB]
    float* part = reinterpret_cast<float*>(red + IB);                   // [splits][c] (channel mode)
    const int64_t base = (int64_t)img * hw * cq;
    const int px0 = sl * slab_pix, px1 = min(hw, px0 + slab_pix);
    const int quads = max(px1 - px0, 0) * cq, qoff = px0 * cq;          // this block's slab of the image
    f32x4 cv[CPF];
#pragma unroll
    for (int j = 0; j < CPF; ++j) {                                     // code: consumed last, requested first
        const int e = tid + j * IB;
        cv[j] = e < quads ? code[base + qoff + e] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // k and this thread's first soft-noise value: requested now, consumed after the score row exists (no dependent round trips later)
    int k = k_dev ? k_dev[0] : k_host;
    const int i0 = (MODE == 0) ? 0 : px0, i1 = (MODE == 0) ? c : px1;   // entries this block ranks: every channel / the pixels of its slab
    // ranking: RK lanes share one entry (each scans every RK-th score, conflict-free LDS reads, shuffle sum): a one-lane scan of
    // L = 128 scores is 128 dependent LDS round trips and was most of this kernel's time
    constexpr int RK = 8;
    float noise0 = 0.f;
    if (soft_noise && i0 + tid / RK < i1) noise0 = soft_noise[(int64_t)img * L + i0 + tid / RK];
    if (MODE == 0) {
        // virtual block vb reduces splits vb, vb + 4, ...: thread lt of it sums pixels p0 + prow, p0 + prow + ppb, ... (in that order)
        const int vb = tid >> 8, lt = tid & 255;
        const int q = lt % cq, prow = lt / cq, ppb = 256 / cq;
        const int iters = (split_pix + ppb - 1) / ppb;                  // loads per thread and split
        for (int sp0 = 0; sp0 < splits; sp0 += 4) {
            const int sp = sp0 + vb;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (iters <= GPF) {
                // every load of the split in flight at once, then the adds in the order of the rolled loop (adding the +0 of a lane
                // past the split leaves a sum unchanged bit for bit: the sums start from +0 and can never become -0)
                f32x4 gv[GPF];
#pragma unroll
                for (int j = 0; j < GPF; ++j) {
                    const int pix = sp * split_pix + prow + j * ppb;
                    const bool ok = sp < splits && j < iters && pix < min(hw, (sp + 1) * split_pix);
                    gv[j] = ok ? grad[base + (int64_t)pix * cq + q] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < GPF; ++j) { s.x += gv[j].x; s.y += gv[j].y; s.z += gv[j].z; s.w += gv[j].w; }
            } else if (sp < splits) {
                const int p0 = sp * split_pix, p1 = min(hw, p0 + split_pix);
                for (int pix = p0 + prow; pix < p1; pix += ppb) {
                    const f32x4 v = grad[base + (int64_t)pix * cq + q];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
            red[tid] = s;
            __syncthreads();
            if (lt < cq && sp < splits) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                for (int kk = lt; kk < 256; kk += cq) { const f32x4 v = red[(vb << 8) + kk]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
                reinterpret_cast<f32x4*>(part)[sp * cq + lt] = t;
            }
            __syncthreads();
        }
        const float inv = 1.f / (float)hw;
        for (int ch = tid; ch < c; ch += IB) {
            float a = 0.f;
            for (int s2 = 0; s2 < splits; ++s2) a += part[s2 * c + ch];
            srow[ch] = a * inv;
        }
    } else {
        // a group of cq lanes owns a pixel (cq | 64): the order of score_spatial_kernel
        // (quad tid + j*IB of the image = channel quad tid % cq of pixel tid / cq + j * IB / cq: all of a thread's loads go out at once)
        const int q = tid % cq, prow = tid / cq, ppb = IB / cq;
        const int allq = hw * cq;
        f32x4 gv[GPF];
#pragma unroll
        for (int j = 0; j < GPF; ++j) {
            const int e = tid + j * IB;
            gv[j] = e < allq ? grad[base + e] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < GPF; ++j) {
            const int pix = prow + j * ppb;
            float sc = (gv[j].x + gv[j].y) + (gv[j].z + gv[j].w);
            for (int o = cq >> 1; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
            if (q == 0 && pix < hw) srow[pix] = sc * (1.f / (float)c);
        }
    }
    __syncthreads();
    k = k < 0 ? 0 : (k > L - 1 ? L - 1 : k);
    for (int e0 = 0; e0 < i1 - i0; e0 += IB / RK) {         // (uniform trip count: every lane takes part in the shuffles)
        const int e = e0 + tid / RK, sub = tid % RK;
        const bool ok = i0 + e < i1;
        const int i = ok ? i0 + e : i0;
        const float si = srow[i];
        int ge = 0;
#pragma unroll 8
        for (int j = sub; j < L; j += RK) ge += (srow[j] >= si) ? 1 : 0;
        ge += __shfl_xor(ge, 1); ge += __shfl_xor(ge, 2); ge += __shfl_xor(ge, 4);
        if (ok && sub == 0) {
            const float nz = (e0 == 0) ? noise0 : (soft_noise ? soft_noise[(int64_t)img * L + i] : 0.f);
            const float mv = (ge <= k) ? (soft_noise ? 0.5f * nz : 0.f) : 1.f;
            mval[i] = mv;
            if (MODE == 1 || sl == 0) {
                mask_out[(int64_t)img * L + i] = mv;
                if (score_out) score_out[(int64_t)img * L + i] = si;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CPF; ++j) {
        const int e = tid + j * IB;
        if (e < quads) {
            f32x4 v = cv[j];
            if (MODE == 0) {
                const f32x4 m = reinterpret_cast<const f32x4*>(mval)[e % cq];      // (qoff is a multiple of cq)
                v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            } else {
                const float m = mval[px0 + e / cq];
                v.x *= m; v.y *= m; v.z *= m; v.w *= m;
            }
            masked[base + qoff + e] = v;
        }
    }
}

// ---- dropout2d and uniform noise from a counter hash (splitmix64): stateless, graph-replay safe given a seed buffer
__device__ __forceinline__ float hash_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);   // 24 random bits -> [0,1)
}
// Device-resident RNG state (HIP-graph replays: nothing may be baked into the launch): state[0] = seed, state[1] = step counter
// advanced once per training step by ctl_step_tick; `salt` separates the call sites of one step.
__device__ __forceinline__ uint64_t state_seed(const int64_t* __restrict__ state, uint64_t salt) {
    uint64_t z = (uint64_t)state[0] + ((uint64_t)state[1] + 1) * 0xD1B54A32D192ED03ull + salt * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 32)) * 0xBF58476D1CE4E5B9ull;
    return z ^ (z >> 29);
}
// `m`: storage of z (bit 0) / out (bit 1): set = bf16 (network-internal tensors of BASELINE config 3); the product is formed in fp32 and
// rounded once by the store.  mask_full is fp32 (latent-code form only).
__global__ __launch_bounds__(MB) void dropout2d_kernel(const void* __restrict__ z, const float* __restrict__ keep,
                                                        uint64_t seed, const int64_t* __restrict__ state, float p,
                                                        void* __restrict__ out, float* __restrict__ keep_out,
                                                        f32x4* __restrict__ mask_full, int hw, int cq, int slab_pix, unsigned m) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [c] multipliers
    const int c = cq * 4, n = blockIdx.y;
    const float inv = 1.f / (1.f - p);
    if (state) seed = state_seed(state, seed);
    for (int ch = threadIdx.x; ch < c; ch += MB) {
        float kp = keep ? keep[(int64_t)n * c + ch] : (hash_uniform(seed, (uint64_t)n * c + ch) >= p ? 1.f : 0.f);
        if (keep_out && blockIdx.x == 0) keep_out[(int64_t)n * c + ch] = kp;
        sm[ch] = kp * inv;
    }
    __syncthreads();
    const int p0 = blockIdx.x * slab_pix, p1 = min(hw, p0 + slab_pix);
    const int64_t base = ((int64_t)n * hw + p0) * cq;
    const int quads = (p1 - p0) * cq;
    for (int e = threadIdx.x; e < quads; e += MB) {
        f32x4 v = ldq(z, base + e, m & 1);
        const f32x4 k4 = reinterpret_cast<const f32x4*>(sm)[e % cq];
        const f32x4 in = v;
        v.x *= k4.x; v.y *= k4.y; v.z *= k4.z; v.w *= k4.w;
        stq(out, base + e, v, m & 2);
        // upstream's `mask` (model.py:334-336): 1 where the dropped-out tensor EQUALS the input, else 0
        if (mask_full) mask_full[base + e] = f32x4{v.x == in.x ? 1.f : 0.f, v.y == in.y ? 1.f : 0.f, v.z == in.z ? 1.f : 0.f, v.w == in.w ? 1.f : 0.f};
    }
}
__global__ __launch_bounds__(MB) void uniform_kernel(float* __restrict__ out, int64_t count, uint64_t seed,
                                                      const int64_t* __restrict__ state) {
    if (state) seed = state_seed(state, seed);
    const int64_t stride = (int64_t)gridDim.x * MB;
    for (int64_t i = (int64_t)blockIdx.x * MB + threadIdx.x; i < count; i += stride) out[i] = hash_uniform(seed, (uint64_t)i);
}

// ------------------------------------------------------------------------------------------------ launchers
static bool cq_ok(int c) { return c >= 4 && c % 4 == 0 && (c / 4) <= 64 && (64 % (c / 4)) == 0; }
static int slab_pixels(int n, int hw, int c) {
    // ~32 KiB of code per block (8 quads per thread in flight), but at least ~512 blocks when the problem is large enough
    static const int forced_kb = ctl_tune_int("CTL_MASK_SLAB_KB", 0);   // tuning hook
    int sp = ((forced_kb > 0 ? forced_kb : 32) * 1024) / (c * 4);
    if (sp < 1) sp = 1;
    while (sp > 1 && (int64_t)n * ctl_cdiv(hw, sp) < 512) sp >>= 1;
    return sp;
}

extern "C" size_t ctl_latent_score_ws_floats(int32_t mode, int32_t n, int32_t hw, int32_t c) {
    return mode == 0 ? (size_t)n * ctl_cdiv(hw, score_split_pix(n, hw)) * c : 0;
}

extern "C" int ctl_latent_score(int32_t mode, const float* grad, float* score, float* scratch, int32_t n, int32_t hw,
                                int32_t c, ctl_stream stream) {
    CTL_REQUIRE(grad && score && n > 0 && hw > 0 && cq_ok(c), "latent_score: bad arguments (c=%d)", c);
    hipStream_t s = (hipStream_t)stream;
    const int cq = c / 4;
    if (mode == 0) {
        CTL_REQUIRE(scratch, "latent_score: channel mode needs scratch");
        const int sp_pix = score_split_pix(n, hw);
        const int splits = ctl_cdiv(hw, sp_pix);
        score_channel_partial_kernel<<<dim3(splits, n), dim3(MB), 0, s>>>((const f32x4*)grad, scratch, hw, cq, splits, sp_pix);
        score_channel_finalize_kernel<<<dim3(ctl_cdiv(c, 64), n), dim3(64), 0, s>>>(scratch, score, c, splits, 1.f / (float)hw);
        ctl_count_launches(1);
    } else if (mode == 1) {
        const int64_t pixels = (int64_t)n * hw;
        const int ppb = MB / cq;
        int64_t blocks = ctl_cdiv64(pixels, ppb);
        if (blocks > 2048) blocks = 2048;
        score_spatial_kernel<<<dim3((unsigned)blocks), dim3(MB), 0, s>>>((const f32x4*)grad, score, pixels, cq, 1.f / (float)c);
    } else {
        CTL_FAIL(CTL_EINVAL, "latent_score: mode %d", mode);
    }
    CTL_LAUNCH_CHECK("latent_score");
    return CTL_OK;
}

extern "C" size_t ctl_latent_mask_apply_ws_floats(int32_t mode, int32_t n, int32_t hw, int32_t c) {
    return ((mode == 0 ? c : hw) > 1024) ? (size_t)n : 0;
}

extern "C" int ctl_latent_mask_apply(int32_t mode, const float* code, const float* score, const float* soft_noise,
                                     int32_t k_host, const int32_t* k_dev, float* masked, float* mask_out, float* scratch,
                                     int32_t n, int32_t hw, int32_t c, ctl_stream stream) {
    CTL_REQUIRE(code && score && masked && mask_out && n > 0 && hw > 0 && cq_ok(c), "latent_mask_apply: bad arguments");
    const int L = mode == 0 ? c : hw;
    CTL_REQUIRE(L <= 8192, "latent_mask_apply: row length %d > 8192", L);
    CTL_REQUIRE(k_dev || (k_host >= 0 && k_host < L), "latent_mask_apply: k=%d out of range [0,%d)", k_host, L);
    hipStream_t s = (hipStream_t)stream;
    const int sp = slab_pixels(n, hw, c);
    const dim3 grid(ctl_cdiv(hw, sp), n);
    if (L > 1024) {
        // long rows: per-image threshold by bitonic sort (caller-provided scratch holds the n thresholds), then a streaming apply
        CTL_REQUIRE(mode == 1, "latent_mask_apply: channel rows longer than 1024 are not supported");
        CTL_REQUIRE(scratch, "latent_mask_apply: rows longer than 1024 need ctl_latent_mask_apply_ws_floats() floats of scratch");
        int lp2 = 1;
        while (lp2 < L) lp2 <<= 1;
        latent_threshold_kernel<<<dim3(n), dim3(MB), (size_t)lp2 * sizeof(float), s>>>(score, L, lp2, k_host, k_dev, scratch);
        mask_apply_thr_kernel<1><<<grid, dim3(MB), (size_t)sp * sizeof(float), s>>>((const f32x4*)code, score, soft_noise, scratch,
                                                                                   (f32x4*)masked, mask_out, hw, c / 4, sp);
        ctl_count_launches(1);      // threshold + apply
        CTL_LAUNCH_CHECK("latent_mask_apply(thr)");
        return CTL_OK;
    }
    if (mode == 0) {
        const size_t lds = (size_t)(L + c) * sizeof(float);
        mask_apply_kernel<0><<<grid, dim3(MB), lds, s>>>((const f32x4*)code, score, nullptr, 0, 0.f, nullptr, soft_noise, k_host, k_dev,
                                                        (f32x4*)masked, mask_out, hw, c / 4, sp);
    } else if (mode == 1) {
        const size_t lds = (size_t)(L + sp) * sizeof(float);
        mask_apply_kernel<1><<<grid, dim3(MB), lds, s>>>((const f32x4*)code, score, nullptr, 0, 0.f, nullptr, soft_noise, k_host, k_dev,
                                                        (f32x4*)masked, mask_out, hw, c / 4, sp);
    } else {
        CTL_FAIL(CTL_EINVAL, "latent_mask_apply: mode %d", mode);
    }
    CTL_LAUNCH_CHECK("latent_mask_apply");
    return CTL_OK;
}


// ---- fused generator tail, host side
static bool image_kernel_ok(int mode, int hw, int c) {
    const int L = mode == 0 ? c : hw, cq = c / 4;
    return (int64_t)hw * c <= (int64_t)IB * IPF * 4 && L <= 1024 && 256 % cq == 0;
}
extern "C" size_t ctl_latent_mask_fused_ws_floats(int32_t mode, int32_t n, int32_t hw, int32_t c) {
    if (image_kernel_ok(mode, hw, c)) return 0;
    const size_t L = mode == 0 ? c : hw;          // multi-launch path: [score n*L][ctl_latent_score scratch][ctl_latent_mask_apply scratch]
    return (size_t)n * L + ctl_latent_score_ws_floats(mode, n, hw, c) + ctl_latent_mask_apply_ws_floats(mode, n, hw, c);
}
extern "C" int ctl_latent_mask_fused(int32_t mode, const float* grad, const float* code, const float* soft_noise, int32_t k_host,
                                     const int32_t* k_dev, float* masked, float* mask_out, float* score_out, float* workspace,
                                     int32_t n, int32_t hw, int32_t c, ctl_stream stream) {
    CTL_REQUIRE(grad && code && masked && mask_out && n > 0 && hw > 0 && cq_ok(c), "latent_mask_fused: bad arguments (c=%d)", c);
    CTL_REQUIRE(mode == 0 || mode == 1, "latent_mask_fused: mode %d", mode);
    const int L = mode == 0 ? c : hw;
    CTL_REQUIRE(L <= 8192, "latent_mask_fused: row length %d > 8192", L);
    CTL_REQUIRE(k_dev || (k_host >= 0 && k_host < L), "latent_mask_fused: k=%d out of range [0,%d)", k_host, L);
    hipStream_t s = (hipStream_t)stream;
    if (image_kernel_ok(mode, hw, c)) {
        const int split_pix = score_split_pix(n, hw), splits = ctl_cdiv(hw, split_pix);
        const size_t lds = ((size_t)2 * L + (mode == 0 ? (size_t)splits * c : 0)) * sizeof(float) + (size_t)IB * sizeof(f32x4);
        int S = 1;                                    // blocks per image: at least ~64 blocks in flight when the batch is small
        while (S < 8 && n * S < 64 && hw / (2 * S) >= 8) S *= 2;
        static const int forced_s = ctl_tune_int("CTL_MASK_S", 0);      // tuning hook
        if (forced_s > 0) S = forced_s;
        const int slab_pix = ctl_cdiv(hw, S);
        // registers: the slab's code quads (CPF) and the grad quads in flight (GPF: channel mode = loads per thread and split,
        // spatial mode = the whole image spread over the block); the rolled loop of the channel mode takes whatever exceeds GPF
        const int cpf = ctl_cdiv(slab_pix * (c / 4), IB);
        const int gpf = mode == 0 ? ctl_cdiv(split_pix, 256 / (c / 4)) : ctl_cdiv(hw * (c / 4), IB);
        const bool small = cpf <= 4 && gpf <= 8;
        const int xcd_map = (S > 1 && n * S >= 8) ? 1 : 0;                                   // (see the kernel: an image's S blocks on one XCD)
        const int grid_x = xcd_map ? 8 * ctl_cdiv(n, 8) * S : n * S;
#define CTL_IMG_LAUNCH(M, CP, GP)                                                                                                  \
        latent_mask_image_kernel<M, CP, GP><<<dim3(grid_x), dim3(IB), lds, s>>>((const f32x4*)grad, (const f32x4*)code, soft_noise, k_host, \
                                                                                k_dev, (f32x4*)masked, mask_out, score_out, hw, c / 4,  \
                                                                                split_pix, splits, S, slab_pix, n, xcd_map)
        if (mode == 0) { if (small) CTL_IMG_LAUNCH(0, 4, 8); else CTL_IMG_LAUNCH(0, IPF, 8); }      // (more than 8 loads per split: rolled loop)
        else { if (small) CTL_IMG_LAUNCH(1, 4, 8); else CTL_IMG_LAUNCH(1, IPF, IPF); }
#undef CTL_IMG_LAUNCH
        CTL_LAUNCH_CHECK("latent_mask_fused");
        return CTL_OK;
    }
    // HBM-streaming sizes.  Channel mode: split sums of grad, then ONE apply launch whose blocks finalise the score row from the
    // split sums themselves (2 launches).  Spatial mode: score pass, (long rows) threshold, apply.
    CTL_REQUIRE(workspace, "latent_mask_fused: this size needs ctl_latent_mask_fused_ws_floats() floats of workspace");
    float* score = workspace;
    float* ws_score = score + (size_t)n * L;
    float* ws_apply = ws_score + ctl_latent_score_ws_floats(mode, n, hw, c);
    if (mode == 0 && L <= 1024) {
        const int cq = c / 4, sp_pix = score_split_pix(n, hw), splits = ctl_cdiv(hw, sp_pix);
        score_channel_partial_kernel<<<dim3(splits, n), dim3(MB), 0, s>>>((const f32x4*)grad, ws_score, hw, cq, splits, sp_pix);
        const int sp = slab_pixels(n, hw, c);
        mask_apply_kernel<0><<<dim3(ctl_cdiv(hw, sp), n), dim3(MB), (size_t)(L + c) * sizeof(float), s>>>(
            (const f32x4*)code, nullptr, ws_score, splits, 1.f / (float)hw, score_out, soft_noise, k_host, k_dev, (f32x4*)masked, mask_out,
            hw, cq, sp);
        ctl_count_launches(1);      // split sums + apply
        CTL_LAUNCH_CHECK("latent_mask_fused(stream)");
        return CTL_OK;
    }
    int rc = ctl_latent_score(mode, grad, score, ws_score, n, hw, c, stream);
    if (rc != CTL_OK) return rc;
    rc = ctl_latent_mask_apply(mode, code, score, soft_noise, k_host, k_dev, masked, mask_out, ws_apply, n, hw, c, stream);
    if (rc != CTL_OK) return rc;
    if (score_out) {
        hipError_t e = hipMemcpyAsync(score_out, score, (size_t)n * L * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) CTL_FAIL(CTL_ELAUNCH, "latent_mask_fused: %s", hipGetErrorString(e));
    }
    return CTL_OK;
}

static int dropout2d_launch(const void* z, const float* keep, uint64_t seed, const int64_t* state, float p, void* out,
                            float* keep_out, float* mask_full, int32_t n, int32_t hw, int32_t c, uint32_t bf16_mask, ctl_stream stream) {
    CTL_REQUIRE(z && out && n > 0 && hw > 0 && cq_ok(c) && p >= 0.f && p < 1.f, "dropout2d: bad arguments");
    CTL_REQUIRE(!mask_full || !(bf16_mask & 3), "dropout2d: the full-size mask goes with fp32 tensors");
    const int sp = slab_pixels(n, hw, c);
    dropout2d_kernel<<<dim3(ctl_cdiv(hw, sp), n), dim3(MB), (size_t)c * sizeof(float), (hipStream_t)stream>>>(
        z, keep, seed, state, p, out, keep_out, (f32x4*)mask_full, hw, c / 4, sp, bf16_mask);
    CTL_LAUNCH_CHECK("dropout2d");
    return CTL_OK;
}
extern "C" int ctl_dropout2d(const float* z, const float* keep, uint64_t seed, float p, float* out, float* keep_out,
                             int32_t n, int32_t hw, int32_t c, ctl_stream stream) {
    return dropout2d_launch(z, keep, seed, nullptr, p, out, keep_out, nullptr, n, hw, c, 0, stream);
}
extern "C" int ctl_dropout2d_ex(const float* z, const float* keep, uint64_t seed_or_salt, const int64_t* state, float p,
                                float* out, float* keep_out, float* mask_full, int32_t n, int32_t hw, int32_t c,
                                ctl_stream stream) {
    return dropout2d_launch(z, keep, seed_or_salt, state, p, out, keep_out, mask_full, n, hw, c, 0, stream);
}
extern "C" int ctl_dropout2d_dt(const void* z, const float* keep, uint64_t seed_or_salt, const int64_t* state, float p, void* out,
                                float* keep_out, int32_t n, int32_t hw, int32_t c, uint32_t bf16_mask, ctl_stream stream) {
    return dropout2d_launch(z, keep, seed_or_salt, state, p, out, keep_out, nullptr, n, hw, c, bf16_mask, stream);
}

static int uniform_launch(float* out, int64_t count, uint64_t seed, const int64_t* state, ctl_stream stream) {
    CTL_REQUIRE(out && count > 0, "uniform: bad arguments");
    int64_t blocks = ctl_cdiv64(count, MB);
    if (blocks > 2048) blocks = 2048;
    uniform_kernel<<<dim3((unsigned)blocks), dim3(MB), 0, (hipStream_t)stream>>>(out, count, seed, state);
    CTL_LAUNCH_CHECK("uniform");
    return CTL_OK;
}
extern "C" int ctl_uniform(float* out, int64_t count, uint64_t seed, ctl_stream stream) {
    return uniform_launch(out, count, seed, nullptr, stream);
}
extern "C" int ctl_uniform_dev(float* out, int64_t count, uint64_t salt, const int64_t* state, ctl_stream stream) {
    CTL_REQUIRE(state, "uniform_dev: null state");
    return uniform_launch(out, count, salt, state, stream);
}

// One launch at the head of a (graph-replayed) training step: advances the device-resident step state
//   state[0] RNG seed (constant)   state[1] RNG step counter   state[2] Adam step count
__global__ void step_tick_kernel(int64_t* state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { state[1] += 1; state[2] += 1; }
}
// A kernel that does nothing for `microseconds` (one wave; 100 MHz constant clock): the probe with which the host checks that two
// streams really run side by side.  HIP streams share a handful of hardware queues, and two streams on ONE queue execute in order.
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int ctl_spin(int32_t microseconds, ctl_stream stream) {
    CTL_REQUIRE(microseconds >= 0 && microseconds <= 100000, "spin: 0..100000 us");
    spin_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>((unsigned long long)microseconds * 100ull);
    CTL_LAUNCH_CHECK("spin");
    return CTL_OK;
}
extern "C" int ctl_step_tick(int64_t* state, ctl_stream stream) {
    CTL_REQUIRE(state, "step_tick: null state");
    step_tick_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(state);
    CTL_LAUNCH_CHECK("step_tick");
    return CTL_OK;
}
