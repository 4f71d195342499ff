// HBM-bound kernels of the hot path: BatchNorm finalize/apply/backward, residual/activation backward, nearest-upsample
// backward, STN input builders, fused losses, argmax and the flat multi-tensor Adam.  All fp32 NHWC; 16 bytes per lane
// wherever the channel count allows, wavefront(64)-shuffle + LDS block reductions, two-stage deterministic sums
// (per-block partials in fp32, finalisation in fp64) instead of float atomics.
#include "ctl_common.h"

#define EB 256               // threads per block for the streaming kernels
#ifndef CTL_FIN_THREADS
#define CTL_FIN_THREADS EB      // threads per block of the BatchNorm finalize kernels (one block per channel)
#endif
#define MAX_STREAM_BLOCKS 2048

static inline unsigned stream_blocks(int64_t work_items) {
    int64_t b = ctl_cdiv64(work_items, EB);
    if (b > MAX_STREAM_BLOCKS) b = MAX_STREAM_BLOCKS;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// sums two values over the block at once: thread 0 gets both (one LDS round instead of two)
__device__ __forceinline__ void block_sum_double2(double& a, double& b, double* sm) {
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = a; sm[8 + w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = b = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += sm[i]; b += sm[8 + i]; }
    }
    __syncthreads();
}
// the rows a thread owns (b = tid, tid + 256, ...; at most 4 * 256 rows) requested back to back: ONE memory round trip instead of one per
// 256 rows -- these kernels are pure latency (a few KB of partial sums), and there are ~220 of them in a training step
__device__ __forceinline__ void sum_rows2(const float* __restrict__ partial, int64_t row0, int blocks, int c, int ch, double& s1, double& s2) {
    const int nthr = (int)blockDim.x;          // (the finalize kernels may run narrower than EB, see CTL_FIN_THREADS)
    for (int b0 = threadIdx.x; b0 < blocks; b0 += 4 * nthr) {
        float v1[4], v2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + u * nthr;
            const bool ok = b < blocks;
            v1[u] = ok ? partial[((row0 + b) * 2 + 0) * c + ch] : 0.f;
            v2[u] = ok ? partial[((row0 + b) * 2 + 1) * c + ch] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
    }
}
__device__ __forceinline__ double block_sum_double(double v, double* sm) {
    // 256 threads: wave shuffle then LDS
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sm[i];
    }
    return r;  // valid on thread 0
}

// (ldq / stq, the storage-type aware quad accessors, live in ctl_common.h)

// ------------------------------------------------------------------------------------------------ BatchNorm forward
// Phase timers of the forward finalize (variant build -DCTL_TIMING_FIN; tools/debug/fin_timing.py reads them with ctl_debug_timing_fin): s_memtime
// cycles of thread 0 of block 0 summed over the launches: [0] until the rows have arrived, [1] block sum, [2] coefficient arithmetic + stores
// acknowledged, [3] launches, [4] sum of rows, [5] sum of groups; [6] realtime (100 MHz) from the kernel's first instruction to its last
#ifdef CTL_TIMING_FIN
__device__ unsigned long long ctl_tmf[8];
#define FT_DECL unsigned long long ft_prev = __builtin_amdgcn_s_memtime(), ft_acc[3] = {0, 0, 0}; const unsigned long long ft_r0 = __builtin_amdgcn_s_memrealtime();
#define FT(i) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long ft_now = __builtin_amdgcn_s_memtime(); ft_acc[i] += ft_now - ft_prev; ft_prev = ft_now; }
#define FT_FLUSH if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 3; ++i_) atomicAdd(&ctl_tmf[i_], ft_acc[i_]); atomicAdd(&ctl_tmf[3], 1ull); \
                     atomicAdd(&ctl_tmf[4], (unsigned long long)blocks); atomicAdd(&ctl_tmf[5], (unsigned long long)groups); \
                     atomicAdd(&ctl_tmf[6], __builtin_amdgcn_s_memrealtime() - ft_r0); }
extern "C" int ctl_debug_timing_fin(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(ctl_tmf), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(ctl_tmf), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#else
#define FT_DECL
#define FT(i)
#define FT_FLUSH
#endif
__global__ __launch_bounds__(EB) void bn_finalize_kernel(const float* __restrict__ partial, int blocks, int c,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum,
                                                          int update_running, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var,
                                                          int64_t* __restrict__ nbt, float* __restrict__ scale,
                                                          float* __restrict__ shift, float* __restrict__ save_mean,
                                                          float* __restrict__ save_invstd, int groups, float* __restrict__ save_uvar) {
    __shared__ double sm[16];
    const int ch = blockIdx.x;
    FT_DECL
    ctl_bn_chan p = {};
    if (threadIdx.x == 0) p = ctl_bn_chan_load(ch, gamma, beta, update_running, running_mean, running_var);
    // groups = independent passes batched along n: one set of coefficients each; the running statistics see them in order,
    // exactly as consecutive forward calls would
    for (int g = 0; g < groups; ++g) {
        double s1 = 0.0, s2 = 0.0;
        sum_rows2(partial, (int64_t)g * blocks, blocks, c, ch, s1, s2);
        FT(0)
        block_sum_double2(s1, s2, sm);
        FT(1)
        if (threadIdx.x == 0) {
            ctl_bn_coefs(s1, s2, count, c, g, ch, p, eps, momentum, update_running, running_mean, running_var, scale, shift, save_mean,
                         save_invstd, save_uvar);
            if (update_running && ch == 0 && nbt) nbt[0] += 1;
        }
        FT(2)
    }
    FT_FLUSH
}

__global__ void bn_eval_kernel(int c, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale, float* shift, int groups) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < c) {
        const float sc = gamma[ch] / sqrtf(rv[ch] + eps);
        const float sh = beta[ch] - rm[ch] * sc;
        for (int g = 0; g < groups; ++g) { scale[g * c + ch] = sc; shift[g * c + ch] = sh; }
    }
}

__global__ __launch_bounds__(EB) void bn_act_kernel(const void* __restrict__ x, const f32x4* __restrict__ scale,
                                                     const f32x4* __restrict__ shift, float slope,
                                                     void* __restrict__ y, int64_t quads, int cq, int64_t group_quads, unsigned m) {
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < quads; i += stride) {
        const int q = (int)(i % cq) + (int)(i / group_quads) * cq;
        const f32x4 sc = scale[q], sh = shift[q];
        f32x4 v = ldq(x, i, m & 1);
        v.x = ctl_leaky(v.x * sc.x + sh.x, slope); v.y = ctl_leaky(v.y * sc.y + sh.y, slope);
        v.z = ctl_leaky(v.z * sc.z + sh.z, slope); v.w = ctl_leaky(v.w * sc.w + sh.w, slope);
        stq(y, i, v, m & 2);
    }
}

// ------------------------------------------------------------------------------------------------ backward reductions
// rows of partial sums per group: enough blocks to stream a large tensor, few for the low-resolution layers (whose old fixed 512
// rows cost more to write and re-read than the tensor itself)
static inline int red_rows_for(int64_t quads_per_group) {
    int64_t b = ctl_cdiv64(quads_per_group, (int64_t)EB * 8);
    if (b > CTL_RED_BLOCKS) b = CTL_RED_BLOCKS;
    if (b < 16) b = 16;
    return (int)b;
}

// BatchNorm backward coefficients of one (group, channel) from its two sums
struct ctl_bnb_chan { float gamma, dgamma, dbeta; };      // requested before the reduction (dgamma / dbeta carried over the groups)
__device__ __forceinline__ ctl_bnb_chan bnb_chan_load(int ch, const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                      const float* __restrict__ dbeta, int accumulate) {
    ctl_bnb_chan p;
    p.gamma = gamma[ch];
    p.dgamma = (dgamma && accumulate) ? dgamma[ch] : 0.f;
    p.dbeta = (dbeta && accumulate) ? dbeta[ch] : 0.f;
    return p;
}
__device__ __forceinline__ void bn_bwd_coefs(double s1, double s2, double count, int c, int gi, int ch, ctl_bnb_chan& p, float mean,
                                             float invstd, float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta, bool affine = true) {
    const double mu = mean, is = invstd, g = p.gamma;
    const double sum_g = s1;
    const double sum_gxhat = is * (s2 - mu * s1);
    const double m1 = sum_g / count, m2 = sum_gxhat / count;
    // dx = gamma*is*(g - m1 - xhat*m2),  xhat = (x-mu)*is   ==>  dx = A*g + B*x + C
    const double A = g * is;
    const double B = -g * is * is * m2;
    const double C = -g * is * m1 + g * is * is * m2 * mu;
    coef[(gi * 3 + 0) * c + ch] = (float)A;
    coef[(gi * 3 + 1) * c + ch] = (float)B;
    coef[(gi * 3 + 2) * c + ch] = (float)C;
    // (first group without `accumulate`: p.dgamma = p.dbeta = 0 and 0 + x == x exactly)
    // affine = false: a group whose pass ran with frozen gamma / beta (BatchNorm mode B, model_util.py:414-451) adds nothing to their gradients
    if (affine) {
        p.dgamma += (float)sum_gxhat;
        p.dbeta += (float)sum_g;
    }
    if (dgamma) dgamma[ch] = p.dgamma;
    if (dbeta) dbeta[ch] = p.dbeta;
}

// grid = (rows, groups) x 256; the global stride (rows * 256) is a multiple of every C/4 in use, so a thread always sees
// the same channel quad and accumulates it in registers.
template <int MODE>
__global__ __launch_bounds__(EB) void bwd_reduce_kernel(const void* __restrict__ dy_, const void* __restrict__ act_src_,
                                                         const void* __restrict__ bn_src_,
                                                         const f32x4* __restrict__ scale,
                                                         const f32x4* __restrict__ shift, float slope, int64_t quads,
                                                         int cq, float* __restrict__ partial, unsigned m,      // m: bit 0 dy, 1 act_src, 2 bn_src, 3 ds
                                                         void* __restrict__ ds_) {
    // blockIdx.y = BatchNorm group: `quads` is the size of one group, its data start at blockIdx.y * quads
    __shared__ f32x4 sm[2][EB];
    const int64_t gtid = (int64_t)blockIdx.x * EB + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * EB;
    const int q = (int)(gtid % cq);
    const int64_t gbase = (int64_t)blockIdx.y * quads;
    f32x4 sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
    if (MODE == 1) { sc = scale[blockIdx.y * cq + q]; sh = shift[blockIdx.y * cq + q]; }
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    for (int64_t i = gtid; i < quads; i += stride) {
        f32x4 g = ldq(dy_, gbase + i, m & 1);
        if (MODE == 0) {
            const f32x4 o = ldq(act_src_, gbase + i, m & 2);
            g.x *= ctl_leaky_grad(o.x, slope); g.y *= ctl_leaky_grad(o.y, slope);
            g.z *= ctl_leaky_grad(o.z, slope); g.w *= ctl_leaky_grad(o.w, slope);
            // the activation-gradient product does not depend on the sums: written here, the apply pass reads it back instead of
            // recomputing it from dout and out (one tensor read less per residual tail)
            if (ds_) stq(ds_, gbase + i, g, m & 8);
        }
        if (MODE == 2) {
            s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        } else {
            const f32x4 u = ldq(bn_src_, gbase + i, m & 4);
            if (MODE == 1) {
                g.x *= ctl_leaky_grad(u.x * sc.x + sh.x, slope); g.y *= ctl_leaky_grad(u.y * sc.y + sh.y, slope);
                g.z *= ctl_leaky_grad(u.z * sc.z + sh.z, slope); g.w *= ctl_leaky_grad(u.w * sc.w + sh.w, slope);
            }
            s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
            s2.x += g.x * u.x; s2.y += g.y * u.y; s2.z += g.z * u.z; s2.w += g.w * u.w;
        }
    }
    sm[0][threadIdx.x] = s1;
    sm[1][threadIdx.x] = s2;
    __syncthreads();
    const int c = cq * 4;
    for (int t = threadIdx.x; t < 2 * c; t += EB) {   // one (stat, channel) per thread
        const int stat = t / c, ch = t % c;
        const int qq = ch >> 2, comp = ch & 3;
        float v = 0.f;
        // threads with (tid % cq) == qq hold this quad (EB % cq == 0 for every cq in use)
        for (int k = qq; k < EB; k += cq) v += sm[stat][k][comp];
        partial[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + stat) * c + ch] = v;
    }
}

__global__ __launch_bounds__(EB) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int blocks, int c,
                                                              double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ save_mean,
                                                              const float* __restrict__ save_invstd,
                                                              float* __restrict__ coef, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate, int groups, unsigned affine_groups) {
    __shared__ double sm[16];
    const int ch = blockIdx.x;
    ctl_bnb_chan p = {};
    if (threadIdx.x == 0) p = bnb_chan_load(ch, gamma, dgamma, dbeta, accumulate);
    for (int gi = 0; gi < groups; ++gi) {       // coef[group][3][c]; dgamma/dbeta sum the groups in order
        float mu = 0.f, is = 0.f;
        if (threadIdx.x == 0) { mu = save_mean[gi * c + ch]; is = save_invstd[gi * c + ch]; }
        double s1 = 0.0, s2 = 0.0;
        sum_rows2(partial, (int64_t)gi * blocks, blocks, c, ch, s1, s2);
        block_sum_double2(s1, s2, sm);
        if (threadIdx.x == 0) bn_bwd_coefs(s1, s2, count, c, gi, ch, p, mu, is, coef, dgamma, dbeta, (affine_groups >> gi) & 1u);
    }
}

template <int MODE>
__global__ __launch_bounds__(EB) void bwd_apply_kernel(const void* __restrict__ dy, const void* __restrict__ act_src,
                                                        const void* __restrict__ bn_src,
                                                        const f32x4* __restrict__ scale,
                                                        const f32x4* __restrict__ shift, float slope,
                                                        const f32x4* __restrict__ coef, int64_t quads, int cq,
                                                        void* __restrict__ ds, void* __restrict__ dx, int64_t group_quads,
                                                        unsigned m) {      // m: bit 0 dy, 1 act_src, 2 bn_src, 3 ds, 4 dx
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < quads; i += stride) {
        const int q = (int)(i % cq);
        const int gi = (int)(i / group_quads);
        f32x4 g = ldq(dy, i, m & 1);
        const f32x4 u = ldq(bn_src, i, m & 4);
        if (MODE == 0) {
            const f32x4 o = ldq(act_src, i, m & 2);
            g.x *= ctl_leaky_grad(o.x, slope); g.y *= ctl_leaky_grad(o.y, slope);
            g.z *= ctl_leaky_grad(o.z, slope); g.w *= ctl_leaky_grad(o.w, slope);
            if (ds) stq(ds, i, g, m & 8);
        } else if (MODE == 1) {
            const f32x4 sc = scale[gi * cq + q], sh = shift[gi * cq + q];
            g.x *= ctl_leaky_grad(u.x * sc.x + sh.x, slope); g.y *= ctl_leaky_grad(u.y * sc.y + sh.y, slope);
            g.z *= ctl_leaky_grad(u.z * sc.z + sh.z, slope); g.w *= ctl_leaky_grad(u.w * sc.w + sh.w, slope);
        }
        const f32x4 A = coef[(gi * 3 + 0) * cq + q], B = coef[(gi * 3 + 1) * cq + q], C = coef[(gi * 3 + 2) * cq + q];
        f32x4 r;
        r.x = A.x * g.x + B.x * u.x + C.x; r.y = A.y * g.y + B.y * u.y + C.y;
        r.z = A.z * g.z + B.z * u.z + C.z; r.w = A.w * g.w + B.w * u.w + C.w;
        stq(dx, i, r, m & 16);
    }
}

// ---- all-bf16 forms of the two BatchNorm-backward passes: 8 channels = 16 bytes per lane (the 4-channel forms above move 8 bytes per
// lane when the tensors are bf16 -- half the bytes per memory instruction, 0.54-0.70x the 16-byte rate on this part), two octets in
// flight per thread and tensor.  Arithmetic, rounding points and the partial-row layout are those of the quad forms.
typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
struct f32x8 { float v[8]; };
__device__ __forceinline__ f32x8 unpack8(u32x4e u) {
    f32x8 r;
    r.v[0] = __builtin_bit_cast(float, u.x << 16); r.v[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    r.v[2] = __builtin_bit_cast(float, u.y << 16); r.v[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
    r.v[4] = __builtin_bit_cast(float, u.z << 16); r.v[5] = __builtin_bit_cast(float, u.z & 0xffff0000u);
    r.v[6] = __builtin_bit_cast(float, u.w << 16); r.v[7] = __builtin_bit_cast(float, u.w & 0xffff0000u);
    return r;
}
__device__ __forceinline__ u32x4e pack8(const f32x8& a) {
    u32x4e r;
    r.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2e{a.v[0], a.v[1]}, bf16x2e));
    r.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2e{a.v[2], a.v[3]}, bf16x2e));
    r.z = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2e{a.v[4], a.v[5]}, bf16x2e));
    r.w = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2e{a.v[6], a.v[7]}, bf16x2e));
    return r;
}
__device__ __forceinline__ f32x8 load8f(const float* __restrict__ p) {      // 8 consecutive fp32 coefficients (32-byte aligned)
    const f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
    return f32x8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}

template <int MODE>
__global__ __launch_bounds__(EB) void bwd_reduce16_kernel(const u32x4e* __restrict__ dy, const u32x4e* __restrict__ act_src,
                                                           const u32x4e* __restrict__ bn_src, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float slope, int64_t octs, int co,
                                                           float* __restrict__ partial, u32x4e* __restrict__ ds) {      // ds (mode 0): g = dy * leaky'(act_src), stored
    // blockIdx.y = BatchNorm group: `octs` is the size of one group; a thread always sees the same channel octet (256 % co == 0)
    __shared__ float sm[2][8][EB];
    const int64_t gtid = (int64_t)blockIdx.x * EB + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * EB;
    const int o = (int)(gtid % co), c = co * 8;
    const int64_t gbase = (int64_t)blockIdx.y * octs;
    f32x8 sc, sh;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc.v[k] = 1.f; sh.v[k] = 0.f; }
    if (MODE == 1) { sc = load8f(scale + blockIdx.y * c + o * 8); sh = load8f(shift + blockIdx.y * c + o * 8); }
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
    for (int64_t i0 = gtid; i0 < octs; i0 += 2 * stride) {
        u32x4e gq[2], uq[2], oq[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = i0 + u * stride;
            const bool ok = i < octs;
            const u32x4e z = {0u, 0u, 0u, 0u};
            gq[u] = ok ? dy[gbase + i] : z;
            uq[u] = ok ? bn_src[gbase + i] : z;
            if (MODE == 0) oq[u] = ok ? act_src[gbase + i] : z;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x8 g = unpack8(gq[u]);
            const f32x8 x = unpack8(uq[u]);
            if (MODE == 0) {
                const f32x8 a = unpack8(oq[u]);
#pragma unroll
                for (int k = 0; k < 8; ++k) g.v[k] *= ctl_leaky_grad(a.v[k], slope);
                if (ds && i0 + u * stride < octs) ds[gbase + i0 + u * stride] = pack8(g);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) g.v[k] *= ctl_leaky_grad(x.v[k] * sc.v[k] + sh.v[k], slope);
            }
            // (a lane past the end contributes g = 0 * leaky'(..) = 0 and 0 * u = 0: nothing)
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1[k] += g.v[k]; s2[k] += g.v[k] * x.v[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { sm[0][k][threadIdx.x] = s1[k]; sm[1][k][threadIdx.x] = s2[k]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * c; t += EB) {   // one (stat, channel) per thread: threads with (tid % co) == channel / 8 hold it
        const int stat = t / c, ch = t % c;
        const int oo = ch >> 3, comp = ch & 7;
        float v = 0.f;
        for (int k = oo; k < EB; k += co) v += sm[stat][comp][k];
        partial[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + stat) * c + ch] = v;
    }
}

template <int MODE>
__global__ __launch_bounds__(EB) void bwd_apply16_kernel(const u32x4e* __restrict__ dy, const u32x4e* __restrict__ act_src,
                                                          const u32x4e* __restrict__ bn_src, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope, const float* __restrict__ coef,
                                                          int64_t octs, int co, u32x4e* __restrict__ ds, u32x4e* __restrict__ dx,
                                                          int64_t group_octs) {
    const int64_t stride = (int64_t)gridDim.x * EB;
    const int c = co * 8;
    for (int64_t i0 = (int64_t)blockIdx.x * EB + threadIdx.x; i0 < octs; i0 += 2 * stride) {
        u32x4e gq[2], uq[2], oq[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = i0 + u * stride;
            const bool ok = i < octs;
            const u32x4e z = {0u, 0u, 0u, 0u};
            gq[u] = ok ? dy[i] : z;
            uq[u] = ok ? bn_src[i] : z;
            if (MODE == 0) oq[u] = ok ? act_src[i] : z;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= octs) continue;
            const int o = (int)(i % co), gi = (int)(i / group_octs);
            f32x8 g = unpack8(gq[u]);
            const f32x8 x = unpack8(uq[u]);
            if (MODE == 0) {
                const f32x8 a = unpack8(oq[u]);
#pragma unroll
                for (int k = 0; k < 8; ++k) g.v[k] *= ctl_leaky_grad(a.v[k], slope);
                if (ds) ds[i] = pack8(g);
            } else if (MODE == 1) {      // (MODE 2: dy is g already)
                const f32x8 sc = load8f(scale + gi * c + o * 8), sh = load8f(shift + gi * c + o * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) g.v[k] *= ctl_leaky_grad(x.v[k] * sc.v[k] + sh.v[k], slope);
            }
            const f32x8 A = load8f(coef + (gi * 3 + 0) * c + o * 8), B = load8f(coef + (gi * 3 + 1) * c + o * 8), C = load8f(coef + (gi * 3 + 2) * c + o * 8);
            f32x8 r;
#pragma unroll
            for (int k = 0; k < 8; ++k) r.v[k] = A.v[k] * g.v[k] + B.v[k] * x.v[k] + C.v[k];
            dx[i] = pack8(r);
        }
    }
}

__global__ __launch_bounds__(EB) void chan_sum_finalize_kernel(const float* __restrict__ partial, int blocks, int c,
                                                                float* __restrict__ out, int accumulate) {
    __shared__ double sm[8];
    const int ch = blockIdx.x;
    double s1 = 0.0;
    for (int b = threadIdx.x; b < blocks; b += EB) s1 += (double)partial[((int64_t)b * 2) * c + ch];
    s1 = block_sum_double(s1, sm);
    if (threadIdx.x == 0) out[ch] = accumulate ? out[ch] + (float)s1 : (float)s1;
}

__global__ __launch_bounds__(EB) void sumpool2_kernel(const void* __restrict__ dup, void* __restrict__ dx, int n, int h,
                                                       int w, int cq, int accumulate, unsigned m) {       // m: bit 0 dup, 1 dx
    const int64_t quads = (int64_t)n * h * w * cq;
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < quads; i += stride) {
        const int q = (int)(i % cq);
        int64_t r = i / cq;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h);
        const int64_t b = r / h;
        const int64_t row0 = ((b * 2 * h + 2 * y) * 2 * w + 2 * x) * cq + q;
        const int64_t row1 = row0 + (int64_t)2 * w * cq;
        const f32x4 a0 = ldq(dup, row0, m & 1), a1 = ldq(dup, row0 + cq, m & 1), a2 = ldq(dup, row1, m & 1), a3 = ldq(dup, row1 + cq, m & 1);
        f32x4 v;
        v.x = (a0.x + a1.x) + (a2.x + a3.x); v.y = (a0.y + a1.y) + (a2.y + a3.y);
        v.z = (a0.z + a1.z) + (a2.z + a3.z); v.w = (a0.w + a1.w) + (a2.w + a3.w);
        if (accumulate) { const f32x4 o = ldq(dx, i, m & 2); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        stq(dx, i, v, m & 2);
    }
}

__global__ __launch_bounds__(EB) void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          float* __restrict__ dx, int64_t count) {
    // (dy and y are network outputs: fp32 in every configuration; dx is 1-channel and stays fp32 as well)
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < count; i += stride) {
        const float s = y[i];
        dx[i] = dy[i] * s * (1.f - s);
    }
}

// ------------------------------------------------------------------------------------------------ STN input builders
#define MAXC 16
// Per-pixel channel rows of the label-space tensors.  CT = 4 (the 4-class maps of the path): one 16-byte load / store per pixel
// and fully unrolled loops; CT = 0: runtime channel count.  The arithmetic and its order are the same in both.
template <int CT> __device__ __forceinline__ void row_load(const float* __restrict__ base, int64_t i, int c, float* v) {
    if (CT == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(base + i * 4);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        for (int k = 0; k < c; ++k) v[k] = base[i * c + k];
    }
}
template <int CT> __device__ __forceinline__ void row_store(float* __restrict__ base, int64_t i, int c, const float* v) {
    if (CT == 4) {
        *reinterpret_cast<f32x4*>(base + i * 4) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
        for (int k = 0; k < c; ++k) base[i * c + k] = v[k];
    }
}
#define CTL_ROW_DISPATCH(kernel, vec4, grid, ...)                                \
    do {                                                                         \
        if (vec4) kernel<4><<<grid, dim3(EB), 0, S_>>>(__VA_ARGS__);             \
        else kernel<0><<<grid, dim3(EB), 0, S_>>>(__VA_ARGS__);                  \
    } while (0)
static inline bool rows_vec4(int c, const void* a, const void* b = nullptr, const void* d = nullptr) {
    return c == 4 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)d) & 15) == 0;      // 16-byte rows need 16-byte aligned tensors
}
template <int CT>
__global__ __launch_bounds__(EB) void softmax_t_fwd_kernel(const float* __restrict__ x, float inv_t, float* __restrict__ p,
                                                            int64_t pixels, int c_rt) {
    const int c = CT ? CT : c_rt;
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < pixels; i += stride) {
        float v[CT ? CT : MAXC];
        row_load<CT>(x, i, c, v);
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < c; ++k) { v[k] = v[k] * inv_t; m = fmaxf(m, v[k]); }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < c; ++k) { v[k] = expf(v[k] - m); s += v[k]; }
        const float r = 1.f / s;
#pragma unroll
        for (int k = 0; k < c; ++k) v[k] = v[k] * r;
        row_store<CT>(p, i, c, v);
    }
}
template <int CT>
__global__ __launch_bounds__(EB) void softmax_t_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                            float inv_t, float* __restrict__ dx, int64_t pixels, int c_rt) {
    const int c = CT ? CT : c_rt;
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < pixels; i += stride) {
        float pv[CT ? CT : MAXC], dv[CT ? CT : MAXC];
        row_load<CT>(p, i, c, pv);
        row_load<CT>(dp, i, c, dv);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < c; ++k) dot += pv[k] * dv[k];
#pragma unroll
        for (int k = 0; k < c; ++k) dv[k] = pv[k] * (dv[k] - dot) * inv_t;
        row_store<CT>(dx, i, c, dv);
    }
}
__global__ __launch_bounds__(EB) void onehot_kernel(const int64_t* __restrict__ label, float* __restrict__ y,
                                                     int64_t pixels, int c) {
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < pixels; i += stride) {
        const int l = (int)label[i];
        for (int k = 0; k < c; ++k) y[i * c + k] = (k == l) ? 1.f : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ losses
template <int CT>
__global__ __launch_bounds__(EB) void ce2d_partial_kernel(const float* __restrict__ logit,
                                                           const int64_t* __restrict__ label, int64_t pixels, int c_rt,
                                                           double* __restrict__ partial) {
    __shared__ double sm[8];
    const int c = CT ? CT : c_rt;
    const int64_t stride = (int64_t)gridDim.x * EB;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < pixels; i += stride) {
        float v[CT ? CT : MAXC];
        row_load<CT>(logit, i, c, v);
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < c; ++k) m = fmaxf(m, v[k]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < c; ++k) s += expf(v[k] - m);
        const int l = (int)label[i];
        float xl = 0.f;
#pragma unroll
        for (int k = 0; k < c; ++k) xl = (k == l) ? v[k] : xl;
        acc += (double)(-(xl - m - logf(s)));
    }
    acc = block_sum_double(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(EB) void scalar_finalize_kernel(const double* __restrict__ partial, int blocks, double mul,
                                                              float* __restrict__ out) {
    __shared__ double sm[8];
    double s = 0.0;
    for (int b = threadIdx.x; b < blocks; b += EB) s += partial[b];
    s = block_sum_double(s, sm);
    if (threadIdx.x == 0) out[0] = (float)(s * mul);
}
template <int CT>
__global__ __launch_bounds__(EB) void ce2d_bwd_kernel(const float* __restrict__ logit, const int64_t* __restrict__ label,
                                                       const float* __restrict__ gout, int64_t pixels, int c_rt,
                                                       float* __restrict__ dlogit) {
    const int c = CT ? CT : c_rt;
    const float gs = gout[0] / (float)pixels;
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < pixels; i += stride) {
        float v[CT ? CT : MAXC];
        row_load<CT>(logit, i, c, v);
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < c; ++k) m = fmaxf(m, v[k]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < c; ++k) { v[k] = expf(v[k] - m); s += v[k]; }
        const float r = 1.f / s;
        const int l = (int)label[i];
#pragma unroll
        for (int k = 0; k < c; ++k) v[k] = gs * (v[k] * r - ((k == l) ? 1.f : 0.f));
        row_store<CT>(dlogit, i, c, v);
    }
}
__global__ __launch_bounds__(EB) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          int64_t count, double* __restrict__ partial) {
    __shared__ double sm[8];
    const int64_t stride = (int64_t)gridDim.x * EB;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < count; i += stride) {
        const float d = a[i] - b[i];
        acc += (double)(d * d);
    }
    acc = block_sum_double(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(EB) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ gout, int64_t count, float scale,
                                                      float* __restrict__ da) {
    const float gs = gout[0] * 2.f * scale / (float)count;
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < count; i += stride) da[i] = gs * (a[i] - b[i]);
}
__global__ __launch_bounds__(EB) void argmax_kernel(const float* __restrict__ logit, uint8_t* __restrict__ out,
                                                     int64_t pixels, int c) {
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < pixels; i += stride) {
        float best = logit[i * c];
        int bi = 0;
        for (int k = 1; k < c; ++k) {
            const float v = logit[i * c + k];
            if (v > best) { best = v; bi = k; }
        }
        out[i] = (uint8_t)bi;
    }
}

// ------------------------------------------------------------------------------------------------ Adam
__global__ __launch_bounds__(EB) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t count, float lr, float b1, float b2,
                                                   float eps, float bc1, float bc2_sqrt, float gscale,
                                                   const int64_t* __restrict__ state) {
    // torch.optim.Adam (no amsgrad, no weight decay): denom = sqrt(v)/sqrt(bc2) + eps; p -= lr/bc1 * m/denom
    if (state) {      // step count lives on the device (HIP-graph replays): same double-precision bias corrections as the host path
        const double st = (double)state[2];
        bc1 = (float)(1.0 - pow((double)b1, st));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, st));
    }
    const int64_t stride = (int64_t)gridDim.x * EB;
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < count; i += stride) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// dst += src[0] + src[1] + ... (fixed order): the parameter gradients of the passes of one network, summed once per step
struct ctl_ptr8 { const float* p[8]; };
__global__ __launch_bounds__(EB) void accumulate_kernel(float* __restrict__ dst, ctl_ptr8 src, int k, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < count; i += stride) {
        float a = dst[i];
        for (int j = 0; j < k; ++j) a += src.p[j][i];
        dst[i] = a;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
#define S_ (hipStream_t) stream

extern "C" int ctl_accumulate(float* dst, const float* const* srcs, int32_t k, int64_t count, ctl_stream stream) {
    CTL_REQUIRE(dst && srcs && k >= 1 && k <= 8 && count > 0, "accumulate: bad arguments (1 <= k <= 8)");
    ctl_ptr8 a;
    for (int j = 0; j < 8; ++j) a.p[j] = j < k ? srcs[j] : nullptr;
    for (int j = 0; j < k; ++j) CTL_REQUIRE(a.p[j], "accumulate: null source %d", j);
    accumulate_kernel<<<dim3(stream_blocks(count)), dim3(EB), 0, S_>>>(dst, a, k, count);
    CTL_LAUNCH_CHECK("accumulate");
    return CTL_OK;
}

// Running-statistics update of BatchNorm layers REPLAYED from saved batch statistics: a second forward pass of a network over the same
// input in training mode (the saliency pass of the targeted latent masks, model_util.py:214: `decoder_function(code)` on the code the
// standard pass has just decoded) computes the same batch statistics and moves the running statistics once more by the same two
// floats.  The engine re-uses the first pass' activations and replays only this update.  table: n_rec records of 6 int64
// {mean byte offset in `act`, uvar byte offset in `act`, running_mean float offset in `buffers`, running_var float offset, nbt index, c}.
__global__ void bn_replay_running_kernel(const char* __restrict__ act, float* __restrict__ buffers, int64_t* __restrict__ nbt,
                                         const int64_t* __restrict__ table, float momentum) {
    const int64_t* r = table + (int64_t)blockIdx.x * 6;
    const float* mean = reinterpret_cast<const float*>(act + r[0]);
    const float* uvar = reinterpret_cast<const float*>(act + r[1]);
    float* rm = buffers + r[2];
    float* rv = buffers + r[3];
    const int c = (int)r[5];
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        rm[ch] = (1.f - momentum) * rm[ch] + momentum * mean[ch];
        rv[ch] = (1.f - momentum) * rv[ch] + momentum * uvar[ch];
    }
    if (threadIdx.x == 0) nbt[r[4]] += 1;
}
extern "C" int ctl_bn_replay_running(const void* act, float* buffers, int64_t* nbt, const int64_t* table, int32_t n_rec, float momentum,
                                     ctl_stream stream) {
    CTL_REQUIRE(act && buffers && nbt && table && n_rec > 0, "bn_replay_running: bad arguments");
    bn_replay_running_kernel<<<dim3((unsigned)n_rec), dim3(128), 0, (hipStream_t)stream>>>((const char*)act, buffers, nbt, table, momentum);
    CTL_LAUNCH_CHECK("bn_replay_running");
    return CTL_OK;
}

#ifdef CTL_TUNING
__global__ void fin_empty_kernel(float* p) { if (p == nullptr) p[0] = 0.f; }
#endif
extern "C" int ctl_bn_finalize(const float* partial, int32_t blocks, int32_t c, int64_t count, const float* gamma,
                               const float* beta, float eps, float momentum, int32_t update_running,
                               float* running_mean, float* running_var, int64_t* nbt, float* scale, float* shift,
                               float* save_mean, float* save_invstd, int32_t groups, ctl_stream stream) {
    return ctl_bn_finalize_ex(partial, blocks, c, count, gamma, beta, eps, momentum, update_running, running_mean, running_var, nbt, scale, shift,
                              save_mean, save_invstd, nullptr, groups, stream);
}
extern "C" int ctl_bn_finalize_ex(const float* partial, int32_t blocks, int32_t c, int64_t count, const float* gamma,
                                  const float* beta, float eps, float momentum, int32_t update_running,
                                  float* running_mean, float* running_var, int64_t* nbt, float* scale, float* shift,
                                  float* save_mean, float* save_invstd, float* save_uvar, int32_t groups, ctl_stream stream) {
    CTL_REQUIRE(partial && gamma && beta && scale && shift && blocks > 0 && c > 0 && count > 0 && groups >= 1, "bn_finalize: bad arguments");
    CTL_REQUIRE(!update_running || (running_mean && running_var), "bn_finalize: update_running without buffers");
#ifdef CTL_TUNING      // ceiling probe (wrong numbers): 1 = an empty kernel of the same grid, 2 = ONE wave that does nothing (tools/debug/fin_empty_probe.sh)
    static const int fin_empty = ctl_tune_int("CTL_FIN_EMPTY", 0);
    if (fin_empty) {
        fin_empty_kernel<<<dim3(fin_empty == 2 ? 1 : c), dim3(fin_empty == 2 ? 64 : CTL_FIN_THREADS), 0, S_>>>(scale);
        CTL_LAUNCH_CHECK("bn_finalize");
        return CTL_OK;
    }
#endif
    bn_finalize_kernel<<<dim3(c), dim3(CTL_FIN_THREADS), 0, S_>>>(partial, blocks, c, (double)count, gamma, beta, eps, momentum,
                                                      update_running, running_mean, running_var, nbt, scale, shift,
                                                      save_mean, save_invstd, groups, save_uvar);
    CTL_LAUNCH_CHECK("bn_finalize");
    return CTL_OK;
}
extern "C" int ctl_bn_eval_coeffs(int32_t c, const float* gamma, const float* beta, const float* rm, const float* rv,
                                  float eps, float* scale, float* shift, int32_t groups, ctl_stream stream) {
    CTL_REQUIRE(c > 0 && gamma && beta && rm && rv && scale && shift && groups >= 1, "bn_eval_coeffs: bad arguments");
    bn_eval_kernel<<<dim3(ctl_cdiv(c, 64)), dim3(64), 0, S_>>>(c, gamma, beta, rm, rv, eps, scale, shift, groups);
    CTL_LAUNCH_CHECK("bn_eval_coeffs");
    return CTL_OK;
}
extern "C" int ctl_bn_act_dt(const float* x, const float* scale, const float* shift, float slope, float* y, int64_t pixels,
                             int32_t c, int32_t groups, uint32_t bf16_mask, ctl_stream stream) {
    CTL_REQUIRE(x && y && scale && shift && c % 4 == 0 && pixels > 0 && groups >= 1 && pixels % groups == 0,
                "bn_act: bad arguments (c must be a multiple of 4, pixels of groups)");
    const int64_t quads = pixels * (c / 4);
    bn_act_kernel<<<dim3(stream_blocks(quads)), dim3(EB), 0, S_>>>(x, (const f32x4*)scale, (const f32x4*)shift, slope, y, quads, c / 4,
                                                                   quads / groups, bf16_mask);
    CTL_LAUNCH_CHECK("bn_act");
    return CTL_OK;
}
extern "C" int ctl_bn_act(const float* x, const float* scale, const float* shift, float slope, float* y, int64_t pixels,
                          int32_t c, int32_t groups, ctl_stream stream) {
    return ctl_bn_act_dt(x, scale, shift, slope, y, pixels, c, groups, 0, stream);
}
static bool red_c_ok(int c) { return c >= 4 && c % 4 == 0 && (EB % (c / 4)) == 0; }

extern "C" int ctl_red_blocks(void) { return CTL_RED_BLOCKS; }
extern "C" int ctl_bwd_reduce_rows(int32_t mode, int64_t pixels_per_group, int32_t c) {
    return mode == 2 ? CTL_RED_BLOCKS : red_rows_for(pixels_per_group * (c / 4));
}
extern "C" int ctl_bwd_reduce_dt(int32_t mode, const float* dy, const float* act_src, const float* bn_src,
                                 const float* scale, const float* shift, float slope, int64_t pixels, int32_t c,
                                 float* partial, int32_t groups, uint32_t bf16_mask, float* ds, ctl_stream stream) {
    CTL_REQUIRE(!ds || mode == 0, "bwd_reduce: ds (= dy * leaky'(act_src)) is an output of mode 0 only");
    CTL_REQUIRE(dy && partial && pixels > 0 && red_c_ok(c) && groups >= 1 && pixels % groups == 0, "bwd_reduce: bad arguments (c=%d)", c);
    const int64_t quads = (pixels / groups) * (c / 4);           // per group
    // modes 0 / 1 feed ctl_bn_bwd_finalize, which derives the same row count from (count, c); mode 2 feeds ctl_chan_sum_finalize (fixed rows)
    const dim3 grid((unsigned)(mode == 2 ? CTL_RED_BLOCKS : red_rows_for(quads)), (unsigned)groups), blk(EB);
    // every tensor stored as bf16 and whole channel octets: 16 bytes per lane
    const bool oct = c % 8 == 0 && (EB % (c / 8)) == 0 && (pixels / groups) * (int64_t)(c / 8) >= 1 && (!ds || (bf16_mask & 8u)) &&
                     ((mode == 0 && (bf16_mask & 7u) == 7u) || (mode == 1 && (bf16_mask & 5u) == 5u));
    if (mode == 0) {
        CTL_REQUIRE(act_src && bn_src, "bwd_reduce mode 0 needs act_src and bn_src");
        if (oct)
            bwd_reduce16_kernel<0><<<grid, blk, 0, S_>>>((const u32x4e*)dy, (const u32x4e*)act_src, (const u32x4e*)bn_src, nullptr, nullptr, slope,
                                                        quads / 2, c / 8, partial, (u32x4e*)ds);
        else
            bwd_reduce_kernel<0><<<grid, blk, 0, S_>>>(dy, act_src, bn_src, nullptr, nullptr, slope, quads, c / 4, partial, bf16_mask, ds);
    } else if (mode == 1) {
        CTL_REQUIRE(bn_src && scale && shift, "bwd_reduce mode 1 needs bn_src, scale, shift");
        if (oct)
            bwd_reduce16_kernel<1><<<grid, blk, 0, S_>>>((const u32x4e*)dy, nullptr, (const u32x4e*)bn_src, scale, shift, slope, quads / 2, c / 8,
                                                        partial, nullptr);
        else
            bwd_reduce_kernel<1><<<grid, blk, 0, S_>>>(dy, nullptr, bn_src, (const f32x4*)scale, (const f32x4*)shift, slope, quads, c / 4, partial,
                                                      bf16_mask, nullptr);
    } else if (mode == 2) {
        CTL_REQUIRE(groups == 1, "bwd_reduce mode 2 sums everything: groups must be 1");
        bwd_reduce_kernel<2><<<grid, blk, 0, S_>>>(dy, nullptr, nullptr, nullptr, nullptr, slope, quads, c / 4, partial, bf16_mask, nullptr);
    } else {
        CTL_FAIL(CTL_EINVAL, "bwd_reduce: mode %d", mode);
    }
    CTL_LAUNCH_CHECK("bwd_reduce");
    return CTL_OK;
}
extern "C" int ctl_bwd_reduce(int32_t mode, const float* dy, const float* act_src, const float* bn_src,
                              const float* scale, const float* shift, float slope, int64_t pixels, int32_t c,
                              float* partial, int32_t groups, ctl_stream stream) {
    return ctl_bwd_reduce_dt(mode, dy, act_src, bn_src, scale, shift, slope, pixels, c, partial, groups, 0, nullptr, stream);
}
extern "C" int ctl_bn_bwd_finalize_ex(const float* partial, int32_t c, int64_t count, const float* gamma,
                                      const float* save_mean, const float* save_invstd, float* coef, float* dgamma,
                                      float* dbeta, int32_t accumulate, int32_t groups, int32_t blocks, uint32_t affine_groups, ctl_stream stream) {
    CTL_REQUIRE(partial && gamma && save_mean && save_invstd && coef && c > 0 && count > 0 && groups >= 1 && groups <= 32 && blocks >= 0, "bn_bwd_finalize: bad arguments");
    // blocks == 0: rows as written by ctl_bwd_reduce for a group of `count` pixels.  affine_groups: bit g = group g adds to dgamma / dbeta (0 = every group)
    bn_bwd_finalize_kernel<<<dim3(c), dim3(CTL_FIN_THREADS), 0, S_>>>(partial, blocks > 0 ? blocks : red_rows_for(count * (c / 4)), c, (double)count, gamma, save_mean,
                                                          save_invstd, coef, dgamma, dbeta, accumulate, groups, affine_groups ? affine_groups : 0xffffffffu);
    CTL_LAUNCH_CHECK("bn_bwd_finalize");
    return CTL_OK;
}
extern "C" int ctl_bn_bwd_finalize(const float* partial, int32_t c, int64_t count, const float* gamma,
                                   const float* save_mean, const float* save_invstd, float* coef, float* dgamma,
                                   float* dbeta, int32_t accumulate, int32_t groups, int32_t blocks, ctl_stream stream) {
    return ctl_bn_bwd_finalize_ex(partial, c, count, gamma, save_mean, save_invstd, coef, dgamma, dbeta, accumulate, groups, blocks, 0u, stream);
}
extern "C" int ctl_bwd_apply_dt(int32_t mode, const float* dy, const float* act_src, const float* bn_src,
                                const float* scale, const float* shift, float slope, const float* coef, int64_t pixels,
                                int32_t c, float* ds, float* dx, int32_t groups, uint32_t bf16_mask, ctl_stream stream) {
    CTL_REQUIRE(dy && bn_src && coef && dx && pixels > 0 && c % 4 == 0 && groups >= 1 && pixels % groups == 0, "bwd_apply: bad arguments");
    const int64_t quads = pixels * (c / 4);
    const dim3 grid(stream_blocks(quads)), blk(EB);
    // every tensor stored as bf16 and whole channel octets: 16 bytes per lane (mask bits: 0 dy, 1 act_src, 2 bn_src, 3 ds, 4 dx)
    const unsigned need = mode == 0 ? (1u | 2u | 4u | 16u | (ds ? 8u : 0u)) : (1u | 4u | 16u);
    const bool oct = c % 8 == 0 && (bf16_mask & need) == need;
    const dim3 grid8(stream_blocks(quads / 4));      // two octets per thread and trip
    if (mode == 0) {
        CTL_REQUIRE(act_src, "bwd_apply mode 0 needs act_src");
        if (oct)
            bwd_apply16_kernel<0><<<grid8, blk, 0, S_>>>((const u32x4e*)dy, (const u32x4e*)act_src, (const u32x4e*)bn_src, nullptr, nullptr, slope,
                                                        coef, quads / 2, c / 8, (u32x4e*)ds, (u32x4e*)dx, quads / 2 / groups);
        else
            bwd_apply_kernel<0><<<grid, blk, 0, S_>>>(dy, act_src, bn_src, nullptr, nullptr, slope, (const f32x4*)coef, quads, c / 4, ds, dx,
                                                     quads / groups, bf16_mask);
    } else if (mode == 1) {
        CTL_REQUIRE(scale && shift, "bwd_apply mode 1 needs scale and shift");
        if (oct)
            bwd_apply16_kernel<1><<<grid8, blk, 0, S_>>>((const u32x4e*)dy, nullptr, (const u32x4e*)bn_src, scale, shift, slope, coef, quads / 2,
                                                        c / 8, nullptr, (u32x4e*)dx, quads / 2 / groups);
        else
            bwd_apply_kernel<1><<<grid, blk, 0, S_>>>(dy, nullptr, bn_src, (const f32x4*)scale, (const f32x4*)shift, slope, (const f32x4*)coef, quads,
                                                     c / 4, nullptr, dx, quads / groups, bf16_mask);
    } else if (mode == 2) {
        if (oct)
            bwd_apply16_kernel<2><<<grid8, blk, 0, S_>>>((const u32x4e*)dy, nullptr, (const u32x4e*)bn_src, nullptr, nullptr, slope, coef, quads / 2,
                                                        c / 8, nullptr, (u32x4e*)dx, quads / 2 / groups);
        else
            bwd_apply_kernel<2><<<grid, blk, 0, S_>>>(dy, nullptr, bn_src, nullptr, nullptr, slope, (const f32x4*)coef, quads, c / 4, nullptr, dx,
                                                     quads / groups, bf16_mask);
    } else {
        CTL_FAIL(CTL_EINVAL, "bwd_apply: mode %d", mode);
    }
    CTL_LAUNCH_CHECK("bwd_apply");
    return CTL_OK;
}
extern "C" int ctl_bwd_apply(int32_t mode, const float* dy, const float* act_src, const float* bn_src,
                             const float* scale, const float* shift, float slope, const float* coef, int64_t pixels,
                             int32_t c, float* ds, float* dx, int32_t groups, ctl_stream stream) {
    return ctl_bwd_apply_dt(mode, dy, act_src, bn_src, scale, shift, slope, coef, pixels, c, ds, dx, groups, 0, stream);
}
extern "C" int ctl_chan_sum_finalize(const float* partial, int32_t c, float* out, int32_t accumulate, ctl_stream stream) {
    CTL_REQUIRE(partial && out && c > 0, "chan_sum_finalize: bad arguments");
    chan_sum_finalize_kernel<<<dim3(c), dim3(EB), 0, S_>>>(partial, CTL_RED_BLOCKS, c, out, accumulate);
    CTL_LAUNCH_CHECK("chan_sum_finalize");
    return CTL_OK;
}
extern "C" int ctl_sumpool2_dt(const float* dup, float* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t accumulate,
                               uint32_t bf16_mask, ctl_stream stream) {
    CTL_REQUIRE(dup && dx && n > 0 && h > 0 && w > 0 && c % 4 == 0, "sumpool2: bad arguments");
    const int64_t quads = (int64_t)n * h * w * (c / 4);
    sumpool2_kernel<<<dim3(stream_blocks(quads)), dim3(EB), 0, S_>>>(dup, dx, n, h, w, c / 4, accumulate, bf16_mask);
    CTL_LAUNCH_CHECK("sumpool2");
    return CTL_OK;
}
extern "C" int ctl_sumpool2(const float* dup, float* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t accumulate,
                            ctl_stream stream) {
    return ctl_sumpool2_dt(dup, dx, n, h, w, c, accumulate, 0, stream);
}
extern "C" int ctl_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t count, ctl_stream stream) {
    CTL_REQUIRE(dy && y && dx && count > 0, "sigmoid_bwd: bad arguments");
    sigmoid_bwd_kernel<<<dim3(stream_blocks(count)), dim3(EB), 0, S_>>>(dy, y, dx, count);
    CTL_LAUNCH_CHECK("sigmoid_bwd");
    return CTL_OK;
}
extern "C" int ctl_softmax_t_fwd(const float* x, float inv_t, float* p, int64_t pixels, int32_t c, ctl_stream stream) {
    CTL_REQUIRE(x && p && pixels > 0 && c > 0 && c <= MAXC, "softmax_t_fwd: bad arguments");
    CTL_ROW_DISPATCH(softmax_t_fwd_kernel, rows_vec4(c, x, p), dim3(stream_blocks(pixels)), x, inv_t, p, pixels, c);
    CTL_LAUNCH_CHECK("softmax_t_fwd");
    return CTL_OK;
}
extern "C" int ctl_softmax_t_bwd(const float* p, const float* dp, float inv_t, float* dx, int64_t pixels, int32_t c,
                                 ctl_stream stream) {
    CTL_REQUIRE(p && dp && dx && pixels > 0 && c > 0 && c <= MAXC, "softmax_t_bwd: bad arguments");
    CTL_ROW_DISPATCH(softmax_t_bwd_kernel, rows_vec4(c, p, dp, dx), dim3(stream_blocks(pixels)), p, dp, inv_t, dx, pixels, c);
    CTL_LAUNCH_CHECK("softmax_t_bwd");
    return CTL_OK;
}
extern "C" int ctl_onehot(const int64_t* label, float* y, int64_t pixels, int32_t c, ctl_stream stream) {
    CTL_REQUIRE(label && y && pixels > 0 && c > 0, "onehot: bad arguments");
    onehot_kernel<<<dim3(stream_blocks(pixels)), dim3(EB), 0, S_>>>(label, y, pixels, c);
    CTL_LAUNCH_CHECK("onehot");
    return CTL_OK;
}
extern "C" int ctl_ce2d_fwd(const float* logit, const int64_t* label, int64_t pixels, int32_t c, double* partial,
                            float* loss, ctl_stream stream) {
    CTL_REQUIRE(logit && label && partial && loss && pixels > 0 && c > 0 && c <= MAXC, "ce2d_fwd: bad arguments");
    CTL_ROW_DISPATCH(ce2d_partial_kernel, rows_vec4(c, logit), dim3(CTL_RED_BLOCKS), logit, label, pixels, c, partial);
    scalar_finalize_kernel<<<dim3(1), dim3(EB), 0, S_>>>(partial, CTL_RED_BLOCKS, 1.0 / (double)pixels, loss);
    ctl_count_launches(1);      // partial + finalize
    CTL_LAUNCH_CHECK("ce2d_fwd");
    return CTL_OK;
}
extern "C" int ctl_ce2d_bwd(const float* logit, const int64_t* label, const float* gout, int64_t pixels, int32_t c,
                            float* dlogit, ctl_stream stream) {
    CTL_REQUIRE(logit && label && gout && dlogit && pixels > 0 && c > 0 && c <= MAXC, "ce2d_bwd: bad arguments");
    CTL_ROW_DISPATCH(ce2d_bwd_kernel, rows_vec4(c, logit, dlogit), dim3(stream_blocks(pixels)), logit, label, gout, pixels, c, dlogit);
    CTL_LAUNCH_CHECK("ce2d_bwd");
    return CTL_OK;
}
extern "C" int ctl_mse_fwd(const float* a, const float* b, int64_t count, float scale, double* partial, float* loss,
                           ctl_stream stream) {
    CTL_REQUIRE(a && b && partial && loss && count > 0, "mse_fwd: bad arguments");
    mse_partial_kernel<<<dim3(CTL_RED_BLOCKS), dim3(EB), 0, S_>>>(a, b, count, partial);
    scalar_finalize_kernel<<<dim3(1), dim3(EB), 0, S_>>>(partial, CTL_RED_BLOCKS, (double)scale / (double)count, loss);
    ctl_count_launches(1);      // partial + finalize
    CTL_LAUNCH_CHECK("mse_fwd");
    return CTL_OK;
}
extern "C" int ctl_mse_bwd(const float* a, const float* b, const float* gout, int64_t count, float scale, float* da,
                           ctl_stream stream) {
    CTL_REQUIRE(a && b && gout && da && count > 0, "mse_bwd: bad arguments");
    mse_bwd_kernel<<<dim3(stream_blocks(count)), dim3(EB), 0, S_>>>(a, b, gout, count, scale, da);
    CTL_LAUNCH_CHECK("mse_bwd");
    return CTL_OK;
}
extern "C" int ctl_argmax_c(const float* logit, uint8_t* out, int64_t pixels, int32_t c, ctl_stream stream) {
    CTL_REQUIRE(logit && out && pixels > 0 && c > 0 && c < 256, "argmax_c: bad arguments");
    argmax_kernel<<<dim3(stream_blocks(pixels)), dim3(EB), 0, S_>>>(logit, out, pixels, c);
    CTL_LAUNCH_CHECK("argmax_c");
    return CTL_OK;
}
extern "C" int ctl_adam(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                        float eps, int32_t step, float grad_scale, ctl_stream stream) {
    CTL_REQUIRE(p && g && m && v && count > 0 && step >= 1, "adam: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    adam_kernel<<<dim3(stream_blocks(count)), dim3(EB), 0, S_>>>(p, g, m, v, count, lr, beta1, beta2, eps, (float)bc1,
                                                                 (float)sqrt(bc2), grad_scale, nullptr);
    CTL_LAUNCH_CHECK("adam");
    return CTL_OK;
}
extern "C" int ctl_adam_dev(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                            float eps, const int64_t* state, float grad_scale, ctl_stream stream) {
    CTL_REQUIRE(p && g && m && v && count > 0 && state, "adam_dev: bad arguments");
    adam_kernel<<<dim3(stream_blocks(count)), dim3(EB), 0, S_>>>(p, g, m, v, count, lr, beta1, beta2, eps, 1.f, 1.f, grad_scale, state);
    CTL_LAUNCH_CHECK("adam_dev");
    return CTL_OK;
}
