// Internal helpers shared by the libctl_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "ctl_hip.h"

void ctl_set_error(const char* fmt, ...);

#define CTL_FAIL(code, ...)            \
    do {                               \
        ctl_set_error(__VA_ARGS__);    \
        return (code);                 \
    } while (0)

#define CTL_REQUIRE(cond, ...)                        \
    do {                                              \
        if (!(cond)) CTL_FAIL(CTL_EINVAL, __VA_ARGS__); \
    } while (0)

// every launcher ends with this: catches bad launch configurations without synchronising
void ctl_count_launches(int n);      // launch census (ctl_launch_count): every launcher counts ONE here, multi-kernel launchers add the rest
#define CTL_LAUNCH_CHECK(name)                                                         \
    do {                                                                               \
        ctl_count_launches(1);                                                         \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) CTL_FAIL(CTL_ELAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

__host__ __device__ static inline int ctl_cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline int64_t ctl_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ctl_leaky(float v, float slope) { return v > 0.f ? v : v * slope; }
// derivative factor chosen from the sign of the activation *output* (== sign of its input for slope >= 0)
__device__ __forceinline__ float ctl_leaky_grad(float out, float slope) { return out > 0.f ? 1.f : slope; }

// Workgroup barriers for the pipelined loops.  __syncthreads() is a workgroup-scope FENCE + barrier: hipcc puts
// s_waitcnt vmcnt(0) in front of it, which drains every in-flight global load AND store of the wave (seen in the ISA; it made
// prefetch and epilogue stores strictly serial with the MFMA phase).  The loops only need LDS ordering:
//   ctl_barrier_lds_reads_done : all waves finished READING an LDS image (their ds_reads were consumed by MFMAs, i.e.
//                                already waited for) -> bare s_barrier
//   ctl_barrier_lds_writes_done: ds_writes of this wave have landed (lgkmcnt(0)) and everybody arrived
__device__ __forceinline__ void ctl_barrier_lds_reads_done() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void ctl_barrier_lds_writes_done() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Tuning hooks: the shipped library reads NO environment variables.  A -DCTL_TUNING build (tools/ab.sh) turns the call sites below
// back into getenv() reads for A/B experiments.
#ifdef CTL_TUNING
#include <stdlib.h>
static inline int ctl_tune_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline const char* ctl_tune_str(const char* name) { return getenv(name); }
#else
static inline constexpr int ctl_tune_int(const char*, int dflt) { return dflt; }
static inline constexpr const char* ctl_tune_str(const char*) { return nullptr; }
#endif

// Storage-type aware quad access for the kernels that touch network-internal tensors: with BASELINE config 3 those are stored as bf16
// (`m` = bit mask over the kernel's tensor arguments, bit set = bf16 storage); arithmetic is fp32 either way, a store rounds once (RNE).
typedef unsigned int u32x2e __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2e __attribute__((ext_vector_type(2)));
typedef float f32x2e __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 ldq(const void* __restrict__ p, int64_t i, bool b16) {
    if (b16) {
        const u32x2e u = reinterpret_cast<const u32x2e*>(p)[i];
        return f32x4{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
    }
    return reinterpret_cast<const f32x4*>(p)[i];
}
#ifndef CTL_ELEM_NT
#define CTL_ELEM_NT 0        // experiment hook: 1 = non-temporal stores of the element-wise kernels' output tensors
#endif
__device__ __forceinline__ void stq(void* __restrict__ p, int64_t i, f32x4 v, bool b16) {
    if (b16) {
        const u32x2e pk = u32x2e{__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2e{v.x, v.y}, bf16x2e)),
                                 __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2e{v.z, v.w}, bf16x2e))};
        if (CTL_ELEM_NT) __builtin_nontemporal_store(pk, reinterpret_cast<u32x2e*>(p) + i); else reinterpret_cast<u32x2e*>(p)[i] = pk;
    } else {
        if (CTL_ELEM_NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p) + i); else reinterpret_cast<f32x4*>(p)[i] = v;
    }
}

__device__ __forceinline__ double wave_sum_double(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// BatchNorm forward finalize of one channel from its two sums over `count` pixels (bn_finalize_kernel): biased variance for the
// normalisation, unbiased for running_var, momentum update in place.
struct ctl_bn_chan { float gamma, beta, rm, rv; };       // a channel's parameters, requested BEFORE the reduction whose result they meet
__device__ __forceinline__ ctl_bn_chan ctl_bn_chan_load(int ch, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int update_running, const float* __restrict__ running_mean,
                                                        const float* __restrict__ running_var) {
    ctl_bn_chan p;
    p.gamma = gamma[ch]; p.beta = beta[ch];
    p.rm = update_running ? running_mean[ch] : 0.f;
    p.rv = update_running ? running_var[ch] : 0.f;
    return p;
}
// p.rm / p.rv are carried from group to group (the running statistics see the groups in order) and stored every time
__device__ __forceinline__ void ctl_bn_coefs(double s1, double s2, double count, int c, int g, int ch, ctl_bn_chan& p, float eps,
                                             float momentum, int update_running, float* __restrict__ running_mean,
                                             float* __restrict__ running_var, float* __restrict__ scale, float* __restrict__ shift,
                                             float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                             float* __restrict__ save_uvar = nullptr) {
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;   // biased variance, as F.batch_norm normalises with
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = p.gamma * invstd;
    scale[g * c + ch] = sc;
    shift[g * c + ch] = p.beta - (float)mean * sc;
    if (save_mean) save_mean[g * c + ch] = (float)mean;
    if (save_invstd) save_invstd[g * c + ch] = invstd;
    if (update_running) {   // nn.BatchNorm2d: momentum 0.1, running_var uses the unbiased estimate
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        if (save_uvar) save_uvar[g * c + ch] = (float)unbiased;      // (with save_mean: the two floats of the running update, for ctl_bn_replay_running)
        p.rm = (1.f - momentum) * p.rm + momentum * (float)mean;
        p.rv = (1.f - momentum) * p.rv + momentum * (float)unbiased;
        running_mean[ch] = p.rm;
        running_var[ch] = p.rv;
    }
}
// conv descriptors shared between ctl_conv.hip and ctl_plan.cpp
struct ctl_conv_cfg {
    int mt, tw, nt;      // M-tiles per wave, tile width in pixels, cout tiles (of 16) per block
    int th;              // tile height
    int tiles_h, tiles_w;
    int g;               // cin chunks of 16
    int cot;             // cout tiles of 16 (total)
    int pc;              // X3 forward-type launches: the producer / consumer form (ctl_conv_igemm.h, PC) was chosen
};
int ctl_conv_pick_cfg(const ctl_conv* d, ctl_conv_cfg* c, int for_wgrad);
int ctl_conv_grid_x(int ntiles, int other, int occ);      // persistent grid: the resident capacity (ctl_conv.hip)
// CUs a launch may count on: 256 on MI355X.  (tuning builds: CTL_NUM_CUS, for experiments with CU-masked streams -- tools/cumask_probe.py)
static inline int ctl_num_cus() { static const int n = ctl_tune_int("CTL_NUM_CUS", 256); return n; }

// bf16 kernel family (ctl_conv_bf16.hip), reached through the public entry points when ctl_conv.dt has CTL_DT_BF16
int ctl_conv_forward_bf16(const ctl_conv* d, const void* x, const void* x2, const void* wpack, const float* bias, const float* pro_scale,
                          const float* pro_shift, const void* res, const float* res_scale, const float* res_shift, const void* res2, void* y,
                          float* stats_partial, void* pool, void* xout, ctl_stream stream);
int ctl_conv_bf16_stats_blocks(const ctl_conv* d);
int ctl_wgrad_bf16_splits(const ctl_conv* d);
int ctl_conv_wgrad_bf16(const ctl_conv* d, const void* x, const float* pro_scale, const float* pro_shift, const void* dy, const void* dy2,
                        const float* dy_coef, float* w_partial, float* b_partial, ctl_stream stream);

// stacked grouped launches of the bf16 weight gradients (ctl_wgrad_bf16.hip), reached through ctl_wgrad_group_class / ctl_conv_wgrad_group
int ctl_wgrad_bf16_group_class(const ctl_conv* d, int has_dy2);      // 0x100 | instantiation key, or -1
int ctl_wgrad_bf16_group_plan(const ctl_conv* descs, int n, int32_t* splits);
int ctl_conv_wgrad_bf16_group(int n, const ctl_conv* descs, const int32_t* splits, const void* const* x, const float* const* pro_scale,
                              const float* const* pro_shift, const void* const* dy, const void* const* dy2, const float* const* dy_coef,
                              float* const* w_partial, float* const* b_partial, ctl_stream stream);

// X3 half of the fp32-storage family (ctl_conv_x3.hip), reached through the public entry points when ctl_conv.dt has CTL_DT_X3
int ctl_conv_x3_ok(const ctl_conv* d);
int ctl_conv_x3_stats_blocks(const ctl_conv* d);
int ctl_conv_forward_x3(const ctl_conv* d, const float* x, const float* wpack, const float* bias, const float* pro_scale, const float* pro_shift,
                        const float* res, const float* res_scale, const float* res_shift, const float* res2, const float* x2, float* y,
                        float* stats_partial, float* pool, float* xout, ctl_stream stream);

int ctl_wgrad_x3_ok(const ctl_conv* d);
int ctl_wgrad_x3_splits(const ctl_conv* d);
int ctl_conv_wgrad_x3(const ctl_conv* d, const float* x, const float* pro_scale, const float* pro_shift, const float* dy, const float* dy2,
                      const float* dy_coef, float* w_partial, float* b_partial, ctl_stream stream);

// in-process profiling (ctl_plan.cpp): returns a token >= 0 if this launch is being timed
int ctl_prof_begin(const char* kind, const ctl_conv* d, const ctl_conv_cfg* c, int nt, hipStream_t stream, bool dy2 = false);
void ctl_prof_end(int token, hipStream_t stream);
int ctl_prof_begin_raw(const char* id, double flops, double bytes, hipStream_t stream);
