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
#define CTL_LAUNCH_CHECK(name)                                                         \
    do {                                                                               \
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

// conv descriptors shared between ctl_conv.hip and ctl_plan.cpp
struct ctl_conv_cfg {
    int mt, tw, nt;      // M-tiles per wave, tile width in pixels, cout tiles (of 16) per block
    int th;              // tile height
    int tiles_h, tiles_w;
    int g;               // cin chunks of 16
    int cot;             // cout tiles of 16 (total)
};
int ctl_conv_pick_cfg(const ctl_conv* d, ctl_conv_cfg* c, int for_wgrad);
int ctl_conv_grid_x(int ntiles, int other, int occ);      // persistent grid: the resident capacity (ctl_conv.hip)

// bf16 kernel family (ctl_conv_bf16.hip), reached through the public entry points when ctl_conv.dt has CTL_DT_BF16
int ctl_conv_forward_bf16(const ctl_conv* d, const void* x, const void* wpack, const float* bias, const float* pro_scale,
                          const float* pro_shift, const void* res, const float* res_scale, const float* res_shift, void* y,
                          float* stats_partial, ctl_stream stream);
int ctl_conv_bf16_stats_blocks(const ctl_conv* d);
int ctl_wgrad_bf16_splits(const ctl_conv* d);
int ctl_conv_wgrad_bf16(const ctl_conv* d, const void* x, const float* pro_scale, const float* pro_shift, const void* dy,
                        float* w_partial, float* b_partial, ctl_stream stream);

// in-process profiling (ctl_plan.cpp): returns a token >= 0 if this launch is being timed
int ctl_prof_begin(const char* kind, const ctl_conv* d, const ctl_conv_cfg* c, int nt, hipStream_t stream);
void ctl_prof_end(int token, hipStream_t stream);
