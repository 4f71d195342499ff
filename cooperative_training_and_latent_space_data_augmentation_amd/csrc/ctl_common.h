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

// Hand-off of per-block partial rows to the block that arrives last at a device-scope counter (it finalises them in the same
// launch: no separate finalize kernel, no kernel boundary).  MI355X: the 8 XCD L2s are not coherent with each other, so the rows are
// stored write-through (agent-scope relaxed atomic store = global_store sc1), every storing wave drains its stores (vmcnt(0)) before
// the workgroup barrier in front of the arrival add, and the last block takes an agent-scope acquire before it reads the rows with
// agent-scope loads (global_load sc1).  The counter is left at zero for the next launch.
__device__ __forceinline__ void ctl_store_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ctl_load_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Arrival counters are SHARDED: `base` points at CTL_ARRIVE_LINES 128-byte lines -- line 0 the top counter, lines 1..8 one shard
// each (block L arrives at shard L % 8; the last arriver of a shard arrives at the top).  Hundreds of adds to ONE address serialise at
// ~12 ns each on this part (768 blocks: 9 us per launch, measured); 8 shards on lines of their own run side by side.
#ifndef CTL_ARRIVE_ACQUIRE
#define CTL_ARRIVE_ACQUIRE 1      /* experiment hook: 0 = no agent-scope acquire (L2 invalidate) in the last block; it reads the rows with agent-scope loads anyway */
#endif
#define CTL_ARRIVE_SHARDS 8
#define CTL_ARRIVE_LINES (CTL_ARRIVE_SHARDS + 1)
#define CTL_ARRIVE_STRIDE 32      /* uint32 per line */
__device__ __forceinline__ bool ctl_arrive_last(unsigned* base, unsigned block_linear, unsigned nblocks, int* flag_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's partial-row stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned shard = block_linear % CTL_ARRIVE_SHARDS;
        const unsigned in_shard = (nblocks - shard + CTL_ARRIVE_SHARDS - 1) / CTL_ARRIVE_SHARDS;
        const unsigned nshards = nblocks < CTL_ARRIVE_SHARDS ? nblocks : CTL_ARRIVE_SHARDS;
        unsigned* sc = base + (1 + shard) * CTL_ARRIVE_STRIDE;
        int last = 0;
        if (__hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
            __hip_atomic_store(sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(base, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nshards - 1) {
                __hip_atomic_store(base, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        *flag_lds = last;
    }
    __syncthreads();
    if (!*flag_lds) return false;
#if CTL_ARRIVE_ACQUIRE
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    return true;
}
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

// BatchNorm forward finalize of one channel from its two sums over `count` pixels (bn_finalize_kernel and the fused tail of the
// convolution kernels): biased variance for the normalisation, unbiased for running_var, momentum update in place.
struct ctl_bn_fin_dev {
    const float* gamma; const float* beta; float* running_mean; float* running_var; int64_t* nbt;
    float* scale; float* shift; float* save_mean; float* save_invstd;
    double count; float eps, momentum; int update_running;
    // role 0: the producer's last block finalises (ctl_bn_finalize_tail).  role 1 / 2: CONSUMER-side finalize -- the kernel that first
    // uses the coefficients (1: as its prologue, 2: as its residual affine) computes them in its first few blocks from the producer's
    // statistics rows `partial` ([groups][rows][2][c]) while the other blocks wait at the record's counter (ctl_bn_consume)
    int role; int rows; const float* partial;
};
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
                                             float* __restrict__ save_mean, float* __restrict__ save_invstd) {
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
        p.rm = (1.f - momentum) * p.rm + momentum * (float)mean;
        p.rv = (1.f - momentum) * p.rv + momentum * (float)unbiased;
        running_mean[ch] = p.rm;
        running_var[ch] = p.rv;
    }
}
// Consumer-side finalize (role 1 / 2).  The first min(nblocks, c) blocks of the launch are the writers: a block per channel sums the
// producer's rows with all 256 threads (one round trip, double), writes scale / shift (write-through: the other XCDs' blocks read them
// in this launch), mean / invstd, the running statistics, and arrives at the record's counter; the LAST writer raises the go flags.
// Every block then waits for a go flag and reads the coefficients with agent-scope loads.  The writers are the first blocks of the grid and wait for
// nobody: no deadlock whatever is resident.  The wait overlaps the waiting blocks' first-tile loads (requested before the call); the
// stand-alone finalize launch and its kernel boundary disappear.  Polling: ~1000 blocks polling ONE line starve the writers' arrival
// adds (measured: +24 us per launch), so there are CTL_GO_LINES flag lines (block L polls line L % CTL_GO_LINES, with a sleep between
// polls).  `lines` = the record's counter lines: line 0 the arrival counter, lines 1.. the go flags; all zero at launch
// (ctl_bn_fin_table_write zeroes them every plan run).  `sm`: LDS scratch, 16 doubles, free at the call.
// The kernel side of this path is compiled in with -DCTL_CONSUMER_FINALIZE=1 only (tools/build_variant.sh): its mere presence in the
// convolution kernels cost the DEFAULT path 1.9 % of the fp32 step (same-box A/B against the commit before it, 18.00 -> 18.34 ms:
// a few more spilled SGPRs and a longer prologue in every instantiation), for a variant that measured slower anyway.
#ifndef CTL_CONSUMER_FINALIZE
#define CTL_CONSUMER_FINALIZE 0
#endif
#define CTL_GO_LINES 32
__device__ __forceinline__ void ctl_bn_consume(const ctl_bn_fin_dev& f, unsigned* lines, int groups, int c, unsigned lin_block, unsigned nblocks,
                                               double* sm) {
    const unsigned nwr = nblocks < (unsigned)c ? nblocks : (unsigned)c;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (lin_block < nwr) {
        for (int ch = (int)lin_block; ch < c; ch += (int)nwr) {
            ctl_bn_chan p = {};
            if (tid == 0) p = ctl_bn_chan_load(ch, f.gamma, f.beta, f.update_running, f.running_mean, f.running_var);
            for (int g = 0; g < groups; ++g) {
                const float* base = f.partial + ((int64_t)g * f.rows * 2) * c + ch;
                double s1 = 0.0, s2 = 0.0;
                for (int r0 = tid; r0 < f.rows; r0 += 256 * 4) {
                    float v1[4], v2[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + u * 256;
                        const bool ok = r < f.rows;
                        v1[u] = ok ? base[(int64_t)r * 2 * c] : 0.f;
                        v2[u] = ok ? base[((int64_t)r * 2 + 1) * c] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
                }
                s1 = wave_sum_double(s1);
                s2 = wave_sum_double(s2);
                if (lane == 0) { sm[wave] = s1; sm[4 + wave] = s2; }
                __syncthreads();
                if (tid == 0) {
                    s1 = (sm[0] + sm[1]) + (sm[2] + sm[3]);
                    s2 = (sm[4] + sm[5]) + (sm[6] + sm[7]);
                    const double mean = s1 / f.count;
                    double var = s2 / f.count - mean * mean;
                    if (var < 0.0) var = 0.0;
                    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
                    const float sc = p.gamma * invstd;
                    ctl_store_wt(f.scale + g * c + ch, sc);
                    ctl_store_wt(f.shift + g * c + ch, p.beta - (float)mean * sc);
                    if (f.save_mean) f.save_mean[g * c + ch] = (float)mean;
                    if (f.save_invstd) f.save_invstd[g * c + ch] = invstd;
                    if (f.update_running) {
                        const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
                        p.rm = (1.f - f.momentum) * p.rm + f.momentum * (float)mean;
                        p.rv = (1.f - f.momentum) * p.rv + f.momentum * (float)unbiased;
                        f.running_mean[ch] = p.rm;
                        f.running_var[ch] = p.rv;
                    }
                }
                __syncthreads();
            }
        }
        if (lin_block == 0 && tid == 0 && f.update_running && f.nbt) f.nbt[0] += groups;
        if (tid < 64) {       // (only thread 0 stored: its wave drains, releases and arrives; the last writer raises the flags)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int last = 0;
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                last = __hip_atomic_fetch_add(lines, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwr - 1;
            }
            last = __shfl(last, 0);
            if (last && tid < CTL_GO_LINES) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                __hip_atomic_store(lines + (1 + tid) * CTL_ARRIVE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (tid == 0) {
        const unsigned* flag = lines + (1 + lin_block % CTL_GO_LINES) * CTL_ARRIVE_STRIDE;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(16);
    }
    __syncthreads();
    // NO acquire fence here: an agent-scope acquire invalidates the XCD's L2 for every block of the launch (measured: +20 us per launch).
    // The callers read the coefficients with agent-scope loads (ctl_load_wt) into LDS instead.
}

// Device-resident record of one fused finalize (written by ctl_bn_fin_table_write before the producing kernel runs): the finalize
// arguments and the arrival counters (one sharded set per block row of output-channel tiles; zero between launches).  The convolution kernels
// take ONE pointer to it: passing the fields as kernel arguments cost the tuned kernels 50-60 spilled SGPRs.
struct alignas(128) ctl_bn_rec {
    ctl_bn_fin_dev f;                                                      // (first line)
    alignas(128) unsigned counters[CTL_FIN_MAX_Y][CTL_ARRIVE_LINES][CTL_ARRIVE_STRIDE];  // per block row of output-channel tiles: sharded arrival counters
};
static_assert(sizeof(ctl_bn_rec) == CTL_FIN_REC_BYTES, "ctl_bn_rec is one table slot");
__device__ __forceinline__ void ctl_bn_finalize_tail(ctl_bn_rec* __restrict__ rec, const float* __restrict__ partial, int rows, int groups,
                                                     int cout, int co_first, int nco, unsigned nblocks, int* flag_lds) {
#if CTL_CONSUMER_FINALIZE
    if (rec->f.role != 0) return;             // (a consumer-side record: nothing to do behind the tiles)
#endif
    if (!ctl_arrive_last(&rec->counters[blockIdx.y][0][0], blockIdx.z * gridDim.x + blockIdx.x, nblocks, flag_lds)) return;
    const ctl_bn_fin_dev f = rec->f;
    const int lane = threadIdx.x & 63;
    for (int cl = threadIdx.x >> 6; cl < nco; cl += (int)(blockDim.x >> 6)) {
        const int ch = co_first + cl;
        if (ch >= cout) continue;
        ctl_bn_chan p = ctl_bn_chan_load(ch, f.gamma, f.beta, f.update_running, f.running_mean, f.running_var);
        for (int g = 0; g < groups; ++g) {      // the running statistics see the groups in order, as consecutive forward calls would
            double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
            for (int b = lane; b < rows; b += 64) {
                s1 += (double)ctl_load_wt(partial + (((int64_t)g * rows + b) * 2 + 0) * cout + ch);
                s2 += (double)ctl_load_wt(partial + (((int64_t)g * rows + b) * 2 + 1) * cout + ch);
            }
            s1 = wave_sum_double(s1);
            s2 = wave_sum_double(s2);
            if (lane == 0)
                ctl_bn_coefs(s1, s2, f.count, cout, g, ch, p, f.eps, f.momentum, f.update_running, f.running_mean, f.running_var, f.scale,
                             f.shift, f.save_mean, f.save_invstd);
        }
    }
    if (f.update_running && f.nbt && co_first == 0 && threadIdx.x == 0) f.nbt[0] += groups;
}

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
                          float* stats_partial, ctl_bn_rec* rec, ctl_stream stream);
int ctl_conv_bf16_stats_blocks(const ctl_conv* d);
int ctl_wgrad_bf16_splits(const ctl_conv* d);
int ctl_conv_wgrad_bf16(const ctl_conv* d, const void* x, const float* pro_scale, const float* pro_shift, const void* dy,
                        float* w_partial, float* b_partial, ctl_stream stream);

// in-process profiling (ctl_plan.cpp): returns a token >= 0 if this launch is being timed
int ctl_prof_begin(const char* kind, const ctl_conv* d, const ctl_conv_cfg* c, int nt, hipStream_t stream);
void ctl_prof_end(int token, hipStream_t stream);
