// Micro-test: per-channel statistics finalised by a second launch vs by the last block of the producing launch (ticket + fences).
// hipcc --offload-arch=gfx950 -O3 -o lastblock lastblock.hip ; ./lastblock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int C = 16;

template <bool COH>
__device__ __forceinline__ float4 ldp4(const float4* p) {
    if (!COH) return *p;
    float4 v;       // agent-scope (sc1) load: served past the non-coherent per-XCD L2
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <bool COH = false>
__device__ __forceinline__ void finalize_rows(const float* partial, int blocks, float* coef) {
    // 256 threads, all loads in flight at once: thread t -> float4 column t % 8 of rows t / 8 + 32 k (k < 24)
    __shared__ double sm[32][32];
    const int c4 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    float4 v[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) v[k] = (r0 + 32 * k < blocks) ? ldp4<COH>(reinterpret_cast<const float4*>(partial) + (r0 + 32 * k) * 8 + c4) : float4{0, 0, 0, 0};
    if (COH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int k = 0; k < 24; ++k) { s0 += (double)v[k].x; s1 += (double)v[k].y; s2 += (double)v[k].z; s3 += (double)v[k].w; }
    sm[r0][c4 * 4 + 0] = s0; sm[r0][c4 * 4 + 1] = s1; sm[r0][c4 * 4 + 2] = s2; sm[r0][c4 * 4 + 3] = s3;
    __syncthreads();
    if (threadIdx.x < C) {
        double s1 = 0, s2 = 0;
        for (int k = 0; k < 32; ++k) { s1 += sm[k][threadIdx.x]; s2 += sm[k][16 + threadIdx.x]; }
        const double mean = s1 / 1048576.0, var = s2 / 1048576.0 - mean * mean;
        coef[threadIdx.x] = (float)(1.0 / sqrt(var + 1e-5));
        coef[16 + threadIdx.x] = (float)mean;
    }
}
// the shape of the library's finalize launch: one block per channel, every thread a few rows of both statistics
__global__ __launch_bounds__(256) void finalize_per_channel(const float* partial, int blocks, float* coef) {
    __shared__ double sm[2][4];
    const int ch = blockIdx.x;
    double s1 = 0, s2 = 0;
    float a[3], b[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { const int r = threadIdx.x + 256 * k; a[k] = r < blocks ? partial[r * 32 + ch] : 0.f; b[k] = r < blocks ? partial[r * 32 + 16 + ch] : 0.f; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { s1 += (double)a[k]; s2 += (double)b[k]; }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s1; sm[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3]; s2 = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
        const double mean = s1 / 1048576.0, var = s2 / 1048576.0 - mean * mean;
        coef[ch] = (float)(1.0 / sqrt(var + 1e-5));
        coef[16 + ch] = (float)mean;
    }
}

template <int FUSED>
__global__ __launch_bounds__(256) void producer(float4* __restrict__ y, long quads, float* __restrict__ partial, float* __restrict__ coef, unsigned* counter, float seed) {
    const long per = (quads + gridDim.x - 1) / gridDim.x;
    const long lo = per * blockIdx.x, hi = min(quads, lo + per);
    float s1 = 0.f, s2 = 0.f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = seed + (float)(i & 1023) * 1e-3f;
        y[i] = float4{v, v, v, v};
        s1 += v; s2 += v * v;
    }
    // block partial row (crudely: lanes 0..31 of wave 0 write the row)
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    __shared__ float w[8];
    __shared__ unsigned tk;
    if ((threadIdx.x & 63) == 0) { w[threadIdx.x >> 6] = s1; w[4 + (threadIdx.x >> 6)] = s2; }
    __syncthreads();
    const float rowv = (threadIdx.x < 16 ? w[0] + w[1] + w[2] + w[3] : w[4] + w[5] + w[6] + w[7]);
    if (FUSED == 2) {          // relaxed agent-scope atomics for the row and the ticket: no L2 write-back / invalidate
        if (threadIdx.x < 32) __hip_atomic_store(partial + blockIdx.x * 32 + threadIdx.x, rowv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) tk = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tk == gridDim.x - 1) {
            finalize_rows<true>(partial, gridDim.x, coef);
            if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (threadIdx.x < 32) partial[blockIdx.x * 32 + threadIdx.x] = rowv;
    if (FUSED == 1) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) tk = atomicAdd(counter, 1u);
        __syncthreads();
        if (tk == gridDim.x - 1) {
            __threadfence();
            finalize_rows(partial, gridDim.x, coef);
            if (threadIdx.x == 0) *counter = 0;
        }
    }
}
__global__ __launch_bounds__(256) void finalize(const float* partial, int blocks, float* coef) { finalize_rows(partial, blocks, coef); }
// the consumer of the coefficients: a small dependent kernel (stands for the next conv's first loads)
__global__ void consumer(const float* coef, float* out) { if (threadIdx.x < 32) out[threadIdx.x] += coef[threadIdx.x]; }

int main() {
    const int blocks = 768;
    for (long mb : {1L, 8L, 67L}) {
        const long quads = mb * 1000000 / 16;
        float4* y; float *partial, *coef, *out; unsigned* counter;
        CK(hipMalloc(&y, quads * 16)); CK(hipMalloc(&partial, blocks * 32 * 4)); CK(hipMalloc(&coef, 32 * 4)); CK(hipMalloc(&out, 32 * 4)); CK(hipMalloc(&counter, 4));
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(out, 0, 128));
        hipStream_t s; CK(hipStreamCreate(&s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 200;
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9;
            for (int trial = 0; trial < 5; ++trial) {
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < reps; ++r) {
                    if (mode == 0) { producer<0><<<blocks, 256, 0, s>>>(y, quads, partial, coef, counter, (float)r); finalize_per_channel<<<16, 256, 0, s>>>(partial, blocks, coef); }
                    else if (mode == 1) producer<1><<<blocks, 256, 0, s>>>(y, quads, partial, coef, counter, (float)r);
                    else if (mode == 3) producer<2><<<blocks, 256, 0, s>>>(y, quads, partial, coef, counter, (float)r);
                    else producer<0><<<blocks, 256, 0, s>>>(y, quads, partial, coef, counter, (float)r);
                    consumer<<<1, 64, 0, s>>>(coef, out);
                }
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<float> h(32); CK(hipMemcpy(h.data(), coef, 128, hipMemcpyDeviceToHost));
            printf("%3ld MB written per launch, %s: %7.2f us per (producer%s, consumer)   coef[0]=%g coef[16]=%g\n", mb,
                   mode == 0 ? "separate finalize launch" : mode == 1 ? "last block, threadfence " : mode == 3 ? "last block, relaxed atom" : "no finalize at all      ", 1e3 * best / reps, mode == 0 ? ", finalize" : "", h[0], h[16]);
        }
        CK(hipFree(y)); CK(hipFree(partial)); CK(hipFree(coef)); CK(hipFree(out)); CK(hipFree(counter));
    }
    return 0;
}
