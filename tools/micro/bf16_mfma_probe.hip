// Does the bf16 matrix pipe share the VALU issue port the way fp32 MFMA does (tools/micro/valu_in_mfma.hip)?  Per MFMA the wave
// issues K independent v_fma_f32; reported: cycles per MFMA with 1 and 2 waves per SIMD, for
//   f32 16x16x4 (2048 flop), bf16 16x16x16 (8192 flop), bf16 16x16x32 (16384 flop, gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int K>
__global__ __launch_bounds__(512) void k(const float* in, float* out, unsigned long long* cyc, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 w = *reinterpret_cast<const f32x4*>(in + lane * 4), x = *reinterpret_cast<const f32x4*>(in + 1024 + lane * 4);
    s16x4 ws = *reinterpret_cast<const s16x4*>(in + lane * 2), xs = *reinterpret_cast<const s16x4*>(in + 512 + lane * 2);
    bf16x8 w8 = *reinterpret_cast<const bf16x8*>(in + lane * 4), x8 = *reinterpret_cast<const bf16x8*>(in + 1024 + lane * 4);
    f32x4 acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0, 0, 0, 0};
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = in[lane + i * 64];
    const float b = in[2048 + lane], c = in[2100 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == 0) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc[r & 3], 0, 0, 0);
            if (KIND == 1) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ws, xs, acc[r & 3], 0, 0, 0);
            if (KIND == 2) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w8, x8, acc[r & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < K; ++j) a[(r * K + j) & 7] = __builtin_fmaf(a[(r * K + j) & 7], b, c);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m].x + acc[m].y + acc[m].z + acc[m].w;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int K> void run(const char* name, float* in, float* out, unsigned long long* cyc) {
    const int grid = 256, iters = 1000;
    unsigned long long h[256 * 8];
    double res[2];
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int threads = cfg ? 512 : 256;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        k<KIND, K><<<grid, threads>>>(in, out, cyc, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0;
        for (int b = 0; b < grid; ++b) for (int wv = 0; wv < threads / 64; ++wv) m += (double)h[b * 8 + wv];
        res[cfg] = m / (grid * threads / 64) / (iters * 8.0);
        if (cfg) printf("   (%d waves/SIMD kernel: %.1f us for %d MFMAs per wave)\n", threads / 256, ms * 1e3, iters * 8);
    }
    printf("%-14s K=%2d VALU per MFMA: 1 wave/SIMD %.1f, 2 waves/SIMD %.1f memtime ticks per MFMA per wave\n", name, K, res[0], res[1]);
}
int main() {
    float *in, *out; unsigned long long* cyc;
    (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 2000.f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0>("f32 16x16x4", in, out, cyc); run<0, 4>("f32 16x16x4", in, out, cyc); run<0, 8>("f32 16x16x4", in, out, cyc);
    run<1, 0>("bf16 16x16x16", in, out, cyc); run<1, 2>("bf16 16x16x16", in, out, cyc); run<1, 4>("bf16 16x16x16", in, out, cyc); run<1, 8>("bf16 16x16x16", in, out, cyc);
    run<2, 0>("bf16 16x16x32", in, out, cyc); run<2, 2>("bf16 16x16x32", in, out, cyc); run<2, 4>("bf16 16x16x32", in, out, cyc); run<2, 8>("bf16 16x16x32", in, out, cyc);
    return 0;
}
