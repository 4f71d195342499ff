// Register-only MFMA f32 16x16x4 throughput probe: WAVES waves per SIMD, NACC independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu, int iters) {
    float* out; hipMalloc(&out, 256 * 256 * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 4 * NACC * 2048.0;
    printf("NACC=%d waves/SIMD=%d: %.1f TFLOP/s (%.3f ms)\n", NACC, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; ++w) { run<1>(w, 20000); run<2>(w, 10000); run<4>(w, 5000); run<8>(w, 2500); }
    return 0;
}
