// Micro-test: sustained rate of v_mfma_f32_16x16x32_bf16 with 1, 2 and 4 waves per SIMD on every CU, and the shader clock the chip holds
// under that load (s_memtime cycles against the 100 MHz wall clock).   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk, int random_data) {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    bf8 a8, b8;
    for (int i = 0; i < 8; ++i) {
        unsigned h = (threadIdx.x * 8 + i + blockIdx.x * 2048) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        a8[i] = random_data ? (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f)) : (__bf16)0.0f;
        b8[i] = random_data ? (__bf16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f)) : (__bf16)0.0f;
    }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));      // (the builtin form
        // made the compiler rotate the accumulators through AGPRs: ~50 v_accvgpr moves per 8 MFMAs, which is what that loop then measured)
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    f4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int random_data = 0; random_data < 2; ++random_data)
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int iters = 400000 / waves, blocks = 256 * waves;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k<<<blocks, 256>>>(d, iters, clk, random_data);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            if (rep == 2) printf("%s operands, %d wave(s) per SIMD: %.3f ms, %.0f TFLOP/s dense bf16; s_memtime/wall = %.0f MHz; %.1f shader cycles per MFMA and wave\n", random_data ? "random" : "all-zero", waves, ms,
                            16384.0 * iters * 8 * 4 * blocks / (ms * 1e-3) / 1e12, 100.0 * (double)h[0] / (double)h[1], (double)h[0] / (iters * 8.0));
        }
    }
    return 0;
}
