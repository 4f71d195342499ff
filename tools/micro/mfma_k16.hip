// Micro-test: issue rate of v_mfma_f32_16x16x16_bf16 against v_mfma_f32_16x16x32_bf16 on gfx950 (is the K = 16 form half the cycles?).
// hipcc --offload-arch=gfx950 -O3 -o mfma_k16 mfma_k16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int K32>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    bf8 a8, b8; s4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3f80 + threadIdx.x); b4[i] = (short)(0x3f80 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // inline asm: with the builtins the compiler rotated the accumulators through AGPRs (dozens of v_accvgpr moves per iteration)
            if (K32) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
            else asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
        }
    }
    f4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}
int main() {
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int k32 = 0; k32 < 2; ++k32) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (k32) k<1><<<256, 256>>>(d, iters); else k<0><<<256, 256>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // one wave per SIMD (256 blocks x 4 waves on 256 CUs x 4 SIMDs): cycles per MFMA at ~2.1 GHz
            if (rep) printf("%s: %.3f ms for %d x 8 MFMAs per wave -> %.2f ns per MFMA (%.1f cycles at 2.1 GHz), %.0f TFLOP/s\n", k32 ? "16x16x32_bf16" : "16x16x16_bf16",
                            ms, iters, 1e6 * ms / (iters * 8.0), 2.1 * 1e6 * ms / (iters * 8.0), (k32 ? 16384.0 : 8192.0) * iters * 8 * 1024 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
