// Does vector work hide behind matrix work on a gfx950 SIMD?  (round 5: the question behind every X3 kernel -- their staging is ~5 vector
// instructions per MFMA.)  Two experiments, cycles by s_memtime, 256 blocks (one per CU):
//   A  ONE wave per SIMD: a loop of { 1 MFMA (dependent chain), N independent v_fma_f32 }: cycles per iteration vs N
//   B  TWO waves per SIMD (512-thread blocks): waves 0-3 run the MFMA loop, waves 4-7 a v_fma loop of the same length: each alone, then together
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip ; run: ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, int SHAPE>      // SHAPE 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(256, 1) void k_same_wave(unsigned long long* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x16 acc = {};
    f32x4 acc4 = {};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 0.5f + i;
    const float m = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 12]) : "v"(m), "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 4; ++i) s += acc4[i];
    if (s == 123.456f) out[1023] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// ROLE mask: bit 0 = the MFMA waves (0-3) work, bit 1 = the VALU waves (4-7) work; NV v_fma per "iteration" of the VALU waves
template <int SHAPE, int NV>
__global__ __launch_bounds__(512, 2) void k_two_waves(unsigned long long* out, int iters, int role) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x16 acc = {};
    f32x4 acc4 = {};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 0.5f + i;
    const float m = 1.0001f, c = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (role & 1)
            for (int it = 0; it < iters; ++it) {
                if (SHAPE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4) : "v"(a), "v"(b));
            }
    } else {
        if (role & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 12]) : "v"(m), "v"(c));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 4; ++i) s += acc4[i];
    if (s == 123.456f) out[1023] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[1 + wave] = t1 - t0;
}

// A': the same with ACCS independent accumulators in rotation (the MFMA chain is no longer dependent): { MFMA acc[k] ; N v_fma } for k = 0 .. ACCS-1
template <int N, int ACCS>
__global__ __launch_bounds__(256, 1) void k_same_wave_indep(unsigned long long* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x16 acc[ACCS];
    for (int k = 0; k < ACCS; ++k) acc[k] = f32x16{};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 0.5f + i;
    const float m = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ACCS; ++k) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[(k * N + j) % 12]) : "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int k = 0; k < ACCS; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (s == 123.456f) out[1023] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    (void)m;
}
template <int N, int ACCS>
static void run_a2(unsigned long long* d, int iters) {
    unsigned long long h = 0;
    for (int r = 0; r < 3; ++r) k_same_wave_indep<N, ACCS><<<256, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("A' 32x32x16  %d independent accumulators, N=%2d v_sub per MFMA, one wave per SIMD: %.1f cycles per MFMA\n", ACCS, N, (double)h / iters / ACCS);
}
template <int N, int SHAPE>
static void run_a(unsigned long long* d, int iters) {
    unsigned long long h = 0;
    for (int r = 0; r < 3; ++r) k_same_wave<N, SHAPE><<<256, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("A  %s  N=%2d v_fma per MFMA, one wave per SIMD: %.1f cycles per iteration\n", SHAPE ? "16x16x32" : "32x32x16", N, (double)h / iters);
}
template <int SHAPE, int NV>
static void run_b(unsigned long long* d, int iters) {
    unsigned long long h[9];
    for (int role = 1; role <= 3; ++role) {
        for (int r = 0; r < 3; ++r) k_two_waves<SHAPE, NV><<<256, 512>>>(d, iters, role);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 72, hipMemcpyDeviceToHost);
        printf("B  %s NV=%2d role %s: MFMA wave %.1f cycles/iter, VALU wave %.1f cycles/iter (%.2f per v_fma)\n", SHAPE ? "16x16x32" : "32x32x16", NV,
               role == 1 ? "MFMA only " : role == 2 ? "VALU only " : "both      ", (double)h[1] / iters, (double)h[5] / iters, (double)h[5] / iters / NV);
    }
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 8192);
    hipMemset(d, 0, 8192);
    const int iters = 20000;
    run_a<0, 0>(d, iters); run_a<2, 0>(d, iters); run_a<4, 0>(d, iters); run_a<6, 0>(d, iters); run_a<8, 0>(d, iters); run_a<12, 0>(d, iters); run_a<16, 0>(d, iters);
    run_a<0, 1>(d, iters); run_a<2, 1>(d, iters); run_a<4, 1>(d, iters); run_a<6, 1>(d, iters); run_a<8, 1>(d, iters);
    run_a2<0, 2>(d, iters); run_a2<2, 2>(d, iters); run_a2<4, 2>(d, iters); run_a2<6, 2>(d, iters); run_a2<8, 2>(d, iters);
    run_a2<0, 4>(d, iters); run_a2<2, 4>(d, iters); run_a2<4, 4>(d, iters); run_a2<5, 4>(d, iters); run_a2<6, 4>(d, iters); run_a2<8, 4>(d, iters);
    run_b<0, 4>(d, iters); run_b<0, 8>(d, iters); run_b<0, 16>(d, iters);
    run_b<1, 2>(d, iters); run_b<1, 4>(d, iters); run_b<1, 8>(d, iters);
    return 0;
}
