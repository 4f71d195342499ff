// What does each instruction type of the X3 staging cost when it is issued by the PARTNER wave of an MFMA wave on the same SIMD?  (round 5)
// 512-thread blocks, one per CU: waves 0-3 loop over v_mfma_f32_32x32x16_bf16 (32 cycles each alone), waves 4-7 loop over 16 instructions of ONE type.
// Printed: cycles per iteration of both waves, each alone and together.  A type that shares hardware with the matrix pipe shows up as the sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int TYPE>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, int iters, int role) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x16 acc = {};
    float v[16];
    f32x2 p[8];
    unsigned u[16];
    u32x4 q = {1, 2, 3, 4};
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 0.5f + i; u[i] = threadIdx.x * 77u + i; }
    for (int i = 0; i < 8; ++i) p[i] = f32x2{v[2 * i], v[2 * i + 1]};
    const float m = 1.0001f, c = 0.5f;
    const f32x2 pm = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    unsigned sel;
    asm volatile("s_mov_b32 %0, 0x0000bf80" : "=s"(sel));
    unsigned char* lp = lds + (wave & 3) * 16384 + lane * 16;
    const unsigned laddr = (unsigned)(size_t)lp;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (role & 1)
            for (int it = 0; it < iters; ++it) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    } else if (role & 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (TYPE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
                else if (TYPE == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[j]) : "v"(v[j]), "v"(v[(j + 1) & 15]));
                else if (TYPE == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[j]) : "s"(sel), "v"(u[j]));
                else if (TYPE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 7]) : "v"(pm), "v"(pc));
                else if (TYPE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(pm));
                else if (TYPE == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
                else if (TYPE == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 15]));
                else if (TYPE == 7) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
                else if (TYPE == 8) { if (j < 4) asm volatile("ds_write_b128 %0, %1" :: "v"(laddr + j * 1024), "v"(q) : "memory"); }
                else if (TYPE == 9) { if (j < 4) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(laddr + j * 1024) : "memory"); }
                else if (TYPE == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 15]));
                else if (TYPE == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 15]));
            }
            if (TYPE == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + (float)u[i] + acc[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    s += (float)q.x;
    if (s == 123.456f) out[1023] = 1;
    if (lane == 0 && blockIdx.x == 0) out[1 + wave] = t1 - t0;
}
template <int TYPE>
static void run(unsigned long long* d, const char* name, int per_iter) {
    const int iters = 20000;
    unsigned long long h[9];
    double r[4][2];
    for (int role = 1; role <= 3; ++role) {
        for (int rep = 0; rep < 3; ++rep) k<TYPE><<<256, 512>>>(d, iters, role);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 72, hipMemcpyDeviceToHost);
        r[role][0] = (double)h[1] / iters; r[role][1] = (double)h[5] / iters;
    }
    printf("%-22s x%2d per iteration: MFMA wave alone %5.1f, partner alone %6.1f (%5.2f each) | together: MFMA wave %5.1f  partner %6.1f (%5.2f each)  -> partner +%.0f %%, MFMA +%.0f %%\n", name,
           per_iter, r[1][0], r[2][1], r[2][1] / per_iter, r[3][0], r[3][1], r[3][1] / per_iter, 100.0 * (r[3][1] / r[2][1] - 1.0), 100.0 * (r[3][0] / r[1][0] - 1.0));
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 8192);
    hipMemset(d, 0, 8192);
    run<0>(d, "v_fma_f32", 16); run<1>(d, "v_cvt_pk_bf16_f32", 16); run<2>(d, "v_dot2c_f32_bf16", 16); run<3>(d, "v_pk_fma_f32", 16); run<4>(d, "v_pk_mul_f32", 16);
    run<5>(d, "v_max_f32", 16); run<6>(d, "v_and_b32", 16); run<7>(d, "v_sub_f32", 16); run<10>(d, "v_mov_b32", 16); run<11>(d, "v_cndmask_b32", 16);
    run<8>(d, "ds_write_b128", 4); run<9>(d, "ds_read_b128 + wait", 4);
    return 0;
}
