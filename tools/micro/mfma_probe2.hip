// MFMA f32 16x16x4 issue-rate probe with the operand pattern of the conv loop: 9 taps x 4 M-tiles x 4 k-slices,
// A from 36 distinct registers (weights), B from 16 distinct registers (inputs), 4 accumulators.
// MODE 0: operands in registers only; MODE 1: B re-read from LDS per tap (ds_read_b128, as the conv does); MODE 2: A and B from LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x4 w[9], x[4];
    for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const f32x4*>(in + t * 256 + lane * 4);
    for (int m = 0; m < 4; ++m) x[m] = *reinterpret_cast<const f32x4*>(in + 4096 + m * 256 + lane * 4);
    f32x4 acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f32x4 wf = w[t];
            if (MODE == 2) wf = *reinterpret_cast<const f32x4*>(lds + 4096 + ((t * 64 + lane) * 4 + (it & 3) * 16) % 4096);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 xf = x[m];
                if (MODE >= 1) xf = *reinterpret_cast<const f32x4*>(lds + ((t * 5 + m * 37 + (it & 7)) * 64 + lane * 4) % 4096);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.x, xf.x, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.y, xf.y, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.z, xf.z, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.w, xf.w, acc[m], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m].x + acc[m].y + acc[m].z + acc[m].w;
    out[blockIdx.x * 256 + tid] = s;
}
template <int MODE> void run(int bpc, int iters, float* in, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int grid = 256 * bpc;
    k<MODE><<<grid, 256>>>(in, out, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<grid, 256>>>(in, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 144.0 * 2048.0;
    printf("MODE=%d waves/SIMD=%d: %.1f TFLOP/s (%.3f ms)\n", MODE, bpc, flops / ms / 1e9, ms);
}
int main() {
    float *in, *out; (void)hipMalloc(&in, 8192 * 4); (void)hipMalloc(&out, 256 * 256 * 8 * 4);
    float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; ++w) { run<0>(w, 400, in, out); run<1>(w, 400, in, out); run<2>(w, 400, in, out); }
    return 0;
}
