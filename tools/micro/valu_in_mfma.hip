// Can a wave hide its own VALU / LDS / VMEM work between its MFMAs?  Per f32 16x16x4 MFMA (32 cycles of matrix pipe) the
// wave issues K independent instructions of a kind; reported: cycles per MFMA with 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int K>
__global__ __launch_bounds__(512) void k(const float* in, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    f32x4 w = *reinterpret_cast<const f32x4*>(in + lane * 4), x = *reinterpret_cast<const f32x4*>(in + 1024 + lane * 4);
    f32x4 acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0, 0, 0, 0};
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = in[lane + i * 64];
    const float b = in[2048 + lane], c = in[2100 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc[r & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (KIND == 0) a[(r * K + j) & 7] = __builtin_fmaf(a[(r * K + j) & 7], b, c);
                if (KIND == 1) { const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((r * 64 + j * 640 + lane + it) * 4) % 4096); a[j & 7] += v.x; }
                if (KIND == 2) { const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((r * 64 + j * 640 + lane + it * 7) * 4) % 4096); a[j & 7] += v.x; }
            }
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
        k<KIND, K><<<grid, threads>>>(in, out, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0;
        for (int b = 0; b < grid; ++b) for (int wv = 0; wv < threads / 64; ++wv) m += (double)h[b * 8 + wv];
        res[cfg] = m / (grid * threads / 64) / (iters * 8.0);
    }
    printf("%-14s K=%d per MFMA: 1 wave/SIMD %.1f cyc/MFMA, 2 waves/SIMD %.1f cyc/MFMA per wave (ideal 32 / 64)\n", name, K, res[0], res[1]);
}
int main() {
    float *in, *out; unsigned long long* cyc;
    (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 2000.f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0>("v_fma", in, out, cyc); run<0, 1>("v_fma", in, out, cyc); run<0, 2>("v_fma", in, out, cyc); run<0, 4>("v_fma", in, out, cyc);
    run<0, 6>("v_fma", in, out, cyc); run<0, 7>("v_fma", in, out, cyc); run<0, 8>("v_fma", in, out, cyc); run<0, 12>("v_fma", in, out, cyc);
    run<1, 1>("ds_read_b128", in, out, cyc); run<1, 2>("ds_read_b128", in, out, cyc);
    run<2, 1>("global_b128", in, out, cyc); run<2, 2>("global_b128", in, out, cyc);
    return 0;
}
