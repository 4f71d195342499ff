// Does a wave's VALU / VMEM / LDS instruction stream slow down while ANOTHER wave on the same SIMD streams MFMAs?
// One 512-thread block per CU: waves 0-3 (one per SIMD) run an f32 16x16x4 MFMA stream, waves 4-7 (the second wave of each
// SIMD) run a "worker" stream of WORK instructions and time it with s_memtime.  Reported: worker cycles alone, worker cycles
// under MFMA, MFMA cycles alone, MFMA cycles with worker.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: independent v_fma chains (8 accumulators); 1: dependent v_fma chain; 2: ds_read_b128 + fma; 3: global loads (L2 hits)
template <int KIND>
__global__ __launch_bounds__(512) void k(const float* in, float* out, unsigned long long* cyc, int mfma_iters, int work_iters,
                                          int run_mfma, int run_work, int prio) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 512) lds[i] = in[i];
    __syncthreads();
    if (wave < 4) {
        if (!run_mfma) return;
        f32x4 w = *reinterpret_cast<const f32x4*>(in + lane * 4), x = *reinterpret_cast<const f32x4*>(in + 1024 + lane * 4);
        f32x4 acc[4];
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0, 0, 0, 0};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc[m], 0, 0, 0);
                }
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int m = 0; m < 4; ++m) s += acc[m].x + acc[m].y + acc[m].z + acc[m].w;
        out[blockIdx.x * 512 + tid] = s;
        if (lane == 0) cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
    } else {
        if (!run_work) return;
        if (prio == 1) __builtin_amdgcn_s_setprio(3);
        if (prio == 2) __builtin_amdgcn_s_setprio(1);
        float a[8];
        for (int i = 0; i < 8; ++i) a[i] = in[lane + i * 64];
        const float b = in[2048 + lane], c = in[2100 + lane];
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < work_iters; ++it) {
            if (KIND == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], b, c);
            } else if (KIND == 1) {
#pragma unroll
                for (int r = 0; r < 64; ++r) a[0] = __builtin_fmaf(a[0], b, c);
            } else if (KIND == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((r * 64 + lane + it) * 4) % 4096);
                    a[r & 7] += v.x + v.w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((r * 64 + lane + it * 7) * 4) % 4096);
                    a[r & 7] += v.x + v.w;
                }
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int i = 0; i < 8; ++i) s += a[i];
        out[blockIdx.x * 512 + tid] = s;
        if (lane == 0) cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
    }
}

template <int KIND> void run(const char* name, int instr_per_iter, float* in, float* out, unsigned long long* cyc, int prio) {
    const int grid = 256, mi = 2000, wi = 400;
    unsigned long long h[256 * 16];
    double res[3][2];
    for (int cfg = 0; cfg < 3; ++cfg) {   // 0: worker alone, 1: mfma alone, 2: both
        const int rm = cfg != 0, rw = cfg != 1;
        (void)hipMemset(cyc, 0, sizeof(h));
        k<KIND><<<grid, 512>>>(in, out, cyc, mi, wi, rm, rw, prio);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, w = 0;
        for (int b = 0; b < grid; ++b) for (int wv = 0; wv < 8; ++wv) (wv < 4 ? m : w) += (double)h[(b * 8 + wv) * 2];
        res[cfg][0] = m / (grid * 4) / (mi * 32.0);           // cycles per MFMA
        res[cfg][1] = w / (grid * 4) / ((double)wi * instr_per_iter);  // cycles per worker instruction
    }
    printf("%-28s worker cyc/instr alone %.2f  under MFMA %.2f | MFMA cyc/instr alone %.2f  with worker %.2f\n", name, res[0][1],
           res[2][1], res[1][0], res[2][0]);
}
int main() {
    float *in, *out; unsigned long long* cyc;
    (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 2000.f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int prio = 0; prio < 3; ++prio) {
        printf("-- worker s_setprio %d\n", prio == 0 ? 0 : (prio == 1 ? 3 : 1));
        run<0>("independent v_fma x8", 64, in, out, cyc, prio);
        run<1>("dependent v_fma chain", 64, in, out, cyc, prio);
        run<2>("ds_read_b128 + 2 add", 16, in, out, cyc, prio);
        run<3>("global_load_b128 + 2 add", 16, in, out, cyc, prio);
    }
    return 0;
}
