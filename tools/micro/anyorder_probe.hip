// Does hipExtAnyOrderLaunch clear the AQL barrier bit on gfx950 (ROCm 7)?  Two independent ~50 us kernels of 64 blocks in ONE stream:
// serialized = ~2x, overlapped = ~1x.  Build: hipcc --offload-arch=gfx950 -O2 tools/micro/anyorder_probe.hip -o /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(unsigned long long cycles, int* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    int* d;
    hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned long long cyc = 5000;       // 100 MHz wall clock: 50 us
    for (int mode = 0; mode < 3; ++mode) {      // 0: plain <<<>>>, 1: hipExtLaunchKernelGGL flags 0, 2: any-order on every second launch
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(s);
            hipEventRecord(e0, s);
            for (int i = 0; i < 10; ++i) {
                if (mode == 0) {
                    spin<<<64, 64, 0, s>>>(cyc, d);
                    spin<<<64, 64, 0, s>>>(cyc, d + 1);
                } else {
                    hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, 0, cyc, d);
                    hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0, cyc, d + 1);
                }
            }
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d rep %d: 20 x 50 us kernels in %.1f us (%s)\n", mode, rep, ms * 1e3, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
