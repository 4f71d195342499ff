// Micro-test: v_dot2c_f32_bf16 with the (-1, 0) / (0, -1) selector as a compile-time constant (the compiler may encode it as an inline
// constant) vs as an opaque register value.   hipcc --offload-arch=gfx950 -O3 -o dot2sel dot2sel.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, unsigned sel_lo, unsigned sel_hi) {
    const unsigned p = 0x40404000u;        // packed pair: low = 0x4000 = 2.0, high = 0x4040 = 3.0
    const bf2 pp = __builtin_bit_cast(bf2, p);
    const bf2 c0 = {(__bf16)-1.0f, (__bf16)0.0f}, c1 = {(__bf16)0.0f, (__bf16)-1.0f};
    out[0] = __builtin_amdgcn_fdot2_f32_bf16(pp, c0, 10.f, false);                                  // want 10 - 2 = 8
    out[1] = __builtin_amdgcn_fdot2_f32_bf16(pp, c1, 10.f, false);                                  // want 10 - 3 = 7
    out[2] = __builtin_amdgcn_fdot2_f32_bf16(pp, __builtin_bit_cast(bf2, sel_lo), 10.f, false);     // want 8
    out[3] = __builtin_amdgcn_fdot2_f32_bf16(pp, __builtin_bit_cast(bf2, sel_hi), 10.f, false);     // want 7
    // exactness probe: x - rne_bf16(x) for a value with a full mantissa
    const float x = 1.2345678f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{x, x}, bf2));
    out[4] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, h), __builtin_bit_cast(bf2, sel_lo), x, false);
    out[5] = x - __builtin_bit_cast(float, h << 16);
}
int main() {
    float* d; hipMalloc(&d, 64);
    k<<<1, 1>>>(d, 0x0000BF80u, 0xBF800000u);
    float h[6]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("constant (-1,0): %g (want 8)   constant (0,-1): %g (want 7)   register (-1,0): %g   register (0,-1): %g\n", h[0], h[1], h[2], h[3]);
    printf("x - rne_bf16(x): dot2 %.9g  subtract %.9g  %s\n", h[4], h[5], h[4] == h[5] ? "equal" : "DIFFERENT");
    return 0;
}
