// probe: do v_mfma_f32_32x32x16_f16, v_dot2_f32_f16 and v_fma_mix_f32 keep fp16 SUBNORMAL inputs (probabilities under 2^-14 of their
// reference) or flush them to zero?  hipcc --offload-arch=gfx950 -O2 -o denorm_probe denorm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float x, float* out) {
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)x; b[e] = (_Float16)1.f; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    const half2v p = {(_Float16)x, (_Float16)x}, ones = {(_Float16)1.f, (_Float16)1.f};
    const float d = __builtin_amdgcn_fdot2(p, ones, 0.f, false);
    float m = 0.f;
    const unsigned pw = __builtin_bit_cast(unsigned, p);
    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(m) : "v"(pw));
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d; out[2] = m; out[3] = (float)(_Float16)x; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    for (float x : {1.0f, 6.1035156e-5f /* 2^-14: smallest normal */, 3.0517578e-5f /* 2^-15 */, 9.5367432e-7f /* 2^-20 */, 5.9604645e-8f /* 2^-24 */}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, d);
        float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("x = %-12g fp16(x) = %-12g | mfma 32x32x16 (16 terms of x * 1): %-12g (expected %g) | dot2 (2 terms): %-12g (expected %g) | fma_mix: %g\n",
               x, h[3], h[0], 16.0 * h[3], h[1], 2.0 * h[3], h[2]);
    }
    return 0;
}
