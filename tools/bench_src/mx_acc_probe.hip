// probe: how precisely does v_mfma_scale_f32_32x32x64_f8f6f4 add a SMALL sum of products to a LARGE accumulator?
//   every product is (1.0 * scale_small) so the exact sum is 64 * 2^-e; C = big.  Prints D - C against the exact increment.
// hipcc --offload-arch=gfx950 -O2 -o mx_acc_probe mx_acc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__global__ void k(float cbig, int scale_a, float* out, int f16mode) {
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) { a[w] = 0x38383838; b[w] = 0x38383838; }       // e4m3 0x38 = 1.0
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = cbig;
    if (f16mode) {
        half8 ha, hb;
        for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)ldexpf(1.f, scale_a - 127); hb[e] = (_Float16)1.f; }
        for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c, 0, 0, 0);
    } else {
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, 127);
    }
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    for (int f16mode = 0; f16mode < 2; ++f16mode)
        for (float cbig : {0.f, 1.f, 60.f, 1000.f}) {
            printf("%s C = %-6g:", f16mode ? "fp16 x4 " : "MX fp8  ", cbig);
            for (int e = 0; e <= 24; e += 2) {
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cbig, 127 - e, d, f16mode);
                float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
                const double inc = 64.0 * ldexp(1.0, -e);
                printf("  2^-%d: %.4f", e, ((double)h - cbig) / inc);
            }
            printf("\n");
        }
    return 0;
}
