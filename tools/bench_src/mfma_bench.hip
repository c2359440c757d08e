// micro-benchmark: v_mfma_f32_32x32x16_f16 issue rate vs dependency structure and waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, int iters) {
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 12 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, int iters) {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
    const int threads = 256 * waves_per_simd;          // 4 SIMDs x waves_per_simd waves
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * 4 * waves_per_simd * iters * 12;
    const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
    printf("NACC=%d waves/SIMD=%d: %.3f ms, %.0f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", NACC, waves_per_simd, ms, tf,
           ms * 1e-3 * 2.4e9 / (iters * 12.0 * waves_per_simd));
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; ++w) { run<1>(w, 20000); run<2>(w, 20000); run<4>(w, 20000); run<12>(w, 20000); }
    return 0;
}
