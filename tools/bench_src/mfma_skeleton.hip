// micro-benchmark: the dependency skeleton of the attention stage (QK chain -> max -> cvt -> PV) vs pure MFMA rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(float* out, int iters, float thr) {
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(((threadIdx.x * 7 + e * 3) % 13) * 0.01f); b[e] = (_Float16)(((threadIdx.x + e) % 7) * 0.02f); }
    f32x16 o0, o1;
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = 0.f, l = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s;
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, s, 0, 0, 0);
            half8 p0 = a, p1 = b;
            if (MODE >= 1) {            // max over the accumulator + cross-half shuffle + (rare) branch
                float mx = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (__any(mx > m_run + thr)) { m_run = mx; for (int r = 0; r < 16; ++r) { o0[r] *= 0.5f; o1[r] *= 0.5f; } }
            }
            if (MODE >= 2) {            // P operands derived from the accumulator (exp + split)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float x0 = __builtin_amdgcn_exp2f(s[e] - m_run), x1 = __builtin_amdgcn_exp2f(s[e + 1] - m_run);
                    l += x0 + x1;
                    half2v h = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
                    p0[e] = h[0]; p0[e + 1] = h[1];
                    half2v g = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]));
                    p1[e] = g[0]; p1[e + 1] = g[1];
                }
            } else if (MODE >= 1) {
                p0[0] = (_Float16)s[3];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, (i & 1) ? p1 : p0, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, (i & 1) ? p1 : p0, o1, 0, 0, 0);
            }
        }
    }
    float sum = l + m_run;
    for (int r = 0; r < 16; ++r) sum += o0[r] + o1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int MODE>
void run(int waves_per_simd, int iters) {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
    const int threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 10, 1e30f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 1e30f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * 4 * waves_per_simd * iters * 48;
    printf("MODE=%d waves/SIMD=%d: %.3f ms, %.0f TFLOP/s issued, %.1f cycles/MFMA/SIMD @2.4GHz\n", MODE, waves_per_simd, ms,
           mfma * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 48.0 * waves_per_simd));
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 3; ++w) { run<0>(w, 4000); run<1>(w, 4000); run<2>(w, 4000); }
    return 0;
}
