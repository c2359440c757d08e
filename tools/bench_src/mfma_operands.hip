// micro-benchmark: does the issue rate of v_mfma_f32_32x32x16_f16 depend on WHICH registers hold its A / B operands?
//   mode 0: the same A and B registers for every MFMA            (tools/bench_src/mfma_valu_overlap.hip measured this: 32 cycles)
//   mode 1: four A and four B register quads in rotation         (tools/bench_src/mx_energy.hip measured this: 50.9 cycles)
//   mode 2: A rotates, B fixed;  mode 3: A fixed, B rotates
//   mode 4: as 1, the second accumulator's MFMA uses the SAME operands as the first one's (what an attention step does: one K
//           fragment against hi and lo of Q)
// 2 accumulators alternate in every mode; grid = `wgs` workgroups of 8 waves; operands zero or random bits.
// hipcc --offload-arch=gfx950 -O3 -o mfma_operands mfma_operands.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ src, float* out, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = __builtin_bit_cast(half8, src[(size_t)t * 8 + i]); b[i] = __builtin_bit_cast(half8, src[(size_t)t * 8 + 4 + i]); }
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int ia = (MODE == 0 || MODE == 3) ? 0 : (i & 3), ib = (MODE == 0 || MODE == 2) ? 0 : ((i + (i >> 2)) & 3);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                // mode 5: the pattern of tools/bench_src/mx_energy.hip — consecutive MFMAs share A, then B, then A ...
                const int ja = MODE == 4 || MODE == 5 ? ia : (MODE == 1 ? ((ia + blk) & 3) : ia);
                const int jb = MODE == 5 ? ((i + (i >> 2) + blk) & 3) : MODE == 4 ? ib : (MODE == 1 ? ((ib + 2 * blk) & 3) : ib);
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ja], b[jb], acc[blk], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
}

int main(int argc, char** argv) {
    const int iters = 4000;
    uint4* src; float* out;
    const size_t n = 256 * 512;
    CK(hipMalloc(&src, n * 8 * sizeof(uint4))); CK(hipMalloc(&out, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int data = 0; data < 2; ++data) {
        if (data == 0) CK(hipMemset(src, 0, n * 8 * sizeof(uint4)));
        else {
            unsigned short* h = (unsigned short*)malloc(n * 8 * 16);
            srand(1);
            for (size_t i = 0; i < n * 64; ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15));     // |x| in [0.125, 0.5)
            CK(hipMemcpy(src, h, n * 8 * 16, hipMemcpyHostToDevice));
            free(h);
        }
        for (int wgs : {16, 256}) {
            printf("%s operands, %3d workgroups:", data ? "random" : "zero  ", wgs);
            for (int mode = 0; mode < 6; ++mode) {
                auto launch = [&]() {
                    switch (mode) {
                        case 0: hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(512), 0, 0, src, out, iters); break;
                        case 1: hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(512), 0, 0, src, out, iters); break;
                        case 2: hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(512), 0, 0, src, out, iters); break;
                        case 3: hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(512), 0, 0, src, out, iters); break;
                        case 4: hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(512), 0, 0, src, out, iters); break;
                        default: hipLaunchKernelGGL(k<5>, dim3(wgs), dim3(512), 0, 0, src, out, iters); break;
                    }
                };
                for (int w = 0; w < 30; ++w) launch();                    // ~0.1 s of load: the power controller settles
                CK(hipEventRecord(e0));
                for (int w = 0; w < 20; ++w) launch();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double ns_per_mfma_simd = ms / 20 * 1e6 / (iters * 24.0 * 2);      // 2 waves per SIMD
                printf("  mode %d %.2f ns (%.1f cyc @2.4)", mode, ns_per_mfma_simd, ns_per_mfma_simd * 2.4);
            }
            printf("\n");
        }
    }
    // time course: sustained load, the rate every ~0.25 s (is the rate a function of the data, or of what ran before?)
    for (int phase = 0; phase < 3; ++phase) {
        const int data = phase == 1;
        if (!data) CK(hipMemset(src, 0, n * 8 * sizeof(uint4)));
        else {
            unsigned short* h = (unsigned short*)malloc(n * 8 * 16);
            srand(1);
            for (size_t i = 0; i < n * 64; ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15));
            CK(hipMemcpy(src, h, n * 8 * 16, hipMemcpyHostToDevice));
            free(h);
        }
        printf("sustained, 256 workgroups, %s operands (cycles @2.4 GHz per MFMA and SIMD, every 100 launches):", data ? "random" : "zero");
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0));
            for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, src, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %.1f", ms / 100 * 1e6 / (iters * 24.0 * 2) * 2.4);
        }
        printf("\n");
    }
    return 0;
}
