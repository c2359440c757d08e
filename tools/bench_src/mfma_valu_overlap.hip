// micro-benchmark: do MFMA and VALU work overlap on a CDNA4 SIMD?
//   one workgroup of 8 waves per CU (2 waves per SIMD, like the attention kernels).  Per wave and iteration: NM independent
//   v_mfma_f32_32x32x16_f16 (4 accumulators round-robin) and NV VALU instructions (full-rate FMAs, or v_exp_f32).
//   mode 0: every wave runs its MFMAs, then its VALU block (one stream, phases back to back)
//   mode 1: every wave interleaves them (one VALU group behind each MFMA)
//   mode 2: waves 0-3 run ONLY MFMAs (2 NM), waves 4-7 run ONLY VALU (2 NV): the two waves of a SIMD are different kinds
//   mode 3: MFMA only (NM per wave, all waves); mode 4: VALU only (NV per wave, all waves)
// If the pipes overlap, modes 1 / 2 take max(mode 3, mode 4); if they serialise, the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int NM, int NV, int KIND, int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384 + 4];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = 0.001f * i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(((threadIdx.x * 7 + e * 3) % 13) * 0.01f); b[e] = (_Float16)(((threadIdx.x + e) % 7) * 0.02f); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
    auto mf = [&](int i) { acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i % NACC], 0, 0, 0); };
    auto va = [&](int i) {
        if (KIND == 1) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
        else if (KIND == 2) {                 // 16-byte LDS read per lane (1 KB per wave), address varies with i: conflict-free rows
            const float4 q = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + (i & 7) * 2048) & 16383));
            v[i & 7] += q.x + q.w;
        } else v[i & 7] = __builtin_fmaf(v[i & 7], 0.999f, 0.0001f);
    };
    const bool mfma_wave = MODE != 2 || wave < 4, valu_wave = MODE != 2 || wave >= 4;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NM; ++i) mf(i);
#pragma unroll
            for (int i = 0; i < NV; ++i) va(i);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                mf(i);
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) va(i * (NV / NM) + j);
            }
        } else if (MODE == 2) {
            if (mfma_wave) {
#pragma unroll
                for (int i = 0; i < 2 * NM; ++i) mf(i);
            }
            if (valu_wave) {
#pragma unroll
                for (int i = 0; i < 2 * NV; ++i) va(i);
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < NM; ++i) mf(i);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) va(i);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE, int NM, int NV, int KIND, int NACC>
float run(float* d, int iters, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, NM, NV, KIND, NACC>), dim3(grid), dim3(512), 0, 0, d, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE, NM, NV, KIND, NACC>), dim3(grid), dim3(512), 0, 0, d, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f;
}

template <int NM, int NV, int KIND, int NACC>
void sweep(float* d, int iters, const char* name, int grid) {
    const float t3 = run<3, NM, NV, KIND, NACC>(d, iters, grid), t4 = run<4, NM, NV, KIND, NACC>(d, iters, grid);
    const float t0 = run<0, NM, NV, KIND, NACC>(d, iters, grid), t1 = run<1, NM, NV, KIND, NACC>(d, iters, grid), t2 = run<2, NM, NV, KIND, NACC>(d, iters, grid);
    printf("%-34s MFMA only %7.1f us  VALU only %7.1f us  | phases back to back %7.1f  interleaved %7.1f  split by wave %7.1f   (sum %7.1f, max %7.1f)\n",
           name, t3, t4, t0, t1, t2, t3 + t4, t3 > t4 ? t3 : t4);
}

int main() {
    float* d; CK(hipMalloc(&d, 4096));
    const int iters = 2000;
    // grid = workgroups = CUs in use: 256 = the whole chip (socket power cap in play), 16 = a fraction of it (full clock)
    for (int grid : {256, 16}) {
        printf("---- %d workgroups (one per CU); accumulators the MFMAs rotate over: 4 / 2 / 1 (1 = one dependent chain)\n", grid);
        sweep<12, 48, 0, 4>(d, iters, "12 MFMA (4 acc) + 48 FMA", grid);
        sweep<12, 48, 0, 2>(d, iters, "12 MFMA (2 acc) + 48 FMA", grid);
        sweep<12, 48, 0, 1>(d, iters, "12 MFMA (1 acc) + 48 FMA", grid);
        sweep<12, 24, 1, 4>(d, iters, "12 MFMA (4 acc) + 24 v_exp_f32", grid);
        sweep<12, 24, 1, 1>(d, iters, "12 MFMA (1 acc) + 24 v_exp_f32", grid);
        sweep<12, 12, 2, 4>(d, iters, "12 MFMA (4 acc) + 12 ds_read_b128", grid);
        sweep<12, 24, 2, 4>(d, iters, "12 MFMA (4 acc) + 24 ds_read_b128", grid);
    }
    return 0;
}
