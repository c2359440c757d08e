// micro-benchmark: what do the cross terms of a split-precision product cost on the MX-scaled fp8 matrix instruction?
//   per wave and "block" (one 32 x 32 output block over a contraction length of 64, the cross-attention's S block):
//   mode 0: 12 x v_mfma_f32_32x32x16_f16                      (hi.hi + hi.lo + lo.hi, the shipped split product)
//   mode 1:  4 x v_mfma_f32_32x32x16_f16 + 2 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands: the cross terms in 8 bits)
//   mode 2:  4 x f16 only (the single-product modes);   mode 3: 6 x MX only;   mode 4: 8 x f16 (two terms)
//   operands: random bits from memory (`a`, `b`: 64 B per lane), or zeros (zero != 0), so the data-dependent power shows.
// One workgroup of 8 waves per CU slot, 2 accumulators per wave (two blocks in flight, like the pipelined kernel).
// Built as a shared library and driven by tools/mx_energy.py (hip events + sysfs clock / power under load).
#include <hip/hip_runtime.h>
#include <cstdint>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void mx_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ a8p,
                                                 const uint4* __restrict__ b8p, float* out, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    uint4 ra[4], rb[4], qa[4], qb[4];
    for (int i = 0; i < 4; ++i) { ra[i] = a[(size_t)t * 4 + i]; rb[i] = b[(size_t)t * 4 + i]; qa[i] = a8p[(size_t)t * 4 + i]; qb[i] = b8p[(size_t)t * 4 + i]; }
    half8 ah[4], bh[4];
    i32x8 a8[2], b8[2];
    for (int i = 0; i < 4; ++i) {
        ah[i] = __builtin_bit_cast(half8, ra[i]);
        bh[i] = __builtin_bit_cast(half8, rb[i]);
    }
    for (int i = 0; i < 2; ++i) {
        a8[i] = i32x8{(int)qa[2 * i].x, (int)qa[2 * i].y, (int)qa[2 * i].z, (int)qa[2 * i].w, (int)qa[2 * i + 1].x, (int)qa[2 * i + 1].y, (int)qa[2 * i + 1].z, (int)qa[2 * i + 1].w};
        b8[i] = i32x8{(int)qb[2 * i].x, (int)qb[2 * i].y, (int)qb[2 * i].z, (int)qb[2 * i].w, (int)qb[2 * i + 1].x, (int)qb[2 * i + 1].y, (int)qb[2 * i + 1].z, (int)qb[2 * i + 1].w};
    }
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int one = 127;                                            // E8M0 scale 2^0
    for (int it = 0; it < iters; ++it) {
        constexpr int NF = MODE == 0 ? 12 : MODE == 1 ? 4 : MODE == 2 ? 4 : MODE == 3 ? 0 : 8;
        constexpr int NX = MODE == 1 ? 2 : MODE == 3 ? 6 : 0;
        // the two blocks of a step alternate, so that no instruction waits for the accumulator of the one in front of it
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 3], bh[(i + (i >> 2) + blk) & 3], acc[blk], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
                acc[blk] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i & 1], b8[((i >> 1) + blk) & 1], acc[blk], 0, 0, 0, one, 0, one);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
}

extern "C" int mx_launch(int mode, int iters, const void* a, const void* b, const void* a8, const void* b8, void* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const uint4* A = (const uint4*)a; const uint4* Bp = (const uint4*)b; const uint4* A8 = (const uint4*)a8; const uint4* B8 = (const uint4*)b8; float* o = (float*)out;
    switch (mode) {
        case 0: hipLaunchKernelGGL(mx_kernel<0>, dim3(256), dim3(512), 0, s, A, Bp, A8, B8, o, iters); break;
        case 1: hipLaunchKernelGGL(mx_kernel<1>, dim3(256), dim3(512), 0, s, A, Bp, A8, B8, o, iters); break;
        case 2: hipLaunchKernelGGL(mx_kernel<2>, dim3(256), dim3(512), 0, s, A, Bp, A8, B8, o, iters); break;
        case 3: hipLaunchKernelGGL(mx_kernel<3>, dim3(256), dim3(512), 0, s, A, Bp, A8, B8, o, iters); break;
        case 4: hipLaunchKernelGGL(mx_kernel<4>, dim3(256), dim3(512), 0, s, A, Bp, A8, B8, o, iters); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
