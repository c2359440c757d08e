// Weight / bias gradient of the hoisted K/V projection (training backward):
//
//     dW_kv[n][k] += sum_m g[m][n] * tokens[m][k]          db_kv[n] += sum_m g[m][n]
//
// with m running over ALL tokens of the scene batch (768 000 rows at BASELINE cfg 4 per GPU), n over the 2C = 512 K|V columns and
// k over the C = 256 token channels: 201 GFLOP whose output is 0.5 MB.  The generic row-split fp32 kernel (gemm_tn_kernel,
// backward.hip: one 4-byte load per lane per MFMA operand) runs it at 45 TFLOP/s; here the same contraction goes through the
// fp16 matrix pipe with the hi/lo split used by the forward projection (three 32x32x16 MFMAs per product, fp32 accumulation,
// fp32-class result) and every token / gradient element is fetched and converted once per workgroup.
//
// Both operands are row-major [m][col] and the contraction runs over m, so both MFMA fragments (8 consecutive m for one column)
// are transposes of the memory layout.  The transpose happens in the staging pass: thread = one column, it loads 32 consecutive
// rows of its column (the 64 lanes of a wave read 256 contiguous bytes of each row), splits them and writes the LDS image
// [column][32 m] as 8-byte pieces; fragments are then plain 16-byte reads.
//
//   workgroup = 512 threads = 256 columns of g (one of the two n-slabs) + the 256 token channels, 32 rows per step,
//               double-buffered LDS (2 x 64 KB), one barrier per step, the next step's rows in flight in registers;
//   wave (wn, wk) of the 4 x 2 wave grid owns the 64 x 128 block of the slab's 256 x 256 outputs: 2 x 4 accumulators;
//   grid      = 2 slabs x S row ranges (S = CUs / 2); partial results are added to the gradient arena with float atomics
//               (as the row-split generic kernel does); the bias gradient falls out of the staging pass (column sums of g).
//
// g is multiplied by a power of two read from `scale` (max |g| -> ~2^10, measured by the attention backward's epilogue) so that
// the low halves of small gradient values stay clear of the fp16 subnormals; the result is multiplied back.
#include "common.hpp"

namespace parq {

namespace {

constexpr int kTM = 32;                          // rows per step
constexpr int kCols = 256;                       // columns of each operand per workgroup
constexpr int kImgHalfs = kCols * kTM;           // one operand image (hi or lo), 16 KB
constexpr int kBufHalfs = 4 * kImgHalfs;         // A_hi | A_lo | B_hi | B_lo

struct KvBwdArgs {
    const float* g; int64_t ldg;                 // [M][2C] gradient of K | V (token-major)
    const float* x; int64_t ldx;                 // [M][C] tokens
    float* dW; int64_t ldw;                      // [2C][C], +=
    float* db;                                   // [2C], +=
    const float* scale;                          // power of two applied to g
    int M, nsplit;
};

// position (in 16-bit units) of rows [8 q, 8 q + 8) of column c inside an operand image: 64 bytes per column, the 16-byte chunk
// index q is XOR-ed with bits 2..3 of the column so that the 16 lanes a ds_read_b128 serves per cycle hit 16 different slots
__device__ __forceinline__ int img_off(int c, int q) { return c * kTM + ((q ^ ((c >> 2) & 3)) << 3); }

__global__ __launch_bounds__(512, 1) void kvproj_bwd_split_kernel(KvBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];          // [2 buffers][A_hi | A_lo | B_hi | B_lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int slab = blockIdx.x & 1, split = blockIdx.x >> 1;
    const float sc = *a.scale, inv_sc = 1.f / sc;

    // rows of this workgroup: a contiguous range, multiple of 32 except for the last one
    const int steps_total = (a.M + kTM - 1) / kTM;
    const int s_begin = (int)((int64_t)split * steps_total / a.nsplit);
    const int s_end = (int)((int64_t)(split + 1) * steps_total / a.nsplit);

    // staging: thread -> one column (threads 0..255: g column slab*256 + tid, threads 256..511: token channel tid - 256)
    const bool isA = tid < kCols;
    const int col = isA ? tid : tid - kCols;
    const float* src = isA ? a.g + slab * kCols + col : a.x + col;
    const int64_t ld = isA ? a.ldg : a.ldx;
    const float mul = isA ? sc : 1.f;
    float stg[kTM];
    float csum = 0.f;
    auto fetch = [&](int step) {
        const int m0 = step * kTM;
        if (m0 + kTM <= a.M) {
#pragma unroll
            for (int r = 0; r < kTM; ++r) stg[r] = src[(int64_t)(m0 + r) * ld];
        } else {
#pragma unroll
            for (int r = 0; r < kTM; ++r) stg[r] = m0 + r < a.M ? src[(int64_t)(m0 + r) * ld] : 0.f;
        }
    };
    auto stage = [&](int buf) {
        _Float16* hi = lds + buf * kBufHalfs + (isA ? 0 : 2 * kImgHalfs);
        _Float16* lo = hi + kImgHalfs;
        typedef _Float16 half4v __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int p = 0; p < kTM / 4; ++p) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                csum += stg[4 * p + e];
                v[e] = stg[4 * p + e] * mul;
            }
            half2v h0, l0, h1, l1;
            split_pair(v[0], v[1], h0, l0);
            split_pair(v[2], v[3], h1, l1);
            const int off = img_off(col, p >> 1) + (p & 1) * 4;
            *reinterpret_cast<half4v*>(hi + off) = half4v{h0[0], h0[1], h1[0], h1[1]};
            *reinterpret_cast<half4v*>(lo + off) = half4v{l0[0], l0[1], l1[0], l1[1]};
        }
    };

    const int wn = wave >> 1, wk = wave & 1;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (s_begin < s_end) {
        fetch(s_begin);
        stage(0);
    }
    __syncthreads();
    for (int step = s_begin; step < s_end; ++step) {
        const int buf = (step - s_begin) & 1;
        const bool more = step + 1 < s_end;
        if (more) fetch(step + 1);
        const _Float16* Ah = lds + buf * kBufHalfs;
        const _Float16* Al = Ah + kImgHalfs;
        const _Float16* Bh = Ah + 2 * kImgHalfs;
        const _Float16* Bl = Ah + 3 * kImgHalfs;
#pragma unroll
        for (int s = 0; s < 2; ++s) {              // two 16-row MFMA steps
            const int q = 2 * s + kh;
            half8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = wn * 64 + i * 32 + li;
                ah[i] = *reinterpret_cast<const half8*>(Ah + img_off(c, q));
                al[i] = *reinterpret_cast<const half8*>(Al + img_off(c, q));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = wk * 128 + j * 32 + li;
                bh[j] = *reinterpret_cast<const half8*>(Bh + img_off(c, q));
                bl[j] = *reinterpret_cast<const half8*>(Bl + img_off(c, q));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) stage(buf ^ 1);                  // the other buffer was last read in step - 1, closed by that step's barrier
        __syncthreads();
    }

    // outputs: accumulator register r of lane (li, kh) is row n = mfma32_row(r, lane), column k = li of its 32 x 32 block
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* o = a.dW + (int64_t)(slab * kCols + wn * 64 + i * 32) * a.ldw + wk * 128 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(o + (int64_t)mfma32_row(r, lane) * a.ldw, acc[i][j][r] * inv_sc);
        }
    if (isA && a.db) atomicAdd(a.db + slab * kCols + col, csum);
}

// scale[0] = 2^(10 - exponent(max)) from the float bit pattern in bits[0] (1 when the maximum is 0 or not finite)
__global__ void pow2_scale_kernel(const unsigned int* __restrict__ bits, float* __restrict__ scale) {
    const float mx = __uint_as_float(bits[0]);
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) frexpf(mx, &ex);
    scale[0] = ldexpf(1.f, 10 - ex);
}

}  // namespace

bool kvproj_bwd_split_supported(int C) { return C == kCols; }

// g [M][2C], tokens [M][C] -> dW [2C][C] (+=), db [2C] (+=).  absmax_bits: device word holding the bit pattern of max |g|
// (non-negative floats order like unsigned integers); scale_scratch: one device float.
// The same contraction for any [M][512-column slice] operand: out[n][k] += sum_m g[m][n] x[m][k], n < 512, k < 256; g rows ldg apart
// (the head-dim-256 attention backward forms dQ^T slices this way: g = dS^T, x = K).  db may be null.
hipError_t launch_tn_split_512x256(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t M, float* out, int64_t ldo,
                                   float* db, const unsigned int* absmax_bits, float* scale_scratch, hipStream_t s) {
    if (M < 1 || M > (int64_t)INT32_MAX) return hipErrorInvalidValue;
    static DynLdsOnce once;
    const size_t ldsb = (size_t)2 * kBufHalfs * sizeof(_Float16);          // 128 KB
    if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&kvproj_bwd_split_kernel), ldsb); e != hipSuccess) return e;
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, s, absmax_bits, scale_scratch);
    KvBwdArgs a;
    a.g = g; a.ldg = ldg; a.x = x; a.ldx = ldx; a.dW = out; a.ldw = ldo; a.db = db; a.scale = scale_scratch; a.M = (int)M;
    const int steps = (int)((M + kTM - 1) / kTM);
    int nsplit = device_num_cus() / 2;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > steps) nsplit = steps;
    a.nsplit = nsplit;
    hipLaunchKernelGGL(kvproj_bwd_split_kernel, dim3(2 * nsplit), dim3(512), ldsb, s, a);
    return hipGetLastError();
}

hipError_t launch_kvproj_bwd_split(const float* g, const float* tokens, int64_t M, int C, float* dW, float* db,
                                   const unsigned int* absmax_bits, float* scale_scratch, hipStream_t s) {
    if (C != kCols || M < 1 || M > (int64_t)INT32_MAX) return hipErrorInvalidValue;
    static DynLdsOnce once;
    const size_t ldsb = (size_t)2 * kBufHalfs * sizeof(_Float16);          // 128 KB
    if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&kvproj_bwd_split_kernel), ldsb); e != hipSuccess) return e;
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, s, absmax_bits, scale_scratch);
    KvBwdArgs a;
    a.g = g; a.ldg = 2 * C; a.x = tokens; a.ldx = C; a.dW = dW; a.ldw = C; a.db = db; a.scale = scale_scratch; a.M = (int)M;
    const int steps = (int)((M + kTM - 1) / kTM);
    int nsplit = device_num_cus() / 2;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > steps) nsplit = steps;
    a.nsplit = nsplit;
    hipLaunchKernelGGL(kvproj_bwd_split_kernel, dim3(2 * nsplit), dim3(512), ldsb, s, a);
    return hipGetLastError();
}

}  // namespace parq
