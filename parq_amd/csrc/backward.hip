// Backward kernels of the decoder chain (training step of model/parq_lightning.py:97-100).
//
// The reference trains through PyTorch autograd; here every backward op of one recurrent iteration is a HIP
// kernel behind the C ABI (parq_backward).  GEMM-shaped pieces reuse the forward small-GEMM (linear.hip):
//     dX = dY W        -> launch_linear with a transposed copy of W (transpose_kernel, refreshed per backward call)
//     dW += dY^T X     -> gemm_tn_kernel below (contraction over the rows, exact fp32 MFMA, accumulates in place)
// and everything row-local or scene-local has its own kernel:
//     LayerNorm backward (+ gamma/beta gradients), GroupNorm(1,C)+ReLU backward in two passes (scene-wide sums in
//     float64), box-decode backward (sigmoid / exp / size gather), position-embedding backward (iteration 0 only:
//     later reference points are detached, transformer_parq.py:331-332), project+sample backward (bilinear scatter
//     into the token gradient with atomics, coordinate gradient for iteration 0), attention backwards.
#include "common.hpp"
#include <cstdint>

#include <cstring>

namespace parq {

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ transpose: dst[c][r] = src[r][c]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst,
                                                        int64_t ldd, int R, int Cc) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? src[(int64_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cc && r < R) dst[(int64_t)c * ldd + r] = tile[tx][i];
    }
}

// ------------------------------------------------------------------ out[N][K] (+)= A[M][N]^T B[M][K]
// v_mfma_f32_16x16x4_f32: lane (i = l&15, kq = l>>4); the contraction runs over the rows m, so lane (i, kq) reads
// A[m0 + kq][n0 + i] and B[m0 + kq][k0 + i]: 16 consecutive floats of a row per kq group (coalesced).  A workgroup of
// 4 waves owns a 32x32 output tile, the waves split the rows and reduce through LDS.
struct TnArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* out; int64_t ldo;
    int M, N, K;
    int accumulate;     // 0 overwrite, 1 add (the launch owns out), 2 add with atomics (launches on other streams add to out too)
    float* bias;        // optional: bias[n] (+)= sum_m A[m][n] for n >= bias_from (the bias gradient of the same layer), by the
    int bias_from;      // workgroups of the first k-tile from the A values they hold anyway (gemm_tn_kernel only)
};

// The same contraction for LARGE outputs with FEW rows (the C = 1024 layers of the reference's shipped size: out 1024 x 1024 ..
// 3072 x 1024 from 256 .. 1024 rows), where the kernel below (one 4-byte load per lane and MFMA operand, 32 x 32 tiles) runs at
// 5 TFLOP/s.  A workgroup owns a 64 x 64 output tile; 32 rows of both operands are staged per step as they lie in memory
// ([m][64 columns], coalesced 256-byte row pieces) — for a contraction over m that layout IS the operand layout of
// v_mfma_f32_32x32x2_f32 (lane = column, two consecutive m per instruction), so no transposes: 16 MFMAs per wave and step.
__global__ __launch_bounds__(256) void gemm_tn_tile64_kernel(TnArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][32][64 + 32];      // dY rows m, columns n0 .. n0 + 63 (96-float rows: the two m of
    __shared__ __attribute__((aligned(16))) float Bs[2][32][64 + 32];      // an instruction read disjoint halves of the 64 banks)
    const int ntk = (a.K + 63) / 64;
    const int k0 = (int)(blockIdx.x % ntk) * 64, n0 = (int)(blockIdx.x / ntk) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;                              // 32 x 32 quadrant of the tile
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // staging: thread -> (row tid >> 3 of the 32, 8 consecutive columns) of each operand
    const int sr = tid >> 3, sc = (tid & 7) * 8;
    float4 ra[2], rb[2];
    // 16-byte loads where base and row stride allow them (row slices of wider buffers need not be aligned)
    const bool avec = ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0) && (a.lda % 4 == 0);
    const bool bvec = ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0) && (a.ldb % 4 == 0);
    auto fetch = [&](int m0) {
        const int m = m0 + sr;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ra[e] = float4{0.f, 0.f, 0.f, 0.f};
            rb[e] = float4{0.f, 0.f, 0.f, 0.f};
        }
        if (m < a.M) {
            const float* ap = a.A + (int64_t)m * a.lda + n0 + sc;
            const float* bp = a.B + (int64_t)m * a.ldb + k0 + sc;
            if (n0 + sc + 8 <= a.N && avec) { ra[0] = *reinterpret_cast<const float4*>(ap); ra[1] = *reinterpret_cast<const float4*>(ap + 4); }
            else {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = n0 + sc + e < a.N ? ap[e] : 0.f;
                ra[0] = float4{t[0], t[1], t[2], t[3]}; ra[1] = float4{t[4], t[5], t[6], t[7]};
            }
            if (k0 + sc + 8 <= a.K && bvec) { rb[0] = *reinterpret_cast<const float4*>(bp); rb[1] = *reinterpret_cast<const float4*>(bp + 4); }
            else {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = k0 + sc + e < a.K ? bp[e] : 0.f;
                rb[0] = float4{t[0], t[1], t[2], t[3]}; rb[1] = float4{t[4], t[5], t[6], t[7]};
            }
        }
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<float4*>(&As[buf][sr][sc]) = ra[0];
        *reinterpret_cast<float4*>(&As[buf][sr][sc + 4]) = ra[1];
        *reinterpret_cast<float4*>(&Bs[buf][sr][sc]) = rb[0];
        *reinterpret_cast<float4*>(&Bs[buf][sr][sc + 4]) = rb[1];
    };
    const int steps = (a.M + 31) / 32;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        if (st + 1 < steps) fetch((st + 1) * 32);
#pragma unroll
        for (int mm = 0; mm < 32; mm += 2)          // lane (column li, kh) supplies row m = mm + kh of both operands
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][mm + kh][wn * 32 + li], Bs[buf][mm + kh][wk * 32 + li], acc, 0, 0, 0);
        if (st + 1 < steps) stage(buf ^ 1);         // the other buffer was last read in step st - 1, closed by that step's barrier
        __syncthreads();
    }
    // accumulator register r of lane (li, kh): row n = mfma32_row(r, lane) of the quadrant, column k = li
    const int k = k0 + wk * 32 + li;
    if (k < a.K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + mfma32_row(r, lane);
            if (n < a.N) {
                float* o = a.out + (int64_t)n * a.ldo + k;
                *o = a.accumulate ? *o + acc[r] : acc[r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * 4 * 64];
    __shared__ float cred[4][32];
    const int ntk = (a.K + 31) / 32;
    const int k0 = (int)(blockIdx.x % ntk) * 32, n0 = (int)(blockIdx.x / ntk) * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    f32x4v acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[s][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    bool nok[2], kok[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        nok[s] = n0 + s * 16 + li < a.N;
        kok[s] = k0 + s * 16 + li < a.K;
    }
    const float* Ap = a.A + n0 + li;
    const float* Bp = a.B + k0 + li;
    // rows: blockIdx.y owns a contiguous chunk; inside it groups of 4 rows are dealt round-robin to the 4 waves
    const int chunk = (((a.M + (int)gridDim.y - 1) / (int)gridDim.y) + 15) / 16 * 16;
    const int m_begin = (int)blockIdx.y * chunk;
    const int m_end = m_begin + chunk < a.M ? m_begin + chunk : a.M;
    float cs[2] = {0.f, 0.f};                     // column sums of this lane's A values (bias gradient)
    constexpr int kUp = 16;                       // row groups of a wave requested together (chunks of up to 256 rows)
    if (chunk <= 16 * kUp) {
        // short chunk (the row-split launches of the small dW GEMMs: up to 256 rows per workgroup): EVERY operand load of the wave
        // is requested before the first MFMA — the loop below costs one dependent memory round trip per 4 rows (a conditional load
        // per iteration; 34 us for a 1024-row chunk).  Out-of-range rows / columns read a valid
        // address and are zeroed by a select.
        float av[kUp][2], bv[kUp][2];
        const int ncl[2] = {nok[0] ? 0 : -li, nok[1] ? 16 : -li};        // column offsets clamped into the matrix
        const int kcl[2] = {kok[0] ? 0 : -li, kok[1] ? 16 : -li};
#pragma unroll
        for (int i = 0; i < kUp; ++i) {
            const int m = m_begin + wave * 4 + i * 16 + kq;
            const int mc = m < m_end ? m : a.M - 1;                       // any valid row: zeroed below
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                av[i][s] = Ap[(int64_t)mc * a.lda + ncl[s]];
                bv[i][s] = Bp[(int64_t)mc * a.ldb + kcl[s]];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < kUp; ++i) {
            const int m = m_begin + wave * 4 + i * 16 + kq;
            const bool mok = m < m_end;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                av[i][s] = (mok && nok[s]) ? av[i][s] : 0.f;
                bv[i][s] = (mok && kok[s]) ? bv[i][s] : 0.f;
                cs[s] += av[i][s];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s], bv[i][t], acc[s][t], 0, 0, 0);
        }
    } else
    for (int m0 = m_begin + wave * 4; m0 < m_end; m0 += 16) {
        const int m = m0 + kq;
        const bool mok = m < m_end;
        float av[2], bv[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            av[s] = (mok && nok[s]) ? Ap[(int64_t)m * a.lda + s * 16] : 0.f;
            bv[s] = (mok && kok[s]) ? Bp[(int64_t)m * a.ldb + s * 16] : 0.f;
            cs[s] += av[s];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[t], acc[s][t], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * 2 + s) * 2 + t) * 4 + r) * 64 + lane] = acc[s][t][r];
    const bool do_bias = a.bias != nullptr && k0 == 0;       // workgroup-uniform
    if (do_bias) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            cs[s] += __shfl_xor(cs[s], 16);
            cs[s] += __shfl_xor(cs[s], 32);
            if (kq == 0) cred[wave][s * 16 + li] = cs[s];
        }
    }
    __syncthreads();
    if (do_bias && tid < 32) {
        const int n = n0 + tid;
        if (n < a.N && n >= a.bias_from) {
            const float v = (cred[0][tid] + cred[1][tid]) + (cred[2][tid] + cred[3][tid]);
            if (gridDim.y > 1 || a.accumulate == 2) atomicAdd(a.bias + n, v);
            else a.bias[n] = a.accumulate ? a.bias[n] + v : v;
        }
    }
    // thread -> (row n, 4 consecutive k) of the 32x32 tile; sub-tile accumulator holds rows 4*(lane>>4)+r, column lane&15
    const int row = tid >> 3, c4 = (tid & 7) * 4;
    const int s = row >> 4, rr = row & 15, t = c4 >> 4, cc = c4 & 15;
    const int src = (((0 * 2 + s) * 2 + t) * 4 + (rr & 3)) * 64 + (rr >> 2) * 16 + cc;
    float sum[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sum[e] = red[src + e];
#pragma unroll
    for (int w = 1; w < 4; ++w)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += red[src + w * 2 * 2 * 4 * 64 + e];
    const int n = n0 + row;
    if (n < a.N) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + c4 + e;
            if (k < a.K) {
                float* o = a.out + (int64_t)n * a.ldo + k;
                if (gridDim.y > 1 || a.accumulate == 2) atomicAdd(o, sum[e]);   // row-split launch, or other streams add to the same out
                else *o = a.accumulate ? *o + sum[e] : sum[e];
            }
        }
    }
}

// ------------------------------------------------------------------ out[n] (+)= sum_m X[m][n]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ldx, int M, int N, float* __restrict__ out,
                                                     int accumulate) {
    __shared__ float part[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    const int chunk = (M + (int)gridDim.y - 1) / (int)gridDim.y;
    const int m_begin = (int)blockIdx.y * chunk;
    const int m_end = m_begin + chunk < M ? m_begin + chunk : M;
    float s = 0.f;
    if (n < N)
        for (int m = m_begin + w; m < m_end; m += 4) s += X[(int64_t)m * ldx + n];
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && n < N) {
        const float t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (gridDim.y > 1) atomicAdd(out + n, t);
        else out[n] = accumulate ? out[n] + t : t;
    }
}

// ------------------------------------------------------------------ y = a + b
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}
// y[m][c] (+)= x[m][c]
__global__ void axpy_rows_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int M, int N,
                                 int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int m = (int)(i / N), c = (int)(i - (int64_t)m * N);
    const float v = x[(int64_t)m * ldx + c];
    float* o = y + (int64_t)m * ldy + c;
    *o = accumulate ? *o + v : v;
}

// dst[m][n] = keep(m * N + n) ? src[m][n] / (1 - p) : 0   (gradient through a dropout site; also used to dump masks in tests)
__global__ void dropout_apply_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n_total, int N, float p, uint32_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const uint32_t row = (uint32_t)(i / N), col = (uint32_t)(i - (int64_t)row * N);
    dst[i] = drop_keep(drop_rowhash(seed, row), col, p) ? src[i] / (1.f - p) : 0.f;
}

// ------------------------------------------------------------------ LayerNorm backward, one wave per row
//   y = xhat * gamma + beta, xhat = (x - mean) * rstd   ->   gx = rstd (gxh - mean(gxh) - xhat mean(gxh xhat)), gxh = gy gamma
constexpr int kLnPer = 16;      // C <= 1024
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     float* __restrict__ gx, int M, int C, int accumulate) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const float mean = stats[(int64_t)row * 2], rstd = stats[(int64_t)row * 2 + 1];
    float gxh[kLnPer], xh[kLnPer];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kLnPer; ++i) {
        const int c = lane + i * 64;
        if (c < C) {
            xh[i] = (x[(int64_t)row * C + c] - mean) * rstd;
            gxh[i] = gy[(int64_t)row * C + c] * gamma[c];
            s1 += gxh[i];
            s2 += gxh[i] * xh[i];
        } else {
            xh[i] = 0.f;
            gxh[i] = 0.f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
    for (int i = 0; i < kLnPer; ++i) {
        const int c = lane + i * 64;
        if (c < C) {
            const float v = rstd * (gxh[i] - m1 - xh[i] * m2);
            float* o = gx + (int64_t)row * C + c;
            *o = accumulate ? *o + v : v;
        }
    }
}
// dgamma[c] += sum_m gy[m][c] xhat[m][c], dbeta[c] += sum_m gy[m][c]
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                            const float* __restrict__ stats, int M, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float p1[4][64], p2[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    const int chunk = (M + (int)gridDim.y - 1) / (int)gridDim.y;          // rows of this workgroup
    const int m_begin = (int)blockIdx.y * chunk, m_end = m_begin + chunk < M ? m_begin + chunk : M;
    float sg = 0.f, sb = 0.f;
    if (c < C)
        for (int m = m_begin + w; m < m_end; m += 4) {
            const float g = gy[(int64_t)m * C + c];
            const float xh = (x[(int64_t)m * C + c] - stats[(int64_t)m * 2]) * stats[(int64_t)m * 2 + 1];
            sg += g * xh;
            sb += g;
        }
    p1[w][threadIdx.x & 63] = sg;
    p2[w][threadIdx.x & 63] = sb;
    __syncthreads();
    if (w == 0 && c < C) {
        atomicAdd(dgamma + c, p1[0][threadIdx.x] + p1[1][threadIdx.x] + p1[2][threadIdx.x] + p1[3][threadIdx.x]);
        atomicAdd(dbeta + c, p2[0][threadIdx.x] + p2[1][threadIdx.x] + p2[2][threadIdx.x] + p2[3][threadIdx.x]);
    }
}

// ------------------------------------------------------------------ GroupNorm(1, C) + ReLU over (rows_per_scene x C) blocks
// forward moments come from the forward's slot accumulators: sums[(scene * ngroups + g) * kGnSlots + slot][2] (float64)
__device__ __forceinline__ void gn_moments(const double* sums, int scene, int ngroups, int g, double cnt, float eps, float& mean,
                                           float& rstd) {
    double S = 0.0, Q = 0.0;
    for (int sl = 0; sl < kGnSlots; ++sl) {
        S += sums[((int64_t)(scene * ngroups + g) * kGnSlots + sl) * 2];
        Q += sums[((int64_t)(scene * ngroups + g) * kGnSlots + sl) * 2 + 1];
    }
    gn_mean_rstd(S, Q, 1.0 / cnt, eps, mean, rstd);
}

struct GnArgs {
    const float* x; int64_t ldx;       // pre-norm activations [M][ngroups*C] (group g at column g*C)
    const double* sums;                // forward moments
    const float* gamma; const float* beta;    // [ngroups][C]
    int M, C, ngroups, rows_per_scene;
    float eps;
    // apply: y = relu(gn(x))
    float* y; int64_t ldy;
    // backward
    const float* gy; int64_t ldgy;     // gradient w.r.t. y
    float* gz; int64_t ldgz;           // scratch: gradient w.r.t. z = gamma xhat + beta (after the ReLU mask)
    double* bsums;                     // [B][ngroups][2] backward sums (zeroed by the caller)
    float* gx; int64_t ldgx;           // gradient w.r.t. x
    float* dgamma; float* dbeta;       // accumulated
};

// grid (ceil(rows_per_scene*C / 1024), ngroups, B)
__global__ __launch_bounds__(256) void gn_apply_kernel(GnArgs a) {
    const int g = blockIdx.y, b = blockIdx.z;
    __shared__ float st[2];
    if (threadIdx.x == 0) gn_moments(a.sums, b, a.ngroups, g, (double)a.rows_per_scene * a.C, a.eps, st[0], st[1]);
    __syncthreads();
    const float mean = st[0], rstd = st[1];
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < (int64_t)a.rows_per_scene * a.C && i < (int64_t)(blockIdx.x + 1) * 1024;
         i += 256) {
        const int r = (int)(i / a.C), c = (int)(i - (int64_t)r * a.C);
        const int64_t m = (int64_t)b * a.rows_per_scene + r;
        if (m >= a.M) continue;
        const float z = (a.x[m * a.ldx + g * a.C + c] - mean) * rstd * a.gamma[g * a.C + c] + a.beta[g * a.C + c];
        a.y[m * a.ldy + g * a.C + c] = z > 0.f ? z : 0.f;
    }
}
// pass 1: gz = gy [z > 0]; bsums += (sum gz gamma, sum gz gamma xhat)
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(GnArgs a) {
    const int g = blockIdx.y, b = blockIdx.z;
    __shared__ float st[2];
    __shared__ double red[2][4];
    if (threadIdx.x == 0) gn_moments(a.sums, b, a.ngroups, g, (double)a.rows_per_scene * a.C, a.eps, st[0], st[1]);
    __syncthreads();
    const float mean = st[0], rstd = st[1];
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < (int64_t)a.rows_per_scene * a.C && i < (int64_t)(blockIdx.x + 1) * 1024;
         i += 256) {
        const int r = (int)(i / a.C), c = (int)(i - (int64_t)r * a.C);
        const int64_t m = (int64_t)b * a.rows_per_scene + r;
        if (m >= a.M) continue;
        const float xh = (a.x[m * a.ldx + g * a.C + c] - mean) * rstd;
        const float gm = a.gamma[g * a.C + c];
        const float z = xh * gm + a.beta[g * a.C + c];
        const float gzv = z > 0.f ? a.gy[m * a.ldgy + g * a.C + c] : 0.f;
        a.gz[m * a.ldgz + g * a.C + c] = gzv;
        s1 += (double)(gzv * gm);
        s2 += (double)(gzv * gm) * (double)xh;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s1;
        red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(a.bsums + ((int64_t)b * a.ngroups + g) * 2, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(a.bsums + ((int64_t)b * a.ngroups + g) * 2 + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}
// pass 2: gx = rstd (gz gamma - S1/n - xhat S2/n)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnArgs a) {
    const int g = blockIdx.y, b = blockIdx.z;
    __shared__ float st[4];
    if (threadIdx.x == 0) {
        gn_moments(a.sums, b, a.ngroups, g, (double)a.rows_per_scene * a.C, a.eps, st[0], st[1]);
        const double n = (double)a.rows_per_scene * a.C;
        st[2] = (float)(a.bsums[((int64_t)b * a.ngroups + g) * 2] / n);
        st[3] = (float)(a.bsums[((int64_t)b * a.ngroups + g) * 2 + 1] / n);
    }
    __syncthreads();
    const float mean = st[0], rstd = st[1], m1 = st[2], m2 = st[3];
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < (int64_t)a.rows_per_scene * a.C && i < (int64_t)(blockIdx.x + 1) * 1024;
         i += 256) {
        const int r = (int)(i / a.C), c = (int)(i - (int64_t)r * a.C);
        const int64_t m = (int64_t)b * a.rows_per_scene + r;
        if (m >= a.M) continue;
        const float xh = (a.x[m * a.ldx + g * a.C + c] - mean) * rstd;
        a.gx[m * a.ldgx + g * a.C + c] = rstd * (a.gz[m * a.ldgz + g * a.C + c] * a.gamma[g * a.C + c] - m1 - xh * m2);
    }
}
// dgamma[g][c] += sum_m gz xhat, dbeta[g][c] += sum_m gz     grid (ceil(C/64), ngroups)
__global__ __launch_bounds__(256) void gn_param_grad_kernel(GnArgs a) {
    __shared__ float p1[4][64], p2[4][64];
    const int g = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    float sg = 0.f, sb = 0.f;
    const int chunk = (a.M + (int)gridDim.z - 1) / (int)gridDim.z;
    const int m_begin = (int)blockIdx.z * chunk, m_end = m_begin + chunk < a.M ? m_begin + chunk : a.M;
    if (c < a.C) {
        int cur_scene = -1;
        float mean = 0.f, rstd = 1.f;
        for (int m = m_begin + w; m < m_end; m += 4) {
            const int scene = m / a.rows_per_scene;
            if (scene != cur_scene) {
                gn_moments(a.sums, scene, a.ngroups, g, (double)a.rows_per_scene * a.C, a.eps, mean, rstd);
                cur_scene = scene;
            }
            const float gzv = a.gz[(int64_t)m * a.ldgz + g * a.C + c];
            sg += gzv * (a.x[(int64_t)m * a.ldx + g * a.C + c] - mean) * rstd;
            sb += gzv;
        }
    }
    p1[w][threadIdx.x & 63] = sg;
    p2[w][threadIdx.x & 63] = sb;
    __syncthreads();
    if (w == 0 && c < a.C) {
        atomicAdd(a.dgamma + g * a.C + c, p1[0][threadIdx.x] + p1[1][threadIdx.x] + p1[2][threadIdx.x] + p1[3][threadIdx.x]);
        atomicAdd(a.dbeta + g * a.C + c, p2[0][threadIdx.x] + p2[1][threadIdx.x] + p2[2][threadIdx.x] + p2[3][threadIdx.x]);
    }
}

// ------------------------------------------------------------------ box decode backward (transformer_parq.py:242-279)
// One thread per query row.  Produces the gradients w.r.t. the 9 last-layer dot products (centre 3, rotation 6) and the
// class / size columns of the fused first head layer; for iteration 0 also d(centre)/d(ref) through inverse_sigmoid.
struct DecodeBwdArgs {
    const float* g_logits; const float* g_center; const float* g_size; const float* g_rot;   // may be null (zero)
    const float* center; const float* size;     // forward outputs
    const float* ref;                           // forward reference points (normalised)
    ScaleBox sb;
    int M, ncls, NH1, C;
    float* g_h3;          // [M][16]: 0..2 centre offsets, 3..8 rotation
    float* g_h1;          // [M][NH1]: this kernel fills columns 2C .. NH1-1 (class logits, size_raw, padding)
    float* g_ref;         // [M][3] or null: += d loss / d ref through the centre update (iteration 0)
};
__global__ void decode_bwd_kernel(DecodeBwdArgs a) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.M) return;
    float* h3 = a.g_h3 + (int64_t)m * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float lo = a.sb.lo[i], hi = a.sb.hi[i];
        const float gc = a.g_center ? a.g_center[(int64_t)m * 3 + i] : 0.f;
        const float sg = (a.center[(int64_t)m * 3 + i] - lo) / (hi - lo);
        const float goff = gc * (hi - lo) * sg * (1.f - sg);                 // d sigmoid
        h3[i] = goff;
        if (a.g_ref) {
            // off = head + log(x1 / x2), x1 = max(r, eps), x2 = max(1 - r, eps), r = clamp(ref, 0, 1)
            const float r0 = a.ref[(int64_t)m * 3 + i];
            const float r = fminf(fmaxf(r0, 0.f), 1.f);
            float d = 0.f;
            if (r0 > 0.f && r0 < 1.f) {
                if (r > 1e-3f) d += 1.f / r;
                if (1.f - r > 1e-3f) d += 1.f / (1.f - r);
            }
            a.g_ref[(int64_t)m * 3 + i] += goff * d;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) h3[3 + j] = a.g_rot ? a.g_rot[(int64_t)m * 6 + j] : 0.f;
    for (int j = 9; j < 16; ++j) h3[j] = 0.f;
    float* h1 = a.g_h1 + (int64_t)m * a.NH1 + 2 * a.C;
    for (int j = 0; j < a.ncls; ++j) h1[j] = a.g_logits ? a.g_logits[(int64_t)m * a.ncls + j] : 0.f;
    // size = exp(size_raw) * mean_size[argmax]: d size / d size_raw = size (the arg-max gather carries no gradient)
    for (int j = 0; j < 3; ++j) h1[a.ncls + j] = a.g_size ? a.g_size[(int64_t)m * 3 + j] * a.size[(int64_t)m * 3 + j] : 0.f;
    for (int j = a.ncls + 3; j < a.NH1 - 2 * a.C; ++j) h1[j] = 0.f;
}
// g_act[m][g*C + c] = sum_j g_h3[m][off_g + j] w3[(6 g + j)][c]   (last head layers: centre 3 rows, rotation 6 rows)
__global__ void head3_bwd_kernel(const float* __restrict__ g_h3, const float* __restrict__ w3, float* __restrict__ g_act, int M, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * 2 * C) return;
    const int m = (int)(i / (2 * C)), gc = (int)(i - (int64_t)m * 2 * C);
    const int g = gc / C, c = gc - g * C;
    const float* h3 = g_h3 + (int64_t)m * 16 + (g ? 3 : 0);
    const int nj = g ? 6 : 3;
    float s = 0.f;
    for (int j = 0; j < nj; ++j) s += h3[j] * w3[(int64_t)(6 * g + j) * C + c];
    g_act[i] = s;
}

// ------------------------------------------------------------------ sine embedding backward (transformer_parq.py:45-64), iteration 0
// emb[m][blk*128 + i] = (i odd ? cos : sin)(2 pi r_axis / dim_t[i]), blocks ordered (y, x, z)
__global__ void posemb_bwd_kernel(const float* __restrict__ g_emb, const float* __restrict__ ref, const float* __restrict__ dim_t,
                                  int M, float* __restrict__ g_ref) {
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int k = lane; k < 384; k += 64) {
        const int blk = k >> 7, i = k & 127;
        const int axis = blk == 0 ? 1 : (blk == 1 ? 0 : 2);
        const float sc = 6.283185307179586f / dim_t[i];
        const float ang = ref[(int64_t)m * 3 + axis] * sc;
        const float d = (i & 1) ? -sinf(ang) : cosf(ang);
        const float v = g_emb[(int64_t)m * 384 + k] * d * sc;
        if (axis == 0) acc[0] += v; else if (axis == 1) acc[1] += v; else acc[2] += v;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
    if (lane < 3) g_ref[(int64_t)m * 3 + lane] += lane == 0 ? acc[0] : (lane == 1 ? acc[1] : acc[2]);
}
// g_refpoint[q][j] += sum_b g_ref[b][q][j] * s (1 - s), s = sigmoid(w) = ref0   (transformer_parq.py:122,309)
__global__ void refpoint_bwd_kernel(const float* __restrict__ g_ref, const float* __restrict__ ref0, int B, int Q, float* __restrict__ g_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q * 3) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += g_ref[(int64_t)b * Q * 3 + i];
    const float r = ref0[i];
    g_w[i] += s * r * (1.f - r);
}

// ------------------------------------------------------------------ project + sample backward (transformer_parq.py:129-161)
// tgt = sum_views bilinear(tokens, u_v) / max(#valid views, 1): the gradient of every view's sample is g_tgt / denom, scattered
// to the 4 corners with the bilinear weights (float atomics into g_tokens).  With g_ref (iteration 0, where the reference
// points come from the learnable embedding) the coordinate gradient is propagated too:
//   d sample / du = sum_c g_c ((a01 - a00) wy0 + (a11 - a10) wy1)  (out-of-range corners count as zero), likewise dv;
//   u = x / zc fx + cx, zc = max(z, eps): dz only while z > eps;  (x, y, z) = R P + t;  P = ref (hi - lo) + lo.
// One workgroup per (scene, query), one wave per view slot — the same geometry code path as the forward kernel.
__global__ __launch_bounds__(1024) void sample_bwd_kernel(const float* __restrict__ tokens, const double* __restrict__ T_cl,
                                                          const float* __restrict__ cam, const float* __restrict__ ref, ScaleBox sb,
                                                          int V, int h, int w, int C, int Q, const float* __restrict__ g_tgt,
                                                          float* __restrict__ g_tokens, float* __restrict__ g_ref) {
    __shared__ int cnt[16];
    __shared__ double gP[16][3];
    const int bq = blockIdx.x, b = bq / Q;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    double P[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) P[i] = (double)ref[(int64_t)bq * 3 + i] * ((double)sb.hi[i] - (double)sb.lo[i]) + (double)sb.lo[i];
    // pass 1: number of valid views (the forward's denominator)
    int nvalid = 0;
    for (int v = wv; v < V; v += nwv) {
        const double* T = T_cl + ((int64_t)b * V + v) * 12;
        const float* cm = cam + ((int64_t)b * V + v) * 6;
        const double x = P[0] * T[0] + P[1] * T[1] + P[2] * T[2] + T[9];
        const double y = P[0] * T[3] + P[1] * T[4] + P[2] * T[5] + T[10];
        const double z = P[0] * T[6] + P[1] * T[7] + P[2] * T[8] + T[11];
        const double eps = (double)1e-3f;
        const double zc = z > eps ? z : eps;
        const double u = (x / zc) * (double)cm[2] + (double)cm[4];
        const double vv = (y / zc) * (double)cm[3] + (double)cm[5];
        nvalid += (z > eps && u >= 0.0 && u <= (double)cm[0] - 1.0 && vv >= 0.0 && vv <= (double)cm[1] - 1.0) ? 1 : 0;
    }
    if (lane == 0) {
        cnt[wv] = nvalid;
        gP[wv][0] = gP[wv][1] = gP[wv][2] = 0.0;
    }
    __syncthreads();
    int total = 0;
    for (int i = 0; i < nwv; ++i) total += cnt[i];
    const float inv_denom = 1.f / (float)(total > 0 ? total : 1);
    const float* g = g_tgt + (int64_t)bq * C;
    double gp0 = 0.0, gp1 = 0.0, gp2 = 0.0;
    for (int v = wv; v < V; v += nwv) {
        const double* T = T_cl + ((int64_t)b * V + v) * 12;
        const float* cm = cam + ((int64_t)b * V + v) * 6;
        const double x = P[0] * T[0] + P[1] * T[1] + P[2] * T[2] + T[9];
        const double y = P[0] * T[3] + P[1] * T[4] + P[2] * T[5] + T[10];
        const double z = P[0] * T[6] + P[1] * T[7] + P[2] * T[8] + T[11];
        const double eps = (double)1e-3f;
        const bool front = z > eps;
        const double zc = front ? z : eps;
        const double u = (x / zc) * (double)cm[2] + (double)cm[4];
        const double vv = (y / zc) * (double)cm[3] + (double)cm[5];
        const double fx0 = floor(u), fy0 = floor(vv);
        if (!(fx0 >= -1.0 && fx0 <= (double)(w - 1) && fy0 >= -1.0 && fy0 <= (double)(h - 1))) continue;
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx1 = (float)(u - fx0), wx0 = (float)(1.0 - (u - fx0));
        const float wy1 = (float)(vv - fy0), wy0 = (float)(1.0 - (vv - fy0));
        const bool x0ok = x0 >= 0, x1ok = x0 + 1 <= w - 1, y0ok = y0 >= 0, y1ok = y0 + 1 <= h - 1;
        const int64_t base = (((int64_t)b * V + v) * h) * (int64_t)w * C;
        const int64_t o00 = base + ((int64_t)(y0ok ? y0 : 0) * w + (x0ok ? x0 : 0)) * C;
        const int64_t o01 = base + ((int64_t)(y0ok ? y0 : 0) * w + (x1ok ? x0 + 1 : 0)) * C;
        const int64_t o10 = base + ((int64_t)(y1ok ? y0 + 1 : 0) * w + (x0ok ? x0 : 0)) * C;
        const int64_t o11 = base + ((int64_t)(y1ok ? y0 + 1 : 0) * w + (x1ok ? x0 + 1 : 0)) * C;
        const bool k00 = x0ok && y0ok, k01 = x1ok && y0ok, k10 = x0ok && y1ok, k11 = x1ok && y1ok;
        float su = 0.f, sv = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float gc = g[c] * inv_denom;
            if (g_tokens) {
                if (k00) atomicAdd(g_tokens + o00 + c, gc * wy0 * wx0);
                if (k01) atomicAdd(g_tokens + o01 + c, gc * wy0 * wx1);
                if (k10) atomicAdd(g_tokens + o10 + c, gc * wy1 * wx0);
                if (k11) atomicAdd(g_tokens + o11 + c, gc * wy1 * wx1);
            }
            if (g_ref) {
                const float a00 = k00 ? tokens[o00 + c] : 0.f, a01 = k01 ? tokens[o01 + c] : 0.f;
                const float a10 = k10 ? tokens[o10 + c] : 0.f, a11 = k11 ? tokens[o11 + c] : 0.f;
                su += gc * ((a01 - a00) * wy0 + (a11 - a10) * wy1);
                sv += gc * ((a10 - a00) * wx0 + (a11 - a01) * wx1);
            }
        }
        if (g_ref) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                su += __shfl_xor(su, o);
                sv += __shfl_xor(sv, o);
            }
            const double gu = (double)su, gv = (double)sv;
            const double gx = gu * (double)cm[2] / zc, gy = gv * (double)cm[3] / zc;
            const double gz = front ? -(gu * (double)cm[2] * x + gv * (double)cm[3] * y) / (zc * zc) : 0.0;
            gp0 += gx * T[0] + gy * T[3] + gz * T[6];
            gp1 += gx * T[1] + gy * T[4] + gz * T[7];
            gp2 += gx * T[2] + gy * T[5] + gz * T[8];
        }
    }
    if (g_ref) {
        if (lane == 0) {
            gP[wv][0] = gp0;
            gP[wv][1] = gp1;
            gP[wv][2] = gp2;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double sacc = 0.0;
            for (int i = 0; i < nwv; ++i) sacc += gP[i][threadIdx.x];
            g_ref[(int64_t)bq * 3 + threadIdx.x] += (float)(sacc * ((double)sb.hi[threadIdx.x] - (double)sb.lo[threadIdx.x]));
        }
    }
}

}  // namespace

// =============================================================================== launchers
hipError_t launch_transpose(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int R, int Cc, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(Cc, 32), ceil_div(R, 32)), dim3(256), 0, s, src, ld_src, dst, ld_dst, R, Cc);
    return hipGetLastError();
}
hipError_t launch_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t ldo, int M, int N, int K,
                          int accumulate, hipStream_t s, float* bias, int bias_from) {
    TnArgs a;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.accumulate = accumulate;
    a.bias = bias; a.bias_from = bias_from;
    // large outputs from few rows (C = 1024 layers): 64 x 64 tiles with LDS-staged rows
    if ((int64_t)N * K >= (1 << 19) && M <= 4096) {
        hipLaunchKernelGGL(gemm_tn_tile64_kernel, dim3(ceil_div(N, 64) * ceil_div(K, 64)), dim3(256), 0, s, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || !bias) return e;
        return launch_colsum(A + bias_from, lda, M, N - bias_from, bias + bias_from, accumulate, s);     // not fused in that kernel
    }
    // many rows, few output tiles (the K/V projection backward: M = all tokens): split the rows, accumulate with atomics
    // whenever the result is accumulated anyway, split the rows — but the split launch adds its partial tiles with float atomics
    // (N*K*splits of them, executed memory-side) and THOSE bound the small dW GEMMs: 1024 rows into a 256 x 256 output ran 16.7 us
    // at 64 rows per workgroup, 12.9 at 128, 8.95 at 256 (all operand loads of a wave in flight together, kernel above); 512-row
    // chunks with twice the loads in flight measured the same 9.1 us.  Every workgroup keeps at least 256 rows.
    int splits = 1;
    const int tiles = ceil_div(N, 32) * ceil_div(K, 32);
    if (accumulate && M >= 256) {
        splits = ceil_div(4 * device_num_cus(), tiles);
        static const int min_rows = [] { const char* e = dev_env("PARQ_TN_ROWS"); return e && atoi(e) > 0 ? atoi(e) : 256; }();
        const int cap = M >= 8192 ? M / 1024 : M / min_rows;
        if (splits > cap) splits = cap;
        if (splits < 1) splits = 1;
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, hipStream_t s) {
    int splits = 1;
    if (accumulate && M >= 512) {                 // few columns, many rows: spread the rows over workgroups (float atomics)
        splits = M >= 8192 ? M / 1024 : M / 128;
        if (splits > 1024) splits = 1024;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 64), splits), dim3(256), 0, s, X, ldx, M, N, out, accumulate);
    return hipGetLastError();
}
hipError_t launch_add(const float* a, const float* b, float* y, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, a, b, y, n);
    return hipGetLastError();
}
hipError_t launch_axpy_rows(const float* x, int64_t ldx, float* y, int64_t ldy, int M, int N, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(axpy_rows_kernel, dim3((unsigned)ceil_div64((int64_t)M * N, 256)), dim3(256), 0, s, x, ldx, y, ldy, M, N, accumulate);
    return hipGetLastError();
}
hipError_t launch_dropout_apply(const float* src, float* dst, int M, int N, float p, uint32_t seed, hipStream_t s) {
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, src, dst, n, N, p, seed);
    return hipGetLastError();
}
hipError_t launch_ln_bwd(const float* gy, const float* x, const float* stats, const float* gamma, float* gx, int M, int C,
                         int accumulate, float* dgamma, float* dbeta, hipStream_t s) {
    if (C > 64 * kLnPer) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, s, gy, x, stats, gamma, gx, M, C, accumulate);
    if (dgamma) {
        int splits = M / 64;
        splits = splits < 1 ? 1 : (splits > 64 ? 64 : splits);
        hipLaunchKernelGGL(ln_param_grad_kernel, dim3(ceil_div(C, 64), splits), dim3(256), 0, s, gy, x, stats, M, C, dgamma, dbeta);
    }
    return hipGetLastError();
}
// y = relu(GroupNorm(x)) from the forward moments
hipError_t launch_gn_apply(const float* x, int64_t ldx, const double* sums, const float* gamma, const float* beta, int M, int C,
                           int ngroups, int rows_per_scene, float eps, float* y, int64_t ldy, hipStream_t s) {
    GnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.sums = sums; a.gamma = gamma; a.beta = beta; a.M = M; a.C = C; a.ngroups = ngroups;
    a.rows_per_scene = rows_per_scene; a.eps = eps; a.y = y; a.ldy = ldy;
    const int B = ceil_div(M, rows_per_scene);
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)ceil_div64((int64_t)rows_per_scene * C, 1024), ngroups, B), dim3(256), 0, s, a);
    return hipGetLastError();
}
// backward of y = relu(GroupNorm(x)): gx from gy; gz / bsums are scratch ([M][ngroups*C] floats, [B][ngroups][2] doubles)
hipError_t launch_gn_bwd(const float* x, int64_t ldx, const double* sums, const float* gamma, const float* beta, int M, int C,
                         int ngroups, int rows_per_scene, float eps, const float* gy, int64_t ldgy, float* gz, double* bsums,
                         float* gx, int64_t ldgx, float* dgamma, float* dbeta, hipStream_t s) {
    GnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.sums = sums; a.gamma = gamma; a.beta = beta; a.M = M; a.C = C; a.ngroups = ngroups;
    a.rows_per_scene = rows_per_scene; a.eps = eps; a.gy = gy; a.ldgy = ldgy; a.gz = gz; a.ldgz = (int64_t)ngroups * C;
    a.bsums = bsums; a.gx = gx; a.ldgx = ldgx; a.dgamma = dgamma; a.dbeta = dbeta;
    const int B = ceil_div(M, rows_per_scene);
    hipError_t e = hipMemsetAsync(bsums, 0, (size_t)B * ngroups * 2 * sizeof(double), s);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)ceil_div64((int64_t)rows_per_scene * C, 1024), ngroups, B);
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, dim3(256), 0, s, a);
    if (dgamma) {
        int splits = M / 64;
        splits = splits < 1 ? 1 : (splits > 64 ? 64 : splits);
        hipLaunchKernelGGL(gn_param_grad_kernel, dim3(ceil_div(C, 64), ngroups, splits), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_decode_bwd(const float* g_logits, const float* g_center, const float* g_size, const float* g_rot,
                             const float* center, const float* size, const float* ref, ScaleBox sb, int M, int ncls, int NH1, int C,
                             float* g_h3, float* g_h1, float* g_ref, hipStream_t s) {
    DecodeBwdArgs a;
    a.g_logits = g_logits; a.g_center = g_center; a.g_size = g_size; a.g_rot = g_rot; a.center = center; a.size = size;
    a.ref = ref; a.sb = sb; a.M = M; a.ncls = ncls; a.NH1 = NH1; a.C = C; a.g_h3 = g_h3; a.g_h1 = g_h1; a.g_ref = g_ref;
    hipLaunchKernelGGL(decode_bwd_kernel, dim3(ceil_div(M, 64)), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_head3_bwd(const float* g_h3, const float* w3, float* g_act, int M, int C, hipStream_t s) {
    hipLaunchKernelGGL(head3_bwd_kernel, dim3((unsigned)ceil_div64((int64_t)M * 2 * C, 256)), dim3(256), 0, s, g_h3, w3, g_act, M, C);
    return hipGetLastError();
}
hipError_t launch_posemb_bwd(const float* g_emb, const float* ref, const float* dim_t, int M, float* g_ref, hipStream_t s) {
    hipLaunchKernelGGL(posemb_bwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, s, g_emb, ref, dim_t, M, g_ref);
    return hipGetLastError();
}
hipError_t launch_refpoint_bwd(const float* g_ref, const float* ref0, int B, int Q, float* g_w, hipStream_t s) {
    hipLaunchKernelGGL(refpoint_bwd_kernel, dim3(ceil_div(Q * 3, 64)), dim3(64), 0, s, g_ref, ref0, B, Q, g_w);
    return hipGetLastError();
}

hipError_t launch_sample_bwd(const float* tokens, const double* T_cl, const float* cam, const float* ref, ScaleBox sb, int B, int V,
                             int h, int w, int C, int Q, const float* g_tgt, float* g_tokens, float* g_ref, hipStream_t s) {
    const int nwv = V < 16 ? V : 16;
    hipLaunchKernelGGL(sample_bwd_kernel, dim3(B * Q), dim3(nwv * 64), 0, s, tokens, T_cl, cam, ref, sb, V, h, w, C, Q, g_tgt, g_tokens,
                       g_ref);
    return hipGetLastError();
}

}  // namespace parq
