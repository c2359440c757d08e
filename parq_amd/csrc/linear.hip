// Small-GEMM kernel of the per-iteration chain:  Y = act(pro(X) @ W^T + b) (+ R)
//
// These GEMMs have M = B*Q rows (256..2048) and K, N of a few hundred: they are
// latency-bound, with both operands L2-resident.  Design for gfx950:
//   * one workgroup = 4 waves = one 32x32 output tile, so a 256x256 output is 64
//     workgroups and a 256x768 one 192 — enough to spread over the 256 CUs;
//   * the 4 waves split K four ways (in-workgroup split-K) and reduce through LDS,
//     which cuts the dependent MFMA chain per tile to K/8 instructions;
//   * v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain), lane l supplies A[i=l&31][k=l>>5]
//     and B[k=l>>5][j=l&31].  The contraction index may be permuted freely, so lane
//     half kh takes a CONTIGUOUS run of k: one 16-byte global load feeds 4 MFMAs and
//     both operands are read in their native K-contiguous layouts (X [M][K], W [N][K]);
//   * prologues on A (add a second matrix, GroupNorm+ReLU) and epilogues (bias, ReLU,
//     residual, head-major scatter) are fused so the chain needs no extra passes.
#include "common.hpp"

namespace parq {

namespace {

constexpr int kTile = 32;
constexpr int kWaves = 4;
constexpr int kChunkBatch = 8;   // float4 chunks of A and B kept in flight per lane

__global__ __launch_bounds__(kWaves * kWave) void linear_f32_kernel(LinearArgs a) {
    __shared__ __attribute__((aligned(16))) float red[kWaves * 16 * kWave];

    // 1-D tile index, column tiles fastest: the column tiles of one row block run together
    // and share its A rows through L2
    const int g = blockIdx.y;
    const int ntn = (a.N + kTile - 1) / kTile;
    const int n0 = (int)(blockIdx.x % ntn) * kTile;
    const int m0 = (int)(blockIdx.x / ntn) * kTile;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;

    const float* X = a.X + g * a.gX;
    const float* W = a.W + g * a.gW;

    const int KS = a.K / kWaves;          // this wave's K slice
    const int KH = KS / 2;                // this lane-half's run
    const int kbase = wave * KS + kh * KH;
    const int nchunks = KH / 4;

    const int m = m0 + li;
    const int n = n0 + li;
    const bool m_ok = m < a.M;
    const bool n_ok = n < a.N;
    const float* xrow = X + (int64_t)(m_ok ? m : 0) * a.ldx + kbase;
    const float* wrow = W + (int64_t)(n_ok ? n : 0) * a.ldw + kbase;
    const bool add2 = (a.X2 != nullptr) && (n0 < a.x2_ncols);
    const float* x2row = add2 ? a.X2 + (int64_t)(m_ok ? m : 0) * a.ldx2 + kbase : nullptr;

    float gn_mean = 0.f, gn_rstd = 1.f;
    const float* gam = nullptr;
    const float* bet = nullptr;
    if (a.gn_stats) {
        const int scene = (m_ok ? m : 0) / a.gn_rows_per_scene;
        gn_mean = a.gn_stats[(scene * a.gn_ngroups + g) * 2 + 0];
        gn_rstd = a.gn_stats[(scene * a.gn_ngroups + g) * 2 + 1];
        gam = a.gn_gamma + g * a.gGamma + kbase;
        bet = a.gn_beta + g * a.gGamma + kbase;
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int c0 = 0; c0 < nchunks; c0 += kChunkBatch) {
        f32x4 av[kChunkBatch], bv[kChunkBatch];
#pragma unroll
        for (int c = 0; c < kChunkBatch; ++c) {
            const int cc = c0 + c;
            if (cc < nchunks) {
                av[c] = *reinterpret_cast<const f32x4*>(xrow + cc * 4);
                bv[c] = *reinterpret_cast<const f32x4*>(wrow + cc * 4);
            } else {
                av[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                bv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (add2) {
#pragma unroll
            for (int c = 0; c < kChunkBatch; ++c)
                if (c0 + c < nchunks) av[c] += *reinterpret_cast<const f32x4*>(x2row + (c0 + c) * 4);
        }
        if (a.gn_stats) {
#pragma unroll
            for (int c = 0; c < kChunkBatch; ++c) {
                if (c0 + c < nchunks) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(gam + (c0 + c) * 4);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(bet + (c0 + c) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = (av[c][e] - gn_mean) * gn_rstd * gv[e] + be[e];
                        av[c][e] = y > 0.f ? y : 0.f;
                    }
                }
            }
        }
        if (!m_ok) {
#pragma unroll
            for (int c = 0; c < kChunkBatch; ++c) av[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!n_ok) {
#pragma unroll
            for (int c = 0; c < kChunkBatch; ++c) bv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < kChunkBatch; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][e], bv[c][e], acc, 0, 0, 0);
        }
    }

    // in-workgroup split-K reduction: red[wave][reg][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * kWave + lane] = acc[r];
    __syncthreads();

    // thread -> (row, 4 consecutive cols) of the 32x32 tile
    const int row = tid >> 3;
    const int c4 = (tid & 7) * 4;
    const int reg = (row & 3) + 4 * (row >> 3);
    const int src_lane = c4 + 32 * ((row >> 2) & 1);
    f32x4 sum = *reinterpret_cast<const f32x4*>(&red[(0 * 16 + reg) * kWave + src_lane]);
#pragma unroll
    for (int w = 1; w < kWaves; ++w) sum += *reinterpret_cast<const f32x4*>(&red[(w * 16 + reg) * kWave + src_lane]);

    const int om = m0 + row;
    if (om >= a.M) return;
    const float* bias = a.bias ? a.bias + g * a.gBias : nullptr;
    float* Ybase = a.Y + g * a.gY + (int64_t)(om / a.rows_per_batch) * a.y_batch +
                   (int64_t)(om % a.rows_per_batch) * a.y_row;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int on = n0 + c4 + e;
        if (on < a.N) {
            float y = sum[e];
            if (bias) y += bias[on];
            if (a.relu) y = y > 0.f ? y : 0.f;
            if (a.R) y += a.R[(int64_t)om * a.ldr + on];
            Ybase[(int64_t)(on / a.col_blk) * a.y_blk + (on % a.col_blk)] = y;
        }
    }
}

}  // namespace

hipError_t launch_linear(const LinearArgs& a, int groups, hipStream_t s) {
    if (a.K % 32 != 0 || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
    const int64_t tiles = (int64_t)ceil_div(a.N, kTile) * ceil_div(a.M, kTile);
    if (tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)tiles, groups, 1);
    hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(kWaves * kWave), 0, s, a);
    return hipGetLastError();
}

}  // namespace parq
