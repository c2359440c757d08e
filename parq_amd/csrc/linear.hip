// Small-GEMM kernel of the per-iteration chain:  Y = act(pro(X) @ W^T + b) (+ R)
//
// These GEMMs have M = B*Q rows (256..2048) and K, N of a few hundred: they are
// latency-bound (every dependent launch costs ~4-5 us of dispatch + cross-XCD memory latency
// on this chip), so the design goal is FEW launches with enough workgroups each:
//   * one workgroup = 4 waves = one 32x32 output tile, so a 256x256 output is 64
//     workgroups and a 256x768 one 192 — enough to spread over the 256 CUs;
//   * the 4 waves split K four ways (in-workgroup split-K) and reduce through LDS,
//     which cuts the dependent MFMA chain per tile to K/8 instructions;
//   * v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain), lane l supplies A[i=l&31][k=l>>5]
//     and B[k=l>>5][j=l&31].  The contraction index may be permuted freely, so lane
//     half kh takes a CONTIGUOUS run of k: one 16-byte global load feeds 4 MFMAs and
//     both operands are read in their native K-contiguous layouts (X [M][K], W [N][K]);
//   * everything that is row-local or a scene-wide scalar is fused instead of launched:
//       prologues on A : LayerNorm of the input rows (statistics computed by the tile
//                        itself, optionally published for later residual use), + second
//                        matrix (pos-embed), GroupNorm(1,C)+ReLU from scene-wide moments
//       epilogues      : bias, ReLU, residual (plain, or LayerNorm of a pre-norm buffer
//                        recomputed from published row statistics), head-major scatter,
//                        scene-wide GroupNorm moments of the OUTPUT accumulated with fp64
//                        atomics for the next layer's prologue.
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

    // ---- GroupNorm(1,C) prologue: scene-wide moments accumulated by the producer's epilogue
    float gn_mean = 0.f, gn_rstd = 1.f;
    const float* gam = nullptr;
    const float* bet = nullptr;
    if (a.gn_sums) {
        const int scene = (m_ok ? m : 0) / a.gn_rows_per_scene;
        const double cnt = (double)a.gn_rows_per_scene * (double)a.K;
        // the producer spread its atomics over kGnSlots accumulators per (scene, group)
        double S = 0.0, Qs = 0.0;
#pragma unroll
        for (int sl = 0; sl < kGnSlots; ++sl) {
            S += a.gn_sums[((scene * a.gn_ngroups + g) * kGnSlots + sl) * 2 + 0];
            Qs += a.gn_sums[((scene * a.gn_ngroups + g) * kGnSlots + sl) * 2 + 1];
        }
        const double mean = S / cnt;
        double var = Qs / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        gn_mean = (float)mean;
        gn_rstd = (float)(1.0 / sqrt(var + (double)a.norm_eps));
        gam = a.gn_gamma + g * a.gGamma + kbase;
        bet = a.gn_beta + g * a.gGamma + kbase;
    }

    // ---- issue every independent global load up front: this kernel is latency-bound (operands come
    // from the other XCDs' writes or from HBM/MALL after the attention kernel swept the L2s), so the
    // first A/B batch and all epilogue operands are requested before anything waits on anything.
    f32x4 av[kChunkBatch], bv[kChunkBatch];
    auto load_batch = [&](int c0) {
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
    };
    load_batch(0);
    // epilogue operands of this thread's (row, 4 cols)
    const int erow = tid >> 3;
    const int ec4 = (tid & 7) * 4;
    const int eom = m0 + erow;
    const bool erow_ok = eom < a.M;
    const float* bias = a.bias ? a.bias + g * a.gBias : nullptr;
    float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_r[4] = {0.f, 0.f, 0.f, 0.f}, e_rg[4] = {1.f, 1.f, 1.f, 1.f}, e_rb[4] = {0.f, 0.f, 0.f, 0.f};
    float rmean = 0.f, rrstd = 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int on = n0 + ec4 + e;
        if (on < a.N) {
            if (bias) e_bias[e] = bias[on];
            if (a.R && erow_ok) e_r[e] = a.R[(int64_t)eom * a.ldr + on];
            if (a.rln_stats) {
                e_rg[e] = a.rln_gamma[on];
                e_rb[e] = a.rln_beta[on];
            }
        }
    }
    if (a.rln_stats && erow_ok) {
        rmean = a.rln_stats[(int64_t)eom * 2 + 0];
        rrstd = a.rln_stats[(int64_t)eom * 2 + 1];
    }

    // ---- LayerNorm prologue: row statistics of A over the full K, reduced across the lane halves and
    // the 4 K-slices through LDS.  One pass over data shifted by the row's first element (the shift
    // removes the cancellation of E[x^2] - mean^2).
    float ln_mean = 0.f, ln_rstd = 1.f;
    if (a.ln_gamma) {
        const float shift = X[(int64_t)(m_ok ? m : 0) * a.ldx];
        float s = 0.f, q = 0.f;
        if (nchunks <= kChunkBatch) {                  // the lane's whole K run is already in registers
#pragma unroll
            for (int c = 0; c < kChunkBatch; ++c)
                if (c < nchunks) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = av[c][e] - shift;
                        s += d;
                        q += d * d;
                    }
                }
        } else {
            for (int c = 0; c < nchunks; ++c) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[e] - shift;
                    s += d;
                    q += d * d;
                }
            }
        }
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (kh == 0) {
            red[wave * 64 + li] = s;
            red[wave * 64 + 32 + li] = q;
        }
        __syncthreads();
        const float S = red[li] + red[64 + li] + red[128 + li] + red[192 + li];
        const float Q2 = red[32 + li] + red[96 + li] + red[160 + li] + red[224 + li];
        __syncthreads();
        const float invK = 1.f / (float)a.K;
        const float dm = S * invK;
        ln_mean = shift + dm;
        const float var = fmaxf(Q2 * invK - dm * dm, 0.f);
        ln_rstd = 1.f / sqrtf(var + a.norm_eps);
        if (a.ln_stats_out && n0 == 0 && wave == 0 && kh == 0 && m_ok) {
            a.ln_stats_out[(int64_t)m * 2 + 0] = ln_mean;
            a.ln_stats_out[(int64_t)m * 2 + 1] = ln_rstd;
        }
    }
    const float* lng = a.ln_gamma ? a.ln_gamma + kbase : nullptr;
    const float* lnb = a.ln_gamma ? a.ln_beta + kbase : nullptr;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int c0 = 0; c0 < nchunks; c0 += kChunkBatch) {
        if (c0 > 0) load_batch(c0);
        if (a.ln_gamma) {
#pragma unroll
            for (int c = 0; c < kChunkBatch; ++c) {
                if (c0 + c < nchunks) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(lng + (c0 + c) * 4);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(lnb + (c0 + c) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[c][e] = (av[c][e] - ln_mean) * ln_rstd * gv[e] + be[e];
                }
            }
        }
        if (add2) {
#pragma unroll
            for (int c = 0; c < kChunkBatch; ++c)
                if (c0 + c < nchunks) av[c] += *reinterpret_cast<const f32x4*>(x2row + (c0 + c) * 4);
        }
        if (a.gn_sums) {
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

    const int om = eom;
    const bool row_ok = erow_ok;
    double gs = 0.0, gq = 0.0;                       // moments of this thread's outputs (GroupNorm of the output)
    const bool gn_out = a.gn_out_sums != nullptr && n0 < a.gn_out_ncols;
    if (row_ok) {
        float* Ybase = a.Y + g * a.gY + (int64_t)(om / a.rows_per_batch) * a.y_batch +
                       (int64_t)(om % a.rows_per_batch) * a.y_row;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int on = n0 + c4 + e;
            if (on < a.N) {
                float y = sum[e] + e_bias[e];
                if (a.relu) y = y > 0.f ? y : 0.f;
                if (a.R) y += a.rln_stats ? (e_r[e] - rmean) * rrstd * e_rg[e] + e_rb[e] : e_r[e];
                Ybase[(int64_t)(on / a.col_blk) * a.y_blk + (on % a.col_blk)] = y;
                gs += (double)y;
                gq += (double)y * (double)y;
            }
        }
    }
    if (gn_out) {
        // scene-wide moments of the output block: one fp64 atomic pair per workgroup when the tile lies
        // in one scene (always, when Q % 32 == 0), else per thread
        const int sc_first = m0 / a.gn_out_rows_per_scene;
        const int last_row = (m0 + kTile - 1 < a.M ? m0 + kTile - 1 : a.M - 1);
        const int grp = (n0 + g * a.N) / a.gn_out_group_cols;
        if (sc_first == last_row / a.gn_out_rows_per_scene) {
            __syncthreads();
            double* dred = reinterpret_cast<double*>(red);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                gs += __shfl_xor(gs, o);
                gq += __shfl_xor(gq, o);
            }
            if (lane == 0) {
                dred[wave * 2 + 0] = gs;
                dred[wave * 2 + 1] = gq;
            }
            __syncthreads();
            if (tid == 0) {
                double S = 0.0, Q2 = 0.0;
                for (int w = 0; w < kWaves; ++w) {
                    S += dred[w * 2 + 0];
                    Q2 += dred[w * 2 + 1];
                }
                // slot = row tile index: only the column tiles of one row block contend on an address
                const int slot = (m0 / kTile) % kGnSlots;
                double* dst = a.gn_out_sums + (((int64_t)sc_first * a.gn_out_ngroups + grp) * kGnSlots + slot) * 2;
                atomicAdd(dst, S);
                atomicAdd(dst + 1, Q2);
            }
        } else if (row_ok) {
            const int slot = (m0 / kTile) % kGnSlots;
            double* dst = a.gn_out_sums + (((int64_t)(om / a.gn_out_rows_per_scene) * a.gn_out_ngroups + grp) * kGnSlots + slot) * 2;
            atomicAdd(dst, gs);
            atomicAdd(dst + 1, gq);
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
