// Small-GEMM kernel of the per-iteration chain:  Y = act(pro(X) @ W^T + b) (+ R)
//
// These GEMMs have M = B*Q rows (256..2048) and K, N of a few hundred.  At one scene they are
// LATENCY-bound: measured on MI355X (tools/bench_src/small_linear.hip) a dependent launch costs
// ~2.6 us and the old row-per-lane kernel spent another ~3 us waiting for its operands, because a
// 16-byte load whose 64 lanes sit on 64 different rows is 64 separate requests for the CU's
// texture/L1 pipe.  Design:
//   * v_mfma_f32_16x16x4_f32 (exact fp32, same rate as 32x32x2): lane (i = l&15, kq = l>>4)
//     supplies A[i][kq] and B[kq][i].  The contraction index may be permuted freely, so for each
//     16-wide K chunk lane (i, kq) loads the float4 at k = chunk*16 + kq*4: the four kq groups of a
//     row read 64 contiguous bytes, one load instruction = 16 rows x 64 B = 16 requests instead
//     of 64, and the float4 feeds 4 MFMAs.  Both operands are read in their native K-contiguous
//     layouts (X [M][K], W [N][K]);
//   * one workgroup = 4 waves = one TxT output tile; the 4 waves split K four ways (in-workgroup
//     split-K) and reduce through LDS, so the dependent MFMA chain is K/16 (T=16) instructions.
//     T = 16 when the 32x32 tiling would not fill the chip (256x256 output = 256 workgroups, whole
//     K slice in registers with every load in flight at once), T = 32 (2x2 sub-tiles) otherwise;
//   * everything that is row-local or a scene-wide scalar is fused instead of launched:
//       prologues on A : LayerNorm of the input rows (statistics computed by the tile
//                        itself, optionally published for later residual use), + second
//                        matrix (pos-embed), GroupNorm(1,C)+ReLU from scene-wide moments
//       epilogues      : bias, ReLU, residual (plain, or LayerNorm of a pre-norm buffer
//                        recomputed from published row statistics), head-major scatter,
//                        scene-wide GroupNorm moments of the OUTPUT accumulated with fp64
//                        atomics for the next layer's prologue.
#include "common.hpp"

#include <cstdlib>

namespace parq {

namespace {

constexpr int kWaves = 4;
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int T, int CB>                            // CB: 16-wide K chunks held in registers per batch
__global__ __launch_bounds__(kWaves * kWave) void linear_f32_kernel(LinearArgs a) {
    PARQ_TL_KERNEL(kTlLinear);
    publish_progress(a);
    constexpr int S = T / 16;                       // 16x16 sub-tiles per tile edge
    constexpr int OPT = T * T / (kWaves * kWave);   // outputs per thread in the epilogue (1 or 4)
    constexpr int TPR = T / OPT;                    // threads per output row
    __shared__ __attribute__((aligned(16))) float red[kWaves * S * S * 4 * kWave];

    // 1-D tile index, column tiles fastest: the column tiles of one row block run together
    // and share its A rows through L2
    const int g = blockIdx.y;
    const int ntn = (a.N + T - 1) / T;
    const int n0 = (int)(blockIdx.x % ntn) * T;
    const int m0 = (int)(blockIdx.x / ntn) * T;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int kq = lane >> 4;

    const float* X = a.X + g * a.gX;
    const float* W = a.W + g * a.gW;

    // 16-wide K chunks are dealt round-robin to the 4 waves (wave w takes chunks w, w+4, ...): the
    // workgroup's concurrent loads of a row are contiguous and any K % 16 == 0 works
    const int nchunks = (a.K / 16 - wave + kWaves - 1) / kWaves;
    const int kbase = wave * 16 + kq * 4; // lane's float4 of its c-th chunk sits at kbase + 64 c

    bool m_ok[S], n_ok[S];
    const float* xrow[S];
    const float* wrow[S];
    const float* x2row[S];
    const bool add2 = (a.X2 != nullptr) && (n0 < a.x2_ncols);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int m = m0 + s * 16 + li, n = n0 + s * 16 + li;
        m_ok[s] = m < a.M;
        n_ok[s] = n < a.N;
        xrow[s] = X + (int64_t)(m_ok[s] ? m : 0) * a.ldx + kbase;
        wrow[s] = W + (int64_t)(n_ok[s] ? n : 0) * a.ldw + kbase;
        x2row[s] = add2 ? a.X2 + (int64_t)(m_ok[s] ? m : 0) * a.ldx2 + kbase : nullptr;
    }

    const float* gam = a.gn_sums ? a.gn_gamma + g * a.gGamma + kbase : nullptr;
    const float* bet = a.gn_sums ? a.gn_beta + g * a.gGamma + kbase : nullptr;

    // ---- issue every independent global load up front: this kernel is latency-bound (operands come
    // from the other XCDs' writes or from HBM/MALL after the attention kernel swept the L2s), so the
    // first A/B batch and all epilogue operands are requested before anything waits on anything.
    f32x4v av[S][CB], bv[S][CB];
    auto load_batch = [&](int c0) {
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (c0 + c < nchunks) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    av[s][c] = *reinterpret_cast<const f32x4v*>(xrow[s] + (c0 + c) * 64);
                    bv[s][c] = *reinterpret_cast<const f32x4v*>(wrow[s] + (c0 + c) * 64);
                }
            }
        }
    };
    load_batch(0);
    // prologue operands of the first PB chunks (covers K = 256): requested now, consumed after the row statistics
    constexpr int PB = 4;
    f32x4v pgam[PB], pbet[PB], px2[S][PB];
    {
        const float* pg = a.ln_gamma ? a.ln_gamma + kbase : gam;
        const float* pb = a.ln_gamma ? a.ln_beta + kbase : bet;
#pragma unroll
        for (int c = 0; c < PB; ++c) {
            if (c < nchunks) {
                if (pg) {
                    pgam[c] = *reinterpret_cast<const f32x4v*>(pg + c * 64);
                    pbet[c] = *reinterpret_cast<const f32x4v*>(pb + c * 64);
                }
                if (add2) {
#pragma unroll
                    for (int s = 0; s < S; ++s) px2[s][c] = *reinterpret_cast<const f32x4v*>(x2row[s] + c * 64);
                }
            }
        }
    }
    // epilogue operands of this thread's (row, OPT cols)
    const int erow = tid / TPR;
    const int ec = (tid % TPR) * OPT;
    const int eom = m0 + erow;
    const bool erow_ok = eom < a.M;
    const float* bias = a.bias ? a.bias + g * a.gBias : nullptr;
    float e_bias[OPT], e_r[OPT], e_rg[OPT], e_rb[OPT];
    float rmean = 0.f, rrstd = 1.f;
#pragma unroll
    for (int e = 0; e < OPT; ++e) {
        e_bias[e] = 0.f; e_r[e] = 0.f; e_rg[e] = 1.f; e_rb[e] = 0.f;
        const int on = n0 + ec + e;
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

    // ---- GroupNorm(1,C) prologue: scene-wide moments accumulated by the producer's epilogue
    // (a tile's rows lie in one scene whenever rows_per_scene % T == 0; otherwise per sub-tile row)
    float gn_mean[S], gn_rstd[S];
    if (a.gn_sums) {
        const double cnt = (double)a.gn_rows_per_scene * (double)a.K;
        const int sc_lo = m0 / a.gn_rows_per_scene;
        const int sc_hi = (m0 + T - 1 < a.M ? m0 + T - 1 : a.M - 1) / a.gn_rows_per_scene;
        if (sc_lo == sc_hi) {
            // the producer left its moments in kGnSlots slots per (scene, group): kGnSlots / 64 per lane, then one shuffle tree
            const double* src = a.gn_sums + ((int64_t)(sc_lo * a.gn_ngroups + g) * kGnSlots + lane) * 2;
            double Sm = 0.0, Qs = 0.0;
#pragma unroll
            for (int i = 0; i < kGnSlots / 64; ++i) { Sm += src[i * 128]; Qs += src[i * 128 + 1]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                Sm += __shfl_xor(Sm, o);
                Qs += __shfl_xor(Qs, o);
            }
            float mean, rstd;
            gn_mean_rstd(Sm, Qs, 1.0 / cnt, a.norm_eps, mean, rstd);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                gn_mean[s] = mean;
                gn_rstd[s] = rstd;
            }
        } else {
            // tile straddles scenes (rows_per_scene not a multiple of the tile): per-row serial sum
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int m = m0 + s * 16 + li;
                const int scene = (m_ok[s] ? m : 0) / a.gn_rows_per_scene;
                double Sm = 0.0, Qs = 0.0;
                for (int sl = 0; sl < kGnSlots; ++sl) {
                    Sm += a.gn_sums[((int64_t)(scene * a.gn_ngroups + g) * kGnSlots + sl) * 2 + 0];
                    Qs += a.gn_sums[((int64_t)(scene * a.gn_ngroups + g) * kGnSlots + sl) * 2 + 1];
                }
                gn_mean_rstd(Sm, Qs, 1.0 / cnt, a.norm_eps, gn_mean[s], gn_rstd[s]);
            }
        }
    }

    // ---- LayerNorm prologue: row statistics of A over the full K, reduced across the 4 kq lane groups
    // and the 4 K-slices through LDS.  One pass over data shifted by the row's first element (the shift
    // removes the cancellation of E[x^2] - mean^2).
    float ln_mean[S], ln_rstd[S];
#pragma unroll
    for (int s = 0; s < S; ++s) { ln_mean[s] = 0.f; ln_rstd[s] = 1.f; }
    if (a.ln_gamma) {
        float sm[S], sq[S], shift[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            shift[s] = X[(int64_t)(m_ok[s] ? m0 + s * 16 + li : 0) * a.ldx];
            sm[s] = 0.f;
            sq[s] = 0.f;
        }
        if (nchunks <= CB) {                           // the lane's whole K share is already in registers
#pragma unroll
            for (int c = 0; c < CB; ++c)
                if (c < nchunks) {
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d = av[s][c][e] - shift[s];
                            sm[s] += d;
                            sq[s] += d * d;
                        }
                }
        } else {
            for (int c = 0; c < nchunks; ++c) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const f32x4v v = *reinterpret_cast<const f32x4v*>(xrow[s] + c * 64);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = v[e] - shift[s];
                        sm[s] += d;
                        sq[s] += d * d;
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            sm[s] += __shfl_xor(sm[s], 16);
            sq[s] += __shfl_xor(sq[s], 16);
            sm[s] += __shfl_xor(sm[s], 32);
            sq[s] += __shfl_xor(sq[s], 32);
            if (kq == 0) {
                red[(wave * T + s * 16 + li) * 2 + 0] = sm[s];
                red[(wave * T + s * 16 + li) * 2 + 1] = sq[s];
            }
        }
        __syncthreads();
        const float invK = 1.f / (float)a.K;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float Ssum = 0.f, Q2 = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                Ssum += red[(w * T + s * 16 + li) * 2 + 0];
                Q2 += red[(w * T + s * 16 + li) * 2 + 1];
            }
            const float dm = Ssum * invK;
            ln_mean[s] = shift[s] + dm;
            const float var = fmaxf(Q2 * invK - dm * dm, 0.f);
            ln_rstd[s] = 1.f / sqrtf(var + a.norm_eps);
            if (a.ln_stats_out && n0 == 0 && wave == 0 && kq == 0 && m_ok[s]) {
                a.ln_stats_out[(int64_t)(m0 + s * 16 + li) * 2 + 0] = ln_mean[s];
                a.ln_stats_out[(int64_t)(m0 + s * 16 + li) * 2 + 1] = ln_rstd[s];
            }
        }
        __syncthreads();
    }
    const float* lng = a.ln_gamma ? a.ln_gamma + kbase : nullptr;
    const float* lnb = a.ln_gamma ? a.ln_beta + kbase : nullptr;

    f32x4v acc[S][S];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int t = 0; t < S; ++t) acc[s][t] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < nchunks; c0 += CB) {
        if (c0 > 0) load_batch(c0);
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            if (c0 + c < nchunks) {
                if (a.ln_gamma) {
                    const bool pre = (c < PB) && (c0 == 0);
                    const f32x4v gv = pre ? pgam[c < PB ? c : 0] : *reinterpret_cast<const f32x4v*>(lng + (c0 + c) * 64);
                    const f32x4v be = pre ? pbet[c < PB ? c : 0] : *reinterpret_cast<const f32x4v*>(lnb + (c0 + c) * 64);
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[s][c][e] = (av[s][c][e] - ln_mean[s]) * ln_rstd[s] * gv[e] + be[e];
                }
                if (add2) {
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        av[s][c] += ((c < PB) && (c0 == 0)) ? px2[s][c < PB ? c : 0] : *reinterpret_cast<const f32x4v*>(x2row[s] + (c0 + c) * 64);
                }
                if (a.gn_sums) {
                    const bool pre = (c < PB) && (c0 == 0) && !a.ln_gamma;
                    const f32x4v gv = pre ? pgam[c < PB ? c : 0] : *reinterpret_cast<const f32x4v*>(gam + (c0 + c) * 64);
                    const f32x4v be = pre ? pbet[c < PB ? c : 0] : *reinterpret_cast<const f32x4v*>(bet + (c0 + c) * 64);
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float y = (av[s][c][e] - gn_mean[s]) * gn_rstd[s] * gv[e] + be[e];
                            av[s][c][e] = y > 0.f ? y : 0.f;
                        }
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (!m_ok[s]) av[s][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    if (!n_ok[s]) bv[s][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int t = 0; t < S; ++t)
                            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][c][e], bv[t][c][e], acc[s][t], 0, 0, 0);
            }
        }
    }

    // in-workgroup split-K reduction: red[wave][sub-tile][reg][lane]; the accumulator of sub-tile (s, t)
    // holds rows 4*(lane>>4) + reg, column lane&15
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int t = 0; t < S; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * S + s) * S + t) * 4 + r) * kWave + lane] = acc[s][t][r];
    __syncthreads();

    // thread -> (row, OPT consecutive cols) of the tile
    float sum[OPT];
    {
        const int s = erow >> 4, rr = erow & 15, t = ec >> 4, cc = ec & 15;
        const int src = (((0 * S + s) * S + t) * 4 + (rr & 3)) * kWave + (rr >> 2) * 16 + cc;
#pragma unroll
        for (int e = 0; e < OPT; ++e) sum[e] = red[src + e];
#pragma unroll
        for (int w = 1; w < kWaves; ++w)
#pragma unroll
            for (int e = 0; e < OPT; ++e) sum[e] += red[src + w * S * S * 4 * kWave + e];
    }

    const int om = eom;
    const bool row_ok = erow_ok;
    double gs = 0.0, gq = 0.0;                       // moments of this thread's outputs (GroupNorm of the output)
    const bool gn_out = a.gn_out_sums != nullptr && n0 < a.gn_out_ncols;
    if (row_ok) {
        float* Ybase = a.Y + g * a.gY + (int64_t)(om / a.rows_per_batch) * a.y_batch +
                       (int64_t)(om % a.rows_per_batch) * a.y_row;
#pragma unroll
        for (int e = 0; e < OPT; ++e) {
            const int on = n0 + ec + e;
            if (on < a.N) {
                float y = sum[e] + e_bias[e];
                if (a.relu) y = y > 0.f ? y : 0.f;
                if (a.relu_mask) y = a.relu_mask[(int64_t)om * a.ldmask + on] > 0.f ? y * (a.mask_scale != 0.f ? a.mask_scale : 1.f) : 0.f;
                if (a.drop_p > 0.f) y = drop_keep(drop_rowhash(a.drop_seed, (uint32_t)om), (uint32_t)on, a.drop_p) ? y / (1.f - a.drop_p) : 0.f;
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
        const int last_row = (m0 + T - 1 < a.M ? m0 + T - 1 : a.M - 1);
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
                // slot = tile index mod kGnSlots: a handful of workgroups contend on one address
                const int slot = (int)(blockIdx.x % kGnSlots);
                double* dst = a.gn_out_sums + (((int64_t)sc_first * a.gn_out_ngroups + grp) * kGnSlots + slot) * 2;
                atomicAdd(dst, S);
                atomicAdd(dst + 1, Q2);
            }
        } else if (row_ok) {
            const int slot = (int)(blockIdx.x % kGnSlots);
            double* dst = a.gn_out_sums + (((int64_t)(om / a.gn_out_rows_per_scene) * a.gn_out_ngroups + grp) * kGnSlots + slot) * 2;
            atomicAdd(dst, gs);
            atomicAdd(dst + 1, gq);
        }
    }
}

}  // namespace

// tile edge: 16 while the 32x32 tiling would leave CUs idle (the latency-bound single-scene case)
static int pick_tile(int64_t tiles32) {
    static const int forced = [] {
        const char* e = dev_env("PARQ_LINEAR_TILE");
        return e ? atoi(e) : 0;
    }();
    if (forced == 16 || forced == 32) return forced;
    static const int below = [] {
        const char* e = dev_env("PARQ_LINEAR_T16_BELOW");
        return e ? atoi(e) : 0;
    }();
    return tiles32 < (below > 0 ? below : device_num_cus()) ? 16 : 32;
}

hipError_t launch_linear(const LinearArgs& a, int groups, hipStream_t s) {
    if (a.K % 16 != 0 || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
    // the latency-bound regime (one or a few scenes): compile-time specialised kernels of the decoder chain (chain.hip)
    {
        const hipError_t e = launch_chain_linear(a, groups, s);
        if (e != hipErrorNotSupported) return e;
    }
    if (a.W2) return hipErrorInvalidValue;          // a second operand pair exists in chain.hip only (callers test chain_linear_supported)
    const int64_t tiles32 = (int64_t)ceil_div(a.N, 32) * ceil_div(a.M, 32) * groups;
    const int T = pick_tile(tiles32);
    const int64_t tiles = (int64_t)ceil_div(a.N, T) * ceil_div(a.M, T);
    if (tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)tiles, groups, 1);
    const int wave_chunks = ceil_div(a.K / 16, kWaves);       // whole K share in registers when it fits
    if (T == 32)
        hipLaunchKernelGGL((linear_f32_kernel<32, 4>), grid, dim3(kWaves * kWave), 0, s, a);
    else if (wave_chunks <= 4)
        hipLaunchKernelGGL((linear_f32_kernel<16, 4>), grid, dim3(kWaves * kWave), 0, s, a);
    else if (wave_chunks <= 6)
        hipLaunchKernelGGL((linear_f32_kernel<16, 6>), grid, dim3(kWaves * kWave), 0, s, a);
    else
        hipLaunchKernelGGL((linear_f32_kernel<16, 12>), grid, dim3(kWaves * kWave), 0, s, a);
    return hipGetLastError();
}

PARQ_TL_DEFINE_SETTER(tl_set_linear)

}  // namespace parq
