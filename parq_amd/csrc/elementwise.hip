// Gather / elementwise / reduction kernels of the PARQ decoder chain (gfx950).
//
//   camera_local    T_camera_local = T_cp ∘ (inv(T_wp) ∘ T_wl)          transformer_parq.py:298-300
//   initial_ref     sigmoid(refpoint.weight) tiled over scenes           transformer_parq.py:122,309
//   posemb          384-d sine embedding of the reference points         transformer_parq.py:45-64
//   project_sample  3D->2D projection + bilinear gather + view mean      transformer_parq.py:129-161
//   layernorm       post-norm LayerNorm                                  transformer_parq.py:354-356
//   gn_stats        GroupNorm(1,C) statistics over a whole scene         generic_mlp.py:85-86
//   box_decode      softmax / arg-max size gather / centre update        transformer_parq.py:242-279
//
// All of these are HBM- or latency-bound: wide coalesced loads, one wave per row
// (or per query-view pair), wavefront/LDS reductions, no MFMA.
#include "common.hpp"
#include "sample_body.hpp"
#include <cstdlib>
#include <cstring>

namespace parq {

namespace {

// ---------------------------------------------------------------- SE(3) on 12-vectors
// Geometry is evaluated in float64 from the float32 inputs.  The reference does these 3x3
// products through torch matmul in float32, whose rounding order is a BLAS detail; pixel
// coordinates amplify a 1-ulp pose difference to ~1e-5 px, which white-noise features turn
// into ~5e-5 output noise.  Computing the geometry (near-)exactly keeps OUR contribution
// to that noise at zero; it costs a few dozen fp64 flops per (query, view).
struct Pose12 { double R[9]; double t[3]; };

__device__ __forceinline__ Pose12 load_pose(const float* p) {
    Pose12 o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.R[i] = (double)p[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) o.t[i] = (double)p[9 + i];
    return o;
}

// utils/wrappers.py:247-251
__device__ __forceinline__ Pose12 pose_inverse(const Pose12& a) {
    Pose12 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = a.R[j * 3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o.t[i] = -(o.R[i * 3 + 0] * a.t[0] + o.R[i * 3 + 1] * a.t[1] + o.R[i * 3 + 2] * a.t[2]);
    return o;
}

// utils/wrappers.py:253-257
__device__ __forceinline__ Pose12 pose_compose(const Pose12& a, const Pose12& b) {
    Pose12 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.R[i * 3 + j] = a.R[i * 3 + 0] * b.R[0 * 3 + j] + a.R[i * 3 + 1] * b.R[1 * 3 + j] +
                             a.R[i * 3 + 2] * b.R[2 * 3 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o.t[i] = a.t[i] + (a.R[i * 3 + 0] * b.t[0] + a.R[i * 3 + 1] * b.t[1] + a.R[i * 3 + 2] * b.t[2]);
    return o;
}

template <typename TOut>
__global__ void camera_local_kernel(const float* T_cp, const float* T_wp, const float* T_wl, int B, int V,
                                    TOut* T_cl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * V) return;
    const int b = i / V;
    const Pose12 cp = load_pose(T_cp + (int64_t)i * 12);
    const Pose12 wp = load_pose(T_wp + (int64_t)i * 12);
    const Pose12 wl = load_pose(T_wl + (int64_t)b * 12);
    const Pose12 o = pose_compose(cp, pose_compose(pose_inverse(wp), wl));
#pragma unroll
    for (int k = 0; k < 9; ++k) T_cl[(int64_t)i * 12 + k] = (TOut)o.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) T_cl[(int64_t)i * 12 + 9 + k] = (TOut)o.t[k];
}

__global__ void initial_ref_kernel(const float* w, int B, int Q, float* ref) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Q * 3) return;
    const float x = w[i % (Q * 3)];
    ref[i] = 1.f / (1.f + expf(-x));
}

// ---------------------------------------------------------------- sine embedding
// emb[m][blk*128 + i], blk order (y, x, z); a = ref*2pi / dim_t[i]; even i -> sin, odd i -> cos.
__global__ void posemb_kernel(const float* ref, const float* dim_t, int M, float* emb) {
    PARQ_TL_KERNEL(kTlPosemb);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * 384) return;
    const int m = idx / 384;
    const int k = idx - m * 384;
    const int blk = k >> 7;
    const int i = k & 127;
    const int axis = blk == 0 ? 1 : (blk == 1 ? 0 : 2);
    const float p = ref[m * 3 + axis] * 6.283185307179586f;
    const float a = p / dim_t[i];
    emb[idx] = (i & 1) ? cosf(a) : sinf(a);
}

// ---------------------------------------------------------------- forward prologue in one launch
// T_camera_local of every (scene, view) in float64, the initial reference points sigmoid(refpoint.weight) tiled over the scenes,
// their sine embedding (what posemb_kernel would compute from them: same arithmetic) and the cleared range flags: four tiny
// launches (~5 us each behind a dependent boundary) as one.  Thread i plays every role its index is in range for.
__global__ void forward_prologue_kernel(const float* T_cp, const float* T_wp, const float* T_wl, int B, int V, double* T_cl,
                                        const float* w, int Q, float* ref, const float* dim_t, float* emb, float* flags, int nflags,
                                        PrologueCall call) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nflags) flags[i] = 0.f;
    if (call.ind != nullptr) {                     // a captured forward: this call's pointers and cameras for the recorded iterations
        if (i < 8) call.ind[i] = call.ptrs.p[i];
        if (i < call.ncam) call.cam_dst[i] = call.cam_src[i];
    }
    if (i < B * V) {
        const int b = i / V;
        const Pose12 cp = load_pose(T_cp + (int64_t)i * 12);
        const Pose12 wp = load_pose(T_wp + (int64_t)i * 12);
        const Pose12 wl = load_pose(T_wl + (int64_t)b * 12);
        const Pose12 o = pose_compose(cp, pose_compose(pose_inverse(wp), wl));
#pragma unroll
        for (int k = 0; k < 9; ++k) T_cl[(int64_t)i * 12 + k] = (double)o.R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) T_cl[(int64_t)i * 12 + 9 + k] = (double)o.t[k];
    }
    if (i < B * Q * 3) ref[i] = 1.f / (1.f + expf(-w[i % (Q * 3)]));
    if (i < B * Q * 384) {
        const int m = i / 384;
        const int k = i - m * 384;
        const int blk = k >> 7;
        const int j = k & 127;
        const int axis = blk == 0 ? 1 : (blk == 1 ? 0 : 2);
        const float r = 1.f / (1.f + expf(-w[(m % Q) * 3 + axis]));
        const float p = r * 6.283185307179586f;
        const float a = p / dim_t[j];
        emb[i] = (j & 1) ? cosf(a) : sinf(a);
    }
}

// ---------------------------------------------------------------- project + sample
// One workgroup per (scene, query); wave wv handles views wv, wv+nwv, ...; each lane owns
// float4 channel groups, so one texel row (C floats, contiguous in the channels-last stack)
// is a single fully coalesced wave load.  Views are reduced through LDS in a fixed order.
constexpr int kMaxChunks = 4;   // C <= 1024

template <int NCH, typename TPose>
__global__ __launch_bounds__(1024) void project_sample_kernel(
    const float* __restrict__ tokens, const TPose* __restrict__ T_cl, const float* __restrict__ cam,
    const float* __restrict__ ref, ScaleBox sb, int V, int h, int w, int C, int Q, float* __restrict__ tgt,
    float* __restrict__ coord_pos, double* __restrict__ zero_f64, int zero_n, float* __restrict__ raw_count,
    const void* const* __restrict__ ind, int64_t coord_off) {
    PARQ_TL_KERNEL(kTlProjectSample);
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [nwv][C] + [nwv] counts + footprints
    if (ind != nullptr) {                                           // a captured forward: this call's pointers (CallPtrs; wave-uniform loads)
        tokens = reinterpret_cast<const float*>(ind[0]);
        coord_pos = reinterpret_cast<float*>(const_cast<void*>(ind[6])) + coord_off;
    }
    project_sample_body<NCH, TPose>(tokens, T_cl, cam, ref, sb, V, h, w, C, Q, tgt, coord_pos, zero_f64, zero_n, raw_count,
                                    (int)blockIdx.x, (int)gridDim.x, smem);
}

// ---------------------------------------------------------------- LayerNorm: one wave per row
constexpr int kLnMaxPerLane = 16;   // C <= 1024

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ Y,
                                                        int M, int C, float eps) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const float* x = X + (int64_t)row * C;
    float v[kLnMaxPerLane];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int c = lane + i * 64;
        v[i] = c < C ? x[c] : 0.f;
        s += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int c = lane + i * 64;
        const float d = c < C ? v[i] - mean : 0.f;
        q += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q / (float)C + eps);
    float* y = Y + (int64_t)row * C;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int c = lane + i * 64;
        if (c < C) y[c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
}

// ---------------------------------------------------------------- GroupNorm(1,C) statistics
// grid (ngroups, B): one workgroup reduces rows_per_scene x ncols elements (double accumulators).
__global__ __launch_bounds__(1024) void gn_stats_kernel(const float* __restrict__ X, int64_t ldx, int col0,
                                                         int ncols, int ngroups, int rows_per_scene, float eps,
                                                         float* __restrict__ stats) {
    __shared__ double sh[2][16];
    const int g = blockIdx.x;
    const int b = blockIdx.y;
    const float* base = X + (int64_t)b * rows_per_scene * ldx + col0 + (int64_t)g * ncols;
    const int n4 = ncols >> 2;
    double s = 0.0, q = 0.0;
    const int64_t total4 = (int64_t)rows_per_scene * n4;
    for (int64_t i = threadIdx.x; i < total4; i += blockDim.x) {
        const int r = (int)(i / n4);
        const int c4 = (int)(i - (int64_t)r * n4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)r * ldx + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s += (double)v[e];
            q += (double)v[e] * (double)v[e];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[0][wv] = s;
        sh[1][wv] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0.0, Qs = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
            S += sh[0][i];
            Qs += sh[1][i];
        }
        const double n = (double)rows_per_scene * (double)ncols;
        float mean, rstd;
        gn_mean_rstd(S, Qs, 1.0 / n, eps, mean, rstd);
        stats[((int64_t)b * ngroups + g) * 2 + 0] = mean;
        stats[((int64_t)b * ngroups + g) * 2 + 1] = rstd;
    }
}

// ---------------------------------------------------------------- heads' last layer + box decode + next reference point
constexpr int kMaxCls = 32;

__global__ __launch_bounds__(256) void box_decode_kernel(BoxDecodeArgs a) {
    if (a.ind != nullptr) {                        // a captured forward: this call's output pointers (CallPtrs; wave-uniform loads)
        a.logits = reinterpret_cast<float*>(const_cast<void*>(a.ind[1])) + a.out_row0 * a.ncls;
        a.center = reinterpret_cast<float*>(const_cast<void*>(a.ind[2])) + a.out_row0 * 3;
        a.size = reinterpret_cast<float*>(const_cast<void*>(a.ind[3])) + a.out_row0 * 3;
        a.rot = reinterpret_cast<float*>(const_cast<void*>(a.ind[4])) + a.out_row0 * 6;
        a.prob = reinterpret_cast<float*>(const_cast<void*>(a.ind[5])) + a.out_row0 * a.ncls;
    }
    PARQ_TL_KERNEL(kTlBoxDecode);
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int lane = threadIdx.x & 63;
    const int C = a.C;
    const int scene = m / a.rows_per_scene;
    const int range_word = a.poison != nullptr ? (*a.poison & a.poison_mask) : 0;         // wave-uniform scalar loads.  Bit 2: an in-launch hand-off timed out
    const bool range_poison = range_word != 0;
    const int peaky = a.peaky != nullptr ? *a.peaky : 0;
    const bool poison = range_poison || (a.peaky_poison && peaky != 0);
    // tell the host (pinned word): bit 0 range violation, bit 2 hand-off timeout, bit 1 + bits 8.. the heads attention mode 4 met
    // too-peaked rows on
    if (a.poison_mirror != nullptr && m == 0 && lane == 0 && (range_poison || peaky != 0))
        atomicOr(a.poison_mirror, ((range_word & ~4) ? 1 : 0) | (range_word & 4) | (peaky != 0 ? (2 | (peaky << 8)) : 0));
    // every independent load is issued before the first dependent use (this kernel is pure latency)
    const float* h1 = a.h1 + (int64_t)m * a.ld1;
    const float lg_in = lane < a.ncls ? h1[lane] : -INFINITY;
    const float sz_in = lane < 3 ? h1[a.ncls + lane] : 0.f;
    const float ref_in = lane < 3 ? a.ref[(int64_t)m * 3 + lane] : 0.5f;
    const float b3c = lane < 3 ? a.b3[lane] : 0.f;
    const float b3r = lane < 6 ? a.b3[6 + lane] : 0.f;
    // scene-wide GroupNorm moments: kGnSlots slots per (scene, head), kGnSlots / 64 per lane
    double sums[4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const double* src = a.gn_sums + ((int64_t)(scene * 2 + g) * kGnSlots + lane) * 2;
        sums[2 * g] = 0.0;
        sums[2 * g + 1] = 0.0;
#pragma unroll
        for (int i = 0; i < kGnSlots / 64; ++i) { sums[2 * g] += src[i * 128]; sums[2 * g + 1] += src[i * 128 + 1]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sums[j] += __shfl_xor(sums[j], o);
    // GroupNorm(1,C) of the second hidden layer from the scene-wide moments (generic_mlp.py:85-86)
    float mean[2], rstd[2];
    const double inv_cnt = 1.0 / ((double)a.rows_per_scene * (double)C);
#pragma unroll
    for (int g = 0; g < 2; ++g) gn_mean_rstd(sums[2 * g], sums[2 * g + 1], inv_cnt, a.eps, mean[g], rstd[g]);
    // output layers: centre (3 rows of group 0) and rotation (6 rows of group 1), K = C each
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
    const float* h2 = a.h2 + (int64_t)m * a.ld2;
#pragma unroll 4
    for (int c = lane; c < C; c += 64) {
        float y0 = (h2[c] - mean[0]) * rstd[0] * a.gn_gamma[c] + a.gn_beta[c];
        float y1 = (h2[C + c] - mean[1]) * rstd[1] * a.gn_gamma[C + c] + a.gn_beta[C + c];
        y0 = y0 > 0.f ? y0 : 0.f;
        y1 = y1 > 0.f ? y1 : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] += y0 * a.w3[(int64_t)j * C + c];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[3 + j] += y1 * a.w3[(int64_t)(6 + j) * C + c];
    }
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);

    // class probabilities: softmax over num_classes, one class per lane (utils/parq_utils.py:101-105)
    float mx = lg_in;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float ex = lane < a.ncls ? expf(lg_in - mx) : 0.f;
    float sum = ex;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float prob = ex / sum;
    if (lane < a.ncls) {
        a.logits[(int64_t)m * a.ncls + lane] = poison ? NAN : lg_in;
        a.prob[(int64_t)m * a.ncls + lane] = poison ? NAN : prob;
    }
    // arg-max with torch.argmax tie-breaking (first maximum)
    float bestp = lane < a.ncls ? prob : -1.f;
    int besti = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float op = __shfl_xor(bestp, o);
        const int oi = __shfl_xor(besti, o);
        if (op > bestp || (op == bestp && oi < besti)) {
            bestp = op;
            besti = oi;
        }
    }
    int arg = besti < a.n_mean ? besti : a.n_mean - 1;
    // size = exp(size_scale) * mean_size[argmax]  (utils/parq_utils.py:94-99)
    if (lane < 3) a.size[(int64_t)m * 3 + lane] = poison ? NAN : expf(sz_in) * a.mean_sizes[arg * 3 + lane];
    // ortho6d: lane j picks its own reduced dot product
    float rotv = 0.f, ctrv = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) rotv = lane == j ? acc[3 + j] : rotv;
#pragma unroll
    for (int j = 0; j < 3; ++j) ctrv = lane == j ? acc[j] : ctrv;
    if (lane < 6) a.rot[(int64_t)m * 6 + lane] = poison ? NAN : rotv + b3r;
    // centre = denorm(sigmoid(offset + inverse_sigmoid(ref)))  (transformer_parq.py:242-245, 38-42)
    float nref = 0.f;
    if (lane < 3) {
        const float lo = lane == 0 ? a.sb.lo[0] : (lane == 1 ? a.sb.lo[1] : a.sb.lo[2]);
        const float hi = lane == 0 ? a.sb.hi[0] : (lane == 1 ? a.sb.hi[1] : a.sb.hi[2]);
        const float r = fminf(fmaxf(ref_in, 0.f), 1.f);
        const float x1 = fmaxf(r, 1e-3f);
        const float x2 = fmaxf(1.f - r, 1e-3f);
        const float off = (ctrv + b3c) + logf(x1 / x2);
        const float sg = 1.f / (1.f + expf(-off));
        const float ctr = __fadd_rn(__fmul_rn(sg, __fsub_rn(hi, lo)), lo);      // mul, then add: as torch
        a.center[(int64_t)m * 3 + lane] = poison ? NAN : ctr;
        // next reference point = normalize(centre), detached (transformer_parq.py:331-332)
        nref = __fsub_rn(ctr, lo) / __fsub_rn(hi, lo);
        if (a.ref_next) a.ref_next[(int64_t)m * 3 + lane] = nref;
    }
    // sine embedding of the next reference point for the next iteration's position MLP
    if (a.emb_next) {
        const float n0 = __shfl(nref, 0), n1 = __shfl(nref, 1), n2 = __shfl(nref, 2);
        // entries 2 p and 2 p + 1 of an axis block are sin and cos of the SAME angle (dim_t[2p] == dim_t[2p+1]): one sincosf per pair
#pragma unroll
        for (int blk = 0; blk < 3; ++blk) {
            const float r = blk == 0 ? n1 : (blk == 1 ? n0 : n2);
            const float ang = (r * 6.283185307179586f) / a.dim_t[2 * lane];
            float sn, cs;
            sincosf(ang, &sn, &cs);
            *reinterpret_cast<float2*>(a.emb_next + (int64_t)m * 384 + blk * 128 + 2 * lane) = float2{sn, cs};
        }
    }
}

// The same decode for C = 256 with every global load requested before the first wait (the generic kernel above reads the weights
// inside a strided loop, the mean-size row after the arg-max and the frequency table at the end: four dependent round trips in a
// launch that is pure latency).  One wave per query row; lane l owns channels 4 l .. 4 l + 3 of both hidden blocks (float4 loads);
// the mean-size table sits in two registers per lane and the arg-max row is fetched with a lane permute.
__global__ __launch_bounds__(256) void box_decode256_kernel(BoxDecodeArgs a) {
    if (a.ind != nullptr) {                        // a captured forward: this call's output pointers (CallPtrs; wave-uniform loads)
        a.logits = reinterpret_cast<float*>(const_cast<void*>(a.ind[1])) + a.out_row0 * a.ncls;
        a.center = reinterpret_cast<float*>(const_cast<void*>(a.ind[2])) + a.out_row0 * 3;
        a.size = reinterpret_cast<float*>(const_cast<void*>(a.ind[3])) + a.out_row0 * 3;
        a.rot = reinterpret_cast<float*>(const_cast<void*>(a.ind[4])) + a.out_row0 * 6;
        a.prob = reinterpret_cast<float*>(const_cast<void*>(a.ind[5])) + a.out_row0 * a.ncls;
    }
    PARQ_TL_KERNEL(kTlBoxDecode);
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    constexpr int C = 256;
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int lane = threadIdx.x & 63;
    const int scene = m / a.rows_per_scene;
    const float* h1 = a.h1 + (int64_t)m * a.ld1;
    const float* h2 = a.h2 + (int64_t)m * a.ld2;
    const int l3 = lane < 3 ? lane : 2, l6 = lane < 6 ? lane : 5, lc = lane < a.ncls ? lane : a.ncls - 1;
    const float lg_raw = h1[lc];
    const float sz_raw = h1[a.ncls + l3];
    const float ref_raw = a.ref[(int64_t)m * 3 + l3];
    const float b3c_raw = a.b3[l3];
    const float b3r_raw = a.b3[6 + l6];
    double sums[4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const double* src = a.gn_sums + ((int64_t)(scene * 2 + g) * kGnSlots + lane) * 2;
        sums[2 * g] = 0.0;
        sums[2 * g + 1] = 0.0;
#pragma unroll
        for (int i = 0; i < kGnSlots / 64; ++i) { sums[2 * g] += src[i * 128]; sums[2 * g + 1] += src[i * 128 + 1]; }
    }
    const f32x4v x0 = *reinterpret_cast<const f32x4v*>(h2 + 4 * lane);
    const f32x4v x1 = *reinterpret_cast<const f32x4v*>(h2 + C + 4 * lane);
    const f32x4v g0 = *reinterpret_cast<const f32x4v*>(a.gn_gamma + 4 * lane), g1 = *reinterpret_cast<const f32x4v*>(a.gn_gamma + C + 4 * lane);
    const f32x4v be0 = *reinterpret_cast<const f32x4v*>(a.gn_beta + 4 * lane), be1 = *reinterpret_cast<const f32x4v*>(a.gn_beta + C + 4 * lane);
    f32x4v w[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) w[j] = *reinterpret_cast<const f32x4v*>(a.w3 + (int64_t)j * C + 4 * lane);
#pragma unroll
    for (int j = 0; j < 6; ++j) w[3 + j] = *reinterpret_cast<const f32x4v*>(a.w3 + (int64_t)(6 + j) * C + 4 * lane);
    const float dimt = a.dim_t[2 * lane];
    const int nms = a.n_mean * 3;                                       // <= 128 (launcher)
    const float ms0 = a.mean_sizes[lane < nms ? lane : nms - 1];
    const float ms1 = a.mean_sizes[64 + lane < nms ? 64 + lane : nms - 1];
    __builtin_amdgcn_sched_barrier(0);                                   // keep every load above the first wait
    const int range_word = a.poison != nullptr ? (*a.poison & a.poison_mask) : 0;         // wave-uniform scalar loads.  Bit 2: an in-launch hand-off timed out
    const bool range_poison = range_word != 0;
    const int peaky = a.peaky != nullptr ? *a.peaky : 0;
    const bool poison = range_poison || (a.peaky_poison && peaky != 0);
    // tell the host (pinned word): bit 0 range violation, bit 2 hand-off timeout, bit 1 + bits 8.. the heads attention mode 4 met
    // too-peaked rows on
    if (a.poison_mirror != nullptr && m == 0 && lane == 0 && (range_poison || peaky != 0))
        atomicOr(a.poison_mirror, ((range_word & ~4) ? 1 : 0) | (range_word & 4) | (peaky != 0 ? (2 | (peaky << 8)) : 0));

    const float lg_in = lane < a.ncls ? lg_raw : -INFINITY;
    const float sz_in = lane < 3 ? sz_raw : 0.f;
    const float ref_in = lane < 3 ? ref_raw : 0.5f;
    const float b3c = lane < 3 ? b3c_raw : 0.f;
    const float b3r = lane < 6 ? b3r_raw : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sums[j] += __shfl_xor(sums[j], o);
    float mean[2], rstd[2];
    const double inv_cnt = 1.0 / ((double)a.rows_per_scene * (double)C);
#pragma unroll
    for (int g = 0; g < 2; ++g) gn_mean_rstd(sums[2 * g], sums[2 * g + 1], inv_cnt, a.eps, mean[g], rstd[g]);
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float y0 = (x0[e] - mean[0]) * rstd[0] * g0[e] + be0[e];
        float y1 = (x1[e] - mean[1]) * rstd[1] * g1[e] + be1[e];
        y0 = y0 > 0.f ? y0 : 0.f;
        y1 = y1 > 0.f ? y1 : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] += y0 * w[j][e];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[3 + j] += y1 * w[3 + j][e];
    }
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);

    float mx = lg_in;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float ex = lane < a.ncls ? expf(lg_in - mx) : 0.f;
    float sum = ex;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float prob = ex / sum;
    if (lane < a.ncls) {
        a.logits[(int64_t)m * a.ncls + lane] = poison ? NAN : lg_in;
        a.prob[(int64_t)m * a.ncls + lane] = poison ? NAN : prob;
    }
    float bestp = lane < a.ncls ? prob : -1.f;
    int besti = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float op = __shfl_xor(bestp, o);
        const int oi = __shfl_xor(besti, o);
        if (op > bestp || (op == bestp && oi < besti)) {
            bestp = op;
            besti = oi;
        }
    }
    const int arg = besti < a.n_mean ? besti : a.n_mean - 1;
    {
        const int idx = arg * 3 + l3;                                     // row `arg` of the mean-size table, from the lanes that hold it
        const float lo_half = __shfl(ms0, idx & 63), hi_half = __shfl(ms1, idx & 63);
        const float msz = idx < 64 ? lo_half : hi_half;
        if (lane < 3) a.size[(int64_t)m * 3 + lane] = poison ? NAN : expf(sz_in) * msz;
    }
    float rotv = 0.f, ctrv = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) rotv = lane == j ? acc[3 + j] : rotv;
#pragma unroll
    for (int j = 0; j < 3; ++j) ctrv = lane == j ? acc[j] : ctrv;
    if (lane < 6) a.rot[(int64_t)m * 6 + lane] = poison ? NAN : rotv + b3r;
    float nref = 0.f;
    if (lane < 3) {
        const float lo = lane == 0 ? a.sb.lo[0] : (lane == 1 ? a.sb.lo[1] : a.sb.lo[2]);
        const float hi = lane == 0 ? a.sb.hi[0] : (lane == 1 ? a.sb.hi[1] : a.sb.hi[2]);
        const float r = fminf(fmaxf(ref_in, 0.f), 1.f);
        const float x1c = fmaxf(r, 1e-3f);
        const float x2c = fmaxf(1.f - r, 1e-3f);
        const float off = (ctrv + b3c) + logf(x1c / x2c);
        const float sg = 1.f / (1.f + expf(-off));
        const float ctr = __fadd_rn(__fmul_rn(sg, __fsub_rn(hi, lo)), lo);      // mul, then add: as torch
        a.center[(int64_t)m * 3 + lane] = poison ? NAN : ctr;
        nref = __fsub_rn(ctr, lo) / __fsub_rn(hi, lo);
        if (a.ref_next) a.ref_next[(int64_t)m * 3 + lane] = nref;
    }
    if (a.emb_next) {
        const float n0 = __shfl(nref, 0), n1 = __shfl(nref, 1), n2 = __shfl(nref, 2);
#pragma unroll
        for (int blk = 0; blk < 3; ++blk) {
            const float r = blk == 0 ? n1 : (blk == 1 ? n0 : n2);
            const float ang = (r * 6.283185307179586f) / dimt;
            float sn, cs;
            sincosf(ang, &sn, &cs);
            *reinterpret_cast<float2*>(a.emb_next + (int64_t)m * 384 + blk * 128 + 2 * lane) = float2{sn, cs};
        }
    }
}

__global__ void zero_f64_kernel(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

__global__ void copy_rows_kernel(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int rows, int cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols);
    const int c = (int)(i - (int64_t)r * cols);
    dst[(int64_t)r * dst_ld + c] = src[(int64_t)r * src_ld + c];
}

__global__ void fill_kernel(float* dst, float value, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

}  // namespace

hipError_t launch_camera_local(const float* T_cp, const float* T_wp, const float* T_wl, int B, int V, float* T_cl,
                               hipStream_t s) {
    hipLaunchKernelGGL(camera_local_kernel<float>, dim3(ceil_div(B * V, 64)), dim3(64), 0, s, T_cp, T_wp, T_wl, B, V, T_cl);
    return hipGetLastError();
}

hipError_t launch_camera_local_f64(const float* T_cp, const float* T_wp, const float* T_wl, int B, int V, double* T_cl,
                                   hipStream_t s) {
    hipLaunchKernelGGL(camera_local_kernel<double>, dim3(ceil_div(B * V, 64)), dim3(64), 0, s, T_cp, T_wp, T_wl, B, V, T_cl);
    return hipGetLastError();
}

hipError_t launch_initial_ref(const float* w, int B, int Q, float* ref, hipStream_t s) {
    hipLaunchKernelGGL(initial_ref_kernel, dim3(ceil_div(B * Q * 3, 256)), dim3(256), 0, s, w, B, Q, ref);
    return hipGetLastError();
}

// ------------------------------------------------------------------ many small device-to-device copies in one launch
// blockIdx.y = copy, blockIdx.x strides over its floats (16-byte pieces where both pointers allow)
__global__ __launch_bounds__(256) void gather_copy_kernel(GatherArgs g) {
    const int e = blockIdx.y;
    const float* __restrict__ src = g.src[e];
    float* __restrict__ dst = g.dst[e];
    const int64_t n = g.n[e];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = t; i < n4; i += nt) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        for (int64_t i = (n4 << 2) + t; i < n; i += nt) dst[i] = src[i];
    } else {
        for (int64_t i = t; i < n; i += nt) dst[i] = src[i];
    }
}
hipError_t launch_gather_copy(const GatherArgs& g, hipStream_t s) {
    if (g.count <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_copy_kernel, dim3(32, g.count), dim3(256), 0, s, g);
    return hipGetLastError();
}

hipError_t launch_forward_prologue(const float* T_cp, const float* T_wp, const float* T_wl, int B, int V, double* T_cl, const float* w, int Q,
                                   float* ref, const float* dim_t, float* emb, float* flags, int nflags, hipStream_t s, const PrologueCall* call) {
    int n = B * Q * 384;
    if (B * V > n) n = B * V;
    if (nflags > n) n = nflags;
    PrologueCall pc;
    memset(&pc, 0, sizeof(pc));
    if (call) pc = *call;
    if (pc.ncam > n) n = pc.ncam;
    hipLaunchKernelGGL(forward_prologue_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, T_cp, T_wp, T_wl, B, V, T_cl, w, Q, ref, dim_t, emb,
                       flags, nflags, pc);
    return hipGetLastError();
}

hipError_t launch_posemb(const float* ref, const float* dim_t, int M, float* emb, hipStream_t s) {
    hipLaunchKernelGGL(posemb_kernel, dim3(ceil_div(M * 384, 256)), dim3(256), 0, s, ref, dim_t, M, emb);
    return hipGetLastError();
}

template <typename TPose>
static hipError_t launch_project_sample_t(const float* tokens, const TPose* T_cl, const float* cam, const float* ref,
                                          ScaleBox sb, int B, int V, int h, int w, int C, int Q, float* tgt,
                                          float* coord_pos, double* zero_f64, int zero_n, hipStream_t s, float* raw_count = nullptr,
                                          const void* const* ind = nullptr, int64_t coord_off = 0) {
    if (C % 4 != 0 || C > 256 * kMaxChunks || V <= 0) return hipErrorInvalidValue;
    const int nwv = V < 16 ? V : 16;
    const size_t smem = (size_t)nwv * C * sizeof(float) + (size_t)nwv * sizeof(int) + (size_t)V * 32 + 16;      // partial sums, counts, footprints
    const int nch = ceil_div(C / 4, 64);
    dim3 grid(B * Q), block(nwv * 64);
    switch (nch) {
        case 1: hipLaunchKernelGGL((project_sample_kernel<1, TPose>), grid, block, smem, s, tokens, T_cl, cam, ref, sb, V, h, w, C, Q, tgt, coord_pos, zero_f64, zero_n, raw_count, ind, coord_off); break;
        case 2: hipLaunchKernelGGL((project_sample_kernel<2, TPose>), grid, block, smem, s, tokens, T_cl, cam, ref, sb, V, h, w, C, Q, tgt, coord_pos, zero_f64, zero_n, raw_count, ind, coord_off); break;
        case 3: hipLaunchKernelGGL((project_sample_kernel<3, TPose>), grid, block, smem, s, tokens, T_cl, cam, ref, sb, V, h, w, C, Q, tgt, coord_pos, zero_f64, zero_n, raw_count, ind, coord_off); break;
        default: hipLaunchKernelGGL((project_sample_kernel<4, TPose>), grid, block, smem, s, tokens, T_cl, cam, ref, sb, V, h, w, C, Q, tgt, coord_pos, zero_f64, zero_n, raw_count, ind, coord_off); break;
    }
    return hipGetLastError();
}

hipError_t launch_project_sample(const float* tokens, const float* T_cl, const float* cam, const float* ref,
                                 ScaleBox sb, int B, int V, int h, int w, int C, int Q, float* tgt,
                                 float* coord_pos, hipStream_t s) {
    return launch_project_sample_t<float>(tokens, T_cl, cam, ref, sb, B, V, h, w, C, Q, tgt, coord_pos, nullptr, 0, s);
}

hipError_t launch_project_sample_f64(const float* tokens, const double* T_cl, const float* cam, const float* ref,
                                     ScaleBox sb, int B, int V, int h, int w, int C, int Q, float* tgt,
                                     float* coord_pos, double* zero_f64, int zero_n, hipStream_t s, float* raw_count,
                                     const void* const* ind, int64_t coord_off) {
    return launch_project_sample_t<double>(tokens, T_cl, cam, ref, sb, B, V, h, w, C, Q, tgt, coord_pos, zero_f64, zero_n, s, raw_count, ind, coord_off);
}

// ---- view-sharded scenes (parq_iterate_sharded): the two merges around the exchanges
// tgt[m][c] = sum[m][c] / max(count[m], 1): the ranks' sample sums and valid-view counts were added by the caller's all-reduce
// range_sum (the float behind the counts): the ranks' fp16-range flags added up by the same all-reduce.  A violation on ANY rank
// poisons the shard it came from, so every rank raises its own device flag (the last kernel of the iteration then writes NaN
// outputs and raises the host mirror): all ranks return NaN and take the same fallback, never unflagged numbers.
__global__ void sample_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ counts, int64_t M, int C, float* __restrict__ tgt,
                                       const float* __restrict__ range_sum, int* __restrict__ range_flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && range_sum != nullptr && range_flag != nullptr && *range_sum != 0.f) atomicOr(range_flag, 1);
    if (i >= M * C) return;
    const float n = counts[i / C];
    tgt[i] = sums[i] / (n > 0.f ? n : 1.f);
}
// this rank's fp16-range flag as one float of the first exchange record (summed over the ranks by the caller's all-reduce)
__global__ void shard_range_flag_kernel(const int* __restrict__ range_flag, float* __restrict__ out) {
    if (threadIdx.x == 0) *out = (range_flag != nullptr && *range_flag != 0) ? 1.f : 0.f;
}
// attention outputs of R key shards -> the attention output over all keys.  parts: R records of [M*C normalised outputs |
// B*H*Lq_pad log2-domain log-sum-exp rows]; weight of shard r for (scene b, head h, query q) = 2^(lse_r - max_r lse)
__global__ void attn_combine_kernel(const float* __restrict__ parts, int R, int64_t rec, int B, int H, int Q, int Lq_pad, int dh,
                                    float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = H * dh;
    const int64_t M = (int64_t)B * Q;
    if (i >= M * C) return;
    const int64_t m = i / C;
    const int c = (int)(i - m * C);
    const int b = (int)(m / Q), q = (int)(m - (int64_t)b * Q), hh = c / dh;
    const int64_t lrow = M * C + ((int64_t)(b * H + hh)) * Lq_pad + q;
    float mx = -INFINITY;
    for (int r = 0; r < R; ++r) mx = fmaxf(mx, parts[r * rec + lrow]);
    float num = 0.f, den = 0.f;
    for (int r = 0; r < R; ++r) {
        const float wgt = __builtin_amdgcn_exp2f(parts[r * rec + lrow] - mx);
        num += wgt * parts[r * rec + i];
        den += wgt;
    }
    out[i] = num / den;
}

hipError_t launch_sample_finalize(const float* sums, const float* counts, int64_t M, int C, float* tgt, hipStream_t s,
                                  const float* range_sum, int* range_flag) {
    hipLaunchKernelGGL(sample_finalize_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, s, sums, counts, M, C, tgt, range_sum, range_flag);
    return hipGetLastError();
}
hipError_t launch_shard_range_flag(const int* range_flag, float* out, hipStream_t s) {
    hipLaunchKernelGGL(shard_range_flag_kernel, dim3(1), dim3(64), 0, s, range_flag, out);
    return hipGetLastError();
}
hipError_t launch_attn_combine(const float* parts, int R, int64_t rec, int B, int H, int Q, int Lq_pad, int dh, float* out, hipStream_t s) {
    if (R < 1) return hipErrorInvalidValue;
    const int64_t n = (int64_t)B * Q * H * dh;
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, parts, R, rec, B, H, Q, Lq_pad, dh, out);
    return hipGetLastError();
}

hipError_t launch_layernorm(const float* X, const float* gamma, const float* beta, float* Y, int M, int C, float eps,
                            hipStream_t s) {
    if (C > 64 * kLnMaxPerLane) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, s, X, gamma, beta, Y, M, C, eps);
    return hipGetLastError();
}

hipError_t launch_gn_stats(const float* X, int64_t ldx, int col0, int ncols, int ngroups, int B, int rows_per_scene,
                           float eps, float* stats, hipStream_t s) {
    if (ncols % 4 != 0 || col0 % 4 != 0 || ldx % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(ngroups, B), dim3(1024), 0, s, X, ldx, col0, ncols, ngroups, rows_per_scene,
                       eps, stats);
    return hipGetLastError();
}

hipError_t launch_box_decode(const BoxDecodeArgs& a, hipStream_t s) {
    if (a.ncls > kMaxCls || a.ncls < 1) return hipErrorInvalidValue;
    // one row per wave; rows per workgroup: 1 while that still leaves CUs idle (pure latency: spread the rows over the chip), else 4
    static const int rows_env = [] { const char* e = dev_env("PARQ_DECODE_ROWS"); return e ? atoi(e) : 0; }();
    const int rows = (rows_env >= 1 && rows_env <= 4) ? rows_env : (a.M <= 2 * device_num_cus() ? 1 : 4);
    static const bool fast_off = [] { const char* e = dev_env("PARQ_DECODE_FAST"); return e && e[0] == '0'; }();
    const bool al = ((reinterpret_cast<uintptr_t>(a.h2) | reinterpret_cast<uintptr_t>(a.gn_gamma) | reinterpret_cast<uintptr_t>(a.gn_beta) |
                      reinterpret_cast<uintptr_t>(a.w3)) & 15u) == 0 && a.ld2 % 4 == 0;
    if (a.C == 256 && a.n_mean * 3 <= 128 && a.n_mean >= 1 && al && !fast_off)
        hipLaunchKernelGGL(box_decode256_kernel, dim3(ceil_div(a.M, rows)), dim3(rows * 64), 0, s, a);
    else
        hipLaunchKernelGGL(box_decode_kernel, dim3(ceil_div(a.M, rows)), dim3(rows * 64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_zero_f64(double* p, int n, hipStream_t s) {
    hipLaunchKernelGGL(zero_f64_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, s, p, n);
    return hipGetLastError();
}

hipError_t launch_copy_rows(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int rows, int cols,
                            hipStream_t s) {
    const int64_t n = (int64_t)rows * cols;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, src, src_ld, dst, dst_ld, rows,
                       cols);
    return hipGetLastError();
}

hipError_t launch_fill(float* dst, float value, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, dst, value, n);
    return hipGetLastError();
}

PARQ_TL_DEFINE_SETTER(tl_set_elementwise)

}  // namespace parq
