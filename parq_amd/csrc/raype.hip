// Ray-point positional encoding (AddRayPE, model/ray_positional_encoding.py:61-139 with
// utils/encoding_utils.py:15-100), once per forward, fused with the tokenisation of
// model/parq_lightning.py:75-85:
//     tokens[b][(v*h+y)*w+x][:] = features[b][v][:][y][x] + W2 relu(W1 p(b,v,y,x) + b1) + b2
// where p is the 192-vector of 64 log-spaced depth samples along the pixel ray, moved to the
// snippet-local frame, box-normalised and passed through inverse_sigmoid.
//
//   raype_points_kernel   p for every token (geometry in float64, like project_sample)
//   gemm_split_kernel     Y = act(X W^T + b) on the fp16 matrix pipe with fp32-class accuracy
//                         (hi/lo split, 3-term products: see flash_split.hip), 128x128 tiles,
//                         row-major output staged through LDS for coalesced stores; optional
//                         epilogue adds the NCHW feature map and writes channels-last tokens.
// HBM traffic at cfg 3: p 147 MB + hidden 197 MB written and re-read, features 197 MB read,
// tokens 197 MB written (a single fused kernel would keep p and the hidden layer in LDS; this
// two-GEMM form is the first correct version).
#include "common.hpp"
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace parq {

namespace {

// ------------------------------------------------------------------------------------------------
struct RayPeArgs {
    const float* cam;    // (B,V,6)
    const float* T_cp;   // (B,V,12)
    const float* T_wp;   // (B,V,12)
    const float* T_wl;   // (B,1,12)
    float lo[3], hi[3];  // RAY_POINTS_SCALE
    float min_depth, max_depth;
    int B, V, h, w, S;   // S = samples per ray (64)
    float* P;            // (B*V*h*w, 3*S)
};

struct P12 { double R[9]; double t[3]; };

__device__ __forceinline__ P12 ldp(const float* p) {
    P12 o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.R[i] = (double)p[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) o.t[i] = (double)p[9 + i];
    return o;
}
__device__ __forceinline__ P12 pinv(const P12& a) {
    P12 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = a.R[j * 3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) o.t[i] = -(o.R[i * 3] * a.t[0] + o.R[i * 3 + 1] * a.t[1] + o.R[i * 3 + 2] * a.t[2]);
    return o;
}
__device__ __forceinline__ P12 pmul(const P12& a, const P12& b) {
    P12 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.R[i * 3 + j] = a.R[i * 3] * b.R[j] + a.R[i * 3 + 1] * b.R[3 + j] + a.R[i * 3 + 2] * b.R[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) o.t[i] = a.t[i] + (a.R[i * 3] * b.t[0] + a.R[i * 3 + 1] * b.t[1] + a.R[i * 3 + 2] * b.t[2]);
    return o;
}

// one thread per (token, sample): writes 3 floats
__global__ __launch_bounds__(256) void raype_points_kernel(RayPeArgs a) {
    __shared__ double Tl[12];      // local <- camera for this (b, v): inv(T_wl) o T_wp o inv(T_cp)
    const int bv = blockIdx.y;
    const int b = bv / a.V;
    if (threadIdx.x == 0) {
        const P12 T = pmul(pmul(pinv(ldp(a.T_wl + (int64_t)b * 12)), ldp(a.T_wp + (int64_t)bv * 12)),
                           pinv(ldp(a.T_cp + (int64_t)bv * 12)));
        for (int i = 0; i < 9; ++i) Tl[i] = T.R[i];
        for (int i = 0; i < 3; ++i) Tl[9 + i] = T.t[i];
    }
    __syncthreads();
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = a.h * a.w;
    if (idx >= (int64_t)hw * a.S) return;
    const int pix = (int)(idx / a.S);
    const int j = (int)(idx - (int64_t)pix * a.S);
    const int y = pix / a.w, x = pix - y * a.w;
    const float* cm = a.cam + (int64_t)bv * 6;
    // integer pixel grid (no +0.5), unproject to z = 1 (utils/encoding_utils.py:15-20, utils/wrappers.py:543-548)
    const double rx = ((double)x - (double)cm[4]) / (double)cm[2];
    const double ry = ((double)y - (double)cm[5]) / (double)cm[3];
    // depth_j = exp(log dmin + log(dmax/dmin) * j/(S-1))   (utils/encoding_utils.py:82-89)
    const double ramp = a.S > 1 ? (double)j / (double)(a.S - 1) : 0.0;
    const double depth = exp(log((double)a.min_depth) + log((double)a.max_depth / (double)a.min_depth) * ramp);
    const double pc[3] = {rx * depth, ry * depth, depth};
    float* out = a.P + (((int64_t)bv * hw + pix) * a.S + j) * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double pl = Tl[i * 3] * pc[0] + Tl[i * 3 + 1] * pc[1] + Tl[i * 3 + 2] * pc[2] + Tl[9 + i];
        double u = (pl - (double)a.lo[i]) / ((double)a.hi[i] - (double)a.lo[i]);
        u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
        const double x1 = u > 1e-3 ? u : 1e-3;
        const double x2 = (1.0 - u) > 1e-3 ? (1.0 - u) : 1e-3;
        out[i] = (float)log(x1 / x2);                     // inverse_sigmoid, eps = 1e-3 (ray_positional_encoding.py:23-27)
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int kBM = 128, kBN = 128, kBK = 64, kThreads = 256;

struct GemmArgs {
    const float* X; int64_t ldx;       // [M][K] fp32
    const _Float16* Whi; const _Float16* Wlo;   // [N][K] fp16 hi/lo
    const float* bias;                 // [N]
    float* Y; int64_t ldy;             // [M][N] row-major
    int M, N, K, relu;
    const float* scale_dev; float scale_mul;   // optional: Y = (X W^T) * scale_mul / *scale_dev (bias-free callers: attention backward)
    const float* xscale_dev;                   // optional: X is multiplied by *xscale_dev while it is split (gradient operands: a power
                                               // of two that keeps small values' low halves out of the fp16 subnormals)
    int accumulate;                            // Y += instead of Y =
    // optional: Y[m][n] += feat[(m / hw) * N * hw + n * hw + (m % hw)]   (NCHW feature maps, (B*V, C, h, w))
    const float* feat; int hw;
};

__global__ __launch_bounds__(kThreads) void gemm_split_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];      // A_hi | A_lo | W_hi | W_lo, each [128][64]
    _Float16* Ahi = lds;
    _Float16* Alo = lds + kBM * kBK;
    _Float16* Bhi = lds + 2 * kBM * kBK;
    _Float16* Blo = lds + 2 * kBM * kBK + kBN * kBK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int nct = (a.N + kBN - 1) / kBN;
    const int nrt = (a.M + kBM - 1) / kBM;
    int rtile, ctile;
    {   // column tiles of one row tile on one XCD (see kvproj_split.hip)
        const int w = blockIdx.x;
        const int per_group = 8 * nct;
        const int grp = w / per_group;
        const int r = w - grp * per_group;
        rtile = grp * 8 + (r & 7);
        ctile = r >> 3;
    }
    if (rtile >= nrt) return;
    const int m0 = rtile * kBM, n0 = ctile * kBN;
    const int nk = a.K / kBK;

    float4 areg[8];
    uint4 wreg[8];
    const float xs = a.xscale_dev ? *a.xscale_dev : 1.f;
    auto gload = [&](int ks) {
        const int k0 = ks * kBK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * kThreads;
            const int row = id >> 3, c = id & 7;
            if (m0 + row < a.M) {
                const float4* p = reinterpret_cast<const float4*>(a.X + (int64_t)(m0 + row) * a.ldx + k0 + c * 8);
                areg[2 * i] = p[0];
                areg[2 * i + 1] = p[1];
            } else {
                areg[2 * i] = float4{0.f, 0.f, 0.f, 0.f};
                areg[2 * i + 1] = float4{0.f, 0.f, 0.f, 0.f};
            }
            if (n0 + row < a.N) {
                const int64_t off = (int64_t)(n0 + row) * a.K + k0 + c * 8;
                wreg[2 * i] = *reinterpret_cast<const uint4*>(a.Whi + off);
                wreg[2 * i + 1] = *reinterpret_cast<const uint4*>(a.Wlo + off);
            } else {
                wreg[2 * i] = uint4{0u, 0u, 0u, 0u};
                wreg[2 * i + 1] = uint4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * kThreads;
            const int row = id >> 3, c = id & 7;
            const int pos = c ^ ((row >> 1) & 7);
            float x[8] = {areg[2 * i].x, areg[2 * i].y, areg[2 * i].z, areg[2 * i].w,
                          areg[2 * i + 1].x, areg[2 * i + 1].y, areg[2 * i + 1].z, areg[2 * i + 1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= xs;
            half8 hi, lo;
            split8(x, hi, lo);
            *reinterpret_cast<half8*>(Ahi + row * kBK + pos * 8) = hi;
            *reinterpret_cast<half8*>(Alo + row * kBK + pos * 8) = lo;
            *reinterpret_cast<uint4*>(Bhi + row * kBK + pos * 8) = wreg[2 * i];
            *reinterpret_cast<uint4*>(Blo + row * kBK + pos * 8) = wreg[2 * i + 1];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    swrite();
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        if (more) gload(ks + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            half8 xh[2], xl[2], wh[2], wlo2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = wr * 64 + t * 32 + li;
                const int posr = (4 * kh + s) ^ ((row >> 1) & 7);
                xh[t] = *reinterpret_cast<const half8*>(Ahi + row * kBK + posr * 8);
                xl[t] = *reinterpret_cast<const half8*>(Alo + row * kBK + posr * 8);
                const int col = wc * 64 + t * 32 + li;
                const int posc = (4 * kh + s) ^ ((col >> 1) & 7);
                wh[t] = *reinterpret_cast<const half8*>(Bhi + col * kBK + posc * 8);
                wlo2[t] = *reinterpret_cast<const half8*>(Blo + col * kBK + posc * 8);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[rt], wh[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[rt], wlo2[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[rt], wh[ct], acc[rt][ct], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            swrite();
            __syncthreads();
        }
    }

    // ---- epilogue: the wave's 64x64 block through LDS ([64][65] floats, conflict-free both ways), then
    // rows of 64 consecutive output columns are written coalesced
    float* ot = reinterpret_cast<float*>(lds) + wave * (64 * 65);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ot[(rt * 32 + mfma32_row(r, lane)) * 65 + ct * 32 + li] = acc[rt][ct][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (a.feat) {
        // NCHW feature maps are pixel-contiguous: lanes walk the 64 rows (pixels) of one channel at a time
        const int m = m0 + wr * 64 + lane;
        if (m < a.M) {
            const int img = m / a.hw, pix = m - img * a.hw;
            const float* fp = a.feat + (int64_t)img * a.N * a.hw + pix;
            const int cbase = n0 + wc * 64;
            for (int c = 0; c < 64; ++c)
                if (cbase + c < a.N) ot[lane * 65 + c] += fp[(int64_t)(cbase + c) * a.hw];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    const int ncol = n0 + wc * 64 + lane;
    const float bv = (a.bias && ncol < a.N) ? a.bias[ncol] : 0.f;
    const float osc = a.scale_dev ? a.scale_mul / *a.scale_dev : a.scale_mul;
    for (int r = 0; r < 64; ++r) {
        const int m = m0 + wr * 64 + r;
        if (m >= a.M) break;
        if (ncol < a.N) {
            float y = ot[r * 65 + lane] * osc + bv;
            if (a.relu) y = y > 0.f ? y : 0.f;
            if (a.accumulate) y += a.Y[(int64_t)m * a.ldy + ncol];
            a.Y[(int64_t)m * a.ldy + ncol] = y;
        }
    }
}


// ================================================================================================
// Fused fast path (C = 256, 64 samples): two persistent W-stationary kernels, modelled on the K/V projection (kvproj_split.hip).
//
//   raype_hidden_kernel  hidden = relu(p W1^T + b1).  A workgroup of 8 waves owns all 256 hidden units (each wave
//                        keeps the hi/lo fragments of its 32 rows of W1 in registers), walks 64-token tiles and
//                        GENERATES the operand tile p (64 x 192) itself: thread (token = tid & 63) computes three
//                        8-value chunks in float64 geometry (one FMA per value: p_axis = depth_j * g_axis + t_axis
//                        with g = R (rx, ry, 1) per token) + fp32 log, splits them hi/lo and writes them straight
//                        into the swizzled LDS operand image.  The 147 MB point tensor never exists.
//   raype_tokens_kernel  tokens = feat + hidden W2^T + b2 (or the NCHW encoding alone).  Same structure with the
//                        hidden tile streamed from memory; products are taken TRANSPOSED (rows = channels,
//                        columns = tokens) so that lanes walk consecutive pixels: the NCHW feature read (and the
//                        NCHW encoding write) is 128-byte coalesced, a lane's 4 consecutive accumulator registers
//                        are 4 consecutive channels = one 16-byte store into the channels-last token row.
struct RayFusedArgs {
    const float* cam;          // (B*V, 6)
    const double* Tl;          // (B*V, 12): local <- camera, float64
    const double* depth;       // (S)
    double lo[3], inv[3];      // box normalisation u = (p - lo) * inv
    int hw, w, S, M;           // M = B*V*h*w tokens
    const _Float16* Whi; const _Float16* Wlo; const float* bias;   // this kernel's weight [256][K] hi/lo, bias [256]
    const _Float16* W2f; const float* bias2;                       // one-pass kernel: W2 in fragment order (raype_pack_w2_kernel), b2
    float* hidden;             // [M][256]
    const float* feat;         // NCHW features (B*V, 256, h, w) or nullptr
    float* out;                // tokens [M][256] (nchw_out = 0) or encoding (B*V, 256, h, w) (nchw_out = 1)
    int nchw_out;
};

constexpr int kFC = 256;       // channels / hidden units of the fused path
constexpr int kFK1 = 192;      // 3 * 64 samples
constexpr int kFTM = 64;       // tokens per tile
constexpr int kFThreads = 512;
constexpr int kOtLd = kFC + 4;   // row stride (floats) of the epilogue transpose tile

__global__ void raype_pose_kernel(const float* T_cp, const float* T_wp, const float* T_wl, int B, int V, float min_depth,
                                  float max_depth, int S, double* Tl, double* depth) {
    const int bv = blockIdx.x * blockDim.x + threadIdx.x;
    if (bv < B * V) {
        const int b = bv / V;
        const P12 T = pmul(pmul(pinv(ldp(T_wl + (int64_t)b * 12)), ldp(T_wp + (int64_t)bv * 12)), pinv(ldp(T_cp + (int64_t)bv * 12)));
        for (int i = 0; i < 9; ++i) Tl[(int64_t)bv * 12 + i] = T.R[i];
        for (int i = 0; i < 3; ++i) Tl[(int64_t)bv * 12 + 9 + i] = T.t[i];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < S) {
        const int j = threadIdx.x;
        const double ramp = S > 1 ? (double)j / (double)(S - 1) : 0.0;
        depth[j] = exp(log((double)min_depth) + log((double)max_depth / (double)min_depth) * ramp);   // encoding_utils.py:82-89
    }
}

__global__ __launch_bounds__(kFThreads, 1) void raype_hidden_kernel(RayFusedArgs a, int ntiles, int P) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];      // [2 buffers][3 k-steps][A_hi 64x64 | A_lo 64x64]
    __shared__ double dtab[64];
    constexpr int kStep = 2 * kFTM * 64;                                // halfs per k-step (hi + lo)
    constexpr int kBuf = 3 * kStep;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int p = blockIdx.x;
    if (tid < 64) dtab[tid] = tid < a.S ? a.depth[tid] : 1.0;
    // W1 fragments of this wave's 32 hidden units: [k-step][s2][hi, lo]
    const int col = wave * 32 + li;
    half8 wfr[3][4][2];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int64_t off = (int64_t)col * kFK1 + ks * 64 + 32 * kh + 8 * s2;
            wfr[ks][s2][0] = *reinterpret_cast<const half8*>(a.Whi + off);
            wfr[ks][s2][1] = *reinterpret_cast<const half8*>(a.Wlo + off);
        }
    const float bias = a.bias[col];
    __syncthreads();

    // operand generator: thread (token row = tid & 63, wave) produces k = 24 wave .. 24 wave + 23, i.e. the three axes of
    // depths j = 8 wave .. 8 wave + 7 (k = 3 j + axis): per value ONE float64 FMA u = depth_j * G_axis + T_axis with
    // G = (R (rx, ry, 1)) / (hi - lo) and T = (t - lo) / (hi - lo) per token, clamps and 1 - u in float64 (the
    // inverse_sigmoid is ill-conditioned at the clamp edges), ratio and log in fp32 (v_rcp / v_log: ~1e-7 relative).
    const int grow = tid & 63;
    auto generate = [&](int tile, int buf) {
        const int m = tile * kFTM + grow;
        double G[3] = {0.0, 0.0, 0.0}, T3[3] = {0.5, 0.5, 0.5};
        if (m < a.M) {
            const int bv = m / a.hw, pix = m - bv * a.hw;
            const int y = pix / a.w, x = pix - y * a.w;
            const float* cm = a.cam + (int64_t)bv * 6;
            // integer pixel grid (no +0.5), unproject to z = 1 (utils/encoding_utils.py:15-20, utils/wrappers.py:543-548)
            const double rx = ((double)x - (double)cm[4]) / (double)cm[2];
            const double ry = ((double)y - (double)cm[5]) / (double)cm[3];
            const double* T = a.Tl + (int64_t)bv * 12;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                G[i] = (T[i * 3] * rx + T[i * 3 + 1] * ry + T[i * 3 + 2]) * a.inv[i];
                T3[i] = (T[9 + i] - a.lo[i]) * a.inv[i];
            }
        }
        _Float16* B0 = lds + buf * kBuf;
        double dj[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) dj[jj] = dtab[8 * wave + jj];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float x8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int q = 8 * i + e;                    // 0..23 inside this thread's run: depth q / 3, axis q % 3
                const int jj = q / 3, ax = q - 3 * jj;      // compile-time after unrolling
                double u = dj[jj] * G[ax] + T3[ax];
                u = fmin(fmax(u, 0.0), 1.0);
                const double x1 = fmax(u, 1e-3);
                const double x2 = fmax(1.0 - u, 1e-3);
                // inverse_sigmoid, eps 1e-3 (ray_positional_encoding.py:23-27)
                x8[e] = __logf(__fdividef((float)x1, (float)x2));
            }
            half8 hi, lo8;
            split8(x8, hi, lo8);
            const int chunk = 3 * wave + i;                 // 8-value chunk index 0..23 of the 192-vector
            const int ks = chunk >> 3, c = chunk & 7;
            const int pos = c ^ ((grow >> 1) & 7);
            _Float16* Ahi = B0 + ks * kStep;
            *reinterpret_cast<half8*>(Ahi + grow * 64 + pos * 8) = hi;
            *reinterpret_cast<half8*>(Ahi + kFTM * 64 + grow * 64 + pos * 8) = lo8;
        }
    };

    int it = 0;
    if (p < ntiles) generate(p, 0);
    __syncthreads();
    for (int tile = p; tile < ntiles; tile += P, ++it) {
        const int buf = it & 1;
        if (tile + P < ntiles) generate(tile + P, buf ^ 1);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const _Float16* B0 = lds + buf * kBuf;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const _Float16* Ahi = B0 + ks * kStep;
            const _Float16* Alo = Ahi + kFTM * 64;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                half8 xh[2], xl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int row = t * 32 + li;
                    const int posr = (4 * kh + s2) ^ ((row >> 1) & 7);
                    xh[t] = *reinterpret_cast<const half8*>(Ahi + row * 64 + posr * 8);
                    xl[t] = *reinterpret_cast<const half8*>(Alo + row * 64 + posr * 8);
                }
                const half8 wh = wfr[ks][s2][0], wl = wfr[ks][s2][1];
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[t], wh, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[t], wl, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[t], wh, acc[t], 0, 0, 0);
            }
        }
        // rows = tokens, columns = hidden units: a register is one token row, lanes 32 consecutive units
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = tile * kFTM + t * 32 + mfma32_row(r, lane);
                if (m < a.M) {
                    const float y = acc[t][r] + bias;
                    a.hidden[(int64_t)m * kFC + col] = y > 0.f ? y : 0.f;
                }
            }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kFThreads, 1) void raype_tokens_kernel(RayFusedArgs a, int ntiles, int P) {
    // LDS: [2 buffers][A_hi 64x64 | A_lo 64x64] (32 KB), overlaid by the epilogue transpose tile ot[64][kOtLd] (66.5 KB);
    // behind it the NCHW feature tile ftile[256 channels][64 pixels] (64 KB), filled by LDS-DMA during the k-loop
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    constexpr int kStep = 2 * kFTM * 64;
    constexpr int nk = kFC / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int p = blockIdx.x;
    const int col = wave * 32 + li;                  // this lane's row of W2 (output channel of the A operand)
    half8 wfr[nk][4][2];
#pragma unroll
    for (int ks = 0; ks < nk; ++ks)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int64_t off = (int64_t)col * kFC + ks * 64 + 32 * kh + 8 * s2;
            wfr[ks][s2][0] = *reinterpret_cast<const half8*>(a.Whi + off);
            wfr[ks][s2][1] = *reinterpret_cast<const half8*>(a.Wlo + off);
        }
    // hidden tile staging: 64 rows x 64 floats per k-step = 512 pieces of 8 floats, one per thread; two k-steps in flight
    float4 areg[2][2];
    const int srow = tid >> 3, sc = tid & 7;
    auto gload = [&](int tile, int ks, float4 (&dst)[2]) {
        const int m = tile * kFTM + srow;
        if (m < a.M) {
            const float4* q = reinterpret_cast<const float4*>(a.hidden + (int64_t)m * kFC + ks * 64 + sc * 8);
            dst[0] = q[0];
            dst[1] = q[1];
        } else {
            dst[0] = float4{0.f, 0.f, 0.f, 0.f};
            dst[1] = float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto swrite = [&](int buf, const float4 (&src)[2]) {
        _Float16* Ahi = lds + buf * kStep;
        _Float16* Alo = Ahi + kFTM * 64;
        const int pos = sc ^ ((srow >> 1) & 7);
        float x[8] = {src[0].x, src[0].y, src[0].z, src[0].w, src[1].x, src[1].y, src[1].z, src[1].w};
        half8 hi, lo;
        split8(x, hi, lo);
        *reinterpret_cast<half8*>(Ahi + srow * 64 + pos * 8) = hi;
        *reinterpret_cast<half8*>(Alo + srow * 64 + pos * 8) = lo;
    };
    const int my_tiles = p < ntiles ? (ntiles - p + P - 1) / P : 0;
    const int total_steps = my_tiles * nk;
    auto issue = [&](int q, float4 (&dst)[2]) {
        if (q < total_steps) gload(p + (q / nk) * P, q % nk, dst);
    };
    int step = 0;
    issue(0, areg[0]);
    issue(1, areg[1]);
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    float* ot = reinterpret_cast<float*>(lds);                           // [64 tokens][kOtLd]
    float* ftile = ot + kFTM * kOtLd;                                    // [256][64]
    for (int tile = p; tile < ntiles; tile += P) {
        // this tile's feature rows: one LDS-DMA instruction = 64 pixels of one channel (lanes = pixels)
        if (a.feat) {
            const int m = tile * kFTM + lane;
            if (m < a.M) {
                const int bv = m / a.hw, pix = m - bv * a.hw;
                const float* fp = a.feat + (int64_t)bv * kFC * a.hw + pix;
#pragma unroll 8
                for (int i = 0; i < 32; ++i) {
                    const int c = wave * 32 + i;
                    __builtin_amdgcn_global_load_lds(fp + (int64_t)c * a.hw, (lds_byte*)(ftile + c * 64), 4, 0, 0);
                }
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < nk; ++ks) {
            const int buf = step & 1;
            swrite(buf, areg[ks & 1]);
            issue(step + 2, areg[ks & 1]);
            __syncthreads();
            const _Float16* Ahi = lds + buf * kStep;
            const _Float16* Alo = Ahi + kFTM * 64;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                half8 xh[2], xl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int row = t * 32 + li;
                    const int posr = (4 * kh + s2) ^ ((row >> 1) & 7);
                    xh[t] = *reinterpret_cast<const half8*>(Ahi + row * 64 + posr * 8);
                    xl[t] = *reinterpret_cast<const half8*>(Alo + row * 64 + posr * 8);
                }
                const half8 wh = wfr[ks][s2][0], wl = wfr[ks][s2][1];
                // transposed product: rows = channels, columns = tokens
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[t], acc[t], 0, 0, 0);
            }
            ++step;
        }
        // epilogue: register r of block t is channel 32 wave + mfma32_row(r, lane) of token tile*64 + 32 t + li.
        // The NCHW encoding is written lane-contiguous as it is; channels-last token rows go through an LDS
        // transpose (ot overlays the operand buffers: every wave is past its last fragment read) so that every store
        // instruction writes one full 1 KB row.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // feature tile landed (hipcc does not track LDS-DMA)
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int m = tile * kFTM + t * 32 + li;
            const bool ok = m < a.M;
            const int bv = ok ? m / a.hw : 0, pix = ok ? m - bv * a.hw : 0;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c0 = wave * 32 + 8 * g4 + 4 * kh;             // 4 consecutive channels in registers 4 g4 .. 4 g4 + 3
                const float4 b4 = *reinterpret_cast<const float4*>(a.bias + c0);
                float y[4] = {acc[t][4 * g4] + b4.x, acc[t][4 * g4 + 1] + b4.y, acc[t][4 * g4 + 2] + b4.z, acc[t][4 * g4 + 3] + b4.w};
                if (a.feat && ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] += ftile[(c0 + e) * 64 + t * 32 + li];
                }
                if (a.nchw_out) {
                    if (ok) {
                        float* op = a.out + (int64_t)bv * kFC * a.hw + pix;
#pragma unroll
                        for (int e = 0; e < 4; ++e) op[(int64_t)(c0 + e) * a.hw] = y[e];
                    }
                } else {
                    *reinterpret_cast<float4*>(ot + (t * 32 + li) * kOtLd + c0) = float4{y[0], y[1], y[2], y[3]};
                }
            }
        }
        __syncthreads();
        if (!a.nchw_out) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int row = wave * 8 + rr;
                const int m = tile * kFTM + row;
                if (m < a.M)
                    *reinterpret_cast<float4*>(a.out + (int64_t)m * kFC + 4 * lane) =
                        *reinterpret_cast<const float4*>(ot + row * kOtLd + 4 * lane);
            }
            __syncthreads();                                             // ot overlays the next tile's operand buffers
        }
    }
}


// ------------------------------------------------------------------------------------------------
// ONE-PASS form (round 4, re-scheduled in round 5): tokens = feat + relu(p W1^T + b1) W2^T + b2 with the 64-token hidden tile kept in LDS — the 197 MB
// hidden tensor of the two-kernel form above (written by raype_hidden_kernel, re-read by raype_tokens_kernel: 786 MB moved for
// 393 MB algorithmic at BASELINE cfg 3) never exists.  model/ray_positional_encoding.py:128-136 is one MLP.
//
//   registers  W1 fragments of the wave's 32 hidden units for the whole launch (96 VGPRs, as in raype_hidden_kernel); W2 does
//              NOT fit beside them (128 more), so its fragments are STREAMED per k-step from a fragment-ordered copy
//              (raype_pack_w2_kernel: one wave-wide 16-byte load = one contiguous KB, L2-resident 256 KB shared by every CU),
//              four (k-step, s2) steps ahead of the MFMAs (ring of 4 x 8 VGPRs, primed behind the first GEMM); the feature
//              tile is loaded at the top of a tile straight INTO the second GEMM's accumulators (no separate registers, no
//              add in the epilogue).  NO SPILLS (round 5): vector memory returns in order, so one scratch reload inside the
//              tile waits for every feature load and row store in front of it — GEMM 1 runs one 32-token block at a time
//              (one accumulator set), image indices are stepped instead of divided, the transpose rows share one LDS address
//   LDS        hid [4 k-steps][hi | lo][64 x 64] 64 KB at offset 0, overlaid by the epilogue's transpose tile ot[64][260]
//              65 KB | pts [3 k-steps][hi | lo][64 x 64] 48 KB (generated, single buffer) | b1, b2 2 KB (a bias load from
//              memory inside the tile would queue behind the feature loads) | depth table
//   per tile   features requested | GEMM 1 TRANSPOSED, block by block (rows = hidden units, columns = tokens: a lane's 4
//              consecutive registers are 4 consecutive units of one token = one 8-byte piece of the hid image), relu, split,
//              hid image | barrier | GEMM 2 in half steps (rows = channels, columns = tokens) | barrier | + b2, transpose
//              through ot | barrier | row stores (1 KB per instruction) in turns with the pieces of generate(next tile) |
//              barrier.  In both GEMMs the operand fragments of step n + 1 are requested in front of the MFMAs of step n and
//              the order is pinned with sched_group_barrier / sched_barrier: left alone, hipcc sinks every LDS read to its
//              first use and folds the W2 ring into one slot, and each step waits for an LDS latency and an L2 round trip.
//   measured   (profiles/r05_raype.txt) AddRayPE.tokens at cfg 3: 0.244 -> 0.18 ms.  What is left adds up by construction:
//              matrix + vector work of one SIMD do not overlap on this part (GEMMs + relu/split 107 us, generator 28 us), and the
//              feature loads / row stores (25 + 25-35 us) run at the ~12 B/clk a CU gets from memory, with the waves blocked
//              at issue.  A phased form (generator inside GEMM 2, row stores inside the next GEMM 1, or stores straight from
//              the accumulators with two barriers per tile) measured the same and was not kept (commit fc35854).
// KEEP: training keeps the fp32 hidden layer for parq_ray_pe_backward (16-byte pieces; the inference path writes nothing).
__global__ void raype_pack_w2_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, _Float16* __restrict__ out) {
    // out[wave 8][ks 4][s2 4][hl 2][lane 64][8] <- W{hl}[col = 32 wave + (lane & 31)][k = 64 ks + 32 (lane >> 5) + 8 s2 + e]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;            // one 16-byte piece
    if (i >= 8 * 4 * 4 * 2 * 64) return;
    const int lane = i & 63, hl = (i >> 6) & 1, s2 = (i >> 7) & 3, ks = (i >> 9) & 3, wave = i >> 11;
    const int col = 32 * wave + (lane & 31), k = 64 * ks + 32 * (lane >> 5) + 8 * s2;
    const _Float16* src = (hl ? lo : hi) + (int64_t)col * kFC + k;
    *reinterpret_cast<half8*>(out + (int64_t)i * 8) = *reinterpret_cast<const half8*>(src);
}

// PROBE (development library only, PARQ_RAYPE_PROBE).  Ingredients taken out, results wrong: 1 W2 fragments loaded once instead of per
// tile, 2 operand image generated once, 4 no feature loads, 8 no token stores.  Results right: 16 s_memrealtime stamps per phase (printed
// by the launcher), 32 feature tile requested one tile ahead instead of at the tile top, 64 all row stores before the generator,
// 128 pose through per-lane loads, 256 W2 ring of three (tools/r05_raype_variants.sh, profiles/r05_raype.txt)
template <bool KEEP, int PROBE = 0, bool NCHW = false>
__global__ __launch_bounds__(kFThreads, 1) void raype_onepass_kernel(RayFusedArgs a, int ntiles, int P) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    __shared__ double dtab[64];
    constexpr int kStep = 2 * kFTM * 64;                                // halfs per k-step (hi + lo)
    _Float16* hid = lds;                                                // 4 k-steps, at offset 0: every fragment read of the second GEMM
    float* ot = reinterpret_cast<float*>(hid);                          // is (one of 4 lane addresses) + a 16-bit immediate.  ot = [64][kOtLd]
    _Float16* pts = lds + kFTM * kOtLd * 2;                             // 3 k-steps, behind ot (65 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int p = blockIdx.x;
    if (tid < 64) dtab[tid] = tid < a.S ? a.depth[tid] : 1.0;
    // b1 | b2 in LDS: vector-memory loads return in order, so a bias load inside the tile would sit behind the tile's feature loads
    __shared__ __attribute__((aligned(16))) float bsh[2 * kFC];
    bsh[tid] = tid < kFC ? a.bias[tid] : a.bias2[tid - kFC];
    const int col = wave * 32 + li;                                     // this lane's row of W1 (A operand of the transposed product)
    half8 wfr[3][4][2];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int64_t off = (int64_t)col * kFK1 + ks * 64 + 32 * kh + 8 * s2;
            wfr[ks][s2][0] = *reinterpret_cast<const half8*>(a.Whi + off);
            wfr[ks][s2][1] = *reinterpret_cast<const half8*>(a.Wlo + off);
        }
    // W2 fragments of (wave, step = 4 ks + s2): two pieces [hi, lo] of 1 KB each, lane-contiguous (raype_pack_w2_kernel)
    // (buffer loads: one lane offset, the piece as a scalar offset — 32 separate 64-bit addresses would cost 64 VGPRs)
    __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.W2f + (int64_t)wave * 4 * 4 * 2 * 64 * 8), 0, 4 * 4 * 2 * 64 * 16, 0x00020000);
    auto load_w2 = [&](int step, half8 (&dst)[2]) {
        if constexpr ((PROBE & 1) != 0) { if (step >= 4) return; }
        dst[0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane * 16, (step * 2) * 1024, 0));
        dst[1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane * 16, (step * 2 + 1) * 1024, 0));
    };
    __syncthreads();

    const int grow = tid & 63;
    // the operand image of a tile (as raype_hidden_kernel), in two steps so that the row stores of the tile before can be issued
    // between its pieces: gen_setup = the ray of this thread's token (G, T3), gen_chunk(i) = eight values of axis-chunk i.
    // The pose and intrinsics come through SCALAR loads when the whole tile lies in one image (always when h w is a multiple of 64): vector
    // loads would queue behind the row stores just issued and the generator would wait for their acknowledgement.
    double G[3], T3[3];
    auto gen_setup = [&](int tile) {
        const int m = tile * kFTM + grow;
        G[0] = G[1] = G[2] = 0.0; T3[0] = T3[1] = T3[2] = 0.5;
        const int bv0 = (tile * kFTM) / a.hw;                             // scalar
        const int mlast = min(tile * kFTM + kFTM - 1, a.M - 1);
        const bool one_image = (PROBE & 128) == 0 && mlast / a.hw == bv0;   // scalar
        float cm[4]; double T[12];                                        // fx, fy, cx, cy | R (row-major), t
        if (one_image) {
            typedef const float __attribute__((address_space(4)))* kcf;   // constant address space: s_load
            typedef const double __attribute__((address_space(4)))* kcd;
            kcf c = (kcf)(a.cam + (int64_t)bv0 * 6 + 2);
            kcd t = (kcd)(a.Tl + (int64_t)bv0 * 12);
#pragma unroll
            for (int i = 0; i < 4; ++i) cm[i] = c[i];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = t[i];
        } else {
            int bv = bv0;
            for (int q = min(m, a.M - 1) - bv0 * a.hw; q >= a.hw; q -= a.hw) ++bv;
#pragma unroll
            for (int i = 0; i < 4; ++i) cm[i] = a.cam[(int64_t)bv * 6 + 2 + i];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = a.Tl[(int64_t)bv * 12 + i];
        }
        if (m < a.M) {
            int pix = m - bv0 * a.hw;
            while (pix >= a.hw) pix -= a.hw;
            const int y = pix / a.w, x = pix - y * a.w;
            const double rx = ((double)x - (double)cm[2]) / (double)cm[0];
            const double ry = ((double)y - (double)cm[3]) / (double)cm[1];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                G[i] = (T[i * 3] * rx + T[i * 3 + 1] * ry + T[i * 3 + 2]) * a.inv[i];
                T3[i] = (T[9 + i] - a.lo[i]) * a.inv[i];
            }
        }
    };
    auto gen_chunk = [&](int i) {
        double dj[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) dj[jj] = dtab[8 * wave + jj];
        {
            float x8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int q = 8 * i + e;
                const int jj = q / 3, ax = q - 3 * jj;
                // u and 1 - u in float64 (the inverse-sigmoid is ill-conditioned at the clamp edges), the clamps on their fp32
                // roundings: rounding is monotone, so max / min commute with it and the values equal the all-float64 clamps
                const double u = dj[jj] * G[ax] + T3[ax];
                const float uf = (float)u, wf = (float)(1.0 - u);
                const float x1 = fmaxf(fminf(uf, 1.f), 1e-3f);             // max(clamp(u, 0, 1), 1e-3)
                const float x2 = fmaxf(fminf(wf, 1.f), 1e-3f);             // max(1 - clamp(u, 0, 1), 1e-3)
                x8[e] = __logf(__fdividef(x1, x2));
            }
            half8 hi, lo8;
            split8(x8, hi, lo8);
            const int chunk = 3 * wave + i;
            const int ks = chunk >> 3, c = chunk & 7;
            const int pos = c ^ ((grow >> 1) & 7);
            _Float16* Ahi = pts + ks * kStep;
            *reinterpret_cast<half8*>(Ahi + grow * 64 + pos * 8) = hi;
            *reinterpret_cast<half8*>(Ahi + kFTM * 64 + grow * 64 + pos * 8) = lo8;
        }
    };

    // the feature tile goes straight INTO the second GEMM's accumulators (register r of block t = channel 32 wave +
    // mfma32_row(r, lane) of token tile*64 + 32 t + li: lanes walk consecutive pixels, 128-byte segments), requested one tile
    // ahead — right after the epilogue of the tile before has read the accumulators — so it travels under the row stores, the
    // generator and the first GEMM, and nothing of a tile's own W2 stream ever queues behind it (vector memory returns in order).
    // (buffer loads: ONE lane-dependent byte offset per block t, the 16 channel strides as scalar offsets — global loads with
    // 32 distinct 64-bit addresses per lane cost 64 VGPRs of addresses and spilled the tile loop.  The descriptor's base is
    // the first image of the tile, so the 32-bit offsets stay small whatever the batch size.)
    f32x16 acc2[2];
    auto load_features = [&](int tile) {
        const int bv0 = (tile * kFTM) / a.hw;                             // scalar: first image this tile touches
        const float* fbase = (a.feat ? a.feat : a.bias) + (int64_t)bv0 * kFC * a.hw;
        // no feature maps (AddRayPE.forward: the encoding alone): zero records -> every load returns 0, no branch per load
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)fbase, 0, a.feat ? 0x7fffffff : 0, 0x00020000);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int m = tile * kFTM + t * 32 + li;
            const bool ok = m < a.M;                                     // rows past the end read (and discard) pixel 0 of image bv0
            int dbv = 0, pix = ok ? m - bv0 * a.hw : 0;                   // image index by stepping (a tile rarely leaves its first
            while (pix >= a.hw) { pix -= a.hw; ++dbv; }                   // image): a per-lane division keeps its reciprocal live
            const int voff = ((dbv * kFC + wave * 32 + 4 * kh) * a.hw + pix) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int soff = (8 * (r >> 2) + (r & 3)) * a.hw * 4;   // mfma32_row(r, lane) = 8 (r >> 2) + 4 kh + (r & 3)
                if constexpr ((PROBE & 4) != 0) acc2[t][r] = 0.f;
                else acc2[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
            }
        }
    };

    if (p < ntiles) {
        gen_setup(p);
#pragma unroll
        for (int i = 0; i < 3; ++i) gen_chunk(i);
        if constexpr ((PROBE & 32) != 0) load_features(p);
    }
    __syncthreads();
    // PROBE 16 (development): s_memrealtime stamps (100 MHz) of waves 0 and 5 at ten points of every tile -> a.hidden as u64[P][2][16][10]
    [[maybe_unused]] unsigned long long* stamps = nullptr;
    if constexpr ((PROBE & 16) != 0) {
        if ((wave == 0 || wave == 5) && lane == 0) stamps = reinterpret_cast<unsigned long long*>(a.hidden) + ((int64_t)p * 2 + (wave == 5)) * 160;
    }
    [[maybe_unused]] int tcount = 0;
#define PARQ_RP_STAMP(k) do { if constexpr ((PROBE & 16) != 0) { if (stamps && tcount < 16) stamps[tcount * 10 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
    for (int tile = p; tile < ntiles; tile += P) {
        PARQ_RP_STAMP(0);
        constexpr int kRing = (PROBE & 256) ? 3 : 4;                      // (k-step, s2) steps of W2 fragments in flight
        half8 w2r[kRing][2];
        if constexpr ((PROBE & 32) == 0) load_features(tile);
        // ---- GEMM 1, transposed, one 32-token block at a time: acc1 rows = this wave's 32 hidden units, columns = tokens
        // 32 t .. 32 t + 31 (one accumulator set live beside W1 and the prefetched feature tile: both blocks at once spill, and a
        // scratch reload waits for every vector-memory operation in front of it — the feature tile and the row stores)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
            const int row = t * 32 + li;
            // operand fragments one step ahead of the MFMAs that use them (sched_barrier: hipcc otherwise sinks every read to
            // its first use and each step pays the LDS latency)
            auto rd1 = [&](int st, half8 (&x)[2]) {
                const int ks = st >> 2, s2 = st & 3;
                const int posr = (4 * kh + s2) ^ ((row >> 1) & 7);
                const _Float16* Ahi = pts + ks * kStep + row * 64 + posr * 8;
                x[0] = *reinterpret_cast<const half8*>(Ahi);
                x[1] = *reinterpret_cast<const half8*>(Ahi + kFTM * 64);
            };
            half8 xq[2][2];
            rd1(0, xq[0]);
#pragma unroll
            for (int st = 0; st < 12; ++st) {
                if (st + 1 < 12) rd1(st + 1, xq[(st + 1) & 1]);
                if (t == 1 && st >= 12 - kRing) load_w2(st - (12 - kRing), w2r[st - (12 - kRing)]);   // the first W2 steps, late: a ring
                const half8 wh = wfr[st >> 2][st & 3][0], wl = wfr[st >> 2][st & 3][1];             // live during this GEMM spills
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xq[st & 1][0], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xq[st & 1][1], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xq[st & 1][0], acc1, 0, 0, 0);
                if (st + 1 < 12) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                   // the reads first,
                if (t == 1 && st >= 12 - kRing) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);     // then the W2 requests,
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                    // then the MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- relu(acc1 + b1) -> hid image (B operand of GEMM 2: rows = tokens, 8-unit chunks, swizzled as every operand image).
            // Registers 4 g .. 4 g + 3 of a lane are units 32 wave + 8 g + 4 kh .. + 3 of token 32 t + li: half a chunk, 8 bytes.
            const int m = tile * kFTM + row;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int u0 = wave * 32 + 8 * g4 + 4 * kh;
                const float4 b4 = *reinterpret_cast<const float4*>(bsh + u0);
                float y[4] = {acc1[4 * g4] + b4.x, acc1[4 * g4 + 1] + b4.y, acc1[4 * g4 + 2] + b4.z, acc1[4 * g4 + 3] + b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = y[e] > 0.f ? y[e] : 0.f;
                if constexpr (KEEP) {
                    if (m < a.M) *reinterpret_cast<float4*>(a.hidden + (int64_t)m * kFC + u0) = float4{y[0], y[1], y[2], y[3]};
                }
                half2v h01, l01, h23, l23;
                split_pair(y[0], y[1], h01, l01);
                split_pair(y[2], y[3], h23, l23);
                const int ks2 = wave >> 1, c = (wave & 1) * 4 + g4;
                const int pos = c ^ ((row >> 1) & 7);
                _Float16* Hh = hid + ks2 * kStep + row * 64 + pos * 8 + 4 * kh;
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(Hh) = u32x2{__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23)};
                *reinterpret_cast<u32x2*>(Hh + kFTM * 64) = u32x2{__builtin_bit_cast(unsigned int, l01), __builtin_bit_cast(unsigned int, l23)};
            }
        }
        PARQ_RP_STAMP(1);
        __syncthreads();                                                  // hid complete
        PARQ_RP_STAMP(2);
        // ---- GEMM 2 on top of the feature tile: rows = channels, columns = tokens (as raype_tokens_kernel); W2 kRing steps ahead,
        // in half steps (one k-step of one 32-token block): the fragments of half step n + 1 are requested in front of the MFMAs of
        // half step n, a ring slot is refilled as soon as its step has been issued (sched_barrier pins that order: left alone, hipcc
        // folds the ring into one slot and sinks the reads, and every step waits for an L2 round trip and an LDS latency)
        {
            auto rd2 = [&](int hs, half8 (&x)[2]) {
                const int step = hs >> 1, t = hs & 1, ks = step >> 2, s2 = step & 3;
                const int row = t * 32 + li;
                const int posr = (4 * kh + s2) ^ ((row >> 1) & 7);
                const _Float16* Ahi = hid + ks * kStep + row * 64 + posr * 8;
                x[0] = *reinterpret_cast<const half8*>(Ahi);
                x[1] = *reinterpret_cast<const half8*>(Ahi + kFTM * 64);
            };
            half8 xq[2][2];
            rd2(0, xq[0]);
#pragma unroll
            for (int hs = 0; hs < 32; ++hs) {
                const int step = hs >> 1, t = hs & 1;
                if (hs + 1 < 32) rd2(hs + 1, xq[(hs + 1) & 1]);
                if (t == 0 && step >= 1 && step - 1 + kRing < 16) load_w2(step - 1 + kRing, w2r[(step - 1) % kRing]);
                const half8 wh = w2r[step % kRing][0], wl = w2r[step % kRing][1];
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xq[hs & 1][0], acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xq[hs & 1][1], acc2[t], 0, 0, 0);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xq[hs & 1][0], acc2[t], 0, 0, 0);
                if (hs + 1 < 32) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (t == 0 && step >= 1 && step - 1 + kRing < 16) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PARQ_RP_STAMP(3);
        __syncthreads();                                                  // hid is free: ot overlays it
        PARQ_RP_STAMP(4);
        // ---- epilogue (as raype_tokens_kernel): + features + b2, NCHW encoding lane-contiguous, or channels-last rows through ot
        {
            const int bv0 = (tile * kFTM) / a.hw;
            __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (NCHW ? (int64_t)bv0 * kFC * a.hw : 0)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int m = tile * kFTM + t * 32 + li;
                const bool ok = m < a.M;
                const int bv = ok ? m / a.hw : bv0, pix = ok ? m - bv * a.hw : 0;
                const int voff = (((bv - bv0) * kFC + wave * 32 + 4 * kh) * a.hw + pix) * 4;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c0 = wave * 32 + 8 * g4 + 4 * kh;
                    const float4 b4 = *reinterpret_cast<const float4*>(bsh + kFC + c0);
                    const float y[4] = {acc2[t][4 * g4] + b4.x, acc2[t][4 * g4 + 1] + b4.y, acc2[t][4 * g4 + 2] + b4.z, acc2[t][4 * g4 + 3] + b4.w};
                    if constexpr (NCHW) {
                        if (ok) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, y[e]), ors, voff, (8 * g4 + e) * a.hw * 4, 0);
                        }
                    } else {
                        *reinterpret_cast<float4*>(ot + (t * 32 + li) * kOtLd + c0) = float4{y[0], y[1], y[2], y[3]};
                    }
                }
            }
        }
        const bool more = tile + P < ntiles;
        if (more) { if constexpr ((PROBE & 32) != 0) load_features(tile + P); }
        const float* otw = ot + wave * 8 * kOtLd + 4 * lane;              // one LDS address, the rows as immediate offsets
        auto store_rows = [&](int r0, int r1) {
#pragma unroll
            for (int rr = r0; rr < r1; ++rr) {
                const int m = tile * kFTM + wave * 8 + rr;
                if (m < a.M && (PROBE & 8) == 0)
                    *reinterpret_cast<float4*>(a.out + (int64_t)m * kFC + 4 * lane) = *reinterpret_cast<const float4*>(otw + rr * kOtLd);
            }
        };
        PARQ_RP_STAMP(5);
        if constexpr (!NCHW) __syncthreads();
        PARQ_RP_STAMP(6);
        // ---- row stores and the next tile's operand image in turns (every wave is done with pts since the first barrier): a
        // wave's stores stall at issue while the write path is full, the generator's vector work fills that time
        if constexpr (!NCHW) store_rows(0, (PROBE & 64) ? 8 : 3);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((PROBE & 2) == 0) { if (more) { gen_setup(tile + P); gen_chunk(0); } }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!NCHW) if ( (PROBE & 64) == 0) store_rows(3, 6);
        __builtin_amdgcn_sched_barrier(0);
        PARQ_RP_STAMP(7);
        if constexpr ((PROBE & 2) == 0) { if (more) gen_chunk(1); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!NCHW) if ( (PROBE & 64) == 0) store_rows(6, 8);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((PROBE & 2) == 0) { if (more) gen_chunk(2); }
        PARQ_RP_STAMP(8);
        __syncthreads();                                                 // pts complete; ot (= hid) free for the next tile
        PARQ_RP_STAMP(9);
        if constexpr ((PROBE & 16) != 0) ++tcount;
    }
#undef PARQ_RP_STAMP
}


}  // namespace

hipError_t launch_raype_points(const float* cam, const float* T_cp, const float* T_wp, const float* T_wl,
                               const float* scale6, float min_depth, float max_depth, int B, int V, int h, int w, int S,
                               float* P, hipStream_t s) {
    RayPeArgs a;
    a.cam = cam; a.T_cp = T_cp; a.T_wp = T_wp; a.T_wl = T_wl;
    for (int i = 0; i < 3; ++i) { a.lo[i] = scale6[2 * i]; a.hi[i] = scale6[2 * i + 1]; }
    a.min_depth = min_depth; a.max_depth = max_depth;
    a.B = B; a.V = V; a.h = h; a.w = w; a.S = S; a.P = P;
    dim3 grid((unsigned)ceil_div64((int64_t)h * w * S, 256), B * V);
    hipLaunchKernelGGL(raype_points_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// Fused path: pose/depth tables, hidden layer, tokens (or NCHW encoding).  C = 256, S = 64 only.
hipError_t launch_raype_fused(const float* cam, const float* T_cp, const float* T_wp, const float* T_wl, const float* scale6,
                              float min_depth, float max_depth, int B, int V, int h, int w, const void* W1hi, const void* W1lo,
                              const float* b1, const void* W2hi, const void* W2lo, const float* b2, const float* feat,
                              float* hidden, double* Tl, double* depth, float* out, int nchw_out, hipStream_t s, void* W2f,
                              int two_kernels) {
    const int S = 64;
    const int64_t M64 = (int64_t)B * V * h * w;
    if (M64 > 0x7fffffffLL) return hipErrorInvalidValue;
    static DynLdsOnce once_a, once_b;
    const size_t lds_a = (size_t)2 * 3 * 2 * kFTM * 64 * sizeof(_Float16);      // 96 KB
    const size_t lds_b = (size_t)kFTM * kOtLd * sizeof(float) + (size_t)kFC * 64 * sizeof(float);   // 66.5 KB (overlays the 32 KB operand buffers) + 64 KB feature tile
    if (hipError_t e = once_a.ensure(reinterpret_cast<const void*>(&raype_hidden_kernel), lds_a); e != hipSuccess) return e;
    if (hipError_t e = once_b.ensure(reinterpret_cast<const void*>(&raype_tokens_kernel), lds_b); e != hipSuccess) return e;
    hipLaunchKernelGGL(raype_pose_kernel, dim3(ceil_div(B * V, 64)), dim3(64), 0, s, T_cp, T_wp, T_wl, B, V, min_depth, max_depth,
                       S, Tl, depth);
    RayFusedArgs a;
    a.cam = cam; a.Tl = Tl; a.depth = depth;
    for (int i = 0; i < 3; ++i) {
        a.lo[i] = (double)scale6[2 * i];
        a.inv[i] = 1.0 / ((double)scale6[2 * i + 1] - (double)scale6[2 * i]);
    }
    a.hw = h * w; a.w = w; a.S = S; a.M = (int)M64;
    a.hidden = hidden; a.feat = nullptr; a.out = nullptr; a.nchw_out = 0;
    const int ntiles = ceil_div((int)M64, kFTM);
    int P = device_num_cus();
    if (P > ntiles) P = ntiles;
    a.Whi = reinterpret_cast<const _Float16*>(W1hi); a.Wlo = reinterpret_cast<const _Float16*>(W1lo); a.bias = b1;
    a.W2f = nullptr; a.bias2 = nullptr;
    const bool w2f_cached = (two_kernels & 2) != 0;                     // bit 1: W2f already holds this W2 (caller's weight cache)
    two_kernels &= 1;
    if (!two_kernels && W2f) {
        // one pass: the hidden tile never leaves the CU (hidden != nullptr: training keeps an fp32 copy for the backward)
        static DynLdsOnce once_k, once_n;
        const size_t lds_f = (size_t)3 * 2 * kFTM * 64 * sizeof(_Float16) + (size_t)kFTM * kOtLd * sizeof(float);      // 48 KB + 65 KB
        if (!w2f_cached)
            hipLaunchKernelGGL(raype_pack_w2_kernel, dim3(8 * 4 * 4 * 2 * 64 / 256), dim3(256), 0, s, reinterpret_cast<const _Float16*>(W2hi),
                               reinterpret_cast<const _Float16*>(W2lo), reinterpret_cast<_Float16*>(W2f));
        a.W2f = reinterpret_cast<const _Float16*>(W2f); a.bias2 = b2;
        a.feat = feat; a.out = out; a.nchw_out = nchw_out;
        if (hidden && nchw_out) {
            static DynLdsOnce once_kn;
            if (hipError_t e = once_kn.ensure(reinterpret_cast<const void*>(&raype_onepass_kernel<true, 0, true>), lds_f); e != hipSuccess) return e;
            hipLaunchKernelGGL((raype_onepass_kernel<true, 0, true>), dim3(P), dim3(kFThreads), lds_f, s, a, ntiles, P);
        } else if (hidden) {
            if (hipError_t e = once_k.ensure(reinterpret_cast<const void*>(&raype_onepass_kernel<true>), lds_f); e != hipSuccess) return e;
            hipLaunchKernelGGL(raype_onepass_kernel<true>, dim3(P), dim3(kFThreads), lds_f, s, a, ntiles, P);
        } else if (nchw_out) {
            static DynLdsOnce once_nn;
            if (hipError_t e = once_nn.ensure(reinterpret_cast<const void*>(&raype_onepass_kernel<false, 0, true>), lds_f); e != hipSuccess) return e;
            hipLaunchKernelGGL((raype_onepass_kernel<false, 0, true>), dim3(P), dim3(kFThreads), lds_f, s, a, ntiles, P);
        } else {
#ifdef PARQ_DEV_PROBES
            static const int probe = [] { const char* e = dev_env("PARQ_RAYPE_PROBE"); return e ? atoi(e) : 0; }();
#define PARQ_RP(PB) case PB: { static DynLdsOnce o; if (hipError_t e = o.ensure(reinterpret_cast<const void*>(&raype_onepass_kernel<false, PB>), lds_f); e != hipSuccess) return e; \
                               hipLaunchKernelGGL((raype_onepass_kernel<false, PB>), dim3(P), dim3(kFThreads), lds_f, s, a, ntiles, P); return hipGetLastError(); }
            switch (probe) { PARQ_RP(1) PARQ_RP(2) PARQ_RP(3) PARQ_RP(4) PARQ_RP(8) PARQ_RP(15) PARQ_RP(32) PARQ_RP(64) PARQ_RP(128) PARQ_RP(96) PARQ_RP(160) PARQ_RP(192) PARQ_RP(224) PARQ_RP(256) default: break; }
#undef PARQ_RP
            if (probe == 16) {                                            // phase stamps: printed to stderr, synchronises
                static unsigned long long* sbuf = nullptr;
                const size_t sbytes = (size_t)P * 2 * 160 * 8;
                if (!sbuf && hipMalloc(&sbuf, sbytes) != hipSuccess) return hipErrorOutOfMemory;
                (void)hipMemsetAsync(sbuf, 0, sbytes, s);
                static DynLdsOnce o;
                if (hipError_t e = o.ensure(reinterpret_cast<const void*>(&raype_onepass_kernel<false, 16>), lds_f); e != hipSuccess) return e;
                a.hidden = reinterpret_cast<float*>(sbuf);
                hipLaunchKernelGGL((raype_onepass_kernel<false, 16>), dim3(P), dim3(kFThreads), lds_f, s, a, ntiles, P);
                (void)hipStreamSynchronize(s);
                std::vector<unsigned long long> hb((size_t)P * 2 * 160);
                (void)hipMemcpy(hb.data(), sbuf, sbytes, hipMemcpyDeviceToHost);
                static int calls = 0;
                if (++calls == 5) {
                    const char* nm[10] = {"top", "gemm1+hid", "barrier1", "gemm2", "barrier2", "epilogue->ot", "barrier3", "row stores", "generate", "barrier4"};
                    for (int wv = 0; wv < 2; ++wv) {
                        double acc[10] = {0}; int n = 0; double tot = 0;
                        for (int pp = 0; pp < P; ++pp)
                            for (int t = 1; t < 10; ++t) {                // tiles 1..9 (steady state)
                                const unsigned long long* r = hb.data() + ((size_t)pp * 2 + wv) * 160 + t * 10;
                                if (!r[9]) continue;
                                for (int k = 1; k < 10; ++k) acc[k] += (double)(r[k] - r[k - 1]) * 10.0;
                                tot += (double)(r[9] - r[0]) * 10.0; ++n;
                            }
                        fprintf(stderr, "[raype stamps] wave %d: %d tiles, %.0f ns per tile:", wv ? 5 : 0, n, tot / n);
                        for (int k = 1; k < 10; ++k) fprintf(stderr, " %s %.0f", nm[k], acc[k] / n);
                        fprintf(stderr, "\n");
                    }
                    const unsigned long long* r0 = hb.data();
                    unsigned long long lo = ~0ull, hi = 0;
                    for (int pp = 0; pp < P; ++pp) { const unsigned long long* r = hb.data() + (size_t)pp * 2 * 160; if (r[0] && r[0] < lo) lo = r[0];
                        for (int t = 0; t < 16; ++t) if (r[t * 10 + 9] > hi) hi = r[t * 10 + 9]; }
                    (void)r0;
                    fprintf(stderr, "[raype stamps] first tile top -> last tile end over all workgroups: %.1f us\n", (double)(hi - lo) * 0.01);
                }
                return hipGetLastError();
            }
#endif
            if (hipError_t e = once_n.ensure(reinterpret_cast<const void*>(&raype_onepass_kernel<false>), lds_f); e != hipSuccess) return e;
            hipLaunchKernelGGL(raype_onepass_kernel<false>, dim3(P), dim3(kFThreads), lds_f, s, a, ntiles, P);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(raype_hidden_kernel, dim3(P), dim3(kFThreads), lds_a, s, a, ntiles, P);
    a.Whi = reinterpret_cast<const _Float16*>(W2hi); a.Wlo = reinterpret_cast<const _Float16*>(W2lo); a.bias = b2;
    a.feat = feat; a.out = out; a.nchw_out = nchw_out;
    hipLaunchKernelGGL(raype_tokens_kernel, dim3(P), dim3(kFThreads), lds_b, s, a, ntiles, P);
    return hipGetLastError();
}

// Y[M][N] = act(X[M][K] @ W^T + bias) (+ NCHW features); W given as fp16 hi/lo [N][K]; K % 64 == 0
hipError_t launch_gemm_split(const float* X, int64_t ldx, const void* Whi, const void* Wlo, const float* bias, float* Y,
                             int64_t ldy, int M, int N, int K, int relu, const float* feat, int hw, hipStream_t s,
                             const float* scale_dev, float scale_mul, const float* xscale_dev, int accumulate) {
    if (K % kBK != 0 || M < 1 || N < 1 || (relu && feat)) return hipErrorInvalidValue;
    static DynLdsOnce once;
    const size_t ldsb = 4 * 64 * 65 * sizeof(float);                    // 66560 B >= the 64 KB of operand staging
    if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&gemm_split_kernel), ldsb); e != hipSuccess) return e;
    GemmArgs a;
    a.X = X; a.ldx = ldx; a.Whi = reinterpret_cast<const _Float16*>(Whi); a.Wlo = reinterpret_cast<const _Float16*>(Wlo);
    a.bias = bias; a.Y = Y; a.ldy = ldy; a.M = M; a.N = N; a.K = K; a.relu = relu; a.feat = feat; a.hw = hw;
    a.scale_dev = scale_dev; a.scale_mul = scale_mul; a.xscale_dev = xscale_dev; a.accumulate = accumulate;
    const int nct = ceil_div(N, kBN), nrt = ceil_div(M, kBM);
    const int64_t wgs = (int64_t)ceil_div(nrt, 8) * 8 * nct;
    if (wgs > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)wgs), dim3(kThreads), ldsb, s, a);
    return hipGetLastError();
}

}  // namespace parq
