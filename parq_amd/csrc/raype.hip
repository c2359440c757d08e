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
    const float bv = ncol < a.N ? a.bias[ncol] : 0.f;
    for (int r = 0; r < 64; ++r) {
        const int m = m0 + wr * 64 + r;
        if (m >= a.M) break;
        if (ncol < a.N) {
            float y = ot[r * 65 + lane] + bv;
            if (a.relu) y = y > 0.f ? y : 0.f;
            a.Y[(int64_t)m * a.ldy + ncol] = y;
        }
    }
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

// Y[M][N] = act(X[M][K] @ W^T + bias) (+ NCHW features); W given as fp16 hi/lo [N][K]; K % 64 == 0
hipError_t launch_gemm_split(const float* X, int64_t ldx, const void* Whi, const void* Wlo, const float* bias, float* Y,
                             int64_t ldy, int M, int N, int K, int relu, const float* feat, int hw, hipStream_t s) {
    if (K % kBK != 0 || M < 1 || N < 1 || (relu && feat)) return hipErrorInvalidValue;
    static bool attr_set = false;
    const size_t ldsb = 4 * 64 * 65 * sizeof(float);                    // 66560 B >= the 64 KB of operand staging
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    GemmArgs a;
    a.X = X; a.ldx = ldx; a.Whi = reinterpret_cast<const _Float16*>(Whi); a.Wlo = reinterpret_cast<const _Float16*>(Wlo);
    a.bias = bias; a.Y = Y; a.ldy = ldy; a.M = M; a.N = N; a.K = K; a.relu = relu; a.feat = feat; a.hw = hw;
    const int nct = ceil_div(N, kBN), nrt = ceil_div(M, kBM);
    const int64_t wgs = (int64_t)ceil_div(nrt, 8) * 8 * nct;
    if (wgs > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)wgs), dim3(kThreads), ldsb, s, a);
    return hipGetLastError();
}

}  // namespace parq
