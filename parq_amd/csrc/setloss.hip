// Set loss of the decoder outputs on the device, given the matching (model/parq_decoder.py:264-370; host mirror: parq_amd/loss.py
// decoder_loss_batched, whose tail this replaces for device tensors).  The matcher (scipy LSAP + the np.random.choice cap) stays on the
// host as in the reference; what remains is ~40 tiny torch launches forward and ~60 in autograd per step, 4 ms in which the device
// idles.  Here: three launches produce the four loss terms AND d term / d output for the four differentiable outputs — each term
// depends on exactly one output tensor (centre <- center_unnormalized, size <- size_unnormalized, rotation <- ortho6d, class <-
// pred_logits), so the backward of the autograd node is four scalings.
//
//   pairs (k, b, q, g): prediction q of scene b in iteration k is matched to box g; coef = 1 / (pairs of (k, b) * valid_bs)
//     centre   : w0 * coef * mean_3 |c_pred - c_box|                                   (:288-292)
//     size     : w1 * coef * mean_3 |s_pred - s_box|                                   (:303-307)
//     rotation : w2 * coef * min over the box's symmetry candidates of mean_9 (R(o6) - R_box Ry(2 pi j / m))^2   (:205-262, :309-330)
//   rows (k, b, q): class target = label of the matched box, else background; row_weight = valid(k, b) * punish / sum(punish) / valid_bs
//     class    : w3 * row_weight * class_weight[c] * (logsumexp(logits) - logits[c])     (:332-362)
#include "common.hpp"
#include <cmath>
#include <cstdint>

namespace parq {

namespace {

struct SetLossArgs {
    const float *logits, *center, *size, *o6;          // (I, B, Q, ncls | 3 | 3 | 6)
    const float *t_center, *t_size, *t_rot;            // (B, nmax, 3 | 3 | 9)
    const int32_t *t_label, *t_sym;                    // (B, nmax); t_sym may be null
    const int32_t* pairs;                              // [4][P]: k, b, q, g
    const float* coef;                                 // [P]
    const float* row_weight;                           // [I*B*Q]
    const float* class_weight;                         // [ncls]
    float w[4];
    int I, B, Q, ncls, nmax, P, background;
    float* terms;                                      // [4]
    float *g_logits, *g_center, *g_size, *g_o6;
    int32_t* cls;                                      // [I*B*Q] scratch: class target per row
    float ry_c[42], ry_s[42];                          // cos / sin of the candidates: 2-fold at 0, 4-fold at 2, 36-fold at 6
};

__global__ __launch_bounds__(256) void setloss_init_kernel(SetLossArgs a) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row == 0) { a.terms[0] = 0.f; a.terms[1] = 0.f; a.terms[2] = 0.f; a.terms[3] = 0.f; }
    if (row >= a.I * a.B * a.Q) return;
    a.cls[row] = a.background;
#pragma unroll
    for (int j = 0; j < 3; ++j) { a.g_center[row * 3 + j] = 0.f; a.g_size[row * 3 + j] = 0.f; }
#pragma unroll
    for (int j = 0; j < 6; ++j) a.g_o6[row * 6 + j] = 0.f;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    const float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void setloss_pairs_kernel(SetLossArgs a) {
    __shared__ float sh[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    float lc = 0.f, ls = 0.f, lr = 0.f;
    // (a pair with an index outside the tensors is skipped: nothing is read or written through it)
    const bool in_range = t < a.P && (unsigned)a.pairs[t] < (unsigned)a.I && (unsigned)a.pairs[a.P + t] < (unsigned)a.B &&
                          (unsigned)a.pairs[2 * a.P + t] < (unsigned)a.Q && (unsigned)a.pairs[3 * a.P + t] < (unsigned)a.nmax;
    if (in_range) {
        const int k = a.pairs[t], b = a.pairs[a.P + t], q = a.pairs[2 * a.P + t], g = a.pairs[3 * a.P + t];
        const int64_t row = ((int64_t)k * a.B + b) * a.Q + q;
        const int tb = b * a.nmax + g;
        const float cf = a.coef[t];
        // ---- centre, size: mean absolute error over the 3 components
        {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float dc = a.center[row * 3 + j] - a.t_center[tb * 3 + j];
                const float ds = a.size[row * 3 + j] - a.t_size[tb * 3 + j];
                s0 += fabsf(dc);
                s1 += fabsf(ds);
                a.g_center[row * 3 + j] = sgn(dc) * (a.w[0] * cf / 3.f);
                a.g_size[row * 3 + j] = sgn(ds) * (a.w[1] * cf / 3.f);
            }
            lc = s0 / 3.f * cf * a.w[0];
            ls = s1 / 3.f * cf * a.w[1];
        }
        // ---- rotation: R = Gram-Schmidt(o6) with columns x, y, z (utils/ortho6d_transforms.py:52-66)
        float av[3], bv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { av[j] = a.o6[row * 6 + j]; bv[j] = a.o6[row * 6 + 3 + j]; }
        const float na_raw = sqrtf(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
        const float na = fmaxf(na_raw, 1e-8f);
        const float x[3] = {av[0] / na, av[1] / na, av[2] / na};
        const float c[3] = {x[1] * bv[2] - x[2] * bv[1], x[2] * bv[0] - x[0] * bv[2], x[0] * bv[1] - x[1] * bv[0]};
        const float nc_raw = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        const float nc = fmaxf(nc_raw, 1e-8f);
        const float z[3] = {c[0] / nc, c[1] / nc, c[2] / nc};
        const float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
        float R[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { R[i][0] = x[i]; R[i][1] = y[i]; R[i][2] = z[i]; }
        float T[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) T[i][j] = a.t_rot[tb * 9 + i * 3 + j];
        const int sy = a.t_sym ? a.t_sym[tb] : 0;
        const int m = sy == 1 ? 2 : (sy == 2 ? 4 : (sy == 3 ? 36 : 0));
        const int off = sy == 1 ? 0 : (sy == 2 ? 2 : 6);
        float best = 0.f, D[3][3];
        {
            float e = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) { D[i][j] = R[i][j] - T[i][j]; e += D[i][j] * D[i][j]; }
            best = e / 9.f;
        }
        if (m > 0) {
            best = INFINITY;
            for (int jc = 0; jc < m; ++jc) {
                // candidate = T Ry(theta): Ry = [[c, 0, s], [0, 1, 0], [-s, 0, c]] (utils/parq_utils.py:214-218)
                const float cc = a.ry_c[off + jc], ss = a.ry_s[off + jc];
                float e = 0.f, Dc[3][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float c0 = T[i][0] * cc - T[i][2] * ss, c1 = T[i][1], c2 = T[i][0] * ss + T[i][2] * cc;
                    Dc[i][0] = R[i][0] - c0; Dc[i][1] = R[i][1] - c1; Dc[i][2] = R[i][2] - c2;
                    e += Dc[i][0] * Dc[i][0] + Dc[i][1] * Dc[i][1] + Dc[i][2] * Dc[i][2];
                }
                e /= 9.f;
                if (e < best) {
                    best = e;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) D[i][j] = Dc[i][j];
                }
            }
        }
        lr = best * cf * a.w[2];
        // backward of the Gram-Schmidt: G = d / d R = 2 D / 9 * (w2 coef), columns gx, gy, gz
        const float sc = 2.f / 9.f * cf * a.w[2];
        float gx[3], gy[3], gz[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { gx[i] = D[i][0] * sc; gy[i] = D[i][1] * sc; gz[i] = D[i][2] * sc; }
        // y = z x x:  dz += x x gy,  dx += gy x z
        gz[0] += x[1] * gy[2] - x[2] * gy[1]; gz[1] += x[2] * gy[0] - x[0] * gy[2]; gz[2] += x[0] * gy[1] - x[1] * gy[0];
        gx[0] += gy[1] * z[2] - gy[2] * z[1]; gx[1] += gy[2] * z[0] - gy[0] * z[2]; gx[2] += gy[0] * z[1] - gy[1] * z[0];
        // z = c / max(|c|, 1e-8)
        float gc[3];
        {
            const float dot = nc_raw > 1e-8f ? z[0] * gz[0] + z[1] * gz[1] + z[2] * gz[2] : 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) gc[i] = (gz[i] - z[i] * dot) / nc;
        }
        // c = x x b:  dx += b x gc,  db = gc x x
        gx[0] += bv[1] * gc[2] - bv[2] * gc[1]; gx[1] += bv[2] * gc[0] - bv[0] * gc[2]; gx[2] += bv[0] * gc[1] - bv[1] * gc[0];
        const float gb[3] = {gc[1] * x[2] - gc[2] * x[1], gc[2] * x[0] - gc[0] * x[2], gc[0] * x[1] - gc[1] * x[0]};
        // x = a / max(|a|, 1e-8)
        {
            const float dot = na_raw > 1e-8f ? x[0] * gx[0] + x[1] * gx[1] + x[2] * gx[2] : 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                a.g_o6[row * 6 + i] = (gx[i] - x[i] * dot) / na;
                a.g_o6[row * 6 + 3 + i] = gb[i];
            }
        }
        const int lab = a.t_label[tb];
        a.cls[row] = (unsigned)lab < (unsigned)a.ncls ? lab : a.background;
    }
    lc = block_sum(lc, sh);
    ls = block_sum(ls, sh);
    lr = block_sum(lr, sh);
    if (threadIdx.x == 0) { atomicAdd(a.terms + 0, lc); atomicAdd(a.terms + 1, ls); atomicAdd(a.terms + 2, lr); }
}

__global__ __launch_bounds__(256) void setloss_rows_kernel(SetLossArgs a) {
    __shared__ float sh[4];
    const int row = blockIdx.x * 256 + threadIdx.x;
    float lk = 0.f;
    if (row < a.I * a.B * a.Q) {
        const float* x = a.logits + (int64_t)row * a.ncls;
        float* g = a.g_logits + (int64_t)row * a.ncls;
        const int c = a.cls[row];
        float mx = x[0];
        for (int j = 1; j < a.ncls; ++j) mx = fmaxf(mx, x[j]);
        float se = 0.f;
        for (int j = 0; j < a.ncls; ++j) se += expf(x[j] - mx);
        const float lse = mx + logf(se);
        const float wc = a.class_weight[c], rw = a.row_weight[row] * a.w[3];
        lk = rw * wc * (lse - x[c]);
        const float s = rw * wc;
        for (int j = 0; j < a.ncls; ++j) g[j] = s * (expf(x[j] - lse) - (j == c ? 1.f : 0.f));
    }
    lk = block_sum(lk, sh);
    if (threadIdx.x == 0) atomicAdd(a.terms + 3, lk);
}

}  // namespace

hipError_t launch_set_loss(const float* logits, const float* center, const float* size, const float* o6, int I, int B, int Q, int ncls,
                           const float* t_center, const float* t_size, const float* t_rot, const int32_t* t_label, const int32_t* t_sym,
                           int nmax, const int32_t* pairs, const float* coef, int P, const float* row_weight, const float* class_weight,
                           const float* w4, int background, float* terms, float* g_logits, float* g_center, float* g_size, float* g_o6,
                           int32_t* cls, hipStream_t s) {
    SetLossArgs a;
    a.logits = logits; a.center = center; a.size = size; a.o6 = o6;
    a.t_center = t_center; a.t_size = t_size; a.t_rot = t_rot; a.t_label = t_label; a.t_sym = t_sym;
    a.pairs = pairs; a.coef = coef; a.row_weight = row_weight; a.class_weight = class_weight;
    for (int i = 0; i < 4; ++i) a.w[i] = w4[i];
    a.I = I; a.B = B; a.Q = Q; a.ncls = ncls; a.nmax = nmax; a.P = P; a.background = background;
    a.terms = terms; a.g_logits = g_logits; a.g_center = g_center; a.g_size = g_size; a.g_o6 = g_o6; a.cls = cls;
    // float32 roundings of cos / sin evaluated in double, as torch.tensor([[math.cos(t), ...]]) does (parq_amd/loss.py roty)
    const int ms[3] = {2, 4, 36}, offs[3] = {0, 2, 6};
    for (int c = 0; c < 3; ++c)
        for (int j = 0; j < ms[c]; ++j) {
            const double th = (j * 2.0 / ms[c]) * M_PI;
            a.ry_c[offs[c] + j] = (float)std::cos(th);
            a.ry_s[offs[c] + j] = (float)std::sin(th);
        }
    const int rows = I * B * Q;
    hipLaunchKernelGGL(setloss_init_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, s, a);
    if (P > 0) hipLaunchKernelGGL(setloss_pairs_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(setloss_rows_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace parq
