// project + sample of one (scene, query): the body shared by project_sample_kernel (elementwise.hip) and the fused position-MLP +
// sample launch (chain.hip).  model/transformer_parq.py:129-161; Pose.transform / Camera.project, utils/wrappers.py:259-267,502-522.
#pragma once
#include "common.hpp"

namespace parq {

// bq: (scene, query) index of this workgroup; nbq: number of such workgroups in the launch (the zeroing of `zero_f64` is spread over
// them); smem: [nwv][C] partial sums + [nwv] counts + V footprints of 32 bytes (launch_project_sample_t sizes it)
template <int NCH, typename TPose>
__device__ __forceinline__ void project_sample_body(
    const float* __restrict__ tokens, const TPose* __restrict__ T_cl, const float* __restrict__ cam,
    const float* __restrict__ ref, ScaleBox sb, int V, int h, int w, int C, int Q, float* __restrict__ tgt,
    float* __restrict__ coord_pos, double* __restrict__ zero_f64, int zero_n, float* __restrict__ raw_count, int bq, int nbq,
    float* smem) {
    for (int i = bq * blockDim.x + threadIdx.x; i < zero_n; i += nbq * blockDim.x)
        zero_f64[i] = 0.0;                                          // accumulators of later kernels
    const int b = bq / Q;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int nwv = blockDim.x >> 6;
    const int C4 = C >> 2;

    // denormalize (transformer_parq.py:198-209).  coord_pos mirrors the reference's float32
    // mul-then-add (no FMA contraction); the projection uses the float64 value.
    double P[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float r = ref[(int64_t)bq * 3 + i];
        P[i] = (double)r * ((double)sb.hi[i] - (double)sb.lo[i]) + (double)sb.lo[i];
        if (coord_pos && threadIdx.x == i)
            coord_pos[(int64_t)bq * 3 + i] = __fadd_rn(__fmul_rn(r, __fsub_rn(sb.hi[i], sb.lo[i])), sb.lo[i]);
    }

    // ---- geometry ONCE per (query, view): thread v projects into view v (float64) and leaves the bilinear footprint in LDS.  Every
    // lane of a wave used to repeat its view's projection: ~150 quarter-rate float64 instructions per wave whatever the number of
    // active lanes, i.e. 10 waves x 2400 cycles per workgroup — at 32 scenes the kernel was bound by that arithmetic (75 us per
    // launch against ~150 MB of actual fetches), at one scene it was 1.5 of its 2.8 us in-kernel time.
    struct Foot { int x0, y0; float w00, w01, w10, w11; int flags, pad; };      // flags: bit 0 valid view, bit 1 a corner is in range
    Foot* foot = reinterpret_cast<Foot*>(smem + (size_t)nwv * C + nwv);
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const TPose* T = T_cl + ((int64_t)b * V + v) * 12;
        const float* cm = cam + ((int64_t)b * V + v) * 6;
        // Pose.transform: p @ R^T + t (utils/wrappers.py:259-267)
        const double x = P[0] * (double)T[0] + P[1] * (double)T[1] + P[2] * (double)T[2] + (double)T[9];
        const double y = P[0] * (double)T[3] + P[1] * (double)T[4] + P[2] * (double)T[5] + (double)T[10];
        const double z = P[0] * (double)T[6] + P[1] * (double)T[7] + P[2] * (double)T[8] + (double)T[11];
        // Camera.project (utils/wrappers.py:511-522); eps is the float32 value of 1e-3
        const double eps = (double)1e-3f;
        const bool front = z > eps;
        const double zc = z > eps ? z : eps;
        const double u = (x / zc) * (double)cm[2] + (double)cm[4];
        const double vv = (y / zc) * (double)cm[3] + (double)cm[5];
        const bool valid = front && (u >= 0.0) && (u <= (double)cm[0] - 1.0) && (vv >= 0.0) && (vv <= (double)cm[1] - 1.0);
        // grid_sample(bilinear, zeros, align_corners=True): the normalised-grid round trip of
        // transformer_parq.py:148-152 is the identity on pixel coordinates
        const double fx0 = floor(u);
        const double fy0 = floor(vv);
        // any corner in range?  (tests in floating point: |u| can be ~1e6 when z was clamped)
        const bool on = fx0 >= -1.0 && fx0 <= (double)(w - 1) && fy0 >= -1.0 && fy0 <= (double)(h - 1);
        Foot f;
        f.x0 = on ? (int)fx0 : 0;
        f.y0 = on ? (int)fy0 : 0;
        const float wx1 = (float)(u - fx0), wx0 = (float)(1.0 - (u - fx0));
        const float wy1 = (float)(vv - fy0), wy0 = (float)(1.0 - (vv - fy0));
        const bool x0ok = f.x0 >= 0, x1ok = f.x0 + 1 <= w - 1;
        const bool y0ok = f.y0 >= 0, y1ok = f.y0 + 1 <= h - 1;
        f.w00 = (x0ok && y0ok) ? wy0 * wx0 : 0.f;   // nw
        f.w01 = (x1ok && y0ok) ? wy0 * wx1 : 0.f;   // ne
        f.w10 = (x0ok && y1ok) ? wy1 * wx0 : 0.f;   // sw
        f.w11 = (x1ok && y1ok) ? wy1 * wx1 : 0.f;   // se
        f.flags = (valid ? 1 : 0) | (on ? 2 : 0);
        f.pad = 0;
        foot[v] = f;
    }
    __syncthreads();

    f32x4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int nvalid = 0;

    for (int v = wv; v < V; v += nwv) {
        const Foot f = foot[v];
        nvalid += f.flags & 1;
        if (!(f.flags & 2)) continue;
        const int x0 = f.x0, y0 = f.y0;
        const bool x0ok = x0 >= 0, x1ok = x0 + 1 <= w - 1;
        const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= h - 1;
        const float* base = tokens + (((int64_t)b * V + v) * h) * (int64_t)w * C;
        const float* r00 = base + ((int64_t)(y0ok ? y0 : 0) * w + (x0ok ? x0 : 0)) * C;
        const float* r01 = base + ((int64_t)(y0ok ? y0 : 0) * w + (x1ok ? x0 + 1 : 0)) * C;
        const float* r10 = base + ((int64_t)(y1ok ? y0 + 1 : 0) * w + (x0ok ? x0 : 0)) * C;
        const float* r11 = base + ((int64_t)(y1ok ? y0 + 1 : 0) * w + (x1ok ? x0 + 1 : 0)) * C;
        const float w00 = f.w00, w01 = f.w01, w10 = f.w10, w11 = f.w11;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int c4 = lane + c * 64;
            if (c4 < C4) {
                const f32x4 a00 = reinterpret_cast<const f32x4*>(r00)[c4];
                const f32x4 a01 = reinterpret_cast<const f32x4*>(r01)[c4];
                const f32x4 a10 = reinterpret_cast<const f32x4*>(r10)[c4];
                const f32x4 a11 = reinterpret_cast<const f32x4*>(r11)[c4];
                acc[c] += a00 * w00 + a01 * w01 + a10 * w10 + a11 * w11;
            }
        }
    }

    // cross-view reduction in view-slot order
    float* part = smem;
    int* cnt = reinterpret_cast<int*>(smem + (size_t)nwv * C);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int c4 = lane + c * 64;
        if (c4 < C4) reinterpret_cast<f32x4*>(part + (size_t)wv * C)[c4] = acc[c];
    }
    if (lane == 0) cnt[wv] = nvalid;
    __syncthreads();
    int total = 0;
    for (int i = 0; i < nwv; ++i) total += cnt[i];
    // view-sharded scenes (raw_count != nullptr): this rank holds only some of the scene's views, so it leaves the UNDIVIDED sum of
    // its views in tgt and its number of valid views in raw_count; the caller adds the ranks' pairs and divides (sample_finalize)
    const float denom = raw_count ? 1.f : (float)(total > 0 ? total : 1);
    if (raw_count && threadIdx.x == 0) raw_count[bq] = (float)total;
    for (int c4 = threadIdx.x; c4 < C4; c4 += blockDim.x) {
        f32x4 s = reinterpret_cast<const f32x4*>(part)[c4];
        for (int i = 1; i < nwv; ++i) s += reinterpret_cast<const f32x4*>(part + (size_t)i * C)[c4];
        reinterpret_cast<f32x4*>(tgt + (int64_t)bq * C)[c4] = s / denom;
    }
}

}  // namespace parq
