// Eval post-processing on the device: PARQDecoder.parse_pred (model/parq_decoder.py:372-424) with the
// class-agnostic / same-class 3-D NMS of utils/nms.py:20-70,141-226 — one workgroup per scene, no host round trip
// (the reference moves boxes to the CPU for NumPy NMS and back).
//   per query: rotation from the 6-D representation (Gram-Schmidt, utils/ortho6d_transforms.py:52-66), the box as a
//   19-vector [-s/2, s/2 per axis | R t | label], axis-aligned bounds of its 8 corners in the local frame, score = max class
//   probability, label = first arg-max, validity window on the centre (x and z, strict inequalities);
//   NMS: candidates = non-background boxes, visited by descending score (ties: higher index first, NumPy's order for
//   argsort(...)[-1] is unspecified there), IoU of the axis-aligned bounds in float64, suppress when IoU (times "same class"
//   in the visualisation variant) > threshold.
#include "common.hpp"

namespace parq {

namespace {

constexpr int kMaxQ = 1024;

struct ParseArgs {
    const float* center; const float* size; const float* rot6; const float* prob;
    int Q, ncls, num_semcls;
    float tx0, tx1, tz0, tz1;     // TRACK_SCALE[0], [1], [4], [5]
    int for_vis, enable_nms;
    float* obbs;                  // (B, Q, 19)
    unsigned char* mask;          // (B, Q)
};

__global__ __launch_bounds__(256) void parse_pred_kernel(ParseArgs a) {
    __shared__ float lo[3][kMaxQ], hi[3][kMaxQ], score[kMaxQ];
    __shared__ int label[kMaxQ], order[kMaxQ];
    __shared__ unsigned char alive[kMaxQ], valid[kMaxQ], picked[kMaxQ];
    __shared__ int n_cand;
    const int b = blockIdx.x, Q = a.Q;
    for (int q = threadIdx.x; q < Q; q += blockDim.x) {
        const int64_t m = (int64_t)b * Q + q;
        const float* o6 = a.rot6 + m * 6;
        // x = unit(a), z = unit(x cross b), y = z cross x; matrix columns (x, y, z)
        float x[3] = {o6[0], o6[1], o6[2]}, yr[3] = {o6[3], o6[4], o6[5]};
        float n = fmaxf(sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), 1e-8f);
        x[0] /= n; x[1] /= n; x[2] /= n;
        float z[3] = {x[1] * yr[2] - x[2] * yr[1], x[2] * yr[0] - x[0] * yr[2], x[0] * yr[1] - x[1] * yr[0]};
        n = fmaxf(sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]), 1e-8f);
        z[0] /= n; z[1] /= n; z[2] /= n;
        const float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
        const float R[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]};
        const float* c = a.center + m * 3;
        const float* s = a.size + m * 3;
        // class score / label: first maximum (torch.max)
        float best = a.prob[m * a.ncls];
        int bl = 0;
        for (int k = 1; k < a.ncls; ++k) {
            const float p = a.prob[m * a.ncls + k];
            if (p > best) { best = p; bl = k; }
        }
        float* ob = a.obbs + m * 19;
        for (int i = 0; i < 3; ++i) { ob[2 * i] = -s[i] / 2.f; ob[2 * i + 1] = s[i] / 2.f; }
        for (int i = 0; i < 9; ++i) ob[6 + i] = R[i];
        for (int i = 0; i < 3; ++i) ob[15 + i] = c[i];
        ob[18] = (float)bl;
        // bounds of the 8 corners p = R c + t, c = (+-sx/2, +-sy/2, +-sz/2)
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 8; ++k) {
            const float cx = (k & 1) ? s[0] / 2.f : -s[0] / 2.f, cy = (k & 2) ? s[1] / 2.f : -s[1] / 2.f, cz = (k & 4) ? s[2] / 2.f : -s[2] / 2.f;
            for (int i = 0; i < 3; ++i) {
                const float p = cx * R[i * 3] + cy * R[i * 3 + 1] + cz * R[i * 3 + 2] + c[i];
                mn[i] = fminf(mn[i], p);
                mx[i] = fmaxf(mx[i], p);
            }
        }
        for (int i = 0; i < 3; ++i) { lo[i][q] = mn[i]; hi[i][q] = mx[i]; }
        score[q] = best;
        label[q] = bl;
        valid[q] = a.for_vis ? 1 : (c[0] > a.tx0 && c[0] < a.tx1 && c[2] > a.tz0 && c[2] < a.tz1);
        alive[q] = bl != a.num_semcls;            // background boxes never enter the NMS
        picked[q] = 0;
    }
    __syncthreads();
    if (!a.enable_nms) {
        for (int q = threadIdx.x; q < Q; q += blockDim.x) a.mask[(int64_t)b * Q + q] = valid[q];   // reference: pred_mask undefined; keep valid
        return;
    }
    // rank of every candidate by (score, index) ascending; order[rank] = index
    if (threadIdx.x == 0) n_cand = 0;
    __syncthreads();
    for (int q = threadIdx.x; q < Q; q += blockDim.x) {
        if (!alive[q]) continue;
        int r = 0;
        for (int p = 0; p < Q; ++p)
            if (alive[p] && (score[p] < score[q] || (score[p] == score[q] && p < q))) ++r;
        order[r] = q;
        atomicAdd(&n_cand, 1);
    }
    __syncthreads();
    const float thr = a.for_vis ? 0.2f : 0.1f;
    const int nc = n_cand;
    for (int pos = nc - 1; pos >= 0; --pos) {
        const int i = order[pos];
        if (alive[i]) {                                             // uniform: every thread reads the same flag
            if (threadIdx.x == 0) picked[i] = 1;
            const double vi = ((double)hi[0][i] - lo[0][i]) * ((double)hi[1][i] - lo[1][i]) * ((double)hi[2][i] - lo[2][i]);
            for (int pp = threadIdx.x; pp < pos; pp += blockDim.x) {
                const int p = order[pp];
                if (!alive[p]) continue;
                const double l = fmax(0.0, fmin((double)hi[0][i], (double)hi[0][p]) - fmax((double)lo[0][i], (double)lo[0][p]));
                const double w = fmax(0.0, fmin((double)hi[1][i], (double)hi[1][p]) - fmax((double)lo[1][i], (double)lo[1][p]));
                const double h = fmax(0.0, fmin((double)hi[2][i], (double)hi[2][p]) - fmax((double)lo[2][i], (double)lo[2][p]));
                const double inter = l * w * h;
                const double vp = ((double)hi[0][p] - lo[0][p]) * ((double)hi[1][p] - lo[1][p]) * ((double)hi[2][p] - lo[2][p]);
                double o = inter / (vi + vp - inter);
                if (a.for_vis && label[i] != label[p]) o = 0.0;      // nms_3d_faster_samecls
                if (o > (double)thr) alive[p] = 0;
            }
        }
        __syncthreads();
    }
    for (int q = threadIdx.x; q < Q; q += blockDim.x) a.mask[(int64_t)b * Q + q] = picked[q] && valid[q];
}

}  // namespace

hipError_t launch_parse_pred(const float* center, const float* size, const float* rot6, const float* prob, int B, int Q, int ncls,
                             int num_semcls, const float* track_scale6, int for_vis, int enable_nms, float* obbs,
                             unsigned char* mask, hipStream_t s) {
    if (Q > kMaxQ || Q < 1) return hipErrorInvalidValue;
    ParseArgs a;
    a.center = center; a.size = size; a.rot6 = rot6; a.prob = prob; a.Q = Q; a.ncls = ncls; a.num_semcls = num_semcls;
    a.tx0 = track_scale6[0]; a.tx1 = track_scale6[1]; a.tz0 = track_scale6[4]; a.tz1 = track_scale6[5];
    a.for_vis = for_vis; a.enable_nms = enable_nms; a.obbs = obbs; a.mask = mask;
    hipLaunchKernelGGL(parse_pred_kernel, dim3(B), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace parq
