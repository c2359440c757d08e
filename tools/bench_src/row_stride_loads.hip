// micro-benchmark (round 6): what the A-operand request pattern of the chain tiles costs by row stride.
// A workgroup of 8 waves asks for a 16-row x 1024-float tile the way chain_linear_h3_kernel does — lane (row = l & 15, kq = l >> 4),
// two float4 per 32-float chunk at row * stride + wave * 32 + kq * 8 + 256 c: one request instruction = 16 rows x 64 bytes — and,
// for comparison, the same bytes as whole-kilobyte requests (a wave = one row segment of 1 KB per instruction).  256 row tiles
// worth of workgroups (16 "column tiles" per row tile, as in an N = 1024 launch) over an M x stride matrix; the sum keeps the loads
// alive.  Row strides: 1024 floats (4 KB: every row of a request lands in the same 256-byte interleave slot) and padded ones.
// Build: hipcc --offload-arch=gfx950 -O3 -o row_stride_loads row_stride_loads.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>          // 0: MFMA-operand pattern (16 rows x 64 B per request), 1: one row segment of 1 KB per request
__global__ __launch_bounds__(512) void tile_loads(const float* __restrict__ X, int64_t stride, int row_tiles, float* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (blockIdx.x % row_tiles) * 16;
    f32x4 v[8];
    if (MODE == 0) {
        const int li = lane & 15, kq = lane >> 4;
        const float* p = X + (int64_t)(m0 + li) * stride + wave * 32 + kq * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[2 * c] = *reinterpret_cast<const f32x4*>(p + c * 256); v[2 * c + 1] = *reinterpret_cast<const f32x4*>(p + c * 256 + 4); }
    } else {
        // wave w owns rows 2w, 2w + 1: four 1 KB requests per row
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[4 * r + c] = *reinterpret_cast<const f32x4*>(X + (int64_t)(m0 + 2 * wave + r) * stride + c * 256 + lane * 4);
    }
    f32x4 s = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += v[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[blockIdx.x * 512 + tid] = s[0];
}

// The whole operand set of an N = 1024 launch of chain_linear_h3_kernel (16-row tiles, four 16-column sub-tiles, K = 1024, 8 waves):
// A as above (PAT 0 / 1) or not at all (PAT 2), W = 32 requests of 1 KB per wave from the fragment-ordered mirror (or none: WON = 0).
template <int PAT, int WON>
__global__ __launch_bounds__(512) void operand_loads(const float* __restrict__ X, const float* __restrict__ Wh, int ntn, float* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0t = (blockIdx.x % ntn) * 4, m0 = (blockIdx.x / ntn) * 16;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[8];
    if (PAT == 0) {
        const int li = lane & 15, kq = lane >> 4;
        const float* p = X + (int64_t)(m0 + li) * 1024 + wave * 32 + kq * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[2 * c] = *reinterpret_cast<const f32x4*>(p + c * 256); v[2 * c + 1] = *reinterpret_cast<const f32x4*>(p + c * 256 + 4); }
    } else if (PAT == 1) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[4 * r + c] = *reinterpret_cast<const f32x4*>(X + (int64_t)(m0 + 2 * wave + r) * 1024 + c * 256 + lane * 4);
    }
    f32x4 w[32];
    if (WON) {
        const f32x4* wb = reinterpret_cast<const f32x4*>(Wh) + ((int64_t)n0t * 32 + wave) * 128 + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) { w[(t * 4 + c) * 2] = wb[t * 32 * 128 + c * 8 * 128]; w[(t * 4 + c) * 2 + 1] = wb[t * 32 * 128 + c * 8 * 128 + 64]; }
    }
    if (PAT != 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    if (WON) {
#pragma unroll
        for (int i = 0; i < 32; ++i) s += w[i];
    }
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[blockIdx.x * 512 + tid] = s[0];
}

template <int PAT, int WON>
static void run_operands(const char* what, const float* X, const float* Wh, int ntn, int rows, float* out) {
    const int grid = ntn * rows, reps = 200;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((operand_loads<PAT, WON>), dim3(grid), dim3(512), 0, 0, X, Wh, ntn, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((operand_loads<PAT, WON>), dim3(grid), dim3(512), 0, 0, X, Wh, ntn, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("N = %4d (%3d workgroups)  %-52s %6.2f us per launch\n", ntn * 64, grid, what, ms * 1e3 / reps);
}

int main() {
    {
        float *X, *Wh, *o2;
        CK(hipMalloc(&X, 256 * 1024 * 4)); CK(hipMemset(X, 0, 256 * 1024 * 4));
        CK(hipMalloc(&Wh, (size_t)3072 * 1024 * 4)); CK(hipMemset(Wh, 0, (size_t)3072 * 1024 * 4));
        CK(hipMalloc(&o2, 4096 * 512 * 4));
        for (int ntn : {16, 48}) {
            run_operands<2, 0>("nothing (launch floor)", X, Wh, ntn, 16, o2);
            run_operands<0, 0>("A: 16 rows x 64 B per request", X, Wh, ntn, 16, o2);
            run_operands<1, 0>("A: 1 KB of a row per request", X, Wh, ntn, 16, o2);
            run_operands<2, 1>("W fragments (256 KB per workgroup)", X, Wh, ntn, 16, o2);
            run_operands<0, 1>("A (16 x 64 B) + W", X, Wh, ntn, 16, o2);
            run_operands<1, 1>("A (1 KB requests) + W", X, Wh, ntn, 16, o2);
        }
        CK(hipFree(X)); CK(hipFree(Wh)); CK(hipFree(o2));
    }

    const int M = 256, reps = 200;
    const int strides[] = {1024, 1024 + 16, 1024 + 32, 1024 + 64, 1024 + 128, 3072, 3072 + 64};
    float* out; CK(hipMalloc(&out, 4096 * 512 * 4));
    for (int st : strides) {
        float* X; CK(hipMalloc(&X, (size_t)M * st * 4)); CK(hipMemset(X, 0, (size_t)M * st * 4));
        for (int mode = 0; mode < 2; ++mode)
            for (int col_tiles : {16, 48}) {
                const int grid = (M / 16) * col_tiles;
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int w = 0; w < 20; ++w) { if (mode == 0) hipLaunchKernelGGL(tile_loads<0>, dim3(grid), dim3(512), 0, 0, X, (int64_t)st, M / 16, out); else hipLaunchKernelGGL(tile_loads<1>, dim3(grid), dim3(512), 0, 0, X, (int64_t)st, M / 16, out); }
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) { if (mode == 0) hipLaunchKernelGGL(tile_loads<0>, dim3(grid), dim3(512), 0, 0, X, (int64_t)st, M / 16, out); else hipLaunchKernelGGL(tile_loads<1>, dim3(grid), dim3(512), 0, 0, X, (int64_t)st, M / 16, out); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("row stride %5d floats  %-28s  %3d workgroups  %6.2f us per launch  (%.0f KB per workgroup)\n", st, mode == 0 ? "16 rows x 64 B per request" : "1 KB of one row per request", grid, ms * 1e3 / reps, 64.0);
            }
        CK(hipFree(X));
    }
    return 0;
}
