// micro-benchmark: latency of a dependent chain of small linears Y = relu(X W^T + b), M=256, N=256, K in {256,768}
// variants: fp32 MFMA 32x32x2 with 4- or 8-way in-workgroup split-K, and fp16 hi/lo 3-term MFMA 32x32x16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split8(const float* x, half8& hi, half8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const half2v h = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x[e], x[e + 1]));
        const half2v l = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(x[e] - (float)h[0], x[e + 1] - (float)h[1]));
        hi[e] = h[0]; hi[e + 1] = h[1]; lo[e] = l[0]; lo[e + 1] = l[1];
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void lin_f32(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K,
                                                    long long* stamps) {
    __shared__ __attribute__((aligned(16))) float red[NW * 16 * 64];
    const long long t0 = clock64();
    const int ntn = N / 32;
    const int n0 = (blockIdx.x % ntn) * 32, m0 = (blockIdx.x / ntn) * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int KS = K / NW, KH = KS / 2, kbase = wave * KS + kh * KH, nch = KH / 4;
    const float* xr = X + (size_t)(m0 + li) * K + kbase;
    const float* wr = W + (size_t)(n0 + li) * K + kbase;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    long long t1 = 0;
    for (int c0 = 0; c0 < nch; c0 += 8) {
        f32x4 av[8], bv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c0 + c < nch) { av[c] = *(const f32x4*)(xr + (c0 + c) * 4); bv[c] = *(const f32x4*)(wr + (c0 + c) * 4); }
            else { av[c] = f32x4{0, 0, 0, 0}; bv[c] = f32x4{0, 0, 0, 0}; }
        }
        if (c0 == 0 && stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t1 = clock64(); }
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][e], bv[c][e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    const long long t2 = clock64();
    __syncthreads();
    for (int t = tid; t < 256; t += NW * 64) {
        const int row = t >> 3, c4 = (t & 7) * 4, reg = (row & 3) + 4 * (row >> 3), sl = c4 + 32 * ((row >> 2) & 1);
        f32x4 s = *(const f32x4*)&red[reg * 64 + sl];
        for (int w = 1; w < NW; ++w) s += *(const f32x4*)&red[(w * 16 + reg) * 64 + sl];
        f32x4 o;
        for (int e = 0; e < 4; ++e) { float y = s[e] + bias[n0 + c4 + e]; o[e] = y > 0.f ? y : 0.f; }
        *(f32x4*)(Y + (size_t)(m0 + row) * N + n0 + c4) = o;
    }
    if (stamps && tid == 0) { const long long t3 = clock64(); long long* s = stamps + blockIdx.x * 4; s[0] = t1 - t0; s[1] = t2 - t1; s[2] = t3 - t2; s[3] = t3 - t0; }
}

// fp16 hi/lo 3-term.  W pre-split as [N][K] half hi and lo.  lane (li, kh): k run of 8 per 16-wide step.
template <int NW>
__global__ __launch_bounds__(NW * 64) void lin_split(const float* X, const _Float16* Whi, const _Float16* Wlo, const float* bias, float* Y,
                                                      int M, int N, int K, long long* stamps) {
    __shared__ __attribute__((aligned(16))) float red[NW * 16 * 64];
    const long long t0 = clock64();
    const int ntn = N / 32;
    const int n0 = (blockIdx.x % ntn) * 32, m0 = (blockIdx.x / ntn) * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int KS = K / NW;                 // wave slice; steps of 16, lane half takes 8 contiguous
    const int nst = KS / 16;
    const float* xr = X + (size_t)(m0 + li) * K + wave * KS + kh * 8;
    const _Float16* whr = Whi + (size_t)(n0 + li) * K + wave * KS + kh * 8;
    const _Float16* wlr = Wlo + (size_t)(n0 + li) * K + wave * KS + kh * 8;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    long long t1 = 0;
    constexpr int MAXST = 12;
    f32x4 xa[MAXST][2]; half8 wh[MAXST], wl[MAXST];
#pragma unroll
    for (int s = 0; s < MAXST; ++s)
        if (s < nst) {
            xa[s][0] = *(const f32x4*)(xr + s * 16); xa[s][1] = *(const f32x4*)(xr + s * 16 + 4);
            wh[s] = *(const half8*)(whr + s * 16); wl[s] = *(const half8*)(wlr + s * 16);
        }
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t1 = clock64(); }
#pragma unroll
    for (int s = 0; s < MAXST; ++s)
        if (s < nst) {
            float xv[8] = {xa[s][0][0], xa[s][0][1], xa[s][0][2], xa[s][0][3], xa[s][1][0], xa[s][1][1], xa[s][1][2], xa[s][1][3]};
            half8 xh, xl;
            split8(xv, xh, xl);
            // D[i=w row n][j=x row m]: A operand = W, B operand = X  -> accumulators hold Y^T? keep A=X,B=W: D[i=m][j=n]
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl[s], acc, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    const long long t2 = clock64();
    __syncthreads();
    for (int t = tid; t < 256; t += NW * 64) {
        const int row = t >> 3, c4 = (t & 7) * 4, reg = (row & 3) + 4 * (row >> 3), sl = c4 + 32 * ((row >> 2) & 1);
        f32x4 s = *(const f32x4*)&red[reg * 64 + sl];
        for (int w = 1; w < NW; ++w) s += *(const f32x4*)&red[(w * 16 + reg) * 64 + sl];
        f32x4 o;
        for (int e = 0; e < 4; ++e) { float y = s[e] + bias[n0 + c4 + e]; o[e] = y > 0.f ? y : 0.f; }
        *(f32x4*)(Y + (size_t)(m0 + row) * N + n0 + c4) = o;
    }
    if (stamps && tid == 0) { const long long t3 = clock64(); long long* s = stamps + blockIdx.x * 4; s[0] = t1 - t0; s[1] = t2 - t1; s[2] = t3 - t2; s[3] = t3 - t0; }
}

__global__ void split_w(const float* W, _Float16* hi, _Float16* lo, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const _Float16 h = (_Float16)W[i]; hi[i] = h; lo[i] = (_Float16)(W[i] - (float)h); }
}

int main() {
    const int M = 256, N = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int K : {256, 768}) {
        // chain needs N == K for ping-pong; use separate X (M x K) buffers: Y (M x 256) feeds only when K == 256;
        // for K=768 the input is a fixed buffer rewritten by a tiny dependent kernel (still a dependent launch chain).
        float *X, *Y, *W, *b; _Float16 *Wh, *Wl; long long* st;
        hipMalloc(&X, M * 768 * 4); hipMalloc(&Y, M * 768 * 4); hipMalloc(&W, N * K * 4); hipMalloc(&b, N * 4);
        hipMalloc(&Wh, N * K * 2); hipMalloc(&Wl, N * K * 2); hipMalloc(&st, 64 * 4 * 8);
        std::vector<float> h(M * 768);
        for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
        hipMemcpy(X, h.data(), M * 768 * 4, hipMemcpyHostToDevice); hipMemcpy(Y, h.data(), M * 768 * 4, hipMemcpyHostToDevice);
        std::vector<float> hw(N * K);
        for (auto& v : hw) v = (rand() % 2001 - 1000) * 6e-5f;
        hipMemcpy(W, hw.data(), N * K * 4, hipMemcpyHostToDevice); hipMemset(b, 0, N * 4);
        hipLaunchKernelGGL(split_w, dim3((N * K + 255) / 256), dim3(256), 0, 0, W, Wh, Wl, N * K);
        const int reps = 200;
        auto timeit = [&](const char* name, auto launch) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int i = 0; i < reps; ++i) launch((i & 1) ? Y : X, (i & 1) ? X : Y, (long long*)nullptr);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            launch(X, Y, st); hipDeviceSynchronize();
            long long hs[64 * 4]; hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
            double a[4] = {0, 0, 0, 0};
            for (int i = 0; i < 64; ++i) for (int j = 0; j < 4; ++j) a[j] += hs[i * 4 + j] / 64.0;
            printf("K=%d %-22s %.2f us/launch | in-kernel clocks (100MHz ticks x10ns): load %.0f  mfma %.0f  epi %.0f  total %.0f\n", K, name,
                   ms * 1e3 / reps, a[0], a[1], a[2], a[3]);
        };
        timeit("f32 4 waves", [&](float* in, float* out, long long* s) { hipLaunchKernelGGL(lin_f32<4>, dim3(64), dim3(256), 0, 0, in, W, b, out, M, N, K, s); });
        timeit("f32 8 waves", [&](float* in, float* out, long long* s) { hipLaunchKernelGGL(lin_f32<8>, dim3(64), dim3(512), 0, 0, in, W, b, out, M, N, K, s); });
        timeit("split 4 waves", [&](float* in, float* out, long long* s) { hipLaunchKernelGGL(lin_split<4>, dim3(64), dim3(256), 0, 0, in, Wh, Wl, b, out, M, N, K, s); });
        timeit("split 2 waves", [&](float* in, float* out, long long* s) { hipLaunchKernelGGL(lin_split<2>, dim3(64), dim3(128), 0, 0, in, Wh, Wl, b, out, M, N, K, s); });
        if (K == 256) timeit("split 8 waves", [&](float* in, float* out, long long* s) { hipLaunchKernelGGL(lin_split<8>, dim3(64), dim3(512), 0, 0, in, Wh, Wl, b, out, M, N, K, s); });
    }
    return 0;
}
