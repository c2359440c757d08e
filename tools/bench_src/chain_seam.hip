// micro-benchmark: a dependent chain of L small linears  X <- relu(X W_s^T + b_s)  (M = 256 rows, 256 -> 256, exact fp32 MFMA,
// 16x16 output tiles: the geometry of the decoder's per-iteration chain at one scene), run
//   A. as L dependent kernel launches (what the library does), with the argument block / tile->XCD mapping varied, and
//   B. as ONE persistent launch in which the 16 workgroups that share a 16-row block ("team") hand their tiles to each other
//      through memory: write-through (sc1) stores -> vmcnt(0) -> one agent-scope counter add per workgroup; the consumers poll that
//      counter relaxed and read the rows with sc1 loads (MI355X guide, Guideline 16 form R1: no fences, placement-independent).
//      Variants: team spread over all XCDs / team on one XCD; plain stores instead of sc1 (valid on one XCD only: measured to price
//      the same-L2 path, NOT a form the library may use); every k-th seam chip-wide (all 16 teams) instead of team-local.
// Every variant is checked against a float64 host evaluation of the chain (a stale read shows up as a wrong result).
// Build: hipcc --offload-arch=gfx950 -O3 -o chain_seam chain_seam.hip      (add -mllvm -amdgpu-kernarg-preload-count=8 for the
// preload variant)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int M = 256, C = 256;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum { LD_PLAIN = 0, LD_SC1 = 1 };
enum { ST_PLAIN = 0, ST_SC1 = 1 };

template <int LD>
__device__ __forceinline__ f32x4 load16(const float* base, int elem_off) {
    if constexpr (LD == LD_PLAIN) {
        return *reinterpret_cast<const f32x4*>(base + elem_off);
    } else {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, elem_off * 4, 0, 16));
    }
}
template <int ST>
__device__ __forceinline__ void store16(float* base, int elem_off, f32x4 v) {
    if constexpr (ST == ST_PLAIN) {
        *reinterpret_cast<f32x4*>(base + elem_off) = v;
    } else {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, elem_off * 4, 0, 16);
    }
}

// one 16x16 output tile (rows m0.., cols n0..) of relu(X W^T + b); NW waves split K = 256 (16 chunks of 16, dealt round-robin);
// lane (li = l & 15, kq = l >> 4) loads the float4 at k = chunk * 16 + kq * 4 of row li of X and of W (row-contiguous requests).
// W / bias loads are issued by `tile_load_w` so that a caller can put them in front of a wait.
template <int NW>
struct TileW { f32x4 bv[16 / NW]; f32x4 bias; };

template <int NW>
__device__ __forceinline__ void tile_load_w(const float* W, const float* bias, int n0, TileW<NW>& t) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int c = 0; c < 16 / NW; ++c) t.bv[c] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + li) * C + (wave + c * NW) * 16 + kq * 4);
    t.bias = *reinterpret_cast<const f32x4*>(bias + n0 + (tid & 3) * 4);   // unconditional: a branch here makes hipcc wait vmcnt(0) inside it
}

template <int NW, int LD, int ST>
__device__ __forceinline__ void tile_compute(const float* X, float* Y, int m0, int n0, const TileW<NW>& t, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    f32x4 av[16 / NW];
#pragma unroll
    for (int c = 0; c < 16 / NW; ++c) av[c] = load16<LD>(X, (m0 + li) * C + (wave + c * NW) * 16 + kq * 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 16 / NW; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][e], t.bv[c][e], acc, 0, 0, 0);
    // acc[r]: row 4 * (lane >> 4) + r, col lane & 15
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (tid < 64) {
        const int row = tid >> 2, c4 = (tid & 3) * 4;
        const int src = (row & 3) * 64 + (row >> 2) * 16 + c4;
        f32x4 s = *reinterpret_cast<const f32x4*>(&red[src]);
#pragma unroll
        for (int w = 1; w < NW; ++w) s += *reinterpret_cast<const f32x4*>(&red[w * 256 + src]);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float y = s[e] + t.bias[e]; o[e] = y > 0.f ? y : 0.f; }
        store16<ST>(Y, (m0 + row) * C + n0 + c4, o);
    }
}

// tile -> (row block, column block) of launch block b.  map 0: column tiles fastest (library order: a row block's 16 tiles sit on
// all 8 XCDs); map 1: XCD-affine (block b runs on XCD b % 8 [observed]: row blocks 2x, 2x+1 live on XCD x)
__device__ __forceinline__ void tile_of(int b, int map, int& rb, int& cb) {
    if (map == 0) { rb = b >> 4; cb = b & 15; }
    else { const int x = b & 7, loc = b >> 3; rb = x * 2 + (loc >> 4); cb = loc & 15; }
}

// ---------------------------------------------------------------- A: one launch per layer
struct FatArgs { const float* X; const float* W; const float* bias; float* Y; int map; int pad[83]; };   // 368 bytes like LinearArgs

template <int NW>
__global__ __launch_bounds__(NW * 64) void layer_fat(FatArgs a) {
    __shared__ __attribute__((aligned(16))) float red[NW * 256];
    int rb, cb; tile_of(blockIdx.x, a.map, rb, cb);
    TileW<NW> t;
    tile_load_w<NW>(a.W, a.bias, cb * 16, t);
    tile_compute<NW, LD_PLAIN, ST_PLAIN>(a.X, a.Y, rb * 16, cb * 16, t, red);
}
template <int NW>
__global__ __launch_bounds__(NW * 64) void layer_thin(const float* X, const float* W, const float* bias, float* Y, int map) {
    __shared__ __attribute__((aligned(16))) float red[NW * 256];
    int rb, cb; tile_of(blockIdx.x, map, rb, cb);
    TileW<NW> t;
    tile_load_w<NW>(W, bias, cb * 16, t);
    tile_compute<NW, LD_PLAIN, ST_PLAIN>(X, Y, rb * 16, cb * 16, t, red);
}
// 14 copies of the same layer at 14 code addresses: a chain of DIFFERENT kernels (what the decoder iteration is) against a chain of
// one kernel whose instructions stay in the instruction cache
template <int ID>
__global__ __launch_bounds__(256) void layer_id(const float* X, const float* W, const float* bias, float* Y, int map) {
    __shared__ __attribute__((aligned(16))) float red[4 * 256];
    if (map == 1000 + ID) return;                      // keeps the instantiations distinct
    int rb, cb; tile_of(blockIdx.x, map, rb, cb);
    TileW<4> t;
    tile_load_w<4>(W, bias, cb * 16, t);
    tile_compute<4, LD_PLAIN, ST_PLAIN>(X, Y, rb * 16, cb * 16, t, red);
}
typedef void (*layer_fn)(const float*, const float*, const float*, float*, int);
static layer_fn layer_table[14] = {layer_id<0>, layer_id<1>, layer_id<2>, layer_id<3>, layer_id<4>, layer_id<5>, layer_id<6>,
                                   layer_id<7>, layer_id<8>, layer_id<9>, layer_id<10>, layer_id<11>, layer_id<12>, layer_id<13>};
// D: the same layer with BOTH operands and the output in tile order: 16 x 16 blocks of 1 KB, element (r, k) of block (rb, kc) at
// float ((k % 16) / 4 * 16 + r) * 4 + k % 4 — every wave-wide float4 load / store is one contiguous KB instead of 16 segments of 64
// bytes at the row stride (1 KB here, 4 KB at K = 1024: the segments of a load then share an L2 channel)
__global__ __launch_bounds__(256) void layer_tiled(const float* Xt, const float* Wt, const float* bias, float* Yt, int map) {
    __shared__ __attribute__((aligned(16))) float red[4 * 256];
    int rb, cb; tile_of(blockIdx.x, map, rb, cb);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 av[4], bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        av[c] = *reinterpret_cast<const f32x4*>(Xt + ((size_t)(rb * 16 + wave + c * 4) * 64 + lane) * 4);
        bv[c] = *reinterpret_cast<const f32x4*>(Wt + ((size_t)(cb * 16 + wave + c * 4) * 64 + lane) * 4);
    }
    const f32x4 bs = *reinterpret_cast<const f32x4*>(bias + cb * 16 + (tid & 3) * 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][e], bv[c][e], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (tid < 64) {
        const int row = tid >> 2, c4 = (tid & 3) * 4;
        const int src = (row & 3) * 64 + (row >> 2) * 16 + c4;
        f32x4 s = *reinterpret_cast<const f32x4*>(&red[src]);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(&red[w * 256 + src]);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float y = s[e] + bs[e]; o[e] = y > 0.f ? y : 0.f; }
        // output block (rb, cb) of the NEXT layer's A operand: (row, k = c4 .. c4 + 3) -> float ((c4 / 4) * 16 + row) * 4
        *reinterpret_cast<f32x4*>(Yt + ((size_t)(rb * 16 + cb) * 64 + (c4 >> 2) * 16 + row) * 4) = o;
    }
}
__global__ void empty_kernel(int) {}
// keeps the device busy for `us` microseconds so that the host can enqueue the whole chain behind it: the timed region then
// measures the DEVICE's dependent-dispatch rate, not the host's launch rate (3 - 4 us per hipLaunchKernelGGL)
__global__ void delay_kernel(int us) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
}

// ---------------------------------------------------------------- B: persistent chain with in-launch seams
// cnt[s * 16 + team]: tiles of layer s finished by the team (zeroed by the host before every launch); fail: raised on a spin timeout
template <int NW, int ST>
__global__ __launch_bounds__(NW * 64) void chain_persistent(const float* W, const float* bias, float* buf0, float* buf1, unsigned* cnt,
                                                             int L, int map, int gseam, unsigned* fail, long long* stamps) {
    __shared__ __attribute__((aligned(16))) float red[NW * 256];
    int rb, cb; tile_of(blockIdx.x, map, rb, cb);
    const int tid = threadIdx.x;
    const bool stamp = stamps != nullptr && (blockIdx.x == 0 || blockIdx.x == 255);
    long long* st = stamps ? stamps + (blockIdx.x == 0 ? 0 : 1) * 64 * 4 : nullptr;
    for (int s = 0; s < L; ++s) {
        const float* Ws = W + (size_t)s * C * C;
        TileW<NW> t;
        tile_load_w<NW>(Ws, bias + s * C, cb * 16, t);          // independent of the seam: in flight while we wait
        const long long t0 = stamp ? clock64() : 0;
        if (s > 0) {
            if (gseam > 0 && (s % gseam) == 0) {
                // chip-wide seam: every team must have finished layer s - 1 (16 counters, one per lane)
                if (tid < 16) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(cnt + (s - 1) * 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) { *fail = 1; break; }
                    }
                }
            } else if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt + (s - 1) * 16 + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { *fail = 1; break; }
                }
            }
            __syncthreads();
        }
        const long long t1 = stamp ? clock64() : 0;
        const float* in = (s & 1) ? buf1 : buf0;
        float* out = (s & 1) ? buf0 : buf1;
        tile_compute<NW, LD_SC1, ST>(in, out, rb * 16, cb * 16, t, red);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its write-through stores
        __syncthreads();
        const long long t2 = stamp ? clock64() : 0;
        if (tid == 0) __hip_atomic_fetch_add(cnt + s * 16 + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stamp && tid == 0 && s < 64) { st[s * 4 + 0] = t0; st[s * 4 + 1] = t1; st[s * 4 + 2] = t2; st[s * 4 + 3] = wall_clock64(); }
    }
}

// counters only: the price of the seam without any payload
template <int NW>
__global__ __launch_bounds__(NW * 64) void seam_only(unsigned* cnt, int L, int map, int gseam, unsigned* fail) {
    int rb, cb; tile_of(blockIdx.x, map, rb, cb);
    const int tid = threadIdx.x;
    for (int s = 0; s < L; ++s) {
        if (s > 0) {
            if (gseam > 0 && (s % gseam) == 0) {
                if (tid < 16) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(cnt + (s - 1) * 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) { *fail = 1; break; }
                    }
                }
            } else if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt + (s - 1) * 16 + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { *fail = 1; break; }
                }
            }
            __syncthreads();
        }
        if (tid == 0) __hip_atomic_fetch_add(cnt + s * 16 + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void census(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(6164) & 15u; }   // HW_REG_XCC_ID
__global__ void thrash(const float4* p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *sink = acc;
}

// ---------------------------------------------------------------- host
static std::vector<float> hW, hB, hX;
static void reference(int L, std::vector<double>& out) {
    std::vector<double> cur(hX.begin(), hX.end()), nxt((size_t)M * C);
    for (int s = 0; s < L; ++s) {
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < C; ++n) {
                double a = hB[(size_t)s * C + n];
                const float* w = &hW[((size_t)s * C + n) * C];
                const double* x = &cur[(size_t)m * C];
                for (int k = 0; k < C; ++k) a += x[k] * (double)w[k];
                nxt[(size_t)m * C + n] = a > 0 ? a : 0;
            }
        cur.swap(nxt);
    }
    out = cur;
}
static double max_err(const float* dev, const std::vector<double>& ref) {
    std::vector<float> h((size_t)M * C);
    CK(hipMemcpy(h.data(), dev, h.size() * 4, hipMemcpyDeviceToHost));
    double e = 0;
    for (size_t i = 0; i < h.size(); ++i) { const double d = fabs((double)h[i] - ref[i]) / fmax(1.0, fabs(ref[i])); if (!(d <= e)) e = d; }
    return e;
}

int main(int argc, char** argv) {
    const int LMAX = 56;
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    hW.resize((size_t)LMAX * C * C); hB.resize((size_t)LMAX * C); hX.resize((size_t)M * C);
    unsigned rng = 12345u;
    auto uni = [&]() { rng = rng * 1664525u + 1013904223u; return (float)((rng >> 8) & 0xffffff) / 16777216.f * 2.f - 1.f; };
    const float a = sqrtf(6.f / C);
    for (auto& w : hW) w = uni() * a;
    for (auto& b : hB) b = uni() * 0.1f;
    for (auto& x : hX) x = uni();
    float *dW, *dB, *d0, *d1, *dBig, *dSink; unsigned *dCnt, *dFail, *dCensus; long long* dSt;
    const size_t big = (size_t)640 << 20;
    CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&d0, hX.size() * 4)); CK(hipMalloc(&d1, hX.size() * 4));
    CK(hipMalloc(&dBig, big)); CK(hipMalloc(&dSink, 4)); CK(hipMalloc(&dCnt, LMAX * 16 * 4)); CK(hipMalloc(&dFail, 4)); CK(hipMalloc(&dCensus, 2048 * 4));
    CK(hipMalloc(&dSt, 2 * 64 * 4 * 8));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dBig, 0, big)); CK(hipMemset(dFail, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> ref14, ref28, ref56;
    reference(14, ref14); reference(28, ref28); reference(56, ref56);
    auto refL = [&](int L) -> const std::vector<double>& { return L == 14 ? ref14 : L == 28 ? ref28 : ref56; };

    {   // census: does block b run on XCD b % 8?
        hipLaunchKernelGGL(census, dim3(2048), dim3(64), 0, 0, dCensus);
        std::vector<unsigned> h(2048); CK(hipMemcpy(h.data(), dCensus, 2048 * 4, hipMemcpyDeviceToHost));
        int ok = 0; for (int b = 0; b < 2048; ++b) ok += (h[b] == (unsigned)(b & 7));
        printf("census: %d of 2048 blocks run on XCD blockIdx %% 8\n", ok);
    }
    auto sweep = [&]() { hipLaunchKernelGGL(thrash, dim3(2048), dim3(256), 0, 0, (const float4*)dBig, big / 16, dSink); };
    // time `fn` (which enqueues one whole chain) `reps` times, each after a cache sweep (cold) or back to back (warm); median in us
    auto timeit = [&](auto fn, bool cold) {
        std::vector<float> ts;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipMemcpyAsync(d0, hX.data(), hX.size() * 4, hipMemcpyHostToDevice, 0));
            CK(hipMemsetAsync(dCnt, 0, LMAX * 16 * 4, 0));
            if (cold) sweep();
            hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, 0, 400);
            CK(hipEventRecord(e0, 0));
            fn();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        return ts[ts.size() / 2];
    };
    auto report = [&](const char* name, int L, float cold, float warm, const float* result) {
        unsigned f = 0; CK(hipMemcpy(&f, dFail, 4, hipMemcpyDeviceToHost));
        printf("%-58s L=%2d  cold %7.2f us (%5.2f/layer)  warm %7.2f us (%5.2f/layer)  max err %.2e%s\n", name, L, cold, cold / L, warm, warm / L,
               max_err(result, refL(L)), f ? "  SPIN TIMEOUT" : "");
        fflush(stdout);
    };

    // ---- A: launches
    for (int L : {14, 28}) {
        for (int map = 0; map < 2; ++map) {
            auto fat = [&]() {
                for (int s = 0; s < L; ++s) {
                    FatArgs fa; memset(&fa, 0, sizeof(fa));
                    fa.X = (s & 1) ? d1 : d0; fa.Y = (s & 1) ? d0 : d1; fa.W = dW + (size_t)s * C * C; fa.bias = dB + s * C; fa.map = map;
                    hipLaunchKernelGGL(layer_fat<4>, dim3(256), dim3(256), 0, 0, fa);
                }
            };
            auto thin4 = [&]() { for (int s = 0; s < L; ++s) hipLaunchKernelGGL(layer_thin<4>, dim3(256), dim3(256), 0, 0, (s & 1) ? d1 : d0, dW + (size_t)s * C * C, dB + s * C, (s & 1) ? d0 : d1, map); };
            auto thin8 = [&]() { for (int s = 0; s < L; ++s) hipLaunchKernelGGL(layer_thin<8>, dim3(256), dim3(512), 0, 0, (s & 1) ? d1 : d0, dW + (size_t)s * C * C, dB + s * C, (s & 1) ? d0 : d1, map); };
            char nm[128];
            float c, w;
            c = timeit(fat, true); w = timeit(fat, false);
            snprintf(nm, sizeof nm, "A launches, 368-byte args, 4 waves, map %d", map); report(nm, L, c, w, (L & 1) ? d1 : d0);
            c = timeit(thin4, true); w = timeit(thin4, false);
            snprintf(nm, sizeof nm, "A launches, 5 plain args,   4 waves, map %d", map); report(nm, L, c, w, (L & 1) ? d1 : d0);
            c = timeit(thin8, true); w = timeit(thin8, false);
            snprintf(nm, sizeof nm, "A launches, 5 plain args,   8 waves, map %d", map); report(nm, L, c, w, (L & 1) ? d1 : d0);
        }
    }
    for (int L : {14, 28}) {
        // C: the same chain through 14 distinct kernels (cold = L2 swept before the chain: code AND weights come from HBM for every
        // layer of the first 14; warm = back to back)
        for (int map = 0; map < 2; ++map) {
            auto distinct = [&]() { for (int s = 0; s < L; ++s) hipLaunchKernelGGL(layer_table[s % 14], dim3(256), dim3(256), 0, 0, (s & 1) ? d1 : d0, dW + (size_t)s * C * C, dB + s * C, (s & 1) ? d0 : d1, map); };
            char nm[128];
            const float c = timeit(distinct, true), w = timeit(distinct, false);
            snprintf(nm, sizeof nm, "C launches of 14 DISTINCT kernels, 4 waves, map %d", map); report(nm, L, c, w, (L & 1) ? d1 : d0);
        }
    }
    {   // D: tile-ordered operands.  Host packs W per layer and X0; the result is compared after un-tiling.
        std::vector<float> hWt(hW.size()), hXt(hX.size());
        auto tile_idx = [](int r, int k, int ncb) { return ((size_t)((r / 16) * ncb + k / 16) * 64 + ((k % 16) / 4) * 16 + r % 16) * 4 + k % 4; };
        for (int s2 = 0; s2 < LMAX; ++s2)
            for (int n = 0; n < C; ++n)
                for (int k = 0; k < C; ++k) hWt[(size_t)s2 * C * C + tile_idx(n, k, C / 16)] = hW[((size_t)s2 * C + n) * C + k];
        for (int m = 0; m < M; ++m)
            for (int k = 0; k < C; ++k) hXt[tile_idx(m, k, C / 16)] = hX[(size_t)m * C + k];
        float *dWt, *dUn;
        CK(hipMalloc(&dWt, hWt.size() * 4)); CK(hipMalloc(&dUn, hX.size() * 4));
        CK(hipMemcpy(dWt, hWt.data(), hWt.size() * 4, hipMemcpyHostToDevice));
        for (int L : {14, 28}) {
            for (int map = 0; map < 2; ++map) {
                auto tiled = [&]() { for (int s2 = 0; s2 < L; ++s2) hipLaunchKernelGGL(layer_tiled, dim3(256), dim3(256), 0, 0, (s2 & 1) ? d1 : d0, dWt + (size_t)s2 * C * C, dB + s2 * C, (s2 & 1) ? d0 : d1, map); };
                // timeit uploads hX (row-major) into d0: swap in the tiled image for this section
                std::vector<float> keep = hX; hX = hXt;
                const float c = timeit(tiled, true), w = timeit(tiled, false);
                hX = keep;
                // un-tile the result on the host for the check
                std::vector<float> ht((size_t)M * C), hr((size_t)M * C);
                CK(hipMemcpy(ht.data(), (L & 1) ? d1 : d0, ht.size() * 4, hipMemcpyDeviceToHost));
                for (int m = 0; m < M; ++m) for (int k = 0; k < C; ++k) hr[(size_t)m * C + k] = ht[tile_idx(m, k, C / 16)];
                CK(hipMemcpy(dUn, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
                char nm[128];
                snprintf(nm, sizeof nm, "D launches, TILE-ORDERED X / W / Y, 4 waves, map %d", map); report(nm, L, c, w, dUn);
            }
        }
    }
    {
        auto empties = [&]() { for (int s = 0; s < 14; ++s) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, s); };
        const float w = timeit(empties, false);
        printf("14 empty dependent launches (256 x 256 threads): %.2f us = %.2f per launch\n", w, w / 14);
    }
    // ---- B: persistent, in-launch seams
    for (int L : {14, 28, 56}) {
        for (int map = 0; map < 2; ++map) {
            for (int gseam : {0, 1}) {
                char nm[128];
                float c, w;
                auto p4 = [&]() { hipLaunchKernelGGL((chain_persistent<4, ST_SC1>), dim3(256), dim3(256), 0, 0, dW, dB, d0, d1, dCnt, L, map, gseam, dFail, (long long*)nullptr); };
                auto p8 = [&]() { hipLaunchKernelGGL((chain_persistent<8, ST_SC1>), dim3(256), dim3(512), 0, 0, dW, dB, d0, d1, dCnt, L, map, gseam, dFail, (long long*)nullptr); };
                c = timeit(p4, true); w = timeit(p4, false);
                snprintf(nm, sizeof nm, "B persistent R1 (sc1 st + sc1 ld), 4 waves, map %d, gseam %d", map, gseam); report(nm, L, c, w, (L & 1) ? d1 : d0);
                c = timeit(p8, true); w = timeit(p8, false);
                snprintf(nm, sizeof nm, "B persistent R1 (sc1 st + sc1 ld), 8 waves, map %d, gseam %d", map, gseam); report(nm, L, c, w, (L & 1) ? d1 : d0);
                if (map == 1 && gseam == 0) {
                    auto q4 = [&]() { hipLaunchKernelGGL((chain_persistent<4, ST_PLAIN>), dim3(256), dim3(256), 0, 0, dW, dB, d0, d1, dCnt, L, map, gseam, dFail, (long long*)nullptr); };
                    c = timeit(q4, true); w = timeit(q4, false);
                    snprintf(nm, sizeof nm, "B persistent same-L2 (plain st + sc1 ld), 4 waves, map 1 [one-XCD only]"); report(nm, L, c, w, (L & 1) ? d1 : d0);
                }
                auto so = [&]() { hipLaunchKernelGGL(seam_only<4>, dim3(256), dim3(256), 0, 0, dCnt, L, map, gseam, dFail); };
                w = timeit(so, false);
                printf("%-58s L=%2d  %7.2f us (%5.2f/seam)\n", (std::string("  seams only, map ") + char('0' + map) + ", gseam " + char('0' + gseam)).c_str(), L, w, w / L);
            }
        }
    }
    {   // where a persistent layer spends its time (block 0 and block 255; shader-clock cycles, wall stamps are 10 ns units)
        const int L = 28;
        CK(hipMemcpy(d0, hX.data(), hX.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(dCnt, 0, LMAX * 16 * 4));
        sweep();
        hipLaunchKernelGGL((chain_persistent<4, ST_SC1>), dim3(256), dim3(256), 0, 0, dW, dB, d0, d1, dCnt, L, 1, 0, dFail, dSt);
        CK(hipDeviceSynchronize());
        std::vector<long long> st(2 * 64 * 4); CK(hipMemcpy(st.data(), dSt, st.size() * 8, hipMemcpyDeviceToHost));
        for (int blk = 0; blk < 2; ++blk) {
            const long long* s = st.data() + blk * 256;
            const double cyc_per_10ns = (double)(s[(L - 1) * 4 + 2] - s[0 * 4 + 2]) / (double)(s[(L - 1) * 4 + 3] - s[0 * 4 + 3]);
            double wsum = 0, csum = 0;
            for (int k = 2; k < L; ++k) { wsum += (double)(s[k * 4 + 1] - s[k * 4 + 0]); csum += (double)(s[k * 4 + 2] - s[k * 4 + 1]); }
            printf("persistent R1 map 1, block %3d: shader clock %.0f MHz; per layer: wait at the seam %.2f us, load + MFMA + store + drain %.2f us, layer period %.2f us\n",
                   blk ? 255 : 0, cyc_per_10ns * 100.0, wsum / (L - 2) / (cyc_per_10ns * 100.0), csum / (L - 2) / (cyc_per_10ns * 100.0),
                   (double)(s[(L - 1) * 4 + 3] - s[1 * 4 + 3]) / (L - 2) * 0.01);
        }
    }
    // ---- E: the launch chain of A (5 plain args, 4 waves) on a created stream, as individual launches and as ONE hipGraph replay
    // (stream capture): does the graph form shorten the dependent-launch boundary?
    {
        hipStream_t cs; CK(hipStreamCreate(&cs));
        hipEvent_t g0, g1; CK(hipEventCreate(&g0)); CK(hipEventCreate(&g1));
        for (int L : {14, 28, 56}) {
            auto chain = [&](hipStream_t st) { for (int s = 0; s < L; ++s) hipLaunchKernelGGL(layer_thin<4>, dim3(256), dim3(256), 0, st, (s & 1) ? d1 : d0, dW + (size_t)s * C * C, dB + s * C, (s & 1) ? d0 : d1, 0); };
            hipGraph_t graph; hipGraphExec_t exec;
            CK(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
            chain(cs);
            CK(hipStreamEndCapture(cs, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            auto time_on = [&](auto fn) {
                std::vector<float> ts;
                for (int r = 0; r < reps + 2; ++r) {
                    CK(hipMemcpyAsync(d0, hX.data(), hX.size() * 4, hipMemcpyHostToDevice, cs));
                    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, cs, 400);
                    CK(hipEventRecord(g0, cs));
                    fn();
                    CK(hipEventRecord(g1, cs));
                    CK(hipEventSynchronize(g1));
                    float ms; CK(hipEventElapsedTime(&ms, g0, g1));
                    if (r >= 2) ts.push_back(ms * 1e3f);
                }
                std::sort(ts.begin(), ts.end());
                return ts[ts.size() / 2];
            };
            const float tl = time_on([&]() { chain(cs); });
            const float el = max_err((L & 1) ? d1 : d0, refL(L == 56 ? 56 : L));
            const float tg = time_on([&]() { CK(hipGraphLaunch(exec, cs)); });
            const float eg = max_err((L & 1) ? d1 : d0, refL(L == 56 ? 56 : L));
            printf("E created stream, L=%2d: launches %7.2f us (%5.2f/layer, err %.1e)   hipGraph replay %7.2f us (%5.2f/layer, err %.1e)\n", L, tl, tl / L, el, tg,
                   tg / L, eg);
            fflush(stdout);
            CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
        }
    }
    return 0;
}
