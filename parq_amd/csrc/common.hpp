// Shared declarations of the gfx950 kernel chain behind libparq_hip.so.
// Everything here targets CDNA4 (wave64; v_mfma_f32_32x32x16_f16 split products and fp32 MFMA) directly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

namespace parq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Development knobs (A/B switches, kernels with ingredients removed) are read from the environment ONLY in the development build
// (-DPARQ_DEV_PROBES -> parq_amd/_C/libparq_hip_dev.so, loaded by tools/ through PARQ_HIP_LIB).  The product library never calls
// getenv: every knob takes its default and the probe instantiations (wrong results by construction) are not compiled in.
#ifdef PARQ_DEV_PROBES
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

// ---- in-kernel time stamps (development build only): where a launch spends its time between two kernel boundaries.
// Wave 0 of every workgroup of an instrumented kernel appends {kernel id | grid size | XCD, linear block index, start, end} to a
// device buffer (parq_dev_timeline); stamps are s_memrealtime (100 MHz, chip-wide), `end` is taken after the wave's own stores
// have been acknowledged.  tools/iter_timeline_stamps.py turns the records into a per-launch table (first / last workgroup start,
// last end, gap to the previous launch's last end = the dependent-dispatch boundary).
enum : int { kTlLinear = 1, kTlProjectSample = 2, kTlSelfAttn = 3, kTlFlashSplit = 4, kTlFlashMerge = 5, kTlBoxDecode = 6, kTlPosemb = 7,
             kTlKvProj = 8, kTlChain = 9 };
#ifdef PARQ_DEV_PROBES
static __device__ unsigned long long* parq_tl_buf = nullptr;      // one copy per translation unit (no relocatable device code)
static __device__ unsigned int parq_tl_cap = 0;
struct TlStamp {
    unsigned long long t0, idx; int id;
    unsigned long long ph = 0ull; int nph = 0;        // up to four phase marks inside the kernel: 11-bit offsets from t0 in 10 ns ticks
    __device__ __forceinline__ void mark() {
        if (threadIdx.x == 0 && parq_tl_buf) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long d = wall_clock64() - t0;
            d = d > 2047ull ? 2047ull : d;
            ph |= d << (11 * nph);
            ++nph;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the record slot is claimed at the START (the atomic's round trip overlaps the kernel); the end costs one wait for the wave's
    // own stores and four fire-and-forget stores
    __device__ __forceinline__ TlStamp(int id_) : id(id_) {
        t0 = 0ull; idx = ~0ull;
        if (threadIdx.x == 0 && parq_tl_buf) { t0 = wall_clock64(); idx = atomicAdd(parq_tl_buf, 1ull); }
    }
    __device__ __forceinline__ ~TlStamp() {
        if (threadIdx.x == 0 && parq_tl_buf) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t1 = wall_clock64();
            if (idx < parq_tl_cap) {
                unsigned long long* r = parq_tl_buf + 4 + idx * 4;
                const unsigned long long nblk = (unsigned long long)gridDim.x * gridDim.y * gridDim.z;
                const unsigned long long xcc = __builtin_amdgcn_s_getreg(6164) & 15u;     // HW_REG_XCC_ID
                r[0] = (unsigned long long)id | (nblk << 8) | (xcc << 40);
                r[1] = (blockIdx.x + (unsigned long long)gridDim.x * (blockIdx.y + (unsigned long long)gridDim.y * blockIdx.z)) | (ph << 20);   // (block index < 2^20 wherever marks are used)
                r[2] = t0; r[3] = t1;
            }
        }
    }
};
#define PARQ_TL_KERNEL(id) ::parq::TlStamp parq_tl_stamp_(id)
// phase marks cost a pointer load and a branch each even with the timeline off: only in builds that ask for them
// (PARQ_DEV_EXTRA_FLAGS=-DPARQ_TL_MARKS python -c 'import __graft_entry__ as g; g.build_dev()'; tools/r06_h3_phases.py)
#ifdef PARQ_TL_MARKS
#define PARQ_TL_MARK() parq_tl_stamp_.mark()
#else
#define PARQ_TL_MARK() do { } while (0)
#endif
// each instrumented translation unit defines its setter with this macro; api.hip calls them all from parq_dev_timeline()
#define PARQ_TL_DEFINE_SETTER(fn)                                                                              \
    hipError_t fn(unsigned long long* buf, unsigned int cap) {                                                 \
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(parq_tl_buf), &buf, sizeof(buf));                          \
        if (e != hipSuccess) return e;                                                                         \
        return hipMemcpyToSymbol(HIP_SYMBOL(parq_tl_cap), &cap, sizeof(cap));                                  \
    }
hipError_t tl_set_linear(unsigned long long*, unsigned int);
hipError_t tl_set_elementwise(unsigned long long*, unsigned int);
hipError_t tl_set_flash(unsigned long long*, unsigned int);
hipError_t tl_set_flash_split(unsigned long long*, unsigned int);
hipError_t tl_set_flash_split8(unsigned long long*, unsigned int);
hipError_t tl_set_kvproj_split(unsigned long long*, unsigned int);
hipError_t tl_set_chain(unsigned long long*, unsigned int);
#else
#define PARQ_TL_KERNEL(id) do { } while (0)
#define PARQ_TL_MARK() do { } while (0)
#define PARQ_TL_DEFINE_SETTER(fn)
#endif

constexpr int kWave = 64;
// GroupNorm moment slots per (scene, group).  Producers either OWN a slot (chain.hip: sub-tile (row block, column block) of a
// 256-query x 256-channel block -> slot 16 rb + cb, a plain store: no atomic round trip at the end of the launch, and a summation
// order that does not depend on arrival) or add into slot (tile % kGnSlots) with fp64 atomics (generic kernel; slots zeroed by the
// iteration's first kernel).  Consumers sum all slots: 4 per lane and one shuffle tree.
constexpr int kGnSlots = 256;

// LDS-DMA (global_load_lds, 16 bytes per lane, wave-uniform LDS base + lane * 16) issued through inline asm.  hipcc knows that the
// builtin writes LDS asynchronously and answers with s_waitcnt vmcnt(0) in front of later C++ reads of LDS it cannot prove
// disjoint — which turns a prefetch into a blocking load.  Through asm the compiler does not see it: the caller waits with a
// counted s_waitcnt vmcnt(N) + a barrier before anybody reads the destination.
// M0 (the LDS base of the instruction) is saved and restored inside the statement: hipcc does not honour an "m0" clobber, and it may
// keep its own value there (indexed register access, its own LDS-DMA builtins).
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_base) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base);
    unsigned saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved)
                 : "s"(base), "v"(gsrc)
                 : "memory");
}
// workgroup barrier that only orders LDS traffic: __syncthreads() is a release / acquire fence and makes hipcc wait for every
// outstanding global store and load (s_waitcnt vmcnt(0)) first
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: a process that drives several GPUs must set
// it on each of them.  One static instance per kernel; bit d of `done` = already set on device ordinal d.
struct DynLdsOnce {
    unsigned long long done = 0;
    hipError_t ensure(const void* fn, size_t bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const bool track = dev >= 0 && dev < 64;
        if (track && ((done >> dev) & 1ull)) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess && track) done |= 1ull << dev;
        return e;
    }
};

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Row of a 32x32 MFMA C/D tile held in register `reg` of lane `lane`
// (col = lane & 31):  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ int mfma32_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// ------------------------------------------------------------------ fp32 -> (hi, lo) fp16 split
// x ~= hi + lo with hi = fp16(x) and lo = fp16(x - hi).  `hi` MUST be the value the residual is
// taken against: hipcc is free to materialise `(_Float16)x` twice (v_cvt_f16_f32 for the residual,
// v_cvt_pk_f16_f32 for the packed MFMA operand) and on gfx950 the two disagree on exact ties, which
// silently costs a full fp16 ulp (2^-11 relative).  Going through the packed round-toward-zero
// builtin pins one conversion; the residual is derived from ITS result.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_pair(float x0, float x1, half2v& hi, half2v& lo) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    hi = __builtin_bit_cast(half2v, h);
    // x - hi as v_fma_mix_f32(hi (fp16 half of the packed register), -1.0, x): one instruction per element instead of cvt + sub
    // (hipcc only forms the mixed fma when a multiply feeds the subtraction); the difference is exact either way
    float d0, d1;
    const unsigned int hp = __builtin_bit_cast(unsigned int, h);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hp), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hp), "v"(x1));
    lo = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(d0, d1));
}

__device__ __forceinline__ void split8(const float* x, half8& hi, half8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        half2v h, l;
        split_pair(x[e], x[e + 1], h, l);
        hi[e] = h[0]; hi[e + 1] = h[1];
        lo[e] = l[0]; lo[e + 1] = l[1];
    }
}

// ---- fp8 (OCP e4m3) operand forms of the MX cross terms (flash_split8.hip, the K/V projection's mode-4 epilogue)
__device__ __forceinline__ float clamp_e4m3(float x) { return __builtin_amdgcn_fmed3f(x, -448.f, 448.f); }
// e4m3 of four floats, byte i = x[i]; values past +-448 saturate (v_cvt_pk_fp8_f32 itself returns NaN beyond its rounding range)
__device__ __forceinline__ int pack4_e4m3(float a, float b, float c, float d) {
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(a), clamp_e4m3(b), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(c), clamp_e4m3(d), w, true);
}
// x0, x1 -> packed hi16 (round toward zero: the value split_pair takes its residual against) and the fp32 residuals x - hi (exact)
__device__ __forceinline__ void split_rtz(float x0, float x1, unsigned& hi, float& d0, float& d1) {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hi), "v"(x1));
}
constexpr float kLo8Scale = 1024.f;          // lo parts of K, V, Q enter e4m3 as lo * 2^10 (|lo| < 2^-10 |x|); E8M0 scale 117 undoes it
typedef int i32x4 __attribute__((ext_vector_type(4)));
// 16 floats -> 16 bytes e4m3(x) and 16 bytes e4m3((x - hi16(x)) * 2^10), byte i = x[i]
__device__ __forceinline__ void pieces_e4m3(const float* x, i32x4& hi8, i32x4& lo8) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned h0, h1;
        float d[4];
        split_rtz(x[4 * w], x[4 * w + 1], h0, d[0], d[1]);
        split_rtz(x[4 * w + 2], x[4 * w + 3], h1, d[2], d[3]);
        hi8[w] = pack4_e4m3(x[4 * w], x[4 * w + 1], x[4 * w + 2], x[4 * w + 3]);
        lo8[w] = pack4_e4m3(d[0] * kLo8Scale, d[1] * kLo8Scale, d[2] * kLo8Scale, d[3] * kLo8Scale);
    }
}
// byte offsets inside a 64-key stage of the mode-4 cache (layout: flash_split8.hip): K as hi16 + e4m3 hi8 + e4m3 lo8, V as fp16 (round to
// nearest): 24 KB per stage, 3 bytes per element
constexpr int kStage8Bytes = 24576;
constexpr int kS8Kh16 = 0, kS8K8hi = 8192, kS8K8lo = 12288, kS8Vh16 = 16384;

// ---- dropout (training): counter-based keep decision, identical in forward and backward.  Element (row, col) of stream `seed`
// is kept when a 24-bit hash is >= p * 2^24; kept values are scaled by 1 / (1 - p) (torch.nn.Dropout semantics).
__host__ __device__ constexpr uint32_t rng_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t rng_stream(uint32_t base, uint32_t stream) { return rng_mix(base ^ rng_mix(stream * 0x9e3779b9U + 0x85ebca6bU)); }
// The index space of a dropout site is 2-D, (row, col): activations (m, n); attention probabilities (bh * Lq + q, key).  The row
// part of the hash is computed once per row (per lane in the forward attention kernels, per query tile in the backward ones),
// the column part once per column where a kernel can share it (below).
__host__ __device__ inline uint32_t drop_rowhash(uint32_t seed, uint32_t row) { return rng_mix(seed ^ (row * 0x9e3779b1U)); }
// keep(row, col) = mix(rowhash(seed, row) ^ colhash(col)) >= ceil(p * 2^24) * 256: the column part does not depend on the seed, so
// the attention kernels hash a key ONCE (per stage in the forward, per lane in the backward, where a lane owns a key) and spend
// xor + 24-bit multiply + shift-xor + 24-bit multiply + compare per element.  The final mix matters: comparing the bare xor against a threshold makes the
// dropped set of a row the preimage of `rowhash ^ [0, thr)`, whose large dyadic sub-blocks depend only on the top bits of the
// row hash — 1/16 of all row pairs then share >60 % of their dropped columns at p = 0.1 (a 4-wise xor dependence that uniformity
// and pairwise independence do not show).  multiply / shift-xor / multiply after the xor breaks that linearity: over all pairs of
// 512 rows the worst shared-drop fraction is 0.171 - 0.176 at p = 0.1 (a full 32-bit finaliser gives 0.178; one multiply + shift-xor
// still leaves 0.20-0.23 against the 6-sigma bound 0.196).  tests/test_host_cpu.py simulates the hash in NumPy,
// tests/test_gpu_kernels.py::test_dropout_masks_of_row_pairs_overlap_like_independent_draws checks the masks the kernels use.
// Column hash = hash of the 32-column block ^ one of 16 constants chosen by bits 0, 1, 3, 4 of the column ^ a constant where bit 2
// is set.  In the attention kernels a lane's 16 accumulator registers of a 32-key block are the columns (r & 3) + 8 (r >> 2) + 4 kh
// (kh = lane >> 5): the block part is wave-uniform (scalar unit), the kh part folds into the lane's row hash once, and the register
// part is a literal operand of the xor in drop_keep_h — no per-key hashing, no table in LDS, no registers (the forward kernel
// used to hash 32 keys per block, park them in LDS and read 16 back: two dependent LDS round trips at the head of every step,
// +32 % on the training forward).  The fixed xor offsets between the columns of a block are decorrelated by the two multiplies
// of drop_keep_h like the row offsets: co-drop frequency of every in-block column pair within 1.7 % of p^2 over 6 M samples.
__host__ __device__ constexpr uint32_t drop_blockhash(uint32_t blk) { return rng_mix(blk * 0x85ebca77U + 0x6a09e667U); }
__host__ __device__ constexpr uint32_t drop_regpart(uint32_t r) { return rng_mix(0x3c6ef372U + r); }       // r = 0 .. 15
constexpr uint32_t kDropBit2Part = rng_mix(0xa54ff53aU);
__host__ __device__ constexpr uint32_t drop_colhash(uint32_t col) {
    const uint32_t k = col & 31u;
    return drop_blockhash(col >> 5) ^ drop_regpart((k & 3u) | ((k >> 3) << 2)) ^ ((k & 4u) ? kDropBit2Part : 0u);
}
// threshold in the scale of the full 32-bit hash: (h >> 8) >= ceil(p 2^24)  <=>  h >= ceil(p 2^24) * 256 (saturated: p ~ 1 keeps nothing
// but h = 2^32 - 1)
__host__ __device__ inline uint32_t drop_threshold(float p) {
    const float t = ceilf(p * 16777216.0f);
    return t >= 16777216.0f ? 0xffffffffU : ((uint32_t)t << 8);
}
// Both multiplies are 24 x 24-bit (v_mul_u32_u24, full rate; the 32-bit v_mul_lo_u32 is a quarter-rate instruction and two of them per
// probability were ~2/3 of the mask's issue time in the attention kernels): the low 24 bits of the xor of two full 32-bit hashes,
// times a 24-bit odd constant, shift-xor, low 24 bits times a second constant; the compared top bits depend on every one of the 24
// bits.  Same statistics as the 32-bit form in the simulation of tests/test_host_cpu.py (worst shared-drop fraction over all pairs of
// 512 rows 0.171 - 0.176 for three seeds, 32-bit form 0.175).
__host__ __device__ inline bool drop_keep_h(uint32_t rowhash, uint32_t colhash, uint32_t thr) {
    uint32_t h = ((rowhash ^ colhash) & 0xffffffU) * 0x9E3779U;
    h ^= h >> 16;
    h = (h & 0xffffffU) * 0x85EBCBU;
    return h >= thr;
}
__host__ __device__ inline bool drop_keep(uint32_t rowhash, uint32_t col, float p) {
    return drop_keep_h(rowhash, drop_colhash(col), drop_threshold(p));
}

// 16-bit operand kinds of the matrix pipe.  A half8 is used as the raw 8 x 16-bit container for both.
enum : int { kF16 = 0, kBF16 = 1 };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// round-to-nearest conversion of 8 floats (the single-term "half" attention modes)
template <int KIND>
__device__ __forceinline__ half8 cvt8_rn(const float* x) {
    if constexpr (KIND == kF16) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (_Float16)x[e];
        return h;
    } else {
        bf16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (__bf16)x[e];
        return __builtin_bit_cast(half8, h);
    }
}

template <int KIND>
__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    if constexpr (KIND == kF16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------ linear (small GEMM)
struct LinearArgs {
    const float* X;  int64_t ldx;       // A operand, row-major [M][K] with row stride ldx
    const float* X2; int64_t ldx2;      // optional addend on A (e.g. + pos_feat) ...
    int x2_ncols;                       // ... applied only to output columns < x2_ncols (tile granular)
    const float* W;  int64_t ldw;       // [N][K] row-major (torch Linear / Conv1d(k=1) weight)
    const float* bias;                  // [N] or nullptr
    const float* R;  int64_t ldr;       // residual [M][N] or nullptr
    float* Y;
    int M, N, K;
    int relu;
    const float* relu_mask; int64_t ldmask;   // backward of a ReLU: y = relu_mask[m][n] > 0 ? y * mask_scale : 0 (applied after bias)
    float mask_scale;                         // 0 is read as 1
    // training dropout on the output, after bias / ReLU and before the residual: element (m, n) of stream drop_seed
    float drop_p; uint32_t drop_seed;
    // output address: Y + (m / rows_per_batch) * y_batch + (m % rows_per_batch) * y_row
    //                   + (n / col_blk) * y_blk + (n % col_blk)
    int rows_per_batch; int64_t y_batch; int64_t y_row; int col_blk; int64_t y_blk;
    float norm_eps;                     // eps of the fused LayerNorm / GroupNorm
    // LayerNorm applied to the rows of A on load (statistics over the full K computed by the tile):
    //   a' = (a - mean_m) * rstd_m * ln_gamma[k] + ln_beta[k]   (then + X2 if given)
    // ln_stats_out[m] = (mean, rstd) is published by the first column tile for later residual use.
    const float* ln_gamma; const float* ln_beta; float* ln_stats_out;
    // residual taken as LayerNorm(R) recomputed from published row statistics:
    //   r' = (R[m][n] - mean_m) * rstd_m * rln_gamma[n] + rln_beta[n]
    const float* rln_stats; const float* rln_gamma; const float* rln_beta;
    // GroupNorm(1 group over rows_per_scene x K) + ReLU applied to A on load, from the scene-wide
    // moments (sum, sum of squares; fp64) the producer accumulated:
    //   a' = relu((a - mean) * rstd * gamma[k] + beta[k]),  (S, Q) = gn_sums[scene][group]
    const double* gn_sums; const float* gn_gamma; const float* gn_beta;
    int gn_rows_per_scene; int gn_ngroups;
    // moments of the OUTPUT (columns < gn_out_ncols) accumulated with fp64 atomics into
    // gn_out_sums[scene][(n + g*N) / gn_out_group_cols][kGnSlots][2] (zeroed by an earlier kernel)
    double* gn_out_sums; int gn_out_ncols; int gn_out_group_cols; int gn_out_rows_per_scene; int gn_out_ngroups;
    // grouped launch: blockIdx.y = g adds these element offsets
    int64_t gX, gW, gBias, gY, gGamma;
    // chain.hip only: a SECOND operand pair accumulated into the same output, Y += X2 @ W2^T, for output columns < x2_ncols (the
    // position MLP's last layer folded into its consumers: (x + h W2^T + b2) W^T = x W^T + h (W W2)^T + W b2).  X2 / ldx2 above is then
    // the second A operand (K columns) instead of an addend; W2 is [N][K] row-major with row stride ldw, W2p its tile-ordered copy.
    const float* W2; const float* W2p;
    // chain.hip only: the same matrix in tile order (launch_pack_w_tiles): block (n / 16, k / 16) = 1 KB holding element (n % 16,
    // k % 16) at float ((k % 16) / 4 * 16 + n % 16) * 4 + k % 4, i.e. one wave-wide float4 load = one contiguous KB.  nullptr: read W.
    const float* Wp;
    // chain.hip only: the same matrix as fp16 hi / lo pairs in MFMA-fragment order with one power-of-two scale per output column
    // (pack_w_half_kernel; element offsets as W's) and wh_scale[n] = the inverse scale.  nullptr: the fp32 kernels.
    const float* Wh; const float* wh_scale;
    // LayerNorm folds of that mirror (pack_w_half_kernel): wh_fold 1 = gamma inside Wh and W beta inside wh_bias, 2 = W beta inside wh_bias only
    const float* wh_bias; int wh_fold;
    int tile_map;                       // chain.hip only: 1 = the workgroups of a row block run on one XCD (set by launch_chain_linear)
    // chain.hip only — LayerNorm seams inside one launch (seam_tile): a launch that produces rows x which a LayerNorm normalises
    // publishes, per (row, workgroup tile of 64 columns), the fp64 partial sums (sum x, sum x^2): lnp_out[row][N / 64][2]
    double* lnp_out;
    // IN-LAUNCH publication of lnp_out (chain.hip seam kernels): the partials are stored write-through (sc1) and flag
    // lnp_flags[row block][tile] = lnp_epoch follows once they are acknowledged; consumers are workgroups of the SAME launch with
    // larger block indices (seam_tile)
    unsigned* lnp_flags; unsigned lnp_epoch;
    // Early completion signal of a forward (api.hip: set on the first launch behind the LAST iteration's cross-attention merge — no
    // kernel of the forward can raise a flag after it): thread 0 of workgroup (0, 0) copies what the forward has raised into the caller's
    // host-visible mirror word (the bits of BoxDecodeArgs::poison_mirror) and then stores this call's epoch (left in the workspace by the
    // call's prologue launch) into the host-visible progress word, so that a host that only needs to know WHETHER the forward has to
    // be re-run can stop waiting one chain tail (~36 us at BASELINE cfg 3) before the forward ends.  nullptr: nothing to publish.
    const int* pub_flags; int* pub_mirror; int* pub_word; const int* pub_epoch; int pub_mask; int pub_peaky;
};
__device__ __forceinline__ void publish_progress(const LinearArgs& a) {
    if (a.pub_word != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const int f0 = a.pub_flags[0] & a.pub_mask, pk = a.pub_peaky ? a.pub_flags[1] : 0;
        const int bits = ((f0 & ~4) ? 1 : 0) | (f0 & 4) | (pk != 0 ? (2 | (pk << 8)) : 0);
        if (bits != 0 && a.pub_mirror != nullptr) __hip_atomic_fetch_or(a.pub_mirror, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.pub_word, *a.pub_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // behind the mirror bits
    }
}
// A consumer tile of a LayerNorm seam inside ONE launch (chain.hip seam_tile).  The LayerNorm is pushed through the linear map behind it,
//   LN(x) W^T + b = rstd (x Wg^T - mean s) + b',   Wg = W diag(gamma), s = row sums of Wg, b' = b + W beta,
// and x Wg^T is computed from the operands x itself is made of (x = r + y Wo^T + bo  ->  x Wg^T = r Wg^T + y (Wg Wo)^T + Wg bo), so
// the tile's contraction does not wait for x; only its epilogue needs (mean, rstd), which it takes from the fp64 partial row sums the
// x tiles of the same launch publish.  out = act(rstd (acc1 + b1 - mean s) + acc2 + b2):
//   acc1 = X1' W1^T + X2 W2^T   (X1' = X1, or LayerNorm(X1) with GIVEN statistics ln1_stats and gamma / beta ln1_g / ln1_b)
//   acc2 = X3 W3^T              (optional: a part the LayerNorm does not scale)
struct SeamArgs {
    const float* X1; int64_t ldx1; const float* W1p;          // tile-ordered weights (launch_pack_w_tiles), [N][K1]
    const float* X2; int64_t ldx2; const float* W2p;          // [N][K2]
    const float* X3; int64_t ldx3; const float* W3p;          // [N][K3] or nullptr
    const float* ln1_stats; const float* ln1_g; const float* ln1_b;
    const float* b1; const float* srow; const float* b2;      // [N] each (b2 may be nullptr)
    float* Y; int64_t ldy; int M, N;
    const double* part; const unsigned* flags; unsigned epoch; int nparts; int width;   // the producer's partials / flags (nparts <= 4 per row)
    float eps;
    float* ln_out;                                             // optional [M][2]: (mean, rstd) of x published for later launches
    double* gn_out_sums; int gn_out_ncols; int gn_out_group_cols; int gn_out_rows_per_scene; int gn_out_ngroups;   // as LinearArgs
    int* err;                                                  // raised (bit 2) if a flag never arrives: outputs are then NaN
};
hipError_t launch_linear(const LinearArgs& a, int groups, hipStream_t s);
// chain.hip: compile-time specialised kernels for the launches of one decoder iteration; hipErrorNotSupported = no instantiation
// matches this launch (launch_linear then uses the generic kernel)
hipError_t launch_chain_linear(const LinearArgs& a, int groups, hipStream_t s);
bool chain_linear_supported(const LinearArgs& a, int groups);      // would launch_chain_linear take this launch?
// chain.hip: two stages that a LayerNorm separates as ONE launch (SeamArgs below; seam_tile); hipErrorNotSupported = shapes do not fit
hipError_t launch_seam_q(const LinearArgs& xa, const SeamArgs& q, hipStream_t s);
// pack time: Wg = W diag(gamma), srow = row sums of Wg, bb = b + W beta | out = A Bm, ob = A bv  (float64 accumulation)
hipError_t launch_ln_fold(const float* W, const float* b, const float* gamma, const float* beta, int R, int C, float* Wg, float* srow, float* bb, hipStream_t s);
hipError_t launch_matmul_fold(const float* A, const float* Bm, const float* bv, int R, int C, float* out, float* ob, hipStream_t s);
hipError_t launch_pack_w_half(const float* W, int64_t ldw, int N, int K, float* dst, float* scales, hipStream_t s, const float* gamma = nullptr,
                              const float* beta = nullptr, const float* bias = nullptr, float* bias_out = nullptr);
hipError_t launch_pack_w_tiles(const float* W, int64_t ldw, int N, int K, float* dst, hipStream_t s);   // N % 16 == K % 16 == 0
// pack time: out_w[r][k] = sum_j Wa[r][j] W2[j][k], out_b[r] = ba[r] + sum_j Wa[r][j] b2[j]  (float64 accumulation), r < R; Wa [R][C], W2 [C][C]
hipError_t launch_fold_pos_weights(const float* Wa, const float* ba, const float* W2, const float* b2, int R, int C, float* out_w, float* out_b, hipStream_t s);

// ------------------------------------------------------------------ attention
struct FlashArgs {
    const float* q; int64_t q_batch, q_head, q_row;     // element strides
    const float* k; int64_t k_batch, k_head, k_row;
    const float* v; int64_t v_batch, v_head, v_row;
    int B, H, Lq, Lk, dh;
    int nsplit;                 // key splits (each writes one partial)
    float* o_part;              // [B*H][nsplit][dh][Lq_pad]
    float* m_part;              // [B*H][nsplit][Lq_pad]   (log2 domain running max)
    float* l_part;              // [B*H][nsplit][Lq_pad]
    float* out; int64_t out_batch, out_row;              // merged (b, q, h*dh + d)
    float defer_log2;           // split kernel: running max moves only past this margin (0 = always)
    float* lse;                 // optional [B*H][Lq_pad]: log2-domain log-sum-exp of every query row (training)
    float drop_p; uint32_t drop_seed;   // training: dropout on the attention probabilities, element (row bh * Lq + q, col key)
    int flags;                  // bit 0: raise the priority of the younger half of the workgroup (pipelined split kernel)
    // attention mode 4: its error grows as a row concentrates on few keys (flash_split8.hip).  The merge kernel knows every row's
    // sum l of probabilities relative to the row's reference maximum — roughly the number of keys that carry the row — and
    // raises bit h of *peaky (and of *peaky_it, this iteration's own word, if given) when some row of head h has l under peaky_l
    // (nullptr / 0: no check); *peaky_min (optional) keeps the smallest l seen as 0x7fffffff - its bits (atomicMax; 0 = none)
    int* peaky; float peaky_l;
    int* peaky_it; int* peaky_min;
    int* head_min;              // optional [H]: the smallest l of every head of this launch, same code (what a head on the fp16 x 3 tier would
                                // look like to the guard: PARQDecoder.tier_return_after)
    // a launch over SOME of the heads (attention mode 4 with per-head tiers: the heads whose rows rest on few keys run the fp16 x 3
    // kernel, the others the mode-4 kernel, each class with its own launch, key-split count and partial buffers).  nh = 0: all H heads.
    // nh > 0: grid index z = b * nh + i covers head (hmap >> 4 i) & 15; partials are indexed by z, everything else by b * H + head.
    int nh; unsigned long long hmap;
    int64_t cache_head_bytes;   // bytes between the cache regions of two (scene, head) pairs; 0 = the kernel's own packed layout
};
struct FlashHead { int b, h, bh; };
// grid index z of a partial-producing / merging launch -> (scene, head, b * H + head)
__device__ __forceinline__ FlashHead flash_head(const FlashArgs& a, int z) {
    FlashHead r;
    if (a.nh > 0) { r.b = z / a.nh; r.h = (int)((a.hmap >> (4 * (z - r.b * a.nh))) & 15ull); }
    else { r.b = z / a.H; r.h = z - r.b * a.H; }
    r.bh = r.b * a.H + r.h;
    return r;
}
inline int flash_launch_heads(const FlashArgs& a) { return a.nh > 0 ? a.nh : a.H; }
int flash_key_tile(int dh);                    // keys per LDS tile
int flash_lq_pad(int Lq);
int flash_pick_nw(int B, int H, int Lq, int Lk, int dh, int num_cus);   // waves per workgroup
int flash_pick_splits(int B, int H, int Lq, int Lk, int dh, int num_cus);
int device_num_cus();                          // cached CU count of the current device
size_t flash_scratch_bytes(int B, int H, int Lq, int dh, int nsplit);
hipError_t launch_flash(const FlashArgs& a, hipStream_t s);       // partials
hipError_t launch_flash_merge(const FlashArgs& a, hipStream_t s); // partials -> out

// split-precision (fp16 hi/lo, 3-term products) cross-attention, head dim 64 (flash_split.hip).
// `terms` = 3: split precision (fp32-class accuracy); 1: single fp16 / bf16 products (`kind` = kF16 / kBF16),
// cache blocks hold only [K | V] (8 KB instead of 16 KB).
size_t kvsplit_cache_bytes(int B, int H, int N, int terms = 3);
int flash_split_pick_splits(int B, int H, int Lq, int Lk, int num_cus);
// kvproj_big.hip: the K/V projection for C > 256 (pre-split tokens, LDS-DMA operands); scratch = kvproj_big_scratch_floats floats
size_t kvproj_big_scratch_floats(int B, int N, int C);
hipError_t launch_kvproj_big(const float* tokens, const void* Whi, const void* Wlo, const float* bias, int B, int N, int C,
                             void* cache, int* overflow, float* scratch, hipStream_t s, int terms = 3, int kind = kF16);
// flash_split256.hip: the same for head dim 256 (a head = 4 virtual heads of 64 in the cache; wave pairs split the head dim)
int flash_split256_pick_splits(int B, int H, int Lq, int Lk, int num_cus);
hipError_t launch_flash_split256(const FlashArgs& a, const void* cache, hipStream_t s, int terms = 3, int kind = kF16);
hipError_t launch_kvsplit_convert(const float* K, const float* V, int64_t k_batch, int64_t k_head, int64_t k_row,
                                  int64_t v_batch, int64_t v_head, int64_t v_row, int B, int H, int N, void* cache,
                                  int* overflow_flag, hipStream_t s, int terms = 3, int kind = kF16);
// split cache (terms = 3) -> fp32 head-major K / V ([n][64] rows per (b, h))
hipError_t launch_kvsplit_to_f32(const void* cache, int B, int H, int N, float* K, float* V, int64_t k_batch, int64_t k_head,
                                 int64_t v_batch, int64_t v_head, hipStream_t s, int chunks = 1,   // H cache heads = model heads x chunks
                                 int terms = 3, int kind = kF16);                                   // terms = 1: the single 16-bit cache
hipError_t launch_flash_split(const FlashArgs& a, const void* cache, hipStream_t s, int terms = 3,
                              int kind = kF16);   // partials; merge as usual
// flash_split8.hip: attention mode 4 — the scores with their two cross terms on the MX-scaled fp8 matrix instruction, P V in fp16 with a
// self-consistent normaliser (head dim 64, whole 64-key stages only: flash_split8_supported).  Cache = Lk / 64 stages of 24 KB per
// (scene, head) — within the size kvsplit_cache_bytes(.., 3) gives for such Lk.
bool flash_split8_supported(int dh, int Lk);
hipError_t launch_kvsplit8_convert(const float* K, const float* V, int64_t k_batch, int64_t k_head, int64_t k_row, int64_t v_batch,
                                   int64_t v_head, int64_t v_row, int B, int H, int N, void* cache, hipStream_t s);
hipError_t launch_flash_split8(const FlashArgs& a, const void* cache, hipStream_t s);
// attention modes 2 / 3 on whole 64-key stages of the single-product cache (flash_split8.hip); flash_split8_supported(dh, Lk) says where
hipError_t launch_flash_single_stage(const FlashArgs& a, const void* cache, hipStream_t s, int kind);
// kvproj_split.hip: tokens -> split cache directly (W pre-split with launch_split_f32)
hipError_t launch_split_f32(const float* src, void* hi, void* lo, int64_t n, hipStream_t s);
// elementwise.hip: up to kGatherMax device-to-device float copies in ONE launch (the weight pack: ~50 tensors per training step)
constexpr int kGatherMax = 64;
struct GatherArgs { const float* src[kGatherMax]; float* dst[kGatherMax]; int64_t n[kGatherMax]; int count; };
hipError_t launch_gather_copy(const GatherArgs& g, hipStream_t s);
// setloss.hip: loss terms + d term / d output for a given matching (three launches)
hipError_t launch_set_loss(const float* logits, const float* center, const float* size, const float* o6, int I, int B, int Q, int ncls,
                           const float* t_center, const float* t_size, const float* t_rot, const int32_t* t_label, const int32_t* t_sym,
                           int nmax, const int32_t* pairs, const float* coef, int P, const float* row_weight, const float* class_weight,
                           const float* w4, int background, float* terms, float* g_logits, float* g_center, float* g_size, float* g_o6,
                           int32_t* cls, hipStream_t s);
// terms = 11: attention mode 4 with per-head tiers — head h as mode-4 stages, or in the split layout where bit h of safe_mask is set;
// every (scene, head) region of the cache then spans ceil(N / 32) * 16 KB (the split layout's size)
hipError_t launch_kvproj_split(const float* tokens, const void* Whi, const void* Wlo, const float* bias, int B, int N,
                               int C, int H, void* cache, int* overflow, hipStream_t s, int terms = 3, int kind = kF16, unsigned safe_mask = 0);
// fp32 -> 16-bit (round to nearest) weights of the single-term modes
hipError_t launch_cvt16(const float* src, void* dst, int64_t n, int kind, hipStream_t s);

// raype.hip: ray-point positional encoding + tokenisation
hipError_t launch_raype_points(const float* cam, const float* T_cp, const float* T_wp, const float* T_wl,
                               const float* scale6, float min_depth, float max_depth, int B, int V, int h, int w, int S,
                               float* P, hipStream_t s);
hipError_t launch_raype_fused(const float* cam, const float* T_cp, const float* T_wp, const float* T_wl, const float* scale6,
                              float min_depth, float max_depth, int B, int V, int h, int w, const void* W1hi, const void* W1lo,
                              const float* b1, const void* W2hi, const void* W2lo, const float* b2, const float* feat,
                              float* hidden, double* Tl, double* depth, float* out, int nchw_out, hipStream_t s,
                              void* W2f = nullptr, int two_kernels = 0);
// GroupNorm(1, C) statistics from the float64 moments (sum, sum of squares) of a group, all in float64 (the variance is a difference
// of two nearly equal numbers) with one reciprocal of the element count.  Used by the forward consumers and by the backward that
// re-normalises from the same moments.  (An fp32 reciprocal square root was measured: the consumers' times did not move — the
// statistics hide behind the operand fetch — so the float64 form stays.)
__device__ __forceinline__ void gn_mean_rstd(double S, double Q, double inv_cnt, float eps, float& mean, float& rstd) {
    const double mu = S * inv_cnt;
    double var = Q * inv_cnt - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}
struct ScaleBox { float lo[3]; float hi[3]; };
// ---- backward (backward.hip, attn_bwd.hip)
hipError_t launch_transpose(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int R, int Cc, hipStream_t s);
hipError_t launch_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t ldo, int M, int N, int K,
                          int accumulate, hipStream_t s, float* bias = nullptr, int bias_from = 0);   // bias[n] (+)= column sums of A, n >= bias_from
hipError_t launch_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, hipStream_t s);
hipError_t launch_add(const float* a, const float* b, float* y, int64_t n, hipStream_t s);
hipError_t launch_axpy_rows(const float* x, int64_t ldx, float* y, int64_t ldy, int M, int N, int accumulate, hipStream_t s);
hipError_t launch_ln_bwd(const float* gy, const float* x, const float* stats, const float* gamma, float* gx, int M, int C,
                         int accumulate, float* dgamma, float* dbeta, hipStream_t s);
hipError_t launch_gn_apply(const float* x, int64_t ldx, const double* sums, const float* gamma, const float* beta, int M, int C,
                           int ngroups, int rows_per_scene, float eps, float* y, int64_t ldy, hipStream_t s);
hipError_t launch_gn_bwd(const float* x, int64_t ldx, const double* sums, const float* gamma, const float* beta, int M, int C,
                         int ngroups, int rows_per_scene, float eps, const float* gy, int64_t ldgy, float* gz, double* bsums,
                         float* gx, int64_t ldgx, float* dgamma, float* dbeta, hipStream_t s);
hipError_t launch_decode_bwd(const float* g_logits, const float* g_center, const float* g_size, const float* g_rot,
                             const float* center, const float* size, const float* ref, ScaleBox sb, int M, int ncls, int NH1, int C,
                             float* g_h3, float* g_h1, float* g_ref, hipStream_t s);
hipError_t launch_head3_bwd(const float* g_h3, const float* w3, float* g_act, int M, int C, hipStream_t s);
hipError_t launch_posemb_bwd(const float* g_emb, const float* ref, const float* dim_t, int M, float* g_ref, hipStream_t s);
hipError_t launch_refpoint_bwd(const float* g_ref, const float* ref0, int B, int Q, float* g_w, hipStream_t s);
hipError_t launch_sample_bwd(const float* tokens, const double* T_cl, const float* cam, const float* ref, ScaleBox sb, int B, int V,
                             int h, int w, int C, int Q, const float* g_tgt, float* g_tokens, float* g_ref, hipStream_t s);
hipError_t launch_attn_bwd(const float* q, int64_t q_batch, int64_t q_head, int64_t q_row, const float* k, int64_t k_batch,
                           int64_t k_head, int64_t k_row, const float* v, int64_t v_batch, int64_t v_head, int64_t v_row,
                           const float* dO, int64_t do_batch, int64_t do_head, int64_t do_row, const float* lse, const float* D,
                           float* gq, int64_t gq_batch, int64_t gq_head, int64_t gq_row, float* gk, int64_t gk_batch,
                           int64_t gk_head, int64_t gk_row, float* gv, int64_t gv_batch, int64_t gv_head, int64_t gv_row, int B, int H,
                           int Lq, int Lk, int dh, int accumulate_kv, hipStream_t s, float* gq_part = nullptr, float drop_p = 0.f,
                           uint32_t drop_seed = 0, unsigned int* absmax = nullptr,    // absmax: 8-byte device scratch -> split-precision kernel
                           float* mat_scratch = nullptr);   // attn_bwd_mat_scratch_floats(...) floats: head dims other than 32 / 64
size_t attn_bwd_mat_scratch_floats(int Lq, int Lk, int dh);
// all recurrent iterations that share K / V in one launch (split-precision kernel; see attn_bwd.hip)
hipError_t launch_attn_bwd_batched(const float* q, const int64_t* q_off, int64_t q_batch, int64_t q_head, int64_t q_row, const float* k,
                                   int64_t k_batch, int64_t k_head, int64_t k_row, const float* v, int64_t v_batch, int64_t v_head,
                                   int64_t v_row, const float* dO, int64_t do_it, int64_t do_batch, int64_t do_head, int64_t do_row,
                                   const float* lse, const int64_t* lse_off, const float* D, int64_t D_it, float* gq, int64_t gq_it,
                                   int64_t gq_batch, int64_t gq_head, int64_t gq_row, float* gk, int64_t gk_batch, int64_t gk_head,
                                   int64_t gk_row, float* gv, int64_t gv_batch, int64_t gv_head, int64_t gv_row, int B, int H, int Lq,
                                   int Lk, int dh, int n_it, hipStream_t s, float* gq_part, float drop_p, const uint32_t* seeds,
                                   unsigned int* absmax, unsigned int* kv_absmax = nullptr,    // kv_absmax: out, max |dK|, |dV| (float bits)
                                   void* pack = nullptr,
                                   float* mat_scratch = nullptr,
                                   const void* kvcache = nullptr, int cache_terms = 0, int cache_kind = 0);   // head dim 64: K / V from the 16-bit cache
                                                                   // pack: attn_bwd_pack_floats(...) floats of scratch -> second-version kernel;
                                                                   // mat_scratch (dh == 256): attn_bwd_batched256_scratch_floats(n_it, Lq, Lk) floats
size_t attn_bwd_pack_floats(int B, int H, int Lq, int n_it);
// kvproj_bwd.hip: dW_kv / db_kv of the hoisted projection on the fp16 matrix pipe (hi/lo split), C = 256
bool kvproj_bwd_split_supported(int C);
hipError_t launch_tn_split_512x256(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t M, float* out, int64_t ldo,
                                   float* db, const unsigned int* absmax_bits, float* scale_scratch, hipStream_t s);
hipError_t launch_kvproj_bwd_split(const float* g, const float* tokens, int64_t M, int C, float* dW, float* db,
                                   const unsigned int* absmax_bits, float* scale_scratch, hipStream_t s);
// dst = dropout(src): keep mask of stream `seed` over the (M, N) index space, scaled by 1 / (1 - p)
hipError_t launch_dropout_apply(const float* src, float* dst, int M, int N, float p, uint32_t seed, hipStream_t s);
size_t attn_bwd_dq_partial_floats(int B, int H, int Lq, int Lk, int dh, bool grouped = false);      // grouped: the batched kernel's slots
int attn_bwd_kb_group(int B, int H, int Lk);
size_t attn_bwd_batched256_scratch_floats(int n_it, int Lq, int Lk);
hipError_t launch_absmax(const float* x, int64_t n, unsigned int* out, hipStream_t s);
hipError_t launch_attn_bwd_rowdot(const float* dO, const float* O, int64_t batch, int64_t row, int B, int H, int Lq, int dh, float* D,
                                  hipStream_t s);
// postproc.hip: parse_pred + 3-D NMS on the device
hipError_t launch_parse_pred(const float* center, const float* size, const float* rot6, const float* prob, int B, int Q, int ncls,
                             int num_semcls, const float* track_scale6, int for_vis, int enable_nms, float* obbs,
                             unsigned char* mask, hipStream_t s);
hipError_t launch_gemm_split(const float* X, int64_t ldx, const void* Whi, const void* Wlo, const float* bias, float* Y,
                             int64_t ldy, int M, int N, int K, int relu, const float* feat, int hw, hipStream_t s,
                             const float* scale_dev = nullptr, float scale_mul = 1.f, const float* xscale_dev = nullptr,
                             int accumulate = 0);   // Y (+)= ((x * *xscale_dev) W^T) * scale_mul / *scale_dev + bias

// ------------------------------------------------------------------ elementwise / gather kernels
hipError_t launch_camera_local(const float* T_cp, const float* T_wp, const float* T_wl, int B, int V,
                               float* T_cl, hipStream_t s);
hipError_t launch_camera_local_f64(const float* T_cp, const float* T_wp, const float* T_wl, int B, int V,
                                   double* T_cl, hipStream_t s);
hipError_t launch_initial_ref(const float* refpoint_w, int B, int Q, float* ref, hipStream_t s);
// Per-call pointers of a captured forward (api.hip parq_forward_replay): the recorded iterations must not hold the caller's token and
// output pointers themselves, so the prologue launch of every call (never part of the graph) leaves them in a block of the workspace —
// [0] tokens, [1..5] pred_logits, centre, size, ortho6d, sem_cls_prob, [6] coord_pos — and copies the cameras beside it; the kernels of
// the iterations that touch them (project + sample, box decode) load the pointers from that block when given one (`ind`).
struct CallPtrs { const void* p[8]; };
struct PrologueCall { const float* cam_src; float* cam_dst; int ncam; const void** ind; CallPtrs ptrs; };   // ind == nullptr: nothing to leave
hipError_t launch_forward_prologue(const float* T_cp, const float* T_wp, const float* T_wl, int B, int V, double* T_cl, const float* w, int Q,
                                   float* ref, const float* dim_t, float* emb, float* flags, int nflags, hipStream_t s,
                                   const PrologueCall* call = nullptr);
hipError_t launch_posemb(const float* ref, const float* dim_t, int M, float* emb, hipStream_t s);
hipError_t launch_zero_f64(double* p, int n, hipStream_t s);
hipError_t launch_project_sample(const float* tokens, const float* T_cl, const float* cam, const float* ref,
                                 ScaleBox sb, int B, int V, int h, int w, int C, int Q, float* tgt,
                                 float* coord_pos, hipStream_t s);
// zero_f64/zero_n: accumulators (GroupNorm moments) this kernel clears for later kernels of the iteration
// ind / coord_off: tokens = ind[0] and coord_pos = (float*)ind[6] + coord_off instead of the two pointer arguments (CallPtrs)
hipError_t launch_project_sample_f64(const float* tokens, const double* T_cl, const float* cam, const float* ref,
                                     ScaleBox sb, int B, int V, int h, int w, int C, int Q, float* tgt,
                                     float* coord_pos, double* zero_f64, int zero_n, hipStream_t s, float* raw_count = nullptr,
                                     const void* const* ind = nullptr, int64_t coord_off = 0);
hipError_t launch_pe1_sample(const LinearArgs& pe1, const float* tokens, const double* T_cl, const float* cam, const float* ref, ScaleBox sb,
                             int B, int V, int h, int w, int C, int Q, float* tgt, float* coord_pos, double* zero_f64, int zero_n,
                             float* raw_count, hipStream_t s, const void* const* ind = nullptr, int64_t coord_off = 0);
hipError_t launch_sample_finalize(const float* sums, const float* counts, int64_t M, int C, float* tgt, hipStream_t s,
                                  const float* range_sum = nullptr, int* range_flag = nullptr);
hipError_t launch_shard_range_flag(const int* range_flag, float* out, hipStream_t s);
hipError_t launch_attn_combine(const float* parts, int R, int64_t rec, int B, int H, int Q, int Lq_pad, int dh, float* out, hipStream_t s);
// self-attention of the Q queries in one launch (8 key slices per workgroup combined through LDS)
hipError_t launch_self_attn(const float* qkv, int64_t row_stride, int B, int H, int Lq, int dh, float* out,
                            int64_t out_row, hipStream_t s, float* lse = nullptr, float drop_p = 0.f, uint32_t drop_seed = 0);
hipError_t launch_layernorm(const float* X, const float* gamma, const float* beta, float* Y, int M, int C,
                            float eps, hipStream_t s);
// mean / rstd over (rows_per_scene x ncols) blocks: stats[(b * ngroups + g) * 2 + {0,1}]
hipError_t launch_gn_stats(const float* X, int64_t ldx, int col0, int ncols, int ngroups, int B,
                           int rows_per_scene, float eps, float* stats, hipStream_t s);
// Last kernel of an iteration, one wave per query row: GroupNorm+ReLU of the second hidden layer,
// the two output layers (centre 3, rotation 6), softmax / arg-max size gather / centre update
// (transformer_parq.py:242-279), next reference point and its 384-d sine embedding (:45-64).
struct BoxDecodeArgs {
    const float* h1; int64_t ld1;      // [M][..]: logits at cols [0,ncls), size_raw at [ncls, ncls+3)
    const float* h2; int64_t ld2;      // [M][2C]: centre hidden | rotation hidden (pre-GroupNorm)
    const double* gn_sums;             // [B][2][kGnSlots][2] moments of h2 per scene and head
    const float* gn_gamma; const float* gn_beta;       // [2][C]
    const float* w3; const float* b3;  // [2][6][C] (centre rows 0..2 of group 0), [2][6]
    int C; int rows_per_scene; float eps;
    const float* ref;                  // [M][3] normalised reference points of this iteration
    const float* mean_sizes; int n_mean;
    const float* dim_t;                // [128]
    ScaleBox sb; int M; int ncls;
    float *logits, *center, *size, *rot, *prob;     // outputs (coord_pos is written by project_sample)
    float* ref_next;                   // [M][3] or nullptr
    float* emb_next;                   // [M][384] or nullptr: pos2posemb3d(ref_next)
    int* poison_mirror;                // optional host-visible int: bit 0 set when outputs are poisoned by a range violation, bit 1 when
                                       // *peaky is set, bits 8 + h: head h of *peaky
    const int* peaky;                  // optional device int raised by the cross-attention merge of attention mode 4 (FlashArgs::peaky):
                                       // bit h = head h ran the mode-4 kernel on a row that rests on too few keys
    int peaky_poison;                  // non-zero: such an iteration's outputs are written as NaN too (never plausible wrong numbers)
    const int* poison;                 // optional device int: non-zero (fp16 operand range exceeded while the K/V cache was built; bit 2:
                                       // an in-launch hand-off timed out) -> every output of the iteration is written as NaN instead of
                                       // a plausible wrong number
    int poison_mask;                   // the bits of *poison that count (all of them in the fp16-operand cache modes, else only bit 2)
    const void* const* ind;            // optional CallPtrs block: the five output pointers are ind[1..5] + out_row0 rows (a captured forward)
    int64_t out_row0;
};
hipError_t launch_box_decode(const BoxDecodeArgs& a, hipStream_t s);
// weight packing helpers
hipError_t launch_copy_rows(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int rows, int cols,
                            hipStream_t s);
hipError_t launch_fill(float* dst, float value, int64_t n, hipStream_t s);

}  // namespace parq
