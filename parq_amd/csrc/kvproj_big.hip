// Hoisted K/V in-projection for LARGE model dims (C > 256: the reference's shipped DEC_DIM 1024), written into the split-fp16
// "fragment-ready" cache (layout: flash_split.hip; heads of 64 — a 256-dim head is 4 virtual heads, flash_split256.hip).
//
// The W-stationary kernel of kvproj_split.hip needs W_kv in registers (C <= 256) and its tiled fallback re-converts every token tile
// once per 128-column tile (16 times at C = 1024) between two barriers per k-step.  Here the tokens are split into fp16 hi/lo ONCE
// by a streaming pre-pass, and the GEMM is a pure fp16-pipe kernel: both operands travel global -> LDS by LDS-DMA (no staging
// registers, no conversion, no LDS stores in the loop), one barrier per k-step.
//
//   workgroup = 8 waves = 256 tokens x 256 output columns (4 virtual heads), k-steps of 32, LDS ring of 2 x 64 KB
//               ([A_hi | A_lo | W_hi | W_lo] x [256 rows][32 k]); wave (wr, wc) owns 64 tokens x 128 columns = 2 x 4 accumulators;
//   K heads are computed TRANSPOSED (A = W rows, B = tokens), V heads normally, so that 8 consecutive accumulator registers are one
//   16-byte chunk of the cache layout (as in kvproj_split.hip); the epilogue adds the bias, splits and sends every 8 KB [hi | lo]
//   image of a (32-token block, head) through LDS so that each global store instruction writes 1 KB of contiguous cache.
// At C = 256 the W-stationary kernel stays ahead (0.205 ms against 0.251 ms here: 8 k-steps per tile do not amortise the pre-pass
// and the epilogue; PARQ_KVPROJ_BIG_MINC=256 reproduces the comparison).
#include "common.hpp"
#include <cstdlib>

namespace parq {

namespace {

constexpr int kTM = 256, kTN = 256, kTK = 32;
constexpr int kImg = kTM * kTK;                     // halfs of one [256 rows][32 k] image (16 KB)
constexpr int kStage = 4 * kImg;                    // A_hi | A_lo | W_hi | W_lo
constexpr int kBlkHalfs = 8192;                     // one 32-key cache block of a head (16 KB)

struct BigArgs {
    const _Float16* Xhi; const _Float16* Xlo;       // [B*N][C] split tokens
    const _Float16* Whi; const _Float16* Wlo;       // [2C][C]
    const float* bias;                              // [2C]
    _Float16* cache;                                // [B][C/64 heads][nblk][16 KB]
    int* overflow;
    int N, C, VH;                                   // VH = C / 64 virtual heads
};

// 8 fp32 -> hi/lo fp16, streaming (the pre-pass)
__global__ __launch_bounds__(256) void split_tokens_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                           int64_t n8, int* __restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    bool ovf = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) ovf |= !(fabsf(v[e]) < 60000.f);
    if (ovf) atomicOr(overflow, 1);                               // rare: the token operand left the fp16 range
    half8 h, l;
    split8(v, h, l);
    reinterpret_cast<half8*>(hi)[i] = h;
    reinterpret_cast<half8*>(lo)[i] = l;
}

// chunk position of 8 consecutive k of row r inside an image row of 4 chunks (64 bytes): 4 rows share a 256-byte bank row, the
// 16 lanes one ds_read_b128 cycle serves (rows li in {0-3, 12-15, 20-27} or their complements) must hit 16 different slots
__device__ __forceinline__ int chunk_pos(int r, int c) { return c ^ ((r >> 2) & 3); }

// TERMS = 3: split products (tokens and W as hi/lo planes, cache blocks of 16 KB); TERMS = 1: single fp16 / bf16 (KIND) products —
// a.Xhi / a.Whi hold the rounded operands, the lo pointers are unused, cache blocks of 8 KB [K | V]
template <int TERMS, int KIND>
__global__ __launch_bounds__(512) void kvproj_big_kernel(BigArgs a) {
    constexpr int NIMG = TERMS == 3 ? 4 : 2;                    // operand images per k-step: A_hi | A_lo | W_hi | W_lo, or A | W
    constexpr int kStg = NIMG * kImg;
    constexpr int NDMA = NIMG * 2;                              // DMA instructions per thread and k-step
    constexpr int kBlkH = TERMS == 3 ? 8192 : 4096;             // halfs of one 32-key cache block of a head
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];          // [2][kStg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int C = a.C;
    const int nct = 2 * C / kTN;
    const int nrt = (a.N + kTM - 1) / kTM;
    const int b = blockIdx.y;
    int rtile, ctile;
    {   // the column tiles of one token tile get workgroup ids that are equal mod 8 (same XCD: tokens re-read from its L2)
        const int w = blockIdx.x;
        const int per = 8 * nct, grp = w / per, r = w - grp * per;
        rtile = grp * 8 + (r & 7);
        ctile = r >> 3;
    }
    if (rtile >= nrt) return;
    const int m0 = rtile * kTM, n0 = ctile * kTN;
    const int nk = C / kTK;

    // ---- LDS-DMA of one k-step: 8 instructions per thread, piece p = i * 512 + tid of the 4096 16-byte pieces of the stage
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const _Float16* srcs[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int p = i * 512 + tid;
        const int img = p >> 10, pp = p & 1023;                 // image index, piece inside it
        const int row = pp >> 2, slot = pp & 3;
        const int chunk = slot ^ ((row >> 2) & 3);
        const bool isA = TERMS == 3 ? img < 2 : img < 1;
        const bool lo = TERMS == 3 && (img & 1);
        if (isA) {
            int tok = m0 + row;
            tok = tok < a.N ? tok : a.N - 1;                    // rows past the scene: any finite data (their cache entries are masked keys)
            srcs[i] = (lo ? a.Xlo : a.Xhi) + ((int64_t)b * a.N + tok) * C + chunk * 8;
        } else {
            srcs[i] = (lo ? a.Wlo : a.Whi) + (int64_t)(n0 + row) * C + chunk * 8;
        }
    }
    // (asm-issued DMA + LDS-only barrier, common.hpp: with the builtin hipcc waited for the step it had just requested before every
    // fragment read of the current one)
    auto gload = [&](int ks, int slot) {
        const unsigned dst = (unsigned)(size_t)(lds_byte*)(lds + slot * kStg);
#pragma unroll
        for (int i = 0; i < NDMA; ++i) lds_dma16(srcs[i] + ks * kTK, dst + (i * 512 + wave * 64) * 16);
    };

    const int headcol0 = (n0 >> 6) + 2 * wc;                    // first of this wave's two virtual heads, in [K heads | V heads]
    const bool isK = headcol0 < a.VH;                           // 128 | C: both heads of a wave are on the same side

    f32x16 acc[2][4];                                           // [token block t][column block j]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    gload(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's pieces of step ks have landed
        lds_barrier();                                          // ... everyone's; every wave is past the reads of step ks - 1
        if (ks + 1 < nk) gload(ks + 1, (ks + 1) & 1);
        const _Float16* S = lds + (ks & 1) * kStg;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            half8 xh[2], xl[2], wh[4], wl[4];
            constexpr int kW = TERMS == 3 ? 2 * kImg : kImg;       // offset of the W image(s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int r = wr * 64 + t * 32 + li;
                const int off = r * kTK + chunk_pos(r, 2 * s + kh) * 8;
                xh[t] = *reinterpret_cast<const half8*>(S + off);
                if constexpr (TERMS == 3) xl[t] = *reinterpret_cast<const half8*>(S + kImg + off);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wc * 128 + j * 32 + li;
                const int off = r * kTK + chunk_pos(r, 2 * s + kh) * 8;
                wh[j] = *reinterpret_cast<const half8*>(S + kW + off);
                if constexpr (TERMS == 3) wl[j] = *reinterpret_cast<const half8*>(S + kW + kImg + off);
            }
            if (isK) {          // transposed product: rows = d, cols = tokens
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[t][j] = mfma16<KIND>(wh[j], xh[t], acc[t][j]);
                        if constexpr (TERMS == 3) {
                            acc[t][j] = mfma16<KIND>(wh[j], xl[t], acc[t][j]);
                            acc[t][j] = mfma16<KIND>(wl[j], xh[t], acc[t][j]);
                        }
                    }
            } else {            // rows = tokens, cols = d
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[t][j] = mfma16<KIND>(xh[t], wh[j], acc[t][j]);
                        if constexpr (TERMS == 3) {
                            acc[t][j] = mfma16<KIND>(xh[t], wl[j], acc[t][j]);
                            acc[t][j] = mfma16<KIND>(xl[t], wh[j], acc[t][j]);
                        }
                    }
            }
        }
    }

    // ---- epilogue: per (token block t, head hh) the 8 KB image [x_hi | x_lo] (x = K or V) goes through this wave's LDS scratch and
    // leaves as 8 store instructions of 1 KB each
    __syncthreads();                                            // the ring is free
    const int nblk = (a.N + 31) / 32;
    bool ovf = false;
    _Float16* wl = lds + wave * 4096;                           // 8 KB per wave
    // all bias values of this lane up front (64 for a K wave, 4 for a V wave): between the LDS round trips below each group's loads
    // would otherwise be issued and waited for one after the other (16 dependent L2 round trips per tile)
    float bK[2][2][2][8], bV[2][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float* bias = a.bias + (headcol0 + hh) * 64;
            bV[hh][ct] = 0.f;
            if (isK) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bK[hh][ct][m][e] = bias[32 * ct + 16 * m + 4 * kh + (e & 3) + 8 * (e >> 2)];
            } else {
                bV[hh][ct] = bias[32 * ct + li];
            }
        }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int headcol = headcol0 + hh;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f32x16& A = acc[t][2 * hh + ct];
                    float x[8];
                    if (isK) {      // lane = key li, registers 8m .. 8m+7 = d 32 ct + 16 m + 4 kh + (e & 3) + 8 (e >> 2)
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = A[8 * m + e] + bK[hh][ct][m][e];
                    } else {        // lane = d 32 ct + li, registers = keys 16 m + 4 kh + (e & 3) + 8 (e >> 2)
                        const float bv = bV[hh][ct];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = A[8 * m + e] + bv;
                    }
                    half8 hi, lo;
                    if constexpr (TERMS == 3) split8(x, hi, lo);
                    else hi = cvt8_rn<KIND>(x);
                    if constexpr (KIND == kF16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ovf |= !(fabsf(x[e]) < 60000.f);
                    }
                    if (isK) {
                        const int pos = (4 * kh + 2 * ct + m) ^ ((li >> 1) & 7);
                        *reinterpret_cast<half8*>(wl + li * 64 + pos * 8) = hi;
                        if constexpr (TERMS == 3) *reinterpret_cast<half8*>(wl + 2048 + li * 64 + pos * 8) = lo;
                    } else {
                        const int d = 32 * ct + li;
                        const int pos = (2 * m + kh) ^ ((d >> 2) & 3);
                        *reinterpret_cast<half8*>(wl + d * 32 + pos * 8) = hi;
                        if constexpr (TERMS == 3) *reinterpret_cast<half8*>(wl + 2048 + d * 32 + pos * 8) = lo;
                    }
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): same-wave LDS round trip
            __builtin_amdgcn_wave_barrier();
            const int blk = (m0 + wr * 64 + t * 32) >> 5;
            if (blk < nblk) {                                   // wave-uniform
                const int hv = isK ? headcol : headcol - a.VH;
                _Float16* gout = a.cache + (((int64_t)b * a.VH + hv) * nblk + blk) * kBlkH + (isK ? 0 : kBlkH / 2);
                const uint4* src = reinterpret_cast<const uint4*>(wl);
                uint4* dst = reinterpret_cast<uint4*>(gout);
#pragma unroll
                for (int i = 0; i < (TERMS == 3 ? 8 : 4); ++i) dst[i * 64 + lane] = src[i * 64 + lane];      // the 8 KB [hi | lo] (4 KB single) image
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    if (ovf) atomicOr(a.overflow, 1);
}

}  // namespace

// scratch floats for the split tokens of launch_kvproj_big (0: the kernel does not apply to this shape)
size_t kvproj_big_scratch_floats(int B, int N, int C) {
    static const int minc = [] { const char* e = dev_env("PARQ_KVPROJ_BIG_MINC"); return e ? atoi(e) : 257; }();     // experiment knob
    if (C < minc || C % 128 != 0) return 0;
    return (size_t)B * N * C;                                   // hi + lo fp16 = 4 bytes per element
}

template <int TERMS, int KIND>
static hipError_t launch_big_t(const BigArgs& a, int B, hipStream_t s) {
    static DynLdsOnce once;
    const size_t ldsb = (size_t)2 * (TERMS == 3 ? 4 : 2) * kImg * sizeof(_Float16);              // 128 KB (64 KB single-term); >= the 64 KB of epilogue scratch
    if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&kvproj_big_kernel<TERMS, KIND>), ldsb); e != hipSuccess) return e;
    const int nct = 2 * a.C / kTN, nrt = ceil_div(a.N, kTM);
    dim3 grid(ceil_div(nrt, 8) * 8 * nct, B, 1);
    hipLaunchKernelGGL((kvproj_big_kernel<TERMS, KIND>), grid, dim3(512), ldsb, s, a);
    return hipGetLastError();
}

// terms = 3: Whi / Wlo are the hi / lo planes of W_kv; terms = 1: Whi holds W_kv rounded to `kind`, Wlo is unused
hipError_t launch_kvproj_big(const float* tokens, const void* Whi, const void* Wlo, const float* bias, int B, int N, int C,
                             void* cache, int* overflow, float* scratch, hipStream_t s, int terms, int kind) {
    if (!kvproj_big_scratch_floats(B, N, C) || !scratch || B > 65535) return hipErrorInvalidValue;
    const int64_t n = (int64_t)B * N * C;
    _Float16* xhi = reinterpret_cast<_Float16*>(scratch);
    _Float16* xlo = xhi + n;
    BigArgs a;
    a.Xhi = xhi; a.Xlo = xlo; a.Whi = reinterpret_cast<const _Float16*>(Whi); a.Wlo = reinterpret_cast<const _Float16*>(Wlo);
    a.bias = bias; a.cache = reinterpret_cast<_Float16*>(cache); a.overflow = overflow; a.N = N; a.C = C; a.VH = C / 64;
    if (terms == 3) {
        hipLaunchKernelGGL(split_tokens_kernel, dim3((unsigned)ceil_div64(n / 8, 256)), dim3(256), 0, s, tokens, xhi, xlo, n / 8, overflow);
        return launch_big_t<3, kF16>(a, B, s);
    }
    if (hipError_t e = launch_cvt16(tokens, xhi, n, kind, s); e != hipSuccess) return e;      // (token range: checked on the K / V values in the epilogue)
    return kind == kF16 ? launch_big_t<1, kF16>(a, B, s) : launch_big_t<1, kBF16>(a, B, s);
}

}  // namespace parq
