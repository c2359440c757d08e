// Hoisted K/V in-projection of the memory tokens, written straight into the split-fp16
// "fragment-ready" cache consumed by flash_split_pipe_kernel (layout: flash_split.hip header).
//
//   [K | V][b][n][:] = tokens[b][n][:] @ W_kv^T + b_kv        (transformer_parq.py:377-380, hoisted:
//                                                              SURVEY.md 0.7 — weights shared, memory constant)
//
// fp32-accurate on the fp16 matrix pipe: tokens are split hi/lo on the fly, W is pre-split at
// pack time, each product is hi*hi + hi*lo + lo*hi with fp32 accumulation.  The GEMM is
// HBM-bound in this form (reads N*C*4, writes 2*N*C*4 bytes per scene; 151 GFLOP of fp16 MFMA
// at cfg 3 is ~60 us of matrix time against ~110 us of HBM time).
//
// Tiling: workgroup = 4 waves = 128 tokens x 128 output columns (two heads of 64), wave (wr,wc)
// owns 64 tokens x one head.  K heads are computed TRANSPOSED (A = W rows, B = tokens) and V heads
// normally (A = tokens, B = W rows): in both cases a lane's 8 consecutive accumulator registers are
// exactly one 16-byte chunk of the cache layout, so the epilogue is bias + split + 16-byte stores.
#include "common.hpp"

#include <cstdlib>

namespace parq {

namespace {

constexpr int kBM = 128, kBN = 128, kBK = 64;
constexpr int kThreads = 256;
constexpr int kBlkHalfs = 8192;                 // one 32-key cache block (16 KB)

struct KvProjArgs {
    const float* X;            // tokens [B][N][C]
    const _Float16* Whi;       // [2C][C]
    const _Float16* Wlo;
    const float* bias;         // [2C]
    _Float16* cache;           // [B][H][nblk][16 KB]
    int* overflow;
    int N, C, H;
};

__global__ __launch_bounds__(kThreads) void kvproj_split_kernel(KvProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];      // A_hi | A_lo | W_hi | W_lo, each [128][64]
    _Float16* Ahi = lds;
    _Float16* Alo = lds + kBM * kBK;
    _Float16* Bhi = lds + 2 * kBM * kBK;
    _Float16* Blo = lds + 2 * kBM * kBK + kBN * kBK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    // Tile mapping: workgroup id w runs on XCD w % 8 (observed dispatch rule; used for speed only).  The
    // nct column tiles that share one 128-token A tile get ids that are equal mod 8 and at most 8*nct apart,
    // so the A tile is fetched from HBM once and re-read from that XCD's L2.
    const int nct = 2 * a.C / kBN;                 // column tiles
    const int nrt = (a.N + kBM - 1) / kBM;         // row tiles per scene
    const int b = blockIdx.y;
    int rtile, ctile;
    {
        const int w = blockIdx.x;
        const int per_group = 8 * nct;             // 8 row tiles x nct column tiles
        const int grp = w / per_group;
        const int r = w - grp * per_group;
        rtile = grp * 8 + (r & 7);
        ctile = r >> 3;
    }
    if (rtile >= nrt) return;                      // padding of the last group (uniform per workgroup)
    const int m0 = rtile * kBM;                    // first token of the tile within scene b
    const int n0 = ctile * kBN;                    // first output column
    const int C = a.C;
    const int nk = C / kBK;
    const int headcol = (n0 >> 6) + wc;            // head index in [K heads | V heads]
    const bool isK = headcol < a.H;

    bool ovf = false;            // fp16 operand range (tokens and K / V values)
    // ---- staging assignment: 1024 A chunks (row, c) and 2048 W chunks per stage
    float4 areg[8];
    uint4 wreg[8];
    const float* Xb = a.X + ((int64_t)b * a.N) * C;
    auto gload = [&](int ks) {
        const int k0 = ks * kBK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * kThreads;      // 0..1023
            const int row = id >> 3, c = id & 7;
            const int tok = m0 + row;
            if (tok < a.N) {
                const float4* p = reinterpret_cast<const float4*>(Xb + (int64_t)tok * C + k0 + c * 8);
                areg[2 * i] = p[0];
                areg[2 * i + 1] = p[1];
            } else {
                areg[2 * i] = float4{0.f, 0.f, 0.f, 0.f};
                areg[2 * i + 1] = float4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * kThreads;
            const int row = id >> 3, c = id & 7;
            const int64_t off = (int64_t)(n0 + row) * C + k0 + c * 8;
            wreg[2 * i] = *reinterpret_cast<const uint4*>(a.Whi + off);
            wreg[2 * i + 1] = *reinterpret_cast<const uint4*>(a.Wlo + off);
        }
    };
    // LDS image: row-major [row][8 chunks]; logical chunk c = 4*kh + s lives at position c ^ ((row>>1)&7).
    // A global chunk g (8 consecutive k) is logical chunk g directly: the k -> (kh, s, e) assignment is
    // k = 32*kh + 8*s + e, the same for A and B, so the contraction is consistent.
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * kThreads;
            const int row = id >> 3, c = id & 7;
            const int pos = c ^ ((row >> 1) & 7);
            float x[8] = {areg[2 * i].x, areg[2 * i].y, areg[2 * i].z, areg[2 * i].w,
                          areg[2 * i + 1].x, areg[2 * i + 1].y, areg[2 * i + 1].z, areg[2 * i + 1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) ovf |= !(fabsf(x[e]) < 60000.f);          // the token operand is carried in fp16 too
            half8 hi, lo;
            split8(x, hi, lo);
            *reinterpret_cast<half8*>(Ahi + row * kBK + pos * 8) = hi;
            *reinterpret_cast<half8*>(Alo + row * kBK + pos * 8) = lo;
            *reinterpret_cast<uint4*>(Bhi + row * kBK + pos * 8) = wreg[2 * i];
            *reinterpret_cast<uint4*>(Blo + row * kBK + pos * 8) = wreg[2 * i + 1];
        }
    };

    f32x16 acc[2][2];            // [token tile rt][d tile ct]
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
            half8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = wr * 64 + t * 32 + li;
                const int posr = (4 * kh + s) ^ ((row >> 1) & 7);
                xh[t] = *reinterpret_cast<const half8*>(Ahi + row * kBK + posr * 8);
                xl[t] = *reinterpret_cast<const half8*>(Alo + row * kBK + posr * 8);
                const int col = wc * 64 + t * 32 + li;
                const int posc = (4 * kh + s) ^ ((col >> 1) & 7);
                wh[t] = *reinterpret_cast<const half8*>(Bhi + col * kBK + posc * 8);
                wl[t] = *reinterpret_cast<const half8*>(Blo + col * kBK + posc * 8);
            }
            // three passes over the four accumulators: consecutive MFMAs never share an accumulator
            if (isK) {      // transposed product: rows = d, cols = tokens
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ct], xh[rt], acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ct], xl[rt], acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ct], xh[rt], acc[rt][ct], 0, 0, 0);
            } else {        // rows = tokens, cols = d
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[rt], wh[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[rt], wl[ct], acc[rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[rt], wh[ct], acc[rt][ct], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) {
            swrite();
            __syncthreads();
        }
    }

    // ---- epilogue: bias, split, then through LDS so that every global store instruction writes 1 KB of
    // contiguous cache image (a wave owns, per 32-token block, the contiguous 8 KB [x_hi | x_lo] region of
    // its head: x = K or V).  The staging LDS is free after the last k-step (barrier above).
    const int nblk = (a.N + 31) / 32;
    const int h = isK ? headcol : headcol - a.H;
    const float* bias = a.bias + headcol * 64;
    _Float16* wl = lds + wave * (2 * 4096);                          // 2 blocks x 8 KB per wave = 8192 halfs
#pragma unroll
    for (int rt2 = 0; rt2 < 2; ++rt2) {
        _Float16* out = wl + rt2 * 4096;                             // [hi 2048 halfs | lo 2048 halfs]
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float x[8];
                if (isK) {
                    // lane = key (li), registers 8m..8m+7 = d = 32ct + 16m + 4kh + (e&3) + 8(e>>2): chunk s = 2ct + m
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[e] = acc[rt2][ct][8 * m + e] + bias[32 * ct + 16 * m + 4 * kh + (e & 3) + 8 * (e >> 2)];
                } else {
                    // lane = d (32ct + li), registers 8m..8m+7 = keys 16m + 4kh + (e&3) + 8(e>>2): chunk (m, kh)
                    const float bv = bias[32 * ct + li];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = acc[rt2][ct][8 * m + e] + bv;
                }
                half8 hi, lo;
                split8(x, hi, lo);
#pragma unroll
                for (int e = 0; e < 8; ++e) ovf |= !(fabsf(x[e]) < 60000.f);
                if (isK) {
                    const int c = 4 * kh + 2 * ct + m;
                    const int pos = c ^ ((li >> 1) & 7);
                    *reinterpret_cast<half8*>(out + li * 64 + pos * 8) = hi;
                    *reinterpret_cast<half8*>(out + 2048 + li * 64 + pos * 8) = lo;
                } else {
                    const int d = 32 * ct + li;
                    const int pos = (2 * m + kh) ^ ((d >> 2) & 3);
                    *reinterpret_cast<half8*>(out + d * 32 + pos * 8) = hi;
                    *reinterpret_cast<half8*>(out + 2048 + d * 32 + pos * 8) = lo;
                }
            }
        }
    }
    // same-wave LDS round trip: DS operations of one wave complete in order
    __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rt2 = 0; rt2 < 2; ++rt2) {
        const int blk = (m0 + wr * 64 + rt2 * 32) >> 5;
        if (blk >= nblk) continue;                                   // wave-uniform
        _Float16* gout = a.cache + (((int64_t)b * a.H + h) * nblk + blk) * kBlkHalfs + (isK ? 0 : 4096);
        const uint4* src = reinterpret_cast<const uint4*>(wl + rt2 * 4096);
        uint4* dst = reinterpret_cast<uint4*>(gout);
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i * 64 + lane] = src[i * 64 + lane];
    }
    if (ovf) atomicOr(a.overflow, 1);
}


// ------------------------------------------------------------------------------------------------
// W-stationary persistent variant (the one the decoder uses).
//
// Measured on the tiled kernel above: 2.3 GB of global->CU traffic per scene (the 128-column W tile is
// re-fetched for every row tile: 1.5 GB from L2; tokens re-read once per column tile) at ~21 GB/s per CU
// — the load path, not HBM or MFMA, sets its time.  Here a workgroup OWNS a 128-column slice of W_kv:
// each of its 4 waves keeps the hi/lo fragments of 32 columns x all K in registers (128 VGPRs) for the
// whole launch and the workgroup walks the row tiles of the scene batch, streaming only tokens
// (one 32 KB k-step in flight behind the MFMAs, across tile boundaries too).  LDS holds only the split
// token tile (double-buffered, one barrier per k-step); B operands never touch LDS.
// Measured and rejected on this kernel (MI355X, cfg 3, 205 us as is): 4 token k-steps in flight instead of 2 (no change:
// the loop is not latency-bound); removing the stores -> 179 us, the MFMAs -> 178 us, the loads -> 221 us (no single
// phase dominates); a software-pipelined k-step (fragments of step q read before the conversion + LDS write of step q+1,
// barrier at the end) -> 239 us; that plus the previous tile's stores spread between the next tile's MFMAs -> 336 us
// (64 more live registers, spills).
constexpr int kWsMaxKSteps = 4;          // K = C <= 256
constexpr int kWsDepth = 2;              // token k-steps in flight

// NWV waves own NWV*32 output columns; TM token rows per tile.  <4,128>: one wave per SIMD, 388 registers.
// <8,64>: two waves per SIMD (<= 256 registers each) so one wave's load/barrier stalls are covered by the
// other's MFMAs, and a token tile is re-read by 2 column slices instead of 4.
// TERMS = 3: split products, cache blocks [K_hi|K_lo|V_hi|V_lo]; TERMS = 1: single fp16 / bf16 (KIND) products,
// W given already converted in a.Whi, cache blocks [K|V] of 8 KB.
template <int NWV, int TM, int TERMS, int KIND>
__global__ __launch_bounds__(NWV * 64, 1) void kvproj_ws_kernel(KvProjArgs a, int total_rt, int nrt, int P) {
    constexpr int kBlkH = TERMS == 3 ? 8192 : 4096;      // 16-bit units per 32-key cache block
    constexpr int kVoff = TERMS == 3 ? 4096 : 2048;      // V_hi offset inside a block
    constexpr int kThr = NWV * 64;
    constexpr int kCols = NWV * 32;
    constexpr int RT = TM / 32;                  // 32-row blocks per tile (accumulators per wave)
    constexpr int NI = TM * 8 / kThr;            // 8-float pieces of a k-step tile per thread
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];      // [2 buffers][A_hi TMx64 | A_lo TMx64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int C = a.C;
    const int nk = C / kBK;
    const int nslice = 2 * C / kCols;
    // id -> (persistent slot p, column slice): the slices of one slot share p % 8, i.e. the XCD (L2 reuse of tokens)
    int p, slice;
    {
        const int w = blockIdx.x;
        const int per = 8 * nslice;
        const int grp = w / per, r = w - grp * per;
        p = grp * 8 + (r & 7);
        slice = r >> 3;
    }
    if (p >= P) return;
    const int n0 = slice * kCols;
    const int col = n0 + wave * 32 + li;          // this lane's output column (B operand row of W_kv)
    const int headcol = col >> 6;                 // wave-uniform: 32 columns never straddle a head
    const bool isK = headcol < a.H;
    const int ct = (wave & 1);                    // which 32-wide half of the head this wave owns

    // ---- W fragments, resident for the whole launch: [k-step][s][hi, lo]
    half8 wfr[kWsMaxKSteps][4][2];
#pragma unroll
    for (int ks = 0; ks < kWsMaxKSteps; ++ks)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            if (ks < nk) {
                const int64_t off = (int64_t)col * C + ks * kBK + 32 * kh + 8 * s2;
                wfr[ks][s2][0] = *reinterpret_cast<const half8*>(a.Whi + off);
                if constexpr (TERMS == 3) wfr[ks][s2][1] = *reinterpret_cast<const half8*>(a.Wlo + off);
            }
        }
    const float* bias = a.bias + headcol * 64;
    const int h = isK ? headcol : headcol - a.H;
    const int nblk = (a.N + 31) / 32;

    bool ovf = false;      // fp16 operand range: tokens (checked where they are converted) and K / V values (epilogue)
    // token staging registers, kWsDepth k-steps in flight (HBM latency under load is ~3 us, one k-step of MFMAs
    // ~0.75 us: with a single step in flight the loop runs at latency, not at MFMA or HBM speed)
    float4 areg[kWsDepth][2 * NI];
    auto gload = [&](int tile, int ks, float4 (&dst)[2 * NI]) {
        const int b = tile / nrt, m0 = (tile - b * nrt) * TM;
        const float* Xb = a.X + ((int64_t)b * a.N) * C + ks * kBK;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int id = tid + i * kThr;
            const int row = id >> 3, c = id & 7;
            const int tok = m0 + row;
            if (tok < a.N) {
                const float4* q = reinterpret_cast<const float4*>(Xb + (int64_t)tok * C + c * 8);
                dst[2 * i] = q[0];
                dst[2 * i + 1] = q[1];
            } else {
                dst[2 * i] = float4{0.f, 0.f, 0.f, 0.f};
                dst[2 * i + 1] = float4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto swrite = [&](int buf, const float4 (&src)[2 * NI]) {
        _Float16* Ahi = lds + buf * (2 * TM * kBK);
        _Float16* Alo = Ahi + TM * kBK;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int id = tid + i * kThr;
            const int row = id >> 3, c = id & 7;
            const int pos = c ^ ((row >> 1) & 7);
            float x[8] = {src[2 * i].x, src[2 * i].y, src[2 * i].z, src[2 * i].w,
                          src[2 * i + 1].x, src[2 * i + 1].y, src[2 * i + 1].z, src[2 * i + 1].w};
            if constexpr (KIND == kF16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ovf |= !(fabsf(x[e]) < 60000.f);
            }
            if constexpr (TERMS == 3) {
                half8 hi, lo;
                split8(x, hi, lo);
                *reinterpret_cast<half8*>(Ahi + row * kBK + pos * 8) = hi;
                *reinterpret_cast<half8*>(Alo + row * kBK + pos * 8) = lo;
            } else {
                *reinterpret_cast<half8*>(Ahi + row * kBK + pos * 8) = cvt8_rn<KIND>(x);
            }
        }
    };
    // linear k-step stream over this workgroup's tiles: step q -> (tile p + (q / nk) * P, ks = q % nk)
    const int my_tiles = p < total_rt ? (total_rt - p + P - 1) / P : 0;
    const int total_steps = my_tiles * nk;
    auto issue = [&](int q, float4 (&dst)[2 * NI]) {
        if (q < total_steps) gload(p + (q / nk) * P, q % nk, dst);
    };

    int step = 0;                                  // global k-step counter (LDS buffer parity, staging slot)
    static_assert(kWsDepth == 2 && kWsMaxKSteps % kWsDepth == 0, "slot arithmetic below assumes depth 2");
    issue(0, areg[0]);
    issue(1, areg[1]);
    for (int tile = p; tile < total_rt; tile += P) {
        f32x16 acc[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < kWsMaxKSteps; ++ks) {
            if (ks < nk) {
                const int buf = step & 1;
                // nk is even (C % 128 == 0 on this path), so the staging slot of k-step ks is ks & 1 at compile time
                swrite(buf, areg[ks & 1]);                     // tokens of this k-step (requested two steps ago)
                issue(step + kWsDepth, areg[ks & 1]);          // refill the slot: two k-steps ahead, across tiles
                __syncthreads();
                const _Float16* Ahi = lds + buf * (2 * TM * kBK);
                const _Float16* Alo = Ahi + TM * kBK;
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    half8 xh[RT], xl[RT];
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        const int row = t * 32 + li;
                        const int posr = (4 * kh + s2) ^ ((row >> 1) & 7);
                        xh[t] = *reinterpret_cast<const half8*>(Ahi + row * kBK + posr * 8);
                        if constexpr (TERMS == 3) xl[t] = *reinterpret_cast<const half8*>(Alo + row * kBK + posr * 8);
                    }
                    const half8 wh = wfr[ks][s2][0], wlo = wfr[ks][s2][1];
                    if (isK) {          // transposed product: rows = d, cols = tokens
#pragma unroll
                        for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(wh, xh[t], acc[t]);
                        if constexpr (TERMS == 3) {
#pragma unroll
                            for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(wh, xl[t], acc[t]);
#pragma unroll
                            for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(wlo, xh[t], acc[t]);
                        }
                    } else {            // rows = tokens, cols = d
#pragma unroll
                        for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(xh[t], wh, acc[t]);
                        if constexpr (TERMS == 3) {
#pragma unroll
                            for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(xh[t], wlo, acc[t]);
#pragma unroll
                            for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(xl[t], wh, acc[t]);
                        }
                    }
                }
                ++step;
            }
        }
        // ---- epilogue of this tile: bias, split, 16-byte chunks of the cache blocks (this wave owns one
        // 32-wide half `ct` of its head: chunks 2ct+m (K) / rows 32ct+li (V))
        const int b = tile / nrt, m0 = (tile - b * nrt) * TM;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int blk = (m0 >> 5) + t;
            if (blk >= nblk) continue;                                   // wave-uniform
            _Float16* out = a.cache + (((int64_t)b * a.H + h) * nblk + blk) * kBlkH;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float x[8];
                if (isK) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[e] = acc[t][8 * m + e] + bias[32 * ct + 16 * m + 4 * kh + (e & 3) + 8 * (e >> 2)];
                } else {
                    const float bv = bias[32 * ct + li];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = acc[t][8 * m + e] + bv;
                }
                half8 hi, lo;
                if constexpr (TERMS == 3) split8(x, hi, lo);
                else hi = cvt8_rn<KIND>(x);
                if constexpr (KIND == kF16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ovf |= !(fabsf(x[e]) < 60000.f);
                }
                if (isK) {
                    const int c = 4 * kh + 2 * ct + m;
                    const int pos = c ^ ((li >> 1) & 7);
                    *reinterpret_cast<half8*>(out + li * 64 + pos * 8) = hi;
                    if constexpr (TERMS == 3) *reinterpret_cast<half8*>(out + 2048 + li * 64 + pos * 8) = lo;
                } else {
                    const int d = 32 * ct + li;
                    const int pos = (2 * m + kh) ^ ((d >> 2) & 3);
                    *reinterpret_cast<half8*>(out + kVoff + d * 32 + pos * 8) = hi;
                    if constexpr (TERMS == 3) *reinterpret_cast<half8*>(out + 6144 + d * 32 + pos * 8) = lo;
                }
            }
        }
    }
    if (ovf) atomicOr(a.overflow, 1);
}

// fp32 [rows][cols] -> hi/lo fp16 (weight pre-split at pack time)
__global__ void split_f32_kernel(const float* __restrict__ src, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    half2v h, l;
    split_pair(src[i], 0.f, h, l);
    hi[i] = h[0];
    lo[i] = l[0];
}

}  // namespace

template <int KIND>
__global__ void cvt16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = src[i + e];
        *reinterpret_cast<half8*>(dst + i) = cvt8_rn<KIND>(x);
    } else {
        for (int64_t j = i; j < n; ++j) {
            float x[8] = {src[j], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            dst[j] = cvt8_rn<KIND>(x)[0];
        }
    }
}

template <int NWV, int TM, int TERMS, int KIND>
static hipError_t launch_ws(const KvProjArgs& a, int B, hipStream_t s) {
    static DynLdsOnce once;
    const size_t lds = (size_t)2 * 2 * TM * kBK * sizeof(_Float16);
    if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&kvproj_ws_kernel<NWV, TM, TERMS, KIND>), lds); e != hipSuccess) return e;
    const int nslice = 2 * a.C / (NWV * 32), nrt = ceil_div(a.N, TM);
    const int total_rt = B * nrt;
    int P = device_num_cus() / nslice;
    if (P < 1) P = 1;
    if (P > total_rt) P = total_rt;
    dim3 grid(ceil_div(P, 8) * 8 * nslice, 1, 1);
    hipLaunchKernelGGL((kvproj_ws_kernel<NWV, TM, TERMS, KIND>), grid, dim3(NWV * 64), lds, s, a, total_rt, nrt, P);
    return hipGetLastError();
}

hipError_t launch_cvt16(const float* src, void* dst, int64_t n, int kind, hipStream_t s) {
    const unsigned blocks = (unsigned)ceil_div64(ceil_div64(n, 8), 256);
    if (kind == kF16)
        hipLaunchKernelGGL(cvt16_kernel<kF16>, dim3(blocks), dim3(256), 0, s, src, reinterpret_cast<_Float16*>(dst), n);
    else
        hipLaunchKernelGGL(cvt16_kernel<kBF16>, dim3(blocks), dim3(256), 0, s, src, reinterpret_cast<_Float16*>(dst), n);
    return hipGetLastError();
}

hipError_t launch_split_f32(const float* src, void* hi, void* lo, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, src,
                       reinterpret_cast<_Float16*>(hi), reinterpret_cast<_Float16*>(lo), n);
    return hipGetLastError();
}

// tokens [B][N][C] -> split cache; Whi/Wlo [2C][C] fp16, bias [2C] fp32.  Needs C % 64 == 0, head dim 64.
hipError_t launch_kvproj_split(const float* tokens, const void* Whi, const void* Wlo, const float* bias, int B, int N,
                               int C, int H, void* cache, int* overflow, hipStream_t s, int terms, int kind) {
    if (C % kBK != 0 || C != H * 64 || (2 * C) % kBN != 0) return hipErrorInvalidValue;
    static DynLdsOnce once;
    const size_t ldsb = (size_t)(2 * kBM * kBK + 2 * kBN * kBK) * sizeof(_Float16);      // 64 KB
    if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&kvproj_split_kernel), ldsb); e != hipSuccess) return e;
    KvProjArgs a;
    a.X = tokens; a.Whi = reinterpret_cast<const _Float16*>(Whi); a.Wlo = reinterpret_cast<const _Float16*>(Wlo);
    a.bias = bias; a.cache = reinterpret_cast<_Float16*>(cache); a.overflow = overflow; a.N = N; a.C = C; a.H = H;
    const int nct = 2 * C / kBN, nrt = ceil_div(N, kBM);
    if (C <= kWsMaxKSteps * kBK && C % (2 * kBK) == 0) {
        // W-stationary persistent kernel: one workgroup per CU, column slices of one slot on one XCD
        static const int waves = [] {
            const char* e = getenv("PARQ_KVPROJ_WAVES");
            return e ? atoi(e) : 8;
        }();
        if (terms != 3) {
            if ((2 * C) % 256 != 0) return hipErrorInvalidValue;
            return kind == kF16 ? launch_ws<8, 64, 1, kF16>(a, B, s) : launch_ws<8, 64, 1, kBF16>(a, B, s);
        }
        if (waves == 8 && (2 * C) % 256 == 0) return launch_ws<8, 64, 3, kF16>(a, B, s);
        return launch_ws<4, 128, 3, kF16>(a, B, s);
    }
    if (terms != 3) return hipErrorInvalidValue;            // the single-term modes exist on the persistent kernel only
    dim3 grid(ceil_div(nrt, 8) * 8 * nct, B, 1);
    if (grid.y > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kvproj_split_kernel, grid, dim3(kThreads), ldsb, s, a);
    return hipGetLastError();
}

}  // namespace parq
