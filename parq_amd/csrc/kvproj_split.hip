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
#include <type_traits>

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
    // TERMS == 11 (attention mode 4 with per-head tiers): bit h set = head h is written in the split layout (fp16 x 3 kernel), clear =
    // as mode-4 stages; every (scene, head) region then spans head_bytes (the split layout's size)
    unsigned safe_mask; int64_t head_bytes;
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
// W-stationary persistent kernel (the one the decoder uses for C = 128 / 256).
//
// Measured on the tiled kernel above: 2.3 GB of global->CU traffic per scene (the 128-column W tile is re-fetched for every row
// tile: 1.5 GB from L2; tokens re-read once per column tile) — the load path, not HBM or MFMA, sets its time.  Here a workgroup
// of 8 waves OWNS a 256-column slice of W_kv (all K heads or all V heads at C = 256): each wave keeps the hi/lo fragments of 32
// columns x all K in registers (128 VGPRs) for the whole launch and the workgroup walks 64-token tiles of the scene batch,
// streaming only tokens; B operands never touch LDS.  The token stream is LDS-DMA with a software-pipelined k-step:
//
//   raw[D][TM][64] fp32   filled by global_load_lds (no staging registers), D - 1 k-steps requested ahead
//   hl[2][hi | lo][TM][64] fp16   the split image the MFMA fragments are read from
//
//   iteration q:   s_waitcnt vmcnt(N)      this thread's DMA pieces of k-step q + 1 have landed (N counted by hand, see below)
//                  barrier                 -> everybody's pieces of q + 1 landed; hl[q & 1] (converted in iteration q - 1) is complete;
//                                             hl[(q + 1) & 1] and raw[(q - 1) % D] are no longer read by anyone
//                  DMA of k-step q + D - 1 -> raw[(q - 1) % D]
//                  convert k-step q + 1: raw -> hl[(q + 1) & 1]        } independent instruction streams of one wave:
//                  24 MFMAs of k-step q from hl[q & 1]                 } hipcc interleaves the conversion with the MFMAs
//   ONE barrier per k-step, and the wave that converts is never the wave that waits for the matrix pipe.
//
// vmcnt is one in-order counter for loads AND stores.  The wait for k-step q + 1 (issued in iteration q - (D - 2)) may leave
// outstanding: the DMAs of the D - 3 iterations in between, plus the stores of a tile epilogue if one lies in between (it does
// when ks(q) < D - 2, except in the workgroup's first tile).  A tile with blocks past the scene skips stores, so its epilogue ends
// with vmcnt(0) and the counts that follow are merely conservative.  Rows past the scene are read clamped and zeroed at conversion.
// PROBE (development, results wrong by construction; tools/kvproj_probe.sh): bit 0 no MFMAs, 1 no global stores, 2 no conversion
// (MFMAs on whatever LDS holds), 3 no token DMA, 4 no epilogue at all
// Development build with -DPARQ_KV_STAMPS on top (PARQ_DEV_EXTRA_FLAGS=-DPARQ_KV_STAMPS python -c "import __graft_entry__ as g;
// g.build_dev(force=True)"): cycle-counter stamps of waves 0 and 4 of the first four workgroups over the first 96 k-steps, kept in LDS during
// the launch and dumped at its end (tools/r06_kvproj_stamps.py -> profiles/r06_kvproj_stamps.txt).  Point 0 = top of a k-step, 1 = this
// wave's DMA pieces and LDS writes have landed, 2 = behind the barrier, 3 / 4 = around the epilogue, 5 - 7 inside it.  The stamp code
// perturbs the loop it measures (the plain loop loses ~15 % with it, restructured variants less): use it for WHERE time goes, never for A/B.
#if defined(PARQ_DEV_PROBES) && defined(PARQ_KV_STAMPS)
constexpr int kKvStampSteps = 96, kKvStampPts = 8, kKvStampBytes = 2 * kKvStampSteps * kKvStampPts * 8;      // LDS: [wave 0 | wave 4][k-step][point]
#define PARQ_KV_STAMP(pt)                                                                                                        \
    do {                                                                                                                         \
        if (kv_tl && (tid & 255) == 0 && step < kKvStampSteps)                                                                   \
            kv_tl[((tid >> 8) * kKvStampSteps + step) * kKvStampPts + (pt)] = (unsigned long long)__builtin_readcyclecounter();  \
    } while (0)
#else
#define PARQ_KV_STAMP(pt) do { } while (0)
#endif

template <int TM, int TERMS, int KIND, int NK, int D, int PROBE = 0>
__global__ __launch_bounds__(512, 1) void kvproj_dma_kernel(KvProjArgs a, int total_rt, int nrt, int P) {
    PARQ_TL_KERNEL(kTlKvProj);
    constexpr int NWV = 8;
    constexpr bool SPLIT = TERMS != 1;         // three-term products (TERMS = 3, and 8: the same GEMM with the mode-4 epilogue;
    constexpr bool MIX = TERMS == 11;          // 11: each head in the layout of ITS tier, KvProjArgs::safe_mask)
    static_assert((TERMS != 8 && !MIX) || TM == 64, "a tile is one 64-key stage of the mode-4 cache");
    constexpr int kBlkH = SPLIT ? 8192 : 4096;      // 16-bit units per 32-key cache block
    constexpr int kVoff = SPLIT ? 4096 : 2048;      // V_hi offset inside a block
    constexpr int kThr = NWV * 64;
    constexpr int kCols = NWV * 32;
    constexpr int RT = TM / 32;                  // 32-row blocks per tile (accumulators per wave)
    constexpr int C = NK * kBK;
    constexpr int kRawBytes = TM * kBK * 4;      // one k-step of fp32 tokens
    constexpr int NDMA = kRawBytes / (kThr * 16);            // DMA instructions per thread and k-step
    constexpr int NST_K = RT * 2 * (SPLIT ? 2 : 1);     // store instructions per thread and tile (mode 4: K waves 2 + 1 + 1 per block,
                                                        // V waves 2: the stage cache holds V as one fp16 plane — NST_V inside run())
    constexpr int NI = TM * 8 / kThr;            // 8-float pieces of a k-step per thread (conversion)
    static_assert(D >= 3 && D - 2 <= NK && NDMA >= 1 && NI >= 1, "wait counts below assume at most one epilogue inside the prefetch window");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    float* raw = reinterpret_cast<float*>(ldsb);
    _Float16* hl = reinterpret_cast<_Float16*>(ldsb + D * kRawBytes);
    _Float16* stg = hl + 2 * 2 * TM * kBK;       // epilogue strips: 2 KB per wave
#if defined(PARQ_DEV_PROBES) && defined(PARQ_KV_STAMPS)
    unsigned long long* kv_tl = (parq_tl_buf && blockIdx.x < 4) ? reinterpret_cast<unsigned long long*>(stg + 8 * 1024) : nullptr;
    if (kv_tl) { for (int i = threadIdx.x; i < kKvStampBytes / 8; i += blockDim.x) kv_tl[i] = 0ull; }
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    constexpr int nslice = 2 * C / kCols;
    int p, slice;
    {
        const int w = blockIdx.x;
        constexpr int per = 8 * nslice;
        const int grp = w / per, r = w - grp * per;
        p = grp * 8 + (r & 7);
        slice = r >> 3;
    }
    if (p >= P) return;
    const int n0 = slice * kCols;
    const int headcol = (n0 + wave * 32) >> 6;    // scalar: the two waves ct = 0, 1 of a pair share a head
    const bool isK = headcol < a.H;
    const int ct = (wave & 1);
    // Which 32 of the head's 64 dims a wave owns is chosen so that its outputs form whole 64-byte pieces of the cache image
    // (see the epilogue): V waves own d = 32 ct + li (whole 64-byte V rows); K waves own the dims whose cache chunk has kh' = ct,
    // accumulator row i <-> d = 32 (i >> 4) + 16 ((i >> 2) & 1) + 8 ((i >> 3) & 1) + 4 ct + (i & 3), so that lane (key, kh) holds in
    // register group m the chunk s2 = 2 m + kh of its key's row: an aligned 64-byte half row per key and wave.
    const int dsel = isK ? 32 * (li >> 4) + 16 * ((li >> 2) & 1) + 8 * ((li >> 3) & 1) + 4 * ct + (li & 3) : 32 * ct + li;
    const int col = headcol * 64 + dsel;          // this lane's row of W_kv

    const float* bias = a.bias + headcol * 64;
    const int h = isK ? headcol : headcol - a.H;
    const int nblk = (a.N + 31) / 32;
    const int my_tiles = (total_rt - p + P - 1) / P;          // >= 1: P <= total_rt
    const int last_step = my_tiles * NK - 1;

    // k-step q of this workgroup -> (tile, ks); past the end: the last step again (requested, converted, never multiplied)
    auto dma = [&](int q) {
        const int qq = q < last_step ? q : last_step;
        const int tile = p + (qq / NK) * P, ks = qq % NK;
        const int b = tile / nrt, m0 = (tile - b * nrt) * TM;
        const float* Xb = a.X + ((int64_t)b * a.N) * C + ks * kBK;
        lds_byte* dst = (lds_byte*)(ldsb) + (q % D) * kRawBytes;
        if constexpr (PROBE & 8) return;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int row = (wave * NDMA + j) * 4 + (lane >> 4);              // one instruction = 4 rows x 256 B = 1 KB of LDS
            const int tok = m0 + row < a.N ? m0 + row : a.N - 1;
            const char* src = reinterpret_cast<const char*>(Xb + (int64_t)tok * C) + (lane & 15) * 16;
            __builtin_amdgcn_global_load_lds(src, dst + (wave * NDMA + j) * 1024, 16, 0, 0);
        }
    };
    bool ovf = false;
    // conversion of k-step q (tile origin m0): raw[q % D] -> hl[q & 1]
    auto convert = [&](int q, int m0) {
        if constexpr (PROBE & 4) return;
        float* src = raw + (q % D) * (TM * kBK);
        _Float16* Ahi = hl + (q & 1) * (2 * TM * kBK);
        _Float16* Alo = Ahi + TM * kBK;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int id = tid + i * kThr;
            const int row = id >> 3, c = id & 7;
            const int pos = c ^ ((row >> 1) & 7);
            const bool ok = m0 + row < a.N;         // rows past the scene: zeros
            // read through inline asm: a C++ read of the DMA ring makes hipcc wait for EVERY outstanding global_load_lds first
            // (s_waitcnt vmcnt(0): it cannot tell the ring slots apart), which is exactly the prefetch distance this kernel is about
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            f32x4v v0, v1;
            // a thread's piece is two 16-byte chunks of one 256-byte row; odd rows fetch theirs in the opposite order, so that the
            // 16 lanes an LDS cycle serves ({0-3, 12-15, 20-27}: rows r, r+1, r+2, r+3) hit 16 different chunk positions instead of
            // 8 positions twice (rows are exactly one bank period apart: 3.1e6 conflict cycles per launch before)
            const unsigned odd = (unsigned)(row & 1) * 16u;
            const unsigned addr = (unsigned)(size_t)(lds_byte*)(src + row * kBK + c * 8);
            const unsigned addrA = addr + odd, addrB = addr + 16u - odd;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(v0), "=&v"(v1) : "v"(addrA), "v"(addrB) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1)::"memory");
            const f32x4v lo4 = odd ? v1 : v0, hi4 = odd ? v0 : v1;
            float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = ok ? x[e] : 0.f;
            if constexpr (KIND == kF16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ovf |= !(fabsf(x[e]) < 60000.f);
            }
            if constexpr (SPLIT) {
                half8 hi, lo;
                split8(x, hi, lo);
                *reinterpret_cast<half8*>(Ahi + row * kBK + pos * 8) = hi;
                *reinterpret_cast<half8*>(Alo + row * kBK + pos * 8) = lo;
            } else {
                *reinterpret_cast<half8*>(Ahi + row * kBK + pos * 8) = cvt8_rn<KIND>(x);
            }
        }
    };
    auto tile_m0 = [&](int tile) {
        const int t = tile < total_rt ? tile : total_rt - 1;
        const int b = t / nrt;
        return (t - b * nrt) * TM;
    };

    auto run = [&](auto isk_tag, auto f8_tag) __attribute__((always_inline)) {
        constexpr bool ISK = decltype(isk_tag)::value;
        constexpr bool F8 = decltype(f8_tag)::value;             // this wave's head is written as mode-4 stages
        constexpr int NST_V = F8 ? RT * 2 : NST_K;
        // W fragments of this lane's column, resident for the whole launch: [k-step][s][hi, lo]
        half8 wfr[NK][4][2];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int64_t off = (int64_t)col * C + ks * kBK + 32 * kh + 8 * s2;
                wfr[ks][s2][0] = *reinterpret_cast<const half8*>(a.Whi + off);
                if constexpr (SPLIT) wfr[ks][s2][1] = *reinterpret_cast<const half8*>(a.Wlo + off);
            }
        float bK[ISK ? 16 : 1];
        if constexpr (ISK) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) bK[8 * m + e] = bias[16 * (2 * m + kh) + 4 * ct + (e & 3) + 8 * (e >> 2)];
        } else {
            bK[0] = bias[32 * ct + li];
        }
        // everything requested so far (W fragments, bias) has to be out of the counter before the counted waits start
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < D - 1; ++q) dma(q);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        convert(0, tile_m0(p));
        int step = 0;
        bool first = true;
        for (int tile = p; tile < total_rt; tile += P) {
            const int b = tile / nrt, m0 = (tile - b * nrt) * TM;
            const int m0_next = tile_m0(tile + P);
            f32x16 acc[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                PARQ_KV_STAMP(0);
                if constexpr ((PROBE & (2 | 8 | 16 | 32 | 64 | 128)) != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // counts do not hold
                else if (ks < D - 2 && !first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 3) * NDMA + (ISK ? NST_K : NST_V)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 3) * NDMA) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's conversion writes of k-step q
                PARQ_KV_STAMP(1);
                __builtin_amdgcn_s_barrier();
                PARQ_KV_STAMP(2);
                dma(step + D - 1);
                __builtin_amdgcn_sched_barrier(0);
                convert(step + 1, ks + 1 < NK ? m0 : m0_next);
                const _Float16* Ahi = hl + (ks & 1) * (2 * TM * kBK);      // NK even: buffer parity of step = parity of ks
                const _Float16* Alo = Ahi + TM * kBK;
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    half8 xh[RT], xl[RT];
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        const int row = t * 32 + li;
                        const int posr = (4 * kh + s2) ^ ((row >> 1) & 7);
                        xh[t] = *reinterpret_cast<const half8*>(Ahi + row * kBK + posr * 8);
                        if constexpr (SPLIT) xl[t] = *reinterpret_cast<const half8*>(Alo + row * kBK + posr * 8);
                    }
                    const half8 wh = wfr[ks][s2][0], wlo = wfr[ks][s2][1];
                    if constexpr (PROBE & 1) {
#pragma unroll
                        for (int t = 0; t < RT; ++t) acc[t][0] += (float)xh[t][0] * (float)wh[0] + (SPLIT ? (float)xl[t][1] * (float)wlo[1] : 0.f);
                    } else if constexpr (ISK) {          // transposed product: rows = d, cols = tokens
#pragma unroll
                        for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(wh, xh[t], acc[t]);
                        if constexpr (SPLIT) {
#pragma unroll
                            for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(wh, xl[t], acc[t]);
#pragma unroll
                            for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(wlo, xh[t], acc[t]);
                        }
                    } else {            // rows = tokens, cols = d
#pragma unroll
                        for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(xh[t], wh, acc[t]);
                        if constexpr (SPLIT) {
#pragma unroll
                            for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(xh[t], wlo, acc[t]);
#pragma unroll
                            for (int t = 0; t < RT; ++t) acc[t] = mfma16<KIND>(xl[t], wh, acc[t]);
                        }
                    }
                }
                ++step;
            }
            // ---- epilogue of this tile: bias, split, and the wave's piece of the cache image through a wave-private LDS strip, so
            // that every store instruction writes whole 64-byte pieces (16-byte chunks at a 128-byte stride, as the accumulator
            // layout would give them, reach 3.4 TB/s on this traffic shape against 5.6 TB/s for whole pieces:
            // tools/bench_src/hbm_stream.hip).  Strip = [32 rows][4 chunks of 16 B] in image order: K rows are keys (the wave's 64-byte
            // half of each 128-byte row), V rows are the wave's 32 dims (whole 64-byte rows); one store instruction = 16 rows.
            const bool whole = m0 + TM <= a.N;            // scalar
            PARQ_KV_STAMP(3);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PROBE & 16) {
                if (acc[0][0] == 123.456f) a.cache[tid] = (_Float16)acc[RT - 1][3];
                first = false;
                continue;
            }
            _Float16* strip = stg + wave * 1024;
            const int swz = ISK ? (li >> 1) & 3 : (li >> 2) & 3;     // low bits of the image's chunk swizzle for this lane's row
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int blk = (m0 >> 5) + t;
                if (blk >= nblk) continue;                                   // scalar
                _Float16* out = a.cache + (((int64_t)b * a.H + h) * nblk + blk) * kBlkH;          // (= head_bytes apart when MIX)
                if constexpr (F8) {
                    // mode-4 stage image (flash_split8.hip): this block's fp16 plane at t * 4 KB of the K / V 16-bit region; K lanes also
                    // store their 16 accumulator registers as ONE 16-byte piece of the hi8 plane and one of the lo8 plane
                    unsigned char* stage = reinterpret_cast<unsigned char*>(a.cache) +
                                           (MIX ? ((int64_t)b * a.H + h) * a.head_bytes + (int64_t)(m0 >> 6) * kStage8Bytes
                                                : (((int64_t)b * a.H + h) * (nblk >> 1) + (m0 >> 6)) * kStage8Bytes);
                    out = reinterpret_cast<_Float16*>(stage + (ISK ? kS8Kh16 : kS8Vh16 - 2 * kVoff) + t * 4096);
                    if constexpr (ISK) {
                        float x16[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) x16[r] = acc[t][r] + bK[r];
#if defined(PARQ_DEV_PROBES) && defined(PARQ_KV_STAMPS)
                        if (t == 0) { asm volatile("" : "+v"(x16[0]), "+v"(x16[15])); PARQ_KV_STAMP(5); }
#endif
                        i32x4 hi8, lo8;
                        pieces_e4m3(x16, hi8, lo8);
                        const int piece = ((t * 2 + kh) * 2 + ct) * 32 + li;        // piece (c = kh, h = ct) of key li
                        if constexpr (PROBE & 32) {                                  // development: where does the read-back come from?
                            if (hi8[0] == 0x12345678 && lo8[1] == 0x1234567) stage[tid] = 1;
                        } else if constexpr (PROBE & 256) {                          // non-temporal: the written cache is not read again by this kernel
                            __builtin_nontemporal_store(hi8, reinterpret_cast<i32x4*>(stage + kS8K8hi + piece * 16));
                            __builtin_nontemporal_store(lo8, reinterpret_cast<i32x4*>(stage + kS8K8lo + piece * 16));
                        } else if constexpr (!(PROBE & 2)) {
                            *reinterpret_cast<i32x4*>(stage + kS8K8hi + piece * 16) = hi8;
                            *reinterpret_cast<i32x4*>(stage + kS8K8lo + piece * 16) = lo8;
                            if (t == 0) PARQ_KV_STAMP(6);
                        }
                    }
                }
                half8 hi[2], lo[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = acc[t][8 * m + e] + (ISK ? bK[ISK ? 8 * m + e : 0] : bK[0]);
                    if constexpr (F8 && !ISK) hi[m] = cvt8_rn<kF16>(x);          // mode 4: V as one fp16 value, round to nearest
                    else if constexpr (SPLIT) split8(x, hi[m], lo[m]);
                    else hi[m] = cvt8_rn<KIND>(x);
                    if constexpr (KIND == kF16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ovf |= !(fabsf(x[e]) < 60000.f);
                    }
                }
#pragma unroll
                for (int pl = 0; pl < (SPLIT && !F8 ? 2 : 1); ++pl) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        *reinterpret_cast<half8*>(strip + li * 32 + (((2 * m + kh) ^ swz) << 3)) = pl ? lo[m] : hi[m];
                    // same-wave LDS round trip: DS operations of one wave complete in order
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int row = (lane >> 2) + 16 * j, pc = lane & 3;
                        const half8 v = *reinterpret_cast<const half8*>(strip + row * 32 + pc * 8);
                        if constexpr ((PROBE & 2) || ((PROBE & 64) && ISK) || ((PROBE & 128) && !ISK)) {
                            if (v[0] == (_Float16)123.f) out[tid] = v[1];
                        } else if constexpr (ISK) {
                            const int half_row = ct ^ ((row >> 3) & 1);           // bit 2 of the row's swizzle (key >> 1) & 7
                            half8* dst = reinterpret_cast<half8*>(out + pl * 2048 + row * 64 + ((4 * half_row + pc) << 3));
                            if constexpr (PROBE & 256) __builtin_nontemporal_store(v, dst);
                            else *dst = v;
                        } else {
                            half8* dst = reinterpret_cast<half8*>(out + kVoff + pl * 2048 + (32 * ct + row) * 32 + pc * 8);
                            if constexpr (PROBE & 256) __builtin_nontemporal_store(v, dst);
                            else *dst = v;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (t == 0) PARQ_KV_STAMP(7);
            }
            if (!whole) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // fewer stores than counted: drain
            __builtin_amdgcn_sched_barrier(0);
            PARQ_KV_STAMP(4);
            first = false;
        }
    };
    if constexpr (MIX) {
        const bool safe = (a.safe_mask >> h) & 1u;                  // scalar: a wave works on one head
        if (isK) { if (safe) run(std::true_type{}, std::false_type{}); else run(std::true_type{}, std::true_type{}); }
        else { if (safe) run(std::false_type{}, std::false_type{}); else run(std::false_type{}, std::true_type{}); }
    } else {
        if (isK) run(std::true_type{}, std::integral_constant<bool, TERMS == 8>{});
        else run(std::false_type{}, std::integral_constant<bool, TERMS == 8>{});
    }
    if (ovf) atomicOr(a.overflow, 1);
#if defined(PARQ_DEV_PROBES) && defined(PARQ_KV_STAMPS)
    if (kv_tl) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * kKvStampSteps * kKvStampPts; i += blockDim.x) {
            const unsigned long long v = kv_tl[i];
            if (v == 0ull) continue;
            const unsigned long long slot = atomicAdd(parq_tl_buf, 1ull);
            if (slot < parq_tl_cap) {
                unsigned long long* r = parq_tl_buf + 4 + slot * 4;
                const int wv = i / (kKvStampSteps * kKvStampPts), rem = i % (kKvStampSteps * kKvStampPts);
                r[0] = 100ull | ((unsigned long long)(wv * 4) << 8) | ((unsigned long long)blockIdx.x << 16);
                r[1] = (unsigned long long)(rem / kKvStampPts) * 8 + (rem % kKvStampPts);
                r[2] = v; r[3] = 0ull;
            }
        }
    }
#endif
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

template <int TM, int TERMS, int KIND, int NK, int D, int PROBE = 0>
static hipError_t launch_dma_nk(const KvProjArgs& a, int B, hipStream_t s) {
    static DynLdsOnce once;
#if defined(PARQ_DEV_PROBES) && defined(PARQ_KV_STAMPS)
    const size_t lds = (size_t)D * TM * kBK * 4 + (size_t)2 * 2 * TM * kBK * sizeof(_Float16) + 8 * 2048 + kKvStampBytes;
#else
    const size_t lds = (size_t)D * TM * kBK * 4 + (size_t)2 * 2 * TM * kBK * sizeof(_Float16) + 8 * 2048;
#endif
    if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&kvproj_dma_kernel<TM, TERMS, KIND, NK, D, PROBE>), lds); e != hipSuccess) return e;
    const int nslice = 2 * a.C / 256, nrt = ceil_div(a.N, TM);
    const int total_rt = B * nrt;
    int P = device_num_cus() / nslice;
    if (P < 1) P = 1;
    if (P > total_rt) P = total_rt;
    dim3 grid(ceil_div(P, 8) * 8 * nslice, 1, 1);
    hipLaunchKernelGGL((kvproj_dma_kernel<TM, TERMS, KIND, NK, D, PROBE>), grid, dim3(512), lds, s, a, total_rt, nrt, P);
    return hipGetLastError();
}

template <int TERMS, int KIND, int D>
static hipError_t launch_dma(const KvProjArgs& a, int B, hipStream_t s) {
    if (a.C == 4 * kBK) return launch_dma_nk<64, TERMS, KIND, 4, D>(a, B, s);
    if constexpr (D <= 4) {
        if (a.C == 2 * kBK) return launch_dma_nk<64, TERMS, KIND, 2, D>(a, B, s);
    }
    return hipErrorInvalidValue;
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
                               int C, int H, void* cache, int* overflow, hipStream_t s, int terms, int kind, unsigned safe_mask) {
    if (C % kBK != 0 || C != H * 64 || (2 * C) % kBN != 0) return hipErrorInvalidValue;
    static DynLdsOnce once;
    const size_t ldsb = (size_t)(2 * kBM * kBK + 2 * kBN * kBK) * sizeof(_Float16);      // 64 KB
    if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&kvproj_split_kernel), ldsb); e != hipSuccess) return e;
    KvProjArgs a;
    a.X = tokens; a.Whi = reinterpret_cast<const _Float16*>(Whi); a.Wlo = reinterpret_cast<const _Float16*>(Wlo);
    a.bias = bias; a.cache = reinterpret_cast<_Float16*>(cache); a.overflow = overflow; a.N = N; a.C = C; a.H = H;
    a.safe_mask = safe_mask; a.head_bytes = (int64_t)ceil_div(N, 32) * 16384;
    const int nct = 2 * C / kBN, nrt = ceil_div(N, kBM);
    if (C <= 4 * kBK && C % (2 * kBK) == 0) {
        // W-stationary persistent kernel: one workgroup per CU, the column slices of one slot on one XCD
#ifdef PARQ_DEV_PROBES
        if (terms == 8 && N % 64 == 0 && C == 256) {          // development: the mode-4 epilogue with one kind of store removed (results wrong)
            static const int probe8 = [] { const char* e = dev_env("PARQ_KVPROJ_PROBE8"); return e ? atoi(e) : 0; }();
            switch (probe8) {
                case 1: return launch_dma_nk<64, 8, kF16, 4, 4, 1>(a, B, s);          // the ingredient bits of PARQ_KVPROJ_PROBE on the mode-4 kernel
                case 2: return launch_dma_nk<64, 8, kF16, 4, 4, 2>(a, B, s);
                case 4: return launch_dma_nk<64, 8, kF16, 4, 4, 4>(a, B, s);
                case 16: return launch_dma_nk<64, 8, kF16, 4, 4, 16>(a, B, s);
                case 17: return launch_dma_nk<64, 8, kF16, 4, 4, 17>(a, B, s);
                case 21: return launch_dma_nk<64, 8, kF16, 4, 4, 21>(a, B, s);
                case 29: return launch_dma_nk<64, 8, kF16, 4, 4, 29>(a, B, s);
                case 32: return launch_dma_nk<64, 8, kF16, 4, 4, 32>(a, B, s);
                case 64: return launch_dma_nk<64, 8, kF16, 4, 4, 64>(a, B, s);
                case 128: return launch_dma_nk<64, 8, kF16, 4, 4, 128>(a, B, s);
                case 224: return launch_dma_nk<64, 8, kF16, 4, 4, 224>(a, B, s);
                case 256: return launch_dma_nk<64, 8, kF16, 4, 4, 256>(a, B, s);
                default: break;
            }
        }
#endif
        if (terms == 8) return (N % 64 == 0 && C == 256) ? launch_dma_nk<64, 8, kF16, 4, 4>(a, B, s) : hipErrorInvalidValue;
        if (terms == 11) return (N % 64 == 0 && C == 256) ? launch_dma_nk<64, 11, kF16, 4, 4>(a, B, s) : hipErrorInvalidValue;
        if (terms != 3) return kind == kF16 ? launch_dma<1, kF16, 4>(a, B, s) : launch_dma<1, kBF16, 4>(a, B, s);
#ifdef PARQ_DEV_PROBES
        static const int probe = [] { const char* e = dev_env("PARQ_KVPROJ_PROBE"); return e ? atoi(e) : 0; }();      // development
        if (probe && C == 256) {
            switch (probe) {
                case 1: return launch_dma_nk<64, 3, kF16, 4, 4, 1>(a, B, s);
                case 2: return launch_dma_nk<64, 3, kF16, 4, 4, 2>(a, B, s);
                case 16: return launch_dma_nk<64, 3, kF16, 4, 4, 16>(a, B, s);
                case 17: return launch_dma_nk<64, 3, kF16, 4, 4, 17>(a, B, s);
                case 20: return launch_dma_nk<64, 3, kF16, 4, 4, 20>(a, B, s);
                case 21: return launch_dma_nk<64, 3, kF16, 4, 4, 21>(a, B, s);
                case 29: return launch_dma_nk<64, 3, kF16, 4, 4, 29>(a, B, s);
                default: break;
            }
        }
#endif
        static const int depth = [] { const char* e = dev_env("PARQ_KVPROJ_RING"); return e ? atoi(e) : 4; }();       // 5: one more k-step in flight (no gain measured)
        if (depth == 5 && C == 256) return launch_dma<3, kF16, 5>(a, B, s);
        return launch_dma<3, kF16, 4>(a, B, s);
    }
    if (terms != 3) return hipErrorInvalidValue;            // the single-term modes exist on the persistent kernel only
    dim3 grid(ceil_div(nrt, 8) * 8 * nct, B, 1);
    if (grid.y > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kvproj_split_kernel, grid, dim3(kThreads), ldsb, s, a);
    return hipGetLastError();
}

PARQ_TL_DEFINE_SETTER(tl_set_kvproj_split)

}  // namespace parq
