// Split-precision flash cross-attention for gfx950 (head dim 64).
//
// fp32-accurate attention on the fp16 matrix pipe: every fp32 operand x is carried as
// x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits), and each product
// a*b is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with fp32 accumulation
// (v_mfma_f32_32x32x16_f16).  The dropped lo*lo term and the residual of the split are both
// ~2^-22 relative, i.e. fp32-rounding class: measured error against float64 equals the
// fp32-MFMA kernel's.  Three fp16 MFMAs replace sixteen fp32 ones for the same contraction,
// so the kernel moves from the fp32-MFMA roof (157 TF) towards the HBM roof of streaming
// the K/V cache.  Range: |x| < 65504 (checked when the cache is built).
//
// K/V cache layout ("fragment-ready", written by kvsplit_convert_kernel or directly by the
// K/V projection GEMM): per (scene, head) a sequence of 32-key blocks of 16 KB
//     [K_hi 4 KB][K_lo 4 KB][V_hi 4 KB][V_lo 4 KB]
//   K_x : [32 keys][8 chunks][8 fp16]; chunk c = 4*kh + s holds d = dmap(kh,s,e), stored at chunk
//         position c ^ ((key>>1)&7)                    (conflict-free ds_read_b128 across keys)
//   V_x : [64 d][4 chunks][8 fp16];  chunk c = 2*m + kh holds keys kmap(m,kh,e), stored at chunk
//         position c ^ ((d>>2)&3)
//   dmap(kh,s,e) = 32*(s>>1) + 16*(s&1) + 4*kh + (e&3) + 8*(e>>2)
//   kmap(m,kh,e) = 16*m + 4*kh + (e&3) + 8*(e>>2)
// Both maps are the row enumeration of the 32x32 MFMA accumulator layout, so (a) the
// projection GEMM can store its accumulators as 16-byte chunks without any shuffle, and (b) the
// softmax probabilities (S^T accumulator registers) are directly the B operand of V^T P^T.
// The LDS image of a block equals its global image: staging is a linear 16-byte copy.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

namespace parq {


namespace {

constexpr int kDH = 64;
constexpr int kBlkKeys = 32;
// cache block of 32 keys: TERMS = 3 -> [K_hi|K_lo|V_hi|V_lo] 16 KB; TERMS = 1 -> [K|V] 8 KB (offsets in 16-bit units)
template <int TERMS>
struct Blk {
    static constexpr int bytes = TERMS == 3 ? 16384 : 8192;
    static constexpr int halfs = bytes / 2;
    static constexpr int k_lo = 2048;
    static constexpr int v_hi = TERMS == 3 ? 4096 : 2048;
    static constexpr int v_lo = 6144;
};
constexpr int kStageBlks = 2;                       // 64 keys per LDS stage
constexpr int kNW = 8;                              // waves per workgroup (32 queries each)
constexpr float kDeferLog2 = 10.f;                  // running max moves only past this margin (log2 domain)
constexpr int kFlashVar = 27;                       // step variant of the product build (flash_split_pipe_kernel VAR: 1 + 2 + 8 + 16; profiles/r04_flash_variants.txt)

__device__ __forceinline__ int dmap(int kh, int s, int e) {
    return 32 * (s >> 1) + 16 * (s & 1) + 4 * kh + (e & 3) + 8 * (e >> 2);
}

// ------------------------------------------------------------------------------------------------
// fp32 head-major K/V ([b][h][n][64], as written by the fp32 projection) -> split cache.
// One workgroup per (32-key block, b*h).  Used by tests and as the fallback producer.
template <int TERMS, int KIND>
__global__ __launch_bounds__(256) void kvsplit_convert_kernel(const float* __restrict__ K, const float* __restrict__ V,
                                                              int64_t k_batch, int64_t k_head, int64_t k_row,
                                                              int64_t v_batch, int64_t v_head, int64_t v_row, int H,
                                                              int N, _Float16* __restrict__ cache, int* __restrict__ overflow) {
    __shared__ float ks[32][65];
    __shared__ float vs[32][65];
    const int blk = blockIdx.x;
    const int bh = blockIdx.y;
    const int b = bh / H, h = bh - b * H;
    const int nblk = (N + 31) / 32;
    const float* kp = K + (int64_t)b * k_batch + (int64_t)h * k_head;
    const float* vp = V + (int64_t)b * v_batch + (int64_t)h * v_head;
    bool ovf = false;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int key = i >> 6, d = i & 63;
        const int n = blk * 32 + key;
        const float kvv = n < N ? kp[(int64_t)n * k_row + d] : 0.f;
        const float vvv = n < N ? vp[(int64_t)n * v_row + d] : 0.f;
        if (KIND == kF16) ovf |= !(fabsf(kvv) < 60000.f) || !(fabsf(vvv) < 60000.f);
        ks[key][d] = kvv;
        vs[key][d] = vvv;
    }
    if (ovf) atomicOr(overflow, 1);
    __syncthreads();
    _Float16* out = cache + ((int64_t)bh * nblk + blk) * Blk<TERMS>::halfs;
    // K: 32 keys x 8 chunks = 256 chunks -> one per thread
    {
        const int key = threadIdx.x >> 3, c = threadIdx.x & 7;
        const int kh = c >> 2, s = c & 3;
        half8 hi, lo;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = ks[key][dmap(kh, s, e)];
        const int pos = c ^ ((key >> 1) & 7);
        if constexpr (TERMS == 3) {
            split8(x, hi, lo);
            *reinterpret_cast<half8*>(out + key * 64 + pos * 8) = hi;
            *reinterpret_cast<half8*>(out + Blk<3>::k_lo + key * 64 + pos * 8) = lo;
        } else {
            *reinterpret_cast<half8*>(out + key * 64 + pos * 8) = cvt8_rn<KIND>(x);
        }
    }
    // V: 64 d x 4 chunks = 256 chunks -> one per thread
    {
        const int d = threadIdx.x >> 2, c = threadIdx.x & 3;
        const int m = c >> 1, kh = c & 1;
        half8 hi, lo;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = vs[16 * m + 4 * kh + (e & 3) + 8 * (e >> 2)][d];
        const int pos = c ^ ((d >> 2) & 3);
        if constexpr (TERMS == 3) {
            split8(x, hi, lo);
            *reinterpret_cast<half8*>(out + Blk<3>::v_hi + d * 32 + pos * 8) = hi;
            *reinterpret_cast<half8*>(out + Blk<3>::v_lo + d * 32 + pos * 8) = lo;
        } else {
            *reinterpret_cast<half8*>(out + Blk<1>::v_hi + d * 32 + pos * 8) = cvt8_rn<KIND>(x);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// split cache -> fp32 head-major K / V ([b][h][n][64]): hi + lo is the fp32 value to 2^-22.  Training runs the forward on
// the split cache and hands the backward kernels plain fp32 K / V rebuilt from it (instead of a second, fp32 projection).
// H counts the heads of the CACHE (64 dims each); a model head of 64 * chunks dims is `chunks` consecutive cache heads and comes
// out as one [N][64 * chunks] plane (k_head / v_head are the strides of MODEL heads).
// TERMS = 1 (the fp16 / bf16 modes): the cache holds one rounded 16-bit value per element; the backward then differentiates through
// exactly the values the forward multiplied (straight-through the rounding).
template <int TERMS, int KIND>
__global__ __launch_bounds__(256) void kvsplit_to_f32_kernel(const _Float16* __restrict__ cache, int H, int N, float* __restrict__ K,
                                                             float* __restrict__ V, int64_t k_batch, int64_t k_head, int64_t v_batch,
                                                             int64_t v_head, int chunks) {
    const int blk = blockIdx.x, bh = blockIdx.y;
    const int b = bh / H, h = bh - b * H;
    const int nblk = (N + 31) / 32;
    const int row = 64 * chunks;
    const _Float16* in = cache + ((int64_t)bh * nblk + blk) * Blk<TERMS>::halfs;
    auto val = [](half8 hi, half8 lo, int e) -> float {
        if constexpr (TERMS == 3) return (float)hi[e] + (float)lo[e];
        else if constexpr (KIND == kF16) return (float)hi[e];
        else return (float)__builtin_bit_cast(bf16x8, hi)[e];
    };
    float* kp = K + (int64_t)b * k_batch + (int64_t)(h / chunks) * k_head + (h % chunks) * 64;
    float* vp = V + (int64_t)b * v_batch + (int64_t)(h / chunks) * v_head + (h % chunks) * 64;
    {   // K: thread -> (key, stored chunk position)
        const int key = threadIdx.x >> 3, pos = threadIdx.x & 7;
        const int c = pos ^ ((key >> 1) & 7), kh = c >> 2, s2 = c & 3;
        const half8 hi = *reinterpret_cast<const half8*>(in + key * 64 + pos * 8);
        half8 lo = hi;
        if constexpr (TERMS == 3) lo = *reinterpret_cast<const half8*>(in + Blk<3>::k_lo + key * 64 + pos * 8);
        const int n = blk * 32 + key;
        if (n < N) {
            const int d0 = 32 * (s2 >> 1) + 16 * (s2 & 1) + 4 * kh;
            float4 a = {val(hi, lo, 0), val(hi, lo, 1), val(hi, lo, 2), val(hi, lo, 3)};
            float4 c4 = {val(hi, lo, 4), val(hi, lo, 5), val(hi, lo, 6), val(hi, lo, 7)};
            *reinterpret_cast<float4*>(kp + (int64_t)n * row + d0) = a;
            *reinterpret_cast<float4*>(kp + (int64_t)n * row + d0 + 8) = c4;
        }
    }
    {   // V: thread -> (d, stored chunk position)
        const int d = threadIdx.x >> 2, pos = threadIdx.x & 3;
        const int c = pos ^ ((d >> 2) & 3), m = c >> 1, kh = c & 1;
        const half8 hi = *reinterpret_cast<const half8*>(in + Blk<TERMS>::v_hi + d * 32 + pos * 8);
        half8 lo = hi;
        if constexpr (TERMS == 3) lo = *reinterpret_cast<const half8*>(in + Blk<3>::v_lo + d * 32 + pos * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = blk * 32 + 16 * m + 4 * kh + (e & 3) + 8 * (e >> 2);
            if (n < N) vp[(int64_t)n * row + d] = val(hi, lo, e);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS holds a ring of stages (two 32-key blocks each) filled by LDS-DMA; every wave passes ONE workgroup barrier per stage.
// Round 1 ran a two-blocks-per-stage kernel here (QK(b0) | QK(b1) + softmax(b0) || PV(b0) + softmax(b1) | PV(b1), 148 us at
// BASELINE cfg 3; measured and rejected on it: waves 4..7 half a stage out of phase, K fragments read ahead of the barrier,
// register staging instead of LDS-DMA, one barrier per two stages).  The pipelined kernel below replaced it in round 2 for every
// mode (split / fp16 / bf16, with and without training dropout).
constexpr int kRing = 4;

// ------------------------------------------------------------------------------------------------
// The cross-attention kernel: ONE software-pipelined step per 32-key block,
//     step n :  QK(n+1)  ||  softmax(n)  ||  PV(n-1)
// PMC stall attribution of the two-blocks-per-stage kernel above (profiles/r02_stall_*): MFMA pipe 52 % busy, the waves 36 %
// parked (s_waitcnt / barrier) and 33 % issue-stalled; per stage the SIMD spends MFMA time (QK(b0), PV(b1): no VALU to issue)
// PLUS VALU time (softmax regions with 7+ VALU per MFMA) instead of the maximum of the two.  Here every MFMA of a step has
// softmax work of ANOTHER block to issue behind it (24 MFMAs : ~150 VALU, 6 per MFMA gap), consecutive MFMAs alternate between
// the S^T accumulator and the two O^T accumulators (few back-to-back dependent pairs), the issue order is fixed by scheduling
// fences (the sched_group_barrier pipeline was not honoured for this region), and the cross-half reductions use
// v_permlane32_swap instead of ds_bpermute.  Same cache, same LDS ring / DMA protocol, same partial outputs.
// v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of its second: fed the same value twice, the first
// comes back as the low half broadcast to both halves and the second as the high half.  Written as inline asm: with hipcc of
// ROCm 7.2 the SECOND result of __builtin_amdgcn_permlane32_swap is mis-assigned to the first result's register ("r[0] + r[1]"
// compiled to v0 + v0 even for unrelated inputs: the row sums came out doubled).  s_nop: VALU write -> permlane read hazard.
__device__ __forceinline__ void xhalf_swap(float v, float& lo_bcast, float& hi_bcast) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lo_bcast = a;
    hi_bcast = b;
}
__device__ __forceinline__ float xhalf_max(float v) {       // max over the two 32-lane halves, result in every lane
    float a, b, o;
    xhalf_swap(v, a, b);
    asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
__device__ __forceinline__ float xhalf_sum(float v) {
    float a, b;
    xhalf_swap(v, a, b);
    return a + b;
}

// RING: LDS slots of one 64-key stage each.  A stage is requested AHEAD = RING - 3 barriers before the barrier that publishes it
// (three stages are live at any time: V of the previous block, K / V of the current pair, K of the next pair).
// PROBE (development only, results are wrong when non-zero): 1 no softmax VALU, 2 no PV MFMAs, 4 no QK MFMAs, 8 no barrier / DMA waits,
// 16 no fragment reads from LDS — what each ingredient of a step costs when it is taken out.
// TERMS = 3: fp16 hi/lo split products (24 MFMAs per block).  TERMS = 1: single fp16 / bf16 products (KIND; the reduced-precision
// attention modes of BASELINE configs 2 and 5): 8 MFMAs per block with one softmax pair behind each, 8 KB cache blocks.
// DROP: training-time dropout on the probabilities (counter-based keep mask of FlashArgs::drop_seed, common.hpp): the column hashes
// of a block's 32 keys are computed once per wave (lane = key) into a wave-private LDS strip behind the ring and read back as
// four 16-byte groups; the normaliser stays undropped, 1 / (1 - p) is applied once to the partial output.
// VAR (round 4; bit mask, A/B'd on one box: profiles/r04_flash_variants.txt):
//   1  step tail: the rare "reference moves" test is per lane (any lane's own 16 scores past the threshold — the same condition
//      as the cross-half maximum past it: no v_permlane32_swap on the fast path), and the rare path no longer flushes the
//      pending PV of the previous block (four LDS reads + 12 MFMAs under the branch, after which hipcc began EVERY step with
//      s_waitcnt lgkmcnt(0) over all twelve fragment reads of the tail): it rescales l, moves the new scores and leaves the
//      factor for the O^T accumulators PENDING; the next step applies it after its PV MFMAs have added that block
//      ((O + P V) * alpha, the same value the flush produced)
//   2  K fragments s = 0, 1 of the next step requested in the middle of the step (their registers are dead after the sixth
//      QK MFMA) instead of at its end
//   8  bare s_barrier between the stages (see sync_point)
//   4  row sums with plain v_add_f32 (hipcc SLP-packs the sixteen adds of a step into v_pk_add_f32, which the guide prices
//      above two plain adds beside MFMAs)
template <int RING, int PROBE = 0, int TERMS = 3, int KIND = kF16, bool DROP = false, int VAR = 0>
__global__ __launch_bounds__(kNW * 64) void flash_split_pipe_kernel(FlashArgs a, const _Float16* __restrict__ cache) {
    PARQ_TL_KERNEL(kTlFlashSplit);
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];       // [RING stages][kStageBlks][block]
    constexpr int kBlkBytes = Blk<TERMS>::bytes;
    constexpr int kBlkHalfs = Blk<TERMS>::halfs;
    constexpr int NT = kNW * 64;
    constexpr int STAGE16 = kStageBlks * kBlkBytes / 16;
    constexpr int LD = STAGE16 / NT;
    constexpr int AHEAD = RING - 3;
    static_assert(kStageBlks == 2 && (RING == 4 || RING == 5), "slot arithmetic below");

    const int split = blockIdx.x;
    const FlashHead fh = flash_head(a, blockIdx.z);                          // per-head tiers: this launch may cover some heads only
    const int bh = fh.bh, b = fh.b, h = fh.h;
    // VAR & 16: the wave index as a scalar (readfirstlane): `active` and everything derived from it become scalar branches instead
    // of exec-masked regions, around which hipcc merges the LDS / VMEM counters conservatively
    const int tid = threadIdx.x, lane = tid & 63, wave = (VAR & 16) ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int q0 = (blockIdx.y * kNW + wave) * 32;
    const int q = q0 + li;
    const bool active = q0 < a.Lq;
    const int Lq_pad = (a.Lq + 31) & ~31;

    half8 qhi[4], qlo[4];
    {
        const float scale = 1.4426950408889634f / sqrtf((float)kDH);
        const float* qp = a.q + (int64_t)b * a.q_batch + (int64_t)h * a.q_head + (int64_t)(q < a.Lq ? q : 0) * a.q_row;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int d0 = 32 * (s >> 1) + 16 * (s & 1) + 4 * kh;
            f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + d0);
            f32x4 x1 = *reinterpret_cast<const f32x4*>(qp + d0 + 8);
            if (q >= a.Lq) { x0 = f32x4{0.f, 0.f, 0.f, 0.f}; x1 = x0; }
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (e < 4 ? x0[e & 3] : x1[e & 3]) * scale;
            if constexpr (TERMS == 3) split8(x, qhi[s], qlo[s]);
            else { qhi[s] = cvt8_rn<KIND>(x); qlo[s] = qhi[s]; }
        }
    }

    const int nblk = (a.Lk + kBlkKeys - 1) / kBlkKeys;
    const int nst = (nblk + kStageBlks - 1) / kStageBlks;
    const int t_begin = (int)((int64_t)split * nst / a.nsplit);
    const int t_end = (int)((int64_t)(split + 1) * nst / a.nsplit);
    const uint4* gsrc = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(cache) +
                                                       (int64_t)bh * (a.cache_head_bytes ? a.cache_head_bytes : (int64_t)nblk * kBlkHalfs * 2));
    const int64_t total16 = (int64_t)nblk * (kBlkBytes / 16);
    const int B0 = t_begin * kStageBlks;                                  // blocks of this split: [B0, B0 + nbk)
    const int nbk = ((t_end * kStageBlks < nblk) ? t_end * kStageBlks : nblk) - B0;
    const bool last_partial = (t_end == nst) && (a.Lk & 31) != 0;         // the split's last block is the cache's ragged last block

    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    // every lane always issues its LD loads (addresses past the end of the cache are clamped to its last chunk: a redundant copy of
    // valid data), so the number of outstanding VMEM instructions per stage is a constant and the waits below can be COUNTED
    auto gload = [&](int st, int slot) {
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            int64_t idx = (int64_t)st * STAGE16 + tid + i * NT;
            idx = idx < total16 ? idx : total16 - 1;
            lds_byte* dst = (lds_byte*)(smem_h) + ((size_t)slot * STAGE16 + i * NT + wave * 64) * 16;
            // a.flags bit 2: non-temporal policy (aux = 2) — the stream then does not displace the chain's weights and code from L2
            if (a.flags & 4) __builtin_amdgcn_global_load_lds(gsrc + idx, dst, 16, 0, 2);
            else __builtin_amdgcn_global_load_lds(gsrc + idx, dst, 16, 0, 0);
        }
    };
    // Direction of the sweep over this split's stages.  The K/V cache (393 MB at BASELINE cfg 3) does not fit the 256 MB Infinity
    // Cache, and every recurrent iteration streams all of it: with the same direction each time an LRU-like cache never hits.
    // Odd iterations therefore walk their stages (and the two blocks of a stage) backwards, starting on what the previous launch
    // touched last.  Only for key counts that are a multiple of 64 (whole stages, no ragged block); a.flags bit 1.
    const bool rev = (a.flags & 2) != 0;
    auto src_stage = [&](int j) { return rev ? t_end - 1 - j : t_begin + j; };
    // local block n -> its 16 KB image in the ring (local stage j = n / 2 lives in slot j % RING; backwards: block 1 of a stage first)
    auto lds_blk = [&](int n) -> const _Float16* {
        return smem_h + (size_t)((((n >> 1) % RING) << 1) | (rev ? 1 - (n & 1) : (n & 1))) * kBlkHalfs;
    };

#pragma unroll
    for (int j = 0; j < RING - 1; ++j)
        if (t_begin + j < t_end) gload(src_stage(j), j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // after the step of local block 2 j + 1: stage j + 2 must be visible to every wave (its K is read two steps later); it was
    // requested AHEAD barriers ago, so with RING = 5 the newest request stays in flight across this barrier (counted wait; LDS-DMA
    // returns in order).  Stage j + RING - 1 then goes into the slot of stage j - 1 (last read by step 2 j: V of block 2 j - 1).
    auto sync_point = [&](int j) {
        if constexpr (!(PROBE & 8)) {
            if (AHEAD >= 2 && t_begin + j + 3 < t_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // VAR & 8: the bare barrier.  __syncthreads() carries a workgroup fence, i.e. s_waitcnt lgkmcnt(0): the twelve fragment
            // reads the step in front of it has just issued (slots of stages j and j + 1 — not the slot the DMA below refills)
            // would be drained by every wave at the same moment.  What the barrier must order is covered without the fence: a
            // wave's own DMA by the vmcnt wait above, the last reads of the refilled slot (V of block 2 j - 1, consumed by MFMAs
            // a step ago) by program order.
            if constexpr ((VAR & 8) != 0) __builtin_amdgcn_s_barrier();
            else __syncthreads();
        }
        if (t_begin + j + RING - 1 < t_end) gload(src_stage(j + RING - 1), (j + RING - 1) % RING);
    };

    f32x16 o[2], sacc[2];
    half8 Phi[2][2], Plo[2][2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m) { Phi[c][m] = half8{0, 0, 0, 0, 0, 0, 0, 0}; Plo[c][m] = Phi[c][m]; }
    float m_run = 0.f, l_run = 0.f, l_a = 0.f, l_b = 0.f;                  // row sum = l_run + l_a + l_b (two chains in the steady state)
    float alpha_pend = 1.f;                                                // VAR & 1: factor the O^T accumulators still owe to a moved reference
    bool pend = false;                                                     // (wave-uniform) alpha_pend is to be applied after the running PV
    // -m_run in 16 registers: the C operand of the first QK MFMA of every block, so the scores leave the matrix pipe already relative
    // to the running maximum (saves a v_sub per score); rewritten only when the reference moves
    f32x16 negm16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    half8 kf[8], vh[2], vl[2];                                             // fragments of the NEXT step, requested at the end of a step
    const int ksw = (li >> 1) & 7;

    auto load_k = [&](const _Float16* Kb, half8 (&kf)[8]) {
        if constexpr (PROBE & 16) {
#pragma unroll
            for (int s = 0; s < 8; ++s) kf[s] = qhi[s & 3];
            return;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pos = (4 * kh + s) ^ ksw;
            kf[2 * s] = *reinterpret_cast<const half8*>(Kb + li * 64 + pos * 8);
            if constexpr (TERMS == 3) kf[2 * s + 1] = *reinterpret_cast<const half8*>(Kb + Blk<3>::k_lo + li * 64 + pos * 8);
        }
    };
    // chunks s = S0 .. S0 + 1 only (kf[2 S0] .. kf[2 S0 + 3])
    auto load_k_pair = [&](const _Float16* Kb, half8 (&kf)[8], auto s0) {
        constexpr int S0 = decltype(s0)::value;
        if constexpr (PROBE & 16) return;
#pragma unroll
        for (int s = S0; s < S0 + 2; ++s) {
            const int pos = (4 * kh + s) ^ ksw;
            kf[2 * s] = *reinterpret_cast<const half8*>(Kb + li * 64 + pos * 8);
            if constexpr (TERMS == 3) kf[2 * s + 1] = *reinterpret_cast<const half8*>(Kb + Blk<3>::k_lo + li * 64 + pos * 8);
        }
    };
    auto load_v = [&](const _Float16* Vb, int m, half8 (&vh)[2], half8 (&vl)[2]) {
        if constexpr (PROBE & 16) {
            vh[0] = qhi[0]; vh[1] = qhi[1]; vl[0] = qlo[0]; vl[1] = qlo[1];
            return;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int d = dt * 32 + li;
            const int pos = (2 * m + kh) ^ ((d >> 2) & 3);
            vh[dt] = *reinterpret_cast<const half8*>(Vb + Blk<TERMS>::v_hi + d * 32 + pos * 8);
            if constexpr (TERMS == 3) vl[dt] = *reinterpret_cast<const half8*>(Vb + Blk<3>::v_lo + d * 32 + pos * 8);
        }
    };
    // max of the 16 scores of this lane (keys of one kh half), then over both halves: the block maximum of query li
    // (plain fmaxf: hipcc forms v_max3 chains itself and — unlike for an asm statement — pads the MFMA-result -> VALU-read hazard)
    auto block_max = [&](const f32x16& S) -> float {
        float m0 = fmaxf(S[0], S[1]), m1 = fmaxf(S[8], S[9]);
#pragma unroll
        for (int r = 2; r < 8; ++r) { m0 = fmaxf(m0, S[r]); m1 = fmaxf(m1, S[8 + r]); }
        return xhalf_max(fmaxf(m0, m1));
    };
    const uint32_t drop_rh = DROP ? drop_rowhash(a.drop_seed, (uint32_t)(bh * a.Lq + q)) : 0u;      // this lane's query row
    const uint32_t drop_thr = DROP ? drop_threshold(a.drop_p) : 0u;
    // dropout column part (common.hpp): block hash (wave-uniform) ^ register constant ^ kh constant; the latter folds into the row hash
    const uint32_t drop_rkh = drop_rh ^ (kh ? kDropBit2Part : 0u);
    uint32_t dbase = 0u;                                                   // drop_rkh ^ block hash of the block in softmax
    // global 32-key block of local block n (the dropout column index is the key's index in the head's key axis)
    auto gblk = [&](int n) { return rev ? t_end * kStageBlks - 1 - n : B0 + n; };
    auto load_drop = [&](int n) {
        if constexpr (DROP) dbase = drop_rkh ^ drop_blockhash((uint32_t)gblk(n));
    };
    // probabilities of accumulator half m of block-parity CUR (epilogue; the steady state uses sm_pair)
    auto softmax_half = [&](auto cur, int m) {
        constexpr int CUR = decltype(cur)::value;
        float p[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p[e] = __builtin_amdgcn_exp2f(sacc[CUR][8 * m + e]);
            l_run += p[e];
            if constexpr (DROP) p[e] = drop_keep_h(dbase, drop_regpart(8 * m + e), drop_thr) ? p[e] : 0.f;
        }
        if constexpr (TERMS == 3) split8(p, Phi[CUR][m], Plo[CUR][m]);
        else Phi[CUR][m] = cvt8_rn<KIND>(p);
    };
    auto pv_half = [&](auto par, int m, const half8 (&vh)[2], const half8 (&vl)[2]) {
        constexpr int PAR = decltype(par)::value;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = mfma16<KIND>(vh[dt], Phi[PAR][m], o[dt]);
        if constexpr (TERMS == 3) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt] = mfma16<KIND>(vl[dt], Phi[PAR][m], o[dt]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt] = mfma16<KIND>(vh[dt], Plo[PAR][m], o[dt]);
        }
    };
    // rare, wave-uniform: the maximum of block n + 1 (accumulator parity NXT, relative to m_run) moves the reference.  The
    // probabilities of block n (parity CUR) are still waiting for their PV: it is done here and then cleared, so that the
    // PV of the next step adds zeros; (l, O) and the new scores move to the new reference.
    auto move_reference = [&](auto cur, int n, float mx) {
        constexpr int CUR = decltype(cur)::value, NXT = CUR ^ 1;
        const _Float16* Vb = lds_blk(n);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            half8 vh2[2], vl2[2];
            load_v(Vb, m, vh2, vl2);
            pv_half(cur, m, vh2, vl2);
            Phi[CUR][m] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            Plo[CUR][m] = Phi[CUR][m];
        }
        const float d = mx > a.defer_log2 ? mx : 0.f;
        const float alpha = __builtin_amdgcn_exp2f(-d);
        m_run += d;
        l_run *= alpha;
        l_a *= alpha;
        l_b *= alpha;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[NXT][r] -= d; negm16[r] = -m_run; }     // the scores of block n + 1 move to the new reference
    };

    // VAR & 1 forms of the two halves of move_reference: no LDS traffic, no MFMAs
    auto apply_pending = [&]() {                                           // O^T now holds every block the old reference covered
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dd][r] *= alpha_pend;
        alpha_pend = 1.f;
        pend = false;
    };
    auto move_reference_lazy = [&](auto cur, float mx) {
        constexpr int CUR = decltype(cur)::value, NXT = CUR ^ 1;
        const float d = mx > a.defer_log2 ? mx : 0.f;
        const float alpha = __builtin_amdgcn_exp2f(-d);
        m_run += d;
        l_run *= alpha;                                                    // the row sums already hold the block whose PV is pending
        l_a *= alpha;
        l_b *= alpha;
        alpha_pend = alpha;
        pend = true;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[NXT][r] -= d; negm16[r] = -m_run; }
    };

    // one softmax PAIR: elements (2 j, 2 j + 1) of accumulator half m = j / 4 -> one packed hi and one packed lo word of P[CUR][m]
    // (10 VALU: 2 sub, 2 exp, 2 add, cvt_pkrtz, 2 fma_mix, cvt_pkrtz; 8 pairs per block).  Two partial row sums keep the add chain short.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto sm_pair = [&](auto cur, auto jj) {
        if constexpr (PROBE & 1) return;
        constexpr int CUR = decltype(cur)::value, J = decltype(jj)::value, M = J >> 2, W = J & 3;
        float p0 = __builtin_amdgcn_exp2f(sacc[CUR][2 * J]);                 // the accumulator holds score - m_run
        float p1 = __builtin_amdgcn_exp2f(sacc[CUR][2 * J + 1]);
        if constexpr (VAR & 4) {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(l_a) : "v"(p0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(l_b) : "v"(p1));
        } else {
            l_a += p0;
            l_b += p1;
        }
        if constexpr (DROP) {                                                // the normaliser above stays undropped
            p0 = drop_keep_h(dbase, drop_regpart(2 * J), drop_thr) ? p0 : 0.f;
            p1 = drop_keep_h(dbase, drop_regpart(2 * J + 1), drop_thr) ? p1 : 0.f;
        }
        if constexpr (TERMS == 3) {
            half2v hi, lo;
            split_pair(p0, p1, hi, lo);
            u32x4 h4 = __builtin_bit_cast(u32x4, Phi[CUR][M]), l4 = __builtin_bit_cast(u32x4, Plo[CUR][M]);
            h4[W] = __builtin_bit_cast(unsigned int, hi);
            l4[W] = __builtin_bit_cast(unsigned int, lo);
            Phi[CUR][M] = __builtin_bit_cast(half8, h4);
            Plo[CUR][M] = __builtin_bit_cast(half8, l4);
        } else {                                                             // one round-to-nearest 16-bit value per probability
            u32x4 h4 = __builtin_bit_cast(u32x4, Phi[CUR][M]);
            if constexpr (KIND == kF16) {
                const half2v hv = {(_Float16)p0, (_Float16)p1};
                h4[W] = __builtin_bit_cast(unsigned int, hv);
            } else {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const bf16x2 bv = {(__bf16)p0, (__bf16)p1};
                h4[W] = __builtin_bit_cast(unsigned int, bv);
            }
            Phi[CUR][M] = __builtin_bit_cast(half8, h4);
        }
    };
    // VAR & 32: a pair in two halves behind consecutive MFMAs (a: the two exponentials and the hi word, b: the lo word and the row
    // sums), at most four vector instructions per MFMA gap instead of eight behind every third MFMA
    float sp0 = 0.f, sp1 = 0.f;
    half2v shi = {(_Float16)0.f, (_Float16)0.f};
    auto sm_pair_a = [&](auto cur, auto jj) {
        constexpr int CUR = decltype(cur)::value, J = decltype(jj)::value, M = J >> 2, W = J & 3;
        sp0 = __builtin_amdgcn_exp2f(sacc[CUR][2 * J]);
        sp1 = __builtin_amdgcn_exp2f(sacc[CUR][2 * J + 1]);
        shi = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(sp0, sp1));
        u32x4 h4 = __builtin_bit_cast(u32x4, Phi[CUR][M]);
        h4[W] = __builtin_bit_cast(unsigned int, shi);
        Phi[CUR][M] = __builtin_bit_cast(half8, h4);
    };
    auto sm_pair_b = [&](auto cur, auto jj) {
        constexpr int CUR = decltype(cur)::value, J = decltype(jj)::value, M = J >> 2, W = J & 3;
        l_a += sp0;
        l_b += sp1;
        float d0, d1;                                                     // p - hi as in split_pair (common.hpp): one mixed fma each, exact
        const unsigned int hp = __builtin_bit_cast(unsigned int, shi);
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hp), "v"(sp0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hp), "v"(sp1));
        const half2v lo = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(d0, d1));
        u32x4 l4 = __builtin_bit_cast(u32x4, Plo[CUR][M]);
        l4[W] = __builtin_bit_cast(unsigned int, lo);
        Plo[CUR][M] = __builtin_bit_cast(half8, l4);
    };
#define PARQ_FENCE() __builtin_amdgcn_sched_barrier(0)
    // one pipelined step: QK(n + 1) -> sacc[NXT], softmax(n) from sacc[CUR] -> P[CUR], PV(n - 1) with P[NXT].  The order below IS the
    // issue order (scheduling fences between the pieces): every MFMA is followed by at most one softmax pair, S^T and O^T
    // accumulators alternate, fragment reads are issued a dozen MFMAs ahead of their use.
    auto step = [&](auto cur, int n) {
        constexpr int CUR = decltype(cur)::value, NXT = CUR ^ 1;
        using IC = std::integral_constant<int, CUR>;
        const _Float16* Vb = lds_blk(n > 0 ? n - 1 : 0);                  // n = 0: P[NXT] is zero, any finite V will do
        // kf = K fragments of block n + 1 and vh / vl = V fragments (m = 0) of block n - 1 were requested by the previous step
        load_drop(n);
        PARQ_FENCE();
        float mx_lane_out = 0.f;
#define PARQ_Q(i, A, Bq) if constexpr (!(PROBE & 4)) sacc[NXT] = mfma16<KIND>(A, Bq, (i) == 0 ? negm16 : sacc[NXT]); PARQ_FENCE()
#define PARQ_P(D, A, Bp) if constexpr (!(PROBE & 2)) o[D] = mfma16<KIND>(A, Bp, o[D]); PARQ_FENCE()
#define PARQ_S(J) sm_pair(IC{}, std::integral_constant<int, J>{}); PARQ_FENCE()
#define PARQ_SA(J) sm_pair_a(IC{}, std::integral_constant<int, J>{}); PARQ_FENCE()
#define PARQ_SB(J) sm_pair_b(IC{}, std::integral_constant<int, J>{}); PARQ_FENCE()
        if constexpr (TERMS == 3 && (VAR & 32) != 0 && !DROP && !(PROBE & 1)) {
        // the same 24 MFMAs, every softmax pair in two halves behind consecutive MFMAs
        PARQ_Q(0, kf[0], qhi[0]);
        PARQ_P(0, vh[0], Phi[NXT][0]);  PARQ_SA(0);
        PARQ_Q(1, kf[0], qlo[0]);  PARQ_SB(0);
        PARQ_P(1, vh[1], Phi[NXT][0]);
        PARQ_Q(2, kf[1], qhi[0]);  PARQ_SA(1);
        PARQ_P(0, vl[0], Phi[NXT][0]);  PARQ_SB(1);
        PARQ_Q(3, kf[2], qhi[1]);
        PARQ_P(1, vl[1], Phi[NXT][0]);  PARQ_SA(2);
        PARQ_Q(4, kf[2], qlo[1]);  PARQ_SB(2);
        PARQ_P(0, vh[0], Plo[NXT][0]);
        PARQ_Q(5, kf[3], qhi[1]);  PARQ_SA(3);
        PARQ_P(1, vh[1], Plo[NXT][0]);  PARQ_SB(3);
        load_v(Vb, 1, vh, vl);
        if constexpr ((VAR & 2) != 0) load_k_pair(lds_blk(n + 2 < nbk ? n + 2 : nbk - 1), kf, std::integral_constant<int, 0>{});
        PARQ_FENCE();
        PARQ_Q(6, kf[4], qhi[2]);
        PARQ_Q(7, kf[4], qlo[2]);  PARQ_SA(4);
        PARQ_Q(8, kf[5], qhi[2]);  PARQ_SB(4);
        PARQ_P(0, vh[0], Phi[NXT][1]);
        PARQ_Q(9, kf[6], qhi[3]);  PARQ_SA(5);
        PARQ_P(1, vh[1], Phi[NXT][1]);  PARQ_SB(5);
        PARQ_Q(10, kf[6], qlo[3]);
        PARQ_P(0, vl[0], Phi[NXT][1]);  PARQ_SA(6);
        PARQ_Q(11, kf[7], qhi[3]);  PARQ_SB(6);
        PARQ_P(1, vl[1], Phi[NXT][1]);  PARQ_SA(7);
        PARQ_P(0, vh[0], Plo[NXT][1]);  PARQ_SB(7);
        float mx_lane;
        {
            const f32x16& S = sacc[NXT];
            float m0 = fmaxf(S[0], S[1]), m1 = fmaxf(S[8], S[9]);
#pragma unroll
            for (int r = 2; r < 8; ++r) { m0 = fmaxf(m0, S[r]); m1 = fmaxf(m1, S[8 + r]); }
            mx_lane = fmaxf(m0, m1);
        }
        PARQ_FENCE();
        PARQ_P(1, vh[1], Plo[NXT][1]);
        mx_lane_out = mx_lane;
        } else if constexpr (TERMS == 3 && (VAR & 64) != 0 && !(PROBE & 1)) {
        // VAR & 64: accumulator RUNS — consecutive MFMAs on the same accumulator (6 QK, 3 + 3 PV, 6 QK, 3 + 3 PV) instead of
        // alternating S^T / O^T: a back-to-back dependent MFMA takes its C operand from the pipeline (no 4 KB accumulator read + write
        // through the register file per MFMA), which is what a power-bound kernel would gain from; a softmax pair behind every third
        PARQ_Q(0, kf[0], qhi[0]);
        PARQ_Q(1, kf[0], qlo[0]);
        PARQ_Q(2, kf[1], qhi[0]);  PARQ_S(0);
        PARQ_Q(3, kf[2], qhi[1]);
        PARQ_Q(4, kf[2], qlo[1]);
        PARQ_Q(5, kf[3], qhi[1]);  PARQ_S(1);
        if constexpr ((VAR & 2) != 0) { load_k_pair(lds_blk(n + 2 < nbk ? n + 2 : nbk - 1), kf, std::integral_constant<int, 0>{}); PARQ_FENCE(); }
        PARQ_P(0, vh[0], Phi[NXT][0]);
        PARQ_P(0, vl[0], Phi[NXT][0]);
        PARQ_P(0, vh[0], Plo[NXT][0]);  PARQ_S(2);
        PARQ_P(1, vh[1], Phi[NXT][0]);
        PARQ_P(1, vl[1], Phi[NXT][0]);
        PARQ_P(1, vh[1], Plo[NXT][0]);  PARQ_S(3);
        load_v(Vb, 1, vh, vl);
        PARQ_FENCE();
        PARQ_Q(6, kf[4], qhi[2]);
        PARQ_Q(7, kf[4], qlo[2]);
        PARQ_Q(8, kf[5], qhi[2]);  PARQ_S(4);
        PARQ_Q(9, kf[6], qhi[3]);
        PARQ_Q(10, kf[6], qlo[3]);
        PARQ_Q(11, kf[7], qhi[3]);  PARQ_S(5);
        PARQ_P(0, vh[0], Phi[NXT][1]);
        PARQ_P(0, vl[0], Phi[NXT][1]);
        PARQ_P(0, vh[0], Plo[NXT][1]);  PARQ_S(6);
        PARQ_P(1, vh[1], Phi[NXT][1]);
        PARQ_P(1, vl[1], Phi[NXT][1]);  PARQ_S(7);
        float mx_lane;
        {
            const f32x16& S = sacc[NXT];
            float m0 = fmaxf(S[0], S[1]), m1 = fmaxf(S[8], S[9]);
#pragma unroll
            for (int r = 2; r < 8; ++r) { m0 = fmaxf(m0, S[r]); m1 = fmaxf(m1, S[8 + r]); }
            mx_lane = fmaxf(m0, m1);
        }
        PARQ_FENCE();
        PARQ_P(1, vh[1], Plo[NXT][1]);
        mx_lane_out = mx_lane;
        } else if constexpr (TERMS == 3) {
        // m = 0 half of PV(n - 1), s = 0, 1 of QK(n + 1); 8 softmax pairs (16 scores per lane and block), one behind every third MFMA
        PARQ_Q(0, kf[0], qhi[0]);
        PARQ_P(0, vh[0], Phi[NXT][0]);  PARQ_S(0);
        PARQ_Q(1, kf[0], qlo[0]);
        PARQ_P(1, vh[1], Phi[NXT][0]);
        PARQ_Q(2, kf[1], qhi[0]);  PARQ_S(1);
        PARQ_P(0, vl[0], Phi[NXT][0]);
        PARQ_Q(3, kf[2], qhi[1]);
        PARQ_P(1, vl[1], Phi[NXT][0]);  PARQ_S(2);
        PARQ_Q(4, kf[2], qlo[1]);
        PARQ_P(0, vh[0], Plo[NXT][0]);
        PARQ_Q(5, kf[3], qhi[1]);  PARQ_S(3);
        PARQ_P(1, vh[1], Plo[NXT][0]);
        load_v(Vb, 1, vh, vl);                                            // m = 1 fragments (the m = 0 registers are free now)
        if constexpr ((VAR & 2) != 0) load_k_pair(lds_blk(n + 2 < nbk ? n + 2 : nbk - 1), kf, std::integral_constant<int, 0>{});   // kf[0..3] are dead
        PARQ_FENCE();
        // m = 1 half, s = 2, 3
        PARQ_Q(6, kf[4], qhi[2]);
        PARQ_Q(7, kf[4], qlo[2]);  PARQ_S(4);
        PARQ_Q(8, kf[5], qhi[2]);
        PARQ_P(0, vh[0], Phi[NXT][1]);
        PARQ_Q(9, kf[6], qhi[3]);  PARQ_S(5);
        PARQ_P(1, vh[1], Phi[NXT][1]);
        PARQ_Q(10, kf[6], qlo[3]);
        PARQ_P(0, vl[0], Phi[NXT][1]);  PARQ_S(6);
        PARQ_Q(11, kf[7], qhi[3]);
        PARQ_P(1, vl[1], Phi[NXT][1]);  PARQ_S(7);
        PARQ_P(0, vh[0], Plo[NXT][1]);
        // the maximum of the new scores (QK finished three MFMAs ago) is taken behind the last two PV MFMAs
        float mx_lane;
        {
            const f32x16& S = sacc[NXT];
            float m0 = fmaxf(S[0], S[1]), m1 = fmaxf(S[8], S[9]);
#pragma unroll
            for (int r = 2; r < 8; ++r) { m0 = fmaxf(m0, S[r]); m1 = fmaxf(m1, S[8 + r]); }
            mx_lane = fmaxf(m0, m1);
        }
        PARQ_FENCE();
        PARQ_P(1, vh[1], Plo[NXT][1]);
        mx_lane_out = mx_lane;
        } else {
        // single products: 4 + 4 MFMAs per block, one softmax pair behind each
        PARQ_Q(0, kf[0], qhi[0]);  PARQ_S(0);
        PARQ_P(0, vh[0], Phi[NXT][0]);  PARQ_S(1);
        PARQ_Q(1, kf[2], qhi[1]);  PARQ_S(2);
        PARQ_P(1, vh[1], Phi[NXT][0]);
        load_v(Vb, 1, vh, vl);
        PARQ_FENCE();
        PARQ_S(3);
        PARQ_Q(2, kf[4], qhi[2]);  PARQ_S(4);
        PARQ_P(0, vh[0], Phi[NXT][1]);  PARQ_S(5);
        PARQ_Q(3, kf[6], qhi[3]);  PARQ_S(6);
        float mx_lane;
        {
            const f32x16& S = sacc[NXT];
            float m0 = fmaxf(S[0], S[1]), m1 = fmaxf(S[8], S[9]);
#pragma unroll
            for (int r = 2; r < 8; ++r) { m0 = fmaxf(m0, S[r]); m1 = fmaxf(m1, S[8 + r]); }
            mx_lane = fmaxf(m0, m1);
        }
        PARQ_S(7);
        PARQ_P(1, vh[1], Phi[NXT][1]);
        mx_lane_out = mx_lane;
        }
#undef PARQ_Q
#undef PARQ_P
#undef PARQ_S
#undef PARQ_SA
#undef PARQ_SB
        if constexpr ((VAR & 1) != 0) {
            if constexpr ((VAR & 2) != 0 && TERMS == 3) load_k_pair(lds_blk(n + 2 < nbk ? n + 2 : nbk - 1), kf, std::integral_constant<int, 2>{});
            else load_k(lds_blk(n + 2 < nbk ? n + 2 : nbk - 1), kf);
            load_v(lds_blk(n), 0, vh, vl);
            PARQ_FENCE();
            // rare, wave-uniform, register-only: (1) a factor left pending by the previous step (its PV has been added by now),
            // (2) the maximum of block n + 1 moves the reference (any lane's own maximum past the threshold <=> some query's)
            const bool moves = __any(mx_lane_out > a.defer_log2);
            if (pend || moves) {
                if (pend) apply_pending();
                if (moves) move_reference_lazy(cur, xhalf_max(mx_lane_out));
            }
        } else {
        // fragments of the next step (K of block n + 2, clamped at the split's end; V of block n), in flight during the reduction
        if constexpr ((VAR & 2) != 0 && TERMS == 3) load_k_pair(lds_blk(n + 2 < nbk ? n + 2 : nbk - 1), kf, std::integral_constant<int, 2>{});
        else load_k(lds_blk(n + 2 < nbk ? n + 2 : nbk - 1), kf);
        load_v(lds_blk(n), 0, vh, vl);
        PARQ_FENCE();
        const float mx = xhalf_max(mx_lane_out);                             // relative to m_run
        if (__any(mx > a.defer_log2)) move_reference(cur, n, mx);
        }
    };

    // the second-dispatched half of the workgroup loses every VALU arbitration against the older half (priority, then age):
    // one static s_setprio for it, no per-phase flips (MI355X_MICROARCH.md, two waves per SIMD)
    if ((a.flags & 1) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    if (nbk > 0) {
        if (active) {
            // prologue: scores of block 0 against a zero reference, then the reference becomes their maximum
            load_k(lds_blk(0), kf);
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sacc[0] = mfma16<KIND>(kf[2 * s], qhi[s], s == 0 ? zero16 : sacc[0]);
                if constexpr (TERMS == 3) {
                    sacc[0] = mfma16<KIND>(kf[2 * s], qlo[s], sacc[0]);
                    sacc[0] = mfma16<KIND>(kf[2 * s + 1], qhi[s], sacc[0]);
                }
            }
            if (nbk == 1 && last_partial) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (B0 * 32 + mfma32_row(r, lane) >= a.Lk) sacc[0][r] = -INFINITY;
            }
            m_run = block_max(sacc[0]);
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[0][r] -= m_run; negm16[r] = -m_run; }
            load_k(lds_blk(nbk > 1 ? 1 : 0), kf);                          // fragments of step 0
            load_v(lds_blk(0), 0, vh, vl);
        }
        // steady state: two steps (one LDS stage) per barrier
        int n = 0;
        for (; n + 2 < nbk; n += 2) {
            if (active) {
                step(std::integral_constant<int, 0>{}, n);
                step(std::integral_constant<int, 1>{}, n + 1);
            }
            sync_point(n >> 1);
        }
        if (active) {
            if (n + 1 < nbk) { step(std::integral_constant<int, 0>{}, n); ++n; }
            // epilogue: softmax of the last block (ragged key axis masked here), PV of the last two blocks
            auto finish = [&](auto cur) {
                constexpr int CUR = decltype(cur)::value, NXT = CUR ^ 1;
                load_drop(n);
                if (last_partial) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((B0 + n) * 32 + mfma32_row(r, lane) >= a.Lk) sacc[CUR][r] = -INFINITY;
                }
                softmax_half(cur, 0);
                softmax_half(cur, 1);
                if (n > 0) {
                    const _Float16* Vp = lds_blk(n - 1);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        half8 vh[2], vl[2];
                        load_v(Vp, m, vh, vl);
                        pv_half(std::integral_constant<int, NXT>{}, m, vh, vl);
                    }
                }
                if constexpr ((VAR & 1) != 0) { if (pend) apply_pending(); }   // the last block's maximum moved the reference
                const _Float16* Vb = lds_blk(n);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    half8 vh[2], vl[2];
                    load_v(Vb, m, vh, vl);
                    pv_half(cur, m, vh, vl);
                }
            };
            if (n & 1) finish(std::integral_constant<int, 1>{});
            else finish(std::integral_constant<int, 0>{});
        }
    } else if (active) {
        m_run = -INFINITY;                                                 // a split without keys: weight 0 in the merge
    }

    if (active) {
        const int64_t pbase = (int64_t)blockIdx.z * a.nsplit + split;
        float* op = a.o_part + pbase * kDH * Lq_pad;
        const float drop_scale = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
        if (a.flags & 8) {
            // write-through publication (a.flags bit 3): the O^T tile of this wave goes through the (now idle) K/V ring so that a lane
            // holds four consecutive queries of one dim, and leaves as 16-byte sc1 stores: the 64 KB a workgroup hands to the merge
            // kernel then drain while other workgroups still compute instead of sitting dirty in L2 until the end-of-kernel
            // write-back (MI355X_MICROARCH.md "publish-large")
            __syncthreads();                                                    // every wave is done with the ring
            float* tr = reinterpret_cast<float*>(smem_h) + wave * (64 * 36);
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) tr[(d * 32 + mfma32_row(r, lane)) * 36 + (lane & 31)] = o[d][r] * drop_scale;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // same-wave LDS round trip
            __builtin_amdgcn_wave_barrier();
            const int q0w = q - (lane & 31);
            typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)op, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i * 64 + lane, dim = idx >> 3, q4 = idx & 7;
                const f32x4 v = *reinterpret_cast<const f32x4*>(tr + dim * 36 + q4 * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rs, (int)((dim * Lq_pad + q0w + q4 * 4) * 4), 0, 16);
            }
        } else {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) op[(int64_t)(d * 32 + mfma32_row(r, lane)) * Lq_pad + q] = o[d][r] * drop_scale;
        }
        const float l_tot = xhalf_sum(l_run + l_a + l_b);
        if (kh == 0) {
            a.m_part[pbase * Lq_pad + q] = m_run;
            a.l_part[pbase * Lq_pad + q] = l_tot;
        }
    }
}

}  // namespace

size_t kvsplit_cache_bytes(int B, int H, int N, int terms) {
    return (size_t)B * H * ceil_div(N, kBlkKeys) * (terms == 3 ? Blk<3>::bytes : Blk<1>::bytes);
}

int flash_split_stage_keys() { return kStageBlks * kBlkKeys; }

int flash_split_pick_splits(int B, int H, int Lq, int Lk, int num_cus) {
    const int nst = ceil_div(ceil_div(Lk, kBlkKeys), kStageBlks);
    const int64_t base = (int64_t)B * H * ceil_div(Lq, 32 * kNW);
    int64_t want = ceil_div64((int64_t)num_cus, base);
    if (want < 1) want = 1;
    if (want > nst) want = nst;
    if (want > 256) want = 256;
    return (int)want;
}

hipError_t launch_kvsplit_convert(const float* K, const float* V, int64_t k_batch, int64_t k_head, int64_t k_row,
                                  int64_t v_batch, int64_t v_head, int64_t v_row, int B, int H, int N, void* cache,
                                  int* overflow_flag, hipStream_t s, int terms, int kind) {
    dim3 grid(ceil_div(N, kBlkKeys), B * H);
    _Float16* c = reinterpret_cast<_Float16*>(cache);
    if (terms == 3)
        hipLaunchKernelGGL((kvsplit_convert_kernel<3, kF16>), grid, dim3(256), 0, s, K, V, k_batch, k_head, k_row, v_batch, v_head,
                           v_row, H, N, c, overflow_flag);
    else if (kind == kF16)
        hipLaunchKernelGGL((kvsplit_convert_kernel<1, kF16>), grid, dim3(256), 0, s, K, V, k_batch, k_head, k_row, v_batch, v_head,
                           v_row, H, N, c, overflow_flag);
    else
        hipLaunchKernelGGL((kvsplit_convert_kernel<1, kBF16>), grid, dim3(256), 0, s, K, V, k_batch, k_head, k_row, v_batch,
                           v_head, v_row, H, N, c, overflow_flag);
    return hipGetLastError();
}

hipError_t launch_kvsplit_to_f32(const void* cache, int B, int H, int N, float* K, float* V, int64_t k_batch, int64_t k_head,
                                 int64_t v_batch, int64_t v_head, hipStream_t s, int chunks, int terms, int kind) {
    if (chunks < 1 || H % chunks != 0) return hipErrorInvalidValue;
    const dim3 grid(ceil_div(N, kBlkKeys), B * H);
    const _Float16* c = reinterpret_cast<const _Float16*>(cache);
    if (terms == 3) hipLaunchKernelGGL((kvsplit_to_f32_kernel<3, kF16>), grid, dim3(256), 0, s, c, H, N, K, V, k_batch, k_head, v_batch, v_head, chunks);
    else if (kind == kF16) hipLaunchKernelGGL((kvsplit_to_f32_kernel<1, kF16>), grid, dim3(256), 0, s, c, H, N, K, V, k_batch, k_head, v_batch, v_head, chunks);
    else hipLaunchKernelGGL((kvsplit_to_f32_kernel<1, kBF16>), grid, dim3(256), 0, s, c, H, N, K, V, k_batch, k_head, v_batch, v_head, chunks);
    return hipGetLastError();
}

hipError_t launch_flash_split(const FlashArgs& a, const void* cache, hipStream_t s, int terms, int kind) {
    if (a.dh != kDH || a.nsplit < 1 || a.nsplit > 256) return hipErrorInvalidValue;
    FlashArgs b = a;
    static const float defer = [] { const char* e = dev_env("PARQ_DEFER_LOG2"); return e ? (float)atof(e) : kDeferLog2; }();   // debugging knob, read once
    b.defer_log2 = defer;
    static const int prio = [] { const char* e = dev_env("PARQ_FLASH_PRIO"); return e ? atoi(e) : 0; }();
    const bool alt = [] { const char* e = dev_env("PARQ_FLASH_ALTERNATE"); return !(e && e[0] == '0'); }();     // (per launch: tools/flash_variants.py)
    // bit 1 (set by the caller for every other recurrent iteration): sweep backwards — only for whole 64-key stages
    static const int nt = [] { const char* e = dev_env("PARQ_FLASH_NT"); return e ? atoi(e) : 0; }();
    static const int wt = [] { const char* e = dev_env("PARQ_FLASH_WT"); return e ? atoi(e) : 1; }();      // write-through partials (0: plain stores; measured 1.852 -> 1.846 ms)
    b.flags = (prio & 1) | ((alt && (a.flags & 2) && (a.Lk % (kStageBlks * kBlkKeys)) == 0) ? 2 : 0) | (nt ? 4 : 0) | ((wt && a.Lq % 256 == 0 && terms == 3) ? 8 : 0);
    const dim3 grid(b.nsplit, ceil_div(b.Lq, 32 * kNW), b.B * flash_launch_heads(b));
    const _Float16* c16 = reinterpret_cast<const _Float16*>(cache);
#define PARQ_PIPE_LAUNCH_V(RING, PROBE, T, K, D, V)                                                                                      \
    {                                                                                                                                    \
        static DynLdsOnce once;                                                                                                          \
        const size_t lds = (size_t)RING * kStageBlks * Blk<T>::bytes + ((D) ? kNW * 32 * sizeof(uint32_t) : 0);                          \
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&flash_split_pipe_kernel<RING, PROBE, T, K, D, V>), lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL((flash_split_pipe_kernel<RING, PROBE, T, K, D, V>), grid, dim3(kNW * 64), lds, s, b, c16);                     \
        return hipGetLastError();                                                                                                        \
    }
#define PARQ_PIPE_LAUNCH(RING, PROBE, T, K, D) PARQ_PIPE_LAUNCH_V(RING, PROBE, T, K, D, kFlashVar)
    const bool drop = b.drop_p > 0.f;
    static const bool stage1_off = [] { const char* e = dev_env("PARQ_FLASH_STAGE1"); return e && e[0] == '0'; }();
    // modes 2 / 3 on whole 64-key stages: the step of the mode-4 kernel with one product (flash_split8.hip)
    if (terms != 3 && !stage1_off && flash_split8_supported(a.dh, a.Lk)) return launch_flash_single_stage(a, cache, s, kind);
    if (terms != 3) {                                            // single fp16 / bf16 products (attention modes 2 / 3)
#ifdef PARQ_DEV_PROBES
        if (!drop && kind == kF16) {
            const int var1 = [] { const char* e = dev_env("PARQ_FLASH_VAR"); return e ? atoi(e) : kFlashVar; }();
            if (var1 == 0) PARQ_PIPE_LAUNCH_V(4, 0, 1, kF16, false, 0)
            if (var1 == 25) PARQ_PIPE_LAUNCH_V(4, 0, 1, kF16, false, 25)
        }
#endif
        if (kind == kF16) { if (drop) PARQ_PIPE_LAUNCH(4, 0, 1, kF16, true) else PARQ_PIPE_LAUNCH(4, 0, 1, kF16, false) }
        if (drop) PARQ_PIPE_LAUNCH(4, 0, 1, kBF16, true) else PARQ_PIPE_LAUNCH(4, 0, 1, kBF16, false)
    }
    if (drop) PARQ_PIPE_LAUNCH(4, 0, 3, kF16, true)
#ifdef PARQ_DEV_PROBES
    static const int probe = [] { const char* e = dev_env("PARQ_FLASH_PROBE"); return e ? atoi(e) : 0; }();   // development: see the kernel
    switch (probe) {
        case 0: break;
        case 1: PARQ_PIPE_LAUNCH(4, 1, 3, kF16, false)
        case 2: PARQ_PIPE_LAUNCH(4, 2, 3, kF16, false)
        case 4: PARQ_PIPE_LAUNCH(4, 4, 3, kF16, false)
        case 6: PARQ_PIPE_LAUNCH(4, 6, 3, kF16, false)
        case 7: PARQ_PIPE_LAUNCH(4, 7, 3, kF16, false)
        case 8: PARQ_PIPE_LAUNCH(4, 8, 3, kF16, false)
        case 16: PARQ_PIPE_LAUNCH(4, 16, 3, kF16, false)
        case 25: PARQ_PIPE_LAUNCH(4, 25, 3, kF16, false)
        case 31: PARQ_PIPE_LAUNCH(4, 31, 3, kF16, false)
        default: return hipErrorInvalidValue;
    }
#endif
#ifdef PARQ_DEV_PROBES
    static const int ring = [] { const char* e = dev_env("PARQ_FLASH_RING"); return e && e[0] == '5' ? 5 : 4; }();   // measured 2 % slower (round 2)
    if (ring == 5) PARQ_PIPE_LAUNCH(5, 0, 3, kF16, false)
    const int var = [] { const char* e = dev_env("PARQ_FLASH_VAR"); return e ? atoi(e) : kFlashVar; }();   // step variants; read per launch: tools/flash_variants.py walks them in one process
    switch (var) {
        case 0: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 0)
        case 1: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 1)
        case 3: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 3)
        case 4: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 4)
        case 5: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 5)
        case 7: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 7)
        case 8: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 8)
        case 9: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 9)
        case 11: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 11)
        case 13: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 13)
        case 15: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 15)
        case 16: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 16)
        case 25: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 25)
        case 27: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 27)
        case 29: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 29)
        case 31: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 31)
        case 32: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 32)
        case 59: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 59)
        case 91: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 91)
        case 89: PARQ_PIPE_LAUNCH_V(4, 0, 3, kF16, false, 89)
        default: return hipErrorInvalidValue;
    }
#endif
    PARQ_PIPE_LAUNCH(4, 0, 3, kF16, false)
#undef PARQ_PIPE_LAUNCH
#undef PARQ_PIPE_LAUNCH_V
}

PARQ_TL_DEFINE_SETTER(tl_set_flash_split)

}  // namespace parq
