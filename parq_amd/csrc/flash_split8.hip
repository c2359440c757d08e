// Flash cross-attention of attention mode 4 ("split8"; head dim 64): Q K^T as a split-precision product whose CROSS TERMS run on the
// MX-scaled fp8 matrix instruction, P V as an fp16 product with a self-consistent normaliser.
//
//   scores   q.k  ~=  q_hi16 . k_hi16         v_mfma_f32_32x32x16_f16           (exact 11 x 11-bit products, fp32 accumulate)
//                  +  e4m3(q) . e4m3(k_lo 2^10) 2^-10                           } v_mfma_scale_f32_32x32x64_f8f6f4: one instruction
//                  +  e4m3(q_lo 2^10) 2^-10 . e4m3(k)                           } per cross term and 64-long contraction
//            with hi = fp16(x) rounded toward zero and lo = x - hi.  The cross terms are 2^-11 of the product, so the 4 significant
//            bits of e4m3 put their rounding at ~2^-15 of it.  A score is what the exponential amplifies: it keeps 15 bits.
//   output   sum_j p~_j v~_j / sum_j p~_j   with p~ = fp16(p), v~ = fp16(v), both rounded to nearest, and the row sum taken over the SAME
//            p~ the matrix pipe multiplies: the weights p~ / sum p~ stay self-consistent, a row that one key dominates reproduces that
//            key's v~ and a row that spreads over N keys averages the 2^-12 rounding noise of weights and values down by sqrt(N).
// Measured on the reference's fixtures the decoder outputs sit 4e-6 .. 1.3e-5 from float64 (all three terms of both products in fp16:
// 2e-6; one fp16 product for everything: 1e-4; the reference's own fp32 run: 6e-5 .. 1.4e-4); tests/emulate_attention_arithmetic.py and
// tests/calibrate_split8_guard.py are the CPU models.  The error model needs rows that spread over many keys: the merge kernel flags
// rows that do not (FlashArgs::peaky) and the caller falls back to the fp16 x 3 kernel.
// What it buys: this kernel family is bound by the power its matrix work draws (profiles/r04_flash_power_budget_probe.txt): with real
// operand bits a whole-chip stream of v_mfma_f32_32x32x16_f16 runs at 20 ns per instruction and SIMD (32.7 cycles of the 2.4 GHz clock
// on zero operands, ~48 on random ones: tools/bench_src/mfma_operands.hip), the MX instruction at 36 ns for FOUR times the contraction
// (tools/bench_src/mx_energy.hip).  Matrix work per 32-key block: 8 x 20 + 2 x 36 ns instead of 24 x 20.
//
// Cache ("stage" = 64 keys = 24 KB, the LDS image equals the global image; written by kvsplit8_convert_kernel or by the K/V
// projection), byte offsets inside a stage:
//       0  K hi16  [2 blocks][32 keys][8 chunks][8 fp16]     chunk swizzle and d order as in the split cache (flash_split.hip)
//    8192  K hi8   [2 blocks][2 c][2 h][32 keys][16 B]       piece (c, h) of a key: byte 8 m + e <-> d = 32 m + 16 c + 4 h + (e & 3) + 8 (e >> 2)
//   12288  K lo8   same, e4m3(lo 2^10)
//   16384  V fp16  [2 blocks][64 d][4 chunks][8 fp16]        round to nearest; layout as the V_hi plane of the split cache
// The fp8 byte order is the register order of the 32 x 32 accumulator map, so the K/V projection stores its accumulators as 16-byte
// pieces.  MX operand semantics (lane l: row / column l & 31, 32 k-values; E8M0 scale byte per lane): tools/bench_src/mx_probe.hip.
// Whole stages only: Lk % 64 == 0 (the callers fall back to the fp16 x 3 kernel).
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

namespace parq {

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kDH = 64, kNW = 8, kRing = 4;
constexpr int kStageBytes = kStage8Bytes;
constexpr int oKh16 = kS8Kh16, oK8hi = kS8K8hi, oK8lo = kS8K8lo, oVh16 = kS8Vh16;
constexpr float kDefer8 = 2.f;                       // the reference maximum trails the row maximum by at most 2 (log2 domain): the row sums
                                                     // the peakedness guard reads stay within a factor 4 of the sums relative to the row maximum
constexpr int kE8One = 127, kE8Lo = 117;             // E8M0 scales: 2^0, 2^-10 (lo parts of K, Q)

__device__ __forceinline__ int dmap(int kh, int s, int e) { return 32 * (s >> 1) + 16 * (s & 1) + 4 * kh + (e & 3) + 8 * (e >> 2); }

// ------------------------------------------------------------------------------------------------
// fp32 head-major K / V -> stage cache.  One workgroup per (stage, b * h); tests and the stand-alone attention entry point.
__global__ __launch_bounds__(256) void kvsplit8_convert_kernel(const float* __restrict__ K, const float* __restrict__ V, int64_t k_batch,
                                                               int64_t k_head, int64_t k_row, int64_t v_batch, int64_t v_head,
                                                               int64_t v_row, int H, int N, unsigned char* __restrict__ cache) {
    __shared__ float ks[64][65];
    __shared__ float vs[64][65];
    const int st = blockIdx.x, bh = blockIdx.y, tid = threadIdx.x;
    const int b = bh / H, h = bh - b * H;
    const int nst = N / 64;
    const float* kp = K + (int64_t)b * k_batch + (int64_t)h * k_head;
    const float* vp = V + (int64_t)b * v_batch + (int64_t)h * v_head;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int key = i >> 6, d = i & 63;
        const int64_t n = (int64_t)st * 64 + key;
        ks[key][d] = kp[n * k_row + d];
        vs[key][d] = vp[n * v_row + d];
    }
    __syncthreads();
    unsigned char* out = cache + ((int64_t)bh * nst + st) * kStageBytes;
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
        {   // K hi16 (round toward zero: the value the lo part is taken against): 32 keys x 8 chunks
            const int key = tid >> 3, c = tid & 7, kh = c >> 2, s = c & 3;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = ks[32 * b2 + key][dmap(kh, s, e)];
            const int pos = c ^ ((key >> 1) & 7);
            half8 hi, lo;
            split8(x, hi, lo);
            *reinterpret_cast<half8*>(out + oKh16 + b2 * 4096 + (key * 64 + pos * 8) * 2) = hi;
        }
        {   // V fp16 (round to nearest): 64 d x 4 chunks
            const int d = tid >> 2, c = tid & 3, m = c >> 1, kh = c & 1;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = vs[32 * b2 + 16 * m + 4 * kh + (e & 3) + 8 * (e >> 2)][d];
            const int pos = c ^ ((d >> 2) & 3);
            *reinterpret_cast<half8*>(out + oVh16 + b2 * 4096 + (d * 32 + pos * 8) * 2) = cvt8_rn<kF16>(x);
        }
    }
    {   // K hi8 / lo8: piece id = ((b2 * 2 + c) * 2 + h) * 32 + key = tid
        const int b2 = tid >> 7, c = (tid >> 6) & 1, hh = (tid >> 5) & 1, key = tid & 31;
        float x[16];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[8 * m + e] = ks[32 * b2 + key][32 * m + 16 * c + 4 * hh + (e & 3) + 8 * (e >> 2)];
        i32x4 hi8, lo8;
        pieces_e4m3(x, hi8, lo8);
        *reinterpret_cast<i32x4*>(out + oK8hi + tid * 16) = hi8;
        *reinterpret_cast<i32x4*>(out + oK8lo + tid * 16) = lo8;
    }
}

__device__ __forceinline__ void xhalf_swap(float v, float& lo_bcast, float& hi_bcast) {     // see flash_split.hip
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lo_bcast = a;
    hi_bcast = b;
}
__device__ __forceinline__ float xhalf_max(float v) {
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

__device__ __forceinline__ f32x16 mx64(i32x8 a, i32x8 b, f32x16 c, int scale_a, int scale_b) {
    // cbsz = blgp = 0: both operands fp8 e4m3; the E8M0 scales in byte 0 of their registers
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
}

// ------------------------------------------------------------------------------------------------
// One software-pipelined step per 32-key block, as flash_split_pipe_kernel:   QK(n + 1)  ||  softmax(n)  ||  PV(n - 1).
// Per block: 4 fp16 MFMAs + 2 MX for the scores, 4 fp16 MFMAs for P V (against 24 fp16 MFMAs in the fp16 x 3 kernel); per probability
// pair: two exp2, one packed round-to-nearest convert (v_cvt_pk_f16_f32), one v_dot2_f32_f16 against (1, 1) for the row sum (it adds
// exactly the two values the matrix pipe will multiply; fp16 subnormals are kept by it and by the MFMA: tools/bench_src/denorm_probe.hip).
// Running maximum: moves only when a score exceeds it by more than kDefer8, and then by an INTEGER d (ceil), so that what still waits
// for its P V is rescaled exactly: the O^T accumulators and row sums by 2^-d, the pending fp16 probabilities by 2^-d (a power of two).
// Fragments of the next step are requested as soon as their registers are dead; no scheduling fences (hipcc's own order of a step
// measured 3 % faster than the written one pinned with fences).
// REV: the sweep direction over the split's stages (a template parameter: the block-in-stage selects are then literals).
// PROBE (development, results wrong): 2 no PV MFMAs, 4 no QK MFMAs, 8 no barrier / DMA waits, 16 no fragment reads; 32 (results right):
// the written order of a step pinned with scheduling fences.
// DROP: training-time dropout on the probabilities, the counter-based keep mask of FlashArgs::drop_seed exactly as in flash_split_pipe_kernel
// (common.hpp; the backward kernels rebuild the same mask): the normaliser stays undropped, 1 / (1 - p) is applied once to the partial output.
// TERMS = 1 (attention modes 2 / 3 on whole 64-key stages): the same step with ONE KIND (fp16 / bf16) product for the scores — Q and K
// rounded to nearest once, no cross terms — on the single-product cache of flash_split.hip (a stage = two 8 KB blocks [K | V]).
template <int PROBE = 0, int RING = kRing, bool REV = false, bool DROP = false, int TERMS = 8, int KIND = kF16>
__global__ __launch_bounds__(kNW * 64) void flash_split8_kernel(FlashArgs a, const unsigned char* __restrict__ cache) {
    PARQ_TL_KERNEL(kTlFlashSplit);
    static_assert(TERMS == 8 || TERMS == 1, "mode 4 or a single 16-bit product");
    static_assert(TERMS == 1 || KIND == kF16, "the split is an fp16 split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [RING stages]
    constexpr bool MX = TERMS == 8;
    constexpr int kStageBytes = MX ? kStage8Bytes : 16384;                    // (shadows the file-level constant of the mode-4 stage)
    // byte offset of physical block pb's K / V plane inside a stage
    auto k16_off = [](int pb) { return MX ? oKh16 + pb * 4096 : pb * 8192; };
    auto v16_off = [](int pb) { return MX ? oVh16 + pb * 4096 : pb * 8192 + 4096; };
    constexpr int NT = kNW * 64;
    constexpr int STAGE16 = kStageBytes / 16;
    constexpr int LD = STAGE16 / NT;
    static_assert(STAGE16 % NT == 0, "whole DMA rows per stage");
    constexpr int AHEAD = RING - 3;                                          // see flash_split.hip: a stage is requested AHEAD barriers before the one that publishes it

    const int split = blockIdx.x;
    const FlashHead fh = flash_head(a, blockIdx.z);                          // per-head tiers: this launch may cover some heads only
    const int bh = fh.bh, b = fh.b, h = fh.h;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int q0 = (blockIdx.y * kNW + wave) * 32;
    const int q = q0 + li;
    const bool active = q0 < a.Lq;
    const int Lq_pad = (a.Lq + 31) & ~31;

    // ---- Q operands: hi16 fragments as in the split kernel; the same 32 values per lane as e4m3 (hi8) and e4m3(lo 2^10)
    half8 qh[4];
    i32x8 q8h, q8l;
    {
        const float scale = 1.4426950408889634f / sqrtf((float)kDH);
        const float* qp = a.q + (int64_t)b * a.q_batch + (int64_t)h * a.q_head + (int64_t)(q < a.Lq ? q : 0) * a.q_row;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int d0 = 32 * (s >> 1) + 16 * (s & 1) + 4 * kh;
            f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + d0);
            f32x4 x1 = *reinterpret_cast<const f32x4*>(qp + d0 + 8);
            if (q >= a.Lq) { x0 = f32x4{0.f, 0.f, 0.f, 0.f}; x1 = x0; }
            float x[8], dl[8];
            unsigned hw[4];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (e < 4 ? x0[e & 3] : x1[e & 3]) * scale;
            if constexpr (!MX) { qh[s] = cvt8_rn<KIND>(x); continue; }
#pragma unroll
            for (int e = 0; e < 8; e += 2) split_rtz(x[e], x[e + 1], hw[e >> 1], dl[e], dl[e + 1]);
            qh[s] = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
            // byte 16 c + 8 m + e of the fp8 operands <-> d = 32 m + 16 c + 4 h + (e & 3) + 8 (e >> 2), with s = 2 m + c
            const int m = s >> 1, c = s & 1;
            q8h[4 * c + 2 * m] = pack4_e4m3(x[0], x[1], x[2], x[3]);
            q8h[4 * c + 2 * m + 1] = pack4_e4m3(x[4], x[5], x[6], x[7]);
            q8l[4 * c + 2 * m] = pack4_e4m3(dl[0] * kLo8Scale, dl[1] * kLo8Scale, dl[2] * kLo8Scale, dl[3] * kLo8Scale);
            q8l[4 * c + 2 * m + 1] = pack4_e4m3(dl[4] * kLo8Scale, dl[5] * kLo8Scale, dl[6] * kLo8Scale, dl[7] * kLo8Scale);
        }
    }

    const int nst = a.Lk / 64;
    const int t_begin = (int)((int64_t)split * nst / a.nsplit);
    const int t_end = (int)((int64_t)(split + 1) * nst / a.nsplit);
    const uint4* gsrc = reinterpret_cast<const uint4*>(cache + (int64_t)bh * (a.cache_head_bytes ? a.cache_head_bytes : (int64_t)nst * kStageBytes));
    const int nbk = 2 * (t_end - t_begin);                                  // 32-key blocks of this split: always whole stages

    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    auto gload = [&](int st, int slot) {
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int64_t idx = (int64_t)st * STAGE16 + tid + i * NT;       // st is a stage of this split: always inside the cache
            lds_byte* dst = (lds_byte*)(smem) + ((size_t)slot * STAGE16 + i * NT + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds(gsrc + idx, dst, 16, 0, 0);
        }
    };
    // odd iterations walk the stages (and the two blocks of a stage) backwards: flash_split.hip
    constexpr bool rev = REV;
    auto src_stage = [&](int j) { return rev ? t_end - 1 - j : t_begin + j; };
    auto stage_of = [&](int n) -> const unsigned char* { return smem + (size_t)((n >> 1) % RING) * kStageBytes; };
    auto pblk = [&](int n) { return rev ? 1 - (n & 1) : (n & 1); };       // physical block of local block n inside its stage

#pragma unroll
    for (int j = 0; j < RING - 1; ++j)
        if (t_begin + j < t_end) gload(src_stage(j), j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto sync_point = [&](int j) {
        if constexpr (!(PROBE & 8)) {
            if (AHEAD >= 2 && t_begin + j + 3 < t_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                    // bare barrier: see flash_split.hip (VAR & 8)
        }
        if (t_begin + j + RING - 1 < t_end) gload(src_stage(j + RING - 1), (j + RING - 1) % RING);
    };

    f32x16 o[2], sacc[2];
    half8 Ph[2][2];                                                        // fp16 probabilities: [block parity][accumulator half]
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m) Ph[c][m] = half8{0, 0, 0, 0, 0, 0, 0, 0};
    float m_run = 0.f, l_run = 0.f, l_a = 0.f, l_b = 0.f;
    f32x16 negm16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    half8 kf[4], vh[2];                                                    // fp16 fragments of the NEXT step
    i32x8 k8h, k8l;
    if constexpr (PROBE & 16) { kf[0] = qh[0]; kf[1] = qh[1]; kf[2] = qh[2]; kf[3] = qh[3]; k8h = q8h; k8l = q8l; vh[0] = qh[0]; vh[1] = qh[1]; }
    const int ksw = (li >> 1) & 7;

    auto load_k16 = [&](int n) {
        if constexpr (PROBE & 16) return;
        const _Float16* Kb = reinterpret_cast<const _Float16*>(stage_of(n) + k16_off(pblk(n)));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pos = (4 * kh + s) ^ ksw;
            kf[s] = *reinterpret_cast<const half8*>(Kb + li * 64 + pos * 8);
        }
    };
    auto load_k8 = [&](int n) {
        if constexpr ((PROBE & 16) || !MX) return;
        const unsigned char* st = stage_of(n);
        const int pb = pblk(n);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int off = (((pb * 2 + c) * 2 + kh) * 32 + li) * 16;
            const i32x4 hv = *reinterpret_cast<const i32x4*>(st + oK8hi + off);
            const i32x4 lv = *reinterpret_cast<const i32x4*>(st + oK8lo + off);
#pragma unroll
            for (int w = 0; w < 4; ++w) { k8h[4 * c + w] = hv[w]; k8l[4 * c + w] = lv[w]; }
        }
    };
    auto load_v = [&](int n, int m, half8 (&vh)[2]) {
        if constexpr (PROBE & 16) { vh[0] = qh[0]; vh[1] = qh[1]; return; }
        const _Float16* Vb = reinterpret_cast<const _Float16*>(stage_of(n) + v16_off(pblk(n)));
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int d = dt * 32 + li;
            const int pos = (2 * m + kh) ^ ((d >> 2) & 3);
            vh[dt] = *reinterpret_cast<const half8*>(Vb + d * 32 + pos * 8);
        }
    };
    auto block_max = [&](const f32x16& S) -> float {
        float m0 = fmaxf(S[0], S[1]), m1 = fmaxf(S[8], S[9]);
#pragma unroll
        for (int r = 2; r < 8; ++r) { m0 = fmaxf(m0, S[r]); m1 = fmaxf(m1, S[8 + r]); }
        return xhalf_max(fmaxf(m0, m1));
    };
    const uint32_t drop_rh = DROP ? drop_rowhash(a.drop_seed, (uint32_t)(bh * a.Lq + q)) : 0u;      // this lane's query row
    const uint32_t drop_thr = DROP ? drop_threshold(a.drop_p) : 0u;
    const uint32_t drop_rkh = drop_rh ^ (kh ? kDropBit2Part : 0u);
    uint32_t dbase = 0u;                                                   // drop_rkh ^ block hash of the block in softmax
    auto load_drop = [&](int n) {                                          // n -> its global 32-key block (the dropout column index is the key's)
        if constexpr (DROP) dbase = drop_rkh ^ drop_blockhash((uint32_t)(rev ? 2 * t_end - 1 - n : 2 * t_begin + n));
    };
    typedef float f32x2p __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
    // two probabilities -> one packed 16-bit pair, rounded to nearest (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32: unbiased also among fp16
    // subnormals); l += exactly those two values (v_dot2c against (1, 1))
    auto pack_sum = [&](float p0, float p1, float& l) -> unsigned {
        if constexpr (KIND == kF16) {
            const half2v hp = __builtin_convertvector(f32x2p{p0, p1}, half2v);
            l = __builtin_amdgcn_fdot2(hp, half2v{(_Float16)1.f, (_Float16)1.f}, l, false);
            return __builtin_bit_cast(unsigned, hp);
        } else {
            const bf16x2p bp = __builtin_convertvector(f32x2p{p0, p1}, bf16x2p);
            l = __builtin_amdgcn_fdot2_f32_bf16(bp, bf16x2p{(__bf16)1.f, (__bf16)1.f}, l, false);
            return __builtin_bit_cast(unsigned, bp);
        }
    };
    // one softmax pair: elements (2 J, 2 J + 1) of the accumulator -> one fp16 word of Ph[CUR][J / 4] and their sum into the row sum
    auto sm_pair = [&](auto cur, auto jj) {
        if constexpr (PROBE & 1) return;
        constexpr int CUR = decltype(cur)::value, J = decltype(jj)::value, M = J >> 2, W = J & 3;
        const float p0 = __builtin_amdgcn_exp2f(sacc[CUR][2 * J]);             // the accumulator holds score - m_run
        const float p1 = __builtin_amdgcn_exp2f(sacc[CUR][2 * J + 1]);
        unsigned hw;                                                        // the pair as the matrix pipe will read it, and its sum into the row sum
        if constexpr ((J & 1) == 0) hw = pack_sum(p0, p1, l_a);
        else hw = pack_sum(p0, p1, l_b);
        if constexpr (DROP)                                                 // the row sum above stays undropped
            hw &= (drop_keep_h(dbase, drop_regpart(2 * J), drop_thr) ? 0xffffu : 0u) | (drop_keep_h(dbase, drop_regpart(2 * J + 1), drop_thr) ? 0xffff0000u : 0u);
        u32x4 h4 = __builtin_bit_cast(u32x4, Ph[CUR][M]);
        h4[W] = hw;
        Ph[CUR][M] = __builtin_bit_cast(half8, h4);
    };
#define PARQ_FENCE() do { if constexpr ((PROBE & 32) != 0) __builtin_amdgcn_sched_barrier(0); } while (0)
    // step n: QK(n + 1) -> sacc[NXT]; softmax(n) from sacc[CUR]; P V of block n - 1
    auto step = [&](auto cur, int n) {
        constexpr int CUR = decltype(cur)::value, NXT = CUR ^ 1;
        using IC = std::integral_constant<int, CUR>;
        const int nk = n + 2 < nbk ? n + 2 : nbk - 1;                       // K of the next step (clamped at the split's end)
        load_drop(n);
#define PARQ_Q(i, Bq) if constexpr (!(PROBE & 4)) sacc[NXT] = mfma16<KIND>(kf[i], Bq, (i) == 0 ? negm16 : sacc[NXT]); PARQ_FENCE()
#define PARQ_P(D, Bp) if constexpr (!(PROBE & 2)) o[D] = mfma16<KIND>(vh[D], Bp, o[D]); PARQ_FENCE()
#define PARQ_S(J) sm_pair(IC{}, std::integral_constant<int, J>{}); PARQ_FENCE()
        PARQ_Q(0, qh[0]);
        PARQ_P(0, Ph[NXT][0]);  PARQ_S(0);
        PARQ_Q(1, qh[1]);
        PARQ_P(1, Ph[NXT][0]);  PARQ_S(1);
        load_v(n > 0 ? n - 1 : 0, 1, vh);
        PARQ_FENCE();
        PARQ_Q(2, qh[2]);  PARQ_S(2);
        PARQ_Q(3, qh[3]);
        load_k16(nk);
        PARQ_FENCE();
        PARQ_S(3);
        if constexpr (!(PROBE & 4) && MX) sacc[NXT] = mx64(k8l, q8h, sacc[NXT], kE8Lo, kE8One);
        PARQ_FENCE();
        PARQ_S(4);
        PARQ_P(0, Ph[NXT][1]);  PARQ_S(5);
        if constexpr (!(PROBE & 4) && MX) sacc[NXT] = mx64(k8h, q8l, sacc[NXT], kE8One, kE8Lo);
        PARQ_FENCE();
        load_k8(nk);
        PARQ_FENCE();
        PARQ_S(6);
        PARQ_P(1, Ph[NXT][1]);  PARQ_S(7);
        load_v(n, 0, vh);
        PARQ_FENCE();
#undef PARQ_Q
#undef PARQ_P
#undef PARQ_S
        float mx_lane;
        {
            const f32x16& S = sacc[NXT];
            float m0 = fmaxf(S[0], S[1]), m1 = fmaxf(S[8], S[9]);
#pragma unroll
            for (int r = 2; r < 8; ++r) { m0 = fmaxf(m0, S[r]); m1 = fmaxf(m1, S[8 + r]); }
            mx_lane = fmaxf(m0, m1);
        }
        // rare, wave-uniform, register-only: some query's maximum of block n + 1 is more than kDefer8 past the reference
        if (__any(mx_lane > a.defer_log2)) {
            const float mx = xhalf_max(mx_lane);
            const float d = mx > a.defer_log2 ? ceilf(mx) : 0.f;
            const float alpha = __builtin_ldexpf(1.f, -(int)d);
            m_run += d;
            l_run *= alpha;
            l_a *= alpha;
            l_b *= alpha;
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[NXT][r] -= d; negm16[r] = -m_run; }
            // block n's probabilities (relative to the old reference) still wait for their P V: times 2^-d, exact in fp16 down to its
            // subnormals (what falls under them is under 2^-24 of the new reference)
            if constexpr (KIND == kF16) {
                const _Float16 ah = (_Float16)alpha;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int e = 0; e < 8; ++e) Ph[CUR][m][e] *= ah;
            } else {                                                        // bf16: exact through fp32 (8 significant bits, fp32's exponent range)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    bf16x8 bv = __builtin_bit_cast(bf16x8, Ph[CUR][m]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = (__bf16)((float)bv[e] * alpha);
                    Ph[CUR][m] = __builtin_bit_cast(half8, bv);
                }
            }
        }
    };

    if ((a.flags & 1) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    if (nbk > 0) {
        if (active) {
            // prologue: scores of block 0 against a zero reference, then the reference becomes their maximum
            load_k16(0);
            load_k8(0);
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc[0] = mfma16<KIND>(kf[s], qh[s], s == 0 ? zero16 : sacc[0]);
            if constexpr (MX) {
                sacc[0] = mx64(k8l, q8h, sacc[0], kE8Lo, kE8One);
                sacc[0] = mx64(k8h, q8l, sacc[0], kE8One, kE8Lo);
            }
            m_run = block_max(sacc[0]);
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[0][r] -= m_run; negm16[r] = -m_run; }
            load_k16(1);                                                     // nbk >= 2
            load_k8(1);
            load_v(0, 0, vh);
        }
        int n = 0;
        for (; n + 2 < nbk; n += 2) {
            if (active) {
                step(std::integral_constant<int, 0>{}, n);
                step(std::integral_constant<int, 1>{}, n + 1);
            }
            sync_point(n >> 1);
        }
        if (active) {
            step(std::integral_constant<int, 0>{}, n);                       // n = nbk - 2
            ++n;
            // epilogue: softmax of the last block (odd), P V of the last two blocks
            load_drop(n);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                unsigned hw[4];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(sacc[1][8 * m + e]), p1 = __builtin_amdgcn_exp2f(sacc[1][8 * m + e + 1]);
                    hw[e >> 1] = pack_sum(p0, p1, l_run);
                    if constexpr (DROP)
                        hw[e >> 1] &= (drop_keep_h(dbase, drop_regpart(8 * m + e), drop_thr) ? 0xffffu : 0u) |
                                      (drop_keep_h(dbase, drop_regpart(8 * m + e + 1), drop_thr) ? 0xffff0000u : 0u);
                }
                Ph[1][m] = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    half8 v2[2];
                    load_v(n - 1 + blk, m, v2);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) o[dt] = mfma16<KIND>(v2[dt], Ph[blk][m], o[dt]);
                }
        }
    } else if (active) {
        m_run = -INFINITY;                                                 // a split without keys: weight 0 in the merge
    }

    if (active) {
        const int64_t pbase = (int64_t)blockIdx.z * a.nsplit + split;
        float* op = a.o_part + pbase * kDH * Lq_pad;
        if constexpr (DROP) {
            const float drop_scale = 1.f / (1.f - a.drop_p);
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= drop_scale;
        }
        if (a.flags & 8) {
            // write-through publication of the partial outputs through the (idle) ring: see flash_split.hip
            __syncthreads();
            float* tr = reinterpret_cast<float*>(smem) + wave * (64 * 36);
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) tr[(d * 32 + mfma32_row(r, lane)) * 36 + (lane & 31)] = o[d][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
                for (int r = 0; r < 16; ++r) op[(int64_t)(d * 32 + mfma32_row(r, lane)) * Lq_pad + q] = o[d][r];
        }
        const float l_tot = xhalf_sum(l_run + l_a + l_b);
        if (kh == 0) {
            a.m_part[pbase * Lq_pad + q] = m_run;
            a.l_part[pbase * Lq_pad + q] = l_tot;
        }
    }
}

}  // namespace

bool flash_split8_supported(int dh, int Lk) { return dh == kDH && Lk >= 64 && Lk % 64 == 0; }

hipError_t launch_kvsplit8_convert(const float* K, const float* V, int64_t k_batch, int64_t k_head, int64_t k_row, int64_t v_batch,
                                   int64_t v_head, int64_t v_row, int B, int H, int N, void* cache, hipStream_t s) {
    if (!flash_split8_supported(kDH, N)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kvsplit8_convert_kernel, dim3(N / 64, B * H), dim3(256), 0, s, K, V, k_batch, k_head, k_row, v_batch, v_head,
                       v_row, H, N, reinterpret_cast<unsigned char*>(cache));
    return hipGetLastError();
}

hipError_t launch_flash_split8(const FlashArgs& a, const void* cache, hipStream_t s) {
    if (!flash_split8_supported(a.dh, a.Lk) || a.nsplit < 1 || a.nsplit > 256) return hipErrorInvalidValue;
    FlashArgs b = a;
    b.defer_log2 = kDefer8;
    static const int wt = [] { const char* e = dev_env("PARQ_FLASH_WT"); return e ? atoi(e) : 1; }();
    b.flags = ((a.flags & 2) ? 2 : 0) | ((wt && a.Lq % 256 == 0) ? 8 : 0);
    const dim3 grid(b.nsplit, ceil_div(b.Lq, 32 * kNW), b.B * flash_launch_heads(b));
    const unsigned char* c8 = reinterpret_cast<const unsigned char*>(cache);
#define PARQ_F8_LAUNCH_RR(PROBE, RING, REV)                                                                                    \
    {                                                                                                                          \
        static DynLdsOnce once;                                                                                                \
        const size_t lds = (size_t)(RING) * kStageBytes;                                                                       \
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&flash_split8_kernel<PROBE, RING, REV>), lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL((flash_split8_kernel<PROBE, RING, REV>), grid, dim3(kNW * 64), lds, s, b, c8);                       \
        return hipGetLastError();                                                                                              \
    }
#define PARQ_F8_LAUNCH(PROBE)                                                                                                  \
    {                                                                                                                          \
        if (b.flags & 2) PARQ_F8_LAUNCH_RR(PROBE, kRing, true)                                                                 \
        PARQ_F8_LAUNCH_RR(PROBE, kRing, false)                                                                                 \
    }
#define PARQ_F8_LAUNCH_DROP(REV)                                                                                               \
    {                                                                                                                          \
        static DynLdsOnce once;                                                                                                \
        const size_t lds = (size_t)kRing * kStageBytes;                                                                        \
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&flash_split8_kernel<0, kRing, REV, true>), lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL((flash_split8_kernel<0, kRing, REV, true>), grid, dim3(kNW * 64), lds, s, b, c8);                    \
        return hipGetLastError();                                                                                              \
    }
    if (b.drop_p > 0.f) { if (b.flags & 2) PARQ_F8_LAUNCH_DROP(true) PARQ_F8_LAUNCH_DROP(false) }
#ifdef PARQ_DEV_PROBES
    static const int probe = [] { const char* e = dev_env("PARQ_FLASH_PROBE"); return e ? atoi(e) : 0; }();
    switch (probe) {
        case 0: break;
        case 2: PARQ_F8_LAUNCH(2)
        case 4: PARQ_F8_LAUNCH(4)
        case 6: PARQ_F8_LAUNCH(6)
        case 8: PARQ_F8_LAUNCH(8)
        case 16: PARQ_F8_LAUNCH(16)
        case 22: PARQ_F8_LAUNCH(22)
        case 32: PARQ_F8_LAUNCH(32)
        default: return hipErrorInvalidValue;
    }
#endif
    PARQ_F8_LAUNCH(0)
#undef PARQ_F8_LAUNCH
#undef PARQ_F8_LAUNCH_RR
#undef PARQ_F8_LAUNCH_DROP
}

// Attention modes 2 / 3 (one fp16 / bf16 product per score and per output) on whole 64-key stages: the step of the kernel above without
// cross terms, on the single-product cache.  Six 16 KB stages in the ring (the write-through epilogue needs 72 KB of it).
hipError_t launch_flash_single_stage(const FlashArgs& a, const void* cache, hipStream_t s, int kind) {
    if (!flash_split8_supported(a.dh, a.Lk) || a.nsplit < 1 || a.nsplit > 256 || (kind != kF16 && kind != kBF16)) return hipErrorInvalidValue;
    FlashArgs b = a;
    b.defer_log2 = kDefer8;
    b.flags = ((a.flags & 2) ? 2 : 0) | ((a.Lq % 256 == 0) ? 8 : 0);
    b.peaky = nullptr;
    const dim3 grid(b.nsplit, ceil_div(b.Lq, 32 * kNW), b.B * flash_launch_heads(b));
    const unsigned char* c8 = reinterpret_cast<const unsigned char*>(cache);
    constexpr int kRing1 = 6;
#define PARQ_F1_LAUNCH(REV, DROP, KIND)                                                                                        \
    {                                                                                                                          \
        static DynLdsOnce once;                                                                                                \
        const size_t lds = (size_t)kRing1 * 16384;                                                                             \
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&flash_split8_kernel<0, kRing1, REV, DROP, 1, KIND>), lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL((flash_split8_kernel<0, kRing1, REV, DROP, 1, KIND>), grid, dim3(kNW * 64), lds, s, b, c8);          \
        return hipGetLastError();                                                                                              \
    }
    const bool rev = (b.flags & 2) != 0, drop = b.drop_p > 0.f;
    if (kind == kF16) {
        if (drop) { if (rev) PARQ_F1_LAUNCH(true, true, kF16) PARQ_F1_LAUNCH(false, true, kF16) }
        if (rev) PARQ_F1_LAUNCH(true, false, kF16)
        PARQ_F1_LAUNCH(false, false, kF16)
    }
    if (drop) { if (rev) PARQ_F1_LAUNCH(true, true, kBF16) PARQ_F1_LAUNCH(false, true, kBF16) }
    if (rev) PARQ_F1_LAUNCH(true, false, kBF16)
    PARQ_F1_LAUNCH(false, false, kBF16)
#undef PARQ_F1_LAUNCH
}


PARQ_TL_DEFINE_SETTER(tl_set_flash_split8)

}  // namespace parq
