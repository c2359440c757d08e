// Split-precision flash cross-attention with the CROSS TERMS on the MX-scaled fp8 matrix instruction (head dim 64, attention mode 4).
//
// An fp32 operand x is carried as hi = fp16(x) (round toward zero) and lo = x - hi, and a product is evaluated as
//      a.b  ~=  a_hi16 . b_hi16        v_mfma_f32_32x32x16_f16                 (exact 11 x 11-bit products, fp32 accumulate)
//            +  e4m3(a) . e4m3(b_lo 2^10) 2^-10                                 } v_mfma_scale_f32_32x32x64_f8f6f4: one instruction
//            +  e4m3(a_lo 2^10) 2^-10 . e4m3(b)                                 } per cross term and 64-long contraction
// The cross terms are 2^-11 of the product, so the 4 significant bits of e4m3 put their rounding at ~2^-15 of it.  The decoder's
// instantiation carries the probabilities as one fp16 value each (P16 below), i.e. P V has the V_lo cross term only.  Measured on the
// reference's fixtures the kernel sits 3.5e-6 .. 1e-5 from float64 at the decoder outputs (all three terms in fp16: 2e-6; one fp16
// product for everything: 1e-4; tests/emulate_attention_arithmetic.py and tests/calibrate_split8_guard.py are the CPU models).  The
// error model needs rows that spread over many keys: the merge kernel flags rows that do not (FlashArgs::peaky) and the caller
// falls back to the fp16 x 3 kernel.  What it buys: this kernel family is bound by the
// power the matrix pipe draws (profiles/r04_flash_power_budget_probe.txt): with real operand bits a whole-chip stream of
// v_mfma_f32_32x32x16_f16 runs at 20 ns per instruction and SIMD (32.7 cycles of the 2.4 GHz clock on zero operands, ~48 on random
// ones: tools/bench_src/mfma_operands.hip), and the MX instruction at 36 ns for FOUR times the contraction (tools/bench_src/mx_energy.hip).
// A 32 x 32 x 64 cross term costs 36 ns instead of 80, a whole split product 4 x 20 + 2 x 36 = 152 ns instead of 240 (measured 157).
//
// Cache ("stage" = 64 keys = 28 KB, the LDS image equals the global image; written by kvsplit8_convert_kernel or by the K/V
// projection), byte offsets inside a stage:
//       0  K hi16  [2 blocks][32 keys][8 chunks][8 fp16]     chunk swizzle and d order as in the split cache (flash_split.hip)
//    8192  K hi8   [2 blocks][2 c][2 h][32 keys][16 B]       piece (c, h) of a key: byte 8 m + e <-> d = 32 m + 16 c + 4 h + (e & 3) + 8 (e >> 2)
//   12288  K lo8   same, e4m3(lo 2^10)
//   16384  V hi16  [2 blocks][64 d][4 chunks][8 fp16]        as in the split cache
//   24576  V lo8   [2 blocks][2 dt][2 h][32 li][16 B]        piece of d = 32 dt + li: byte r <-> key (r & 3) + 8 (r >> 2) + 4 h of the block
//  (28672  V hi8   same — only in the 32 KB form of the kernel tests' p_lo instantiation: the decoder's kernel has no P_lo . V_hi term)
// The byte orders are the register orders of the 32 x 32 accumulator map, so (a) the K/V projection stores its accumulators as
// 16-byte pieces, (b) the probabilities of a lane (S^T accumulator registers of two consecutive blocks) ARE the 32 k-values of the
// MX B operand: P^T of a 64-key stage against V^T, one instruction per cross term and 32 output dims.
// The hardware takes the E8M0 scale of the first 16 bytes of a lane pair (l, l + 32) from lane l and of the second 16 bytes from
// lane l + 32 (tools/bench_src/mx_probe.hip): the two 32-key blocks of a stage have separate probability scales, which is what a
// running-max move between them needs (below).  Whole stages only: Lk % 64 == 0 (the callers fall back to the fp16 x 3 kernel).
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

namespace parq {

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kDH = 64, kNW = 8, kRing = 4;
constexpr int oKh16 = kS8Kh16, oK8hi = kS8K8hi, oK8lo = kS8K8lo, oVh16 = kS8Vh16, oV8hi = kS8V8hi, oV8lo = kS8V8lo;
constexpr float kDefer8 = 2.f;                       // probabilities stay under 2^2: p 2^6 fits e4m3 (max 448)
constexpr int kE8One = 127, kE8Lo = 117;             // E8M0 scales: 2^0, 2^-10 (lo parts of K, V, Q)
constexpr int kE8Phi = 121, kE8Plo = 111;            // probabilities: hi8 = e4m3(p 2^6), lo8 = e4m3(p_lo 2^16)

__device__ __forceinline__ int dmap(int kh, int s, int e) { return 32 * (s >> 1) + 16 * (s & 1) + 4 * kh + (e & 3) + 8 * (e >> 2); }

// ------------------------------------------------------------------------------------------------
// fp32 head-major K / V -> stage cache.  One workgroup per (stage, b * h); tests and the stand-alone attention entry point.
template <bool FULL>
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
    constexpr int kStageBytes = FULL ? kStage8BytesFull : kStage8Bytes;
    unsigned char* out = cache + ((int64_t)bh * nst + st) * kStageBytes;
    auto hi16x8 = [](const float* x) {
        half8 hi, lo;
        split8(x, hi, lo);
        return hi;
    };
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
        {   // K hi16: 32 keys x 8 chunks
            const int key = tid >> 3, c = tid & 7, kh = c >> 2, s = c & 3;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = ks[32 * b2 + key][dmap(kh, s, e)];
            const int pos = c ^ ((key >> 1) & 7);
            *reinterpret_cast<half8*>(out + oKh16 + b2 * 4096 + (key * 64 + pos * 8) * 2) = hi16x8(x);
        }
        {   // V hi16: 64 d x 4 chunks
            const int d = tid >> 2, c = tid & 3, m = c >> 1, kh = c & 1;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = vs[32 * b2 + 16 * m + 4 * kh + (e & 3) + 8 * (e >> 2)][d];
            const int pos = c ^ ((d >> 2) & 3);
            *reinterpret_cast<half8*>(out + oVh16 + b2 * 4096 + (d * 32 + pos * 8) * 2) = hi16x8(x);
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
    {   // V hi8 / lo8: piece id = ((b2 * 2 + dt) * 2 + h) * 32 + li = tid
        const int b2 = tid >> 7, dt = (tid >> 6) & 1, hh = (tid >> 5) & 1, li = tid & 31;
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = vs[32 * b2 + (r & 3) + 8 * (r >> 2) + 4 * hh][32 * dt + li];
        i32x4 hi8, lo8;
        pieces_e4m3(x, hi8, lo8);
        if constexpr (FULL) *reinterpret_cast<i32x4*>(out + oV8hi + tid * 16) = hi8;
        *reinterpret_cast<i32x4*>(out + oV8lo + tid * 16) = lo8;
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

__device__ __forceinline__ f32x16 mx64(i32x8 a, i32x8 b, f32x16 c, int sel_a, int scale_a, int sel_b, int scale_b) {
    // cbsz = blgp = 0: both operands fp8 e4m3
    if (sel_b == 0) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 1, scale_b);
}

// ------------------------------------------------------------------------------------------------
// One software-pipelined step per 32-key block, as flash_split_pipe_kernel:   QK(n + 1)  ||  softmax(n)  ||  PV(n - 1),
// with the cross terms of P V taken once per STAGE: the MX contraction is 64 keys long, so P of blocks 2 j and 2 j + 1 is collected
// in one pair of fp8 registers (16 + 16 bytes per lane) and multiplied at the end of step 2 j + 1 — four MX instructions per stage
// (2 cross terms x 2 halves of the head dim) instead of 2 x 16 fp16 ones.  Per block: 8 fp16 MFMAs (4 QK, 4 PV) + 2 MX (QK) + 2 MX (PV,
// amortised; 1 in the decoder's P16 form) against 24 fp16 MFMAs.
// Running maximum: moves only when a score exceeds it by more than kDefer8, and then by an INTEGER d (ceil), so that everything
// still waiting to be multiplied is rescaled exactly: the O^T accumulators and row sums by 2^-d, the pending fp16 probabilities
// by 2^-d (a power of two), the pending fp8 probabilities through the E8M0 scale operand of their block.
// PROBE (development, results wrong): 1 no softmax VALU, 2 no PV MFMAs, 4 no QK MFMAs, 8 no barrier / DMA waits, 16 no fragment reads;
// 32 (results right): the written order of a step pinned with scheduling fences — 108.9 us against 105.3 for hipcc's own order of the
// same instructions (round 4, one box), so the product build has none.
// P16: the probabilities enter P V as ONE fp16 value each (round to nearest; no lo part: the P_lo . V_hi cross term, its conversions and its MX
// instructions are gone) and the NORMALISER sums those same rounded values, so the weights p~ / sum p~ stay self-consistent: a row that
// one key dominates is exact, a spread row averages the 2^-12 relative weight noise away (modelled on the reference's fixtures:
// 3.5e-6 -> 3.9e-6 at 96 000 keys, 2.3e-6 -> 8.6e-6 at 15 360; tests/calibrate_split8_guard.py; measured 4.4e-6 / 3.5e-6 / 9.7e-6 on g19 / g18 / g15).
// V keeps both of its terms.  P16 = false is the form of the kernel tests (parq_k_attention_split8 with p_lo = 1).
template <int PROBE = 0, int RING = kRing, bool REV = false, bool P16 = true>
__global__ __launch_bounds__(kNW * 64) void flash_split8_kernel(FlashArgs a, const unsigned char* __restrict__ cache) {
    PARQ_TL_KERNEL(kTlFlashSplit);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [kRing stages]
    constexpr int NT = kNW * 64;
    constexpr int kStageBytes = P16 ? kStage8Bytes : kStage8BytesFull;        // 28 KB without the V hi8 plane, 32 KB with it
    constexpr int STAGE16 = kStageBytes / 16;
    constexpr int LD = (STAGE16 + NT - 1) / NT;                              // the last DMA row of a 28 KB stage is half a row: waves 0 .. 3
    constexpr int AHEAD = RING - 3;                                          // see flash_split.hip: a stage is requested AHEAD barriers before the one that publishes it

    const int split = blockIdx.x;
    const int bh = blockIdx.z;
    const int b = bh / a.H, h = bh - b * a.H;
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
    const uint4* gsrc = reinterpret_cast<const uint4*>(cache + (int64_t)bh * nst * kStageBytes);
    const int nbk = 2 * (t_end - t_begin);                                  // 32-key blocks of this split: always whole stages

    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    auto gload = [&](int st, int slot) {
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            if ((i + 1) * NT > STAGE16 && i * NT + wave * 64 >= STAGE16) continue;        // scalar: past the end of a 28 KB stage
            const int64_t idx = (int64_t)st * STAGE16 + tid + i * NT;       // st is a stage of this split: always inside the cache
            lds_byte* dst = (lds_byte*)(smem) + ((size_t)slot * STAGE16 + i * NT + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds(gsrc + idx, dst, 16, 0, 0);
        }
    };
    // odd iterations walk the stages (and the two blocks of a stage) backwards: flash_split.hip
    constexpr bool rev = REV;                                              // a template parameter: the block-in-stage selects below are then literals
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
            if (AHEAD >= 2 && STAGE16 % NT == 0 && t_begin + j + 3 < t_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                    // bare barrier: see flash_split.hip (VAR & 8)
        }
        if (t_begin + j + RING - 1 < t_end) gload(src_stage(j + RING - 1), (j + RING - 1) % RING);
    };

    f32x16 o[2], sacc[2];
    half8 Ph[2][2];                                                        // fp16 probabilities: [block parity][accumulator half]
    i32x8 p8h, p8l;                                                        // fp8 probabilities of a stage: bytes 16 (n & 1) + r
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m) Ph[c][m] = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < 8; ++w) { p8h[w] = 0; p8l[w] = 0; }
    float m_run = 0.f, l_run = 0.f, l_a = 0.f, l_b = 0.f;
    constexpr int kPscale0 = kE8Phi | (kE8Plo << 8);
    int pscale = kPscale0;                                                 // byte 0 / 1: E8M0 scale of this lane's block of p8h / p8l
    int pend_d = 0;                                                        // running-max moves since that block's probabilities were taken
    f32x16 negm16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    half8 kf[4], vh[2];                                                    // fp16 fragments of the NEXT step
    i32x8 k8h, k8l;
    if constexpr (PROBE & 16) { kf[0] = qh[0]; kf[1] = qh[1]; kf[2] = qh[2]; kf[3] = qh[3]; k8h = q8h; k8l = q8l; vh[0] = qh[0]; vh[1] = qh[1]; }
    const int ksw = (li >> 1) & 7;

    auto load_k16 = [&](int n) {
        if constexpr (PROBE & 16) return;
        const _Float16* Kb = reinterpret_cast<const _Float16*>(stage_of(n) + oKh16 + pblk(n) * 4096);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pos = (4 * kh + s) ^ ksw;
            kf[s] = *reinterpret_cast<const half8*>(Kb + li * 64 + pos * 8);
        }
    };
    auto load_k8 = [&](int n) {
        if constexpr (PROBE & 16) return;
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
    auto load_k = [&](int n) { load_k16(n); load_k8(n); };
    auto load_v = [&](int n, int m, half8 (&vh)[2]) {
        if constexpr (PROBE & 16) { vh[0] = qh[0]; vh[1] = qh[1]; return; }
        const _Float16* Vb = reinterpret_cast<const _Float16*>(stage_of(n) + oVh16 + pblk(n) * 4096);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int d = dt * 32 + li;
            const int pos = (2 * m + kh) ^ ((d >> 2) & 3);
            vh[dt] = *reinterpret_cast<const half8*>(Vb + d * 32 + pos * 8);
        }
    };
    // fp8 V fragments of the stage of local blocks (n, n + 1), n even: registers 0..3 <- the local-even block, 4..7 <- the odd one
    auto load_v8 = [&](int n, i32x8 (&v8h)[2], i32x8 (&v8l)[2]) {
        if constexpr (PROBE & 16) { v8h[0] = q8h; v8h[1] = q8h; v8l[0] = q8l; v8l[1] = q8l; return; }
        const unsigned char* st = stage_of(n);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int pb = rev ? 1 - par : par;
                const int off = (((pb * 2 + dt) * 2 + kh) * 32 + li) * 16;
                const i32x4 lv = *reinterpret_cast<const i32x4*>(st + oV8lo + off);
#pragma unroll
                for (int w = 0; w < 4; ++w) v8l[dt][4 * par + w] = lv[w];
                if constexpr (!P16) {
                    const i32x4 hv = *reinterpret_cast<const i32x4*>(st + oV8hi + off);
#pragma unroll
                    for (int w = 0; w < 4; ++w) v8h[dt][4 * par + w] = hv[w];
                }
            }
    };
    // cross terms of P V for the stage whose probabilities sit in p8h / p8l
    auto pvx = [&](const i32x8 (&v8h)[2], const i32x8 (&v8l)[2]) {
        if constexpr (PROBE & 2) return;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = mx64(v8l[dt], p8h, o[dt], 0, kE8Lo, 0, pscale);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            if constexpr (!P16) o[dt] = mx64(v8h[dt], p8l, o[dt], 0, kE8One, 1, pscale);
    };
    auto block_max = [&](const f32x16& S) -> float {
        float m0 = fmaxf(S[0], S[1]), m1 = fmaxf(S[8], S[9]);
#pragma unroll
        for (int r = 2; r < 8; ++r) { m0 = fmaxf(m0, S[r]); m1 = fmaxf(m1, S[8 + r]); }
        return xhalf_max(fmaxf(m0, m1));
    };
    // one softmax pair: elements (2 J, 2 J + 1) of the accumulator -> one fp16 word of Ph[CUR][J / 4], two bytes of p8h and of p8l
    auto sm_pair = [&](auto cur, auto jj) {
        if constexpr (PROBE & 1) return;
        constexpr int CUR = decltype(cur)::value, J = decltype(jj)::value, M = J >> 2, W = J & 3;
        const float p0 = __builtin_amdgcn_exp2f(sacc[CUR][2 * J]);             // the accumulator holds score - m_run
        const float p1 = __builtin_amdgcn_exp2f(sacc[CUR][2 * J + 1]);
        unsigned hw;
        float d0 = 0.f, d1 = 0.f;
        if constexpr (P16) {
            // one round-to-nearest conversion (v_cvt_pk_f16_f32: unbiased also in the fp16 subnormal range, where a long tail of small
            // probabilities sits); the row sum takes exactly the values the matrix pipe will multiply — v_dot2_f32_f16 against (1, 1)
            // adds both halves of the packed pair to the fp32 sum in one instruction, subnormals included (tools/bench_src/denorm_probe.hip)
            typedef float f32x2p __attribute__((ext_vector_type(2)));
            const half2v hp = __builtin_convertvector(f32x2p{p0, p1}, half2v);
            hw = __builtin_bit_cast(unsigned, hp);
            const half2v ones = {(_Float16)1.f, (_Float16)1.f};
            if constexpr ((J & 1) == 0) l_a = __builtin_amdgcn_fdot2(hp, ones, l_a, false);
            else l_b = __builtin_amdgcn_fdot2(hp, ones, l_b, false);
        } else {
            l_a += p0;
            l_b += p1;
            split_rtz(p0, p1, hw, d0, d1);
        }
        u32x4 h4 = __builtin_bit_cast(u32x4, Ph[CUR][M]);
        h4[W] = hw;
        Ph[CUR][M] = __builtin_bit_cast(half8, h4);
        // two probabilities -> two e4m3 bytes of p8h / p8l (v_cvt_scalef32_pk_fp8_f32 converts x / scale into the low or, with op_sel[3],
        // the high half of its destination and keeps the other half).  Written as asm on the register itself: the builtin's merging
        // form lost the low-half conversions of three registers out of four to hipcc's dead code elimination (ROCm 7.2: only the
        // high-half writes were left in the loop).  s_nop: a VALU that reads a destination just written through op_sel / dst_sel
        // needs one wait state on this family (the tied operand of the high-half form reads it)
        constexpr int R = 4 * CUR + (J >> 1);
        int wh = p8h[R], wl = p8l[R];
        if constexpr ((J & 1) == 0) {
            asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(wh) : "v"(p0), "v"(p1), "s"(1.f / 64.f));
            if constexpr (!P16) asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(wl) : "v"(d0), "v"(d1), "s"(1.f / 65536.f));
        } else {
            asm("s_nop 0\n\tv_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(wh) : "v"(p0), "v"(p1), "s"(1.f / 64.f));
            if constexpr (!P16) asm("s_nop 0\n\tv_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(wl) : "v"(d0), "v"(d1), "s"(1.f / 65536.f));
        }
        p8h[R] = wh;
        if constexpr (!P16) p8l[R] = wl;
    };
#define PARQ_FENCE() do { if constexpr ((PROBE & 32) != 0) __builtin_amdgcn_sched_barrier(0); } while (0)
    // step n: QK(n + 1) -> sacc[NXT]; softmax(n) from sacc[CUR]; fp16 P V of block n - 1; odd n: at its end the fp8 cross terms of
    // the stage (n - 1, n), whose probabilities are complete by then (their V fragments are requested at the top of the step).
    // Fragments of the next step are requested as soon as their registers are dead (kf after the fourth fp16 Q K, k8 after the
    // second MX one, vh at the end).
    auto step = [&](auto cur, int n) {
        constexpr int CUR = decltype(cur)::value, NXT = CUR ^ 1;
        using IC = std::integral_constant<int, CUR>;
        const int nk = n + 2 < nbk ? n + 2 : nbk - 1;                       // K of the next step (clamped at the split's end)
#define PARQ_Q(i, Bq) if constexpr (!(PROBE & 4)) sacc[NXT] = mfma16<kF16>(kf[i], Bq, (i) == 0 ? negm16 : sacc[NXT]); PARQ_FENCE()
#define PARQ_P(D, Bp) if constexpr (!(PROBE & 2)) o[D] = mfma16<kF16>(vh[D], Bp, o[D]); PARQ_FENCE()
#define PARQ_S(J) sm_pair(IC{}, std::integral_constant<int, J>{}); PARQ_FENCE()
        i32x8 v8h[2], v8l[2];
        if constexpr (CUR == 1) { load_v8(n - 1, v8h, v8l); PARQ_FENCE(); }
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
        if constexpr (!(PROBE & 4)) sacc[NXT] = mx64(k8l, q8h, sacc[NXT], 0, kE8Lo, 0, kE8One);
        PARQ_FENCE();
        PARQ_S(4);
        PARQ_P(0, Ph[NXT][1]);  PARQ_S(5);
        if constexpr (!(PROBE & 4)) sacc[NXT] = mx64(k8h, q8l, sacc[NXT], 0, kE8One, 0, kE8Lo);
        PARQ_FENCE();
        load_k8(nk);
        PARQ_FENCE();
        PARQ_S(6);
        PARQ_P(1, Ph[NXT][1]);  PARQ_S(7);
        load_v(n, 0, vh);
        PARQ_FENCE();
        if constexpr (CUR == 1) {
            pvx(v8h, v8l);
            PARQ_FENCE();
            pscale = kPscale0;
            pend_d = 0;
        }
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
            const int di = (int)d;
            const float alpha = __builtin_ldexpf(1.f, -di);
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
            // block n's probabilities (relative to the old reference) still wait for their P V: fp16 ones times 2^-d ...
            const _Float16 ah = (_Float16)alpha;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) Ph[CUR][m][e] *= ah;
            // ... fp8 ones through the scale of their block: lanes kh = 0 scale the stage's first block, kh = 1 the second.  Only after
            // an even step does one wait (the first block; the second is taken at the new reference by the next step, and an odd step
            // has multiplied its stage already)
            if (CUR == 0 && kh == 0) {
                pend_d += di;
                const int eh = kE8Phi - pend_d, el = kE8Plo - pend_d;
                pscale = (eh > 0 ? eh : 0) | ((el > 0 ? el : 0) << 8);
            }
        }
    };

    if ((a.flags & 1) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    if (nbk > 0) {
        if (active) {
            // prologue: scores of block 0 against a zero reference, then the reference becomes their maximum
            load_k(0);
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) sacc[0] = mfma16<kF16>(kf[s], qh[s], s == 0 ? zero16 : sacc[0]);
            sacc[0] = mx64(k8l, q8h, sacc[0], 0, kE8Lo, 0, kE8One);
            sacc[0] = mx64(k8h, q8l, sacc[0], 0, kE8One, 0, kE8Lo);
            m_run = block_max(sacc[0]);
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[0][r] -= m_run; negm16[r] = -m_run; }
            load_k(1);                                                       // nbk >= 2
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
            // epilogue: softmax of the last block (odd), fp16 P V of the last two blocks, cross terms of the last stage
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float p[8], dl[8];
                unsigned hw[4];
#pragma unroll
                for (int e = 0; e < 8; ++e) p[e] = __builtin_amdgcn_exp2f(sacc[1][8 * m + e]);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    if constexpr (P16) {                                               // as sm_pair: one RNE conversion, summed as converted
                        typedef float f32x2p __attribute__((ext_vector_type(2)));
                        const half2v hp = __builtin_convertvector(f32x2p{p[e], p[e + 1]}, half2v);
                        hw[e >> 1] = __builtin_bit_cast(unsigned, hp);
                        l_run = __builtin_amdgcn_fdot2(hp, half2v{(_Float16)1.f, (_Float16)1.f}, l_run, false);
                        dl[e] = dl[e + 1] = 0.f;
                    } else {
                        split_rtz(p[e], p[e + 1], hw[e >> 1], dl[e], dl[e + 1]);
                        l_run += p[e] + p[e + 1];
                    }
                }
                Ph[1][m] = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    int a0, a1, b0, b1;
                    asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(p[4 * w]), "v"(p[4 * w + 1]), "s"(1.f / 64.f));
                    asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(a1) : "v"(p[4 * w + 2]), "v"(p[4 * w + 3]), "s"(1.f / 64.f));
                    asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(b0) : "v"(dl[4 * w]), "v"(dl[4 * w + 1]), "s"(1.f / 65536.f));
                    asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(b1) : "v"(dl[4 * w + 2]), "v"(dl[4 * w + 3]), "s"(1.f / 65536.f));
                    p8h[4 + 2 * m + w] = (int)__builtin_amdgcn_perm((unsigned)a1, (unsigned)a0, 0x05040100u);
                    if constexpr (!P16) p8l[4 + 2 * m + w] = (int)__builtin_amdgcn_perm((unsigned)b1, (unsigned)b0, 0x05040100u);
                }
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    half8 v2[2];
                    load_v(n - 1 + blk, m, v2);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) o[dt] = mfma16<kF16>(v2[dt], Ph[blk][m], o[dt]);
                }
            i32x8 v8h[2], v8l[2];
            load_v8(n - 1, v8h, v8l);
            pvx(v8h, v8l);
        }
    } else if (active) {
        m_run = -INFINITY;                                                 // a split without keys: weight 0 in the merge
    }

    if (active) {
        const int64_t pbase = (int64_t)bh * a.nsplit + split;
        float* op = a.o_part + pbase * kDH * Lq_pad;
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
                                   int64_t v_head, int64_t v_row, int B, int H, int N, void* cache, hipStream_t s, bool full) {
    if (!flash_split8_supported(kDH, N)) return hipErrorInvalidValue;
    if (full) hipLaunchKernelGGL(kvsplit8_convert_kernel<true>, dim3(N / 64, B * H), dim3(256), 0, s, K, V, k_batch, k_head, k_row, v_batch,
                                 v_head, v_row, H, N, reinterpret_cast<unsigned char*>(cache));
    else hipLaunchKernelGGL(kvsplit8_convert_kernel<false>, dim3(N / 64, B * H), dim3(256), 0, s, K, V, k_batch, k_head, k_row, v_batch,
                            v_head, v_row, H, N, reinterpret_cast<unsigned char*>(cache));
    return hipGetLastError();
}

hipError_t launch_flash_split8(const FlashArgs& a, const void* cache, hipStream_t s, bool p_lo) {
    if (!flash_split8_supported(a.dh, a.Lk) || a.nsplit < 1 || a.nsplit > 256 || a.drop_p > 0.f) return hipErrorInvalidValue;
    FlashArgs b = a;
    b.defer_log2 = kDefer8;
    static const int wt = [] { const char* e = dev_env("PARQ_FLASH_WT"); return e ? atoi(e) : 1; }();
    b.flags = ((a.flags & 2) ? 2 : 0) | ((wt && a.Lq % 256 == 0) ? 8 : 0);
    const dim3 grid(b.nsplit, ceil_div(b.Lq, 32 * kNW), b.B * b.H);
    const unsigned char* c8 = reinterpret_cast<const unsigned char*>(cache);
#define PARQ_F8_LAUNCH_RRP(PROBE, RING, REV, P16)                                                                              \
    {                                                                                                                          \
        static DynLdsOnce once;                                                                                                \
        const size_t lds = (size_t)(RING) * ((P16) ? kStage8Bytes : kStage8BytesFull);                                         \
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&flash_split8_kernel<PROBE, RING, REV, P16>), lds); e != hipSuccess) return e; \
        hipLaunchKernelGGL((flash_split8_kernel<PROBE, RING, REV, P16>), grid, dim3(kNW * 64), lds, s, b, c8);                  \
        return hipGetLastError();                                                                                              \
    }
#define PARQ_F8_LAUNCH_RR(PROBE, RING, REV) PARQ_F8_LAUNCH_RRP(PROBE, RING, REV, true)
#define PARQ_F8_LAUNCH_R(PROBE, RING)                                                                                          \
    {                                                                                                                          \
        if (b.flags & 2) PARQ_F8_LAUNCH_RR(PROBE, RING, true)                                                                  \
        PARQ_F8_LAUNCH_RR(PROBE, RING, false)                                                                                  \
    }
#define PARQ_F8_LAUNCH(PROBE) PARQ_F8_LAUNCH_R(PROBE, kRing)
#ifdef PARQ_DEV_PROBES
    static const int probe = [] { const char* e = dev_env("PARQ_FLASH_PROBE"); return e ? atoi(e) : 0; }();
    switch (probe) {
        case 0: break;
        case 1: PARQ_F8_LAUNCH(1)
        case 2: PARQ_F8_LAUNCH(2)
        case 4: PARQ_F8_LAUNCH(4)
        case 7: PARQ_F8_LAUNCH(7)
        case 8: PARQ_F8_LAUNCH(8)
        case 16: PARQ_F8_LAUNCH(16)
        case 23: PARQ_F8_LAUNCH(23)
        case 32: PARQ_F8_LAUNCH(32)
        default: return hipErrorInvalidValue;
    }
    const int ring = [] { const char* e = dev_env("PARQ_FLASH_RING"); return e && e[0] == '5' ? 5 : 4; }();
    if (ring == 5) PARQ_F8_LAUNCH_R(0, 5)
#endif
    if (p_lo) { if (b.flags & 2) PARQ_F8_LAUNCH_RRP(0, kRing, true, false) PARQ_F8_LAUNCH_RRP(0, kRing, false, false) }
    PARQ_F8_LAUNCH(0)
#undef PARQ_F8_LAUNCH
#undef PARQ_F8_LAUNCH_R
#undef PARQ_F8_LAUNCH_RR
#undef PARQ_F8_LAUNCH_RRP
}

}  // namespace parq
