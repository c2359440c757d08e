// Split-precision flash cross-attention for HEAD DIM 256 — the reference's shipped decoder size (config/train.yaml, eval.yaml:
// DEC_DIM 1024, 4 heads).  Same arithmetic as flash_split.hip (fp16 hi/lo operands, three 32x32x16 MFMAs per product, fp32
// accumulation: fp32-class accuracy), same "fragment-ready" K/V cache: a head of 256 dims is stored as 4 virtual heads of 64
// (chunk c = dims 64 c .. 64 c + 63), i.e. cache region (b, 4 h + c) holds the 16 KB blocks [K_hi | K_lo | V_hi | V_lo] of 32 keys of
// chunk c exactly as the head-dim-64 producers write them (the K/V projection is simply run with C / 64 heads).
//
// What changes is the register budget: 32 queries x 256 dims of output accumulators (128 VGPRs) plus Q fragments for a 256-deep
// contraction (128 VGPRs) do not fit a wave.  So the head dim is split over a PAIR of waves that own the same 32 queries:
//   wave (g, half): queries 32 g .. 32 g + 31 of the workgroup's 128, dims 128 half .. 128 half + 127
//     1. partial S^T over its 128 dims (24 MFMAs per 32-key block, Q fragments of two chunks in registers: 64 VGPRs)
//     2. the partners exchange the partial scores through LDS (4 KB per wave) and both form S = own + partner's
//     3. both run the (identical) online-softmax update — cheap next to the MFMAs at this head dim — and split P
//     4. O^T of its 128 dims += V^T P^T (24 MFMAs; 4 accumulators = 64 VGPRs)
// A workgroup is 4 such pairs = 128 queries; the two query tiles of a (scene, head, key split) get workgroup ids 8 k apart (same
// XCD, adjacent in dispatch order) so that the second one finds the K/V blocks in L2.
//
// LDS (all 160 KB): K ring 2 x 32 KB, V ring 2 x 32 KB ([chunk][hi | lo] images filled by LDS-DMA), exchange 32 KB.  One stage =
// one 32-key block.  K_{t+2} is requested at the mid-stage barrier (every wave is past QK(t)), V_{t+2} at the end-of-stage barrier
// (every wave is past PV(t)): each request has a full stage of compute to land.  VMEM operations retire in order, so the waits
// count the DMA groups issued after the one that is needed (4 instructions per thread and group).  The DMA is issued through asm
// and the barriers order LDS only (common.hpp: lds_dma16 / lds_barrier): with the builtin and __syncthreads() hipcc put
// s_waitcnt vmcnt(0) in front of the barriers and LDS reads, i.e. every stage waited for the requests it had just made.
#include "common.hpp"
#include <cstdlib>

namespace parq {

namespace {

constexpr int kDH = 256;
constexpr int kChunks = 4;                          // 64-dim chunks per head
constexpr int kNWconst = 8;
// TERMS = 3 (split precision): one 16 KB cache block of a chunk is [K_hi | K_lo | V_hi | V_lo] x 2048 halfs, K (or V) of a stage is
// [chunk][hi 2048 | lo 2048] halfs = 32 KB.  TERMS = 1 (single fp16 / bf16 products, KIND): blocks of 8 KB [K | V], 16 KB per stage.
template <int TERMS> struct Lay {
    static constexpr int sub = TERMS == 3 ? 8192 : 4096;          // halfs per cache block of a chunk
    static constexpr int xoff = TERMS == 3 ? 4096 : 2048;         // V offset inside a block
    static constexpr int img = TERMS == 3 ? 4096 : 2048;          // halfs of one chunk's K (or V) image in the ring
    static constexpr int grp = kChunks * img;                     // one ring slot
    static constexpr int ndma = grp * 2 / (kNWconst * 64 * 16);   // DMA instructions per thread and group (4 / 2)
};
constexpr int kNW = 8;
constexpr float kDeferLog2 = 10.f;

// DROP: training dropout on the probabilities (keep decision of element (row = scene-head * Lq + query, column = key), the stream
// the backward regenerates); the row sum l is taken before the mask, 1 / (1 - p) is applied to the output partials.
template <bool DROP, int TERMS, int KIND>
__global__ __launch_bounds__(kNW * 64) void flash_split256_kernel(FlashArgs a, const _Float16* __restrict__ cache) {
    typedef Lay<TERMS> L;
    constexpr int kGrpHalfs = L::grp, kSubHalfs = L::sub, kImgH = L::img;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* Kr = smem_h;                                   // [2][kGrpHalfs]
    _Float16* Vr = smem_h + 2 * kGrpHalfs;                   // [2][kGrpHalfs]
    float* Xb = reinterpret_cast<float*>(smem_h + 4 * kGrpHalfs);      // [8 waves][16 regs][64 lanes]

    const int split = blockIdx.x;
    const int bh = blockIdx.z;
    const int b = bh / a.H, h = bh - b * a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int grp = wave >> 1, half = wave & 1;
    const int q0 = (blockIdx.y * 4 + grp) * 32;
    const int q = q0 + li;
    const bool active = q0 < a.Lq;                           // wave-uniform; inactive waves still take part in staging and barriers
    const int Lq_pad = (a.Lq + 31) & ~31;

    // Q fragments of this wave's two chunks (B operand of S^T = K Q^T), pre-scaled by log2(e) / sqrt(256), split hi/lo
    half8 qhi[2][4], qlo[2][4];
    {
        const float scale = 1.4426950408889634f / 16.f;
        const float* qp = a.q + (int64_t)b * a.q_batch + (int64_t)h * a.q_head + (int64_t)(q < a.Lq ? q : 0) * a.q_row + 128 * half;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int d0 = 64 * cc + 32 * (s >> 1) + 16 * (s & 1) + 4 * kh;       // dmap(kh, s, e) of flash_split.hip inside the chunk
                f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + d0);
                f32x4 x1 = *reinterpret_cast<const f32x4*>(qp + d0 + 8);
                if (q >= a.Lq) { x0 = f32x4{0.f, 0.f, 0.f, 0.f}; x1 = x0; }
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (e < 4 ? x0[e & 3] : x1[e & 3]) * scale;
                if constexpr (TERMS == 3) split8(x, qhi[cc][s], qlo[cc][s]);
                else { qhi[cc][s] = cvt8_rn<KIND>(x); qlo[cc][s] = qhi[cc][s]; }
            }
    }

    const int nblk = (a.Lk + 31) / 32;
    const int t_begin = (int)((int64_t)split * nblk / a.nsplit);
    const int t_end = (int)((int64_t)(split + 1) * nblk / a.nsplit);
    // chunk c of this head is virtual head 4 h + c of the cache
    const _Float16* cbase = cache + ((int64_t)(b * a.H + h) * kChunks) * nblk * kSubHalfs;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    // one DMA group = K (which = 0) or V (which = 1) of block t: instruction c copies the 8 KB [hi | lo] of chunk c
    auto gload = [&](int t, int which) {
        _Float16* ring = which ? Vr : Kr;
        const unsigned dst = (unsigned)(size_t)(lds_byte*)(ring + ((t - t_begin) & 1) * kGrpHalfs);
        if constexpr (TERMS == 3) {
            // instruction c copies the 8 KB [hi | lo] of chunk c
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const _Float16* src = cbase + ((int64_t)c * nblk + t) * kSubHalfs + which * L::xoff + tid * 8;
                lds_dma16(src, dst + (c * 512 + wave * 64) * 16);      // asm-issued: hipcc must not answer with vmcnt(0) before LDS reads
            }
        } else {
            // instruction i copies the 4 KB images of chunks 2 i and 2 i + 1 (256 threads each)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = 2 * i + (tid >> 8), piece = tid & 255;
                const _Float16* src = cbase + ((int64_t)c * nblk + t) * kSubHalfs + which * L::xoff + piece * 8;
                lds_dma16(src, dst + (c * 256 + (piece >> 6) * 64) * 16);
            }
        }
    };
    auto wait_groups = [&](int n) {                           // at most n DMA groups (L::ndma instructions each) still in flight
        if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::ndma) : "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L::ndma) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * L::ndma) : "memory");
    };

    f32x16 o[4];                                             // O^T blocks: dims 128 half + 32 j + mfma32_row(r), column = this lane's query
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const uint32_t drop_row = DROP ? drop_rowhash(a.drop_seed, (uint32_t)(bh * a.Lq + (q < a.Lq ? q : 0))) : 0u;
    const uint32_t drop_thr = DROP ? drop_threshold(a.drop_p) : 0u;

    // issue order: K_t0, V_t0, K_t0+1, V_t0+1, then per stage K_{t+2} (mid barrier), V_{t+2} (end barrier)
    if (t_begin < t_end) { gload(t_begin, 0); gload(t_begin, 1); }
    if (t_begin + 1 < t_end) { gload(t_begin + 1, 0); gload(t_begin + 1, 1); }
    // K of the first block must be complete and visible: groups issued after it = V_t0 (+ K, V of the second block)
    wait_groups(t_begin + 1 < t_end ? 3 : 1);
    lds_barrier();

    const int ksw = (li >> 1) & 7;
    for (int t = t_begin; t < t_end; ++t) {
        const int slot = (t - t_begin) & 1;
        const _Float16* Ks = Kr + slot * kGrpHalfs + (2 * half) * kImgH;         // this wave's two chunks: [cc][hi | lo]
        const _Float16* Vs = Vr + slot * kGrpHalfs + (2 * half) * kImgH;
        const bool more1 = t + 1 < t_end, more2 = t + 2 < t_end;
        const uint32_t drop_blk = DROP ? drop_row ^ drop_blockhash((uint32_t)t) ^ (kh ? kDropBit2Part : 0u) : 0u;   // common.hpp: column hash parts

        // ---- partial S^T over this wave's 128 dims
        f32x16 sacc;
        if (active) {
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            sacc = zero16;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int pos = (4 * kh + s) ^ ksw;
                    const half8 kfh = *reinterpret_cast<const half8*>(Ks + cc * kImgH + li * 64 + pos * 8);
                    sacc = mfma16<KIND>(kfh, qhi[cc][s], sacc);
                    if constexpr (TERMS == 3) {
                        const half8 kfl = *reinterpret_cast<const half8*>(Ks + cc * kImgH + 2048 + li * 64 + pos * 8);
                        sacc = mfma16<KIND>(kfh, qlo[cc][s], sacc);
                        sacc = mfma16<KIND>(kfl, qhi[cc][s], sacc);
                    }
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) Xb[(wave * 16 + r) * 64 + lane] = sacc[r];
        }
        // V_t must be complete before the barrier that publishes it: groups issued after V_t are K_{t+1}, V_{t+1}
        wait_groups(more1 ? 2 : 0);
        lds_barrier();                                       // mid-stage: partial scores exchanged, K slot of block t free, V_t visible
        if (more2) gload(t + 2, 0);
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] += Xb[((wave ^ 1) * 16 + r) * 64 + lane];
            if (t == nblk - 1 && (a.Lk & 31) != 0) {         // keys past Lk in the last block
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 32 + mfma32_row(r, lane) >= a.Lk) sacc[r] = -INFINITY;
            }
            // ---- online softmax with a deferred running maximum (log2 domain), identical in both partner waves
            float mx = sacc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (__any(mx > m_run + kDeferLog2)) {
                const bool need = mx > m_run + kDeferLog2;
                const float m_new = need ? mx : m_run;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
            }
            half8 phi[2], plo[2];
            float rs = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float p[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    p[e] = __builtin_amdgcn_exp2f(sacc[8 * m + e] - m_run);
                    rs += p[e];
                    if constexpr (DROP) {
                        if (!drop_keep_h(drop_blk, drop_regpart(8 * m + e), drop_thr)) p[e] = 0.f;     // = drop_colhash(t * 32 + mfma32_row(8 m + e, lane))
                    }
                }
                if constexpr (TERMS == 3) split8(p, phi[m], plo[m]);
                else { phi[m] = cvt8_rn<KIND>(p); plo[m] = phi[m]; }
            }
            rs += __shfl_xor(rs, 32);
            l_run += rs;
            // ---- O^T of this wave's 128 dims += V^T P^T
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cc = j >> 1, d = (j & 1) * 32 + li;
                    const int pos = (2 * m + kh) ^ ((d >> 2) & 3);
                    const half8 vh = *reinterpret_cast<const half8*>(Vs + cc * kImgH + d * 32 + pos * 8);
                    o[j] = mfma16<KIND>(vh, phi[m], o[j]);
                    if constexpr (TERMS == 3) {
                        const half8 vl = *reinterpret_cast<const half8*>(Vs + cc * kImgH + 2048 + d * 32 + pos * 8);
                        o[j] = mfma16<KIND>(vl, phi[m], o[j]);
                        o[j] = mfma16<KIND>(vh, plo[m], o[j]);
                    }
                }
        }
        // K_{t+1} must be complete before the barrier that publishes it: groups issued after it are V_{t+1}, K_{t+2}
        if (more1) wait_groups(more2 ? 2 : 1);
        lds_barrier();                                       // end of stage: V slot of block t and the exchange area are free, K_{t+1} visible
        if (more2) gload(t + 2, 1);
    }

    if (active) {
        const int64_t pbase = (int64_t)bh * a.nsplit + split;
        float* op = a.o_part + pbase * kDH * Lq_pad;
        const float drop_scale = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) op[(int64_t)(128 * half + j * 32 + mfma32_row(r, lane)) * Lq_pad + q] = o[j][r] * drop_scale;
        if (half == 0 && kh == 0) {
            a.m_part[pbase * Lq_pad + q] = m_run;
            a.l_part[pbase * Lq_pad + q] = l_run;
        }
    }
}

}  // namespace

int flash_split256_pick_splits(int B, int H, int Lq, int Lk, int num_cus) {
    const int nblk = ceil_div(Lk, 32);
    const int64_t base = (int64_t)B * H * ceil_div(Lq, 128);
    int64_t want = ceil_div64((int64_t)num_cus, base);
    if (want < 1) want = 1;
    if (want > nblk) want = nblk;
    if (want > 256) want = 256;
    // keep the workgroup ids of the query tiles of one (split, scene-head) a multiple of 8 apart (same XCD: shared K/V in L2)
    if (want >= 8) want = (want / 8) * 8;
    return (int)want;
}

// partial (O, m, l) of every (scene-head, key split) in the layout flash_merge_kernel<256> combines; cache: virtual-head split cache
template <int TERMS, int KIND>
static hipError_t launch_fs256(const FlashArgs& a, const void* cache, hipStream_t s) {
    static DynLdsOnce once, once_drop;
    const size_t lds = (size_t)4 * Lay<TERMS>::grp * sizeof(_Float16) + (size_t)kNW * 16 * 64 * sizeof(float);       // 160 KB (96 KB single-term)
    dim3 grid(a.nsplit, ceil_div(a.Lq, 128), a.B * a.H);
    if (a.drop_p > 0.f) {
        if (hipError_t e = once_drop.ensure(reinterpret_cast<const void*>(&flash_split256_kernel<true, TERMS, KIND>), lds); e != hipSuccess) return e;
        hipLaunchKernelGGL((flash_split256_kernel<true, TERMS, KIND>), grid, dim3(kNW * 64), lds, s, a, reinterpret_cast<const _Float16*>(cache));
    } else {
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&flash_split256_kernel<false, TERMS, KIND>), lds); e != hipSuccess) return e;
        hipLaunchKernelGGL((flash_split256_kernel<false, TERMS, KIND>), grid, dim3(kNW * 64), lds, s, a, reinterpret_cast<const _Float16*>(cache));
    }
    return hipGetLastError();
}

hipError_t launch_flash_split256(const FlashArgs& a, const void* cache, hipStream_t s, int terms, int kind) {
    if (a.dh != kDH || a.nsplit < 1 || a.nsplit > 256) return hipErrorInvalidValue;
    if (terms == 3) return launch_fs256<3, kF16>(a, cache, s);
    return kind == kF16 ? launch_fs256<1, kF16>(a, cache, s) : launch_fs256<1, kBF16>(a, cache, s);
}

}  // namespace parq
