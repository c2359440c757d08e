// fp32 flash attention for the PARQ decoder (gfx950): dense cross-attention of Q queries
// against all N = V*h*w memory tokens (transformer_parq.py:377-380) and the small
// query self-attention (:372-376), without materialising the (H,Q,N) score tensor.
//
// Mapping onto CDNA4
//   * v_mfma_f32_32x32x2_f32 (exact fp32).  Everything is kept in the "query = lane & 31"
//     layout:   S^T = K Q^T   (A = K[key][d] from LDS, B = Q[q][d] from registers)
//               O^T = V^T P^T (A = V[key][d] from LDS, B = P[q][key] = the S^T registers)
//     The C/D layout of S^T (lane (q, half) holds keys (r&3)+8(r>>2)+4*half) is exactly a
//     legal B-operand enumeration of the key contraction index, so the probabilities feed
//     the second MFMA straight from registers — no cross-lane traffic, and softmax
//     statistics are per-lane scalars (one shuffle joins the two lane halves).
//   * The contraction over d is permuted so lane half kh owns d in [kh*dh/2, (kh+1)*dh/2):
//     one ds_read_b128 of a K row feeds 4 MFMAs.  K tiles are XOR-swizzled in LDS
//     (16-byte chunk index ^ row) so those reads are bank-conflict free; V tiles are read
//     row-contiguously with ds_read_b32 (conflict free as is).
//   * One workgroup = NW waves x 32 queries, streaming its key range through a
//     double-buffered LDS tile (register-staged prefetch: loads for tile t+1 are issued
//     before the MFMAs of tile t, written to the other buffer after them; one barrier/tile).
//   * grid = (key splits, query tiles, B*H): splits give >= 1-2 workgroups per CU even
//     for B = 1; partial (O, m, l) are combined by flash_merge_kernel.
#include "common.hpp"
#include <cstdlib>

namespace parq {

namespace {

template <int DH>
struct Tile {
    static constexpr int KT = DH <= 64 ? 64 : 32;           // keys per LDS tile
    static constexpr int ROW4 = DH / 4;                     // float4 per row
    static constexpr int SWZ = (ROW4 < 16 ? ROW4 : 16) - 1; // XOR mask on the chunk index
    static constexpr size_t lds_bytes() { return (size_t)2 * 2 * KT * DH * sizeof(float); }
};

template <int DH, int NW>
__global__ __launch_bounds__(NW * 64) void flash_f32_kernel(FlashArgs a) {
    using T = Tile<DH>;
    constexpr int KT = T::KT;
    constexpr int NT = NW * 64;
    constexpr int ROW4 = T::ROW4;
    constexpr int TILE4 = KT * ROW4;
    constexpr int LD4 = TILE4 / NT;
    static_assert(TILE4 % NT == 0, "tile must divide evenly over the workgroup");
    constexpr int NDT = DH / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                    // [2][KT*DH] swizzled
    float* Vs = smem + 2 * KT * DH;      // [2][KT*DH]

    const int split = blockIdx.x;
    const int bh = blockIdx.z;
    const int b = bh / a.H;
    const int h = bh - b * a.H;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;
    const int q0 = (blockIdx.y * NW + wave) * 32;
    const int q = q0 + li;
    const bool active = q0 < a.Lq;       // wave-uniform
    const int Lq_pad = (a.Lq + 31) & ~31;

    // Q fragment (B operand): lane (kh, j) holds Q[q0+j][kh*DH/2 + t], pre-scaled by log2(e)/sqrt(dh)
    float qf[DH / 2];
    {
        const float scale = 1.4426950408889634f / sqrtf((float)DH);
        const float* qp = a.q + (int64_t)b * a.q_batch + (int64_t)h * a.q_head +
                          (int64_t)(q < a.Lq ? q : 0) * a.q_row + kh * (DH / 2);
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            f32x4 t4 = *reinterpret_cast<const f32x4*>(qp + c * 4);
            if (q >= a.Lq) t4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[c * 4 + e] = t4[e] * scale;
        }
    }

    const int nt = (a.Lk + KT - 1) / KT;
    const int t_begin = (int)((int64_t)split * nt / a.nsplit);
    const int t_end = (int)((int64_t)(split + 1) * nt / a.nsplit);

    const float* kbase = a.k + (int64_t)b * a.k_batch + (int64_t)h * a.k_head;
    const float* vbase = a.v + (int64_t)b * a.v_batch + (int64_t)h * a.v_head;

    f32x4 kreg[LD4], vreg[LD4];
    auto gload = [&](int tile) {
#pragma unroll
        for (int i = 0; i < LD4; ++i) {
            const int f = tid + i * NT;
            const int row = f / ROW4;
            const int ch = f - row * ROW4;
            const int key = tile * KT + row;
            if (key < a.Lk) {
                kreg[i] = *reinterpret_cast<const f32x4*>(kbase + (int64_t)key * a.k_row + ch * 4);
                vreg[i] = *reinterpret_cast<const f32x4*>(vbase + (int64_t)key * a.v_row + ch * 4);
            } else {
                kreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LD4; ++i) {
            const int f = tid + i * NT;
            const int row = f / ROW4;
            const int ch = f - row * ROW4;
            *reinterpret_cast<f32x4*>(Ks + buf * KT * DH + row * DH + ((ch ^ (row & T::SWZ)) * 4)) = kreg[i];
            *reinterpret_cast<f32x4*>(Vs + buf * KT * DH + row * DH + ch * 4) = vreg[i];
        }
    };

    f32x16 o[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY;
    float l_run = 0.f;

    if (t_begin < t_end) {
        gload(t_begin);
        swrite(0);
    }
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        const bool more = t + 1 < t_end;
        if (more) gload(t + 1);

        if (active) {
            const float* Kt = Ks + buf * KT * DH;
            const float* Vt = Vs + buf * KT * DH;
            const bool tail = (t == nt - 1) && (a.Lk % KT != 0);
#pragma unroll
            for (int kb = 0; kb < KT / 32; ++kb) {
                // ---- S^T = K Q^T  (32 keys x 32 queries, contraction over DH)
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
                const int krow = kb * 32 + li;
                const float* kr = Kt + krow * DH;
                const int sw = krow & T::SWZ;
#pragma unroll
                for (int u = 0; u < DH / 8; ++u) {
                    const int ch = kh * (DH / 8) + u;
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(kr + ((ch ^ sw) * 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[u * 4 + e], sacc, 0, 0, 0);
                }
                if (tail) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * KT + kb * 32 + mfma32_row(r, lane);
                        if (key >= a.Lk) sacc[r] = -INFINITY;
                    }
                }
                // ---- online softmax (log2 domain); lane pair (l, l^32) shares one query
                float mx = sacc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                float rs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] = __builtin_amdgcn_exp2f(sacc[r] - m_new);
                    rs += sacc[r];
                }
                rs += __shfl_xor(rs, 32);
                l_run = l_run * alpha + rs;
                m_run = m_new;
                if (a.drop_p > 0.f) {           // training: dropout on the probabilities; the normaliser stays undropped
                    const float inv = 1.f / (1.f - a.drop_p);
                    const uint32_t rh = drop_rowhash(a.drop_seed, (uint32_t)(bh * a.Lq + q));
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sacc[r] = drop_keep(rh, (uint32_t)(t * KT + kb * 32 + mfma32_row(r, lane)), a.drop_p) ? sacc[r] * inv : 0.f;
                }
#pragma unroll
                for (int d = 0; d < NDT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                // ---- O^T += V^T P^T
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int vrow = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float* vr = Vt + vrow * DH + li;
#pragma unroll
                    for (int d = 0; d < NDT; ++d)
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[d * 32], sacc[r], o[d], 0, 0, 0);
                }
            }
        }
        if (more) swrite(buf ^ 1);
        __syncthreads();
    }

    if (active) {
        const int64_t pbase = (int64_t)bh * a.nsplit + split;
        float* op = a.o_part + pbase * DH * Lq_pad;
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) op[(int64_t)(d * 32 + mfma32_row(r, lane)) * Lq_pad + q] = o[d][r];
        if (kh == 0) {
            a.m_part[pbase * Lq_pad + q] = m_run;
            a.l_part[pbase * Lq_pad + q] = l_run;
        }
    }
}

// Combine the key-split partials:  out[b][q][h*DH+d] = sum_s w_s O_s[d][q] / sum_s w_s l_s,
// w_s = 2^(m_s - max_s m_s).  The partials are tens of MB and every output needs one value from each of
// the S splits, so the kernel is built for bytes in flight: one workgroup per (32 queries, b*h, 16 d-rows)
// (>= 128 workgroups at B = 1), each thread streams 16-byte rows of 4 queries for one d over half of the
// splits, 8 loads deep; halves are reduced and the tile transposed through LDS so the output rows
// (d-contiguous) are written coalesced.
// DG: head dims per workgroup (grid.z = DH / DG).  16 -> 2 split groups per (query, d); 8 -> 4 groups and twice the workgroups:
// at one scene (4 heads x 8 query tiles) 256 workgroups instead of 128, i.e. the whole chip for this latency-bound pass.
template <int DH, int DG>
__global__ __launch_bounds__(256) void flash_merge_kernel(FlashArgs a) {
    PARQ_TL_KERNEL(kTlFlashMerge);
    constexpr int kMergeDG = DG;
    constexpr int NG = 256 / (8 * DG);               // split groups
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wsm = smem;                               // [nsplit][32]
    float* dsm = wsm + a.nsplit * 32;                // [8][32]
    float* part = dsm + 8 * 32;                      // [NG split groups][kMergeDG][32]
    const FlashHead fh = flash_head(a, blockIdx.y);
    const int bh = fh.bh, b = fh.b, h = fh.h;
    const int q0 = blockIdx.x * 32;
    const int dg0 = blockIdx.z * kMergeDG;
    const int tq = threadIdx.x & 31;
    const int td = threadIdx.x >> 5;
    const int Lq_pad = (a.Lq + 31) & ~31;
    const int q = q0 + tq;
    const int64_t pb = (int64_t)blockIdx.y * a.nsplit;

    // softmax statistics of the 32 queries: thread (tq, td) owns splits td, td+8, ...; all of its loads are
    // issued together (a one-load-per-iteration loop here costs nsplit dependent L2 round trips)
    constexpr int kMaxPer = 32;                      // nsplit <= 256
    float mloc[kMaxPer], lloc[kMaxPer];
    float mmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
        const int s = td + i * 8;
        if (i * 8 < a.nsplit) {                      // uniform per i
            mloc[i] = s < a.nsplit ? a.m_part[(pb + s) * Lq_pad + q] : -INFINITY;
            lloc[i] = s < a.nsplit ? a.l_part[(pb + s) * Lq_pad + q] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i)
        if (i * 8 < a.nsplit) mmax = fmaxf(mmax, mloc[i]);
    dsm[td * 32 + tq] = mmax;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) mmax = fmaxf(mmax, dsm[i * 32 + tq]);
    __syncthreads();
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) {
        const int s = td + i * 8;
        if (i * 8 < a.nsplit && s < a.nsplit) {
            const float w = __builtin_amdgcn_exp2f(mloc[i] - mmax);
            wsm[s * 32 + tq] = w;
            den += w * lloc[i];
        }
    }
    dsm[td * 32 + tq] = den;
    __syncthreads();
    den = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) den += dsm[i * 32 + tq];
    const float inv = 1.f / den;
    if (a.lse && blockIdx.z == 0 && td == 0 && q < a.Lq) a.lse[(int64_t)bh * Lq_pad + q] = mmax + log2f(den);
    if ((a.peaky || a.head_min) && blockIdx.z == 0 && td == 0) {     // attention mode 4: a row carried by too few keys (FlashArgs)
        if (a.peaky && __any(q < a.Lq && den < a.peaky_l) && tq == 0) {
            atomicOr(a.peaky, 1 << h);
            if (a.peaky_it) atomicOr(a.peaky_it, 1 << h);
        }
        if (a.peaky_min || a.head_min) {
            float dmin = q < a.Lq ? den : INFINITY;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o));
            if (tq == 0 && a.peaky_min) atomicMax(a.peaky_min, 0x7fffffff - __float_as_int(dmin));
            if (tq == 0 && a.head_min) atomicMax(a.head_min + h, 0x7fffffff - __float_as_int(dmin));
        }
    }

    // thread -> (4 queries, one d, one of NG groups of the splits)
    const int q4 = threadIdx.x & 7;
    const int dd = (threadIdx.x >> 3) & (DG - 1);
    const int half = threadIdx.x / (8 * DG);
    const int s_begin = (int)((int64_t)half * a.nsplit / NG);
    const int s_end = (int)((int64_t)(half + 1) * a.nsplit / NG);
    const int64_t sstride = (int64_t)DH * Lq_pad;
    const float* o0 = a.o_part + (pb * DH + dg0 + dd) * (int64_t)Lq_pad + q0 + q4 * 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int s = s_begin;
    for (; s + 8 <= s_end; s += 8) {
        f32x4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const f32x4*>(o0 + (s + u) * sstride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u] * *reinterpret_cast<const f32x4*>(&wsm[(s + u) * 32 + q4 * 4]);
    }
    for (; s < s_end; ++s)
        acc += *reinterpret_cast<const f32x4*>(o0 + s * sstride) * *reinterpret_cast<const f32x4*>(&wsm[s * 32 + q4 * 4]);
    *reinterpret_cast<f32x4*>(&part[(half * kMergeDG + dd) * 32 + q4 * 4]) = acc;
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * kMergeDG; idx += 256) {
        const int qq = idx / kMergeDG;
        const int d = idx - qq * kMergeDG;
        if (q0 + qq < a.Lq) {
            float dn = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) dn += dsm[i * 32 + qq];
            float acc2 = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < NG; ++g2) acc2 += part[(g2 * kMergeDG + d) * 32 + qq];
            a.out[(int64_t)b * a.out_batch + (int64_t)(q0 + qq) * a.out_row + h * DH + dg0 + d] = acc2 / dn;
        }
    }
    (void)inv;
}

// Self-attention among the Q queries (transformer_parq.py:372-376) in ONE launch: a workgroup owns
// 16 queries of one (scene, head); its 8 waves take disjoint key slices, run an fp32-MFMA online
// softmax and combine their (m, l, O^T) through LDS.  The kernel is pure latency, so it is shaped
// for short dependent chains and cheap loads:
//   * v_mfma_f32_16x16x4_f32, lane (j = l&15, kq = l>>4).  S^T = K Q^T with A = K, B = Q^T: both
//     operands are float4 loads at column c*16 + kq*4 of row j (16 rows x 64 B per instruction,
//     16 requests for the L1 pipe instead of 64 for a row-per-lane layout), the contraction index
//     is permuted identically on both sides;
//   * the S^T accumulator of key sub-tile s holds keys s*16 + 4*kq + r for query j, which is
//     exactly the B operand P^T[key][q] of O^T = V^T P^T when MFMA step (s, r) contracts over
//     key = s*16 + 4*kq + r: probabilities never leave registers.  Its A operand V[key][d = 16 dt + j]
//     is a lane-contiguous scalar load.
template <int DH, int NW = 8>
__global__ __launch_bounds__(NW * 64) void self_attn_kernel(const float* __restrict__ qkv, int64_t row_stride, int H, int L,
                                                        float* __restrict__ out, int64_t out_row, float* __restrict__ lse,
                                                        float drop_p, uint32_t drop_seed) {
    PARQ_TL_KERNEL(kTlSelfAttn);
    constexpr int NDT = DH / 16;                 // 16-wide d sub-tiles of O^T
    constexpr int NC = DH / 16;                  // float4 chunks of a Q / K row per lane
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Os = smem;                         // [NW][DH][17]
    float* Ms = Os + NW * DH * 17;            // [NW][16]
    float* Ls = Ms + NW * 16;                 // [NW][16]
    const int bh = blockIdx.y;
    const int b = bh / H, h = bh - b * H;
    const int C = H * DH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, kq = lane >> 4;
    const int q0 = blockIdx.x * 16;
    const int q = q0 + lj;
    const float* base = qkv + (int64_t)b * L * row_stride + h * DH;     // q at +0, k at +C, v at +2C

    f32x4v qf[NC];
    {
        const float* qp = base + (int64_t)(q < L ? q : 0) * row_stride + kq * 4;
#pragma unroll
        for (int c = 0; c < NC; ++c) qf[c] = *reinterpret_cast<const f32x4v*>(qp + c * 16);
    }
    const int per = ((((L + NW - 1) / NW) + 31) / 32) * 32;              // keys per wave, multiple of 32
    const int k_begin = wave * per;
    const int k_end = (k_begin + per < L) ? k_begin + per : L;
    const float scale = 1.4426950408889634f / sqrtf((float)DH);

    f32x4v o[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d) o[d] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    bool q_scaled = false;

    // K fragments of a 32-key block: row j = key, float4 chunks of the head dim
    constexpr bool kV4 = (NDT % 4) == 0;
    // Head dim 64 (round 6): K and V come in as WHOLE-KILOBYTE requests (a request = four key rows x 256 bytes) and reach the MFMA operand
    // layout through a wave-private LDS tile of 16 keys — the operand pattern itself (16 rows x 64 bytes per request) costs several times
    // more per byte (tools/bench_src/row_stride_loads.hip).  A block's V is requested behind its K staging (it lands behind the scores) and
    // the next block's K behind this block's V staging (it lands behind the P V products), so only the first K pays the round trip.  Same
    // products in the same order: the results are those of the direct loads bit for bit (output digests equal, tools/r06_fwd_time.py;
    // 1.396 -> 1.388 ms per cfg-3 forward on one box).  Keys past L read row 0: their probabilities are exactly 0 (score -inf).
    constexpr bool kStage = DH == 64;
    // head dims >= 128 (4 waves, 2+ blocks per wave at Q = 256) keep the direct loads, pipelined: the next block's K is requested as soon as
    // this block's scores are done and its V as soon as this block's P V products are issued (25.6 -> 23.4 us at head dim 256; the staged
    // form measured 1.959 -> 1.970 ms per shipped-size forward there: one sub-block at a time costs more than the requests save)
    constexpr bool kPipe = DH >= 128;
    constexpr int RQ = DH / 16;                    // float4 requests per lane and 16-key sub-block (K or V)
    constexpr int KP = DH + 16, VP = DH + 4;       // row pitch (floats) of the staging tile as K / as V (fragment reads of 16 rows / of rows 4 apart spread over the banks)
    float* const stg = Ls + NW * 16 + wave * (16 * KP);
    f32x4v kf[2][NC];
    float vv[2][4][NDT];
    constexpr int RQA = kStage ? RQ : 1, S2 = kStage ? 2 : 1;
    f32x4v kr[S2][RQA], vr[S2][RQA];
    auto graw = [&](f32x4v (&dst)[S2][RQA], int kb, int col0) {
        if constexpr (kStage)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < RQ; ++i) {
                const int f = i * 64 + lane, row = f / (DH / 4), c4 = f % (DH / 4);
                const int key = kb + s * 16 + row;
                dst[s][i] = *reinterpret_cast<const f32x4v*>(base + col0 + (int64_t)(key < L ? key : 0) * row_stride + c4 * 4);
            }
    };
    auto to_lds = [&](const f32x4v (&src)[RQA], int pitch) {
#pragma unroll
        for (int i = 0; i < RQA; ++i) {
            const int f = i * 64 + lane, row = f / (DH / 4), c4 = f % (DH / 4);
            *reinterpret_cast<f32x4v*>(stg + row * pitch + c4 * 4) = src[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave's own LDS writes have landed (no other wave touches the tile)
    };
    auto stage_k = [&](int s) {
        to_lds(kr[s < S2 ? s : 0], KP);
#pragma unroll
        for (int c = 0; c < NC; ++c) kf[0][c] = *reinterpret_cast<const f32x4v*>(stg + lj * KP + c * 16 + kq * 4);
        asm volatile("" ::: "memory");
    };
    auto stage_v = [&](int s) {
        to_lds(vr[s < S2 ? s : 0], VP);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int d4 = 0; d4 < NDT / 4; ++d4) {
                const f32x4v v4 = *reinterpret_cast<const f32x4v*>(stg + (4 * kq + r) * VP + NDT * lj + 4 * d4);
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[0][r][4 * d4 + e] = v4[e];
            }
        asm volatile("" ::: "memory");
    };
    auto load_k = [&](int kb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int key = kb + s * 16 + lj;
            const float* kr = base + C + (int64_t)(key < L ? key : 0) * row_stride + kq * 4;
#pragma unroll
            for (int c = 0; c < NC; ++c) kf[s][c] = *reinterpret_cast<const f32x4v*>(kr + c * 16);
        }
    };
    // V^T as the A operand of O^T += V^T P^T: MFMA row i = lj is head dim kVD(lj, dt) of sub-tile dt.  For head dims that are
    // multiples of 64 the rows are dealt so that a lane's NDT values are CONSECUTIVE dims (d = NDT lj + dt): NDT / 4 float4 loads
    // per key instead of NDT scalar ones (the output dim order is a free permutation, undone when O^T goes to LDS).  Keys past L
    // read row 0: their probabilities are exactly 0 (score -inf), in the pipelined form they are not zeroed a second time
    auto load_v = [&](int kb) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int vk = kb + s * 16 + 4 * kq + r;
                if constexpr (kV4) {
                    const float* vr = base + 2 * C + (int64_t)(vk < L ? vk : 0) * row_stride + NDT * lj;
#pragma unroll
                    for (int d4 = 0; d4 < NDT / 4; ++d4) {
                        const f32x4v v4 = *reinterpret_cast<const f32x4v*>(vr + 4 * d4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[s][r][4 * d4 + e] = (kPipe || vk < L) ? v4[e] : 0.f;
                    }
                } else {
                    const float* vr = base + 2 * C + (int64_t)(vk < L ? vk : 0) * row_stride + lj;
#pragma unroll
                    for (int d = 0; d < NDT; ++d) vv[s][r][d] = (kPipe || vk < L) ? vr[d * 16] : 0.f;
                }
            }
    };
    if constexpr (kStage) {
        if (k_begin < k_end) graw(kr, k_begin, C);
    }
    if constexpr (kPipe) {
        if (k_begin < k_end) { load_k(k_begin); load_v(k_begin); }
    }
    for (int kb = k_begin; kb < k_end; kb += 32) {
        if (!q_scaled) {                          // scores in the log2 domain
#pragma unroll
            for (int c = 0; c < NC; ++c) qf[c] *= scale;
            q_scaled = true;
        }
        f32x4v sacc[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) sacc[s] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if constexpr (kStage) {
            // one 16-key sub-block at a time through the staging tile (the fragment registers of one sub-block, not two); each
            // accumulator sees its products in the order of the direct form
            stage_k(0);
            graw(vr, kb, 2 * C);                  // this block's V lands behind its scores, the next block's K behind its P V products
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0][c][e], qf[c][e], sacc[0], 0, 0, 0);
            stage_k(1);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0][c][e], qf[c][e], sacc[1], 0, 0, 0);
        } else {
            if constexpr (!kPipe) { load_k(kb); load_v(kb); }      // head dim 32: all loads of this 32-key block up front
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        sacc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s][c][e], qf[c][e], sacc[s], 0, 0, 0);
            if constexpr (kPipe) {
                if (kb + 32 < k_end) load_k(kb + 32);
            }
        }
        if (kb + 32 > L) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb + s * 16 + 4 * kq + r >= L) sacc[s][r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(sacc[0][0], sacc[0][1]), fmaxf(sacc[0][2], sacc[0][3])),
                         fmaxf(fmaxf(sacc[1][0], sacc[1][1]), fmaxf(sacc[1][2], sacc[1][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sacc[s][r] = __builtin_amdgcn_exp2f(sacc[s][r] - m_new);
                rs += sacc[s][r];
            }
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        if (drop_p > 0.f) {
            const float inv = 1.f / (1.f - drop_p);
            const uint32_t rh = drop_rowhash(drop_seed, (uint32_t)(bh * L + q));
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sacc[s][r] = drop_keep(rh, (uint32_t)(kb + s * 16 + 4 * kq + r), drop_p) ? sacc[s][r] * inv : 0.f;
        }
#pragma unroll
        for (int d = 0; d < NDT; ++d) o[d] *= alpha;
        if constexpr (kStage) {
            stage_v(0);
            if (kb + 32 < k_end) graw(kr, kb + 32, C);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < NDT; ++d) o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[0][r][d], sacc[0][r], o[d], 0, 0, 0);
            stage_v(1);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < NDT; ++d) o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[0][r][d], sacc[1][r], o[d], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int d = 0; d < NDT; ++d)
                        o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[s][r][d], sacc[s][r], o[d], 0, 0, 0);
            if constexpr (kPipe) {
                if (kb + 32 < k_end) load_v(kb + 32);
            }
        }
    }
    // ---- combine the key slices: O^T sub-tile d, accumulator register r holds MFMA row 4 kq + r = head dim 16 d + 4 kq + r
    // (or NDT (4 kq + r) + d with the float4 V layout above), column (query) lj
    constexpr bool kV4o = (NDT % 4) == 0;
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) Os[(wave * DH + (kV4o ? NDT * (4 * kq + r) + d : d * 16 + 4 * kq + r)) * 17 + lj] = o[d][r];
    if (kq == 0) {
        Ms[wave * 16 + lj] = m_run;
        Ls[wave * 16 + lj] = l_run;
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * DH; idx += NW * 64) {
        const int qq = idx / DH;
        const int d = idx - qq * DH;
        float mmax = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) mmax = fmaxf(mmax, Ms[w * 16 + qq]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float wt = __builtin_amdgcn_exp2f(Ms[w * 16 + qq] - mmax);
            num += wt * Os[(w * DH + d) * 17 + qq];
            den += wt * Ls[w * 16 + qq];
        }
        if (q0 + qq < L) {
            out[((int64_t)b * L + q0 + qq) * out_row + h * DH + d] = num / den;
            if (lse && d == 0) lse[(int64_t)bh * ((L + 31) & ~31) + q0 + qq] = mmax + log2f(den);
        }
    }
}

// The same merge with the split count a template parameter (NS key splits, whole 32-query tiles): every partial a thread will
// combine — its NS / 8 (m, l) pairs and its NS / NG float4 rows of O^T — is requested before the first wait.  The generic kernel
// above needs three dependent round trips (statistics, then two batches of 8 partial rows); at one scene this launch is pure
// latency.  Same thread mapping and summation order: bit-identical results.
template <int DH, int DG, int NS>
__global__ __launch_bounds__(256) void flash_merge_fixed_kernel(FlashArgs a) {
    PARQ_TL_KERNEL(kTlFlashMerge);
    constexpr int NG = 256 / (8 * DG);               // split groups
    constexpr int SPT = NS / NG;                     // partial rows per thread
    constexpr int PER = NS / 8;                      // (m, l) pairs per thread
    static_assert(NS % NG == 0 && NS % 8 == 0 && SPT <= 16, "split count");
    __shared__ __attribute__((aligned(16))) float wsm[NS * 32];
    __shared__ float dsm[8 * 32];
    __shared__ __attribute__((aligned(16))) float part[NG * DG * 32];
    const FlashHead fh = flash_head(a, blockIdx.y);
    const int bh = fh.bh, b = fh.b, h = fh.h;
    const int q0 = blockIdx.x * 32;
    const int dg0 = blockIdx.z * DG;
    const int tq = threadIdx.x & 31;
    const int td = threadIdx.x >> 5;
    const int Lq_pad = a.Lq;                         // a.Lq % 32 == 0 (launcher)
    const int q = q0 + tq;
    const int64_t pb = (int64_t)blockIdx.y * NS;
    const int q4 = threadIdx.x & 7;
    const int dd = (threadIdx.x >> 3) & (DG - 1);
    const int half = threadIdx.x / (8 * DG);
    const int s_begin = half * SPT;
    const int64_t sstride = (int64_t)DH * Lq_pad;
    const float* o0 = a.o_part + (pb * DH + dg0 + dd) * (int64_t)Lq_pad + q0 + q4 * 4;

    float mloc[PER], lloc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        mloc[i] = a.m_part[(pb + td + i * 8) * Lq_pad + q];
        lloc[i] = a.l_part[(pb + td + i * 8) * Lq_pad + q];
    }
    f32x4 x[SPT];
#pragma unroll
    for (int u = 0; u < SPT; ++u) x[u] = *reinterpret_cast<const f32x4*>(o0 + (s_begin + u) * sstride);
    __builtin_amdgcn_sched_barrier(0);               // keep every load above the first wait

    float mmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < PER; ++i) mmax = fmaxf(mmax, mloc[i]);
    dsm[td * 32 + tq] = mmax;
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) mmax = fmaxf(mmax, dsm[i * 32 + tq]);
    lds_barrier();
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float w = __builtin_amdgcn_exp2f(mloc[i] - mmax);
        wsm[(td + i * 8) * 32 + tq] = w;
        den += w * lloc[i];
    }
    dsm[td * 32 + tq] = den;
    lds_barrier();
    if (a.lse && blockIdx.z == 0 && td == 0) {
        float dn = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) dn += dsm[i * 32 + tq];
        a.lse[(int64_t)bh * Lq_pad + q] = mmax + log2f(dn);
    }
    if ((a.peaky || a.head_min) && blockIdx.z == 0 && td == 0) {     // attention mode 4: a row carried by too few keys (FlashArgs)
        float dn = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) dn += dsm[i * 32 + tq];
        if (a.peaky && __any(dn < a.peaky_l) && tq == 0) {
            atomicOr(a.peaky, 1 << h);
            if (a.peaky_it) atomicOr(a.peaky_it, 1 << h);
        }
        if (a.peaky_min || a.head_min) {
            float dmin = dn;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o));
            if (tq == 0 && a.peaky_min) atomicMax(a.peaky_min, 0x7fffffff - __float_as_int(dmin));
            if (tq == 0 && a.head_min) atomicMax(a.head_min + h, 0x7fffffff - __float_as_int(dmin));
        }
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < SPT; ++u) acc += x[u] * *reinterpret_cast<const f32x4*>(&wsm[(s_begin + u) * 32 + q4 * 4]);
    *reinterpret_cast<f32x4*>(&part[(half * DG + dd) * 32 + q4 * 4]) = acc;
    lds_barrier();
    for (int idx = threadIdx.x; idx < 32 * DG; idx += 256) {
        const int qq = idx / DG;
        const int d = idx - qq * DG;
        float dn = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) dn += dsm[i * 32 + qq];
        float acc2 = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) acc2 += part[(g2 * DG + d) * 32 + qq];
        a.out[(int64_t)b * a.out_batch + (int64_t)(q0 + qq) * a.out_row + h * DH + dg0 + d] = acc2 / dn;
    }
}

template <int DH, int NW>
hipError_t launch_one(const FlashArgs& a, hipStream_t s) {
    static DynLdsOnce once;
    const size_t lds = Tile<DH>::lds_bytes();
    if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&flash_f32_kernel<DH, NW>), lds); e != hipSuccess) return e;
    dim3 grid(a.nsplit, ceil_div(a.Lq, 32 * NW), a.B * a.H);
    hipLaunchKernelGGL((flash_f32_kernel<DH, NW>), grid, dim3(NW * 64), lds, s, a);
    return hipGetLastError();
}

template <int DH>
hipError_t launch_dh(const FlashArgs& a, int nw, hipStream_t s) {
    switch (nw) {
        case 1: return launch_one<DH, 1>(a, s);
        case 2: return launch_one<DH, 2>(a, s);
        case 4: return launch_one<DH, 4>(a, s);
        case 8:
            if constexpr (DH <= 64) return launch_one<DH, 8>(a, s);
            else return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

template <int DH>
hipError_t merge_dh(const FlashArgs& a, hipStream_t s) {
    const size_t lds = ((size_t)a.nsplit * 32 + 8 * 32 + (size_t)2 * 16 * 32) * sizeof(float);
    static const int dg_env = [] { const char* e = dev_env("PARQ_MERGE_DG"); return e ? atoi(e) : 0; }();
    // 8 dims per workgroup while that is what it takes to cover the chip (one scene), else 16
    const int64_t wg16 = (int64_t)ceil_div(a.Lq, 32) * a.B * flash_launch_heads(a) * (DH / 16);
    const int dg = dg_env == 8 || dg_env == 16 ? dg_env : (wg16 < device_num_cus() ? 8 : 16);
    static const bool fixed_off = [] { const char* e = dev_env("PARQ_MERGE_FIXED"); return e && e[0] == '0'; }();
    if constexpr (DH == 64) {
        if (dg == 8 && a.nsplit == 64 && a.Lq % 32 == 0 && !fixed_off) {       // the one-scene cross-attention merge: all loads up front
            hipLaunchKernelGGL((flash_merge_fixed_kernel<DH, 8, 64>), dim3(a.Lq / 32, a.B * flash_launch_heads(a), DH / 8), dim3(256), 0, s, a);
            return hipGetLastError();
        }
    }
    if constexpr (DH == 256) {
        // the shipped geometry at one scene (32 key splits): the same all-loads-up-front form (10.5 -> ~7 us per launch: 1.986 -> 1.959 ms per forward, outputs bit-identical)
        if (dg == 16 && a.nsplit == 32 && a.Lq % 32 == 0 && !fixed_off) {
            hipLaunchKernelGGL((flash_merge_fixed_kernel<DH, 16, 32>), dim3(a.Lq / 32, a.B * flash_launch_heads(a), DH / 16), dim3(256), 0, s, a);
            return hipGetLastError();
        }
    }
    if (dg == 8) {
        static DynLdsOnce once;
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&flash_merge_kernel<DH, 8>), 96 * 1024); e != hipSuccess) return e;
        hipLaunchKernelGGL((flash_merge_kernel<DH, 8>), dim3(ceil_div(a.Lq, 32), a.B * flash_launch_heads(a), DH / 8), dim3(256), lds, s, a);
    } else {
        static DynLdsOnce once;
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&flash_merge_kernel<DH, 16>), 96 * 1024); e != hipSuccess) return e;
        hipLaunchKernelGGL((flash_merge_kernel<DH, 16>), dim3(ceil_div(a.Lq, 32), a.B * flash_launch_heads(a), DH / 16), dim3(256), lds, s, a);
    }
    return hipGetLastError();
}

}  // namespace

template <int DH>
static hipError_t launch_self_dh(const float* qkv, int64_t row_stride, int B, int H, int L, float* out, int64_t out_row,
                                 hipStream_t s, float* lse, float drop_p, uint32_t drop_seed) {
    // head dims up to 64: 8 waves (two per SIMD); 128 / 256: 4 waves, one per SIMD, so that a wave may hold the Q / K / V fragments
    // of a 32-key block and the whole O^T accumulator in the unified register file
    constexpr int NW = DH <= 64 ? 8 : 4;
    const size_t lds = ((size_t)NW * DH * 17 + 2 * NW * 16 + (DH == 64 ? (size_t)NW * 16 * (DH + 16) : 0)) * sizeof(float);
    if (lds > 64 * 1024) {
        static DynLdsOnce once;
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&self_attn_kernel<DH, NW>), lds); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((self_attn_kernel<DH, NW>), dim3(ceil_div(L, 16), B * H), dim3(NW * 64), lds, s, qkv, row_stride, H, L, out,
                       out_row, lse, drop_p, drop_seed);
    return hipGetLastError();
}

// qkv: (B, L, row_stride) with q | k | v at column offsets 0, H*dh, 2*H*dh.  dh in {32, 64, 128, 256}.
hipError_t launch_self_attn(const float* qkv, int64_t row_stride, int B, int H, int L, int dh, float* out,
                            int64_t out_row, hipStream_t s, float* lse, float drop_p, uint32_t drop_seed) {
    if (dh == 64) return launch_self_dh<64>(qkv, row_stride, B, H, L, out, out_row, s, lse, drop_p, drop_seed);
    if (dh == 32) return launch_self_dh<32>(qkv, row_stride, B, H, L, out, out_row, s, lse, drop_p, drop_seed);
    if (dh == 128) return launch_self_dh<128>(qkv, row_stride, B, H, L, out, out_row, s, lse, drop_p, drop_seed);
    if (dh == 256) return launch_self_dh<256>(qkv, row_stride, B, H, L, out, out_row, s, lse, drop_p, drop_seed);
    return hipErrorInvalidValue;
}

int flash_key_tile(int dh) { return dh <= 64 ? 64 : 32; }
int flash_lq_pad(int Lq) { return (Lq + 31) & ~31; }

static int max_nw(int dh) { return dh <= 64 ? 8 : 4; }

// Waves per workgroup: the largest that still leaves >= one workgroup per CU.
int flash_pick_nw(int B, int H, int Lq, int Lk, int dh, int num_cus) {
    const int nt = ceil_div(Lk, flash_key_tile(dh));
    if (dh >= 128) {          // register budget: keep the staging registers per lane small
        int nw = flash_lq_pad(Lq) / 32;
        return nw >= 4 ? 4 : (nw >= 2 ? 2 : 1);
    }
    for (int nw = max_nw(dh); nw > 1; nw >>= 1) {
        if (nw * 32 > flash_lq_pad(Lq) && nw > 1) continue;
        const int64_t wgs = (int64_t)B * H * ceil_div(Lq, 32 * nw) * nt;
        if (wgs >= num_cus) return nw;
    }
    return 1;
}

int flash_pick_splits(int B, int H, int Lq, int Lk, int dh, int num_cus) {
    const int nw = flash_pick_nw(B, H, Lq, Lk, dh, num_cus);
    const int nt = ceil_div(Lk, flash_key_tile(dh));
    const int64_t base = (int64_t)B * H * ceil_div(Lq, 32 * nw);
    const int per_cu = 1;                         // 8-wave workgroups: registers admit one per CU
    int64_t want = ceil_div64((int64_t)num_cus * per_cu, base);
    if (want < 1) want = 1;
    if (want > nt) want = nt;
    if (want > 256) want = 256;
    return (int)want;
}

size_t flash_scratch_bytes(int B, int H, int Lq, int dh, int nsplit) {
    const size_t lp = (size_t)flash_lq_pad(Lq);
    return ((size_t)B * H * nsplit * lp * (size_t)(dh + 2)) * sizeof(float);
}

int device_num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        else
            cus = 256;
    }
    return cus;
}

static bool dh_ok(int dh) { return dh == 32 || dh == 64 || dh == 128 || dh == 256; }

hipError_t launch_flash(const FlashArgs& a, hipStream_t s) {
    if (!dh_ok(a.dh) || a.nsplit < 1 || a.nsplit > 256) return hipErrorInvalidValue;
    if (a.nsplit > ceil_div(a.Lk, flash_key_tile(a.dh))) return hipErrorInvalidValue;
    const int nw = flash_pick_nw(a.B, a.H, a.Lq, a.Lk, a.dh, device_num_cus());
    switch (a.dh) {
        case 32: return launch_dh<32>(a, nw, s);
        case 64: return launch_dh<64>(a, nw, s);
        case 128: return launch_dh<128>(a, nw, s);
        default: return launch_dh<256>(a, nw, s);
    }
}

hipError_t launch_flash_merge(const FlashArgs& a, hipStream_t s) {
    if (!dh_ok(a.dh)) return hipErrorInvalidValue;
    switch (a.dh) {
        case 32: return merge_dh<32>(a, s);
        case 64: return merge_dh<64>(a, s);
        case 128: return merge_dh<128>(a, s);
        default: return merge_dh<256>(a, s);
    }
}

PARQ_TL_DEFINE_SETTER(tl_set_flash)

}  // namespace parq
