// Attention backward, first correct version.  One kernel serves the self-attention among the queries
// and the dense cross-attention against the memory tokens:
//     P = exp2(s2 - lse2),  s2 = q.k * log2(e)/sqrt(dh)         (lse2 saved by the forward merge / self-attention kernel)
//     dP = dO.v,  dS = P (dP - D),  D_i = dO_i . O_i            (attn_bwd_rowdot_kernel)
//     dV_j = sum_i P_ij dO_i,  dK_j = sum_i dS_ij q_i / sqrt(dh),  dQ_i = sum_j dS_ij k_j / sqrt(dh)
// A workgroup owns 256 keys (one per thread, K_j and V_j in registers, dK_j / dV_j accumulated in registers over all
// queries: complete, so they are stored without atomics) and walks the queries in tiles of 32 staged in LDS; the dS
// tile goes through LDS for the dQ contraction, whose per-workgroup partial is added to global dQ with float atomics.
// All arithmetic is fp32 on the vector ALU: this is the correctness baseline that an MFMA kernel is checked against,
// not the fast path (cfg 3: ~50 GFMA per iteration).
#include "common.hpp"

#include <cstdlib>
#include <type_traits>
#include <cstring>
#include <cmath>

namespace parq {

namespace {

constexpr int kMaxBwdIters = 16;

struct AttnBwdArgs {
    const float* q; int64_t q_batch, q_head, q_row;
    const float* k; int64_t k_batch, k_head, k_row;
    const float* v; int64_t v_batch, v_head, v_row;
    const float* dO; int64_t do_batch, do_head, do_row;
    const float* lse;          // [B*H][Lq_pad], log2 domain
    const float* D;            // [B*H][Lq_pad]
    float* gq; int64_t gq_batch, gq_head, gq_row;     // += (atomic)
    float* gq_part;            // optional [B*H][key blocks of 256][Lq_pad][64]: per-workgroup dQ partials (MFMA kernel), summed
                               // by attn_bwd_dq_reduce_kernel instead of ~50 M global float atomics per scene and iteration
    float* gk; int64_t gk_batch, gk_head, gk_row;     // = or += (accumulate)
    float* gv; int64_t gv_batch, gv_head, gv_row;
    int B, H, Lq, Lk, accumulate_kv;
    float drop_p; uint32_t drop_seed;   // dropout on the probabilities (same stream as the forward)
    // split kernel only: the recurrent iterations that share K / V, processed by ONE launch (dK / dV accumulate in registers over all
    // of them).  Iteration t reads q + q_off[t], lse + lse_off[t], dO + t * do_it, D + t * D_it with dropout stream seeds[t] and
    // writes its dQ partials at gq_part + t * gqp_it.  n_it = 1 with zero offsets is the plain single-iteration call.
    int n_it;
    int64_t do_it, D_it, gqp_it;
    int64_t q_off[kMaxBwdIters], lse_off[kMaxBwdIters];
    uint32_t seeds[kMaxBwdIters];
    unsigned int* kv_absmax;   // optional: atomicMax of the bit pattern of |dK|, |dV| as written (scale of the projection backward)
    // split kernel, second version: K / V straight from the forward's 16-bit cache (flash_split.hip layout: 32-key blocks
    // [K_hi | K_lo | V_hi | V_lo], or [K | V] in the single-product modes) instead of fp32 k / v rebuilt from it
    const _Float16* kvcache;
    int cache_kind;            // kF16 / kBF16 of a single-product cache
    int probe;                     // development build only (PARQ_ATTN_BWD_PROBE): pieces of attn_bwd_split2_kernel left out, for timing
    int kb_group;                  // attn_bwd_split2_kernel: consecutive key blocks per workgroup (they share one slot of dQ partials)
};

template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qt = smem;                     // [32][DH]
    float* Ot = Qt + 32 * DH;             // [32][DH]   dO tile
    float* dS = Ot + 32 * DH;             // [32][257]
    float* Ks = dS + 32 * 257;            // [256][DH + 1]
    float* st = Ks + 256 * (DH + 1);      // [2][32] lse, D
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int j0 = blockIdx.x * 256;
    const int tid = threadIdx.x;
    const int j = j0 + tid;
    const bool jok = j < a.Lk;
    const int Lq_pad = (a.Lq + 31) & ~31;
    const float c2 = 1.4426950408889634f / sqrtf((float)DH);     // log2 domain scale
    const float cn = 1.f / sqrtf((float)DH);
    float kj[DH], vj[DH], gkj[DH], gvj[DH];
    {
        const float* kp = a.k + (int64_t)b * a.k_batch + (int64_t)h * a.k_head + (int64_t)(jok ? j : 0) * a.k_row;
        const float* vp = a.v + (int64_t)b * a.v_batch + (int64_t)h * a.v_head + (int64_t)(jok ? j : 0) * a.v_row;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const float4 k4 = *reinterpret_cast<const float4*>(kp + d);
            const float4 v4 = *reinterpret_cast<const float4*>(vp + d);
            kj[d] = k4.x; kj[d + 1] = k4.y; kj[d + 2] = k4.z; kj[d + 3] = k4.w;
            vj[d] = v4.x; vj[d + 1] = v4.y; vj[d + 2] = v4.z; vj[d + 3] = v4.w;
        }
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            gkj[d] = 0.f;
            gvj[d] = 0.f;
            Ks[tid * (DH + 1) + d] = jok ? kj[d] : 0.f;
        }
    }
    for (int i0 = 0; i0 < a.Lq; i0 += 32) {
        __syncthreads();
        for (int idx = tid; idx < 32 * DH; idx += 256) {
            const int i = idx / DH, d = idx - i * DH;
            const bool ok = i0 + i < a.Lq;
            Qt[idx] = ok ? a.q[(int64_t)b * a.q_batch + (int64_t)h * a.q_head + (int64_t)(i0 + i) * a.q_row + d] : 0.f;
            Ot[idx] = ok ? a.dO[(int64_t)b * a.do_batch + (int64_t)h * a.do_head + (int64_t)(i0 + i) * a.do_row + d] : 0.f;
        }
        if (tid < 32) {
            st[tid] = i0 + tid < a.Lq ? a.lse[(int64_t)bh * Lq_pad + i0 + tid] : 0.f;
            st[32 + tid] = i0 + tid < a.Lq ? a.D[(int64_t)bh * Lq_pad + i0 + tid] : 0.f;
        }
        __syncthreads();
        for (int i = 0; i < 32; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) {
                s += Qt[i * DH + d] * kj[d];
                dp += Ot[i * DH + d] * vj[d];
            }
            const bool ok = jok && (i0 + i < a.Lq);
            const float p = ok ? __builtin_amdgcn_exp2f(s * c2 - st[i]) : 0.f;
            float keep = 1.f;
            if (a.drop_p > 0.f)
                keep = drop_keep(drop_rowhash(a.drop_seed, (uint32_t)(bh * a.Lq + i0 + i)), (uint32_t)j, a.drop_p) ? 1.f / (1.f - a.drop_p) : 0.f;
            const float ds = p * (dp * keep - st[32 + i]);
            dS[i * 257 + tid] = ds;
#pragma unroll
            for (int d = 0; d < DH; ++d) {
                gkj[d] += ds * Qt[i * DH + d];
                gvj[d] += p * keep * Ot[i * DH + d];
            }
        }
        __syncthreads();
        // dQ tile: thread (i = tid >> 3, 8 consecutive d) sums over this workgroup's 256 keys
        {
            const int i = tid >> 3, d0 = (tid & 7) * (DH / 8);
            float acc[DH / 8];
#pragma unroll
            for (int e = 0; e < DH / 8; ++e) acc[e] = 0.f;
            for (int jj = 0; jj < 256; ++jj) {
                const float ds = dS[i * 257 + jj];
#pragma unroll
                for (int e = 0; e < DH / 8; ++e) acc[e] += ds * Ks[jj * (DH + 1) + d0 + e];
            }
            if (i0 + i < a.Lq) {
                float* gp = a.gq + (int64_t)b * a.gq_batch + (int64_t)h * a.gq_head + (int64_t)(i0 + i) * a.gq_row + d0;
#pragma unroll
                for (int e = 0; e < DH / 8; ++e) atomicAdd(gp + e, acc[e] * cn);
            }
        }
    }
    if (jok) {
        float* gkp = a.gk + (int64_t)b * a.gk_batch + (int64_t)h * a.gk_head + (int64_t)j * a.gk_row;
        float* gvp = a.gv + (int64_t)b * a.gv_batch + (int64_t)h * a.gv_head + (int64_t)j * a.gv_row;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            gkp[d] = (a.accumulate_kv ? gkp[d] : 0.f) + gkj[d] * cn;
            gvp[d] = (a.accumulate_kv ? gvp[d] : 0.f) + gvj[d];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// MFMA version for long key sequences (the cross-attention against all view tokens), head dim 64, exact fp32
// (v_mfma_f32_32x32x2_f32).  A workgroup owns 256 keys, each of its 8 waves 32 of them: K_j and V_j stay in registers as
// B operands, dK^T / dV^T (d x keys) accumulate in registers over all queries, and the queries are streamed in tiles of
// 32 through LDS.  With queries as accumulator ROWS and keys as COLUMNS (lanes):
//     S  = Q K^T, dP = dO V^T          A = Q / dO tile from LDS, B = K / V registers
//     P = exp2(S c2 - lse_q), dS = P (dP - D_q)                       (lse, D per accumulator row)
//     dV^T += dO^T P, dK^T += Q^T dS   contraction over the queries = accumulator rows: P / dS registers ARE the B operand
//     dQ  += dS K                      contraction over this wave's keys: dS goes through a per-wave LDS tile to become
//                                      the A operand, K comes from a per-wave LDS copy; the 8 waves add their partial
//                                      tiles into one LDS tile (ds_add_f32), which is added to global dQ with atomics.
constexpr int kQs = 68;      // row stride (floats) of the Q / dO tiles
constexpr int kKc = 80;      // row stride of the per-wave K copy (16 q-lanes x 4 key rows of the 16x16x4 B operand: conflict-free)

template <int NWV>        // waves per workgroup = 32-key blocks per workgroup: 8 for the long cross-attention, 2 for the 256-key self-attention
__global__ __launch_bounds__(NWV * 64) void attn_bwd_mfma_kernel(AttnBwdArgs a) {
    constexpr int NT = NWV * 64;
    constexpr int KW = NWV * 32;            // keys per workgroup
    constexpr int kDs = KW + 4;             // row stride of the workgroup's dS tile [32 queries][KW keys]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                       // [32][kQs]
    float* Os = Qs + 32 * kQs;              // [32][kQs]  dO tile
    float* st = Os + 32 * kQs;              // [2][32] lse, D
    float* Ds = st + 64;                    // [32][kDs]: dS of the current query tile against all 256 keys of the workgroup
    float* Kall = Ds + 32 * kDs;            // [KW][kKc]: K of the workgroup's keys, wave w owns rows 32 w ..
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    float* Kc = Kall + wave * 32 * kKc;     // this wave's [32][kKc]
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int j0 = blockIdx.x * KW + wave * 32;
    const int j = j0 + li;
    const bool jok = j < a.Lk;
    const int Lq_pad = (a.Lq + 31) & ~31;
    const float c2 = 1.4426950408889634f / 8.f;       // log2(e) / sqrt(64)
    const float cn = 1.f / 8.f;

    // K, V of this lane's key: lane (j, kh) holds d = kh*32 .. kh*32+31 (B operands); K also copied to LDS [j][d]
    float kf[32], vf[32];
    {
        const float* kp = a.k + (int64_t)b * a.k_batch + (int64_t)h * a.k_head + (int64_t)(jok ? j : 0) * a.k_row + kh * 32;
        const float* vp = a.v + (int64_t)b * a.v_batch + (int64_t)h * a.v_head + (int64_t)(jok ? j : 0) * a.v_row + kh * 32;
#pragma unroll
        for (int u = 0; u < 32; u += 4) {
            float4 k4 = *reinterpret_cast<const float4*>(kp + u);
            float4 v4 = *reinterpret_cast<const float4*>(vp + u);
            if (!jok) { k4 = float4{0.f, 0.f, 0.f, 0.f}; v4 = k4; }
            kf[u] = k4.x; kf[u + 1] = k4.y; kf[u + 2] = k4.z; kf[u + 3] = k4.w;
            vf[u] = v4.x; vf[u + 1] = v4.y; vf[u + 2] = v4.z; vf[u + 3] = v4.w;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) Kc[li * kKc + kh * 32 + u] = kf[u];
    }
    f32x16 gk[2], gv[2];                    // dK^T, dV^T: rows d (2 x 32), columns keys
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gk[dt][r] = 0.f; gv[dt][r] = 0.f; }

    for (int i0 = 0; i0 < a.Lq; i0 += 32) {
        __syncthreads();                                    // previous tile fully consumed
        for (int idx = tid; idx < 32 * 16; idx += NT) {    // Q and dO tiles: 32 rows x 16 float4
            const int i = idx >> 4, c4 = idx & 15;
            const bool ok = i0 + i < a.Lq;
            float4 q4 = float4{0.f, 0.f, 0.f, 0.f}, o4 = q4;
            if (ok) {
                q4 = *reinterpret_cast<const float4*>(a.q + (int64_t)b * a.q_batch + (int64_t)h * a.q_head + (int64_t)(i0 + i) * a.q_row + c4 * 4);
                o4 = *reinterpret_cast<const float4*>(a.dO + (int64_t)b * a.do_batch + (int64_t)h * a.do_head + (int64_t)(i0 + i) * a.do_row + c4 * 4);
            }
            *reinterpret_cast<float4*>(Qs + i * kQs + c4 * 4) = q4;
            *reinterpret_cast<float4*>(Os + i * kQs + c4 * 4) = o4;
        }
        if (tid < 32) {
            st[tid] = i0 + tid < a.Lq ? a.lse[(int64_t)bh * Lq_pad + i0 + tid] : 0.f;
            st[32 + tid] = i0 + tid < a.Lq ? a.D[(int64_t)bh * Lq_pad + i0 + tid] : 0.f;
        }
        __syncthreads();

        // ---- S = Q K^T, dP = dO V^T   (rows = queries, columns = this wave's keys)
        f32x16 sacc, pacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
        for (int u = 0; u < 32; u += 4) {
            const float4 q4 = *reinterpret_cast<const float4*>(Qs + li * kQs + kh * 32 + u);
            const float4 o4 = *reinterpret_cast<const float4*>(Os + li * kQs + kh * 32 + u);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, kf[u], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, kf[u + 1], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, kf[u + 2], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, kf[u + 3], sacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.x, vf[u], pacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.y, vf[u + 1], pacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.z, vf[u + 2], pacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(o4.w, vf[u + 3], pacc, 0, 0, 0);
        }
        // ---- P, dS (accumulator row r of this lane is query mfma32_row(r, lane))
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = mfma32_row(r, lane);
            const bool ok = jok && (i0 + qi < a.Lq);
            const float p = ok ? __builtin_amdgcn_exp2f(sacc[r] * c2 - st[qi]) : 0.f;
            float keep = 1.f;
            if (a.drop_p > 0.f)
                keep = drop_keep(drop_rowhash(a.drop_seed, (uint32_t)(bh * a.Lq + i0 + qi)), (uint32_t)j, a.drop_p) ? 1.f / (1.f - a.drop_p) : 0.f;
            const float ds = p * (pacc[r] * keep - st[32 + qi]);
            sacc[r] = p * keep;
            pacc[r] = ds;
            Ds[qi * kDs + wave * 32 + li] = ds;
        }
        // ---- dV^T += dO^T P, dK^T += Q^T dS: MFMA step r contracts over the query pair (row(r,0), row(r,1))
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = mfma32_row(r, lane);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                gv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[qi * kQs + dt * 32 + li], sacc[r], gv[dt], 0, 0, 0);
                gk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[qi * kQs + dt * 32 + li], pacc[r], gk[dt], 0, 0, 0);
            }
        }
        // ---- dQ tile = dS K over ALL 256 keys of the workgroup: v_mfma_f32_16x16x4_f32, wave w owns the 16 x 16 block
        // (queries 16 (w >> 2) .., d 16 (w & 3) ..) of the 32 x 64 tile, so no cross-wave reduction is needed
        __syncthreads();
        {
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            const int l15 = lane & 15, kq = lane >> 4;
            for (int blk = wave; blk < 8; blk += NWV) {          // 8 blocks of 16 x 16 cover the 32 x 64 tile
                const int qb = (blk >> 2) * 16, db = (blk & 3) * 16;
                f32x4v g4 = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
                for (int t = 0; t < KW / 4; ++t) {
                    const int jj = 4 * t + kq;                   // key inside the workgroup
                    g4 = __builtin_amdgcn_mfma_f32_16x16x4f32(Ds[(qb + l15) * kDs + jj], Kall[jj * kKc + db + l15], g4, 0, 0, 0);
                }
                // accumulator: rows qb + 4 kq + r, column db + l15
                if (a.gq_part) {
                    float* part = a.gq_part + (((int64_t)bh * gridDim.x + blockIdx.x) * Lq_pad + i0) * 64;
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[(qb + 4 * kq + r) * 64 + db + l15] = g4[r] * cn;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = i0 + qb + 4 * kq + r;
                        if (i < a.Lq)
                            atomicAdd(a.gq + (int64_t)b * a.gq_batch + (int64_t)h * a.gq_head + (int64_t)i * a.gq_row + db + l15, g4[r] * cn);
                    }
                }
            }
        }
    }
    // ---- dK, dV of this wave's keys: transpose (d x keys) -> [key][d] through the wave's LDS copy, then row-contiguous update
    __syncthreads();                                     // every wave is done reading the other waves' K copies
    for (int which = 0; which < 2; ++which) {
        const float scale = which == 0 ? cn : 1.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Kc[li * kKc + dt * 32 + mfma32_row(r, lane)] = (which == 0 ? gk[dt][r] : gv[dt][r]) * scale;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float* g = which == 0 ? a.gk + (int64_t)b * a.gk_batch + (int64_t)h * a.gk_head : a.gv + (int64_t)b * a.gv_batch + (int64_t)h * a.gv_head;
        const int64_t grow = which == 0 ? a.gk_row : a.gv_row;
        for (int idx = lane; idx < 32 * 64; idx += 64) {
            const int jj = idx >> 6, d = idx & 63;
            if (j0 + jj < a.Lk) {
                float* o = g + (int64_t)(j0 + jj) * grow + d;
                *o = (a.accumulate_kv ? *o : 0.f) + Kc[jj * kKc + d];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}


// ------------------------------------------------------------------------------------------------
// Split-precision version of the kernel above (head dim 64, 256 keys per workgroup): every product runs on the fp16 matrix
// pipe as hi*hi + hi*lo + lo*hi with fp32 accumulation (fp32-class accuracy, see flash_split.hip), 5.3x fewer MFMA cycles
// than the exact-fp32 kernel.  dO is multiplied by a power of two `oscale` (chosen by the caller from max|dO|) so that the
// small gradient values keep their low halves out of the fp16 subnormals; results are multiplied back (exact).
//   LDS (fp16 hi and lo images of everything an MFMA reads as A or as a key-indexed B operand):
//     Qa / Oa [32 q][64 d]   rows, 8-d chunks swizzled by (q>>1)&7        -> A of S = Q K^T and dP = dO V^T
//     Qt / Ot [64 d][32 q]   d-major, chunk (m, kh) holds queries 16m+4kh+(e&3)+8(e>>2), swizzled by (d>>2)&3
//                                                                          -> A of dK^T += Q^T dS and dV^T += dO^T P
//     Ds      [32 q][256 j]  8-key chunks swizzled by q&15                -> A of dQ = dS K   (16x16x32 MFMA)
//     Kt      [64 d][256 j]  8-key chunks swizzled by d&15                -> B of dQ = dS K
//   registers: K_j, V_j fragments (B of S / dP), P and dS accumulators re-used as B operands (rows = queries), dK^T, dV^T.
constexpr int kSpKW = 256;

__device__ __forceinline__ void split4(const float* x, _Float16* hi, _Float16* lo) {
    half2v h0, l0, h1, l1;
    split_pair(x[0], x[1], h0, l0);
    split_pair(x[2], x[3], h1, l1);
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
}

// The probabilities feed the dV product as fp16 hi/lo: p = exp2(s - lse) <= 1, and with near-uniform attention over N keys
// p ~ 1/N (5e-6 at BASELINE cfg 3) is an fp16 SUBNORMAL — its hi/lo split then carries 6-7 bits (measured: 0.5 % error of the
// V-projection gradients at N = 192 000).  The split kernels therefore work with p' = 2^14 p in (0, 2^14]: the row statistics are
// staged as lse - 14 and D * oscale * 2^-14, dS = p' (dP * keep * 2^-14 - D') keeps its value with the same instruction count,
// and the dV accumulators are rescaled by 2^-14 in the epilogue.
constexpr float kPShift = 14.f;
constexpr float kPShiftInv = 1.f / 16384.f;
template <bool DROP>
__global__ __launch_bounds__(512) void attn_bwd_split_kernel(AttnBwdArgs a, const float* __restrict__ oscale_ptr) {
    const float oscale = *oscale_ptr;
    extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
    _Float16* Qa = sm;                      // [hi | lo][32][64]
    _Float16* Oa = Qa + 2 * 2048;
    _Float16* Qt = Oa + 2 * 2048;           // [hi | lo][64][32]
    _Float16* Ot = Qt + 2 * 2048;
    _Float16* Ds = Ot + 2 * 2048;           // [hi | lo][32][256]
    _Float16* Kt = Ds + 2 * 8192;           // [hi | lo][64][256]
    float* st = reinterpret_cast<float*>(Kt + 2 * 16384);   // [2][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int jw = wave * 32 + li;                   // key inside the workgroup
    const int j = blockIdx.x * kSpKW + jw;
    const bool jok = j < a.Lk;
    const int Lq_pad = (a.Lq + 31) & ~31;
    const float c2 = 1.4426950408889634f / 8.f, cn = 1.f / 8.f;
    const float inv_os = 1.f / oscale;

    // ---- K, V of this lane's key as B fragments: step t holds d = 16 t + 8 kh + e; K also goes d-major into LDS
    half8 kfh[4], kfl[4], vfh[4], vfl[4];
    {
        const float* kp = a.k + (int64_t)b * a.k_batch + (int64_t)h * a.k_head + (int64_t)(jok ? j : 0) * a.k_row;
        const float* vp = a.v + (int64_t)b * a.v_batch + (int64_t)h * a.v_head + (int64_t)(jok ? j : 0) * a.v_row;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int d0 = 16 * t + 8 * kh;
            float kx[8], vx[8];
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                const float4 k4 = *reinterpret_cast<const float4*>(kp + d0 + e);
                const float4 v4 = *reinterpret_cast<const float4*>(vp + d0 + e);
                kx[e] = jok ? k4.x : 0.f; kx[e + 1] = jok ? k4.y : 0.f; kx[e + 2] = jok ? k4.z : 0.f; kx[e + 3] = jok ? k4.w : 0.f;
                vx[e] = jok ? v4.x : 0.f; vx[e + 1] = jok ? v4.y : 0.f; vx[e + 2] = jok ? v4.z : 0.f; vx[e + 3] = jok ? v4.w : 0.f;
            }
            split8(kx, kfh[t], kfl[t]);
            split8(vx, vfh[t], vfl[t]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = d0 + e;
                const int off = d * 256 + (((jw >> 3) ^ (d & 15)) << 3) + (jw & 7);
                Kt[off] = kfh[t][e];
                Kt[16384 + off] = kfl[t][e];
            }
        }
    }
    // dropout: this lane's key is the column of every element it handles -> one column hash for the whole launch
    const uint32_t drop_col = DROP ? drop_colhash((uint32_t)j) : 0u;
    const uint32_t drop_thr = DROP ? drop_threshold(a.drop_p) : 0u;
    f32x16 gk[2], gv[2];                    // dK^T, dV^T: rows d (2 x 32), columns this wave's keys
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gk[dt][r] = 0.f; gv[dt][r] = 0.f; }

    for (int it = 0; it < a.n_it; ++it) {
    // wave-uniform per-iteration bases (scalar registers)
    const float* q_it = a.q + a.q_off[it] + (int64_t)b * a.q_batch + (int64_t)h * a.q_head;
    const float* do_it = a.dO + (int64_t)it * a.do_it + (int64_t)b * a.do_batch + (int64_t)h * a.do_head;
    const float* lse_it = a.lse + a.lse_off[it] + (int64_t)bh * Lq_pad;
    const float* D_it = a.D + (int64_t)it * a.D_it + (int64_t)bh * Lq_pad;
    const uint32_t seed_it = a.seeds[it];
    for (int i0 = 0; i0 < a.Lq; i0 += 32) {
        __syncthreads();                                    // previous tile fully consumed (also orders the Kt writes)
        {   // tile loader: thread -> (query i, 4 consecutive d); both layouts, hi and lo
            const int i = tid >> 4, d4 = (tid & 15) * 4;
            const bool ok = i0 + i < a.Lq;
            float q4[4] = {0.f, 0.f, 0.f, 0.f}, o4[4] = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float4 qq = *reinterpret_cast<const float4*>(q_it + (int64_t)(i0 + i) * a.q_row + d4);
                const float4 oo = *reinterpret_cast<const float4*>(do_it + (int64_t)(i0 + i) * a.do_row + d4);
                q4[0] = qq.x; q4[1] = qq.y; q4[2] = qq.z; q4[3] = qq.w;
                o4[0] = oo.x * oscale; o4[1] = oo.y * oscale; o4[2] = oo.z * oscale; o4[3] = oo.w * oscale;
            }
            _Float16 qh[4], ql[4], oh[4], ol[4];
            split4(q4, qh, ql);
            split4(o4, oh, ol);
            const int ca = ((d4 >> 3) ^ ((i >> 1) & 7)) * 8 + (d4 & 4);       // natural layout: chunk swizzle, 4-half offset
            typedef _Float16 half4v __attribute__((ext_vector_type(4)));
            *reinterpret_cast<half4v*>(Qa + i * 64 + ca) = half4v{qh[0], qh[1], qh[2], qh[3]};
            *reinterpret_cast<half4v*>(Qa + 2048 + i * 64 + ca) = half4v{ql[0], ql[1], ql[2], ql[3]};
            *reinterpret_cast<half4v*>(Oa + i * 64 + ca) = half4v{oh[0], oh[1], oh[2], oh[3]};
            *reinterpret_cast<half4v*>(Oa + 2048 + i * 64 + ca) = half4v{ol[0], ol[1], ol[2], ol[3]};
            // transposed layout: query i = 16 m + 4 kh' + (e&3) + 8 (e>>2)  ->  chunk 2m + kh', element e; the 4 consecutive d of
            // this thread share (d >> 2), so their addresses differ by the constant row stride
            const int m = i >> 4, r16 = i & 15, khq = (r16 >> 2) & 1, eq = (r16 & 3) + 4 * (r16 >> 3);
            const int tb = d4 * 32 + (((2 * m + khq) ^ ((d4 >> 2) & 3)) << 3) + eq;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Qt[tb + e * 32] = qh[e]; Qt[2048 + tb + e * 32] = ql[e];
                Ot[tb + e * 32] = oh[e]; Ot[2048 + tb + e * 32] = ol[e];
            }
        }
        if (tid < 32) {
            st[tid] = i0 + tid < a.Lq ? lse_it[i0 + tid] - kPShift : 0.f;
            st[32 + tid] = i0 + tid < a.Lq ? D_it[i0 + tid] * oscale * kPShiftInv : 0.f;
            if (DROP) reinterpret_cast<uint32_t*>(st + 64)[tid] = drop_rowhash(seed_it, (uint32_t)(bh * a.Lq + i0 + tid));
        }
        __syncthreads();

        // ---- S = Q K^T, dP = dO V^T  (rows = queries, columns = this wave's keys)
        f32x16 sacc, pacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int pos = ((2 * t + kh) ^ ((li >> 1) & 7)) * 8;
            const half8 qh8 = *reinterpret_cast<const half8*>(Qa + li * 64 + pos);
            const half8 ql8 = *reinterpret_cast<const half8*>(Qa + 2048 + li * 64 + pos);
            const half8 oh8 = *reinterpret_cast<const half8*>(Oa + li * 64 + pos);
            const half8 ol8 = *reinterpret_cast<const half8*>(Oa + 2048 + li * 64 + pos);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh8, kfh[t], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh8, kfl[t], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql8, kfh[t], sacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, vfh[t], pacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, vfl[t], pacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ol8, vfh[t], pacc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // keep the fragment loads of step t+1 from being hoisted (register budget)
        }
        // ---- P (with dropout), dS; accumulator register r is query mfma32_row(r, lane), column = this lane's key
        half8 ph[2], pl[2], sh[2], sl[2];
        const int ds_jk = (jw >> 3) ^ (4 * kh);
        // dropout element = (row bh * Lq + query, col key): the row hashes of the tile's 32 queries sit in LDS (st[64 ..])
        const float drop_inv = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
        unsigned keep_bits = 0xffffu;               // bit r: accumulator register r is kept (a rolled loop keeps the register count down)
        if constexpr (DROP) {
            keep_bits = 0u;
            const uint32_t* rhs = reinterpret_cast<const uint32_t*>(st + 64);
#pragma unroll 1
            for (int r = 0; r < 16; ++r)
                keep_bits |= (drop_keep_h(rhs[(r & 3) + 8 * (r >> 2) + 4 * kh], drop_col, drop_thr) ? 1u : 0u) << r;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float pv[8], dv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * m + e;
                const int qi = mfma32_row(r, lane);
                const bool ok = jok && (i0 + qi < a.Lq);
                const float p = ok ? __builtin_amdgcn_exp2f(sacc[r] * c2 - st[qi]) : 0.f;
                float keep = 1.f;
                if constexpr (DROP) keep = ((keep_bits >> r) & 1u) ? drop_inv : 0.f;
                pv[e] = p * keep;                                                            // p = 2^14 x probability (kPShift)
                dv[e] = p * (pacc[r] * (keep * kPShiftInv) - st[32 + qi]);
            }
            split8(pv, ph[m], pl[m]);
            split8(dv, sh[m], sl[m]);
            // dS of (query qi, key jw) also goes to LDS for the dQ product
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                constexpr int kRow[8] = {0, 1, 2, 3, 8, 9, 10, 11};      // mfma32_row(8 m + e, lane) = 16 m + kRow[e] + 4 kh
                const int qi = 16 * m + kRow[e] + 4 * kh;
                const int off = qi * 256 + ((ds_jk ^ kRow[e]) << 3) + (jw & 7);    // chunk (jw >> 3) ^ (qi & 15), qi & 15 = kRow[e] ^ 4 kh
                Ds[off] = sh[m][e];
                Ds[8192 + off] = sl[m][e];
            }
        }
        // ---- dV^T += dO^T P, dK^T += Q^T dS: contraction over the queries kmap(m, kh, e) = accumulator rows
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int d = dt * 32 + li;
                const int pos = d * 32 + (((2 * m + kh) ^ ((d >> 2) & 3)) << 3);
                const half8 oth = *reinterpret_cast<const half8*>(Ot + pos);
                const half8 otl = *reinterpret_cast<const half8*>(Ot + 2048 + pos);
                const half8 qth = *reinterpret_cast<const half8*>(Qt + pos);
                const half8 qtl = *reinterpret_cast<const half8*>(Qt + 2048 + pos);
                gv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(oth, ph[m], gv[dt], 0, 0, 0);
                gv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(oth, pl[m], gv[dt], 0, 0, 0);
                gv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(otl, ph[m], gv[dt], 0, 0, 0);
                gk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qth, sh[m], gk[dt], 0, 0, 0);
                gk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qth, sl[m], gk[dt], 0, 0, 0);
                gk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qtl, sh[m], gk[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        // ---- dQ tile = dS K over the 256 keys of the workgroup: wave w owns the 16 x 16 block (queries 16 (w>>2).., d 16 (w&3)..)
        __syncthreads();
        {
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            const int l15 = lane & 15, kq = lane >> 4;
            const int qrow = (wave >> 2) * 16 + l15, dcol = (wave & 3) * 16 + l15;
            f32x4v g4 = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int c = 4 * t + kq;                        // 8-key chunk 0..31
                const int pa = qrow * 256 + ((c ^ (qrow & 15)) << 3);
                const int pb = dcol * 256 + ((c ^ (dcol & 15)) << 3);
                const half8 ah = *reinterpret_cast<const half8*>(Ds + pa);
                const half8 al = *reinterpret_cast<const half8*>(Ds + 8192 + pa);
                const half8 bh8 = *reinterpret_cast<const half8*>(Kt + pb);
                const half8 bl8 = *reinterpret_cast<const half8*>(Kt + 16384 + pb);
                g4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh8, g4, 0, 0, 0);
                g4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl8, g4, 0, 0, 0);
                g4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh8, g4, 0, 0, 0);
                if (t & 1) __builtin_amdgcn_sched_barrier(0);
            }
            // accumulator: rows (wave>>2)*16 + 4 kq + r, column dcol
            const int qb = (wave >> 2) * 16;
            if (a.gq_part) {
                float* part = a.gq_part + (int64_t)it * a.gqp_it + (((int64_t)bh * gridDim.x + blockIdx.x) * Lq_pad + i0) * 64;
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(qb + 4 * kq + r) * 64 + dcol] = g4[r] * cn * inv_os;
            } else {                                        // single-iteration calls only (the launcher enforces it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + qb + 4 * kq + r;
                    if (i < a.Lq)
                        atomicAdd(a.gq + (int64_t)b * a.gq_batch + (int64_t)h * a.gq_head + (int64_t)i * a.gq_row + dcol, g4[r] * cn * inv_os);
                }
            }
        }
    }
    }
    // ---- dK, dV of this wave's keys: (d x keys) accumulators -> [key][d] through LDS (the Ds region, free now), row-contiguous update
    __syncthreads();
    float* tr = reinterpret_cast<float*>(Ds) + wave * (32 * 65);         // 8 waves x 8.3 KB <= 32 KB + Kt head room
    float kvmax = 0.f;
    for (int which = 0; which < 2; ++which) {
        const float scale = (which == 0 ? cn : kPShiftInv) * inv_os;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[li * 65 + dt * 32 + mfma32_row(r, lane)] = (which == 0 ? gk[dt][r] : gv[dt][r]) * scale;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float* g = which == 0 ? a.gk + (int64_t)b * a.gk_batch + (int64_t)h * a.gk_head : a.gv + (int64_t)b * a.gv_batch + (int64_t)h * a.gv_head;
        const int64_t grow = which == 0 ? a.gk_row : a.gv_row;
        const int j0 = blockIdx.x * kSpKW + wave * 32;
        for (int idx = lane; idx < 32 * 64; idx += 64) {
            const int jj = idx >> 6, d = idx & 63;
            if (j0 + jj < a.Lk) {
                float* o = g + (int64_t)(j0 + jj) * grow + d;
                const float val = (a.accumulate_kv ? *o : 0.f) + tr[jj * 65 + d];
                *o = val;
                kvmax = fmaxf(kvmax, fabsf(val));
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    if (a.kv_absmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kvmax = fmaxf(kvmax, __shfl_xor(kvmax, o));
        if (lane == 0) atomicMax(a.kv_absmax, __float_as_uint(kvmax));
    }
}

// ================================================================================================ split kernel, second version
// Same arithmetic as attn_bwd_split_kernel, reorganised around two gfx950 features:
//   * ds_read_b64_tr_b16 (LDS transpose read): within a 16-lane group, lane t supplies the address of the 8-byte piece
//     (row t >> 2, 4 columns from 4 (t & 3)) of a [4 rows][16 columns] 16-bit block and lane c receives column c of it, i.e. the four
//     rows of one column (probed on MI355X: tools/bench_src/tr16_probe.hip).  Operands whose contraction index runs along the ROWS of
//     their LDS image (dO^T, Q^T for dV / dK; dS and K for dQ) are read straight from the natural image: no transposed copies of the
//     Q / dO tile, dS written as 8-byte pieces instead of 32 two-byte stores per lane, K written as 16-byte chunks.
//   * the hi/lo-split Q / dO tile (and lse, D, the dropout row hashes) is the same for every key block, so a pack kernel builds it
//     ONCE per (iteration, scene, head, 32 queries) as the exact LDS image and the workgroups fetch it by LDS-DMA one tile ahead:
//     no loader VALU, no loader LDS writes, no staging registers.
// Two workgroup barriers per query tile instead of three.
constexpr int kImgHalfs = 8448;                  // [Q_hi | Q_lo | dO_hi | dO_lo] 4 x 2048 + 256 halfs of statistics (lse, D, row hashes)
// 16-byte chunk swizzle of a [rows][64 halfs] image (128-byte rows, two rows per 256-byte bank row).  With v = (r >> 1) & 7:
//   * bit 2 of the swizzle = bit 1 of r: the transpose reads of dV / dK touch rows r .. r + 3 x 64 bytes, rows r and r + 2 must use
//     different 64-byte halves;
//   * bits 2..1 = (r bit 1, r bit 3): the dQ reads of the K image touch rows {j, j+1, j+2, j+3, j+8, .. j+11} x 32 bytes, the four
//     same-parity rows must use four different 32-byte quarters;
//   * a bijection of v: the 8 same-parity rows of a ds_read_b128 lane group hit 8 different chunks.
__device__ __forceinline__ int img_swz(int r) {
    const int v = (r >> 1) & 7;
    return ((v & 1) << 2) | ((v >> 2) << 1) | ((v >> 1) & 1);
}

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half4v lds_tr16(const _Float16* p) {
    typedef short short4v __attribute__((ext_vector_type(4)));
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p));
    return __builtin_bit_cast(half4v, v);
}
__device__ __forceinline__ half8 cat4(half4v a, half4v b) { return half8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// one workgroup per (32-query tile, scene-head, iteration): the LDS image of that tile in global memory
__global__ __launch_bounds__(256) void attn_bwd_pack_kernel(AttnBwdArgs a, const float* __restrict__ oscale_ptr, _Float16* __restrict__ pack) {
    const float oscale = *oscale_ptr;
    const int tile = blockIdx.x, bh = blockIdx.y, it = blockIdx.z;
    const int b = bh / a.H, h = bh - b * a.H;
    const int ntiles = gridDim.x, Lq_pad = (a.Lq + 31) & ~31;
    _Float16* img = pack + (((int64_t)it * gridDim.y + bh) * ntiles + tile) * kImgHalfs;
    const int i = threadIdx.x >> 3, c = threadIdx.x & 7;
    const int qi = tile * 32 + i;
    float qx[8], ox[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { qx[e] = 0.f; ox[e] = 0.f; }
    if (qi < a.Lq) {
        const float* qp = a.q + a.q_off[it] + (int64_t)b * a.q_batch + (int64_t)h * a.q_head + (int64_t)qi * a.q_row + c * 8;
        const float* op = a.dO + (int64_t)it * a.do_it + (int64_t)b * a.do_batch + (int64_t)h * a.do_head + (int64_t)qi * a.do_row + c * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { qx[e] = qp[e]; ox[e] = op[e] * oscale; }
    }
    half8 qh, ql, oh, ol;
    split8(qx, qh, ql);
    split8(ox, oh, ol);
    const int off = i * 64 + ((c ^ img_swz(i)) << 3);
    *reinterpret_cast<half8*>(img + off) = qh;
    *reinterpret_cast<half8*>(img + 2048 + off) = ql;
    *reinterpret_cast<half8*>(img + 4096 + off) = oh;
    *reinterpret_cast<half8*>(img + 6144 + off) = ol;
    if (threadIdx.x < 32) {
        float* stf = reinterpret_cast<float*>(img + 8192);
        const int q2 = tile * 32 + threadIdx.x;
        const bool ok = q2 < a.Lq;
        // rows past Lq: lse = +inf makes every probability exp2(s - lse) exactly 0
        stf[threadIdx.x] = ok ? a.lse[a.lse_off[it] + (int64_t)bh * Lq_pad + q2] - kPShift : INFINITY;
        stf[32 + threadIdx.x] = ok ? a.D[(int64_t)it * a.D_it + (int64_t)bh * Lq_pad + q2] * oscale * kPShiftInv : 0.f;
        reinterpret_cast<uint32_t*>(stf + 64)[threadIdx.x] = drop_rowhash(a.seeds[it], (uint32_t)(bh * a.Lq + q2));
    }
}

// (no scheduling barriers inside the MFMA groups here: letting hipcc hoist the fragment reads over the previous step's MFMAs
// measured 18.2 -> 16.6 ms; the first version needs them to stay inside its register budget)
// PIPE (round 3): the dQ tile of query tile n - 1 is computed next to the softmax / dS arithmetic of tile n instead of at the end
// of its own tile — same results (same operands, same order inside every accumulation), two barriers per tile as before.
// CACHE: 0 = fp32 K / V (a.k, a.v); 8 = the forward's mode-4 stage cache; 3 / 1 = the forward's split / single-product cache (a.kvcache): the hi / lo pairs the forward
// multiplied are used as they are (no fp32 rebuild pass over 2 N C floats per scene, no second copy of K / V in HBM)
// kGrad1: in the three GRADIENT products of the tile (dV^T += dO^T P, dK^T += Q^T dS, dQ += dS K) the probabilities and dS enter as ONE fp16
// value each, rounded to nearest, instead of a hi / lo pair: 56 MFMAs per tile instead of 72, no lo plane of the dS^T image (a quarter of
// the dQ tile's LDS reads), no residual arithmetic.  These products contract over the 2048 queries of a (scene, head) (dV, dK) or over
// all keys (dQ), so the unbiased 2^-12 rounding noise of the factors averages to ~1e-5 of a gradient and below; the recomputed scores
// S and dP = dO V^T, which feed the exponential and dS itself, keep all three terms.  (false: the round-3 form, kept for A/B.)
constexpr bool kGrad1 = true;

// One key block of 256 keys (kbx) against every query tile of every iteration.  `add`: the workgroup has already written dQ partials of an
// earlier key block of its group into its slot (attn_bwd_split2_kernel): this block's are added to them.
template <bool DROP, bool RAGGED, int PIPE, int CACHE>
__device__ __forceinline__ void attn_bwd_split2_block(const AttnBwdArgs& a, const float oscale, const _Float16* __restrict__ pack,
                                                      const int kbx, const bool add) {
#ifdef PARQ_DEV_PROBES
    const int probe = a.probe;              // 1: no dQ tile, 2: no softmax / dS arithmetic, 4: no dV / dK products, 8: no S / dP products
#else
    constexpr int probe = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
    _Float16* Ds = sm;                      // dS^T [hi | lo][256 keys][32 queries], 8-byte pieces swizzled by (key >> 1) & 7 (64-byte rows:
                                            // distinct for the same-parity rows of a write group; rows j and j + 8 of a read use different halves)
    _Float16* Ki = Ds + 2 * 8192;           // K [hi | lo][256 keys][64 d], 16-byte chunks swizzled by img_swz(key)
    _Float16* Stg = Ki + 2 * 16384;         // [2][kImgHalfs]: the Q / dO tile images, filled by LDS-DMA
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int t16 = lane & 15, g16 = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int jw = wave * 32 + li;                   // key inside the workgroup
    const int j = kbx * kSpKW + jw;
    const bool jok = j < a.Lk;
    const int Lq_pad = (a.Lq + 31) & ~31;
    const int ntiles = Lq_pad >> 5;
    const float c2 = 1.4426950408889634f / 8.f, cn = 1.f / 8.f;
    const float inv_os = 1.f / oscale;

    // ---- K, V of this lane's key as B fragments: step t holds d = 16 t + 8 kh + e; K also goes into LDS (natural layout)
    half8 kfh[4], kfl[4], vfh[4], vfl[4];
    if constexpr (CACHE == 8) {
        // the forward's mode-4 stage cache (flash_split8.hip; 64 keys = 24 KB: K hi16 | K hi8 | K lo8 | V fp16): this wave's 32 keys are
        // block blk & 1 of stage blk >> 1.  K = hi16 + e4m3 lo 2^-10 (what the forward's scores saw, to its 15 bits), V = the fp16 value
        // (what the forward multiplied: the gradient goes straight through the rounding, as in the single-product modes).
        // K hi16 / V planes: the layouts of the split cache's K_hi / V_hi blocks.  K lo8: piece (c, h) of a key holds byte 8 m + e' <->
        // d = 32 m + 16 c + 4 h + (e' & 3) + 8 (e' >> 2); element e of fragment t (d = 16 t + 8 kh + e) is byte 8 (t >> 1) + 4 kh + (e & 3)
        // of piece (c = t & 1, h = e >> 2).
        const int nblk = (a.Lk + 31) >> 5;
        int blk = kbx * (kSpKW / 32) + wave;
        blk = blk < nblk ? blk : nblk - 1;
        const unsigned char* stage = reinterpret_cast<const unsigned char*>(a.kvcache) + ((int64_t)bh * (nblk >> 1) + (blk >> 1)) * kStage8Bytes;
        const int b2 = blk & 1;
        const _Float16* kb = reinterpret_cast<const _Float16*>(stage + kS8Kh16 + b2 * 4096);
        const _Float16* vb = reinterpret_cast<const _Float16*>(stage + kS8Vh16 + b2 * 4096);
        const int ksw = (li >> 1) & 7;
        const int vr = li & 15, vc = 2 * (li >> 4) + ((vr & 7) >> 2), ve = (vr & 3) + 4 * (vr >> 3);
        const half8 zero8 = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int o0 = li * 64 + ((t ^ ksw) << 3) + 4 * kh, o1 = li * 64 + (((4 + t) ^ ksw) << 3) + 4 * kh;
            kfh[t] = cat4(*reinterpret_cast<const half4v*>(kb + o0), *reinterpret_cast<const half4v*>(kb + o1));
            half8 lo8;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int w = *reinterpret_cast<const int*>(stage + kS8K8lo + (((b2 * 2 + (t & 1)) * 2 + hh) * 32 + li) * 16 + 8 * (t >> 1) + 4 * kh);
                lo8[4 * hh + 0] = (_Float16)(__builtin_amdgcn_cvt_f32_fp8(w, 0) * (1.f / kLo8Scale));
                lo8[4 * hh + 1] = (_Float16)(__builtin_amdgcn_cvt_f32_fp8(w, 1) * (1.f / kLo8Scale));
                lo8[4 * hh + 2] = (_Float16)(__builtin_amdgcn_cvt_f32_fp8(w, 2) * (1.f / kLo8Scale));
                lo8[4 * hh + 3] = (_Float16)(__builtin_amdgcn_cvt_f32_fp8(w, 3) * (1.f / kLo8Scale));
            }
            kfl[t] = lo8;
            half8 vh8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = 16 * t + 8 * kh + e;
                vh8[e] = vb[d * 32 + ((vc ^ ((d >> 2) & 3)) << 3) + ve];
            }
            vfh[t] = vh8;
            vfl[t] = zero8;
            if (!jok) { kfh[t] = zero8; kfl[t] = zero8; vfh[t] = zero8; }
            const int off = jw * 64 + (((2 * t + kh) ^ img_swz(jw)) << 3);
            *reinterpret_cast<half8*>(Ki + off) = kfh[t];
            *reinterpret_cast<half8*>(Ki + 16384 + off) = kfl[t];
        }
    } else if constexpr (CACHE != 0) {
        // this wave's 32 keys are ONE cache block (key li of block 8 blockIdx.x + wave).  K_x: [32 keys][8 chunks][8]; chunk 4 kh' + s
        // holds d = 32 (s >> 1) + 16 (s & 1) + 4 kh' + (e & 3) + 8 (e >> 2), stored at chunk position c ^ ((key >> 1) & 7): the eight
        // d = 16 t + 8 kh + e of fragment t are elements 4 kh .. 4 kh + 3 of chunks t (e < 4) and 4 + t (e >= 4).  V_x: [64 d][4 chunks][8];
        // chunk 2 m + kh'' holds keys 16 m + 4 kh'' + (e & 3) + 8 (e >> 2) at position c ^ ((d >> 2) & 3): one 16-bit element per d.
        constexpr int kBH = CACHE == 3 ? 8192 : 4096, kVo = CACHE == 3 ? 4096 : 2048;
        const int nblk = (a.Lk + 31) >> 5;
        int blk = kbx * (kSpKW / 32) + wave;
        blk = blk < nblk ? blk : nblk - 1;                         // past the end: any valid block (masked by jok below)
        const _Float16* cb = a.kvcache + ((int64_t)bh * nblk + blk) * kBH;
        const int ksw = (li >> 1) & 7;
        const int vr = li & 15, vc = 2 * (li >> 4) + ((vr & 7) >> 2), ve = (vr & 3) + 4 * (vr >> 3);
        const half8 zero8 = half8{0, 0, 0, 0, 0, 0, 0, 0};
        auto widen = [&](half8 raw, half8& hi, half8& lo) {          // single-product cache: the 16-bit value is the operand
            if constexpr (CACHE == 3) { hi = raw; return; }
            if (a.cache_kind == kBF16) {                            // bf16 bits -> fp32 -> fp16 hi / lo (exact: 8 significant bits)
                const bf16x8 bv = __builtin_bit_cast(bf16x8, raw);
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (float)bv[e];
                split8(x, hi, lo);
            } else { hi = raw; lo = zero8; }
        };
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int o0 = li * 64 + ((t ^ ksw) << 3) + 4 * kh, o1 = li * 64 + (((4 + t) ^ ksw) << 3) + 4 * kh;
            half8 kh8 = cat4(*reinterpret_cast<const half4v*>(cb + o0), *reinterpret_cast<const half4v*>(cb + o1));
            half8 kl8 = zero8;
            if constexpr (CACHE == 3) kl8 = cat4(*reinterpret_cast<const half4v*>(cb + 2048 + o0), *reinterpret_cast<const half4v*>(cb + 2048 + o1));
            half8 vh8, vl8 = zero8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = 16 * t + 8 * kh + e;
                const int ov = kVo + d * 32 + ((vc ^ ((d >> 2) & 3)) << 3) + ve;
                vh8[e] = cb[ov];
                if constexpr (CACHE == 3) vl8[e] = cb[2048 + ov];
            }
            widen(kh8, kfh[t], kl8);
            if constexpr (CACHE == 3) kfl[t] = kl8; else kfl[t] = kl8;
            widen(vh8, vfh[t], vl8);
            vfl[t] = vl8;
            if (!jok) { kfh[t] = zero8; kfl[t] = zero8; vfh[t] = zero8; vfl[t] = zero8; }
            const int off = jw * 64 + (((2 * t + kh) ^ img_swz(jw)) << 3);
            *reinterpret_cast<half8*>(Ki + off) = kfh[t];
            *reinterpret_cast<half8*>(Ki + 16384 + off) = kfl[t];
        }
    } else {
        const float* kp = a.k + (int64_t)b * a.k_batch + (int64_t)h * a.k_head + (int64_t)(jok ? j : 0) * a.k_row;
        const float* vp = a.v + (int64_t)b * a.v_batch + (int64_t)h * a.v_head + (int64_t)(jok ? j : 0) * a.v_row;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int d0 = 16 * t + 8 * kh;
            float kx[8], vx[8];
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                const float4 k4 = *reinterpret_cast<const float4*>(kp + d0 + e);
                const float4 v4 = *reinterpret_cast<const float4*>(vp + d0 + e);
                kx[e] = jok ? k4.x : 0.f; kx[e + 1] = jok ? k4.y : 0.f; kx[e + 2] = jok ? k4.z : 0.f; kx[e + 3] = jok ? k4.w : 0.f;
                vx[e] = jok ? v4.x : 0.f; vx[e + 1] = jok ? v4.y : 0.f; vx[e + 2] = jok ? v4.z : 0.f; vx[e + 3] = jok ? v4.w : 0.f;
            }
            split8(kx, kfh[t], kfl[t]);
            split8(vx, vfh[t], vfl[t]);
            const int off = jw * 64 + (((2 * t + kh) ^ img_swz(jw)) << 3);
            *reinterpret_cast<half8*>(Ki + off) = kfh[t];
            *reinterpret_cast<half8*>(Ki + 16384 + off) = kfl[t];
        }
    }
    const uint32_t drop_col = DROP ? drop_colhash((uint32_t)j) : 0u;
    const uint32_t drop_thr = DROP ? drop_threshold(a.drop_p) : 0u;
    const float drop_inv = DROP ? 1.f / (1.f - a.drop_p) : 1.f;
    f32x16 gk[2], gv[2];                    // dK^T, dV^T: rows d (2 x 32), columns this wave's keys
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gk[dt][r] = 0.f; gv[dt][r] = 0.f; }

    // tile stream: n = it * ntiles + tile, image n at pack + ((it * BH + bh) * ntiles + tile) * kImgHalfs
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const int ntot = a.n_it * ntiles;
    auto tile_dma = [&](int n, int slot) {
        const int it = n / ntiles, tile = n - it * ntiles;
        const char* src = reinterpret_cast<const char*>(pack + (((int64_t)it * gridDim.y + bh) * ntiles + tile) * kImgHalfs);
        // destination = a SCALAR (the wave index through readfirstlane): kept as a vector value it was spilled, and its scratch reload at
        // the top of every tile — vector memory retires in order — waited for the dQ stores the waitcnt above deliberately leaves in flight
        const unsigned dst = (unsigned)(size_t)(lds_byte*)(Stg + slot * kImgHalfs) + __builtin_amdgcn_readfirstlane(wave) * 1024;
        lds_dma16(src + (size_t)tid * 16, dst);
        lds_dma16(src + 8192 + (size_t)tid * 16, dst + 8192);
        if (tid < 32) lds_dma16(src + 16384 + (size_t)tid * 16, dst + 16384);
    };
    // per-lane pieces of the transpose reads (constant over the tiles)
    //   dV / dK A operand (rows d = 32 dt + li, k = queries): piece row i_base + (t16 >> 2), i_base = 16 m + 4 kh (+ 8), d piece 32 dt + 16 (g16 & 1) + 4 (t16 & 3)
    //   dQ A operand (rows = queries qb + t16, k = keys): piece row j0 + (t16 >> 2), query piece (qb >> 2) + (t16 & 3)
    //   dQ B operand (cols = d db + t16, k = keys): piece row j0 + (t16 >> 2), d piece db + 4 (t16 & 3)
    const int qb = (wave >> 2) * 16, db = (wave & 3) * 16;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    // ---- dQ tile of query tile `nd` = dS K over the 256 keys of the workgroup: wave w owns the 16 x 16 block (queries 16 (w>>2).., d 16 (w&3)..).
    // Reads Ds (the dS^T image of that tile) and Ki; 24 MFMAs + 32 transpose reads, no VALU to speak of.
    auto dq_tile = [&](int nd, bool real = true) {      // real = false: the placeholder call at n == 0 (see the loop)
        if (probe & 1) return;
        f32x4v g4 = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int ja = 32 * t + 8 * g16 + (t16 >> 2), jb = ja + 4;       // keys of this lane's two pieces
            const int qp = (qb >> 2) + (t16 & 3);
            const int pa0 = ja * 32 + ((qp ^ ((ja >> 1) & 7)) << 2), pa1 = jb * 32 + ((qp ^ ((jb >> 1) & 7)) << 2);
            const int dp = db + 4 * (t16 & 3);
            const int pb0 = ja * 64 + (((dp >> 3) ^ img_swz(ja)) << 3) + (dp & 4);
            const int pb1 = jb * 64 + (((dp >> 3) ^ img_swz(jb)) << 3) + (dp & 4);
            const half8 ah = cat4(lds_tr16(Ds + pa0), lds_tr16(Ds + pa1));
            const half8 bh8 = cat4(lds_tr16(Ki + pb0), lds_tr16(Ki + pb1));
            const half8 bl8 = cat4(lds_tr16(Ki + 16384 + pb0), lds_tr16(Ki + 16384 + pb1));
            g4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh8, g4, 0, 0, 0);
            g4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl8, g4, 0, 0, 0);
            if constexpr (!kGrad1) {
                const half8 al = cat4(lds_tr16(Ds + 8192 + pa0), lds_tr16(Ds + 8192 + pa1));
                g4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh8, g4, 0, 0, 0);
            }
        }
        // accumulator: rows qb + 4 g16 + r, column db + t16
        const int itd = nd / ntiles, i0d = (nd - itd * ntiles) * 32;
        float* part = a.gq_part + (int64_t)itd * a.gqp_it + (((int64_t)bh * gridDim.x + blockIdx.x) * Lq_pad + i0d) * 64;
        if (add) {                               // scalar branch: four no-return atomics instead of four stores (same vmcnt count)
            // (the placeholder call adds zeros: as stores its values are overwritten by tile 0's real ones, as atomics they would stay)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(part + (qb + 4 * g16 + r) * 64 + db + t16, real ? g4[r] * cn * inv_os : 0.f);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(qb + 4 * g16 + r) * 64 + db + t16] = g4[r] * cn * inv_os;
        }
    };
    // which waves go first: PIPE 1 = waves 4..7, PIPE 2 = odd waves (whichever pairs share a SIMD).  Wave-uniform: a scalar branch
    const bool upper_wave = PIPE == 2 ? (__builtin_amdgcn_readfirstlane(wave) & 1) != 0 : __builtin_amdgcn_readfirstlane(wave) >= 4;
    tile_dma(0, 0);
    for (int n = 0; n < ntot; ++n) {
        const _Float16* Im = Stg + (n & 1) * kImgHalfs;
        // this thread's DMA pieces of tile n have landed (the 4 dQ stores of the previous tile were issued after them and may stay in
        // flight: VMEM operations retire in order); the barrier makes every wave's pieces visible and closes the previous tile's reads
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        lds_barrier();                                      // LDS-only barrier: the dQ stores stay in flight
        if (n + 1 < ntot) tile_dma(n + 1, (n + 1) & 1);     // the other slot was last read before the barrier above
        const float* st = reinterpret_cast<const float*>(Im + 8192);

        // PIPE: the dQ tile of the PREVIOUS query tile (its dS^T image sits in Ds since the barrier above) is computed in THIS
        // interval, by the upper four waves FIRST (then S / dP, P / dS, dV / dK) and by the lower four LAST.  A SIMD hosts waves w
        // and w + 4: shifted by one phase, one of them is in its VALU phase (softmax, dS, hi/lo splits) while the other one feeds
        // the matrix pipe (upper S/dP ‖ lower P/dS, upper P/dS ‖ lower dV/dK), instead of both running MFMA phase, VALU phase,
        // MFMA phase in lockstep.  The dV / dK products need no barrier (registers + the Q / dO image), only the dS^T image does.  No fences, no extra live
        // registers.  At n == 0 there is no previous tile: the same code runs on whatever Ds holds and its result lands in tile
        // 0's slot, which the next iteration overwrites with the real dQ of tile 0 (same lanes, stores retire in order).
        if constexpr (PIPE) {
            if (upper_wave) dq_tile(n > 0 ? n - 1 : 0, n > 0);
        }

        // ---- S = Q K^T, dP = dO V^T  (rows = queries, columns = this wave's keys)
        f32x16 sacc, pacc;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // inline-constant C of the first MFMAs
        sacc = zero16;
        pacc = zero16;
        if (!(probe & 8))
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int pos = li * 64 + (((2 * t + kh) ^ img_swz(li)) << 3);
            const half8 qh8 = *reinterpret_cast<const half8*>(Im + pos);
            const half8 ql8 = *reinterpret_cast<const half8*>(Im + 2048 + pos);
            const half8 oh8 = *reinterpret_cast<const half8*>(Im + 4096 + pos);
            const half8 ol8 = *reinterpret_cast<const half8*>(Im + 6144 + pos);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh8, kfh[t], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh8, kfl[t], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql8, kfh[t], sacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, vfh[t], pacc, 0, 0, 0);
            if constexpr (CACHE != 8) pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, vfl[t], pacc, 0, 0, 0);     // (the stage cache holds V as one fp16 value)
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ol8, vfh[t], pacc, 0, 0, 0);
        }
        // ---- P (with dropout), dS; accumulator register r is query mfma32_row(r, lane), column = this lane's key
        half8 ph[2], pl[2], sh[2], sl[2];
        {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float pv[8], dv[8];
            // row hashes of this half's 8 queries 16 m + 4 kh + {0..3, 8..11}: two 16-byte reads, then xor + compare per element
            uint32_t rh[8];
            if constexpr (DROP) {
                const uint32_t* rhs = reinterpret_cast<const uint32_t*>(st + 64) + 16 * m + 4 * kh;
                const uint4 h0 = *reinterpret_cast<const uint4*>(rhs), h1 = *reinterpret_cast<const uint4*>(rhs + 8);
                rh[0] = h0.x; rh[1] = h0.y; rh[2] = h0.z; rh[3] = h0.w; rh[4] = h1.x; rh[5] = h1.y; rh[6] = h1.z; rh[7] = h1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * m + e;
                const int qi = mfma32_row(r, lane);
                if (probe & 2) { pv[e] = sacc[r]; dv[e] = pacc[r]; continue; }
                float p = __builtin_amdgcn_exp2f(sacc[r] * c2 - st[qi]);                     // rows past Lq: lse = +inf -> 0
                if constexpr (RAGGED) p = jok ? p : 0.f;                                     // keys past Lk (launches whose Lk is not a multiple of 256)
                float keep = 1.f;
                if constexpr (DROP) keep = drop_keep_h(rh[e], drop_col, drop_thr) ? drop_inv : 0.f;
                pv[e] = p * keep;                                                            // p = 2^14 x probability (kPShift)
                dv[e] = p * (pacc[r] * (keep * kPShiftInv) - st[32 + qi]);
            }
            if constexpr (kGrad1) {
                ph[m] = cvt8_rn<kF16>(pv);             // one fp16 value each, round to nearest (v_cvt_pk_f16_f32): see kGrad1
                sh[m] = cvt8_rn<kF16>(dv);
            } else {
                split8(pv, ph[m], pl[m]);
                split8(dv, sh[m], sl[m]);
            }
        }
        }
        // the dS^T rows of this wave's keys go into Ds: without PIPE right away (nobody reads Ds between the top barrier and the
        // barrier below); with PIPE after a barrier, because every wave's dq_tile above was still reading the previous tile's image
        auto write_ds = [&]() {
#pragma unroll
            for (int m = 0; m < 2; ++m)
                // dS^T row of this key: queries 16 m + 8 hh + 4 kh + (0..3) are registers 4 hh .. 4 hh + 3 -> query piece 4 m + 2 hh + kh
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int off = jw * 32 + (((4 * m + 2 * hh + kh) ^ ((jw >> 1) & 7)) << 2);
                    *reinterpret_cast<half4v*>(Ds + off) = half4v{sh[m][4 * hh], sh[m][4 * hh + 1], sh[m][4 * hh + 2], sh[m][4 * hh + 3]};
                    if constexpr (!kGrad1)
                        *reinterpret_cast<half4v*>(Ds + 8192 + off) = half4v{sl[m][4 * hh], sl[m][4 * hh + 1], sl[m][4 * hh + 2], sl[m][4 * hh + 3]};
                }
        };
        if constexpr (!PIPE) write_ds();
        // ---- dV^T += dO^T P, dK^T += Q^T dS: contraction over the queries; A operands by transpose reads of the natural images
        if (!(probe & 4))
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int dp = 32 * dt + 16 * (g16 & 1) + 4 * (t16 & 3);
                const int r0 = 16 * m + 4 * kh + (t16 >> 2), r1 = r0 + 8;
                const int p0 = r0 * 64 + (((dp >> 3) ^ img_swz(r0)) << 3) + (dp & 4);
                const int p1 = r1 * 64 + (((dp >> 3) ^ img_swz(r1)) << 3) + (dp & 4);
                const half8 qth = cat4(lds_tr16(Im + p0), lds_tr16(Im + p1));
                const half8 qtl = cat4(lds_tr16(Im + 2048 + p0), lds_tr16(Im + 2048 + p1));
                const half8 oth = cat4(lds_tr16(Im + 4096 + p0), lds_tr16(Im + 4096 + p1));
                const half8 otl = cat4(lds_tr16(Im + 6144 + p0), lds_tr16(Im + 6144 + p1));
                gv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(oth, ph[m], gv[dt], 0, 0, 0);
                if constexpr (!kGrad1) gv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(oth, pl[m], gv[dt], 0, 0, 0);
                gv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(otl, ph[m], gv[dt], 0, 0, 0);
                gk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qth, sh[m], gk[dt], 0, 0, 0);
                if constexpr (!kGrad1) gk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qth, sl[m], gk[dt], 0, 0, 0);
                gk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qtl, sh[m], gk[dt], 0, 0, 0);
                }
        if constexpr (!PIPE) {
            lds_barrier();
            dq_tile(n);
        } else {
            if (!upper_wave) dq_tile(n > 0 ? n - 1 : 0, n > 0);    // the lower waves' turn (their P / dS operands are dead by now)
            lds_barrier();                                   // every wave has read the previous tile's dS^T image
            write_ds();                                      // visible to the next interval through the barrier at the top of the loop
        }
    }
    if constexpr (PIPE) {
        lds_barrier();                                      // the last tile's dS^T image is complete
        dq_tile(ntot - 1);
    }
    // ---- dK, dV of this wave's keys: (d x keys) accumulators -> [key][d] through LDS (Ds + head of Ki, free now), row-contiguous stores
    __syncthreads();
    float* tr = reinterpret_cast<float*>(Ds) + wave * (32 * 65);
    float kvmax = 0.f;
    for (int which = 0; which < 2; ++which) {
        const float scale = (which == 0 ? cn : kPShiftInv) * inv_os;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[li * 65 + dt * 32 + mfma32_row(r, lane)] = (which == 0 ? gk[dt][r] : gv[dt][r]) * scale;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float* g = which == 0 ? a.gk + (int64_t)b * a.gk_batch + (int64_t)h * a.gk_head : a.gv + (int64_t)b * a.gv_batch + (int64_t)h * a.gv_head;
        const int64_t grow = which == 0 ? a.gk_row : a.gv_row;
        const int j0 = kbx * kSpKW + wave * 32;
        for (int idx = lane; idx < 32 * 64; idx += 64) {
            const int jj = idx >> 6, d = idx & 63;
            if (j0 + jj < a.Lk) {
                float* o = g + (int64_t)(j0 + jj) * grow + d;
                const float val = (a.accumulate_kv ? *o : 0.f) + tr[jj * 65 + d];
                *o = val;
                kvmax = fmaxf(kvmax, fabsf(val));
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    if (a.kv_absmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kvmax = fmaxf(kvmax, __shfl_xor(kvmax, o));
        if (lane == 0) atomicMax(a.kv_absmax, __float_as_uint(kvmax));
    }
}

// A workgroup works through a.kb_group CONSECUTIVE key blocks and keeps ONE slot of dQ partials for all of them: the first block stores its
// partial tiles, the later ones add theirs with no-return float atomics (the slot belongs to this workgroup alone: no contention, a fixed
// order, the same vmcnt bookkeeping as stores).  The partial array — written by this kernel and read back by attn_bwd_dq_reduce_kernel — shrinks
// by that factor: at BASELINE cfg 4's shard (4 scenes, 8 iterations) from 6.3 GB to 1.6 GB per step, the reduction from 1.05 to ~0.3 ms.
template <bool DROP, bool RAGGED, int PIPE = 1, int CACHE = 0>
__global__ __launch_bounds__(512) void attn_bwd_split2_kernel(AttnBwdArgs a, const float* __restrict__ oscale_ptr,
                                                              const _Float16* __restrict__ pack) {
    const float oscale = *oscale_ptr;
    const int nkb = (a.Lk + kSpKW - 1) / kSpKW;
    for (int pass = 0; pass < a.kb_group; ++pass) {
        const int kbx = (int)blockIdx.x * a.kb_group + pass;
        if (kbx >= nkb) break;
        if (pass > 0) __syncthreads();          // the next block rewrites the K image and the staging slots
        attn_bwd_split2_block<DROP, RAGGED, PIPE, CACHE>(a, oscale, pack, kbx, pass > 0);
    }
}

// scale[0] = 2^(10 - exponent(max |dO|)) from the bit pattern left by absmax_kernel (1 when the gradient is all zero)
__global__ void oscale_kernel(const unsigned int* __restrict__ bits, float* __restrict__ scale) {
    const float mx = __uint_as_float(bits[0]);
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) frexpf(mx, &ex);
    scale[0] = ldexpf(1.f, 10 - ex);
}

// max |x| over a buffer -> *out (float bits via atomicMax on the non-negative pattern); out must be zeroed
__global__ void absmax_kernel(const float* __restrict__ x, int64_t n, unsigned int* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// gq[b][i][h*64 + d] += sum_kb part[bh][kb][i][d]
__global__ __launch_bounds__(256) void attn_bwd_dq_reduce_kernel(const float* __restrict__ part, int nkb, int Lq, int Lq_pad, int H,
                                                                 float* __restrict__ gq, int64_t gq_batch, int64_t gq_head, int64_t gq_row,
                                                                 int64_t part_it, int64_t gq_it) {
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    part += (int64_t)blockIdx.z * part_it;                  // recurrent iteration (batched backward)
    gq += (int64_t)blockIdx.z * gq_it;
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (i, d)
    const int i = idx >> 6, d = idx & 63;
    if (i >= Lq) return;
    const float* p = part + ((int64_t)bh * nkb * Lq_pad + i) * 64 + d;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int kb = 0;
    for (; kb + 4 <= nkb; kb += 4) {
        s0 += p[(int64_t)kb * Lq_pad * 64];
        s1 += p[(int64_t)(kb + 1) * Lq_pad * 64];
        s2 += p[(int64_t)(kb + 2) * Lq_pad * 64];
        s3 += p[(int64_t)(kb + 3) * Lq_pad * 64];
    }
    for (; kb < nkb; ++kb) s0 += p[(int64_t)kb * Lq_pad * 64];
    gq[(int64_t)b * gq_batch + (int64_t)h * gq_head + (int64_t)i * gq_row + d] += (s0 + s1) + (s2 + s3);
}

// D[bh][i] = sum_d dO[i][h*dh + d] * O[i][h*dh + d]; one wave per (bh, i)
__global__ __launch_bounds__(256) void attn_bwd_rowdot_kernel(const float* __restrict__ dO, const float* __restrict__ O, int64_t batch,
                                                              int64_t row, int BH, int H, int Lq, int dh, float* __restrict__ D) {
    const int Lq_pad = (Lq + 31) & ~31;
    const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bh = (int)(idx / Lq), i = (int)(idx - (int64_t)bh * Lq);
    if (bh >= BH) return;
    const int b = bh / H, h = bh - b * H;
    float s = 0.f;
    for (int d = lane; d < dh; d += 64) {
        const int64_t o = (int64_t)b * batch + (int64_t)i * row + h * dh + d;
        s += dO[o] * O[o];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) D[(int64_t)bh * Lq_pad + i] = s;
}

}  // namespace

// ================================================================================================ any head dim: materialised path
// Attention backward for head dims the register-resident kernels do not cover (e.g. 256 = the reference's shipped size): per
// (scene, head) the score matrix is materialised in global scratch, as the reference itself does, and every product is one of the
// chain's generic fp32 GEMMs:  S = Q K^T,  P = exp2(S c - lse),  dV += Pd^T dO,  dP = dO V^T,  dS = P (dP keep/(1-p) - D) / sqrt(dh),
// dQ += dS K,  dK += dS^T Q   (Pd = P with the dropout mask).  Functional, not fast: ~10 passes over Lq x Lk floats per head.
__global__ void attn_mat_probs_kernel(float* __restrict__ S, float* __restrict__ Pd, const float* __restrict__ lse, int Lq, int Lk, int Lkp,
                                      float c2, float drop_p, uint32_t seed, uint32_t row0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)Lq * Lkp) return;
    const int i = (int)(idx / Lkp), j = (int)(idx - (int64_t)i * Lkp);
    float p = 0.f;
    if (j < Lk) p = __builtin_amdgcn_exp2f(S[idx] * c2 - lse[i]);
    S[idx] = p;
    if (Pd) Pd[idx] = (j < Lk && drop_keep(drop_rowhash(seed, row0 + (uint32_t)i), (uint32_t)j, drop_p)) ? p / (1.f - drop_p) : 0.f;
}
__global__ void attn_mat_ds_kernel(const float* __restrict__ P, float* __restrict__ dP, const float* __restrict__ D, int Lq, int Lk, int Lkp,
                                   float cn, float drop_p, uint32_t seed, uint32_t row0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)Lq * Lkp) return;
    const int i = (int)(idx / Lkp), j = (int)(idx - (int64_t)i * Lkp);
    float g = 0.f;
    if (j < Lk) {
        float dp = dP[idx];
        if (drop_p > 0.f) dp = drop_keep(drop_rowhash(seed, row0 + (uint32_t)i), (uint32_t)j, drop_p) ? dp / (1.f - drop_p) : 0.f;
        g = P[idx] * (dp - D[i]) * cn;
    }
    dP[idx] = g;
}

static LinearArgs mat_lin(const float* X, int64_t ldx, const float* W, int64_t ldw, float* Y, int64_t ldy, int M, int N, int K) {
    LinearArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.Y = Y; a.M = M; a.N = N; a.K = K;
    a.rows_per_batch = M; a.y_batch = 0; a.y_row = ldy; a.col_blk = N; a.y_blk = 0;
    return a;
}

size_t attn_bwd_mat_scratch_floats(int Lq, int Lk, int dh) {
    const int64_t Lkp = (Lk + 15) & ~15;
    return (size_t)(2 * (int64_t)Lq * Lkp + (int64_t)dh * Lkp);
}

static hipError_t attn_bwd_materialised(const AttnBwdArgs& a, int dh, float* scratch, hipStream_t s) {
    if (dh % 16 != 0 || !scratch) return hipErrorInvalidValue;
    const int Lq = a.Lq, Lk = a.Lk, Lkp = (Lk + 15) & ~15, Lq_pad = (Lq + 31) & ~31;
    float* Pb = scratch;                                     // [Lq][Lkp]  S, then P
    float* Gb = Pb + (int64_t)Lq * Lkp;                      // [Lq][Lkp]  Pd, then dP, then dS
    float* KT = Gb + (int64_t)Lq * Lkp;                      // [dh][Lkp]
    const float c2 = 1.4426950408889634f / sqrtf((float)dh), cn = 1.f / sqrtf((float)dh);
    const unsigned eb = (unsigned)ceil_div64((int64_t)Lq * Lkp, 256);
    hipError_t e;
    for (int b = 0; b < a.B; ++b)
        for (int h = 0; h < a.H; ++h) {
            const int bh = b * a.H + h;
            const float* q = a.q + (int64_t)b * a.q_batch + (int64_t)h * a.q_head;
            const float* k = a.k + (int64_t)b * a.k_batch + (int64_t)h * a.k_head;
            const float* v = a.v + (int64_t)b * a.v_batch + (int64_t)h * a.v_head;
            const float* dO = a.dO + (int64_t)b * a.do_batch + (int64_t)h * a.do_head;
            float* gq = a.gq + (int64_t)b * a.gq_batch + (int64_t)h * a.gq_head;
            float* gk = a.gk + (int64_t)b * a.gk_batch + (int64_t)h * a.gk_head;
            float* gv = a.gv + (int64_t)b * a.gv_batch + (int64_t)h * a.gv_head;
            const uint32_t row0 = (uint32_t)(bh * Lq);
            LinearArgs l = mat_lin(q, a.q_row, k, a.k_row, Pb, Lkp, Lq, Lk, dh);                 // S = Q K^T
            if ((e = launch_linear(l, 1, s)) != hipSuccess) return e;
            const bool drop = a.drop_p > 0.f;
            hipLaunchKernelGGL(attn_mat_probs_kernel, dim3(eb), dim3(256), 0, s, Pb, drop ? Gb : nullptr, a.lse + (int64_t)bh * Lq_pad, Lq, Lk,
                               Lkp, c2, a.drop_p, a.drop_seed, row0);
            if ((e = launch_gemm_tn(drop ? Gb : Pb, Lkp, dO, a.do_row, gv, a.gv_row, Lq, Lk, dh, a.accumulate_kv, s)) != hipSuccess) return e;
            l = mat_lin(dO, a.do_row, v, a.v_row, Gb, Lkp, Lq, Lk, dh);                           // dP = dO V^T
            if ((e = launch_linear(l, 1, s)) != hipSuccess) return e;
            hipLaunchKernelGGL(attn_mat_ds_kernel, dim3(eb), dim3(256), 0, s, Pb, Gb, a.D + (int64_t)bh * Lq_pad, Lq, Lk, Lkp, cn, a.drop_p,
                               a.drop_seed, row0);
            if ((e = hipMemsetAsync(KT, 0, (size_t)dh * Lkp * sizeof(float), s)) != hipSuccess) return e;
            if ((e = launch_transpose(k, a.k_row, KT, Lkp, Lk, dh, s)) != hipSuccess) return e;
            l = mat_lin(Gb, Lkp, KT, Lkp, gq, a.gq_row, Lq, dh, Lkp);                              // dQ += dS K
            l.R = gq; l.ldr = a.gq_row;
            if ((e = launch_linear(l, 1, s)) != hipSuccess) return e;
            if ((e = launch_gemm_tn(Gb, Lkp, q, a.q_row, gk, a.gk_row, Lq, Lk, dh, a.accumulate_kv, s)) != hipSuccess) return e;   // dK += dS^T Q
        }
    return hipGetLastError();
}

constexpr size_t kSplitBwdLds = (size_t)(8 * 2048 + 2 * 8192 + 2 * 16384) * sizeof(_Float16) + 96 * sizeof(float);
static hipError_t split_bwd_lds_attr() {
    static DynLdsOnce once_a, once_b;
    hipError_t e = once_a.ensure(reinterpret_cast<const void*>(&attn_bwd_split_kernel<false>), kSplitBwdLds);
    if (e == hipSuccess) e = once_b.ensure(reinterpret_cast<const void*>(&attn_bwd_split_kernel<true>), kSplitBwdLds);
    return e;
}

// q/k/v/dO/g*: element strides (batch, head, row); lse, D: [B*H][pad32(Lq)].  gq must be zeroed by the caller.
hipError_t launch_attn_bwd(const float* q, int64_t q_batch, int64_t q_head, int64_t q_row, const float* k, int64_t k_batch,
                           int64_t k_head, int64_t k_row, const float* v, int64_t v_batch, int64_t v_head, int64_t v_row,
                           const float* dO, int64_t do_batch, int64_t do_head, int64_t do_row, const float* lse, const float* D,
                           float* gq, int64_t gq_batch, int64_t gq_head, int64_t gq_row, float* gk, int64_t gk_batch,
                           int64_t gk_head, int64_t gk_row, float* gv, int64_t gv_batch, int64_t gv_head, int64_t gv_row, int B, int H,
                           int Lq, int Lk, int dh, int accumulate_kv, hipStream_t s, float* gq_part, float drop_p, uint32_t drop_seed,
                           unsigned int* absmax, float* mat_scratch) {
    if (dh != 64 && dh != 32 && !mat_scratch) return hipErrorInvalidValue;
    AttnBwdArgs a;
    a.probe = 0;
    a.q = q; a.q_batch = q_batch; a.q_head = q_head; a.q_row = q_row;
    a.k = k; a.k_batch = k_batch; a.k_head = k_head; a.k_row = k_row;
    a.v = v; a.v_batch = v_batch; a.v_head = v_head; a.v_row = v_row;
    a.dO = dO; a.do_batch = do_batch; a.do_head = do_head; a.do_row = do_row;
    a.lse = lse; a.D = D;
    a.gq = gq; a.gq_batch = gq_batch; a.gq_head = gq_head; a.gq_row = gq_row;
    a.gk = gk; a.gk_batch = gk_batch; a.gk_head = gk_head; a.gk_row = gk_row;
    a.gv = gv; a.gv_batch = gv_batch; a.gv_head = gv_head; a.gv_row = gv_row;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.accumulate_kv = accumulate_kv; a.gq_part = nullptr;
    a.drop_p = drop_p; a.drop_seed = drop_seed;
    a.n_it = 1; a.do_it = a.D_it = a.gqp_it = 0; a.kv_absmax = nullptr; a.kb_group = 1;
    memset(a.q_off, 0, sizeof(a.q_off));
    memset(a.lse_off, 0, sizeof(a.lse_off));
    for (int t = 0; t < kMaxBwdIters; ++t) a.seeds[t] = drop_seed;
    if (dh != 64 && dh != 32) return attn_bwd_materialised(a, dh, mat_scratch, s);
    dim3 grid(ceil_div(Lk, 256), B * H);
    static const int force = [] {
        const char* e = dev_env("PARQ_ATTN_BWD");            // "naive" / "mfma" (exact fp32 MFMA): debugging overrides of the split kernel
        return e ? (e[0] == 'n' ? 1 : 2) : 0;
    }();
    if (dh == 64 && force == 0 && Lk >= 2048 && absmax) {
        // split-precision kernel: dO scaled by a power of two so that max |dO| sits near 2^10 (keeps the low halves of the small
        // gradient values out of the fp16 subnormals); the scale is computed and consumed on the device (absmax[0] bits, absmax[1] scale)
        hipError_t e = hipMemsetAsync(absmax, 0, sizeof(unsigned int), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, s, dO, (int64_t)B * do_batch, absmax);
        float* oscale = reinterpret_cast<float*>(absmax + 1);
        hipLaunchKernelGGL(oscale_kernel, dim3(1), dim3(1), 0, s, absmax, oscale);
        const size_t lds = kSplitBwdLds;
        e = split_bwd_lds_attr();
        if (e != hipSuccess) return e;
        dim3 g2(ceil_div(Lk, kSpKW), B * H);
        a.gq_part = gq_part;
        if (drop_p > 0.f) hipLaunchKernelGGL(attn_bwd_split_kernel<true>, g2, dim3(512), lds, s, a, oscale);
        else hipLaunchKernelGGL(attn_bwd_split_kernel<false>, g2, dim3(512), lds, s, a, oscale);
        if (a.gq_part) {
            const int Lq_pad = (Lq + 31) & ~31;
            hipLaunchKernelGGL(attn_bwd_dq_reduce_kernel, dim3(ceil_div(Lq * 64, 256), B * H), dim3(256), 0, s, gq_part, (int)g2.x, Lq,
                               Lq_pad, H, gq, gq_batch, gq_head, gq_row, (int64_t)0, (int64_t)0);
        }
        return hipGetLastError();
    }
    if (dh == 64 && force != 1) {
        const bool big = Lk >= 2048;
        const int nwv = big ? 8 : 2;
        const int KW = nwv * 32;
        const size_t lds = (size_t)(2 * 32 * kQs + 64 + 32 * (KW + 4) + KW * kKc) * sizeof(float);
        static DynLdsOnce once8, once2;
        if (hipError_t e = (big ? once8 : once2).ensure(big ? reinterpret_cast<const void*>(&attn_bwd_mfma_kernel<8>)
                                                             : reinterpret_cast<const void*>(&attn_bwd_mfma_kernel<2>), lds);
            e != hipSuccess) return e;
        dim3 g2(ceil_div(Lk, KW), B * H);
        a.gq_part = big ? gq_part : nullptr;
        if (big) hipLaunchKernelGGL(attn_bwd_mfma_kernel<8>, g2, dim3(512), lds, s, a);
        else hipLaunchKernelGGL(attn_bwd_mfma_kernel<2>, g2, dim3(128), lds, s, a);
        if (a.gq_part) {
            const int Lq_pad = (Lq + 31) & ~31;
            hipLaunchKernelGGL(attn_bwd_dq_reduce_kernel, dim3(ceil_div(Lq * 64, 256), B * H), dim3(256), 0, s, gq_part, (int)g2.x, Lq,
                               Lq_pad, H, gq, gq_batch, gq_head, gq_row, (int64_t)0, (int64_t)0);
        }
        return hipGetLastError();
    }
    if (dh == 64) {
        const size_t lds = (size_t)(2 * 32 * 64 + 32 * 257 + 256 * 65 + 64) * sizeof(float);
        static DynLdsOnce once;
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&attn_bwd_kernel<64>), lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(attn_bwd_kernel<64>, grid, dim3(256), lds, s, a);
    } else {
        const size_t lds = (size_t)(2 * 32 * 32 + 32 * 257 + 256 * 33 + 64) * sizeof(float);
        static DynLdsOnce once;
        if (hipError_t e = once.ensure(reinterpret_cast<const void*>(&attn_bwd_kernel<32>), lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(attn_bwd_kernel<32>, grid, dim3(256), lds, s, a);
    }
    return hipGetLastError();
}

// ================================================================================================ head dim 256, batched iterations
// The reference's shipped decoder size (DEC_DIM 1024 / 4 heads).  A register-resident kernel like attn_bwd_split2_kernel does not fit
// a 256-deep contraction, so the backward of all iterations that share K / V is composed, per (scene, head), from the split-precision
// GEMMs the library already has (fp16 hi/lo operands, 3-term products: fp32-class), with the scores kept TRANSPOSED (rows = keys) so
// that four of the five products are "tall matrix x small weight" launches of gemm_split_kernel (raype.hip):
//     S^T  [Lk][R] = K  [Lk][256] . Q_all^T        R = n_it * Lq query rows of all iterations (padded to 512: QP)
//     dP^T [Lk][R] = V  [Lk][256] . (s dO_all)^T   s = the power-of-two gradient scale of the split kernels
//     P', dS' elementwise in place (P' = 2^14 P with dropout, dS' = s dS: same staging as attn_bwd_split2_kernel)
//     dV [Lk][256] = P'^T-rows . dO_all   dK [Lk][256] = dS'^T-rows . Q_all        (contraction over the R query rows)
//     dQ_all [R][256] = sum over keys dS'[key][r] K[key][d]: the 512 x 256 TN kernel of the K/V-projection backward, per 512-row slice
// 3.1 GB of scratch at cfg 3 (two [Lk][2048] fp32 matrices), reused for every (scene, head).
struct Bwd256Args {
    const float* q; int64_t q_batch, q_head, q_row;
    const float* dO; int64_t do_it, do_batch, do_head, do_row;
    const float* lse; const float* D; int64_t D_it;
    float* gq; int64_t gq_it, gq_batch, gq_head, gq_row;
    int64_t q_off[kMaxBwdIters], lse_off[kMaxBwdIters];
    uint32_t seeds[kMaxBwdIters];
    int n_it, Lq, Lq_pad, Lk, QP, H;
    float drop_p;
};

// fp16 hi/lo planes of Q_all and s dO_all, row-major [QP][256] and transposed [256][QP]; rows past n_it * Lq are zero
__global__ __launch_bounds__(256) void bwd256_planes_kernel(Bwd256Args a, int b, int h, const float* __restrict__ oscale_ptr,
                                                            _Float16* __restrict__ planes) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.QP * 256) return;
    const int r = (int)(idx >> 8), d = (int)(idx & 255);
    float qv = 0.f, ov = 0.f;
    if (r < a.n_it * a.Lq) {
        const int it = r / a.Lq, i = r - it * a.Lq;
        qv = a.q[a.q_off[it] + (int64_t)b * a.q_batch + (int64_t)h * a.q_head + (int64_t)i * a.q_row + d];
        ov = a.dO[(int64_t)it * a.do_it + (int64_t)b * a.do_batch + (int64_t)h * a.do_head + (int64_t)i * a.do_row + d] * *oscale_ptr;
    }
    half2v qh, ql, oh, ol;
    split_pair(qv, 0.f, qh, ql);
    split_pair(ov, 0.f, oh, ol);
    const int64_t P = (int64_t)a.QP * 256;                   // one plane
    planes[0 * P + idx] = qh[0]; planes[1 * P + idx] = ql[0];          // Q   [QP][256]
    planes[2 * P + idx] = oh[0]; planes[3 * P + idx] = ol[0];          // dO  [QP][256]
    const int64_t t = (int64_t)d * a.QP + r;
    planes[4 * P + t] = qh[0]; planes[5 * P + t] = ql[0];              // Q^T  [256][QP]
    planes[6 * P + t] = oh[0]; planes[7 * P + t] = ol[0];              // dO^T [256][QP]
}

// in place: ST -> P' (2^14 x probability, dropout applied), dPT -> dS' (gradient scale s still on it); max |dS'| for the TN kernel's range
__global__ __launch_bounds__(256) void bwd256_elem_kernel(float* __restrict__ ST, float* __restrict__ dPT, Bwd256Args a, int bh,
                                                          const float* __restrict__ oscale_ptr, float c2, unsigned int* __restrict__ ds_absmax) {
    const int64_t n = (int64_t)a.Lk * a.QP;
    float mx = 0.f;
    const float os = *oscale_ptr;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx / a.QP), r = (int)(idx - (int64_t)j * a.QP);
        float p = 0.f, ds = 0.f;
        if (r < a.n_it * a.Lq) {
            const int it = r / a.Lq, i = r - it * a.Lq;
            const float lse = a.lse[a.lse_off[it] + (int64_t)bh * a.Lq_pad + i];
            const float Dv = a.D[(int64_t)it * a.D_it + (int64_t)bh * a.Lq_pad + i];
            p = __builtin_amdgcn_exp2f(ST[idx] * c2 - (lse - kPShift));
            float keep = 1.f;
            if (a.drop_p > 0.f)
                keep = drop_keep(drop_rowhash(a.seeds[it], (uint32_t)(bh * a.Lq + i)), (uint32_t)j, a.drop_p) ? 1.f / (1.f - a.drop_p) : 0.f;
            ds = p * (dPT[idx] * (keep * kPShiftInv) - Dv * os * kPShiftInv);
            p *= keep;
        }
        ST[idx] = p;
        dPT[idx] = ds;
        mx = fmaxf(mx, fabsf(ds));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) atomicMax(ds_absmax, __float_as_uint(mx));
}

// gq[it][b][i][h * 256 + d] += dQ_all[r][d] * cn / s
__global__ __launch_bounds__(256) void bwd256_dq_scatter_kernel(const float* __restrict__ dq_all, Bwd256Args a, int b, int h,
                                                                const float* __restrict__ oscale_ptr, float cn) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.n_it * a.Lq * 256) return;
    const int r = (int)(idx >> 8), d = (int)(idx & 255);
    const int it = r / a.Lq, i = r - it * a.Lq;
    a.gq[(int64_t)it * a.gq_it + (int64_t)b * a.gq_batch + (int64_t)h * a.gq_head + (int64_t)i * a.gq_row + d] += dq_all[idx] * (cn / *oscale_ptr);
}

static int bwd256_qp(int n_it, int Lq) { return (n_it * Lq + 511) & ~511; }

size_t attn_bwd_batched256_scratch_floats(int n_it, int Lq, int Lk) {
    const int64_t QP = bwd256_qp(n_it, Lq);
    return (size_t)(2 * (int64_t)Lk * QP + 1024      // S^T / P' and dP^T / dS' (+ slack: the TN kernel reads whole 512-column slices)
                    + QP * 1024                      // 8 fp16 planes of QP x 256
                    + QP * 256 + QP + 16);           // dQ_all, a column-sum dummy, scalars
}

static hipError_t attn_bwd_batched_256(const AttnBwdArgs& a, int64_t gq_it, const float* oscale, float* scratch, hipStream_t s) {
    const int Lq = a.Lq, Lk = a.Lk, QP = bwd256_qp(a.n_it, Lq);
    Bwd256Args g;
    g.q = a.q; g.q_batch = a.q_batch; g.q_head = a.q_head; g.q_row = a.q_row;
    g.dO = a.dO; g.do_it = a.do_it; g.do_batch = a.do_batch; g.do_head = a.do_head; g.do_row = a.do_row;
    g.lse = a.lse; g.D = a.D; g.D_it = a.D_it;
    g.gq = a.gq; g.gq_it = gq_it; g.gq_batch = a.gq_batch; g.gq_head = a.gq_head; g.gq_row = a.gq_row;
    for (int t = 0; t < kMaxBwdIters; ++t) { g.q_off[t] = a.q_off[t]; g.lse_off[t] = a.lse_off[t]; g.seeds[t] = a.seeds[t]; }
    g.n_it = a.n_it; g.Lq = Lq; g.Lq_pad = (Lq + 31) & ~31; g.Lk = Lk; g.QP = QP; g.H = a.H; g.drop_p = a.drop_p;
    float* ST = scratch;
    float* dPT = ST + (int64_t)Lk * QP;
    _Float16* planes = reinterpret_cast<_Float16*>(dPT + (int64_t)Lk * QP + 1024);
    float* dq_all = reinterpret_cast<float*>(planes) + (int64_t)QP * 1024;
    float* dummy = dq_all + (int64_t)QP * 256;
    unsigned int* ds_absmax = reinterpret_cast<unsigned int*>(dummy + QP);
    float* tn_scale = reinterpret_cast<float*>(ds_absmax + 1);
    const int64_t P = (int64_t)QP * 256;
    const float c2 = 1.4426950408889634f / 16.f, cn = 1.f / 16.f;
    hipError_t e;
    for (int b = 0; b < a.B; ++b)
        for (int h = 0; h < a.H; ++h) {
            const int bh = b * a.H + h;
            const float* k = a.k + (int64_t)b * a.k_batch + (int64_t)h * a.k_head;
            const float* v = a.v + (int64_t)b * a.v_batch + (int64_t)h * a.v_head;
            float* gk = a.gk + (int64_t)b * a.gk_batch + (int64_t)h * a.gk_head;
            float* gv = a.gv + (int64_t)b * a.gv_batch + (int64_t)h * a.gv_head;
            hipLaunchKernelGGL(bwd256_planes_kernel, dim3((unsigned)ceil_div64(P, 256)), dim3(256), 0, s, g, b, h, oscale, planes);
            if ((e = launch_gemm_split(k, a.k_row, planes + 0 * P, planes + 1 * P, nullptr, ST, QP, Lk, QP, 256, 0, nullptr, 1, s)) != hipSuccess) return e;
            if ((e = launch_gemm_split(v, a.v_row, planes + 2 * P, planes + 3 * P, nullptr, dPT, QP, Lk, QP, 256, 0, nullptr, 1, s)) != hipSuccess) return e;
            if ((e = hipMemsetAsync(ds_absmax, 0, sizeof(unsigned int), s)) != hipSuccess) return e;
            hipLaunchKernelGGL(bwd256_elem_kernel, dim3(4096), dim3(256), 0, s, ST, dPT, g, bh, oscale, c2, ds_absmax);
            // dV = P'^T dO' / (2^14 s),  dK = dS'^T Q cn / s
            if ((e = launch_gemm_split(ST, QP, planes + 6 * P, planes + 7 * P, nullptr, gv, a.gv_row, Lk, 256, QP, 0, nullptr, 1, s, oscale, kPShiftInv)) != hipSuccess) return e;
            if ((e = launch_gemm_split(dPT, QP, planes + 4 * P, planes + 5 * P, nullptr, gk, a.gk_row, Lk, 256, QP, 0, nullptr, 1, s, oscale, cn)) != hipSuccess) return e;
            // dQ_all = dS'^T-contraction with K, 512 query rows per launch
            if ((e = hipMemsetAsync(dq_all, 0, (size_t)QP * 256 * sizeof(float), s)) != hipSuccess) return e;
            for (int r0 = 0; r0 < QP; r0 += 512)
                if ((e = launch_tn_split_512x256(dPT + r0, QP, k, a.k_row, Lk, dq_all + (int64_t)r0 * 256, 256, nullptr, ds_absmax, tn_scale, s)) != hipSuccess) return e;
            hipLaunchKernelGGL(bwd256_dq_scatter_kernel, dim3((unsigned)ceil_div64((int64_t)a.n_it * Lq * 256, 256)), dim3(256), 0, s, dq_all, g, b, h, oscale, cn);
        }
    return hipGetLastError();
}

// Cross-attention backward of n_it recurrent iterations that share K / V (hoisted projection, shared layer weights) in ONE launch of
// the split-precision kernel: dK / dV are written once instead of read-modify-written per iteration, and the K / V prologue and the
// transposing epilogue are paid once per key block instead of once per iteration.  q / lse of iteration t live at q + q_off[t] /
// lse + lse_off[t] (activation stash); dO, D, the dQ partials and gq are arrays over the iterations with strides do_it, D_it,
// attn_bwd_dq_partial_floats(...) and gq_it.  gk / gv are overwritten.  absmax: 8-byte device scratch.
hipError_t launch_attn_bwd_batched(const float* q, const int64_t* q_off, int64_t q_batch, int64_t q_head, int64_t q_row, const float* k,
                                   int64_t k_batch, int64_t k_head, int64_t k_row, const float* v, int64_t v_batch, int64_t v_head,
                                   int64_t v_row, const float* dO, int64_t do_it, int64_t do_batch, int64_t do_head, int64_t do_row,
                                   const float* lse, const int64_t* lse_off, const float* D, int64_t D_it, float* gq, int64_t gq_it,
                                   int64_t gq_batch, int64_t gq_head, int64_t gq_row, float* gk, int64_t gk_batch, int64_t gk_head,
                                   int64_t gk_row, float* gv, int64_t gv_batch, int64_t gv_head, int64_t gv_row, int B, int H, int Lq,
                                   int Lk, int dh, int n_it, hipStream_t s, float* gq_part, float drop_p, const uint32_t* seeds,
                                   unsigned int* absmax, unsigned int* kv_absmax, void* pack, float* mat_scratch, const void* kvcache,
                                   int cache_terms, int cache_kind) {
    if (n_it < 1 || n_it > kMaxBwdIters || !absmax) return hipErrorInvalidValue;
    if (dh == 256 ? !mat_scratch : (dh != 64 || Lk < 2048 || !gq_part)) return hipErrorInvalidValue;
    AttnBwdArgs a;
    a.probe = 0;
    a.q = q; a.q_batch = q_batch; a.q_head = q_head; a.q_row = q_row;
    a.k = k; a.k_batch = k_batch; a.k_head = k_head; a.k_row = k_row;
    a.v = v; a.v_batch = v_batch; a.v_head = v_head; a.v_row = v_row;
    a.dO = dO; a.do_batch = do_batch; a.do_head = do_head; a.do_row = do_row;
    a.lse = lse; a.D = D;
    a.gq = gq; a.gq_batch = gq_batch; a.gq_head = gq_head; a.gq_row = gq_row;
    a.gk = gk; a.gk_batch = gk_batch; a.gk_head = gk_head; a.gk_row = gk_row;
    a.gv = gv; a.gv_batch = gv_batch; a.gv_head = gv_head; a.gv_row = gv_row;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.accumulate_kv = 0; a.gq_part = gq_part;
    a.drop_p = drop_p; a.drop_seed = seeds ? seeds[0] : 0;
    a.n_it = n_it; a.do_it = do_it; a.D_it = D_it; a.kv_absmax = kv_absmax;
    a.kvcache = reinterpret_cast<const _Float16*>(kvcache); a.cache_kind = cache_kind;
    if (kv_absmax) {
        hipError_t e0 = hipMemsetAsync(kv_absmax, 0, sizeof(unsigned int), s);
        if (e0 != hipSuccess) return e0;
    }
    for (int t = 0; t < kMaxBwdIters; ++t) {
        a.q_off[t] = t < n_it ? q_off[t] : 0;
        a.lse_off[t] = t < n_it ? lse_off[t] : 0;
        a.seeds[t] = (t < n_it && seeds) ? seeds[t] : 0;
    }
    // one power-of-two scale for all iterations: max |dO| over the whole [n_it] array
    hipError_t e = hipMemsetAsync(absmax, 0, sizeof(unsigned int), s);
    if (e != hipSuccess) return e;
    for (int t = 0; t < n_it; ++t)
        hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, s, dO + (int64_t)t * do_it, (int64_t)B * do_batch, absmax);
    float* oscale = reinterpret_cast<float*>(absmax + 1);
    hipLaunchKernelGGL(oscale_kernel, dim3(1), dim3(1), 0, s, absmax, oscale);
    if (dh == 256) return attn_bwd_batched_256(a, gq_it, oscale, mat_scratch, s);     // (kv_absmax: taken by the caller over its dK | dV buffer)
    static const bool v1 = [] { const char* e = dev_env("PARQ_ATTN_BWD_V"); return e && e[0] == '1'; }();
    // second version: a workgroup takes kb_group consecutive key blocks and keeps one slot of dQ partials for them (see the kernel)
    if (v1 || !pack) return hipErrorNotSupported;          // (the first-version kernel wants one partial slot per key block: the workspace no longer has them)
    const bool grouped = true;
    a.kb_group = grouped ? attn_bwd_kb_group(B, H, Lk) : 1;
    a.gqp_it = (int64_t)attn_bwd_dq_partial_floats(B, H, Lq, Lk, dh, grouped);
    dim3 g2(ceil_div(ceil_div(Lk, kSpKW), a.kb_group), B * H);
    const int Lq_pad = (Lq + 31) & ~31;
    if (kvcache && (v1 || !pack)) return hipErrorNotSupported;          // only the second-version kernel reads the 16-bit cache
    if (pack && !v1) {
        // second version: tile images packed once, fetched by LDS-DMA; transpose reads
        constexpr size_t lds2 = (size_t)(2 * 8192 + 2 * 16384 + 2 * kImgHalfs) * sizeof(_Float16);
        const bool rag = (Lk % kSpKW) != 0;
        static const int pipe = [] { const char* e = dev_env("PARQ_ATTN_BWD_PIPE"); return e ? atoi(e) : 1; }();
        if (kvcache && pipe != 1) return hipErrorNotSupported;
        _Float16* pk = reinterpret_cast<_Float16*>(pack);
        static const int probe = [] { const char* e = dev_env("PARQ_ATTN_BWD_PROBE"); return e ? atoi(e) : 0; }();
        a.probe = probe;
        hipLaunchKernelGGL(attn_bwd_pack_kernel, dim3(Lq_pad / 32, B * H, n_it), dim3(256), 0, s, a, oscale, pk);
#define PARQ_BWD2(DROP_, RAG_, PIPE_)                                                                                          \
        {                                                                                                                      \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_split2_kernel<DROP_, RAG_, PIPE_>),                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
            if (e != hipSuccess) return e;                                                                                     \
            hipLaunchKernelGGL((attn_bwd_split2_kernel<DROP_, RAG_, PIPE_>), g2, dim3(512), lds2, s, a, oscale, pk);           \
        }
#define PARQ_BWD2C(DROP_, RAG_, C_)                                                                                           \
        {                                                                                                                      \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_split2_kernel<DROP_, RAG_, 1, C_>),                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
            if (e != hipSuccess) return e;                                                                                     \
            hipLaunchKernelGGL((attn_bwd_split2_kernel<DROP_, RAG_, 1, C_>), g2, dim3(512), lds2, s, a, oscale, pk);          \
        }
        if (kvcache && pipe == 1 && cache_terms == 8) {                            // K / V from the forward's mode-4 stage cache
            if (Lk & 63) return hipErrorNotSupported;
            if (drop_p > 0.f) { if (rag) PARQ_BWD2C(true, true, 8) else PARQ_BWD2C(true, false, 8) }
            else { if (rag) PARQ_BWD2C(false, true, 8) else PARQ_BWD2C(false, false, 8) }
        } else if (kvcache && pipe == 1 && (cache_terms == 3 || cache_terms == 1)) {       // K / V from the forward's 16-bit cache
            if (cache_terms == 3) {
                if (drop_p > 0.f) { if (rag) PARQ_BWD2C(true, true, 3) else PARQ_BWD2C(true, false, 3) }
                else { if (rag) PARQ_BWD2C(false, true, 3) else PARQ_BWD2C(false, false, 3) }
            } else {
                if (drop_p > 0.f) { if (rag) PARQ_BWD2C(true, true, 1) else PARQ_BWD2C(true, false, 1) }
                else { if (rag) PARQ_BWD2C(false, true, 1) else PARQ_BWD2C(false, false, 1) }
            }
        } else if (pipe == 1) {
            if (drop_p > 0.f) { if (rag) PARQ_BWD2(true, true, 1) else PARQ_BWD2(true, false, 1) }
            else { if (rag) PARQ_BWD2(false, true, 1) else PARQ_BWD2(false, false, 1) }
        } else if (pipe == 2) {
            if (drop_p > 0.f) { if (rag) PARQ_BWD2(true, true, 2) else PARQ_BWD2(true, false, 2) }
            else { if (rag) PARQ_BWD2(false, true, 2) else PARQ_BWD2(false, false, 2) }
        } else {
            if (drop_p > 0.f) { if (rag) PARQ_BWD2(true, true, 0) else PARQ_BWD2(true, false, 0) }
            else { if (rag) PARQ_BWD2(false, true, 0) else PARQ_BWD2(false, false, 0) }
        }
#undef PARQ_BWD2
#undef PARQ_BWD2C
    } else {
        e = split_bwd_lds_attr();
        if (e != hipSuccess) return e;
        if (drop_p > 0.f) hipLaunchKernelGGL(attn_bwd_split_kernel<true>, g2, dim3(512), kSplitBwdLds, s, a, oscale);
        else hipLaunchKernelGGL(attn_bwd_split_kernel<false>, g2, dim3(512), kSplitBwdLds, s, a, oscale);
    }
    hipLaunchKernelGGL(attn_bwd_dq_reduce_kernel, dim3(ceil_div(Lq * 64, 256), B * H, n_it), dim3(256), 0, s, gq_part, (int)g2.x, Lq,
                       Lq_pad, H, gq, gq_batch, gq_head, gq_row, a.gqp_it, gq_it);
    return hipGetLastError();
}

// scratch floats for the packed Q / dO tile images of launch_attn_bwd_batched
size_t attn_bwd_pack_floats(int B, int H, int Lq, int n_it) {
    return (size_t)n_it * B * H * (((Lq + 31) & ~31) / 32) * kImgHalfs / 2;
}

// scratch floats for the dQ partials of launch_attn_bwd (0 when the atomics path is taken)
size_t attn_bwd_dq_partial_floats(int B, int H, int Lq, int Lk, int dh, bool grouped) {
    if (dh != 64 || Lk < 2048) return 0;
    const int slots = grouped ? ceil_div(ceil_div(Lk, 256), attn_bwd_kb_group(B, H, Lk)) : ceil_div(Lk, 256);
    return (size_t)B * H * slots * ((Lq + 31) & ~31) * 64;
}

// key blocks per workgroup of the batched split-precision kernel (attn_bwd_split2_kernel): 4 while that leaves at least 512 workgroups
int attn_bwd_kb_group(int B, int H, int Lk) {
    const int nkb = ceil_div(Lk, 256);
    static const int want = [] { const char* e = dev_env("PARQ_BWD_KB_GROUP"); return e && atoi(e) > 0 ? atoi(e) : 4; }();      // development A/B: 1 = one slot per key block
    int g = want;
    while (g > 1 && (int64_t)B * H * ceil_div(nkb, g) < 512) g >>= 1;
    return g;
}

hipError_t launch_attn_bwd_rowdot(const float* dO, const float* O, int64_t batch, int64_t row, int B, int H, int Lq, int dh, float* D,
                                  hipStream_t s) {
    hipLaunchKernelGGL(attn_bwd_rowdot_kernel, dim3((unsigned)ceil_div64((int64_t)B * H * Lq, 4)), dim3(256), 0, s, dO, O, batch, row,
                       B * H, H, Lq, dh, D);
    return hipGetLastError();
}

// out[0] = max(out[0], bit pattern of max |x|) (out zeroed by the caller)
hipError_t launch_absmax(const float* x, int64_t n, unsigned int* out, hipStream_t s) {
    hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, s, x, n, out);
    return hipGetLastError();
}

}  // namespace parq
