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

namespace parq {

namespace {

struct AttnBwdArgs {
    const float* q; int64_t q_batch, q_head, q_row;
    const float* k; int64_t k_batch, k_head, k_row;
    const float* v; int64_t v_batch, v_head, v_row;
    const float* dO; int64_t do_batch, do_head, do_row;
    const float* lse;          // [B*H][Lq_pad], log2 domain
    const float* D;            // [B*H][Lq_pad]
    float* gq; int64_t gq_batch, gq_head, gq_row;     // += (atomic)
    float* gk; int64_t gk_batch, gk_head, gk_row;     // = or += (accumulate)
    float* gv; int64_t gv_batch, gv_head, gv_row;
    int B, H, Lq, Lk, accumulate_kv;
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
            const float ds = p * (dp - st[32 + i]);
            dS[i * 257 + tid] = ds;
#pragma unroll
            for (int d = 0; d < DH; ++d) {
                gkj[d] += ds * Qt[i * DH + d];
                gvj[d] += p * Ot[i * DH + d];
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

// q/k/v/dO/g*: element strides (batch, head, row); lse, D: [B*H][pad32(Lq)].  gq must be zeroed by the caller.
hipError_t launch_attn_bwd(const float* q, int64_t q_batch, int64_t q_head, int64_t q_row, const float* k, int64_t k_batch,
                           int64_t k_head, int64_t k_row, const float* v, int64_t v_batch, int64_t v_head, int64_t v_row,
                           const float* dO, int64_t do_batch, int64_t do_head, int64_t do_row, const float* lse, const float* D,
                           float* gq, int64_t gq_batch, int64_t gq_head, int64_t gq_row, float* gk, int64_t gk_batch,
                           int64_t gk_head, int64_t gk_row, float* gv, int64_t gv_batch, int64_t gv_head, int64_t gv_row, int B, int H,
                           int Lq, int Lk, int dh, int accumulate_kv, hipStream_t s) {
    if (dh != 64 && dh != 32) return hipErrorInvalidValue;
    AttnBwdArgs a;
    a.q = q; a.q_batch = q_batch; a.q_head = q_head; a.q_row = q_row;
    a.k = k; a.k_batch = k_batch; a.k_head = k_head; a.k_row = k_row;
    a.v = v; a.v_batch = v_batch; a.v_head = v_head; a.v_row = v_row;
    a.dO = dO; a.do_batch = do_batch; a.do_head = do_head; a.do_row = do_row;
    a.lse = lse; a.D = D;
    a.gq = gq; a.gq_batch = gq_batch; a.gq_head = gq_head; a.gq_row = gq_row;
    a.gk = gk; a.gk_batch = gk_batch; a.gk_head = gk_head; a.gk_row = gk_row;
    a.gv = gv; a.gv_batch = gv_batch; a.gv_head = gv_head; a.gv_row = gv_row;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.accumulate_kv = accumulate_kv;
    dim3 grid(ceil_div(Lk, 256), B * H);
    if (dh == 64) {
        const size_t lds = (size_t)(2 * 32 * 64 + 32 * 257 + 256 * 65 + 64) * sizeof(float);
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<64>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr = true;
        }
        hipLaunchKernelGGL(attn_bwd_kernel<64>, grid, dim3(256), lds, s, a);
    } else {
        const size_t lds = (size_t)(2 * 32 * 32 + 32 * 257 + 256 * 33 + 64) * sizeof(float);
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<32>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr = true;
        }
        hipLaunchKernelGGL(attn_bwd_kernel<32>, grid, dim3(256), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_attn_bwd_rowdot(const float* dO, const float* O, int64_t batch, int64_t row, int B, int H, int Lq, int dh, float* D,
                                  hipStream_t s) {
    hipLaunchKernelGGL(attn_bwd_rowdot_kernel, dim3((unsigned)ceil_div64((int64_t)B * H * Lq, 4)), dim3(256), 0, s, dO, O, batch, row,
                       B * H, H, Lq, dh, D);
    return hipGetLastError();
}

}  // namespace parq
