// probe (GPU box): lane semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands and of the fp8 converts, as
// flash_split8 relies on them.
//   A operand: lane l holds row (l & 31), k = 32 (l >> 5) + j in byte j of its 8 registers;  B: column (l & 31), same k;
//   C / D: the 32x32 accumulator map of every other 32x32 MFMA (col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5));
//   scale operands: E8M0 byte 0 of a per-lane register, applied to that lane's 32 k-values.
// hipcc --offload-arch=gfx950 -O2 -o mx_probe mx_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void mx(const uint8_t* A, const uint8_t* B, float* D, int sa0, int sa1, int sb0, int sb1) {
    const int l = threadIdx.x, row = l & 31, h = l >> 5;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        uint32_t x = 0, y = 0;
        for (int j = 0; j < 4; ++j) {
            const int k = 32 * h + 4 * w + j;
            x |= (uint32_t)A[row * 64 + k] << (8 * j);        // A[i][k]
            y |= (uint32_t)B[k * 32 + row] << (8 * j);        // B[k][j]
        }
        a[w] = (int)x; b[w] = (int)y;
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    const int sa = h ? sa1 : sa0, sb = h ? sb1 : sb0;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + row] = c[r];
}

__global__ void mx1(const uint8_t* A, const uint8_t* B, float* D, int lane_a, int lane_b, int val) {
    const int l = threadIdx.x, row = l & 31, h = l >> 5;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        uint32_t x = 0, y = 0;
        for (int j = 0; j < 4; ++j) {
            const int k = 32 * h + 4 * w + j;
            x |= (uint32_t)A[row * 64 + k] << (8 * j);
            y |= (uint32_t)B[k * 32 + row] << (8 * j);
        }
        a[w] = (int)x; b[w] = (int)y;
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    const int sa = l == lane_a ? val : 127, sb = l == lane_b ? val : 127;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + row] = c[r];
}

__global__ void cvt(const float* in, float* out, int n, float sc) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float a = in[i];
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, -a, 0, false);
    out[i] = __builtin_amdgcn_cvt_f32_fp8(w, 0);
    out[n + i] = __builtin_amdgcn_cvt_f32_fp8(w, 1);
    s16x2 o = {0, 0};
    o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, a, a * 0.5f, sc, false);
    const int w2 = (int)(unsigned short)o[0];
    out[2 * n + i] = __builtin_amdgcn_cvt_f32_fp8(w2, 0);
    out[3 * n + i] = __builtin_amdgcn_cvt_f32_fp8(w2, 1);
}

static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
    if (e == 15 && m == 7) x = NAN;
    return s ? -x : x;
}

int main() {
    uint8_t hA[32 * 64], hB[64 * 32];
    srand(5);
    for (int i = 0; i < 32 * 64; ++i) { do hA[i] = rand() & 255; while ((hA[i] & 0x7f) > 0x5f); }     // |x| <= 30, no NaN
    for (int i = 0; i < 64 * 32; ++i) { do hB[i] = rand() & 255; while ((hB[i] & 0x7f) > 0x5f); }
    uint8_t *dA, *dB; float* dD;
    CK(hipMalloc(&dA, sizeof hA)); CK(hipMalloc(&dB, sizeof hB)); CK(hipMalloc(&dD, 32 * 32 * 4));
    CK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice));
    const int cases[3][4] = {{127, 127, 127, 127}, {120, 127, 127, 127}, {124, 130, 126, 125}};
    for (const auto& cs : cases) {
        hipLaunchKernelGGL(mx, dim3(1), dim3(64), 0, 0, dA, dB, dD, cs[0], cs[1], cs[2], cs[3]);
        float hD[32 * 32];
        CK(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
        double worst = 0, big = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                for (int k = 0; k < 64; ++k) {
                    const int hh = k >> 5;
                    ref += (double)e4m3(hA[i * 64 + k]) * e4m3(hB[k * 32 + j]) * ldexp(1.0, (hh ? cs[1] : cs[0]) - 127) * ldexp(1.0, (hh ? cs[3] : cs[2]) - 127);
                }
                worst = fmax(worst, fabs(ref - hD[i * 32 + j]));
                big = fmax(big, fabs(ref));
            }
        printf("mx 32x32x64 e4m3, scales A(%d,%d) B(%d,%d): max |D - ref| = %.3g (max |ref| %.3g)\n", cs[0], cs[1], cs[2], cs[3], worst, big);
    }
    {
        float base[32 * 32], hD[32 * 32];
        hipLaunchKernelGGL(mx1, dim3(1), dim3(64), 0, 0, dA, dB, dD, -1, -1, 127);
        CK(hipMemcpy(base, dD, sizeof base, hipMemcpyDeviceToHost));
        const int tests[6][3] = {{3, -1, 120}, {35, -1, 120}, {-1, 3, 120}, {-1, 35, 120}, {3, -1, 127 + 256 * 120}, {3, -1, 0}};
        for (const auto& t : tests) {
            hipLaunchKernelGGL(mx1, dim3(1), dim3(64), 0, 0, dA, dB, dD, t[0], t[1], t[2]);
            CK(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
            int nrow = 0, ncol = 0, first_r = -1, first_c = -1;
            bool rows[32] = {}, cols[32] = {};
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (hD[i * 32 + j] != base[i * 32 + j]) { rows[i] = true; cols[j] = true; }
            for (int i = 0; i < 32; ++i) { if (rows[i]) { ++nrow; if (first_r < 0) first_r = i; } if (cols[i]) { ++ncol; if (first_c < 0) first_c = i; } }
            // which k-block lost its weight? compare with the reference where block kb of row / column is scaled
            printf("scale %d in lane A %d / B %d: %d rows (first %d), %d columns (first %d) of D changed", t[2], t[0], t[1], nrow, first_r, ncol, first_c);
            if (first_r >= 0) {
                const int i = first_r, j = first_c;
                double G[4] = {0, 0, 0, 0};                  // my k labels: group = k / 16
                for (int k = 0; k < 64; ++k) G[k >> 4] += (double)e4m3(hA[i * 64 + k]) * e4m3(hB[k * 32 + j]);
                const double f = ldexp(1.0, (t[2] & 255) - 127);
                printf("; D[%d][%d] = %g; 16-k group sums %g %g %g %g;", i, j, hD[i * 32 + j], G[0], G[1], G[2], G[3]);
                for (int m = 1; m < 16; ++m) {
                    double ref = 0;
                    for (int g = 0; g < 4; ++g) ref += G[g] * ((m >> g) & 1 ? f : 1.0);
                    if (fabs(ref - hD[i * 32 + j]) < 0.02 * (fabs(ref) + 1)) printf(" MATCH scaled groups mask %d", m);
                }
            }
            printf("\n");
        }
    }
    const float vals[] = {0.3f, 1.0f, 1.0625f, 1.1875f, 1.3125f, 447.f, 448.f, 460.f, 480.f, 500.f, 1000.f, 1e6f, 1e-3f, 0.001953125f, 0.0009765625f, 0.0029f, 0.0175f, 17.f, 19.f, 208.f};
    const int n = sizeof vals / sizeof vals[0];
    float *din, *dout, hout[4 * 32];
    CK(hipMalloc(&din, sizeof vals)); CK(hipMalloc(&dout, 4 * n * 4));
    CK(hipMemcpy(din, vals, sizeof vals, hipMemcpyHostToDevice));
    for (float sc : {1.f, 4.f, 0.25f, 3.f}) {
        hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, din, dout, n, sc);
        CK(hipMemcpy(hout, dout, 4 * n * 4, hipMemcpyDeviceToHost));
        printf("scale operand %g:\n", sc);
        for (int i = 0; i < n; ++i)
            printf("  x = %-12g cvt_pk_fp8(x) = %-10g cvt_pk_fp8(-x) = %-10g | cvt_scalef32_pk_fp8(x, scale) = %-10g (x/2: %g)\n", vals[i], hout[i], hout[n + i], hout[2 * n + i], hout[3 * n + i]);
    }
    return 0;
}
