// micro-benchmark: what a plain streaming kernel reaches on MI355X with the traffic shape of the K/V projection
// (read 197 MB of fp32 tokens, write 393 MB of 16-bit cache), to price kvproj against a measured ceiling rather than
// the 8 TB/s datasheet figure.  Buffers rotate through 4 sets (2.4 GB) so the 256 MB Infinity Cache cannot hold them.
//   mode 0: read only            mode 1: write only (contiguous 1 KB per wave instruction)
//   mode 2: read + write         mode 3: read + write, the writes as the projection issues them (16-byte chunks, 128-byte row stride)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void stream(const f4* __restrict__ in, f4* __restrict__ out, size_t n_in4, float* sink) {
    // each thread: 2 x 16 B in, 4 x 16 B out per step (the projection's 1 : 2 byte ratio)
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t steps = n_in4 / (2 * nthr);
    for (size_t s = 0; s < steps; ++s) {
        const size_t base = s * nthr;
        f4 a = {1.f, 2.f, 3.f, 4.f}, b = {5.f, 6.f, 7.f, 8.f};
        if (MODE != 1) {
            a = in[2 * base + tid];
            b = in[2 * base + nthr + tid];
        }
        if (MODE == 0) { acc += a; acc += b; }
        if (MODE == 1 || MODE == 2) {
            out[4 * base + tid] = a;
            out[4 * base + nthr + tid] = b;
            out[4 * base + 2 * nthr + tid] = a + b;
            out[4 * base + 3 * nthr + tid] = a - b;
        }
        if (MODE == 3) {
            // wave instruction = 64 lanes x 16 B: lane (li = lane & 31, kh = lane >> 5) -> row li (128 B stride), chunk position 4 kh + j
            const int lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
            const size_t wave_id = (base + tid) >> 6;                       // 4 KB block per wave and step
            f4* blk = out + wave_id * 256;
            blk[li * 8 + 4 * kh + 0] = a;
            blk[li * 8 + 4 * kh + 1] = b;
            blk[li * 8 + 4 * kh + 2] = a + b;
            blk[li * 8 + 4 * kh + 3] = a - b;
        }
    }
    if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) *sink = acc[0];
}

int main() {
    const size_t in_bytes = 196608000, out_bytes = 2 * in_bytes;          // cfg 3: 192 000 tokens x 256 x 4 B
    const int sets = 4;
    std::vector<f4*> in(sets), out(sets);
    for (int i = 0; i < sets; ++i) {
        hipMalloc(&in[i], in_bytes);
        hipMalloc(&out[i], out_bytes);
        hipMemset(in[i], 0, in_bytes);
        hipMemset(out[i], 0, out_bytes);
    }
    float* sink;
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t n_in4 = in_bytes / 16;
    for (int grid : {256, 512, 1024, 2048}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f, sum = 0.f;
            const int reps = 12;
            for (int r = 0; r < reps + 2; ++r) {
                const int i = r % sets;
                hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(stream<0>, dim3(grid), dim3(512), 0, 0, in[i], out[i], n_in4, sink); break;
                    case 1: hipLaunchKernelGGL(stream<1>, dim3(grid), dim3(512), 0, 0, in[i], out[i], n_in4, sink); break;
                    case 2: hipLaunchKernelGGL(stream<2>, dim3(grid), dim3(512), 0, 0, in[i], out[i], n_in4, sink); break;
                    default: hipLaunchKernelGGL(stream<3>, dim3(grid), dim3(512), 0, 0, in[i], out[i], n_in4, sink); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (r >= 2) { sum += ms; best = ms < best ? ms : best; }
            }
            const double bytes = (mode == 0 ? in_bytes : mode == 1 ? out_bytes : in_bytes + out_bytes);
            printf("grid %4d mode %d: avg %.1f us best %.1f us  -> %.2f TB/s (avg)\n", grid, mode, sum / reps * 1e3, best * 1e3,
                   bytes / (sum / reps * 1e-3) / 1e12);
        }
    }
    return 0;
}
