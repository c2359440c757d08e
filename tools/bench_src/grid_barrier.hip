// micro-benchmark: cost of a grid-wide barrier + dependent cross-workgroup read on gfx950 (8 XCDs), versus
// the same dependent phases issued as separate kernel launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// one phase: block b reads the 8 KB written by 8 other blocks in the previous phase, writes its own 8 KB
__device__ __forceinline__ void phase_body(const float* in, float* out, int b, int nb, int phase) {
    float acc = 0.f;
    for (int j = 0; j < 8; ++j) {
        const int src = (b * 8 + j * 37 + phase) % nb;
        const float4 v = reinterpret_cast<const float4*>(in + (size_t)src * 2048)[threadIdx.x];
        acc += v.x + v.y + v.z + v.w;
        const float4 u = reinterpret_cast<const float4*>(in + (size_t)src * 2048)[threadIdx.x + 256];
        acc += u.x + u.y + u.z + u.w;
    }
    float4 o = {acc * 1e-3f, acc * 2e-3f, acc * 3e-3f, 1.f};
    reinterpret_cast<float4*>(out + (size_t)b * 2048)[threadIdx.x] = o;
    reinterpret_cast<float4*>(out + (size_t)b * 2048)[threadIdx.x + 256] = o;
}

__global__ void persistent(float* bufA, float* bufB, unsigned* counter, int phases) {
    const int nb = gridDim.x;
    for (int p = 0; p < phases; ++p) {
        const float* in = (p & 1) ? bufB : bufA;
        float* out = (p & 1) ? bufA : bufB;
        phase_body(in, out, blockIdx.x, nb, p);
        grid_barrier(counter, (unsigned)(nb * (p + 1)));
    }
}
__global__ void single(const float* in, float* out, int phase) { phase_body(in, out, blockIdx.x, gridDim.x, phase); }
__global__ void barrier_only(unsigned* counter, int phases) {
    for (int p = 0; p < phases; ++p) grid_barrier(counter, (unsigned)(gridDim.x * (p + 1)));
}

// single-XCD variant: launch 8x the workgroups, only blockIdx % 8 == 0 (all dispatched to XCD 0) participate
__global__ void persistent_xcd(float* bufA, float* bufB, unsigned* counter, int phases) {
    if (blockIdx.x & 7) return;
    const int nb = gridDim.x >> 3, b = blockIdx.x >> 3;
    for (int p = 0; p < phases; ++p) {
        const float* in = (p & 1) ? bufB : bufA;
        float* out = (p & 1) ? bufA : bufB;
        phase_body(in, out, b, nb, p);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // same L2: no writeback needed
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(nb * (p + 1))) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}
__global__ void persistent_xcd_agent(float* bufA, float* bufB, unsigned* counter, int phases) {
    if (blockIdx.x & 7) return;
    const int nb = gridDim.x >> 3, b = blockIdx.x >> 3;
    for (int p = 0; p < phases; ++p) {
        const float* in = (p & 1) ? bufB : bufA;
        float* out = (p & 1) ? bufA : bufB;
        phase_body(in, out, b, nb, p);
        grid_barrier(counter, (unsigned)(nb * (p + 1)));
    }
}

int main() {
    const int phases = 200;
    float *A, *B; unsigned* c;
    hipMalloc(&A, 256 * 2048 * 4); hipMalloc(&B, 256 * 2048 * 4); hipMalloc(&c, 4);
    hipMemset(A, 0, 256 * 2048 * 4); hipMemset(B, 0, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nb : {16, 32, 64, 128, 256}) {
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(c, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(barrier_only, dim3(nb), dim3(256), 0, 0, c, phases);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("G=%3d barrier only        : %.2f us/phase\n", nb, ms * 1e3 / phases);
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(c, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(persistent, dim3(nb), dim3(256), 0, 0, A, B, c, phases);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("G=%3d persistent + barrier: %.2f us/phase\n", nb, ms * 1e3 / phases);
        if (nb <= 32) {
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(c, 0, 4);
                hipEventRecord(e0);
                hipLaunchKernelGGL(persistent_xcd_agent, dim3(nb * 8), dim3(256), 0, 0, A, B, c, phases);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("G=%3d one-XCD, agent-scope barrier: %.2f us/phase\n", nb, ms * 1e3 / phases);
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(c, 0, 4);
                hipEventRecord(e0);
                hipLaunchKernelGGL(persistent_xcd, dim3(nb * 8), dim3(256), 0, 0, A, B, c, phases);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("G=%3d one-XCD, relaxed barrier    : %.2f us/phase (result validity not checked)\n", nb, ms * 1e3 / phases);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int p = 0; p < phases; ++p)
                hipLaunchKernelGGL(single, dim3(nb), dim3(256), 0, 0, (p & 1) ? B : A, (p & 1) ? A : B, p);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("G=%3d separate launches   : %.2f us/phase\n", nb, ms * 1e3 / phases);
    }
    return 0;
}
