#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int h_addr[64]; short h_out[256];
    int *d_addr; short* d_out;
    hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
    for (int variant = 0; variant < 3; ++variant) {
        for (int l = 0; l < 64; ++l) {
            int t = l & 15, g = l >> 4;
            if (variant == 0) h_addr[l] = 0;                                   // uniform address
            if (variant == 1) h_addr[l] = (t >> 2) * 16 + (t & 3) * 4 + g * 64; // contiguous 4x16 block per group
            if (variant == 2) h_addr[l] = (t >> 2) * 100 + (t & 3) * 4 + g * 1000; // row stride 100, group stride 1000
        }
        hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
        printf("variant %d\n", variant);
        for (int l = 0; l < 64; ++l) { printf("l%2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h_out[l * 4 + j]); printf(l % 4 == 3 ? "\n" : "   "); }
    }
    return 0;
}
