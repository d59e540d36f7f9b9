// Scratch probe: which SIMD does each wave of a 512-thread (and 256-thread) workgroup land on?  (HW_REG_HW_ID bits 5:4)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = v;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 4);
    for (int threads : {512, 256, 1024}) {
        const int nb = 6, w = threads / 64;
        hipLaunchKernelGGL(k, dim3(nb), dim3(threads), 0, 0, d);
        unsigned h[4096]; hipMemcpy(h, d, nb * w * 4, hipMemcpyDeviceToHost);
        printf("threads=%d\n", threads);
        for (int b = 0; b < nb; ++b) {
            printf("  block %d: simd of waves:", b);
            for (int i = 0; i < w; ++i) printf(" %u", (h[b * w + i] >> 4) & 3);
            printf("   (cu %u, wave slots:", (h[b * w] >> 8) & 15);
            for (int i = 0; i < w; ++i) printf(" %u", h[b * w + i] & 15);
            printf(")\n");
        }
    }
    return 0;
}
