// What a pure streaming read reaches on this box (the ceiling the kNN sweep's 6.19 TB/s should be judged against):
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_read_probe.hip -o tools/ab/hbm_read_probe && tools/ab/hbm_read_probe [GB]
// Variants: loads per thread in flight (unroll), workgroups per CU, plain vs non-temporal loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ p, size_t n4, float* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += p[i];
    const float s = (acc.x + acc.y) + (acc.z + acc.w);
    if (s == 123.456f) out[0] = s;           // (never true: keeps the loads alive)
}

// the same stream through the LDS-DMA path: every lane moves 16 B straight into LDS (1 KB contiguous per wave instruction),
// U instructions in flight per wave, nothing reads the data back.  AUX = cache-policy bits of the instruction (0 plain, 2 = nt? --
// the probe tries 0, 1, 2, 3 and prints what each reaches).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
template <int U, int AUX>
__global__ __launch_bounds__(256) void dma_kernel(const f32x4* __restrict__ p, size_t n4, float* out) {
    __shared__ uint4 lds[U * 256];
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(p + i + u * stride), (lds_void_t*)&lds[(u * 4 + wave) * 64], 16, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lds[threadIdx.x].x == 0x12345678u && n4 == 1) out[0] = 1.f;
}
template <int U, int AUX> void run_dma(const f32x4* d, size_t n4, float* out, int wg_per_cu, int cus) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = cus * wg_per_cu;
    hipLaunchKernelGGL((dma_kernel<U, AUX>), dim3(grid), dim3(256), 0, 0, d, n4, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((dma_kernel<U, AUX>), dim3(grid), dim3(256), 0, 0, d, n4, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("LDS-DMA unroll %d  wg/CU %2d  aux %d: %.3f ms  %.0f GB/s\n", U, wg_per_cu, AUX, best, n4 * 16.0 / best / 1e6);
}

template <int U, bool NT> void run(const f32x4* d, size_t n4, float* out, int wg_per_cu, int cus) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = cus * wg_per_cu;
    hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, d, n4, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, d, n4, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("unroll %d  wg/CU %2d  %s loads: %.3f ms  %.0f GB/s\n", U, wg_per_cu, NT ? "nontemporal" : "plain      ", best, n4 * 16.0 / best / 1e6);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 30.72;
    const size_t n4 = (size_t)(gb * 1e9 / 16);
    f32x4* d; float* out;
    CK(hipMalloc(&d, n4 * 16)); CK(hipMalloc(&out, 4));
    CK(hipMemset(d, 0, n4 * 16));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, %.2f GB\n", prop.name, cus, n4 * 16.0 / 1e9);
    for (int w : {4, 8}) { run_dma<8, 0>(d, n4, out, w, cus); run_dma<8, 1>(d, n4, out, w, cus); run_dma<8, 2>(d, n4, out, w, cus); run_dma<8, 3>(d, n4, out, w, cus); run_dma<4, 2>(d, n4, out, w, cus); }
    for (int w : {8}) { run<4, false>(d, n4, out, w, cus); run<8, false>(d, n4, out, w, cus); run<8, true>(d, n4, out, w, cus); run<16, false>(d, n4, out, w, cus); }
    return 0;
}
