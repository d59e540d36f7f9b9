// What can ONE CU pull out of L2 per clock, and through which path?  (round 4: the fp16x2 / bf16x3 GEMM loops sit at ~10 - 13 B per
// clock and CU of operand fetch with the texture-data path "stalled on cache" 30 - 38 % of the cycles -- DESIGN 2.3f.)
//   hipcc -O3 --offload-arch=gfx950 tools/l2_fetch_probe.hip -o tools/ab/l2_fetch_probe && tools/ab/l2_fetch_probe
// Every workgroup (one per CU) reads a 1 MB region shared by the workgroups of its XCD over and over (L2 hits after the first
// pass), rotated by the workgroup's index so that CUs do not ask for the same line at the same moment.  Variants:
//   dma  : global_load_lds b128 (1 KB per wave instruction straight into LDS), U instructions per wave between two waits
//   vgpr : global_load_dwordx4 into registers, U in flight per wave
// for 4 / 8 / 16 waves per workgroup.  Prints GB/s of the chip and bytes per clock and CU (shader clock measured in-kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr size_t kRegion = 1 << 20;          // bytes per XCD region

template <int U, int WAVES, bool DMA>
__global__ __launch_bounds__(64 * WAVES) void fetch_kernel(const char* __restrict__ base, int iters, unsigned long long* cycles, float* out) {
    __shared__ uint4 lds[DMA ? U * WAVES * 64 : 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* region = base + (size_t)(blockIdx.x & 7) * kRegion;
    const size_t per_iter = (size_t)U * WAVES * 1024;                      // bytes this workgroup moves per iteration
    size_t off = ((size_t)(blockIdx.x >> 3) * 37 * 1024) % kRegion;        // rotate the start per workgroup
    const unsigned long long t0 = clock64();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if constexpr (DMA) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t o = (off + (size_t)(u * WAVES + wave) * 1024) & (kRegion - 1);
                __builtin_amdgcn_global_load_lds((glb_void_t*)(region + o + lane * 16), (lds_void_t*)&lds[(u * WAVES + wave) * 64], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t o = (off + (size_t)(u * WAVES + wave) * 1024) & (kRegion - 1);
                v[u] = *reinterpret_cast<const f32x4*>(region + o + lane * 16);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
        off = (off + per_iter) & (kRegion - 1);
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    const float s = (acc.x + acc.y) + (acc.z + acc.w) + (DMA ? (float)lds[threadIdx.x & 63].x : 0.f);
    if (s == 123.456f) out[0] = s;
}

template <int U, int WAVES, bool DMA> void run(const char* d, unsigned long long* d_cyc, float* out, int cus) {
    const int iters = (int)((size_t)(8 << 20) / ((size_t)U * WAVES * 1024));       // 8 MB per workgroup
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((fetch_kernel<U, WAVES, DMA>), dim3(cus), dim3(64 * WAVES), 0, 0, d, iters, d_cyc, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fetch_kernel<U, WAVES, DMA>), dim3(cus), dim3(64 * WAVES), 0, 0, d, iters, d_cyc, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    unsigned long long h[1024];
    CK(hipMemcpy(h, d_cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < cus; ++i) mean += (double)h[i];
    mean /= cus;
    const double bytes_wg = (double)iters * U * WAVES * 1024;
    printf("%-4s waves %2d  in flight per wave %d (%3d KB per CU): %7.0f GB/s chip  %6.1f B/clk/CU  (%.0f cycles per wait)\n", DMA ? "dma" : "vgpr", WAVES,
           U, U * WAVES, bytes_wg * cus / best / 1e6, bytes_wg / mean, mean / iters);
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    char* d; CK(hipMalloc(&d, 8 * kRegion)); CK(hipMemset(d, 1, 8 * kRegion));
    unsigned long long* d_cyc; CK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 1024));
    float* out; CK(hipMalloc(&out, 4));
    printf("%d CUs, 1 MB region per XCD (L2-resident), one workgroup per CU\n", cus);
#define ROW(W, D) run<1, W, D>(d, d_cyc, out, cus); run<2, W, D>(d, d_cyc, out, cus); run<4, W, D>(d, d_cyc, out, cus); run<8, W, D>(d, d_cyc, out, cus);
    ROW(4, true) ROW(8, true) ROW(16, true)
    ROW(4, false) ROW(8, false) ROW(16, false)
    return 0;
}
