// Scratch probe: effective shader clock under load = d(s_memtime) / d(s_memrealtime, 100 MHz).
// mode 0: bf16 MFMA back to back (4 waves/CU x 3 blocks)   mode 1: fp32-input MFMA   mode 2: VALU fma only
// mode 3: bf16 MFMA with zero operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int mode>
__global__ __launch_bounds__(256) void load_kernel(int iters, float seed, uint64_t* out, float* sink) {
    uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float v = seed * (1 + (threadIdx.x & 7)) * 1.3717f;
    uint4 u = {__float_as_uint(v), __float_as_uint(v * 1.1f), __float_as_uint(v * 0.7f), __float_as_uint(v * 1.9f)};
    if (mode == 3) u = make_uint4(0, 0, 0, 0);
    bf16x8_t a = __builtin_bit_cast(bf16x8_t, u), b = a;
    float x = v, y = v * 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (mode == 0 || mode == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        } else if (mode == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, y, acc[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) { x = fmaf(x, 1.0001f, y); y = fmaf(y, 0.9999f, x); }
        }
    }
    uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = x + y;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    uint64_t* d; float* sink; hipMalloc(&d, 768 * 16); hipMalloc(&sink, 4);
    uint64_t h[2 * 768];
    const char* names[] = {"bf16 MFMA 32x32x16 dense", "fp32-input MFMA 32x32x2", "VALU fma", "bf16 MFMA zero operands"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
        int iters = mode == 1 ? 40000 : 200000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(load_kernel<0>, dim3(768), dim3(256), 0, 0, iters, 0.37f, d, sink);
        if (mode == 1) hipLaunchKernelGGL(load_kernel<1>, dim3(768), dim3(256), 0, 0, iters, 0.37f, d, sink);
        if (mode == 2) hipLaunchKernelGGL(load_kernel<2>, dim3(768), dim3(256), 0, 0, iters, 0.37f, d, sink);
        if (mode == 3) hipLaunchKernelGGL(load_kernel<3>, dim3(768), dim3(256), 0, 0, iters, 0.37f, d, sink);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double cs = 0, rs = 0; for (int i = 0; i < 768; ++i) { cs += h[2 * i]; rs += h[2 * i + 1]; }
        double ghz = cs / rs * 0.1;     // realtime counter = 100 MHz
        double flops = mode == 0 || mode == 3 ? 768.0 * 4 * iters * 4 * 32768 : mode == 1 ? 768.0 * 4 * iters * 4 * 4096 : 0;
        printf("%-28s %.2f ms  shader clock %.3f GHz  %s%.0f TFLOP/s\n", names[mode], ms, ghz, flops ? "" : "(n/a) ", flops / ms / 1e9);
    }
    return 0;
}
