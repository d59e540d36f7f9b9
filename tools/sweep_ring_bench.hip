// Experiment for the next round (not product code): the fp32 kNN distance sweep with the store rows staged through a
// wave-private LDS ring by whole-line LDS-DMA (optionally non-temporal) and the 16 queries held in REGISTERS as MFMA B
// fragments -- the form DESIGN 2.1 prices at ~6.4 TB/s against the shipped kernel's 6.13 (plain-load ceiling 6.1).
//   hipcc -O3 --offload-arch=gfx950 tools/sweep_ring_bench.hip -o tools/ab/sweep_ring_bench && tools/ab/sweep_ring_bench [rows]
// Per row r and query j it forms d = |p_r|^2 - 2 p_r.q_j on v_mfma_f32_16x16x4_f32 and keeps the minimum per query (a stand-in
// for the candidate lists of the real kernel); the minima are checked against a host loop on the first 4096 rows.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int D = 768, NQ = 32, NCHUNK = D / 32;            // a chunk = 16 rows x 32 floats (one 128-byte line per row)
constexpr int WAVES = 8;

__global__ void fill(float* x, size_t n, unsigned long long seed) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    x[i] = ((float)(z >> 40) * (1.0f / 16777216.0f) - 0.5f) * 0.0722f;      // ~unit-norm rows at D = 768
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// RING = chunks of LDS per wave (2 KB each); AUX = cache policy of the DMA (0 plain, 2 non-temporal)
template <int RING, int AUX>
__global__ __launch_bounds__(64 * WAVES, 2) void sweep_ring(const float* __restrict__ P, long nrows, const float* __restrict__ Q,
                                                            float* __restrict__ out_min /* [total waves][16] */) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];      // [WAVES][RING][2 halves][64 lanes] x 16 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * WAVES + wave, nw = (long)gridDim.x * WAVES;
    uint4* ring = lds + (size_t)wave * RING * 128;
    // queries as B fragments: lane (col j = lane & 15, ksub = lane >> 4) holds -2 q_j[16 kb + 4 ksub .. +3] for every k-block
    f32x4 bq[2 * NCHUNK];
    {
        const int j = lane & 15, ksub = lane >> 4;
#pragma unroll
        for (int kb = 0; kb < 2 * NCHUNK; ++kb) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Q + (size_t)j * D + 16 * kb + 4 * ksub);
            bq[kb] = -2.f * v;
        }
    }
    const long ngroups = nrows / 16;                                    // (rows % 16 == 0 in this harness)
    const long my = ngroups > gw ? (ngroups - 1 - gw) / nw + 1 : 0;      // groups gw, gw + nw, ...
    const long total = my * NCHUNK;                                     // chunks this wave streams
    // DMA lane mapping: instruction h of a chunk covers rows 8 h .. 8 h + 7; lane (a = lane / 8, b = lane % 8) copies the
    // 16-byte piece (b ^ a) of row 8 h + a into LDS slot `lane` -- 8 consecutive lanes = one whole 128-byte line, and the
    // fragment read of (row, piece) finds it at slot 8 a + (piece ^ a): rows of one piece spread over the banks
    const int da = lane >> 3, dp = (lane & 7) ^ da;
    long is_chunk = 0;                                                  // next chunk to issue
    const float* src0 = P + ((size_t)gw * 16 + da) * D + 4 * dp;
    auto issue = [&]() {
        const long c = is_chunk < total ? is_chunk : total - 1;         // past the end: the last chunk again (never consumed)
        const long grp = c / NCHUNK; const int ch = (int)(c - grp * NCHUNK);
        const float* s = src0 + (size_t)grp * nw * 16 * D + 32 * ch;
        uint4* dst = ring + (size_t)(is_chunk % RING) * 128;
        __builtin_amdgcn_global_load_lds((glb_void_t*)s, (lds_void_t*)dst, 16, 0, AUX);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(s + 8 * D), (lds_void_t*)(dst + 64), 16, 0, AUX);
        ++is_chunk;
    };
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    if (total > 0) {
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) issue();
        const int arow = lane & 15, ksub = lane >> 4;
        const int rh = arow >> 3, ra = arow & 7;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float nsq = 0.f;
        long c = 0;
        for (long grp = 0; grp < my; ++grp) {
#pragma unroll
            for (int ch = 0; ch < NCHUNK; ++ch, ++c) {                      // (unrolled: bq[] must be indexed statically)
                wait_vm<(RING - 2) * 2>();                              // chunk c has landed (this wave's own DMA queue)
                __builtin_amdgcn_sched_barrier(0);
                const uint4* slot = ring + (size_t)(c % RING) * 128 + rh * 64 + 8 * ra;
                const f32x4 a0 = __builtin_bit_cast(f32x4, slot[(ksub) ^ ra]);
                const f32x4 a1 = __builtin_bit_cast(f32x4, slot[(4 + ksub) ^ ra]);
                issue();                                                // refills the slot chunk c - 1 used (its reads are consumed)
                const f32x4 b0 = bq[2 * ch], b1 = bq[2 * ch + 1];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc, 0, 0, 0);
                nsq += (a0.x * a0.x + a0.y * a0.y) + (a0.z * a0.z + a0.w * a0.w) + (a1.x * a1.x + a1.y * a1.y) + (a1.z * a1.z + a1.w * a1.w);
            }
            // row group done: fold |p|^2 in (A = partial norms, B = 1)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(nsq, 1.0f, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) best[r] = fminf(best[r], acc[r]);
            acc = f32x4{0.f, 0.f, 0.f, 0.f}; nsq = 0.f;
        }
        wait_vm<0>();
    }
    // lane holds column (query) lane & 15, rows 4 (lane >> 4) + r of every group: minimum over its rows, then over the 4 lane groups
    float m = fminf(fminf(best[0], best[1]), fminf(best[2], best[3]));
    m = fminf(m, __shfl_xor(m, 16)); m = fminf(m, __shfl_xor(m, 32));
    if (lane < 16) out_min[gw * 16 + lane] = m;
}

// ---- 32 resident queries (two sub-tiles of 16): sub-tile 0's fragments in registers for the first 16 chunks, the rest of
// sub-tile 0 and all of sub-tile 1 in LDS (64 KB, written once by wave 0), rows through the same rings ----
template <int RING, int AUX>
__global__ __launch_bounds__(64 * WAVES, 2) void sweep_ring32(const float* __restrict__ P, long nrows, const float* __restrict__ Q,
                                                              float* __restrict__ out_min /* [total waves][32] */) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];      // rings | Qs0 [8 chunks][2][64] | Qs1 [24 chunks][2][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * WAVES + wave, nw = (long)gridDim.x * WAVES;
    uint4* ring = lds + (size_t)wave * RING * 128;
    f32x4* Qs0 = reinterpret_cast<f32x4*>(lds + (size_t)WAVES * RING * 128);
    f32x4* Qs1 = Qs0 + 8 * 2 * 64;
    constexpr int REG = 16;
    f32x4 bq[2 * REG];
    {
        const int j = lane & 15, ksub = lane >> 4;
#pragma unroll
        for (int kb = 0; kb < 2 * NCHUNK; ++kb) {
            const f32x4 v0 = -2.f * *reinterpret_cast<const f32x4*>(Q + (size_t)j * D + 16 * kb + 4 * ksub);
            const f32x4 v1 = -2.f * *reinterpret_cast<const f32x4*>(Q + (size_t)(16 + j) * D + 16 * kb + 4 * ksub);
            if (kb < 2 * REG) bq[kb < 2 * REG ? kb : 0] = v0;
            else if (wave == 0) Qs0[(kb - 2 * REG) * 64 + lane] = v0;
            if (wave == 0) Qs1[kb * 64 + lane] = v1;
        }
    }
    __syncthreads();
    const long ngroups = nrows / 16;
    const long my = ngroups > gw ? (ngroups - 1 - gw) / nw + 1 : 0;
    const long total = my * NCHUNK;
    const int da = lane >> 3, dp = (lane & 7) ^ da;
    long is_chunk = 0;
    const float* src0 = P + ((size_t)gw * 16 + da) * D + 4 * dp;
    auto issue = [&]() {
        const long c = is_chunk < total ? is_chunk : total - 1;
        const long grp = c / NCHUNK; const int ch = (int)(c - grp * NCHUNK);
        const float* s = src0 + (size_t)grp * nw * 16 * D + 32 * ch;
        uint4* dst = ring + (size_t)(is_chunk % RING) * 128;
        __builtin_amdgcn_global_load_lds((glb_void_t*)s, (lds_void_t*)dst, 16, 0, AUX);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(s + 8 * D), (lds_void_t*)(dst + 64), 16, 0, AUX);
        ++is_chunk;
    };
    float best0[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, best1[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    if (total > 0) {
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) issue();
        const int arow = lane & 15, ksub = lane >> 4;
        const int rh = arow >> 3, ra = arow & 7;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float nsq = 0.f;
        long c = 0;
        for (long grp = 0; grp < my; ++grp) {
#pragma unroll
            for (int ch = 0; ch < NCHUNK; ++ch, ++c) {
                wait_vm<(RING - 2) * 2>();
                __builtin_amdgcn_sched_barrier(0);
                const uint4* slot = ring + (size_t)(c % RING) * 128 + rh * 64 + 8 * ra;
                const f32x4 a0 = __builtin_bit_cast(f32x4, slot[(ksub) ^ ra]);
                const f32x4 a1 = __builtin_bit_cast(f32x4, slot[(4 + ksub) ^ ra]);
                issue();
                f32x4 b0, b1;
                if (ch < REG) { b0 = bq[ch < REG ? 2 * ch : 0]; b1 = bq[ch < REG ? 2 * ch + 1 : 0]; }
                else { b0 = Qs0[(2 * (ch - REG)) * 64 + lane]; b1 = Qs0[(2 * (ch - REG) + 1) * 64 + lane]; }
                const f32x4 c0 = Qs1[(2 * ch) * 64 + lane], c1 = Qs1[(2 * ch + 1) * 64 + lane];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, c0.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, c0.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, c0.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, c0.w, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, c1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, c1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, c1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, c1.w, acc1, 0, 0, 0);
                nsq += (a0.x * a0.x + a0.y * a0.y) + (a0.z * a0.z + a0.w * a0.w) + (a1.x * a1.x + a1.y * a1.y) + (a1.z * a1.z + a1.w * a1.w);
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(nsq, 1.0f, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(nsq, 1.0f, acc1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { best0[r] = fminf(best0[r], acc0[r]); best1[r] = fminf(best1[r], acc1[r]); }
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f}; nsq = 0.f;
        }
        wait_vm<0>();
    }
    float m0 = fminf(fminf(best0[0], best0[1]), fminf(best0[2], best0[3])), m1 = fminf(fminf(best1[0], best1[1]), fminf(best1[2], best1[3]));
    m0 = fminf(m0, __shfl_xor(m0, 16)); m0 = fminf(m0, __shfl_xor(m0, 32));
    m1 = fminf(m1, __shfl_xor(m1, 16)); m1 = fminf(m1, __shfl_xor(m1, 32));
    if (lane < 16) { out_min[gw * 32 + lane] = m0; out_min[gw * 32 + 16 + lane] = m1; }
}

template <int RING, int AUX> void run32(const float* P, long rows, const float* Q, float* out, int cus, const std::vector<float>& want32) {
    const int grid = cus;
    const size_t lds = (size_t)WAVES * RING * 128 * 16 + (8 + 24) * 2 * 64 * 16;
    CK(hipFuncSetAttribute((const void*)sweep_ring32<RING, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((sweep_ring32<RING, AUX>), dim3(grid), dim3(64 * WAVES), lds, 0, P, 4096l, Q, out);
    CK(hipDeviceSynchronize());
    std::vector<float> h((size_t)grid * WAVES * 32);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int j = 0; j < 32; ++j) { float m = INFINITY; for (size_t w = 0; w < (size_t)grid * WAVES; ++w) m = fminf(m, h[w * 32 + j]); maxerr = fmax(maxerr, fabs((double)m - want32[j])); }
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((sweep_ring32<RING, AUX>), dim3(grid), dim3(64 * WAVES), lds, 0, P, rows, Q, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("32 queries: ring %2d chunks/wave (%3zu KB LDS/block) aux %d: %.3f ms  %.0f GB/s   max |min-dist error| on 4096 rows %.2e\n", RING, lds >> 10, AUX, best,
           rows * (double)D * 4 / best / 1e6, maxerr);
}

template <int RING, int AUX> void run(const float* P, long rows, const float* Q, float* out, int cus, int blocks_per_cu, const std::vector<float>& want) {
    const int grid = cus * blocks_per_cu;
    const size_t lds = (size_t)WAVES * RING * 128 * 16;
    CK(hipFuncSetAttribute((const void*)sweep_ring<RING, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // correctness on the first 4096 rows
    hipLaunchKernelGGL((sweep_ring<RING, AUX>), dim3(grid), dim3(64 * WAVES), lds, 0, P, 4096l, Q, out);
    CK(hipDeviceSynchronize());
    std::vector<float> h((size_t)grid * WAVES * 16);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int j = 0; j < 16; ++j) { float m = INFINITY; for (size_t w = 0; w < (size_t)grid * WAVES; ++w) m = fminf(m, h[w * 16 + j]); maxerr = fmax(maxerr, fabs((double)m - want[j])); }
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((sweep_ring<RING, AUX>), dim3(grid), dim3(64 * WAVES), lds, 0, P, rows, Q, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("ring %2d chunks/wave (%3zu KB LDS/block) x %d block/CU  aux %d: %.3f ms  %.0f GB/s   max |min-dist error| on 4096 rows %.2e\n", RING, lds >> 10,
           blocks_per_cu, AUX, best, rows * (double)D * 4 / best / 1e6, maxerr);
}

int main(int argc, char** argv) {
    const long rows = argc > 1 ? atol(argv[1]) : 10000000l;
    float *P, *Q, *out;
    CK(hipMalloc(&P, (size_t)rows * D * 4)); CK(hipMalloc(&Q, NQ * D * 4)); CK(hipMalloc(&out, 2048 * WAVES * 32 * 4));
    hipLaunchKernelGGL(fill, dim3((unsigned)(((size_t)rows * D + 255) / 256)), dim3(256), 0, 0, P, (size_t)rows * D, 1ull);
    hipLaunchKernelGGL(fill, dim3((NQ * D + 255) / 256), dim3(256), 0, 0, Q, (size_t)NQ * D, 2ull);
    CK(hipDeviceSynchronize());
    std::vector<float> hp((size_t)4096 * D), hq(NQ * D);
    CK(hipMemcpy(hp.data(), P, hp.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hq.data(), Q, hq.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> want(16, INFINITY), want32(32, INFINITY);
    for (int r = 0; r < 4096; ++r) {
        double n2 = 0; for (int k = 0; k < D; ++k) n2 += (double)hp[(size_t)r * D + k] * hp[(size_t)r * D + k];
        for (int j = 0; j < 32; ++j) { double dot = 0; for (int k = 0; k < D; ++k) dot += (double)hp[(size_t)r * D + k] * hq[j * D + k]; want32[j] = fminf(want32[j], (float)(n2 - 2 * dot)); if (j < 16) want[j] = want32[j]; }
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%d CUs, %ld rows x %d (%.2f GB), 16 queries in registers\n", cus, rows, D, rows * (double)D * 4 / 1e9);
    run<8, 0>(P, rows, Q, out, cus, 1, want);
    run<8, 2>(P, rows, Q, out, cus, 1, want);
    run<4, 2>(P, rows, Q, out, cus, 1, want);
    run<6, 2>(P, rows, Q, out, cus, 1, want);
    run<7, 2>(P, rows, Q, out, cus, 1, want);
    run32<4, 2>(P, rows, Q, out, cus, want32);
    run32<3, 2>(P, rows, Q, out, cus, want32);
    return 0;
}
