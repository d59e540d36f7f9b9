// PrototypeMemory bookkeeping on the device: the per-example loop of memory.py:41-83 /
// :196-217 -- append the example; if the class now holds more than max_examples_per_class, keep the ones
// closest (L2) to the class mean, i.e. drop the farthest -- is sequential per class (every prune changes the
// mean), which makes add_examples() O(n * D) of host work PER EXAMPLE in the reference.  Here one workgroup
// per class runs that sequential loop for all the examples a call adds to the class: fp64 running sum, mean
// rounded to fp32 like the reference's `stack(...).mean(0)`, one wave per row for the distances (coalesced
// row reads), block-wide arg-max, and reports which rows were dropped plus the survivors' distances to the
// mean at the last prune (the reference leaves the list sorted by them).  The host applies the result to its
// (authoritative) Python lists once per call.
#include "common.h"

namespace {

constexpr int kPruneThreads = 1024;
constexpr int kPruneWaves = kPruneThreads / 64;
constexpr int kPruneMaxRows = 8192;
constexpr int kPruneMaxD = 4096;

typedef ac_prune_job PruneArgs;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kPruneThreads) void class_add_prune_kernel(const ac_prune_job* __restrict__ jobs, int D_) {
    const PruneArgs a = jobs[blockIdx.x];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* dist_s = reinterpret_cast<double*>(smem);                          // [n]
    double* sum_s = dist_s + ((a.n_old + a.n_new + 1) & ~1);                  // [D] (even offset: mean_s stays 16-B aligned)
    float* mean_s = reinterpret_cast<float*>(sum_s + D_);                     // [D], 16-byte aligned (n even-padded below)
    uint8_t* alive_s = reinterpret_cast<uint8_t*>(mean_s + D_);               // [n]
    __shared__ double red_d[kPruneWaves];
    __shared__ int red_i[kPruneWaves];
    __shared__ int drop_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n_old + a.n_new, D = D_;
    for (int i = tid; i < n; i += kPruneThreads) { alive_s[i] = i < a.n_old; dist_s[i] = 0.0; }
    for (int d = tid; d < D; d += kPruneThreads) sum_s[d] = a.sum[d];
    __syncthreads();
    int count = a.n_old;
    for (int t = 0; t < a.n_new; ++t) {
        const int r = a.n_old + t;
        for (int d = tid; d < D; d += kPruneThreads) sum_s[d] += (double)a.rows[(int64_t)r * a.ld + d];
        if (tid == 0) alive_s[r] = 1;
        ++count;
        __syncthreads();
        if (count <= a.cap) { if (tid == 0) a.dropped[t] = -1; continue; }
        for (int d = tid; d < D; d += kPruneThreads) mean_s[d] = (float)(sum_s[d] / (double)count);
        __syncthreads();
        // distances: one wave per row, lanes across the embedding (coalesced), fp32 differences like
        // torch.norm(emb - mean), squares accumulated in fp64
        // (rows are read as float4 when D and ld allow it, two rows in flight per wave)
        const bool vec = (D & 3) == 0 && (a.ld & 3) == 0 && ((((uintptr_t)a.rows) & 15) == 0);
        for (int i0 = wave; i0 < n; i0 += 2 * kPruneWaves) {
            const int i1 = i0 + kPruneWaves;
            const bool ok0 = alive_s[i0], ok1 = i1 < n && alive_s[i1];            // wave-uniform
            if (!ok0 && !ok1) continue;
            const float* r0 = a.rows + (int64_t)i0 * a.ld;
            const float* r1 = a.rows + (int64_t)(ok1 ? i1 : i0) * a.ld;
            double acc0 = 0.0, acc1 = 0.0;
            if (vec) {
                for (int d = 4 * lane; d < D; d += 256) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(mean_s + d);
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(r0 + d), x1 = *reinterpret_cast<const f32x4*>(r1 + d);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d0 = x0[e] - m[e], d1 = x1[e] - m[e];
                        acc0 += (double)d0 * (double)d0; acc1 += (double)d1 * (double)d1;
                    }
                }
            } else {
                for (int d = lane; d < D; d += 64) {
                    const float d0 = r0[d] - mean_s[d], d1 = r1[d] - mean_s[d];
                    acc0 += (double)d0 * (double)d0; acc1 += (double)d1 * (double)d1;
                }
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { acc0 += __shfl_xor(acc0, o); acc1 += __shfl_xor(acc1, o); }
            if (lane == 0) { if (ok0) dist_s[i0] = sqrt(acc0); if (ok1) dist_s[i1] = sqrt(acc1); }
        }
        __syncthreads();
        // arg-max over the alive rows (ties: the later row, as a stable ascending sort would drop)
        double bd = -1.0; int bi = -1;
        for (int i = tid; i < n; i += kPruneThreads)
            if (alive_s[i] && (dist_s[i] > bd || (dist_s[i] == bd && i > bi))) { bd = dist_s[i]; bi = i; }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double d2 = __shfl_xor(bd, o); const int i2 = __shfl_xor(bi, o);
            if (d2 > bd || (d2 == bd && i2 > bi)) { bd = d2; bi = i2; }
        }
        if (lane == 0) { red_d[wave] = bd; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kPruneWaves; ++w)
                if (red_d[w] > red_d[0] || (red_d[w] == red_d[0] && red_i[w] > red_i[0])) { red_d[0] = red_d[w]; red_i[0] = red_i[w]; }
            drop_s = red_i[0];
            alive_s[red_i[0]] = 0;
            a.dropped[t] = red_i[0];
        }
        __syncthreads();
        const int j = drop_s;
        for (int d = tid; d < D; d += kPruneThreads) sum_s[d] -= (double)a.rows[(int64_t)j * a.ld + d];
        --count;
        __syncthreads();
    }
    for (int i = tid; i < n; i += kPruneThreads) { a.alive[i] = alive_s[i]; a.dist[i] = dist_s[i]; }
    for (int d = tid; d < D; d += kPruneThreads) a.sum[d] = sum_s[d];
}

}  // namespace

extern "C" int ac_memory_add_prune(const ac_prune_job* h_jobs, const ac_prune_job* d_jobs, int njobs, int D,
                                   ac_stream_t stream) {
    AC_REQUIRE(njobs >= 0 && D >= 1 && (njobs == 0 || (h_jobs && d_jobs)), AC_EINVAL, "memory_add_prune: bad arguments");
    AC_REQUIRE(D <= kPruneMaxD, AC_EUNSUPPORTED, "memory_add_prune: D=%d exceeds %d", D, kPruneMaxD);
    if (njobs == 0) return AC_OK;
    size_t lds = 0;
    for (int j = 0; j < njobs; ++j) {
        const ac_prune_job& a = h_jobs[j];
        AC_REQUIRE(a.rows && a.sum && a.alive && a.dist && a.dropped, AC_EINVAL, "memory_add_prune: job %d has a null pointer", j);
        AC_REQUIRE(a.n_old >= 0 && a.n_new >= 1 && a.cap >= 1 && a.ld >= D && a.n_old <= a.cap, AC_EINVAL,
                   "memory_add_prune: job %d bad sizes (n_old=%d n_new=%d cap=%d)", j, a.n_old, a.n_new, a.cap);
        const int n = a.n_old + a.n_new;
        AC_REQUIRE(n <= kPruneMaxRows, AC_EUNSUPPORTED, "memory_add_prune: job %d has %d rows (max %d)", j, n, kPruneMaxRows);
        const size_t need = (size_t)((n + 1) / 2 * 2) * 8 + (size_t)D * 8 + (size_t)D * 4 + (size_t)n + 16;
        if (need > lds) lds = need;
    }
    AC_REQUIRE(lds <= 150 * 1024, AC_EUNSUPPORTED, "memory_add_prune: needs %zu B of LDS", lds);
    (void)hipFuncSetAttribute((const void*)class_add_prune_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(class_add_prune_kernel, dim3(njobs), dim3(kPruneThreads), lds, (hipStream_t)stream, d_jobs, D);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
