// Deterministic synthetic unit-norm rows, bit-identical to oracle/synth.py.
// Integer hashing + IEEE-correct fp64 sqrt/divide only, so host and device agree bit for bit.
// Stands in for the reference's unit-norm embeddings (classifier.py:1275) at benchmark scale.
#include "common.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ int synth_int(uint64_t rowkey, int col) {
    const uint64_t h = splitmix64(rowkey + (uint64_t)col);
    const int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)(h >> 48);
    return s - 131070;
}

// one wave per row
__global__ __launch_bounds__(256) void synth_rows_kernel(float* out, int64_t n, int64_t ld, int D,
                                                         uint64_t seed, int64_t row_offset) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const uint64_t rowkey = splitmix64(seed ^ ((uint64_t)(r + row_offset) * 0xD1342543DE82EF95ull));
    long long ss = 0;
    for (int c = lane; c < D; c += 64) {
        const long long x = synth_int(rowkey, c);
        ss += x * x;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) ss += __shfl_xor(ss, o);
    double nrm = sqrt((double)ss);
    if (nrm == 0.0) nrm = 1.0;
    float* dst = out + (size_t)r * ld;
    for (int c = lane; c < D; c += 64) dst[c] = (float)((double)synth_int(rowkey, c) / nrm);
    for (int c = D + lane; c < ld; c += 64) dst[c] = 0.f;
}

}  // namespace

extern "C" int ac_synth_unit_rows(float* d_out, int64_t n, int64_t ld, int D, uint64_t seed,
                                  int64_t row_offset, ac_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(d_out && n >= 0 && D >= 1 && ld >= D, AC_EINVAL, "synth: bad arguments");
    if (n == 0) return AC_OK;
    const int64_t blocks = (n + 3) / 4;
    AC_REQUIRE(blocks < 2147483647LL, AC_EINVAL, "synth: too many rows for one launch");
    hipLaunchKernelGGL(synth_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_out, n, ld, D,
                       seed, row_offset);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
