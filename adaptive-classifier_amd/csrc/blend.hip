// Score blend of predict() / predict_batch() on the device (reference classifier.py:447-480 and :1359-1384).
//   combined[c] = wp[c] * sum of the prototype scores of the hits of class c      (fp64, hits in distance order)
//               + wh[c] * head_prob[c]   for the head's top `ncls_head` classes   (stable descending order)
//   order: descending combined score, ties in insertion order (hits in distance order first, then head classes
//   in descending probability) -- Python's stable sort in the reference; scores normalised by their sum.
// One wave per query; classes strided over the lanes; everything in fp64 like the reference's Python floats.
// adaptive_classifier/classifier.py::_blend is the same formula in numpy (general path + test reference).
#include "common.h"

namespace {

constexpr int kBlendMaxC = 2048;

struct Best { double score; int ins; int cls; };

__device__ __forceinline__ bool better(double s, int i, double s2, int i2) { return s > s2 || (s == s2 && i < i2); }

__global__ __launch_bounds__(64) void blend_topk_kernel(const float* __restrict__ S, const int64_t* __restrict__ Cid,
                                                        int kp, const float* __restrict__ P, int C,
                                                        const double* __restrict__ wp, const double* __restrict__ wh,
                                                        int ncls_head, int k, int* __restrict__ out_n,
                                                        int* __restrict__ out_cls, double* __restrict__ out_val) {
    __shared__ double comb[kBlendMaxC];
    __shared__ int ins[kBlendMaxC];          // insertion rank; INT_MAX = class absent
    __shared__ double total_s;
    const int q = blockIdx.x, lane = threadIdx.x;
    constexpr int BIG = 0x7fffffff;
    for (int c = lane; c < C; c += 64) { comb[c] = 0.0; ins[c] = BIG; }
    __syncthreads();
    if (S && lane == 0) {                    // hits in distance order: the summation order of the reference
        for (int j = 0; j < kp; ++j) {
            const int64_t c = Cid[(int64_t)q * kp + j];
            if (c >= 0 && c < C) {
                comb[c] += (double)S[(int64_t)q * kp + j];
                if (ins[c] == BIG) ins[c] = j;
            }
        }
    }
    __syncthreads();
    for (int c = lane; c < C; c += 64) comb[c] *= wp[c];
    if (P) {
        const float* p = P + (int64_t)q * C;
        for (int c = lane; c < C; c += 64) {
            const float pc = p[c];
            int rank = 0;                    // position in the stable descending order (torch.topk order)
            for (int o = 0; o < C; ++o) rank += (p[o] > pc) || (p[o] == pc && o < c);
            if (rank < ncls_head) {
                comb[c] += (double)pc * wh[c];
                if (ins[c] == BIG) ins[c] = kp + rank;
            }
        }
    }
    __syncthreads();
    if (lane == 0) {
        double t = 0.0;
        for (int c = 0; c < C; ++c) if (ins[c] != BIG) t += comb[c];
        total_s = t;
    }
    __syncthreads();
    const double denom = total_s > 0.0 ? total_s : 1.0;
    int n = 0;
    for (int r = 0; r < k; ++r) {
        double bs = 0.0; int bi = BIG, bc = -1;
        for (int c = lane; c < C; c += 64)
            if (ins[c] != BIG && (bc < 0 || better(comb[c], ins[c], bs, bi))) { bs = comb[c]; bi = ins[c]; bc = c; }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double s2 = __shfl_xor(bs, o); const int i2 = __shfl_xor(bi, o), c2 = __shfl_xor(bc, o);
            if (c2 >= 0 && (bc < 0 || better(s2, i2, bs, bi))) { bs = s2; bi = i2; bc = c2; }
        }
        if (bc < 0) break;                   // wave-uniform
        if (lane == 0) {
            out_cls[(int64_t)q * k + r] = bc;
            out_val[(int64_t)q * k + r] = bs / denom;
            ins[bc] = BIG;                   // taken
        }
        ++n;
        __syncthreads();
    }
    if (lane == 0) out_n[q] = n;
}

}  // namespace

extern "C" int ac_blend_topk(const float* d_scores, const int64_t* d_hit_class, int kp, const float* d_head_probs,
                             int C, const double* d_w_proto, const double* d_w_head, int ncls_head, int k, int b,
                             int32_t* d_out_n, int32_t* d_out_class, double* d_out_score, ac_stream_t stream) {
    AC_REQUIRE(d_w_proto && d_w_head && d_out_n && d_out_class && d_out_score, AC_EINVAL, "blend: null pointer");
    AC_REQUIRE((d_scores == nullptr) == (d_hit_class == nullptr), AC_EINVAL, "blend: scores and hit classes go together");
    AC_REQUIRE(d_scores || d_head_probs, AC_EINVAL, "blend: neither prototype hits nor head probabilities");
    AC_REQUIRE(b >= 0 && k >= 1 && kp >= 0 && ncls_head >= 0, AC_EINVAL, "blend: bad sizes");
    AC_REQUIRE(C >= 1 && C <= kBlendMaxC, AC_EUNSUPPORTED, "blend: %d classes (max %d on the device path)", C, kBlendMaxC);
    if (b == 0) return AC_OK;
    hipLaunchKernelGGL(blend_topk_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, d_scores, d_hit_class, kp,
                       d_head_probs, C, d_w_proto, d_w_head, ncls_head, k, d_out_n, d_out_class, d_out_score);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
