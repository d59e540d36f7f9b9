// Score blend of predict() / predict_batch() on the device (reference classifier.py:447-480 and :1359-1384).
//   combined[c] = wp[c] * sum of the prototype scores of the hits of class c      (fp64, hits in distance order)
//               + wh[c] * head_prob[c]   for the head's top `ncls_head` classes   (stable descending order)
//   order: descending combined score, ties in insertion order (hits in distance order first, then head classes
//   in descending probability) -- Python's stable sort in the reference; scores normalised by their sum.
// One wave per query; classes strided over the lanes; everything in fp64 like the reference's Python floats.
// adaptive_classifier/classifier.py::_blend is the same formula in numpy (general path + test reference).
#include "common.h"
#include "grid_sync.h"
#include <atomic>
#include <mutex>
#include <string.h>

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

// ---- the whole tail of a predict batch in ONE launch (ac_predict_post) ------------------------------------------------------------
// proto_scores_kernel (knn_l2.hip: exp(-d), softmax over the hits) + rows_to_class_kernel (hit row -> classifier class) +
// softmax_rows_kernel (head.hip: F.softmax over the head's outputs) + blend_topk_kernel above, the same arithmetic in the same order
// (the three preludes leave their results in LDS instead of global memory), one wave per query.  The packed result may live in
// host-mapped memory: the last workgroup to finish (agent-scope counter; every workgroup made its stores visible system-wide first)
// publishes `epoch` into a host-mapped flag, so the host neither launches a copy nor sleeps in a stream synchronisation.
constexpr int kPostMaxKp = 1024;

__global__ __launch_bounds__(64) void predict_post_kernel(const float* __restrict__ D, const int64_t* __restrict__ I, int kp,
                                                          const int32_t* __restrict__ row_class, int64_t nrows,
                                                          const int64_t* __restrict__ class_lut, int nlut,
                                                          const float* __restrict__ head, int C, int head_softmax,
                                                          const double* __restrict__ wp, const double* __restrict__ wh,
                                                          int ncls_head, int k, int* out_n, int* out_cls, double* out_val,
                                                          unsigned* done, int b, int* host_flag, int epoch,
                                                          const unsigned long long* stage, unsigned long long* host_out, int words) {
    __shared__ double comb[kBlendMaxC];
    __shared__ int ins[kBlendMaxC];          // insertion rank; INT_MAX = class absent
    __shared__ float sP[kBlendMaxC];
    __shared__ float sS[kPostMaxKp];
    __shared__ int sC[kPostMaxKp];
    __shared__ double res_val[kBlendMaxC];
    __shared__ int res_cls[kBlendMaxC];
    __shared__ double total_s;
    const int q = blockIdx.x, lane = threadIdx.x;
    constexpr int BIG = 0x7fffffff;
    for (int c = lane; c < C; c += 64) { comb[c] = 0.0; ins[c] = BIG; }
    if (D) {
        // memory.py:117 (exp(-d)) and :129-130 (softmax over the hits): proto_scores_kernel's expressions
        const float* Dq = D + (size_t)q * kp;
        const int64_t* Iq = I + (size_t)q * kp;
        float mx = -INFINITY;
        for (int e = lane; e < kp; e += 64)
            if (Iq[e] >= 0) mx = fmaxf(mx, expf(-Dq[e]));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int e = lane; e < kp; e += 64)
            if (Iq[e] >= 0) sum += expf(expf(-Dq[e]) - mx);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
        for (int e = lane; e < kp; e += 64) {
            const int64_t id = Iq[e];
            sS[e] = id >= 0 ? expf(expf(-Dq[e]) - mx) / sum : 0.f;
            int64_t c = -1;                  // rows_to_class_kernel
            if (id >= 0 && id < nrows) {
                c = row_class ? (int64_t)row_class[id] : id;
                if (class_lut) c = (c >= 0 && c < nlut) ? class_lut[c] : -1;
            }
            sC[e] = (c >= 0 && c < C) ? (int)c : -1;
        }
    }
    if (head) {
        const float* zr = head + (size_t)q * C;
        if (head_softmax) {                  // softmax_rows_kernel
            float mx = -INFINITY;
            for (int c = lane; c < C; c += 64) mx = fmaxf(mx, zr[c]);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float sum = 0.f;
            for (int c = lane; c < C; c += 64) sum += expf(zr[c] - mx);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
            for (int c = lane; c < C; c += 64) sP[c] = expf(zr[c] - mx) / sum;
        } else {
            for (int c = lane; c < C; c += 64) sP[c] = zr[c];
        }
    }
    __syncthreads();
    if (D && lane == 0) {                    // hits in distance order: the summation order of the reference
        for (int j = 0; j < kp; ++j) {
            const int c = sC[j];
            if (c >= 0) {
                comb[c] += (double)sS[j];
                if (ins[c] == BIG) ins[c] = j;
            }
        }
    }
    __syncthreads();
    for (int c = lane; c < C; c += 64) comb[c] *= wp[c];
    if (head) {
        for (int c = lane; c < C; c += 64) {
            const float pc = sP[c];
            int rank = 0;                    // position in the stable descending order (torch.topk order)
            for (int o = 0; o < C; ++o) rank += (sP[o] > pc) || (sP[o] == pc && o < c);
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
        if (lane == 0) {                     // (kept in LDS: a global store here would be waited for by every round's barrier)
            res_cls[r] = bc;
            res_val[r] = bs / denom;
            ins[bc] = BIG;                   // taken
        }
        ++n;
        __syncthreads();
    }
    // the query's results in one batch of write-through stores (the packed result never sits dirty in an L2, see below)
    for (int r = lane; r < n; r += 64) {
        __hip_atomic_store(out_cls + (int64_t)q * k + r, res_cls[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(out_val) + (int64_t)q * k + r,
                           (unsigned long long)__double_as_longlong(res_val[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) __hip_atomic_store(out_n + q, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (host_flag) {
        // the packed result sits in device memory (`stage`); the workgroup that finishes LAST (agent-scope counter: release of its
        // own stores, acquire of everyone else's) copies it to the host-mapped buffer in whole 512-byte wave stores -- 256
        // queries' worth of 4- and 8-byte stores straight over the bus cost 12 us more than this -- and publishes the flag
        // Fence-free (the hand-off of gemm_pipe.hip's exchanges): the stage is written with write-through stores and read back
        // with L2-bypassing loads, the host buffer is fine-grained memory (never cached on the device), so "acknowledged"
        // (s_waitcnt vmcnt(0)) is "visible" -- a release / acquire pair here writes back and invalidates whole L2s that are
        // full of the batch's other results: +12 us, measured.
        unsigned old = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __shfl(old, 0);
        if (old == (unsigned)b - 1u) {       // wave-uniform
            asm volatile("" ::: "memory");
            // 16 bytes per lane, eight L2-bypassing loads in flight per lane before the first store (a relaxed atomic load per
            // word would be one memory round trip per 512 bytes)
            const __amdgpu_buffer_rsrc_t rs = acp::make_rsrc(stage, (unsigned)words * 8u);
            const int quads = (words + 1) / 2;                  // (reads beyond the descriptor return zero; host_out has room:
            for (int i0 = 0; i0 < quads; i0 += 8 * 64) {        //  the caller's buffers are multiples of 16 bytes)
                acp::u32x4_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(i0 + 64 * u + lane) * 16u, 0, 16);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 64 * u + lane < quads) reinterpret_cast<acp::u32x4_t*>(host_out)[i0 + 64 * u + lane] = v[u];
            }
            // (the flag and the data travel through different L2 channels: ONE system-scope release, by this workgroup only)
            __threadfence_system();
            if (lane == 0) {
                __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (ready for its next use)
                __hip_atomic_store(host_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// completion slots of ac_predict_post(wait_host = 1): a host-mapped flag + a device counter each, handed out round-robin
struct PostSlot { int* flag; unsigned* done; };
PostSlot post_slot(int* epoch_out) {
    // per device: the counters are device memory of the device the kernel runs on (a process that drives several GPUs)
    struct PerDev { int* flags = nullptr; unsigned* counters = nullptr; bool tried = false; };
    static PerDev devs[64];
    static std::mutex mu;
    static std::atomic<unsigned> next{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    PerDev& d = devs[dev];
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!d.tried) {
            d.tried = true;
            void* h = nullptr; void* c = nullptr;
            if (hipHostMalloc(&h, 64 * 64, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) == hipSuccess &&
                hipMalloc(&c, 64 * 64) == hipSuccess && hipMemset(c, 0, 64 * 64) == hipSuccess) {
                memset(h, 0, 64 * 64);
                d.flags = (int*)h; d.counters = (unsigned*)c;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    const unsigned n = next.fetch_add(1, std::memory_order_relaxed) + 1;
    *epoch_out = (int)(n & 0x3fffffff) + 1;
    PostSlot s{nullptr, nullptr};
    if (d.flags) { s.flag = d.flags + 16 * (n & 63); s.done = d.counters + 16 * (n & 63); }
    return s;
}

}  // namespace

extern "C" int ac_host_alloc(size_t bytes, void** p) {
    AC_REQUIRE(p && bytes > 0, AC_EINVAL, "host_alloc: bad arguments");
    *p = nullptr;
    AC_HIP_CHECK(hipHostMalloc(p, bytes, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable));
    return AC_OK;
}
extern "C" int ac_host_free(void* p) {
    if (p) AC_HIP_CHECK(hipHostFree(p));
    return AC_OK;
}

extern "C" int ac_predict_post(const float* d_dist, const int64_t* d_ids, int kp, const int32_t* d_row_class, int64_t nrows,
                               const int64_t* d_class_lut, int nlut, const float* d_head, int C, int head_softmax,
                               const double* d_w_proto, const double* d_w_head, int ncls_head, int k, int b, void* out,
                               size_t out_bytes, void* h_out, ac_stream_t stream_) {
    const int wait_host = h_out != nullptr;
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(d_w_proto && d_w_head && out, AC_EINVAL, "predict_post: null pointer");
    AC_REQUIRE((d_dist == nullptr) == (d_ids == nullptr), AC_EINVAL, "predict_post: distances and row ids go together");
    AC_REQUIRE(d_dist || d_head, AC_EINVAL, "predict_post: neither prototype hits nor head outputs");
    AC_REQUIRE(b >= 0 && k >= 1 && kp >= 0 && ncls_head >= 0 && nrows >= 0 && nlut >= 0, AC_EINVAL, "predict_post: bad sizes");
    AC_REQUIRE(C >= 1 && C <= kBlendMaxC && kp <= kPostMaxKp, AC_EUNSUPPORTED, "predict_post: %d classes / %d hits per query (max %d / %d)", C, kp,
               kBlendMaxC, kPostMaxKp);
    const size_t off_cls = (size_t)4 * b, off_val = (off_cls + (size_t)4 * b * k + 7) / 8 * 8;
    AC_REQUIRE(out_bytes >= (off_val + (size_t)8 * b * k + 15) / 16 * 16 && (((uintptr_t)out) & 15) == 0 && (((uintptr_t)h_out) & 15) == 0, AC_EWORKSPACE, "predict_post: result buffer %zu < %zu bytes",
               out_bytes, off_val + (size_t)8 * b * k);
    if (b == 0) return AC_OK;
    PostSlot slot{nullptr, nullptr};
    int epoch = 0;
    if (wait_host) {
        slot = post_slot(&epoch);
        AC_REQUIRE(slot.flag, AC_EHIP, "predict_post: no host-mapped completion slot");
    }
    char* base = (char*)out;
    hipLaunchKernelGGL(predict_post_kernel, dim3(b), dim3(64), 0, stream, d_dist, d_ids, d_dist ? kp : 0, d_row_class, nrows, d_class_lut, nlut,
                       d_head, C, head_softmax, d_w_proto, d_w_head, ncls_head, k, (int*)base, (int*)(base + off_cls), (double*)(base + off_val),
                       slot.done, b, slot.flag, epoch, (const unsigned long long*)out, (unsigned long long*)h_out,
                       (int)((off_val + (size_t)8 * b * k) / 8));
    AC_LAUNCH_CHECK();
    if (wait_host) {
        // the stream's work ahead of this kernel is the whole batch (~5 ms): a bounded spin with pauses, then the blocking wait
        volatile int* f = slot.flag;
        bool seen = false;
        for (long spin = 0; spin < 40000000L && !seen; ++spin) {
            seen = __atomic_load_n(f, __ATOMIC_ACQUIRE) == epoch;
            if (!seen) __builtin_ia32_pause();
        }
        if (!seen) {
            AC_HIP_CHECK(hipStreamSynchronize(stream));
            seen = __atomic_load_n(f, __ATOMIC_ACQUIRE) == epoch;
            AC_REQUIRE(seen, AC_EHIP, "predict_post: the kernel's completion flag never arrived");
        }
    }
    return AC_OK;
}

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
