// Shared host-side helpers for libacamd.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/acamd.h"

namespace ac {

void set_error(const char* fmt, ...);

struct DevInfo {
    int cus;
    int lds_per_block;   // max dynamic+static LDS per workgroup (bytes)
    size_t hbm_bytes;
};
const DevInfo& dev_info();   // lazily queried for the current device

#define AC_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ac::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                          __FILE__, __LINE__);                                          \
            return AC_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define AC_REQUIRE(cond, code, ...)                                                     \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ac::set_error(__VA_ARGS__);                                                 \
            return (code);                                                              \
        }                                                                               \
    } while (0)

#define AC_LAUNCH_CHECK()                                                               \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            ac::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),    \
                          __FILE__, __LINE__);                                          \
            return AC_EHIP;                                                             \
        }                                                                               \
    } while (0)

__host__ __device__ static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// gemm.hip: fp32 MFMA GEMMs shared by head.hip and bert.hip
//   C[M,N] = epi(A[M,K] . W[N,K]^T): bias, act (0 none / 1 relu / 2 gelu-erf), optional inverted
//   dropout mask (uint8 [M,N], kept values * mask_scale), optional residual added last.
int linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
               const float* residual, int64_t ldr, float* C, int64_t ldc, int M, int N, int K, int act,
               const uint8_t* mask, float mask_scale, hipStream_t stream, float drop_p = 0.f,
               uint64_t drop_seed = 0, const uint16_t* W_planes = nullptr, const uint16_t* A_planes = nullptr);

// counter-based Bernoulli(1-p) keep decision for in-kernel dropout (stateless: seed + element index)
__host__ __device__ static inline bool dropout_keep(uint64_t seed, uint64_t idx, float p) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f) >= p;
}
// arithmetic of the LDS-tiled GEMMs (ac_gemm_set_arith / env AC_GEMM_ARITH = f32 | bf16x3)
int gemm_arith();
void set_gemm_arith(int mode);
//   C = alpha * op(A) op(B) + beta * C; if gate != null: C = gate[m,n] != 0 ? C * gate_scale : 0
int gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda,
             const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const float* gate,
             int64_t ldg, float gate_scale, hipStream_t stream);

}  // namespace ac
