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

}  // namespace ac
