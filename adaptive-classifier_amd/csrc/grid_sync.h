// Shared device helpers of the persistent (one-launch) kernels: head_epoch.hip, bert_small.hip.
//   * sc1 accesses: agent-coherent across the 8 XCD L2s (write-through stores, L2-bypassing loads) -- the only way data
//     moves between workgroups inside a launch here; no cache write-back / invalidate anywhere
//   * fence-free grid barrier over co-resident workgroups (cooperative launch)
//   * wave sums on DPP
#pragma once
#include <hip/hip_runtime.h>

namespace acp {

struct GridCtl {
    unsigned xcd[8][32];              // arrivals per group of workgroups (one 128-byte line each)
    unsigned top;
    unsigned abort_;
    unsigned pad[30];
};

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float2 ld2_sc1(const float* p) {          // p 8-byte aligned
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// sc1 (agent-coherent, write-through / L2-bypassing) accesses through a buffer descriptor: ordinary loads to the
// compiler, so a batch of them is issued back to back and waited for once (relaxed atomics are kept in program order
// with a wait after each group).  Offsets beyond `bytes` read as zero.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float4 ld4_buf(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {      // plain (L2-cached) 16-byte load
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Fence-free grid barrier: every wave drains its stores; thread 0 counts its workgroup into one of eight counters
// (128-byte lines apart) with a relaxed agent-scope atomic; lanes 0-7 of wave 0 then poll the eight counters until each
// has seen all of its workgroups.  One atomic + one poll round trip: 1.8 us for 256 workgroups, against 2.4 us for a
// two-level counter and 4.1 us for a single one (tools/gridbar_probe.hip).  `n` = 1, 2, 3 ... over the launch.
// false = gave up (a workgroup never arrived: cannot happen with a cooperative launch; bounded so a bug cannot hang
// the GPU).  The exchanged data itself is written with sc1 stores and read with sc1 loads: no cache maintenance here.
// The two halves of the barrier, for callers that have independent loads to issue in between (they then overlap with the
// barrier's own latency instead of delaying the arrival): grid_arrive() drains this workgroup's stores and counts it in;
// grid_wait() polls.
__device__ __forceinline__ void grid_arrive(GridCtl* c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&c->xcd[blockIdx.x & 7][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool ACQUIRE>
__device__ __forceinline__ bool grid_wait(GridCtl* c, unsigned n, unsigned G, unsigned* lds_flag) {
    if (threadIdx.x < 64) {
        const unsigned lane = threadIdx.x;
        const unsigned grp = lane & 7, target = n * ((G + 7 - grp) / 8);
        unsigned ok = 1;
        for (long spins = 0;; ++spins) {
            const unsigned v = __hip_atomic_load(&c->xcd[grp][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v >= target)) break;
            __builtin_amdgcn_s_sleep(2);                 // (polling flat out slows the stragglers' own memory traffic)
            if ((spins & 1023) == 1023) {
                if (__hip_atomic_load(&c->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins > (1l << 21)) {
                    if (lane == 0) __hip_atomic_store(&c->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        if (ACQUIRE && lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) *lds_flag = ok;
    }
    __syncthreads();
    return *lds_flag != 0;
}
template <bool ACQUIRE>
__device__ __forceinline__ bool grid_barrier(GridCtl* c, unsigned n, unsigned G, unsigned* lds_flag) {
    grid_arrive(c);
    return grid_wait<ACQUIRE>(c, n, G, lds_flag);
}

// Sums over the wave without LDS traffic: four DPP butterflies inside each row of 16 lanes (every lane of a row then
// holds the row sum), the four row sums combined in a fixed order.  ~60 cycles against ~600 for six ds_bpermute steps.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);      // row_half_mirror
    v = dpp_add<0x140>(v);      // row_mirror
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

}  // namespace acp
