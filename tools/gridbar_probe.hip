// Probe: cost of a grid-wide barrier between co-resident workgroups on MI355X (persistent-kernel phases).
//   hipcc -O2 --offload-arch=gfx950 tools/gridbar_probe.hip -o tools/ab/gridbar_probe && tools/ab/gridbar_probe
// Variants: 0 flat counter, all-thread fences;  1 flat counter, thread-0 fences only;  2 two-level (per-XCD then global).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Ctl { unsigned flat; unsigned pad0[63]; unsigned xcd[64][64]; unsigned top; unsigned pad1[63]; unsigned abort_; };

template <int V>
__device__ __forceinline__ void grid_barrier(Ctl* c, unsigned epoch, unsigned G) {
    if (V >= 16) {                   // one level, V group counters, lanes 0 .. V-1 poll
        constexpr unsigned NG = V;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < 64) {
            const unsigned lane = threadIdx.x;
            if (lane == 0) __hip_atomic_fetch_add(&c->xcd[blockIdx.x % NG][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned grp = lane % NG, ng = (G + NG - 1 - grp) / NG;
            long spins = 0;
            for (;;) {
                const unsigned v = __hip_atomic_load(&c->xcd[grp][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v >= epoch * ng)) break;
                if (++spins > 20000000) { c->abort_ = 1; break; }
            }
        }
        __syncthreads();
        return;
    }
    if (V == 5) {                    // one level: eight group counters, every waiter polls all eight (lanes 0-7)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < 64) {
            const unsigned lane = threadIdx.x;
            if (lane == 0) __hip_atomic_fetch_add(&c->xcd[blockIdx.x & 7][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned grp = lane & 7, ng = (G + 7 - grp) / 8;
            long spins = 0;
            for (;;) {
                const unsigned v = __hip_atomic_load(&c->xcd[grp][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v >= epoch * ng)) break;
                if (++spins > 20000000) { c->abort_ = 1; break; }
            }
        }
        __syncthreads();
        return;
    }
    if (V >= 3) {                    // fence-free: every wave drains its own stores, counters are relaxed agent-scope atomics
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* spin_on; unsigned target;
            if (V == 3) {
                __hip_atomic_fetch_add(&c->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spin_on = &c->flat; target = epoch * G;
            } else {
                const unsigned x = blockIdx.x & 7, nx = (G + 7 - x) / 8;
                const unsigned old = __hip_atomic_fetch_add(&c->xcd[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == epoch * nx) __hip_atomic_fetch_add(&c->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spin_on = &c->top; target = epoch * (G < 8 ? G : 8);
            }
            long spins = 0;
            while (__hip_atomic_load(spin_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 50000000) { c->abort_ = 1; break; }
            }
        }
        __syncthreads();
        return;
    }
    if (V == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (V <= 1) {
            __hip_atomic_fetch_add(&c->flat, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * G;
            long spins = 0;
            while (__hip_atomic_load(&c->flat, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 50000000) { c->abort_ = 1; break; }
            }
        } else {
            const unsigned x = blockIdx.x & 7, nx = (G + 7 - x) / 8;          // blocks on this XCD (round-robin dispatch)
            const unsigned old = __hip_atomic_fetch_add(&c->xcd[x][0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == epoch * nx) __hip_atomic_fetch_add(&c->top, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * (G < 8 ? G : 8);
            long spins = 0;
            while (__hip_atomic_load(&c->top, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 50000000) { c->abort_ = 1; break; }
            }
        }
    }
    __syncthreads();
    if (V == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// each phase: every block writes one value, after the barrier reads its neighbour's value of that phase (correctness check)
template <int V>
__global__ __launch_bounds__(512) void probe(Ctl* c, unsigned* slots, int phases, unsigned* errors) {
    const unsigned G = gridDim.x;
    unsigned bad = 0;
    for (int p = 1; p <= phases; ++p) {
        if (threadIdx.x == 0) slots[blockIdx.x * 32] = (unsigned)p * 1000003u + blockIdx.x;
        if (threadIdx.x == 64) slots[blockIdx.x * 32 + 16] = (unsigned)p * 7u + blockIdx.x;     // a second wave's store
        grid_barrier<V>(c, (unsigned)p, G);
        const unsigned nb = (blockIdx.x + 1 + (p % 7) * 37) % G;
        if (threadIdx.x == 128) { if (slots[nb * 32] != (unsigned)p * 1000003u + nb) ++bad; }
        if (threadIdx.x == 200) { if (slots[nb * 32 + 16] != (unsigned)p * 7u + nb) ++bad; }
        // second barrier so the next phase's writes cannot overtake this phase's reads
        grid_barrier<V>(c, 100000u + (unsigned)p, G);   // (flat counter keeps counting: handled below by using 2p-1 / 2p)
    }
    if (bad) atomicAdd(errors, bad);
}

// the probe above needs monotone epochs: rewrite with epochs 2p-1, 2p
template <int V>
__global__ __launch_bounds__(512) void probe2(Ctl* c, unsigned* slots, int phases, unsigned* errors) {
    const unsigned G = gridDim.x;
    unsigned bad = 0;
    for (int p = 1; p <= phases; ++p) {
        if (V >= 3) {
            if (threadIdx.x == 0) __hip_atomic_store(&slots[blockIdx.x * 32], (unsigned)p * 1000003u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 64) __hip_atomic_store(&slots[blockIdx.x * 32 + 16], (unsigned)p * 7u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0) slots[blockIdx.x * 32] = (unsigned)p * 1000003u + blockIdx.x;
            if (threadIdx.x == 64) slots[blockIdx.x * 32 + 16] = (unsigned)p * 7u + blockIdx.x;
        }
        grid_barrier<V>(c, 2u * p - 1, G);
        const unsigned nb = (blockIdx.x + 1 + (p % 7) * 37) % G;
        if (V >= 3) {
            if (threadIdx.x == 128) { if (__hip_atomic_load(&slots[nb * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)p * 1000003u + nb) ++bad; }
            if (threadIdx.x == 200) { if (__hip_atomic_load(&slots[nb * 32 + 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)p * 7u + nb) ++bad; }
        } else {
        if (threadIdx.x == 128) { if (slots[nb * 32] != (unsigned)p * 1000003u + nb) ++bad; }
        if (threadIdx.x == 200) { if (slots[nb * 32 + 16] != (unsigned)p * 7u + nb) ++bad; }
        }
        grid_barrier<V>(c, 2u * p, G);
    }
    if (bad) atomicAdd(errors, bad);
}

template <int V>
void run(int G, int phases, bool coop) {
    Ctl* c; unsigned *slots, *err;
    CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&slots, 4096 * 32 * 4)); CK(hipMalloc(&err, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(err, 0, 4));
        CK(hipEventRecord(e0, 0));
        if (coop) {
            void* args[] = {&c, &slots, &phases, &err};
            CK(hipLaunchCooperativeKernel((const void*)probe2<V>, dim3(G), dim3(512), args, 0, 0));
        } else {
            hipLaunchKernelGGL(probe2<V>, dim3(G), dim3(512), 0, 0, c, slots, phases, err);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    unsigned herr; Ctl hc; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("  variant %d G=%4d %s: %.3f us per barrier (%d barriers)  errors=%u abort=%u\n", V, G, coop ? "coop" : "plain",
           best * 1e3 / (2.0 * phases), 2 * phases, herr, hc.abort_);
    hipFree(c); hipFree(slots); hipFree(err);
}

int main() {
    const int phases = 2000;
    for (int G : {128, 256}) {
        run<5>(G, phases, false); run<16>(G, phases, false); run<32>(G, phases, false); run<64>(G, phases, false);
    }

    return 0;
}
