// kNN site of the hot path: exact squared-L2 top-k over the prototype store.
// Replaces faiss.IndexFlatL2.search as called at
//   /root/reference/src/adaptive_classifier/memory.py:113-114
// (third-party faiss-cpu>=1.7.4, requirements.txt:4 -- not vendored).
//
// Three kernels, all on one stream, no host synchronisation:
//
//  1. knn_sweep<TQ>   the HBM sweep.  Every prototype row is read from HBM exactly once per
//                     query tile of TQ queries.  The query tile lives in LDS, pre-scaled by -2
//                     and pre-arranged in MFMA B-fragment order; prototype rows stream
//                     HBM -> VGPR (float4 per lane, no LDS round trip: nothing is shared between
//                     waves) and go straight into v_mfma_f32_32x32x2_f32 / _16x16x4_f32 as the
//                     A operand.  acc[row][query] = |p|^2 - 2 q.p  (|p|^2 is folded in by one
//                     extra MFMA whose A operand is the lane's running sum of squares).
//                     Each lane owns ONE query column, so the running threshold tau_q is one
//                     register; a candidate is pushed to the block's per-query LDS list only if
//                     acc < tau_q (rare after warm-up).  Lists are pruned to the k' = k+pad best
//                     by a wave-level rank-by-counting pass, which also tightens tau_q.
//  2. knn_merge_rerank  per query: radix-select the k' best of the G per-block lists, recompute
//                     those k' distances exactly (fp64 sum of (p-q)^2), order by (exact, id),
//                     emit top-k, and certify with an fp32 error bound that no unseen row can
//                     beat the k-th (see acamd.h "exactness contract").
//  3. knn_exact_fallback + knn_exact_fb_merge  only for queries whose certificate failed: a plain fp64
//                     sweep, parallel over row slabs.
//
// Roofline: algorithmic bytes per sweep = N*D*4; MFMA time at TQ=32 is
// 16 B/clk/CU (> the 10.3 B/clk/CU HBM feed), so the sweep is HBM-bound for nq <= 32.
#include "common.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWaves = 8;               // waves per sweep block (2 per SIMD)
constexpr int kThreads = kWaves * 64;
constexpr int kGroup = 8;               // float4 loads in flight per lane per buffer
constexpr int kPad = 8;                 // extra candidates kept beyond k
constexpr int kMergeMaxCand = 32768;    // G * k' limit (merge kernel keeps 32-bit keys in LDS)
constexpr int kLdsLimit = 160 * 1024;

// One MFMA shape for every query-tile width: v_mfma_f32_16x16x4_f32.  A lane (row i = lane & 15,
// k-slice h = lane >> 4) loads float4 P[row0 + i][16*kb + 4*h ..]: 16 rows x 64 contiguous bytes per
// wave load (two instructions per 128-B line; the 32x32x2 shape would touch 32 rows x 32 B).  A query
// tile is J sub-tiles of 16 queries; the A fragment is reused for the J B-fragments.
struct Shape {
    static constexpr int ROWS = 16, KSPLIT = 4, NACC = 4, KCOLS = 16;
    typedef f32x4 acc_t;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = 4 * (lane >> 4) + r
    static __device__ __forceinline__ int acc_row(int r, int lane) { return 4 * (lane >> 4) + r; }
};

__device__ __attribute__((aligned(16))) float g_knn_zeros[64];     // zero-initialised; tail-group loads of the sweep read it

struct SweepParams {
    const float* P;
    int64_t N;
    int64_t ldP;
    const float* Q;
    int64_t ldQ;
    int D;        // logical dim
    int Dp;       // round_up(D, 4): float4 loads at col < Dp are in bounds (zero padded)
    int ng;       // k-groups per tile: ceil(Dp / (KCOLS * kGroup))
    int nq;
    int kp;       // candidates kept per block and per query (k + pad)
    int cap;      // list capacity, power of two, >= 2 * kp
    int G;        // row groups (blocks per query tile)
    int nqt;      // query tiles
    int64_t ntiles;   // ceil(N / (kWaves * ROWS))
    float* part_d;    // [nqt*TQ][G][kp]
    int32_t* part_i;  // [nqt*TQ][G][kp]
    float* part_maxnorm;  // [G * nqt]
    const float* zeros;   // >= 16 B of zeros, 16-byte aligned (tail-group loads)
    int32_t* clear_ctr;   // [64] the merge / fallback kernels' slot counter and ...
    int32_t* clear_stats; // [4] the caller's d_stats (or NULL): zeroed by workgroup 0 here instead of by two memset launches
};

// monotone map float -> uint32 (ascending float order == ascending unsigned order)
__device__ __forceinline__ uint32_t fkey(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// XCD-aware block id remap (cdna guide T1, bijective form): hardware places block b on XCD
// b % 8; give each XCD a contiguous range of virtual ids so that the nqt query-tile blocks
// of one row group (consecutive virtual ids) share one L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int x = b & 7, s = b >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + s;
}

// Wave-level prune of one candidate list: keep the kp smallest by (d, id), compacted to the front (in no particular order --
// nothing downstream reads the lists as sorted: the merge radix-selects over all of them); update cnt and tau.  n <= cap <= 512.
// Round 4: a radix SELECT over the monotone keys (32 ballots per register slot) instead of rank-by-counting (n dependent
// LDS broadcast reads: ~3 us for a 112-entry list, most of what a sweep with many resident queries spent outside its k-loop).
template <int MAXPER>
__device__ __forceinline__ void prune_list(float* ld, int32_t* li, int* cnt_p, float* tau_p,
                                           int cap, int kp, int lane) {
    int n = *cnt_p;
    if (n > cap) n = cap;
    if (n <= kp) {                                    // nothing to drop (final prune of a short list)
        if (lane == 0) *cnt_p = n;
        return;
    }
    uint32_t key[MAXPER];
    int32_t myi[MAXPER];
#pragma unroll
    for (int e = 0; e < MAXPER; ++e) {
        const int s = lane + 64 * e;
        const bool v = s < n;
        key[e] = v ? fkey(ld[s]) : 0xffffffffu;       // (a real key is never 0xffffffff: that would be a NaN)
        myi[e] = v ? li[s] : 0x7fffffff;
    }
    // T = the kp-th smallest key: fix its bits from the top; `want` = its rank among the entries that share the bits fixed so far
    uint32_t T = 0;
    int want = kp;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t hi = bit == 31 ? 0u : ~((2u << bit) - 1u);
        int c0 = 0;
#pragma unroll
        for (int e = 0; e < MAXPER; ++e)
            c0 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[e] != 0xffffffffu && (key[e] & hi) == T && !((key[e] >> bit) & 1u)));
        if (want > c0) { want -= c0; T |= 1u << bit; }
    }
    // entries below T stay, entries equal to T: the `want` lowest ids of them (ids are unique)
    int c_eq = 0;
#pragma unroll
    for (int e = 0; e < MAXPER; ++e) c_eq += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[e] == T));
    uint32_t idT = 0xffffffffu;
    if (c_eq > want) {                                // (wave-uniform, rare: several candidates share the boundary value)
        idT = 0;
        int w2 = want;
#pragma unroll 1
        for (int bit = 30; bit >= 0; --bit) {         // ids are non-negative 31-bit numbers
            const uint32_t hi = ~((2u << bit) - 1u);
            int c0 = 0;
#pragma unroll
            for (int e = 0; e < MAXPER; ++e)
                c0 += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[e] == T && ((uint32_t)myi[e] & hi) == idT && !(((uint32_t)myi[e] >> bit) & 1u)));
            if (w2 > c0) { w2 -= c0; idT |= 1u << bit; }
        }
    }
    // all reads above are complete for the whole wave before any lane writes (the entries sit in registers)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    int base = 0;
#pragma unroll
    for (int e = 0; e < MAXPER; ++e) {
        const bool keep = key[e] < T || (key[e] == T && (uint32_t)myi[e] <= idT);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {
            const int pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            ld[pos] = fkey_inv(key[e]);
            li[pos] = myi[e];
        }
        base += __builtin_popcountll(m);
    }
    if (lane == 0) { *cnt_p = kp; *tau_p = fkey_inv(T); }
}

__device__ __forceinline__ void prune_dispatch(float* ld, int32_t* li, int* cnt_p, float* tau_p,
                                               int cap, int kp, int lane) {
    if (cap <= 64) prune_list<1>(ld, li, cnt_p, tau_p, cap, kp, lane);
    else if (cap <= 128) prune_list<2>(ld, li, cnt_p, tau_p, cap, kp, lane);
    else if (cap <= 256) prune_list<4>(ld, li, cnt_p, tau_p, cap, kp, lane);
    else prune_list<8>(ld, li, cnt_p, tau_p, cap, kp, lane);
}

template <int J>
__global__ __launch_bounds__(kThreads, 2) void knn_sweep(SweepParams prm) {
    typedef Shape S;
    typedef S::acc_t acc_t;
    static_assert(J == 1 || J == 2, "query tile = 1 or 2 sub-tiles of 16 (LDS budget; tau[] reload below)");
    constexpr int TQ = 16 * J;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nblk = prm.G * prm.nqt;
    const int v = xcd_remap(blockIdx.x, nblk);
    const int qt = v % prm.nqt;
    const int g = v / prm.nqt;
    if (blockIdx.x == 0 && tid < 64) {          // consumed by the kernels launched after this one
        prm.clear_ctr[tid] = 0;
        if (tid < 4 && prm.clear_stats) prm.clear_stats[tid] = 0;
    }

    // ---- LDS carve-up (all offsets multiples of 16) ----
    const int nslots = J * prm.ng * kGroup * 64;      // float4 slots of the query tile
    f32x4* Qs = reinterpret_cast<f32x4*>(smem);
    float* list_d = reinterpret_cast<float*>(smem + (size_t)nslots * 16);
    int32_t* list_i = reinterpret_cast<int32_t*>(list_d + (size_t)TQ * prm.cap);
    int* cnt = reinterpret_cast<int*>(list_i + (size_t)TQ * prm.cap);
    float* tau_s = reinterpret_cast<float*>(cnt + TQ);
    float* wmax_s = tau_s + TQ;                       // [kWaves]

    // ---- stage the query tile: slot (jj, kb, ksub, j) = -2 * Q[qt*TQ + 16*jj + j][16*kb + 4*ksub ..+3] ----
    {
        const int c4_per_q = prm.ng * kGroup * S::KSPLIT;   // float4 columns per query (padded)
        const int total = TQ * c4_per_q;
        for (int t = tid; t < total; t += kThreads) {
            const int j = t / c4_per_q;
            const int c4 = t - j * c4_per_q;
            const int kb = c4 / S::KSPLIT, ksub = c4 - kb * S::KSPLIT;
            const int qrow = qt * TQ + j;
            const int col = 4 * c4;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (qrow < prm.nq && col < prm.D) {
                const float* src = prm.Q + (size_t)qrow * prm.ldQ + col;
                val.x = -2.f * src[0];
                if (col + 1 < prm.D) val.y = -2.f * src[1];
                if (col + 2 < prm.D) val.z = -2.f * src[2];
                if (col + 3 < prm.D) val.w = -2.f * src[3];
            }
            Qs[((j >> 4) * (prm.ng * kGroup) + kb) * 64 + ksub * 16 + (j & 15)] = val;
        }
        for (int t = tid; t < TQ; t += kThreads) {
            cnt[t] = 0;
            tau_s[t] = (qt * TQ + t < prm.nq) ? INFINITY : -INFINITY;
        }
    }
    __syncthreads();

    const int j = lane & 15;            // query column (within each sub-tile) this lane owns in C/D
    const int ksub = lane / S::ROWS;    // k sub-slice this lane feeds in the A/B layout
    const int arow = lane % S::ROWS;    // tile row this lane feeds in the A layout
    float tau[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) tau[jj] = tau_s[16 * jj + j];
    float wave_maxnorm = 0.f;

    // tiles of this block: T = it * G + g
    const int64_t my_tiles = (prm.ntiles > g) ? (prm.ntiles - 1 - g) / prm.G + 1 : 0;
    const int ng = prm.ng;
    const int64_t total = my_tiles * ng;

    f32x4 buf[2][kGroup];
    // prefetch state (flattened group counter -> tile, group)
    int64_t pf_tile = 0;
    int pf_grp = 0;
    const float* pf_ptr;
    auto tile_rowptr = [&](int64_t it) -> const float* {
        int64_t row = (it * prm.G + g) * (int64_t)(kWaves * S::ROWS) + wave * S::ROWS + arow;
        if (row > prm.N - 1) row = prm.N - 1;
        return prm.P + (size_t)row * prm.ldP;
    };
    pf_ptr = tile_rowptr(0);

#define AC_PREFETCH(B)                                                                   \
    do {                                                                                 \
        const int kb0 = pf_grp * kGroup;                                                 \
        if ((kb0 + kGroup) * S::KCOLS <= prm.Dp) { /* wave-uniform: whole group in bounds */ \
            _Pragma("unroll") for (int u = 0; u < kGroup; ++u)                           \
                buf[B][u] = *reinterpret_cast<const f32x4*>(pf_ptr + 4 * ksub + (kb0 + u) * S::KCOLS); \
        } else { /* tail group: out-of-range float4s are fetched from a zero block instead */ \
            _Pragma("unroll") for (int u = 0; u < kGroup; ++u) {                         \
                const int col = (kb0 + u) * S::KCOLS + 4 * ksub;                         \
                const float* src = col < prm.Dp ? pf_ptr + col : prm.zeros;              \
                buf[B][u] = *reinterpret_cast<const f32x4*>(src);                        \
            }                                                                            \
        }                                                                                \
        if (++pf_grp == ng) { pf_grp = 0; ++pf_tile; pf_ptr = tile_rowptr(pf_tile); }    \
    } while (0)

    acc_t acc[J];
    float nsq = 0.f;
    int64_t cur_tile = 0;
    int cur_grp = 0;

    auto epilogue = [&]() {
        // fold |p|^2 in: A = this lane's partial sum of squares, B = 1
#pragma unroll
        for (int jj = 0; jj < J; ++jj) acc[jj] = S::mfma(nsq, 1.0f, acc[jj]);
        // row norm for the error-bound certificate
        float rn = nsq;                     // lanes i, i+16, i+32, i+48 hold the 4 k-slices of row i
        rn += __shfl_xor(rn, 16);
        rn += __shfl_xor(rn, 32);
        wave_maxnorm = fmaxf(wave_maxnorm, rn);

        const int64_t row_base = (cur_tile * prm.G + g) * (int64_t)(kWaves * S::ROWS) + wave * S::ROWS;
        bool maybe = false;
#pragma unroll
        for (int jj = 0; jj < J; ++jj)
#pragma unroll
            for (int r = 0; r < S::NACC; ++r) maybe |= (acc[jj][r] < tau[jj]);
        unsigned done = 0;
        bool pend = __any(maybe) != 0;
        for (;;) {
            bool lane_pend = false;
            if (pend) {
#pragma unroll
                for (int jj = 0; jj < J; ++jj) {
                    const int q = 16 * jj + j;
#pragma unroll
                    for (int r = 0; r < S::NACC; ++r) {
                        const int64_t row = row_base + S::acc_row(r, lane);
                        const float d = acc[jj][r];
                        const unsigned bit = 1u << (jj * S::NACC + r);
                        if (!(done & bit) && d < tau[jj] && row < prm.N) {
                            const int slot = atomicAdd(&cnt[q], 1);
                            if (slot < prm.cap) {
                                list_d[q * prm.cap + slot] = d;
                                list_i[q * prm.cap + slot] = (int32_t)row;
                                done |= bit;
                            } else {
                                lane_pend = true;
                            }
                        }
                    }
                }
            }
            // one barrier per tile: decide (uniformly) whether lists must be pruned
            int over = 0;
            if (tid < TQ) over = cnt[tid] > (prm.cap - prm.cap / 4);
            const int need = __syncthreads_or((int)lane_pend | over);
            if (!need) break;
            for (int q = wave; q < TQ; q += kWaves) {
                if (cnt[q] > prm.kp)
                    prune_dispatch(list_d + q * prm.cap, list_i + q * prm.cap, &cnt[q], &tau_s[q],
                                   prm.cap, prm.kp, lane);
            }
            __syncthreads();
            tau[0] = tau_s[j];
            if (J > 1) tau[J - 1] = tau_s[16 * (J - 1) + j];     // J is 1 or 2
            pend = true;   // re-test un-pushed entries against the tightened tau
        }
        nsq = 0.f;
    };

#define AC_COMPUTE(B)                                                                    \
    do {                                                                                 \
        if (cur_grp == 0) {                                                              \
            _Pragma("unroll") for (int jj = 0; jj < J; ++jj)                             \
                _Pragma("unroll") for (int r = 0; r < S::NACC; ++r) acc[jj][r] = 0.f;    \
        }                                                                                \
        const f32x4* qsrc = Qs + (size_t)cur_grp * kGroup * 64 + lane;                   \
        const size_t jstride = (size_t)ng * kGroup * 64;                                 \
        f32x4 bq[J];                                                                     \
        _Pragma("unroll") for (int jj = 0; jj < J; ++jj) bq[jj] = qsrc[jj * jstride];    \
        _Pragma("unroll") for (int u = 0; u < kGroup; ++u) {                             \
            const f32x4 a = buf[B][u];                                                   \
            f32x4 b[J];                                                                  \
            _Pragma("unroll") for (int jj = 0; jj < J; ++jj) b[jj] = bq[jj];             \
            if (u + 1 < kGroup) { /* LDS reads one step ahead */                         \
                _Pragma("unroll") for (int jj = 0; jj < J; ++jj) bq[jj] = qsrc[jj * jstride + (u + 1) * 64]; \
            }                                                                            \
            _Pragma("unroll") for (int jj = 0; jj < J; ++jj) acc[jj] = S::mfma(a.x, b[jj].x, acc[jj]); \
            nsq = fmaf(a.x, a.x, nsq); nsq = fmaf(a.y, a.y, nsq);                        \
            _Pragma("unroll") for (int jj = 0; jj < J; ++jj) acc[jj] = S::mfma(a.y, b[jj].y, acc[jj]); \
            nsq = fmaf(a.z, a.z, nsq); nsq = fmaf(a.w, a.w, nsq);                        \
            _Pragma("unroll") for (int jj = 0; jj < J; ++jj) acc[jj] = S::mfma(a.z, b[jj].z, acc[jj]); \
            _Pragma("unroll") for (int jj = 0; jj < J; ++jj) acc[jj] = S::mfma(a.w, b[jj].w, acc[jj]); \
            __builtin_amdgcn_sched_barrier(0); /* keep the per-load consume order */     \
        }                                                                                \
        if (++cur_grp == ng) { epilogue(); cur_grp = 0; ++cur_tile; }                    \
    } while (0)

    // Loads are issued unconditionally (past the end they re-read the last row, clamped in
    // tile_rowptr) so that every path has the same number of loads in flight: a load inside a
    // branch makes hipcc's s_waitcnt accounting wait on the buffer it has just issued.
    if (total > 0) {
        AC_PREFETCH(0);
        for (int64_t gg = 0; gg < total; gg += 2) {
            AC_PREFETCH(1);
            AC_COMPUTE(0);
            AC_PREFETCH(0);
            if (gg + 1 < total) AC_COMPUTE(1);
        }
    }
#undef AC_PREFETCH
#undef AC_COMPUTE

    // ---- final: sort + cut every list to kp, write the block's partial result ----
    __syncthreads();
    for (int q = wave; q < TQ; q += kWaves) {
        prune_dispatch(list_d + q * prm.cap, list_i + q * prm.cap, &cnt[q], &tau_s[q], prm.cap,
                       prm.kp, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int n = cnt[q];
        const size_t base = ((size_t)(qt * TQ + q) * prm.G + g) * prm.kp;
        for (int e = lane; e < prm.kp; e += 64) {
            prm.part_d[base + e] = e < n ? list_d[q * prm.cap + e] : INFINITY;
            prm.part_i[base + e] = e < n ? list_i[q * prm.cap + e] : -1;
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) wave_maxnorm = fmaxf(wave_maxnorm, __shfl_xor(wave_maxnorm, o));
    if (lane == 0) wmax_s[wave] = wave_maxnorm;
    __syncthreads();
    if (tid == 0) {
        float m = 0.f;
        for (int w = 0; w < kWaves; ++w) m = fmaxf(m, wmax_s[w]);
        prm.part_maxnorm[v] = m;
    }
}

// --------------------------------------------------------------------------------------
// knn_sweep_ring: the sweep for <= 16 resident queries (the HBM-bound regime the roofline target is stated for), with the
// store rows staged through wave-private LDS rings by whole-line NON-TEMPORAL LDS-DMA and the query tile held in REGISTERS.
//   Why (profiles/r03/hbm_read_ceiling.txt, tools/sweep_ring_bench.hip): plain 16-byte loads stream at <= 6.1 TB/s on an
//   MI355X whatever the occupancy -- knn_sweep<1> sits at that ceiling (6.13 - 6.19) -- while non-temporal accesses reach
//   6.8 - 7.0.  knn_sweep's operand layout (4 lanes per row: 64 bytes of 16 rows per instruction, the other half of every line
//   with the NEXT instruction out of the L1) cannot use them: a non-temporal load does not stay in the L1 (6.13 -> 5.21 TB/s).
//   Here one DMA instruction moves 8 rows x 128 bytes = eight whole lines (8 consecutive lanes per line) into LDS; the
//   fragments are read back in the MFMA layout.  LDS slot of (row a of the 8, 16-byte piece p) = 8 a + (p ^ a): the four
//   k-slices of a row group spread over the banks.  With the rows in LDS the query tile no longer fits there (48 KB at
//   D = 768) -- its B fragments live in registers for the first 512 dimensions (128 VGPRs; all 192 made hipcc spill and reload
//   inside the k-loop) and in 16 KB of LDS for the rest (16 KB at D <= 768, 32 KB at D <= 1024); hence D <= 1024, D % 32 == 0 and <= 16 queries.
//   Per output element the same MFMA sequence in the same k order as knn_sweep<1>; candidate lists, pruning, the row-norm
//   maximum and the partial results are knn_sweep's, unchanged.
// A chunk = 16 rows x 32 floats (2 DMA instructions); RINGC chunks per wave; each wave waits on its OWN DMA queue
// (counted vmcnt), no barrier in the k-loop; the block meets once per 128-row tile in the list-maintenance barrier.
// MAXCH = chunks per row the instantiation is unrolled for: 24 (D <= 768) or 32 (D <= 1024); the query fragments of the first
// ring_reg_chunks() chunks stay in registers (8 VGPRs per chunk), the rest sit in LDS (the longer unrolled loop of the 32-chunk
// form left hipcc 12 bytes short with 16 chunks in registers)
constexpr int ring_reg_chunks(int maxch) { return maxch <= 24 ? 16 : 14; }
constexpr int ring_qs_bytes(int maxch) { return (maxch - ring_reg_chunks(maxch)) * 2 * 64 * 16; }

template <int N> __device__ __forceinline__ void sweep_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int RINGC, int MAXCH>
__global__ __launch_bounds__(kThreads, 2) void knn_sweep_ring(SweepParams prm) {
    constexpr int kRingMaxChunks = MAXCH, kRingRegChunks = ring_reg_chunks(MAXCH), kRingQsBytes = ring_qs_bytes(MAXCH);
    typedef Shape S;
    typedef S::acc_t acc_t;
    constexpr int J = 1, TQ = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void glb_void_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int v = xcd_remap(blockIdx.x, prm.G);       // (nqt == 1)
    const int g = v;
    if (blockIdx.x == 0 && tid < 64) {          // consumed by the kernels launched after this one
        prm.clear_ctr[tid] = 0;
        if (tid < 4 && prm.clear_stats) prm.clear_stats[tid] = tid == 1 ? 1 : 0;      // [1] = 1: the rows went through the LDS ring
    }
    // ---- LDS: rings [kWaves][RINGC][2 halves][64 lanes] x 16 B | lists | counters ----
    uint4* ring = reinterpret_cast<uint4*>(smem) + (size_t)wave * RINGC * 128;
    f32x4* Qs = reinterpret_cast<f32x4*>(smem + (size_t)kWaves * RINGC * 2048);      // B fragments of chunks >= kRingRegChunks: [kb][lane]
    float* list_d = reinterpret_cast<float*>(smem + (size_t)kWaves * RINGC * 2048 + kRingQsBytes);
    int32_t* list_i = reinterpret_cast<int32_t*>(list_d + (size_t)TQ * prm.cap);
    int* cnt = reinterpret_cast<int*>(list_i + (size_t)TQ * prm.cap);
    float* tau_s = reinterpret_cast<float*>(cnt + TQ);
    float* wmax_s = tau_s + TQ;                       // [kWaves]
    // [2][8] per-tile verdicts of the waves, accessed with relaxed workgroup-scope atomics (= plain ds_read / ds_write).  As
    // `volatile int*` accesses they were FLAT loads -- address-space inference leaves volatile accesses alone -- and a flat load
    // counts on vmcnt too: every tile waited vmcnt(0), i.e. drained the whole DMA ring, for them.
    __shared__ int need_s[2 * 8];
    int bar_parity = 0;
    for (int t = tid; t < TQ; t += kThreads) {
        cnt[t] = 0;
        tau_s[t] = (t < prm.nq) ? INFINITY : -INFINITY;
    }
    const int nch = prm.D >> 5;                       // chunks per row (D % 32 == 0)
    const int j = lane & 15;            // query column this lane owns in C/D and feeds in B
    const int ksub = lane / S::ROWS;    // k sub-slice this lane feeds in the A/B layout
    const int arow = lane % S::ROWS;    // tile row this lane feeds in the A layout
    // ---- the query tile as B fragments: -2 * Q[j][16 kb + 4 ksub ..] ----
    f32x4 bq[2 * kRingRegChunks];
    const bool q_vec = (prm.ldQ & 3) == 0 && (((uintptr_t)prm.Q) & 15) == 0;        // 16-byte aligned query rows
#pragma unroll
    for (int kb = 0; kb < 2 * kRingMaxChunks; ++kb) {
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (kb < 2 * nch && j < prm.nq) {
            const float* src = prm.Q + (size_t)j * prm.ldQ + 16 * kb + 4 * ksub;
            if (q_vec) val = -2.f * *reinterpret_cast<const f32x4*>(src);
            else { val.x = -2.f * src[0]; val.y = -2.f * src[1]; val.z = -2.f * src[2]; val.w = -2.f * src[3]; }
        }
        if (kb < 2 * kRingRegChunks) bq[kb < 2 * kRingRegChunks ? kb : 0] = val;
        else if (wave == 0) Qs[(kb - 2 * kRingRegChunks) * 64 + lane] = val;
    }
    __syncthreads();
    float tau[J];
    tau[0] = tau_s[j];
    float wave_maxnorm = 0.f;

    const int64_t my_tiles = (prm.ntiles > g) ? (prm.ntiles - 1 - g) / prm.G + 1 : 0;
    const int64_t total = my_tiles * nch;             // chunks this wave streams

    // ---- DMA stream of this wave: lane (a = lane / 8, piece = (lane % 8) ^ a) of instruction h copies 16 bytes of row 8 h + a
    const int da = lane >> 3, dp = (lane & 7) ^ da;
    int64_t is_tile = 0;                              // tile / chunk of the next DMA issue
    int is_ch = 0, is_slot = 0;
    const float* rp0;
    const float* rp1;
    auto tile_rows = [&](int64_t it) {
        int64_t r0 = (it * prm.G + g) * (int64_t)(kWaves * S::ROWS) + wave * S::ROWS + da, r1 = r0 + 8;
        if (r0 > prm.N - 1) r0 = prm.N - 1;
        if (r1 > prm.N - 1) r1 = prm.N - 1;
        rp0 = prm.P + (size_t)r0 * prm.ldP + 4 * dp;
        rp1 = prm.P + (size_t)r1 * prm.ldP + 4 * dp;
    };
    tile_rows(0);
    auto issue = [&]() {                              // always two DMA instructions (exact vmcnt accounting); past the end: the last chunk again
        uint4* dst = ring + (size_t)is_slot * 128;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(rp0 + 32 * is_ch), (lds_void_t*)dst, 16, 0, 2);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(rp1 + 32 * is_ch), (lds_void_t*)(dst + 64), 16, 0, 2);
        is_slot = is_slot + 1 == RINGC ? 0 : is_slot + 1;
        if (is_ch + 1 < nch) ++is_ch;
        else if (is_tile + 1 < my_tiles) { is_ch = 0; ++is_tile; tile_rows(is_tile); }
    };

    acc_t acc[J];
    float nsq = 0.f;
    int64_t cur_tile = 0;

    auto epilogue = [&]() {
        // fold |p|^2 in: A = this lane's partial sum of squares, B = 1
        acc[0] = S::mfma(nsq, 1.0f, acc[0]);
        // row norm for the error-bound certificate
        float rn = nsq;                     // lanes i, i+16, i+32, i+48 hold the 4 k-slices of row i
        rn += __shfl_xor(rn, 16);
        rn += __shfl_xor(rn, 32);
        wave_maxnorm = fmaxf(wave_maxnorm, rn);

        const int64_t row_base = (cur_tile * prm.G + g) * (int64_t)(kWaves * S::ROWS) + wave * S::ROWS;
        bool maybe = false;
#pragma unroll
        for (int r = 0; r < S::NACC; ++r) maybe |= (acc[0][r] < tau[0]);
        unsigned done = 0;
        bool pend = __any(maybe) != 0;
        for (;;) {
            bool lane_pend = false;
            if (pend) {
                const int q = j;
#pragma unroll
                for (int r = 0; r < S::NACC; ++r) {
                    const int64_t row = row_base + S::acc_row(r, lane);
                    const float d = acc[0][r];
                    const unsigned bit = 1u << r;
                    if (!(done & bit) && d < tau[0] && row < prm.N) {
                        const int slot = atomicAdd(&cnt[q], 1);
                        if (slot < prm.cap) {
                            list_d[q * prm.cap + slot] = d;
                            list_i[q * prm.cap + slot] = (int32_t)row;
                            done |= bit;
                        } else {
                            lane_pend = true;
                        }
                    }
                }
            }
            // one barrier per tile: decide (uniformly) whether lists must be pruned.  NOT __syncthreads_or: its fence waits
            // vmcnt(0), i.e. drains this wave's DMA ring once per 128 rows.  Only LDS traffic has to be ordered here: every
            // wave posts its verdict in a flag word of this tile's parity, waits for its LDS operations, meets the others at
            // a raw s_barrier and reads the eight words (the other parity is rewritten only after the next barrier).
            int over = 0;
            if (tid < TQ) over = cnt[tid] > (prm.cap - prm.cap / 4);
            const int w_need = __any((int)lane_pend | over) != 0;
            const int fl = 8 * (bar_parity & 1);
            if (lane == 0) __hip_atomic_store(&need_s[fl + wave], w_need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int need = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) need |= __hip_atomic_load(&need_s[fl + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++bar_parity;
            if (!need) break;
            for (int q = wave; q < TQ; q += kWaves) {
                if (cnt[q] > prm.kp)
                    prune_dispatch(list_d + q * prm.cap, list_i + q * prm.cap, &cnt[q], &tau_s[q],
                                   prm.cap, prm.kp, lane);
            }
            __syncthreads();
            tau[0] = tau_s[j];
            pend = true;   // re-test un-pushed entries against the tightened tau
        }
        nsq = 0.f;
    };

    if (total > 0) {
#pragma unroll
        for (int i = 0; i < RINGC - 1; ++i) issue();
        const int rh = arow >> 3, ra = arow & 7;
        int slot = 0;
        for (int64_t it = 0; it < my_tiles; ++it) {
            acc[0] = acc_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < kRingMaxChunks; ++ch) {               // (unrolled: bq[] is indexed statically)
                if (ch < nch) {                                         // wave-uniform
                    sweep_wait_vm<(RINGC - 2) * 2>();                   // this chunk has landed (the wave's own DMA queue)
                    __builtin_amdgcn_sched_barrier(0);
                    const uint4* sl = ring + (size_t)slot * 128 + rh * 64 + 8 * ra;
                    const f32x4 a0 = __builtin_bit_cast(f32x4, sl[ksub ^ ra]);
                    const f32x4 a1 = __builtin_bit_cast(f32x4, sl[(4 + ksub) ^ ra]);
                    slot = slot + 1 == RINGC ? 0 : slot + 1;
                    issue();                                            // refills the slot the PREVIOUS chunk used (its reads are consumed)
                    f32x4 b0, b1;
                    if (ch < kRingRegChunks) { b0 = bq[ch < kRingRegChunks ? 2 * ch : 0]; b1 = bq[ch < kRingRegChunks ? 2 * ch + 1 : 0]; }
                    else { b0 = Qs[(2 * (ch - kRingRegChunks)) * 64 + lane]; b1 = Qs[(2 * (ch - kRingRegChunks) + 1) * 64 + lane]; }
                    acc[0] = S::mfma(a0.x, b0.x, acc[0]);
                    nsq = fmaf(a0.x, a0.x, nsq); nsq = fmaf(a0.y, a0.y, nsq);
                    acc[0] = S::mfma(a0.y, b0.y, acc[0]);
                    nsq = fmaf(a0.z, a0.z, nsq); nsq = fmaf(a0.w, a0.w, nsq);
                    acc[0] = S::mfma(a0.z, b0.z, acc[0]);
                    acc[0] = S::mfma(a0.w, b0.w, acc[0]);
                    acc[0] = S::mfma(a1.x, b1.x, acc[0]);
                    nsq = fmaf(a1.x, a1.x, nsq); nsq = fmaf(a1.y, a1.y, nsq);
                    acc[0] = S::mfma(a1.y, b1.y, acc[0]);
                    nsq = fmaf(a1.z, a1.z, nsq); nsq = fmaf(a1.w, a1.w, nsq);
                    acc[0] = S::mfma(a1.z, b1.z, acc[0]);
                    acc[0] = S::mfma(a1.w, b1.w, acc[0]);
                }
            }
            epilogue();
            ++cur_tile;
        }
        sweep_wait_vm<0>();                                             // the over-issued tail chunks must land before LDS is released
    }

    // ---- final: sort + cut every list to kp, write the block's partial result ----
    __syncthreads();
    for (int q = wave; q < TQ; q += kWaves) {
        prune_dispatch(list_d + q * prm.cap, list_i + q * prm.cap, &cnt[q], &tau_s[q], prm.cap,
                       prm.kp, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int n = cnt[q];
        const size_t base = ((size_t)q * prm.G + g) * prm.kp;
        for (int e = lane; e < prm.kp; e += 64) {
            prm.part_d[base + e] = e < n ? list_d[q * prm.cap + e] : INFINITY;
            prm.part_i[base + e] = e < n ? list_i[q * prm.cap + e] : -1;
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) wave_maxnorm = fmaxf(wave_maxnorm, __shfl_xor(wave_maxnorm, o));
    if (lane == 0) wmax_s[wave] = wave_maxnorm;
    __syncthreads();
    if (tid == 0) {
        float m = 0.f;
        for (int w = 0; w < kWaves; ++w) m = fmaxf(m, wmax_s[w]);
        prm.part_maxnorm[v] = m;
    }
}

// --------------------------------------------------------------------------------------
// knn_plane_sweep: the HBM-bound sweep over the PREPARED store's fp16 plane for 1 .. 64 resident queries.
//   The fp32 sweeps above stream 4 B per store element; a prepared store (ac_knn_prepare_store) also holds every row as ONE
//   fp16 plane (2 B per element) whose one-product MFMA dot differs from the exact value by a bound E the merge's certificate
//   knows (knn_batch.hip: |v - exact| <= gamma (max|p| + |q|)^2) -- the machinery the batched search uses from 64 queries up.
//   This kernel is its small-batch, bandwidth-bound form: ONE pass over the plane (half the bytes of the fp32 sweep), the query
//   tile of 32 or 64 columns resident in LDS, v_mfma_f32_32x32x16_f16 at ~20 % of the matrix pipe, the candidate lists, pruning
//   and partial results of knn_sweep (exact results come from knn_merge_rerank's fp64 re-rank + certificate, as everywhere).
//   * The plane is tile-major: a 256-row tile is one contiguous run of 256 * Kp * 2 bytes, [k-slot][row % 256][8 fp16] inside.
//     In that layout the MFMA A fragment of (32 rows, 16 k) is two 512-byte runs -- lane (i = lane & 31, kg = lane >> 5) loads
//     the 16 bytes of row i, k-slot 2 s + kg -- i.e. every load instruction consumes whole 128-byte lines, which is what lets
//     it be NON-TEMPORAL straight into registers (knn_sweep's fp32 operand layout takes half a line per instruction and loses
//     with nt loads; knn_sweep_ring goes through LDS for that reason).  No LDS round trip for the rows, no barrier in the k-loop.
//   * A workgroup = 8 waves x 32 rows = one 256-row tile per step, tiles interleaved over the grid (T = it * G + g), one
//     residency round.  Per wave 16 loads (16 KB) in flight: four register buffers of four k-steps, issued unconditionally.
//   * Query B fragments: LDS [k-step][sub-tile][lane] x 16 B, copied from the query plane knn_prepare_queries builds (same
//     scaling and rounding as the batched path, so the same bound).  Sweep value v = |p|^2 + f_q acc; |p|^2 comes from the
//     prepared norms through SCALAR loads (constant address space: off the vmcnt queue the prefetched rows sit in).
//   * The block meets once per tile at a raw s_barrier to decide about pruning (as knn_sweep_ring does).
// --------------------------------------------------------------------------------------
struct PlaneSweepParams {
    const uint16_t* Pp;       // store plane, tile-major (knn_batch.hip knn_plane_kernel)
    const float* pnorm;       // [round_up(N, 256)] |p|^2, +inf past N
    const uint16_t* Qp;       // query plane [Kp/8][q_rows][8]
    int64_t q_rows;
    const float* qfac;        // [q_rows] -2 2^(e_p + e_q)
    int64_t N;
    int64_t ntiles;           // ceil(N / 256)
    int Kp;                   // multiple of 64
    int q0, nq;               // this launch's query tile: queries q0 .. q0 + nq - 1 (nq <= TQ)
    int kp, cap, G;
    float* part_d;            // [query][G][kp]
    int32_t* part_i;
    int32_t* clear_ctr;       // as SweepParams (first tile's launch only; else NULL)
    int32_t* clear_stats;
};

typedef _Float16 kf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t ku32x4 __attribute__((ext_vector_type(4)));
// C/D layout of v_mfma_f32_32x32x16_f16: lane owns column (lane & 31) and these 16 rows
__device__ __forceinline__ int plane_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int TQ>
__global__ __launch_bounds__(kThreads, 2) void knn_plane_sweep(PlaneSweepParams prm) {
    constexpr int NJ = TQ / 32;                       // 32-column sub-tiles of the query tile
    constexpr int NB = 4, GK = 4;                     // register buffers x k-steps per buffer (16 loads in flight)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, kg = lane >> 5;
    const int g = xcd_remap(blockIdx.x, prm.G);
    if (blockIdx.x == 0 && tid < 64 && prm.clear_ctr) {            // consumed by the kernels launched after this one
        prm.clear_ctr[tid] = 0;
        if (tid < 4 && prm.clear_stats) prm.clear_stats[tid] = tid == 1 ? 2 : 0;     // [1] = 2: the fp16-plane sweep ran
    }
    const int nk = prm.Kp >> 4;                       // 16-k MFMA steps per row
    const int ngrp = nk / GK;                         // (Kp % 64 == 0)
    // ---- LDS: query fragments [nk][NJ][64] x 16 B | lists | counters ----
    ku32x4* Qs = reinterpret_cast<ku32x4*>(smem);
    float* list_d = reinterpret_cast<float*>(smem + (size_t)nk * NJ * 1024);
    int32_t* list_i = reinterpret_cast<int32_t*>(list_d + (size_t)TQ * prm.cap);
    int* cnt = reinterpret_cast<int*>(list_i + (size_t)TQ * prm.cap);
    float* tau_s = reinterpret_cast<float*>(cnt + TQ);
    __shared__ int need_s[2 * 8];                                  // [2][8] per-tile verdicts of the waves (static LDS: see knn_sweep_ring)
    for (int t = tid; t < nk * NJ * 64; t += kThreads) {
        const int l = t & 63, jj = (t >> 6) % NJ, s = (t >> 6) / NJ;
        const int64_t qrow = prm.q0 + 32 * jj + (l & 31);          // (< q_rows: the plane is padded to 256 queries with zeros)
        Qs[t] = *reinterpret_cast<const ku32x4*>(prm.Qp + ((int64_t)(2 * s + (l >> 5)) * prm.q_rows + qrow) * 8);
    }
    for (int t = tid; t < TQ; t += kThreads) {
        cnt[t] = 0;
        tau_s[t] = (t < prm.nq) ? INFINITY : -INFINITY;
    }
    __syncthreads();
    float tau[NJ], qf[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) { tau[jj] = tau_s[32 * jj + i32]; qf[jj] = prm.qfac[prm.q0 + 32 * jj + i32]; }
    int bar_parity = 0;

    const int64_t my_tiles = (prm.ntiles > g) ? (prm.ntiles - 1 - g) / prm.G + 1 : 0;
    const int64_t total = my_tiles * ngrp;            // groups of GK k-steps this wave consumes
    const size_t tile_bytes = (size_t)prm.Kp * 512;   // 256 rows x Kp x 2 B
    const uint32_t lane_off = (uint32_t)(kg * 256 + 32 * wave + i32) * 16u;    // byte offset of this lane's piece inside a k-slot pair
    // prefetch state: next group to load
    int64_t pf_tile = 0;
    int pf_grp = 0;
    auto tile_base = [&](int64_t it) -> const char* {
        int64_t T = it * prm.G + g;
        if (T > prm.ntiles - 1) T = prm.ntiles - 1;    // past the end: the last tile again (never consumed)
        return reinterpret_cast<const char*>(prm.Pp) + (size_t)T * tile_bytes;
    };
    const char* pf_base = tile_base(0);
    ku32x4 buf[NB][GK];
#define AC_PL_PREFETCH(B)                                                                                     \
    do {                                                                                                      \
        const char* src_ = pf_base + (size_t)pf_grp * (GK * 8192) + lane_off;                                 \
        _Pragma("unroll") for (int u = 0; u < GK; ++u)                                                        \
            buf[B][u] = __builtin_nontemporal_load(reinterpret_cast<const ku32x4*>(src_ + u * 8192));          \
        if (++pf_grp == ngrp) { pf_grp = 0; ++pf_tile; pf_base = tile_base(pf_tile); }                        \
    } while (0)

    f32x16 acc[NJ];
    int64_t cur_tile = 0;
    int cur_grp = 0;

    auto epilogue = [&]() {
        const int row0 = (int)((cur_tile * prm.G + g) * 256) + 32 * wave;      // wave-uniform, multiple of 32 (N < 2^31)
        // |p|^2 of the wave's 32 rows through the scalar unit: lane half kg needs rows (r & 3) + 8 (r >> 2) + 4 kg
        typedef const float __attribute__((address_space(4)))* cfp;
        const cfp pn = (cfp)(uintptr_t)(prm.pnorm + __builtin_amdgcn_readfirstlane(row0));
        bool maybe = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float lo = pn[(r & 3) + 8 * (r >> 2)], hi = pn[(r & 3) + 8 * (r >> 2) + 4];
            const float pnr = kg ? hi : lo;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                acc[jj][r] = fmaf(acc[jj][r], qf[jj], pnr);                    // the sweep value, in place
                maybe |= acc[jj][r] < tau[jj];
            }
        }
        unsigned done = 0;
        bool pend = __any(maybe) != 0;
        for (;;) {
            bool lane_pend = false;
            if (pend) {
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int q = 32 * jj + i32;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = acc[jj][r];
                        const unsigned bit = 1u << (16 * jj + r);
                        if (!(done & bit) && d < tau[jj]) {                    // (rows past N carry +inf norms: never below tau)
                            const int slot = atomicAdd(&cnt[q], 1);
                            if (slot < prm.cap) {
                                list_d[q * prm.cap + slot] = d;
                                list_i[q * prm.cap + slot] = row0 + plane_acc_row(r, lane);
                                done |= bit;
                            } else {
                                lane_pend = true;
                            }
                        }
                    }
                }
            }
            // one RAW barrier per tile (knn_sweep_ring: __syncthreads_or would fence vmcnt(0) and drain the prefetched rows)
            int over = 0;
            if (tid < TQ) over = cnt[tid] > (prm.cap - prm.cap / 4);
            const int w_need = __any((int)lane_pend | over) != 0;
            const int fl = 8 * (bar_parity & 1);
            if (lane == 0) __hip_atomic_store(&need_s[fl + wave], w_need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int need = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) need |= __hip_atomic_load(&need_s[fl + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++bar_parity;
            if (!need) break;
            for (int q = wave; q < TQ; q += kWaves) {
                if (cnt[q] > prm.kp)
                    prune_dispatch(list_d + q * prm.cap, list_i + q * prm.cap, &cnt[q], &tau_s[q], prm.cap, prm.kp, lane);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) tau[jj] = tau_s[32 * jj + i32];
            pend = true;                                  // re-test un-pushed entries against the tightened tau
        }
    };

#define AC_PL_COMPUTE(B)                                                                                      \
    do {                                                                                                      \
        if (cur_grp == 0) {                                                                                   \
            _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj)                                                 \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;                              \
        }                                                                                                     \
        const ku32x4* qsrc_ = Qs + (size_t)cur_grp * (GK * NJ * 64) + lane;                                    \
        ku32x4 bq_[NJ];                                                                                        \
        _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) bq_[jj] = qsrc_[jj * 64];                           \
        _Pragma("unroll") for (int u = 0; u < GK; ++u) {                                                      \
            const kf16x8 a_ = __builtin_bit_cast(kf16x8, buf[B][u]);                                          \
            kf16x8 b_[NJ];                                                                                    \
            _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) b_[jj] = __builtin_bit_cast(kf16x8, bq_[jj]);   \
            if (u + 1 < GK) { /* LDS reads one step ahead */                                                  \
                _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) bq_[jj] = qsrc_[((u + 1) * NJ + jj) * 64];  \
            }                                                                                                 \
            _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj)                                                 \
                acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_[jj], acc[jj], 0, 0, 0);               \
            __builtin_amdgcn_sched_barrier(0); /* keep the per-load consume order */                          \
        }                                                                                                     \
        if (++cur_grp == ngrp) { epilogue(); cur_grp = 0; ++cur_tile; }                                       \
    } while (0)

    // Loads are issued unconditionally (past the end they re-read the last tile) so that every path has the same number of
    // loads in flight: a load inside a branch makes hipcc's s_waitcnt accounting wait on the buffer it has just issued.
    if (total > 0) {
        AC_PL_PREFETCH(0);
        AC_PL_PREFETCH(1);
        AC_PL_PREFETCH(2);
        for (int64_t gg = 0; gg < total; gg += NB) {      // (ngrp may be odd: the tile boundary falls anywhere in this body)
            AC_PL_PREFETCH(3);
            AC_PL_COMPUTE(0);
            AC_PL_PREFETCH(0);
            if (gg + 1 < total) AC_PL_COMPUTE(1);
            AC_PL_PREFETCH(1);
            if (gg + 2 < total) AC_PL_COMPUTE(2);
            AC_PL_PREFETCH(2);
            if (gg + 3 < total) AC_PL_COMPUTE(3);
        }
    }
#undef AC_PL_PREFETCH
#undef AC_PL_COMPUTE

    // ---- final: sort + cut every list to kp, write the block's partial result ----
    __syncthreads();
    for (int q = wave; q < TQ; q += kWaves) {
        if (q >= prm.nq) continue;
        prune_dispatch(list_d + q * prm.cap, list_i + q * prm.cap, &cnt[q], &tau_s[q], prm.cap, prm.kp, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int n = cnt[q];
        const size_t base = ((size_t)(prm.q0 + q) * prm.G + g) * prm.kp;
        for (int e = lane; e < prm.kp; e += 64) {
            prm.part_d[base + e] = e < n ? list_d[q * prm.cap + e] : INFINITY;
            prm.part_i[base + e] = e < n ? list_i[q * prm.cap + e] : -1;
        }
    }
}

// --------------------------------------------------------------------------------------
// merge + exact re-rank + certificate.  One block (256 threads) per query.
// --------------------------------------------------------------------------------------
const float* zeros_device() {                 // (the symbol has one address per device)
    static const float* cache[64] = {nullptr};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!cache[dev]) { void* q = nullptr; (void)hipGetSymbolAddress(&q, HIP_SYMBOL(g_knn_zeros)); cache[dev] = (const float*)q; }
    return cache[dev];
}

struct MergeParams {
    const float* P;
    int64_t N;
    int64_t ldP;
    const float* Q;
    int64_t ldQ;
    int D;
    int Dp;
    int k;
    int kp;
    int G;
    int nblk;       // entries of part_maxnorm
    double gamma;   // |sweep value - exact| <= gamma (|p| + |q|)^2: n * 2^-24 for the fp32 fma chain (n roundings per term);
                    // the fp16 GEMM-form sweep uses its own bound (knn_batch_gamma, knn_batch.hip)
    // candidate-buffer mode (knn_batch.hip): one list of up to cand_cap entries per query instead of G lists of kp;
    // the list is cand_segs segments of cand_cap / cand_segs entries; cand_cnt[q * cand_segs + s] = entries offered to segment s
    // (may exceed the segment: overflow -> exact fallback)
    const int32_t* cand_cnt;
    int cand_cap, cand_segs;
    int32_t* cand_cnt_clear;   // (threshold stages) = cand_cnt: this query's counters are zeroed once read, for the next sweep's appends
    float* thr_out;            // (threshold stages) thr_out[q] = min(thr_out[q], tau_q - |q|^2 + E rounded up), tau_q = the k-th (= k'-th)
                               // exact distance of this stage -- what knn_thr_kernel computed in a launch of its own (round 3)
    int64_t run_stride;      // 0, or 8 * stride of a threshold stage's sample: candidate id i is store row (i >> 3) * run_stride + (i & 7)
    int64_t row_offset;
    const float* part_d;
    const int32_t* part_i;
    const float* part_maxnorm;
    float* outD;
    double* outD64;   // optional: the exact fp64 distances next to their fp32 roundings (shard merges order by these)
    int64_t* outI;
    int32_t* flags;   // [nq] 0 = certified; slot + 1 = exact fallback over fb_S row slabs; -1 = fallback, no slot
    int32_t* stats;   // optional
    // slab-parallel exact fallback: fb_F slots of fb_S slabs x k (exact distance, id) partial results
    int fb_S, fb_F;
    double* fb_d;
    int32_t* fb_i;
    int32_t* fb_slotctr;
    // (threshold stages of the batch path, round 6) 1 = stop after the selection: thr_out[q] = the k'-th smallest SWEEP value of this
    // stage's candidates, one ulp up (the sweep keeps v < thr).  Those candidates are k' real rows, so at least k' rows of the whole
    // store pass the next sweep -- all the final merge's certificate asks of a threshold ("nreal >= k'"; a short list sends the
    // query to the exact fallback).  No row is gathered, nothing is re-ranked: the stage's merge drops from 26 us to its loads +
    // four radix rounds, and the bound is tighter than tau - |q|^2 + E (no error term: both sides are sweep values).
    int thr_only = 0;
};

constexpr int kMergeThreads = 256;

// Block-wide radix select (8 bits per round) over 32-bit keys held in LDS: returns the `want`-th
// smallest (1-based) among the entries with active(t) != 0.  On return *rank_in_ties is how many of
// the entries equal to the result are needed to reach `want`, *n_ties how many such entries exist.
template <typename KeyFn, typename ActiveFn>
__device__ __forceinline__ uint32_t block_radix_select(int n, int want, KeyFn key_of, ActiveFn active,
                                                       int* hist, int* bcast, int* rank_in_ties, int* n_ties) {
    const int tid = threadIdx.x;
    __shared__ int wave_tot[kMergeThreads / 64];
    uint32_t prefix = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;                       // kMergeThreads == 256 bins
        __syncthreads();
        const uint32_t himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
        // distances of one query's candidates share their leading bits, so in the first rounds nearly every key lands in
        // the same bin: run-length aggregation per thread (one LDS atomic per run instead of one per key)
        int run_bin = -1, run_cnt = 0;
        for (int t = tid; t < n; t += kMergeThreads) {
            if (!active(t)) continue;
            const uint32_t key = key_of(t);
            if ((key & himask) != (prefix & himask)) continue;
            const int bin = (int)((key >> shift) & 255u);
            if (bin == run_bin) { ++run_cnt; continue; }
            if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
            run_bin = bin; run_cnt = 1;
        }
        if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
        __syncthreads();
        // exclusive prefix of bin `tid`: wave scan + the totals of the waves below (kMergeThreads == 256 = 4 waves)
        const int mine_cnt = hist[tid];
        int c = mine_cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(c, o); if ((tid & 63) >= o) c += v; }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = c;
        __syncthreads();
        for (int w = 0; w < (tid >> 6); ++w) c += wave_tot[w];
        c -= mine_cnt;
        const int mine = hist[tid];
        if (c < want && want <= c + mine) { bcast[0] = tid; bcast[1] = want - c; bcast[2] = mine; }
        __syncthreads();
        prefix |= (uint32_t)bcast[0] << shift;
        want = bcast[1];
        *n_ties = bcast[2];
        __syncthreads();
    }
    *rank_in_ties = want;
    return prefix;
}

__global__ __launch_bounds__(kMergeThreads) void knn_merge_rerank(MergeParams prm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n = prm.cand_cnt ? prm.cand_cap : prm.G * prm.kp;
    const int kp = prm.kp;
    // candidate mode (knn_batch.hip): the list is cand_segs segments of segcap entries with a count each; the entries in use
    // are compacted into LDS (keys + ids), so the selection below walks the ~k' * stride real candidates, not the capacity
    const int segs = prm.cand_cnt ? prm.cand_segs : 1, segcap = n / segs;
    const bool cand = segs > 1;                            // (one segment: the list is walked in place, like the per-block lists)

    // LDS: keys[n] u32 | (cand) cid[n] i32 | qrow[Dp] f32 | sel[kp] u64 | exact[kp] f64 | hist[256] | misc
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);
    size_t off = ac::align_up((size_t)n * 4, 16);
    int32_t* cid = reinterpret_cast<int32_t*>(smem + off);
    if (cand) off += ac::align_up((size_t)n * 4, 16);
    float* qrow = reinterpret_cast<float*>(smem + off);
    off += ac::align_up((size_t)prm.Dp * 4, 16);
    unsigned long long* sel = reinterpret_cast<unsigned long long*>(smem + off);
    off += (size_t)kp * 8;
    double* exact = reinterpret_cast<double*>(smem + off);
    off += (size_t)kp * 8;
    int* hist = reinterpret_cast<int*>(smem + off);
    off += 256 * 4;
    int* misc = reinterpret_cast<int*>(smem + off);       // [0..2] radix broadcast, [4] nsel counter, [5] nreal, [6] segment overflow
    double* dmisc = reinterpret_cast<double*>(misc + 8);  // [0] qnorm2, [1] exact k-th

    const float* pd = prm.part_d + (size_t)q * n;
    const int32_t* pi = prm.part_i + (size_t)q * n;
    if (tid < 8) misc[tid] = 0;
    for (int c = tid; c < prm.Dp; c += kMergeThreads)
        qrow[c] = c < prm.D ? prm.Q[(size_t)q * prm.ldQ + c] : 0.f;
    int nkeys = n;                                         // key slots the selection walks
    if (cand) {
        // segment s holds min(count, segcap) entries; exclusive offsets by a block scan (segs <= 256 = one per thread)
        __shared__ int seg_off[kMergeThreads + 1];
        __shared__ int scan_tot[kMergeThreads / 64];
        const int raw = tid < segs ? prm.cand_cnt[(size_t)q * segs + tid] : 0;
        const int mine = raw < segcap ? raw : segcap;
        int c = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(c, o); if (lane >= o) c += v; }
        if (lane == 63) scan_tot[wave] = c;
        __syncthreads();
        if (raw > segcap) misc[6] = 1;                     // (benign race: every writer stores 1)
        for (int w = 0; w < wave; ++w) c += scan_tot[w];
        seg_off[tid + 1] = c;
        if (tid == 0) seg_off[0] = 0;
        __syncthreads();
        nkeys = seg_off[kMergeThreads];
        for (int j = tid; j < nkeys; j += kMergeThreads) {
            int lo = 0, hi = segs;                         // largest s with seg_off[s] <= j
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= j) lo = mid; else hi = mid; }
            const int t = lo * segcap + (j - seg_off[lo]);
            keys[j] = fkey(pd[t]);
            cid[j] = pi[t];
        }
        if (tid == 0) misc[5] = nkeys;
        if (prm.cand_cnt_clear && tid < segs) prm.cand_cnt_clear[(size_t)q * segs + tid] = 0;       // (read above, before the barriers)
    } else {
        // monotone 32-bit key of the sweep value; padding (id < 0) and slots past the count sort last
        const int raw = prm.cand_cnt ? prm.cand_cnt[q] : n;
        const int nfill = raw < n ? raw : n;
        if (tid == 0 && raw > n) misc[6] = 1;
        int nreal_local = 0;
#pragma unroll 8
        for (int t = tid; t < n; t += kMergeThreads) {      // (unrolled: the loads of several candidates in flight)
            const bool real = t < nfill && pi[t] >= 0;
            keys[t] = real ? fkey(pd[t]) : 0xffffffffu;
            nreal_local += real ? 1 : 0;
        }
        __syncthreads();
        atomicAdd(&misc[5], nreal_local);
        if (prm.cand_cnt_clear && tid == 0) prm.cand_cnt_clear[q] = 0;
    }
    __syncthreads();
    const bool overflow = misc[6] != 0;
    const int nreal = misc[5];
    const int nsel = nreal < kp ? nreal : kp;     // how many candidates we re-rank
    auto id_of = [&](int t) -> int32_t { return cand ? cid[t] : pi[t]; };

    // ---- the nsel-th smallest sweep value T; ties at T are resolved by the lowest ids ----
    uint32_t T = 0xffffffffu;
    int32_t tie_id_max = 0x7fffffff;
    if (nsel > 0) {
        int r = 0, c_eq = 0;
        T = block_radix_select(nkeys, nsel, [&](int t) { return keys[t]; },
                               [&](int t) { return keys[t] != 0xffffffffu; }, hist, misc, &r, &c_eq);
        if (c_eq != r) {     // rare: several candidates share the boundary value -> r lowest ids of them
            int r2 = 0, c2 = 0;
            tie_id_max = (int32_t)block_radix_select(
                nkeys, r, [&](int t) { return (uint32_t)id_of(t); },
                [&](int t) { return keys[t] != 0xffffffffu && keys[t] == T; }, hist, misc, &r2, &c2);
        }
    }
    if (prm.thr_only) {
        if (tid == 0 && prm.thr_out && nreal >= kp && nsel > 0) {
            const float t = nextafterf(fkey_inv(T), INFINITY);
            if (t < prm.thr_out[q]) prm.thr_out[q] = t;
        }
        return;
    }
    // ---- compact the selected candidates ----
    for (int t = tid; t < nkeys; t += kMergeThreads) {
        const uint32_t key0 = keys[t];
        const int32_t id = key0 != 0xffffffffu ? id_of(t) : -1;
        if (nsel > 0 && id >= 0) {
            const uint32_t key = key0;
            if (key < T || (key == T && id <= tie_id_max)) {
                const int s = atomicAdd(&misc[4], 1);
                if (s < kp) sel[s] = ((unsigned long long)key << 32) | (uint32_t)id;
            }
        }
    }
    __syncthreads();
    const int ns = misc[4] < kp ? misc[4] : kp;
    const unsigned long long T64 = (unsigned long long)T << 32;

    // ---- exact fp64 distances of the selected rows; |q|^2 ----
    // Four rows per wave at a time: their loads are issued together, so a wave pays the (random-row, HBM) latency once per
    // group instead of once per row -- the per-row arithmetic (four accumulators over c4 = lane, lane + 64, ..., the
    // (a0 + a1) + (a2 + a3) fold, the xor-shuffle tree) is unchanged, hence the same bits.  (One row at a time this loop was
    // most of the kernel: ~20 of its 27 - 32 us.)
    const int nc4 = prm.Dp >> 2;
    constexpr int RU = 4;
    for (int s0 = wave * RU; s0 < ns; s0 += (kMergeThreads / 64) * RU) {
        const f32x4* prow[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int s = s0 + u < ns ? s0 + u : s0;                  // (a short last group re-reads its first row)
            const int32_t id = (int32_t)(uint32_t)(sel[s] & 0xffffffffull);
            const int64_t prow_i = prm.run_stride ? (int64_t)(id >> 3) * prm.run_stride + (id & 7) : (int64_t)id;     // (threshold stages: sample row -> store row)
            prow[u] = reinterpret_cast<const f32x4*>(prm.P + (size_t)prow_i * prm.ldP);
        }
        double acc[RU][4];
#pragma unroll
        for (int u = 0; u < RU; ++u) { acc[u][0] = 0; acc[u][1] = 0; acc[u][2] = 0; acc[u][3] = 0; }
#pragma unroll 2
        for (int c4 = lane; c4 < nc4; c4 += 64) {
            f32x4 p[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) p[u] = prow[u][c4];
            const f32x4 qq = *reinterpret_cast<const f32x4*>(qrow + 4 * c4);
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const double e0 = (double)p[u].x - (double)qq.x, e1 = (double)p[u].y - (double)qq.y;
                const double e2 = (double)p[u].z - (double)qq.z, e3 = (double)p[u].w - (double)qq.w;
                acc[u][0] = fma(e0, e0, acc[u][0]); acc[u][1] = fma(e1, e1, acc[u][1]);
                acc[u][2] = fma(e2, e2, acc[u][2]); acc[u][3] = fma(e3, e3, acc[u][3]);
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            double a = (acc[u][0] + acc[u][1]) + (acc[u][2] + acc[u][3]);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o);
            if (lane == 0 && s0 + u < ns) exact[s0 + u] = a;
        }
    }
    if (wave == 0) {
        double a = 0;
        for (int c = lane; c < prm.Dp; c += 64) a = fma((double)qrow[c], (double)qrow[c], a);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o);
        if (lane == 0) { dmisc[0] = a; dmisc[1] = INFINITY; }
    }
    __syncthreads();

    // ---- final order by (exact, id); emit top-k ----
    const int kout = prm.k;
    for (int t = tid; t < ns; t += kMergeThreads) {
        const double dt = exact[t];
        const uint32_t it = (uint32_t)(sel[t] & 0xffffffffull);
        int rank = 0;
        for (int s = 0; s < ns; ++s) {
            const double ds = exact[s];
            const uint32_t is = (uint32_t)(sel[s] & 0xffffffffull);
            rank += (ds < dt || (ds == dt && is < it)) ? 1 : 0;
        }
        if (rank < kout) {
            prm.outD[(size_t)q * kout + rank] = (float)dt;
            if (prm.outD64) prm.outD64[(size_t)q * kout + rank] = dt;
            prm.outI[(size_t)q * kout + rank] = (int64_t)it + prm.row_offset;
        }
        if (rank == kout - 1) dmisc[1] = dt;
    }
    for (int t = ns + tid; t < kout; t += kMergeThreads) {   // k > N: faiss-style padding
        prm.outD[(size_t)q * kout + t] = FLT_MAX;
        if (prm.outD64) prm.outD64[(size_t)q * kout + t] = INFINITY;
        prm.outI[(size_t)q * kout + t] = -1;
    }
    __syncthreads();

    // ---- certificate ----
    // largest row norm seen by the sweep: the per-block maxima reduced by the whole workgroup (a one-thread loop over up to
    // 512 global loads was most of this kernel's time for a single query)
    float mx_all = 0.f;
    for (int b = tid; b < prm.nblk; b += kMergeThreads) mx_all = fmaxf(mx_all, prm.part_maxnorm[b]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx_all = fmaxf(mx_all, __shfl_xor(mx_all, o));
    if (lane == 0) hist[wave] = (int)__float_as_uint(mx_all);           // (norms are non-negative: their bits order like the values)
    __syncthreads();
    if (tid == 0 && prm.thr_out) {
        // threshold stage: the k-th (= k'-th) exact distance of this sample bounds the store's k'-th smallest distance
        float mx = 0.f;
        for (int w = 0; w < kMergeThreads / 64; ++w) mx = fmaxf(mx, __uint_as_float((uint32_t)hist[w]));
        const double qn2 = dmisc[0], tau = dmisc[1];
        const double pn = sqrt((double)mx * 1.001), qn = sqrt(qn2);
        const double E = prm.gamma * (pn + qn) * (pn + qn) + 1e-30;
        float t = (float)(tau - qn2 + E);
        if ((double)t < tau - qn2 + E) t = nextafterf(t, INFINITY);
        // a stage that kept fewer than k' rows for this query has no k'-th distance to offer: the threshold of the stage before it
        // (still a valid bound) stays; a valid new bound only ever tightens it
        if (isfinite(tau) && ns >= kout && t < prm.thr_out[q]) prm.thr_out[q] = t;
    }
    if (tid == 0) {
        int ok = 1;
        if (prm.N > (int64_t)kp) {
            float mx = 0.f;
            for (int w = 0; w < kMergeThreads / 64; ++w) mx = fmaxf(mx, __uint_as_float((uint32_t)hist[w]));
            const double qn2 = dmisc[0];
            const double pn = sqrt((double)mx * 1.001), qn = sqrt(qn2);
            // fp32 fma-chain roundoff of |p|^2 - 2 q.p: every term passes through at most
            // nterms roundings, so |err| <= gamma_n * (|p|^2 + 2 sum|q_i p_i|) <= gamma_n (|p|+|q|)^2
            const double E = prm.gamma * (pn + qn) * (pn + qn) + 1e-30;
            const double a_last = (double)fkey_inv((uint32_t)(T64 >> 32));
            // every row that was NOT re-ranked has sweep value >= a_last, hence exact
            // distance >= a_last - E + |q|^2.  The k-th re-ranked must beat that strictly.
            const double kth = dmisc[1];
            ok = (ns >= kout) && (kth < a_last - E + qn2) && !overflow;
            if (prm.cand_cnt && nreal < kp) ok = 0;      // fewer than k' candidates kept: the "unseen rows >= a_last" premise is gone
        }
        int flag = 0;
        if (!ok) {
            const int slot = atomicAdd(prm.fb_slotctr, 1);
            flag = slot < prm.fb_F ? slot + 1 : -1;
            if (prm.stats) atomicAdd(&prm.stats[0], 1);
        }
        prm.flags[q] = flag;
    }
}

// --------------------------------------------------------------------------------------
// exact fallback: fp64 sweep for the (rare) queries whose certificate failed.
// One block per query; exits immediately unless flagged.
// --------------------------------------------------------------------------------------
constexpr int kFbThreads = 512;
constexpr int kFbWaves = kFbThreads / 64;
constexpr int kFbCap = 1024;          // list capacity (k <= 248 -> prune keeps k)
constexpr int kFbRound = 32;          // rows per wave between barriers

// grid = (fb_S, nq): block (s, q) scans row slab s of a flagged query and writes its exact top-k to the
// query's slot; knn_exact_fb_merge then merges the slabs.  A flagged query without a slot (more than fb_F
// failures in one call) is handled by its s == 0 block alone over the whole store.
// (round 6: the grid is (fb_S, min(nq, kFbQueryGroups)) and a block walks the queries q = blockIdx.y, + gridDim.y, ...: with
//  no query flagged -- every call of an ordinary batch -- dispatching fb_S x nq = 16 384 empty 512-thread blocks cost 8.5 us;
//  fb_S x 8 cost 2.  A device holds <= ~1000 of these blocks at once, so flagged batches lose nothing.)
constexpr int kFbQueryGroups = 8, kFbMergeGroups = 32;
__device__ __forceinline__ void knn_exact_fallback_query(const MergeParams& prm, const int q, const int slab, char* smem) {
    const int flag = prm.flags[q];
    if (flag == 0 || (flag < 0 && slab != 0)) return;
    const bool direct = flag < 0;
    const int64_t row_lo = direct ? 0 : (prm.N * slab) / prm.fb_S;
    const int64_t row_hi = direct ? prm.N : (prm.N * (slab + 1)) / prm.fb_S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* ld = reinterpret_cast<double*>(smem);                  // [kFbCap]
    int32_t* li = reinterpret_cast<int32_t*>(ld + kFbCap);         // [kFbCap]
    float* qrow = reinterpret_cast<float*>(li + kFbCap);           // [Dp]
    int* misc = reinterpret_cast<int*>(qrow + ac::align_up((size_t)prm.Dp, 4));  // [0] cnt
    double* tau_d = reinterpret_cast<double*>(misc + 4);
    int32_t* tau_i = reinterpret_cast<int32_t*>(tau_d + 1);

    for (int c = tid; c < prm.Dp; c += kFbThreads)
        qrow[c] = c < prm.D ? prm.Q[(size_t)q * prm.ldQ + c] : 0.f;
    if (tid == 0) { misc[0] = 0; *tau_d = INFINITY; *tau_i = 0x7fffffff; }
    __syncthreads();
    const int nc4 = prm.Dp >> 2;
    const int k = prm.k;

    auto prune = [&]() {
        // block-wide rank-by-counting over cnt <= kFbCap entries (2 per thread)
        const int n = misc[0] < kFbCap ? misc[0] : kFbCap;
        double myd[2]; int32_t myi[2]; int rank[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int s = tid + kFbThreads * e;
            myd[e] = s < n ? ld[s] : INFINITY;
            myi[e] = s < n ? li[s] : 0x7fffffff;
            rank[e] = 0;
        }
        for (int s = 0; s < n; ++s) {
            const double d = ld[s]; const int32_t i = li[s];
#pragma unroll
            for (int e = 0; e < 2; ++e) rank[e] += (d < myd[e] || (d == myd[e] && i < myi[e])) ? 1 : 0;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int s = tid + kFbThreads * e;
            if (s < n && rank[e] < k) {
                ld[rank[e]] = myd[e]; li[rank[e]] = myi[e];
                if (rank[e] == k - 1) { *tau_d = myd[e]; *tau_i = myi[e]; }
            }
        }
        if (tid == 0) misc[0] = n < k ? n : k;
        __syncthreads();
    };

    for (int64_t base = row_lo; base < row_hi; base += (int64_t)kFbWaves * kFbRound) {
        const double td = *tau_d; const int32_t ti = *tau_i;
        for (int m = 0; m < kFbRound; ++m) {
            const int64_t row = base + (int64_t)m * kFbWaves + wave;
            if (row >= row_hi) break;
            const f32x4* prow = reinterpret_cast<const f32x4*>(prm.P + (size_t)row * prm.ldP);
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int c4 = lane; c4 < nc4; c4 += 64) {
                const f32x4 p = prow[c4];
                const f32x4 qq = *reinterpret_cast<const f32x4*>(qrow + 4 * c4);
                const double e0 = (double)p.x - (double)qq.x, e1 = (double)p.y - (double)qq.y;
                const double e2 = (double)p.z - (double)qq.z, e3 = (double)p.w - (double)qq.w;
                a0 = fma(e0, e0, a0); a1 = fma(e1, e1, a1); a2 = fma(e2, e2, a2); a3 = fma(e3, e3, a3);
            }
            double a = (a0 + a1) + (a2 + a3);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o);
            if (lane == 0 && (a < td || (a == td && (int32_t)row < ti))) {
                const int s = atomicAdd(&misc[0], 1);
                if (s < kFbCap) { ld[s] = a; li[s] = (int32_t)row; }
            }
        }
        __syncthreads();
        const int c_now = misc[0];      // read between two barriers: identical for every thread
        __syncthreads();
        if (c_now > kFbCap - kFbWaves * kFbRound) prune();
    }
    prune();
    const int n = misc[0];
    if (direct) {
        for (int t = tid; t < k; t += kFbThreads) {
            prm.outD[(size_t)q * k + t] = t < n ? (float)ld[t] : FLT_MAX;
            if (prm.outD64) prm.outD64[(size_t)q * k + t] = t < n ? ld[t] : (double)INFINITY;
            prm.outI[(size_t)q * k + t] = t < n ? (int64_t)li[t] + prm.row_offset : -1;
        }
    } else {
        const size_t base = ((size_t)(flag - 1) * prm.fb_S + slab) * k;
        for (int t = tid; t < k; t += kFbThreads) {
            prm.fb_d[base + t] = t < n ? ld[t] : INFINITY;
            prm.fb_i[base + t] = t < n ? li[t] : 0x7fffffff;
        }
    }
}
__global__ __launch_bounds__(kFbThreads) void knn_exact_fallback(MergeParams prm, int nq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int q = blockIdx.y; q < nq; q += gridDim.y) {
        knn_exact_fallback_query(prm, q, blockIdx.x, smem);
        __syncthreads();                                            // (the next query reuses the lists)
    }
}

// merge the fb_S slab results of a flagged query: bitonic sort of fb_S * k (<= 4096) exact entries
__device__ __forceinline__ void knn_exact_fb_merge_query(const MergeParams& prm, const int npow2, const int q, char* smem) {
    const int tid = threadIdx.x;
    const int flag = prm.flags[q];
    if (flag <= 0) return;
    double* ds = reinterpret_cast<double*>(smem);
    int32_t* is = reinterpret_cast<int32_t*>(ds + npow2);
    const int n = prm.fb_S * prm.k;
    const size_t base = (size_t)(flag - 1) * n;
    for (int t = tid; t < npow2; t += 256) {
        ds[t] = t < n ? prm.fb_d[base + t] : INFINITY;
        is[t] = t < n ? prm.fb_i[base + t] : 0x7fffffff;
    }
    __syncthreads();
    for (int size = 2; size <= npow2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (npow2 >> 1); t += 256) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const double dl = ds[lo], dh = ds[hi];
                const int32_t il = is[lo], ih = is[hi];
                const bool gt = dl > dh || (dl == dh && il > ih);
                if (gt == asc) { ds[lo] = dh; ds[hi] = dl; is[lo] = ih; is[hi] = il; }
            }
            __syncthreads();
        }
    for (int t = tid; t < prm.k; t += 256) {
        const bool real = is[t] != 0x7fffffff;
        prm.outD[(size_t)q * prm.k + t] = real ? (float)ds[t] : FLT_MAX;
        if (prm.outD64) prm.outD64[(size_t)q * prm.k + t] = real ? ds[t] : (double)INFINITY;
        prm.outI[(size_t)q * prm.k + t] = real ? (int64_t)is[t] + prm.row_offset : -1;
    }
}
__global__ __launch_bounds__(256) void knn_exact_fb_merge(MergeParams prm, int npow2, int nq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
        knn_exact_fb_merge_query(prm, npow2, q, smem);
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------
// shard merge and prototype scores
// --------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(256) void topk_merge_kernel(const DT* Din, const int64_t* Iin,
                                                         int shards, int nq, int k, float* outD,
                                                         int64_t* outI) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = shards * k;
    int64_t* ids = reinterpret_cast<int64_t*>(smem);
    DT* ds = reinterpret_cast<DT*>(ids + n);
    for (int t = tid; t < n; t += 256) {
        const int s = t / k, e = t - s * k;
        ids[t] = Iin[((size_t)s * nq + q) * k + e];
        ds[t] = Din[((size_t)s * nq + q) * k + e];
    }
    for (int t = tid; t < k; t += 256) { outD[(size_t)q * k + t] = FLT_MAX; outI[(size_t)q * k + t] = -1; }
    __syncthreads();
    for (int t = tid; t < n; t += 256) {
        const int64_t it = ids[t];
        if (it < 0) continue;
        const DT dt = ds[t];
        int rank = 0;
        for (int s = 0; s < n; ++s) {
            const int64_t is = ids[s];
            if (is < 0) continue;
            const DT d = ds[s];
            rank += (d < dt || (d == dt && (is < it || (is == it && s < t)))) ? 1 : 0;
        }
        if (rank < k) { outD[(size_t)q * k + rank] = (float)dt; outI[(size_t)q * k + rank] = it; }
    }
}

// memory.py:117 (exp(-d)) and :129-130 (softmax over the hits), one wave per query
__global__ __launch_bounds__(64) void proto_scores_kernel(const float* D, const int64_t* I, int nq,
                                                          int k, float* out) {
    const int q = blockIdx.x, lane = threadIdx.x;
    float mx = -INFINITY;
    for (int e = lane; e < k; e += 64)
        if (I[(size_t)q * k + e] >= 0) mx = fmaxf(mx, expf(-D[(size_t)q * k + e]));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int e = lane; e < k; e += 64)
        if (I[(size_t)q * k + e] >= 0) sum += expf(expf(-D[(size_t)q * k + e]) - mx);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
    for (int e = lane; e < k; e += 64) {
        const bool v = I[(size_t)q * k + e] >= 0;
        out[(size_t)q * k + e] = v ? expf(expf(-D[(size_t)q * k + e]) - mx) / sum : 0.f;
    }
}

// row ids of the hits -> class ids through the row->class map (index_to_label, memory.py:123,174;
// generalised int32 map of SURVEY 8a M6); padding (id < 0) and out-of-range ids give -1
__global__ __launch_bounds__(256) void rows_to_class_kernel(const int64_t* I, int64_t n, const int32_t* row_class,
                                                            int64_t nrows, const int64_t* class_lut, int nlut,
                                                            int64_t* out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int64_t id = I[t];
    int64_t c = -1;
    if (id >= 0 && id < nrows) {
        c = row_class ? (int64_t)row_class[id] : id;
        if (class_lut) c = (c >= 0 && c < nlut) ? class_lut[c] : -1;
    }
    out[t] = c;
}

// --------------------------------------------------------------------------------------
// small-store exact path: N <= kSmallN rows, ANY k <= N and ANY D.  The reference searches with
// k = #classes (classifier.py:424-425), so k can exceed the fused sweep's limit while N (= #classes,
// one prototype per class) stays tiny.  One block per query: fp64 distance of every row, full bitonic
// sort of (distance, id) in LDS, emit the first k.  Exact by construction (no certificate needed).
// --------------------------------------------------------------------------------------
constexpr int kSmallN = 8192;
constexpr int kSmallThreads = 256;

__global__ __launch_bounds__(kSmallThreads) void knn_small_exact(MergeParams prm, int npow2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ds = reinterpret_cast<double*>(smem);                // [npow2]
    int32_t* is = reinterpret_cast<int32_t*>(ds + npow2);        // [npow2]
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* qv = prm.Q + (size_t)q * prm.ldQ;
    for (int r = tid; r < npow2; r += kSmallThreads) {
        double a = INFINITY;
        if (r < prm.N) {
            const float* p = prm.P + (size_t)r * prm.ldP;
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            int c = 0;
            for (; c + 3 < prm.D; c += 4) {
                const double e0 = (double)p[c] - (double)qv[c], e1 = (double)p[c + 1] - (double)qv[c + 1];
                const double e2 = (double)p[c + 2] - (double)qv[c + 2], e3 = (double)p[c + 3] - (double)qv[c + 3];
                a0 = fma(e0, e0, a0); a1 = fma(e1, e1, a1); a2 = fma(e2, e2, a2); a3 = fma(e3, e3, a3);
            }
            for (; c < prm.D; ++c) { const double e = (double)p[c] - (double)qv[c]; a0 = fma(e, e, a0); }
            a = (a0 + a1) + (a2 + a3);
        }
        ds[r] = a;
        is[r] = r < prm.N ? r : 0x7fffffff;
    }
    __syncthreads();
    for (int size = 2; size <= npow2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (npow2 >> 1); t += kSmallThreads) {
                const int lo = 2 * t - (t & (stride - 1));       // index with bit `stride` cleared
                const int hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const double dl = ds[lo], dh = ds[hi];
                const int32_t il = is[lo], ih = is[hi];
                const bool gt = dl > dh || (dl == dh && il > ih);
                if (gt == asc) { ds[lo] = dh; ds[hi] = dl; is[lo] = ih; is[hi] = il; }
            }
            __syncthreads();
        }
    for (int t = tid; t < prm.k; t += kSmallThreads) {
        const bool real = t < prm.N;
        prm.outD[(size_t)q * prm.k + t] = real ? (float)ds[t] : FLT_MAX;
        if (prm.outD64) prm.outD64[(size_t)q * prm.k + t] = real ? ds[t] : (double)INFINITY;
        prm.outI[(size_t)q * prm.k + t] = real ? (int64_t)is[t] + prm.row_offset : -1;
    }
}

// ---- host-side planning ----
struct Plan {
    bool small;          // knn_small_exact instead of the fused sweep
    int small_pow2;
    int TQ, kp, cap, ng, Dp, G, nqt;
    int ring;            // 0, or the chunks per wave of knn_sweep_ring (<= 16 queries, D % 32 == 0, D <= 768)
    int64_t ntiles;
    size_t sweep_lds, merge_lds, fb_lds;
    size_t off_part_d, off_part_i, off_maxnorm, off_flags, off_zeros, off_fb_d, off_fb_i, off_fb_ctr, total;
    int fb_S, fb_F;
};

static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

static int make_plan(int64_t N, int D, int nq, int k, Plan* pl) {
    AC_REQUIRE(N >= 0 && N < 2147483647LL, AC_EINVAL, "knn: N=%lld out of range", (long long)N);
    AC_REQUIRE(D >= 1 && nq >= 0 && k >= 1, AC_EINVAL, "knn: bad D=%d nq=%d k=%d", D, nq, k);
    pl->small = false;
    pl->ring = 0;
    pl->Dp = (D + 3) / 4 * 4;
    {   // does the fused sweep cover (D, k)?  LDS: one 16-query tile + its candidate lists
        const int kp = k + kPad;
        int cap = next_pow2(2 * kp);
        if (cap < 64) cap = 64;
        const int ng = (pl->Dp + 16 * kGroup - 1) / (16 * kGroup);
        const size_t lds16 = (size_t)ng * kGroup * 64 * 16 + (size_t)16 * cap * 8 + 16 * 8 + kWaves * 4 + 64;
        if (k > AC_KNN_MAX_K || lds16 > (size_t)kLdsLimit) {
            AC_REQUIRE(N <= kSmallN, AC_EUNSUPPORTED,
                       "knn: k=%d, D=%d is outside the fused sweep (k <= %d, query tile + lists <= %d B of LDS) and "
                       "N=%lld exceeds the small-store path (N <= %d)", k, D, AC_KNN_MAX_K, kLdsLimit, (long long)N,
                       kSmallN);
            pl->small = true;
            pl->small_pow2 = next_pow2((int)(N > 2 ? N : 2));
            pl->TQ = 16; pl->kp = 0; pl->cap = 0; pl->ng = 0; pl->G = 1; pl->nqt = 1; pl->ntiles = 0;
            pl->sweep_lds = pl->merge_lds = pl->fb_lds = 0;
            pl->off_part_d = pl->off_part_i = pl->off_maxnorm = pl->off_flags = pl->off_zeros = 0;
            pl->off_fb_d = pl->off_fb_i = pl->off_fb_ctr = 0; pl->fb_S = pl->fb_F = 1;
            pl->total = 256;
            return AC_OK;
        }
    }
    pl->kp = k + kPad;
    pl->cap = next_pow2(2 * pl->kp);
    if (pl->cap < 64) pl->cap = 64;
    int TQ = nq > 16 ? 32 : 16;
    for (;;) {
        const int kcols = 16;
        pl->ng = (pl->Dp + kcols * kGroup - 1) / (kcols * kGroup);
        pl->sweep_lds = (size_t)(TQ / 16) * pl->ng * kGroup * 64 * 16 + (size_t)TQ * pl->cap * 8 + TQ * 8 +
                        kWaves * 4 + 64;
        if (pl->sweep_lds <= (size_t)kLdsLimit) break;
        AC_REQUIRE(TQ == 32, AC_EUNSUPPORTED,
                   "knn: D=%d with k=%d needs %zu B of LDS (> %d); unsupported", D, k, pl->sweep_lds, kLdsLimit);
        TQ = 16;
    }
    pl->TQ = TQ;
    pl->nqt = nq > 0 ? (nq + TQ - 1) / TQ : 1;
    const int rows_per_tile = kWaves * 16;
    pl->ntiles = (N + rows_per_tile - 1) / rows_per_tile;
    const ac::DevInfo& di = ac::dev_info();
    // blocks that are actually co-resident on a CU (VGPR/LDS limited); the grid is sized to exactly
    // one residency round so that no CU idles in a second, partial round
    // <= 16 queries, D a multiple of 32 up to 768: the LDS-ring form (rows by non-temporal DMA, queries in registers)
    static const int ring_env = getenv("AC_KNN_RING") ? atoi(getenv("AC_KNN_RING")) : -1;      // 0 = never (A/B)
    if (TQ == 16 && pl->nqt == 1 && (D % 32) == 0 && D <= 1024 && ring_env != 0) {
        const size_t lists = ring_qs_bytes(D <= 768 ? 24 : 32) + (size_t)TQ * pl->cap * 8 + TQ * 8 + kWaves * 4 + 64 + 64;
        // (tools/sweep_ring_bench.hip: 4 chunks per wave stream as fast as 8)
        pl->ring = (size_t)kWaves * 4 * 2048 + lists <= (size_t)kLdsLimit ? 4 : 0;
        if (pl->ring) pl->sweep_lds = (size_t)kWaves * pl->ring * 2048 + lists;
    }
    int per_cu = 0;
    hipError_t oe = pl->ring ? hipSuccess : (TQ == 32)
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, knn_sweep<2>, kThreads, pl->sweep_lds)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, knn_sweep<1>, kThreads, pl->sweep_lds);
    if (pl->ring) per_cu = 1;                        // (8 waves at the 256-register budget fill a CU)
    if (oe != hipSuccess) { (void)hipGetLastError(); per_cu = 1; }
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    int64_t G = ((int64_t)di.cus * per_cu) / pl->nqt;
    if (G < 1) G = 1;
    if (G > pl->ntiles) G = pl->ntiles;
    if (G > kMergeMaxCand / pl->kp) {
        G = kMergeMaxCand / pl->kp;
        if (pl->nqt == 1 && G > di.cus) G = G / di.cus * di.cus;
    }
    if (const char* e = getenv("AC_KNN_G")) { int64_t v = atoll(e); if (v >= 1) G = v; }   // tuning experiments
    if (G > pl->ntiles) G = pl->ntiles;
    if (G > kMergeMaxCand / pl->kp) G = kMergeMaxCand / pl->kp;
    if (G < 1) G = 1;
    pl->G = (int)G;
    const size_t nqpad = (size_t)pl->nqt * TQ;
    const size_t ncand = nqpad * pl->G * pl->kp;
    size_t off = 0;
    pl->off_part_d = off; off += ac::align_up(ncand * 4, 256);
    pl->off_part_i = off; off += ac::align_up(ncand * 4, 256);
    pl->off_maxnorm = off; off += ac::align_up((size_t)pl->G * pl->nqt * 4, 256);
    pl->off_flags = off; off += ac::align_up((size_t)(nq > 0 ? nq : 1) * 4, 256);
    pl->off_zeros = off; off += 256;
    // slab-parallel exact fallback: fb_S * k <= 4096 entries per slot, up to 64 slots
    pl->fb_S = 4096 / next_pow2(k);
    if (pl->fb_S > 64) pl->fb_S = 64;
    if (pl->fb_S < 1) pl->fb_S = 1;
    pl->fb_F = nq < 64 ? (nq > 0 ? nq : 1) : 64;
    const size_t fb_entries = (size_t)pl->fb_F * pl->fb_S * k;
    pl->off_fb_d = off; off += ac::align_up(fb_entries * 8, 256);
    pl->off_fb_i = off; off += ac::align_up(fb_entries * 4, 256);
    pl->off_fb_ctr = off; off += 256;
    pl->total = off;
    pl->merge_lds = ac::align_up((size_t)pl->G * pl->kp * 4, 16) + ac::align_up((size_t)pl->Dp * 4, 16) +
                    (size_t)pl->kp * 16 + 256 * 4 + 64;
    pl->fb_lds = (size_t)kFbCap * 12 + ac::align_up((size_t)pl->Dp, 4) * 4 + 64;
    return AC_OK;
}

thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

}  // namespace

extern "C" int ac_knn_set_profile_events(void* start_event, void* stop_event) {
    g_prof_start = (hipEvent_t)start_event;
    g_prof_stop = (hipEvent_t)stop_event;
    return AC_OK;
}

extern "C" int ac_knn_l2_topk_workspace(int64_t N, int D, int nq, int k, size_t* bytes) {
    AC_REQUIRE(bytes != nullptr, AC_EINVAL, "knn workspace: bytes is NULL");
    Plan pl;
    int rc = make_plan(N, D, nq, k, &pl);
    if (rc != AC_OK) return rc;
    *bytes = pl.total;
    return AC_OK;
}

extern "C" int ac_knn_l2_topk(const float* d_P, int64_t N, int64_t ldP, int D, const float* d_Q,
                              int nq, int64_t ldQ, int k, int64_t row_offset, float* d_outD,
                              int64_t* d_outI, void* d_ws, size_t ws_bytes, int32_t* d_stats,
                              ac_stream_t stream_) {
    return ac_knn_l2_topk_x(d_P, N, ldP, D, d_Q, nq, ldQ, k, row_offset, d_outD, nullptr, d_outI, d_ws, ws_bytes,
                            d_stats, stream_);
}

extern "C" int ac_knn_l2_topk_x(const float* d_P, int64_t N, int64_t ldP, int D, const float* d_Q,
                                int nq, int64_t ldQ, int k, int64_t row_offset, float* d_outD, double* d_outD64,
                                int64_t* d_outI, void* d_ws, size_t ws_bytes, int32_t* d_stats,
                                ac_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Plan pl;
    int rc = make_plan(N, D, nq, k, &pl);
    if (rc != AC_OK) return rc;
    if (nq == 0) return AC_OK;
    AC_REQUIRE(d_Q && d_outD && d_outI, AC_EINVAL, "knn: null pointer");
    AC_REQUIRE(ldQ >= D, AC_EINVAL, "knn: ldQ=%lld < D=%d", (long long)ldQ, D);
    AC_REQUIRE(ws_bytes >= pl.total && (d_ws || pl.total == 0), AC_EWORKSPACE,
               "knn: workspace %zu < required %zu", ws_bytes, pl.total);
    if (N > 0) {
        AC_REQUIRE(d_P != nullptr, AC_EINVAL, "knn: d_P is NULL");
        AC_REQUIRE(ldP >= pl.Dp && (ldP % 4) == 0, AC_EINVAL,
                   "knn: ldP=%lld must be a multiple of 4 and >= round_up(D,4)=%d", (long long)ldP, pl.Dp);
        AC_REQUIRE((((uintptr_t)d_P) & 15) == 0, AC_EINVAL, "knn: d_P must be 16-byte aligned");
    }
    char* ws = (char*)d_ws;
    if (d_stats && (pl.small || N == 0)) AC_HIP_CHECK(hipMemsetAsync(d_stats, 0, 4 * sizeof(int32_t), stream));
    if (pl.small) {
        MergeParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.P = d_P; sp.N = N; sp.ldP = ldP; sp.Q = d_Q; sp.ldQ = ldQ; sp.D = D; sp.k = k; sp.row_offset = row_offset;
        sp.outD = d_outD; sp.outD64 = d_outD64; sp.outI = d_outI;
        const size_t lds = (size_t)pl.small_pow2 * 12;
        (void)hipFuncSetAttribute((const void*)knn_small_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(knn_small_exact, dim3(nq), dim3(kSmallThreads), lds, stream, sp, pl.small_pow2);
        AC_LAUNCH_CHECK();
        return AC_OK;
    }

    MergeParams mp;
    mp.P = d_P; mp.N = N; mp.ldP = ldP; mp.Q = d_Q; mp.ldQ = ldQ; mp.D = D; mp.Dp = pl.Dp;
    mp.k = k; mp.kp = pl.kp; mp.G = pl.G; mp.nblk = pl.G * pl.nqt;
    mp.gamma = 1.01 * (double)(pl.ng * kGroup * 16 + 16) * 5.9604644775390625e-08;    // n * 2^-24, n roundings per term
    mp.cand_cnt = nullptr; mp.cand_cap = 0; mp.cand_segs = 1; mp.run_stride = 0; mp.cand_cnt_clear = nullptr; mp.thr_out = nullptr;
    mp.row_offset = row_offset;
    mp.part_d = (const float*)(ws + pl.off_part_d);
    mp.part_i = (const int32_t*)(ws + pl.off_part_i);
    mp.part_maxnorm = (const float*)(ws + pl.off_maxnorm);
    mp.outD = d_outD; mp.outD64 = d_outD64; mp.outI = d_outI;
    mp.flags = (int32_t*)(ws + pl.off_flags);
    mp.stats = d_stats;
    mp.fb_S = pl.fb_S; mp.fb_F = pl.fb_F;
    mp.fb_d = (double*)(ws + pl.off_fb_d);
    mp.fb_i = (int32_t*)(ws + pl.off_fb_i);
    mp.fb_slotctr = (int32_t*)(ws + pl.off_fb_ctr);
    if (N == 0) AC_HIP_CHECK(hipMemsetAsync(ws + pl.off_fb_ctr, 0, 256, stream));      // (otherwise the sweep's workgroup 0 clears it)

    if (N == 0) {
        // empty shard: everything is padding; reuse the merge kernel with all-padding partials
        AC_HIP_CHECK(hipMemsetAsync(ws + pl.off_part_i, 0xff, (size_t)pl.nqt * pl.TQ * pl.G * pl.kp * 4, stream));
        AC_HIP_CHECK(hipMemsetAsync(ws + pl.off_maxnorm, 0, (size_t)pl.G * pl.nqt * 4, stream));
    } else {
        SweepParams sp;
        sp.P = d_P; sp.N = N; sp.ldP = ldP; sp.Q = d_Q; sp.ldQ = ldQ; sp.D = D; sp.Dp = pl.Dp;
        sp.ng = pl.ng; sp.nq = nq; sp.kp = pl.kp; sp.cap = pl.cap; sp.G = pl.G; sp.nqt = pl.nqt;
        sp.ntiles = pl.ntiles;
        sp.part_d = (float*)(ws + pl.off_part_d);
        sp.part_i = (int32_t*)(ws + pl.off_part_i);
        sp.part_maxnorm = (float*)(ws + pl.off_maxnorm);
        sp.zeros = zeros_device();                    // a zero-initialised __device__ array: no memset launch per call
        sp.clear_ctr = (int32_t*)(ws + pl.off_fb_ctr);
        sp.clear_stats = d_stats;
        const int nblk = pl.G * pl.nqt;
        if (g_prof_start && g_prof_stop) AC_HIP_CHECK(hipEventRecord(g_prof_start, stream));
        if (pl.ring) {
            if (D <= 768) {
                AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_sweep_ring<4, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.sweep_lds));
                hipLaunchKernelGGL((knn_sweep_ring<4, 24>), dim3(nblk), dim3(kThreads), pl.sweep_lds, stream, sp);
            } else {
                AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_sweep_ring<4, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.sweep_lds));
                hipLaunchKernelGGL((knn_sweep_ring<4, 32>), dim3(nblk), dim3(kThreads), pl.sweep_lds, stream, sp);
            }
        } else if (pl.TQ == 32) {
            (void)hipFuncSetAttribute((const void*)knn_sweep<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)pl.sweep_lds);
            hipLaunchKernelGGL(knn_sweep<2>, dim3(nblk), dim3(kThreads), pl.sweep_lds, stream, sp);
        } else {
            (void)hipFuncSetAttribute((const void*)knn_sweep<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)pl.sweep_lds);
            hipLaunchKernelGGL(knn_sweep<1>, dim3(nblk), dim3(kThreads), pl.sweep_lds, stream, sp);
        }
        AC_LAUNCH_CHECK();
        if (g_prof_start && g_prof_stop) AC_HIP_CHECK(hipEventRecord(g_prof_stop, stream));
    }
    (void)hipFuncSetAttribute((const void*)knn_merge_rerank, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)pl.merge_lds);
    hipLaunchKernelGGL(knn_merge_rerank, dim3(nq), dim3(kMergeThreads), pl.merge_lds, stream, mp);
    AC_LAUNCH_CHECK();
    if (N > 0) {
        (void)hipFuncSetAttribute((const void*)knn_exact_fallback, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)pl.fb_lds);
        hipLaunchKernelGGL(knn_exact_fallback, dim3(pl.fb_S, nq < kFbQueryGroups ? nq : kFbQueryGroups), dim3(kFbThreads), pl.fb_lds, stream, mp, nq);
        AC_LAUNCH_CHECK();
        const int np2 = next_pow2(pl.fb_S * k > 2 ? pl.fb_S * k : 2);
        (void)hipFuncSetAttribute((const void*)knn_exact_fb_merge, hipFuncAttributeMaxDynamicSharedMemorySize, np2 * 12);
        hipLaunchKernelGGL(knn_exact_fb_merge, dim3(nq < kFbMergeGroups ? nq : kFbMergeGroups), dim3(256), (size_t)np2 * 12, stream, mp, np2, nq);
        AC_LAUNCH_CHECK();
    }
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------------------
// batched search with a prepared store (fp16 plane + row norms): sample -> thresholds -> GEMM-form filter ->
// the same merge / fp64 re-rank / certificate / exact fallback as above (knn_batch.hip explains the scheme)
// ---------------------------------------------------------------------------------------------------------
namespace {

constexpr int kBatchPad = 24;            // k' = k + 24 candidates re-ranked per query (the certificate's slack)
constexpr int kBatchMaxK = 100;
constexpr int64_t kBatchMinRows = 65536;

constexpr int kTwoPhaseCtlInts = 512;       // >= sizeof(acp::GridCtl) / 4
struct BatchPlan {
    int kp, cap, Dp;
    int64_t stride, stride_a, S, q_rows;
    int segs;
    size_t off_sD32, off_sD64, off_sI, off_thr, off_qfac, off_cnt, off_ctl, off_wgmin, off_qp, off_cd, off_ci, off_flags, off_fb_d, off_fb_i,
        off_fb_ctr, off_sub, sub_bytes, total;
    int fb_S, fb_F;
    size_t merge_lds, fb_lds;
};

int make_batch_plan(int64_t N, int D, int nq, int k, BatchPlan* bp) {
    AC_REQUIRE(N >= kBatchMinRows && N < 2147483647LL, AC_EUNSUPPORTED, "knn batch: N=%lld outside [%lld, 2^31)", (long long)N,
               (long long)kBatchMinRows);
    AC_REQUIRE(D >= 1 && nq >= 1 && k >= 1 && k <= kBatchMaxK, AC_EUNSUPPORTED, "knn batch: D=%d nq=%d k=%d unsupported (k <= %d)",
               D, nq, k, kBatchMaxK);
    bp->kp = k + kBatchPad;
    bp->Dp = (D + 3) / 4 * 4;
    // Thresholds come from strided SAMPLES of the store swept by the same GEMM-form kernel (knn_batch.hip) and re-ranked exactly
    // by the merge kernel: tau_q = the largest exact distance among the k' sample rows with the smallest sweep values -- k' rows
    // of the store within tau_q, so the store's k'-th smallest distance is <= tau_q whatever the sweep's rounding did.
    //   stage A: every `stride_a`-th row (max(4096, 128 k') rows), no threshold yet: every lane offers the best of its 32 rows per query;
    //   stage B (stores beyond ~0.5 M rows): every `stride`-th row filtered by stage A's tau (k' stride_a / stride expected);
    //   main:    all rows filtered by the last tau: k' * stride candidates expected, at most half the candidate buffer.
    // (Round 2 searched ONE sample of N / 64 rows exactly with the fp32 sweep: 10 ms at 4096 x 10M, 0.17 ms of the 0.39 ms a
    //  256 x 100k call takes.)
    int64_t smax = 8192 / bp->kp;
    if (smax > 128) smax = 128;
    if (smax < 1) smax = 1;
    // stage A offers one row per 32 sample rows and query (the best of each lane's rows): with >= 4 k' offers the k' best of them
    // are, up to rare collisions, the sample's k' best, so tau_A is as tight as an exact search of the sample would make it
    const int64_t rows_a = 128 * (int64_t)bp->kp > 4096 ? 128 * (int64_t)bp->kp : 4096;
    int64_t stride_a = (N + rows_a - 1) / rows_a, stride = stride_a;
    if (stride > smax) {                               // stage B's (or, for small stores, stage A's own) stride
        // Stage B keeps the sample rows within tau_A and needs k' of them: k' S_B / S_A are expected, so its sample must be a few
        // times stage A's.  (Round 3 used smax whatever stride_a was: at 0.92 - 1.5 M rows -- stride_a just above smax, e.g. a
        // 10M-row store sharded 8 ways -- that is k' x 1.1 .. 1.4 expected candidates, a good share of the queries came out of
        // stage B with fewer than k', lost their threshold, overflowed their candidate buffer in the main sweep and took the
        // exact fallback: 1.3 s instead of 8 ms per 4096-query batch at 1M rows.)
        stride = smax;
        if (stride > stride_a / 3) stride = stride_a / 3;
        if (stride < 1) stride = 1;
    }
    bp->stride_a = stride_a;
    bp->stride = stride;
    bp->S = ac::knn_sample_rows(N, stride_a);            // rows of the stage-A sample
    const int64_t expect = (int64_t)bp->kp * stride;             // E[candidates per query] of the main sweep = k' * stride
    int cap = 1024;
    while (cap < 2 * expect && cap < 16384) cap <<= 1;
    // Short launches (a sample stage, or the whole sweep of a small store) fire all their appends in one burst, and the
    // device-scope atomics that reserve the slots serialise per address (~0.1 - 0.5 us each: 84 of the 140 us of a 256 x 100k
    // sweep with one counter per query).  Such launches split a query's list into up to 64 segments with a counter each
    // (>= 256 entries per segment: a segment overflows no more easily than the whole list would); long launches keep ONE list
    // -- their workgroups drift apart, and a single hot counter line per query is then the cheaper form (10M x 768 x 4096: 68.8 ms
    // against 75.8 with 16 segments).  batch_segs() picks per launch.
    const int segs_want = nq <= 256 ? 64 : (nq <= 1024 ? 32 : 16);
    if (cap < 256 * segs_want) cap = 256 * segs_want;
    bp->cap = cap;
    bp->segs = cap / 256 < segs_want ? cap / 256 : segs_want;          // the most any launch of this call uses
    bp->q_rows = ((int64_t)nq + 255) / 256 * 256;
    const size_t sub = 256;                                // (the exact sample search of round 2 needed a workspace of its own)
    bp->sub_bytes = sub;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += ac::align_up(n, 256); return o; };
    bp->off_sD32 = take((size_t)nq * bp->kp * 4);
    bp->off_sD64 = take((size_t)nq * bp->kp * 8);
    bp->off_sI = take((size_t)nq * bp->kp * 8);
    bp->off_thr = take((size_t)bp->q_rows * 4);
    bp->off_qfac = take((size_t)bp->q_rows * 4);
    bp->off_cnt = take((size_t)bp->q_rows * 4 * bp->segs);
    bp->off_ctl = take(kTwoPhaseCtlInts * 4);             // (right behind the counters: zeroed by the same launch) grid-barrier words
    bp->off_wgmin = take(ac::knn_batch_two_phase_bytes());   // two-phase thresholds (knn_batch.hip): per-workgroup minima
    bp->off_qp = take(ac::knn_planes_bytes(nq, D));
    bp->off_cd = take((size_t)bp->q_rows * cap * 4);
    bp->off_ci = take((size_t)bp->q_rows * cap * 4);
    bp->off_flags = take((size_t)nq * 4);
    bp->fb_S = 4096 / next_pow2(k);
    if (bp->fb_S > 64) bp->fb_S = 64;
    if (bp->fb_S < 1) bp->fb_S = 1;
    bp->fb_F = nq < 64 ? nq : 64;
    bp->off_fb_d = take((size_t)bp->fb_F * bp->fb_S * k * 8);
    bp->off_fb_i = take((size_t)bp->fb_F * bp->fb_S * k * 4);
    bp->off_fb_ctr = take(256);
    bp->off_sub = take(sub);
    bp->total = off;
    // (the static LDS of the merge kernel -- segment offsets, scan totals -- is ~1.1 KB)
    bp->merge_lds = 2 * ac::align_up((size_t)cap * 4, 16) + ac::align_up((size_t)bp->Dp * 4, 16) + (size_t)bp->kp * 16 + 256 * 4 + 64;   // (segmented form; one segment needs cap * 4 less)
    bp->fb_lds = (size_t)kFbCap * 12 + ac::align_up((size_t)bp->Dp, 4) * 4 + 64;
    return AC_OK;
}


// ---- the fp16-plane sweep for small batches (knn_plane_sweep): 1 .. kPlaneMaxQueries queries against a prepared store ----
constexpr int kPlaneMaxQueries = 64;

struct PlanePlan {
    int TQ, nqt, kp, cap, G, Dp, Kp;
    int64_t ntiles, q_rows;
    size_t sweep_lds, merge_lds, fb_lds;
    size_t off_part_d, off_part_i, off_flags, off_qp, off_qfac, off_thr, off_fb_d, off_fb_i, off_fb_ctr, total;
    int fb_S, fb_F;
};

// false: the shape does not fit (LDS) or the form is switched off -- the caller uses the GEMM-form path instead
bool make_plane_plan(int64_t N, int D, int nq, int k, PlanePlan* pp) {
    static const int plane_env = getenv("AC_KNN_PLANE") ? atoi(getenv("AC_KNN_PLANE")) : -1;       // 0 = never (A/B)
    if (plane_env == 0 || nq < 1 || nq > kPlaneMaxQueries || N < 1 || N >= 2147483647LL - 512 || k < 1 || k > kBatchMaxK) return false;
    pp->kp = k + kBatchPad;
    pp->Dp = (D + 3) / 4 * 4;
    pp->Kp = (D + 63) / 64 * 64;
    const int nk = pp->Kp / 16;
    auto lds_for = [&](int TQ, int cap) {
        return (size_t)nk * (TQ / 32) * 1024 + (size_t)TQ * cap * 8 + (size_t)TQ * 8 + 2 * 8 * 4 + 64;
    };
    const int cap_full = (2 * pp->kp + 15) / 16 * 16, cap_min = (pp->kp + 32 + 15) / 16 * 16;
    const int tq_first = nq > 32 ? 64 : 32;
    pp->TQ = 0;
    for (int TQ = tq_first; TQ >= 32 && !pp->TQ; TQ -= 32)
        for (int cap = cap_full; cap >= cap_min; cap -= 16)
            if (lds_for(TQ, cap) <= (size_t)kLdsLimit) { pp->TQ = TQ; pp->cap = cap; break; }
    if (!pp->TQ || pp->cap > 512) return false;
    pp->sweep_lds = lds_for(pp->TQ, pp->cap);
    pp->nqt = (nq + pp->TQ - 1) / pp->TQ;
    pp->ntiles = (N + 255) / 256;
    int64_t G = ac::dev_info().cus;                       // one 8-wave workgroup per CU: exactly one residency round
    if (G > pp->ntiles) G = pp->ntiles;
    if (G > kMergeMaxCand / pp->kp) G = kMergeMaxCand / pp->kp;
    if (const char* e = getenv("AC_KNN_G")) { int64_t v = atoll(e); if (v >= 1 && v <= G) G = v; }   // tuning experiments
    pp->G = (int)G;
    pp->q_rows = 256;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += ac::align_up(n, 256); return o; };
    const size_t ncand = (size_t)pp->nqt * pp->TQ * pp->G * pp->kp;
    pp->off_part_d = take(ncand * 4);
    pp->off_part_i = take(ncand * 4);
    pp->off_flags = take((size_t)nq * 4);
    pp->off_qp = take(ac::knn_planes_bytes(nq, D));
    pp->off_qfac = take((size_t)pp->q_rows * 4);
    pp->off_thr = take((size_t)pp->q_rows * 4);
    pp->fb_S = 4096 / next_pow2(k);
    if (pp->fb_S > 64) pp->fb_S = 64;
    if (pp->fb_S < 1) pp->fb_S = 1;
    pp->fb_F = nq;
    pp->off_fb_d = take((size_t)pp->fb_F * pp->fb_S * k * 8);
    pp->off_fb_i = take((size_t)pp->fb_F * pp->fb_S * k * 4);
    pp->off_fb_ctr = take(256);
    pp->total = off;
    pp->merge_lds = ac::align_up((size_t)pp->G * pp->kp * 4, 16) + ac::align_up((size_t)pp->Dp * 4, 16) + (size_t)pp->kp * 16 + 256 * 4 + 64;
    pp->fb_lds = (size_t)kFbCap * 12 + ac::align_up((size_t)pp->Dp, 4) * 4 + 64;
    return true;
}

int plane_search(const PlanePlan& pp, const float* d_P, int64_t N, int64_t ldP, int D, const uint16_t* d_planes, const float* d_norms,
                 const float* d_Q, int nq, int64_t ldQ, int k, int64_t row_offset, float* d_outD, double* d_outD64, int64_t* d_outI,
                 char* ws, int32_t* d_stats, hipStream_t stream) {
    const int64_t np = (N + 255) / 256 * 256;
    const uint32_t* d_maxnorm = reinterpret_cast<const uint32_t*>(d_norms + np);
    const double gamma = ac::knn_batch_gamma(D);
    // 1. the queries' fp16 plane (each scaled by its own power of two) and epilogue factors -2 2^(e_p + e_q): the same kernel,
    //    scaling and rounding as the GEMM-form path, hence the same error bound
    int rc = ac::knn_prepare_queries(nullptr, pp.kp, d_Q, ldQ, D, nq, d_maxnorm, gamma, (uint16_t*)(ws + pp.off_qp),
                                     (float*)(ws + pp.off_thr), (float*)(ws + pp.off_qfac), stream);
    if (rc != AC_OK) return rc;
    // 2. one pass over the plane per query tile
    PlaneSweepParams sp;
    sp.Pp = d_planes; sp.pnorm = d_norms; sp.Qp = (const uint16_t*)(ws + pp.off_qp); sp.q_rows = pp.q_rows;
    sp.qfac = (const float*)(ws + pp.off_qfac); sp.N = N; sp.ntiles = pp.ntiles; sp.Kp = pp.Kp;
    sp.kp = pp.kp; sp.cap = pp.cap; sp.G = pp.G;
    sp.part_d = (float*)(ws + pp.off_part_d); sp.part_i = (int32_t*)(ws + pp.off_part_i);
    if (g_prof_start && g_prof_stop) AC_HIP_CHECK(hipEventRecord(g_prof_start, stream));
    for (int qt = 0; qt < pp.nqt; ++qt) {
        sp.q0 = qt * pp.TQ; sp.nq = nq - sp.q0 < pp.TQ ? nq - sp.q0 : pp.TQ;
        sp.clear_ctr = qt == 0 ? (int32_t*)(ws + pp.off_fb_ctr) : nullptr;
        sp.clear_stats = qt == 0 ? d_stats : nullptr;
        if (pp.TQ == 64) {
            AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_plane_sweep<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp.sweep_lds));
            hipLaunchKernelGGL(knn_plane_sweep<64>, dim3(pp.G), dim3(kThreads), pp.sweep_lds, stream, sp);
        } else {
            AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_plane_sweep<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp.sweep_lds));
            hipLaunchKernelGGL(knn_plane_sweep<32>, dim3(pp.G), dim3(kThreads), pp.sweep_lds, stream, sp);
        }
        AC_LAUNCH_CHECK();
    }
    if (g_prof_start && g_prof_stop) AC_HIP_CHECK(hipEventRecord(g_prof_stop, stream));
    // 3. merge of the G per-workgroup lists + exact fp64 re-rank + certificate with the fp16 bound, then the exact fallback
    MergeParams mp;
    mp.P = d_P; mp.N = N; mp.ldP = ldP; mp.Q = d_Q; mp.ldQ = ldQ; mp.D = D; mp.Dp = pp.Dp;
    mp.k = k; mp.kp = pp.kp; mp.G = pp.G; mp.nblk = 1; mp.gamma = gamma;
    mp.cand_cnt = nullptr; mp.cand_cap = 0; mp.cand_segs = 1; mp.run_stride = 0; mp.cand_cnt_clear = nullptr; mp.thr_out = nullptr;
    mp.row_offset = row_offset;
    mp.part_d = (const float*)(ws + pp.off_part_d); mp.part_i = (const int32_t*)(ws + pp.off_part_i);
    mp.part_maxnorm = reinterpret_cast<const float*>(d_maxnorm);
    mp.outD = d_outD; mp.outD64 = d_outD64; mp.outI = d_outI;
    mp.flags = (int32_t*)(ws + pp.off_flags);
    mp.stats = d_stats;
    mp.fb_S = pp.fb_S; mp.fb_F = pp.fb_F;
    mp.fb_d = (double*)(ws + pp.off_fb_d); mp.fb_i = (int32_t*)(ws + pp.off_fb_i); mp.fb_slotctr = (int32_t*)(ws + pp.off_fb_ctr);
    AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_merge_rerank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp.merge_lds));
    hipLaunchKernelGGL(knn_merge_rerank, dim3(nq), dim3(kMergeThreads), pp.merge_lds, stream, mp);
    AC_LAUNCH_CHECK();
    (void)hipFuncSetAttribute((const void*)knn_exact_fallback, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp.fb_lds);
    hipLaunchKernelGGL(knn_exact_fallback, dim3(pp.fb_S, nq < kFbQueryGroups ? nq : kFbQueryGroups), dim3(kFbThreads), pp.fb_lds, stream, mp, nq);
    AC_LAUNCH_CHECK();
    const int np2 = next_pow2(pp.fb_S * k > 2 ? pp.fb_S * k : 2);
    (void)hipFuncSetAttribute((const void*)knn_exact_fb_merge, hipFuncAttributeMaxDynamicSharedMemorySize, np2 * 12);
    hipLaunchKernelGGL(knn_exact_fb_merge, dim3(nq < kFbMergeGroups ? nq : kFbMergeGroups), dim3(256), (size_t)np2 * 12, stream, mp, np2, nq);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

}  // namespace

extern "C" int ac_knn_store_bytes(int64_t N, int D, size_t* planes_bytes, size_t* norms_bytes) {
    AC_REQUIRE(planes_bytes && norms_bytes && N >= 0 && D >= 1, AC_EINVAL, "knn_store_bytes: bad arguments");
    *planes_bytes = ac::knn_planes_bytes(N, D);
    *norms_bytes = (size_t)((N + 255) / 256 * 256 + 64) * sizeof(float);      // |p|^2 per row, +inf tile padding, the maximum at [round_up(N, 256)]
    return AC_OK;
}

extern "C" int ac_knn_prepare_store(const float* d_P, int64_t N, int64_t ldP, int D, uint16_t* d_planes, float* d_norms,
                                    ac_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(d_P && d_planes && d_norms && N >= 1 && D >= 1 && ldP >= D, AC_EINVAL, "knn_prepare_store: bad arguments");
    AC_REQUIRE((((uintptr_t)d_planes) & 15) == 0 && (((uintptr_t)d_norms) & 15) == 0, AC_EINVAL,
               "knn_prepare_store: planes / norms must be 16-byte aligned");
    const int64_t np = (N + 255) / 256 * 256;
    AC_HIP_CHECK(hipMemsetAsync(d_norms + np, 0, 64 * sizeof(float), stream));          // [np] = max |p|^2 (float bits)
    return ac::knn_prepare_store(d_P, ldP, N, D, d_planes, d_norms, reinterpret_cast<uint32_t*>(d_norms + np), stream);
}

extern "C" int ac_knn_update_store(const float* d_P, int64_t N_old, int64_t N_new, int64_t ldP, int D, uint16_t* d_planes,
                                   float* d_norms, int64_t row0, int64_t nrows, int32_t* d_exponent_changed, ac_stream_t stream_) {
    AC_REQUIRE(d_P && d_planes && d_norms && d_exponent_changed && D >= 1 && ldP >= D, AC_EINVAL, "knn_update_store: bad arguments");
    AC_REQUIRE(N_old >= 1 && N_new >= N_old && row0 >= 0 && nrows >= 1 && row0 + nrows <= N_new, AC_EINVAL,
               "knn_update_store: rows [%lld, %lld) outside a store of %lld rows (was %lld)", (long long)row0, (long long)(row0 + nrows),
               (long long)N_new, (long long)N_old);
    AC_REQUIRE(N_new == N_old || (row0 + nrows == N_new && row0 <= N_old), AC_EINVAL,
               "knn_update_store: an append must cover every new row: [row0, row0 + nrows) = [<= N_old, N_new)");
    return ac::knn_update_store(d_P, ldP, N_old, N_new, D, d_planes, d_norms, row0, nrows, d_exponent_changed, (hipStream_t)stream_);
}

extern "C" int ac_knn_l2_topk_batch_workspace(int64_t N, int D, int nq, int k, size_t* bytes) {
    AC_REQUIRE(bytes != nullptr, AC_EINVAL, "knn batch workspace: bytes is NULL");
    BatchPlan bp;
    int rc = make_batch_plan(N, D, nq, k, &bp);
    if (rc != AC_OK) return rc;
    *bytes = bp.total;
    PlanePlan pp;                                       // small batches take the fp16-plane sweep (knn_plane_sweep)
    if (make_plane_plan(N, D, nq, k, &pp) && pp.total > *bytes) *bytes = pp.total;
    return AC_OK;
}

// segments of a query's candidate list for a launch that sweeps `rows` rows (see make_batch_plan)
static int batch_segs(const BatchPlan& bp, int64_t rows, int nq) {
    const int64_t tiles = ((rows + 255) / 256) * (((int64_t)nq + 255) / 256);
    const int64_t per_wg = (tiles + ac::dev_info().cus - 1) / ac::dev_info().cus;
    return per_wg <= 8 ? bp.segs : 1;
}

extern "C" int ac_knn_l2_topk_batch(const float* d_P, int64_t N, int64_t ldP, int D, const uint16_t* d_planes,
                                    const float* d_norms, const float* d_Q, int nq, int64_t ldQ, int k, int64_t row_offset,
                                    float* d_outD, double* d_outD64, int64_t* d_outI, void* d_ws, size_t ws_bytes,
                                    int32_t* d_stats, ac_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    BatchPlan bp;
    int rc = make_batch_plan(N, D, nq, k, &bp);
    if (rc != AC_OK) return rc;
    AC_REQUIRE(d_P && d_planes && d_norms && d_Q && d_outD && d_outI, AC_EINVAL, "knn batch: null pointer");
    AC_REQUIRE(ldQ >= D && ldP >= bp.Dp && (ldP % 4) == 0 && (((uintptr_t)d_P) & 15) == 0, AC_EINVAL, "knn batch: bad leading dimension / alignment");
    AC_REQUIRE(d_ws && ws_bytes >= bp.total, AC_EWORKSPACE, "knn batch: workspace %zu < required %zu", ws_bytes, bp.total);
    char* ws = (char*)d_ws;
    {   // <= 64 queries: bandwidth-bound, ONE pass over the fp16 plane with the query tile resident (knn_plane_sweep)
        PlanePlan pp;
        if (make_plane_plan(N, D, nq, k, &pp)) {
            AC_REQUIRE(ws_bytes >= pp.total, AC_EWORKSPACE, "knn batch: workspace %zu < required %zu", ws_bytes, pp.total);
            return plane_search(pp, d_P, N, ldP, D, d_planes, d_norms, d_Q, nq, ldQ, k, row_offset, d_outD, d_outD64, d_outI, ws,
                                d_stats, stream);
        }
    }
    const int64_t np = (N + 255) / 256 * 256;
    const uint32_t* d_maxnorm = reinterpret_cast<const uint32_t*>(d_norms + np);
    // |v - exact| <= gamma (max|p| + |q|)^2 for the one-product fp16 sweep (knn_batch.hip; derivation at the declaration of
    // this entry point in include/acamd.h)
    const double gamma = ac::knn_batch_gamma(D);

    // 1. per query: epilogue factor -2 2^(e_p + e_q), fp16 plane of q 2^-e_q, threshold +inf (keep everything)
    //    (the same launch zeroes the candidate counters of every segmentation this call uses: no memset launches below)
    rc = ac::knn_prepare_queries(nullptr, bp.kp, d_Q, ldQ, D, nq, d_maxnorm, gamma,
                                 (uint16_t*)(ws + bp.off_qp), (float*)(ws + bp.off_thr), (float*)(ws + bp.off_qfac), stream,
                                 (int32_t*)(ws + bp.off_cnt), (int64_t)bp.q_rows * bp.segs + kTwoPhaseCtlInts);
    if (rc != AC_OK) return rc;
    // 2. threshold stages: sweep a strided sample, re-rank its k' best exactly (knn_merge_rerank in candidate mode, asked for
    //    k' results; its certificate is irrelevant here -- ANY k' rows bound the k'-th smallest distance from above)
    MergeParams sp;
    sp.P = d_P; sp.Q = d_Q; sp.ldQ = ldQ; sp.D = D; sp.Dp = bp.Dp;
    sp.k = bp.kp; sp.kp = bp.kp; sp.G = 1; sp.nblk = 1; sp.gamma = gamma;
    sp.cand_cnt = (const int32_t*)(ws + bp.off_cnt); sp.cand_cap = bp.cap; sp.cand_segs = bp.segs; sp.row_offset = 0;
    // a threshold stage's merge writes the new threshold itself and zeroes the counters it has read (the next sweep appends
    // into them): round 3 spent a knn_thr_kernel launch and a memset launch per stage on that
    sp.cand_cnt_clear = (int32_t*)(ws + bp.off_cnt); sp.thr_out = (float*)(ws + bp.off_thr);
    sp.part_d = (const float*)(ws + bp.off_cd); sp.part_i = (const int32_t*)(ws + bp.off_ci);
    sp.part_maxnorm = reinterpret_cast<const float*>(d_maxnorm);
    sp.outD = (float*)(ws + bp.off_sD32); sp.outD64 = (double*)(ws + bp.off_sD64); sp.outI = (int64_t*)(ws + bp.off_sI);
    sp.flags = (int32_t*)(ws + bp.off_flags); sp.stats = nullptr;
    sp.fb_S = bp.fb_S; sp.fb_F = 0; sp.fb_d = nullptr; sp.fb_i = nullptr; sp.fb_slotctr = (int32_t*)(ws + bp.off_fb_ctr) + 8;
    static const bool thr_exact = [] { const char* e = getenv("AC_KNN_THR_EXACT"); return e && atoi(e) != 0; }();     // (A/B: the re-ranked form)
    sp.thr_only = thr_exact ? 0 : 1;
    AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_merge_rerank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp.merge_lds));
    static const bool dbg = getenv("AC_KNN_BATCH_DEBUG") != nullptr;
    const int64_t stage_stride[2] = {bp.stride_a, bp.stride};
    // the main sweep takes its thresholds from its own first tile round where it can (knn_batch.hip two_phase): no sample stages
    const bool two_phase = ac::knn_batch_two_phase_applies(N, nq, bp.kp, batch_segs(bp, N, nq));
    const int nstages = two_phase ? 0 : (bp.stride_a > bp.stride ? 2 : 1);
    for (int st = 0; st < nstages; ++st) {
        const int64_t sst = stage_stride[st];
        const int segs = batch_segs(bp, ac::knn_sample_rows(N, sst), nq);
        const size_t mlds = bp.merge_lds - (segs > 1 ? 0 : ac::align_up((size_t)bp.cap * 4, 16));
        sp.cand_segs = segs;
        rc = ac::knn_batch_launch(d_planes, d_norms, N, D, (const uint16_t*)(ws + bp.off_qp), nq, (const float*)(ws + bp.off_thr),
                                  (const float*)(ws + bp.off_qfac), (float*)(ws + bp.off_cd), (int32_t*)(ws + bp.off_ci),
                                  (int32_t*)(ws + bp.off_cnt), bp.cap, segs, sst, st == 0 ? 1 : 0, stream);
        if (rc != AC_OK) return rc;
        sp.N = ac::knn_sample_rows(N, sst); sp.ldP = ldP; sp.run_stride = sst > 1 ? 8 * sst : 0;   // sample row i = store row (i >> 3) * 8 sst + (i & 7)
        hipLaunchKernelGGL(knn_merge_rerank, dim3(nq), dim3(kMergeThreads), mlds, stream, sp);
        AC_LAUNCH_CHECK();
        if (dbg) {
            AC_HIP_CHECK(hipStreamSynchronize(stream));
            int32_t cnt[4]; float thr[4]; double tau[4];
            AC_HIP_CHECK(hipMemcpy(cnt, ws + bp.off_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
            AC_HIP_CHECK(hipMemcpy(thr, ws + bp.off_thr, sizeof(thr), hipMemcpyDeviceToHost));
            for (int i = 0; i < 4; ++i)
                AC_HIP_CHECK(hipMemcpy(&tau[i], ws + bp.off_sD64 + ((size_t)(i < nq ? i : 0) * bp.kp + bp.kp - 1) * 8, 8, hipMemcpyDeviceToHost));
            fprintf(stderr, "knn batch stage %d: stride %lld rows %lld cap %d kp %d | cand %d %d %d %d | tau %g %g %g %g | thr %g %g %g %g\n", st,
                    (long long)sst, (long long)sp.N, bp.cap, bp.kp, cnt[0], cnt[1], cnt[2], cnt[3], tau[0], tau[1], tau[2], tau[3], thr[0], thr[1], thr[2], thr[3]);
        }
    }
    const int msegs = batch_segs(bp, N, nq);
    const size_t mlds = bp.merge_lds - (msegs > 1 ? 0 : ac::align_up((size_t)bp.cap * 4, 16));
    // 3. the GEMM-form sweep: candidates (row, v) with v below the query's threshold (its workgroup 0 also zeroes the caller's
    //    d_stats and the fallback's slot counter, which the merge after it increments)
    if (g_prof_start && g_prof_stop) AC_HIP_CHECK(hipEventRecord(g_prof_start, stream));
    rc = ac::knn_batch_launch(d_planes, d_norms, N, D, (const uint16_t*)(ws + bp.off_qp), nq, (const float*)(ws + bp.off_thr),
                              (const float*)(ws + bp.off_qfac), (float*)(ws + bp.off_cd), (int32_t*)(ws + bp.off_ci), (int32_t*)(ws + bp.off_cnt), bp.cap, msegs, 1, 0, stream,
                              (int32_t*)(ws + bp.off_fb_ctr), d_stats, two_phase ? bp.kp : 0, (unsigned*)(ws + bp.off_wgmin), ws + bp.off_ctl);
    if (rc != AC_OK) return rc;
    if (g_prof_start && g_prof_stop) AC_HIP_CHECK(hipEventRecord(g_prof_stop, stream));
    // 4. merge + exact re-rank + certificate, then the exact fallback for uncertified queries
    MergeParams mp;
    mp.P = d_P; mp.N = N; mp.ldP = ldP; mp.Q = d_Q; mp.ldQ = ldQ; mp.D = D; mp.Dp = bp.Dp;
    mp.k = k; mp.kp = bp.kp; mp.G = 1; mp.nblk = 1; mp.gamma = gamma;
    mp.cand_cnt = (const int32_t*)(ws + bp.off_cnt); mp.cand_cap = bp.cap; mp.cand_segs = msegs; mp.run_stride = 0;
    mp.cand_cnt_clear = nullptr; mp.thr_out = nullptr;
    mp.row_offset = row_offset;
    mp.part_d = (const float*)(ws + bp.off_cd); mp.part_i = (const int32_t*)(ws + bp.off_ci);
    mp.part_maxnorm = reinterpret_cast<const float*>(d_maxnorm);
    mp.outD = d_outD; mp.outD64 = d_outD64; mp.outI = d_outI;
    mp.flags = (int32_t*)(ws + bp.off_flags);
    mp.stats = d_stats;
    mp.fb_S = bp.fb_S; mp.fb_F = bp.fb_F;
    mp.fb_d = (double*)(ws + bp.off_fb_d); mp.fb_i = (int32_t*)(ws + bp.off_fb_i); mp.fb_slotctr = (int32_t*)(ws + bp.off_fb_ctr);
    AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_merge_rerank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp.merge_lds));
    hipLaunchKernelGGL(knn_merge_rerank, dim3(nq), dim3(kMergeThreads), mlds, stream, mp);
    AC_LAUNCH_CHECK();
    (void)hipFuncSetAttribute((const void*)knn_exact_fallback, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bp.fb_lds);
    hipLaunchKernelGGL(knn_exact_fallback, dim3(bp.fb_S, nq < kFbQueryGroups ? nq : kFbQueryGroups), dim3(kFbThreads), bp.fb_lds, stream, mp, nq);
    AC_LAUNCH_CHECK();
    const int np2 = next_pow2(bp.fb_S * k > 2 ? bp.fb_S * k : 2);
    (void)hipFuncSetAttribute((const void*)knn_exact_fb_merge, hipFuncAttributeMaxDynamicSharedMemorySize, np2 * 12);
    hipLaunchKernelGGL(knn_exact_fb_merge, dim3(nq < kFbMergeGroups ? nq : kFbMergeGroups), dim3(256), (size_t)np2 * 12, stream, mp, np2, nq);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_topk_merge(const float* d_D_in, const int64_t* d_I_in, int shards, int nq, int k,
                             float* d_outD, int64_t* d_outI, ac_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(shards >= 1 && nq >= 0 && k >= 1, AC_EINVAL, "topk_merge: bad shape");
    AC_REQUIRE(d_D_in && d_I_in && d_outD && d_outI, AC_EINVAL, "topk_merge: null pointer");
    if (nq == 0) return AC_OK;
    const size_t lds = (size_t)shards * k * 12;
    AC_REQUIRE(lds <= 96 * 1024, AC_EUNSUPPORTED, "topk_merge: shards*k=%d too large", shards * k);
    (void)hipFuncSetAttribute((const void*)topk_merge_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(topk_merge_kernel<float>, dim3(nq), dim3(256), lds, stream, d_D_in, d_I_in, shards, nq, k,
                       d_outD, d_outI);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_topk_merge_f64(const double* d_D_in, const int64_t* d_I_in, int shards, int nq, int k,
                                 float* d_outD, int64_t* d_outI, ac_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(shards >= 1 && nq >= 0 && k >= 1, AC_EINVAL, "topk_merge_f64: bad shape");
    AC_REQUIRE(d_D_in && d_I_in && d_outD && d_outI, AC_EINVAL, "topk_merge_f64: null pointer");
    if (nq == 0) return AC_OK;
    const size_t lds = (size_t)shards * k * 16;
    AC_REQUIRE(lds <= 128 * 1024, AC_EUNSUPPORTED, "topk_merge_f64: shards*k=%d too large", shards * k);
    (void)hipFuncSetAttribute((const void*)topk_merge_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(topk_merge_kernel<double>, dim3(nq), dim3(256), lds, stream, d_D_in, d_I_in, shards, nq, k,
                       d_outD, d_outI);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_rows_to_class(const int64_t* d_I, int64_t n, const int32_t* d_row_class, int64_t nrows,
                                const int64_t* d_class_lut, int nlut, int64_t* d_out, ac_stream_t stream_) {
    AC_REQUIRE(d_I && d_out && n >= 0 && nrows >= 0, AC_EINVAL, "rows_to_class: bad arguments");
    if (n == 0) return AC_OK;
    hipLaunchKernelGGL(rows_to_class_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       d_I, n, d_row_class, nrows, d_class_lut, nlut, d_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_proto_scores(const float* d_D, const int64_t* d_I, int nq, int k, float* d_out,
                               ac_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(nq >= 0 && k >= 1 && d_D && d_I && d_out, AC_EINVAL, "proto_scores: bad arguments");
    if (nq == 0) return AC_OK;
    hipLaunchKernelGGL(proto_scores_kernel, dim3(nq), dim3(64), 0, stream, d_D, d_I, nq, k, d_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
