// gemm_ring_nt: the large-M planes GEMM of the encoder (transformers BertModel projections, called at
// /root/reference/src/adaptive_classifier/classifier.py:1271) as ONE persistent workgroup per CU that streams
// its share of (tile, k-stage) units through a deep LDS ring.
//
// Why (profiles/r01/gemm_planes_pmc.json): the 128x128 / three-blocks-per-CU kernel keeps only
// 3 x 24 KB of operand stages in flight per CU and needs 32 B/clk/CU of L2->LDS traffic at full matrix rate; with
// the ~2 us load-to-LDS latency seen under load that is a Little's-law ceiling of ~50 % matrix-pipe busy (52 %
// measured).  This kernel
//   * uses a 256 x 128 tile (8 waves of 64 x 64): 24 B/clk/CU at full rate, and
//   * keeps R - 1 = 3 stages (108 KB) in flight: the ring is filled by global_load_lds and drained with COUNTED
//     s_waitcnt vmcnt(N) + raw s_barrier -- no vmcnt(0) in the loop;
//   * runs its 8 waves as two groups of 4 (one wave per SIMD each), staggered by one barrier: while one group
//     issues the next stage's DMA and reads its fragments (L phase) the other group owns the matrix pipe
//     (M phase: 24 v_mfma_f32_32x32x16_bf16), so fragment-read latency and DMA issue never idle the pipe;
//   * is persistent: a block's units are a contiguous run of k-stages across tiles, so the ring never drains
//     between tiles, and the work is split STREAM-K style -- every CU gets the same number of k-stages whatever
//     the tile count (8192-row encoder shapes are 0.75 - 2.25 tiles per CU).  A tile cut between blocks is
//     finished by the block holding its tail: the others publish fp32 partial accumulators (agent-scope
//     release / acquire, cdna guide G16) and the owner adds them in a fixed order (deterministic).  Tiles are
//     dealt to XCDs first, so a cut tile's partials stay inside one XCD's L2 and an owner only ever waits for
//     LOWER hardware block ids of its own XCD (no deadlock under partial residency).
// Arithmetic: identical to gemm_planes_nt (bf16x3 split, six products smallest first, fp32 accumulate).
#include "common.h"
#include "gemm_common.h"

#include <stdlib.h>

namespace {
using namespace acg;

constexpr int RBM = 256, RBN = 128;            // block tile
constexpr int RGA = RBM / 32, RGW = RBN / 32;  // 32-row groups per operand
constexpr int RNG = RGA + RGW;
constexpr int RNP = 3;                         // planes per operand
constexpr int RPIECES = RNG * RNP;             // 1 KB DMA pieces per stage (36)
constexpr int RPPW = (RPIECES + 7) / 8;        // pieces per wave and stage (5; waves 4-7 re-issue pieces 0-3)
constexpr int RSBK = 16;                       // k per stage
constexpr int kRingThreads = 512;

struct RingParams {
    const uint16_t* Ap; int64_t a_rows;
    const uint16_t* Wp; int64_t w_rows;
    float* C; int64_t ldc;            // fp32 result, or (C_PLANES) the planes of the next GEMM's operand
    int M, N, K;
    int tiles_n, tiles;               // column tiles, total tiles
    int nk;                           // k-stages per tile
    int nblk;                         // grid size (multiple of 8)
    float* partials;                  // [nblk][2][RBM * RBN] fp32: slot 0 = contributed (head / middle), 1 = own deferred tail
    int* flags;                       // [nblk] waves of the block that have published their contributed partial
    Epilogue epi;
};

// Values that only the rare blocks (tile ends, stream-K hand-offs, tile changes of the DMA stream) need are passed
// through an empty asm first: nothing derived from them can be hoisted out of the persistent loop, where it would
// sit in registers (or spill) beside the accumulators and the two fragment sets.
template <typename T> __device__ __forceinline__ T launder_v(T x) { asm volatile("" : "+v"(x)); return x; }
template <typename T> __device__ __forceinline__ T launder_s(T x) { asm volatile("" : "+s"(x)); return x; }

__device__ __forceinline__ void waitcnt_vm(int n) {
    // immediate operand: dispatch on the few values used
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// ABL (ablation builds, tools/gemm_bench only): bit 0 = no MFMAs, bit 1 = no DMA in the loop, bit 2 = no fragment reads
// PIPE = 0: two wave groups staggered by a barrier (L phase of one beside the M phase of the other).
// PIPE = 1: every wave software-pipelined -- while the 24 MFMAs of stage s run it issues the DMA of stage s + R - 1
//           and reads the fragments of stage s + 1 into a second register set; one barrier per stage.
template <int EPI, int R, bool C_PLANES, int ABL = 0, int PIPE = 1>
__global__ __launch_bounds__(kRingThreads, 2) void gemm_ring_nt(RingParams prm) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];     // [R][RNP][RNG][64] (+ transpose scratch)
    constexpr int SLOT = RNP * RNG * 64;                            // uint4 per ring slot
    constexpr int WAIT = (R - 2) * RPPW;                            // own pieces allowed outstanding when a stage must have landed

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                      // waves w and w + 4 share a SIMD
    const int wm = wave >> 1, wn = wave & 1;                        // 4 x 2 waves of 64 x 64

    // ---- this block's units: XCD x owns a contiguous run of whole tiles; its nblk/8 blocks split that run's k-stages evenly
    const int nk = prm.nk;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nb = prm.nblk >> 3;
    const int tq = prm.tiles >> 3, tr = prm.tiles & 7;
    const int tx0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int txn = tq + (xcd < tr ? 1 : 0);
    const int64_t ux = (int64_t)txn * nk;                           // units of this XCD
    const int64_t ubase = (int64_t)tx0 * nk;
    // (cut points are even unit indices: the software-pipelined loop runs two stages per iteration; nk is even)
    const int64_t u0 = ubase + ((ux / 2) * jb / nb) * 2, u1 = ubase + ((ux / 2) * (jb + 1) / nb) * 2;
    const int nst = (int)(u1 - u0);                                 // stages this block streams
    if (nst <= 0) return;                                           // (whole block: uniform)

    // ---- DMA issue stream ----
    const uint16_t* pp[RPPW];                                       // this lane's source of each of its pieces, current stage
    const int64_t a_step = 2 * prm.a_rows * 8, w_step = 2 * prm.w_rows * 8;
    int iss = 0;                                                    // next stage (0-based within the block) to issue
    const int64_t a_plane = prm.a_rows * (int64_t)prm.K, w_plane = prm.w_rows * (int64_t)prm.K;
    int iss_tile = (int)(u0 / nk), iss_ks = (int)(u0 % nk);        // unit the next issued stage belongs to
    auto setup_pieces = [&](int tile, int ks) {                     // point the pieces at unit (tile, ks)
        const int bn = tile % prm.tiles_n, bm = tile / prm.tiles_n;
        const int i32 = launder_v(lane) & 31, kg = launder_v(lane) >> 5;
#pragma unroll
        for (int t = 0; t < RPPW; ++t) {
            const int j = (wave + 8 * t) % RPIECES, p = j / RNG, g = j % RNG;
            if (g < RGA) {
                int row = bm * RBM + 32 * g + i32; if (row > prm.M - 1) row = prm.M - 1;
                pp[t] = prm.Ap + p * a_plane + ((int64_t)(2 * ks + kg) * prm.a_rows + row) * 8;
            } else {
                int row = bn * RBN + 32 * (g - RGA) + i32; if (row > prm.N - 1) row = prm.N - 1;
                pp[t] = prm.Wp + p * w_plane + ((int64_t)(2 * ks + kg) * prm.w_rows + row) * 8;
            }
        }
    };
    setup_pieces(iss_tile, iss_ks);
    auto issue = [&]() {                                            // DMA of stage `iss` into ring slot iss % R (always PPW instructions)
        const int slot = iss % R;
#pragma unroll
        for (int t = 0; t < RPPW; ++t) {
            const int j = (wave + 8 * t) % RPIECES, p = j / RNG, g = j % RNG;
            __builtin_amdgcn_global_load_lds((glb_void_t*)pp[t], (lds_void_t*)&lds[slot * SLOT + (p * RNG + g) * 64], 16, 0, 0);
        }
        if (iss + 1 < nst) {                                        // past the end: re-issue the last stage (harmless duplicates)
            if (++iss_ks == nk) { iss_ks = 0; ++iss_tile; setup_pieces(iss_tile, 0); }
            else {
#pragma unroll
                for (int t = 0; t < RPPW; ++t) pp[t] += ((wave + 8 * t) % RPIECES) % RNG < RGA ? a_step : w_step;
            }
        }
        ++iss;
    };

    f32x16 acc[2][2];
    struct Frags { bf16x8_t a[2][RNP], b[2][RNP]; };
    Frags F0, F1;
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    auto read_frags = [&](Frags& F, int slot) {
        const uint4* base = lds + slot * SLOT + lane;
#pragma unroll
        for (int p = 0; p < RNP; ++p) {
#pragma unroll
            for (int a = 0; a < 2; ++a) F.a[a][p] = __builtin_bit_cast(bf16x8_t, base[(p * RNG + 2 * wm + a) * 64]);
#pragma unroll
            for (int b = 0; b < 2; ++b) F.b[b][p] = __builtin_bit_cast(bf16x8_t, base[(p * RNG + RGA + 2 * wn + b) * 64]);
        }
    };
    auto mfmas = [&](const Frags& F) {
        constexpr int PAIRS[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};   // smallest products first
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[a][PAIRS[pr][0]], F.b[b][PAIRS[pr][1]], acc[a][b], 0, 0, 0);
    };

    // partial accumulators in fragment order: [wave][a][b][r][lane] (coalesced 256-B runs)
    float* my_part = prm.partials + (size_t)blockIdx.x * 2 * (RBM * RBN);
    auto store_partial = [&](float* dst) {
        float* d = launder_v(dst + (size_t)wave * (4 * 16 * 64) + lane);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) d[((a * 2 + b) * 16 + r) * 64] = acc[a][b][r];
    };
    auto add_partial = [&](const float* src) {
        const float* d = launder_v(src + (size_t)wave * (4 * 16 * 64) + lane);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += d[((a * 2 + b) * 16 + r) * 64];
    };
    auto epilogue = [&](int tile) {
        const int bn = tile % prm.tiles_n, bm = tile / prm.tiles_n;
        const int ln = launder_v(lane);                             // (see launder_v: keep the epilogue's address math out of the loop)
        Epilogue e = prm.epi;
        e.bias = launder_s(e.bias);
        e.residual = launder_s(e.residual);
        float* Cc = launder_s(prm.C);
        if (C_PLANES) {
            float* scratch = reinterpret_cast<float*>(lds + R * SLOT) + wave * kTrFloats;
            store_tile_planes<EPI, 2>(acc, reinterpret_cast<uint16_t*>(Cc), prm.M, prm.N, bm * RBM, bn * RBN, wm, wn, ln, e, scratch);
        } else {
            store_tile<EPI, 2, RBM>(acc, Cc, prm.ldc, prm.M, prm.N, bm * RBM, bn * RBN, wm, wn, ln, e);
        }
    };
    // publish this wave's share of a contributed partial: stores -> vmcnt(0) -> agent release -> counter
    auto publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&prm.flags[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto wait_block = [&](int blk) {                                // all 8 waves of `blk` have published
        if (lane == 0) {
            while (__hip_atomic_load(&prm.flags[blk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8) __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };

    // what this block does with its first and last tile
    const int first_tile = (int)(u0 / nk);
    const bool first_has_head = (u0 % nk) == 0;
    bool deferred = false;                                          // own tail partial parked in my_part slot 1

    // ---- prologue: R - 1 stages in flight, stage 0 visible ----
#pragma unroll
    for (int s = 0; s < R - 1; ++s) issue();
    waitcnt_vm(WAIT);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (PIPE == 0 && grp == 1) {                                    // stagger: group 1 runs one slot behind group 0
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    zero_acc();
    int slot = 0;                                                   // ring slot of the stage being consumed
    if (PIPE == 1) { read_frags(F0, 0); slot = 1 % R; }             // `slot` = ring slot of the NEXT stage
    if (ABL & 4) { read_frags(F0, 0); read_frags(F1, 0); }          // ablation without fragment reads: MFMAs still see real data

    auto segment_end = [&](int tile, int ks0, int seg_n) {
        const bool has_head = ks0 == 0, has_tail = ks0 + seg_n == nk;
        if (has_head && has_tail) {
            epilogue(tile);
        } else if (has_tail) {                                      // owner of a cut tile: park the tail, finish after the stream
            store_partial(my_part + RBM * RBN);
            deferred = true;
        } else {                                                    // head or middle segment: contribute
            store_partial(my_part);
            publish();
        }
        zero_acc();
    };

    int cur_tile = first_tile, ks0 = (int)(u0 % nk);
    for (int s = 0; s < nst;) {
        // one segment: the k-stages of `cur_tile` this block holds, [ks0, ks0 + seg_n)
        const int seg_n = (nk - ks0) < (nst - s) ? (nk - ks0) : (nst - s);
        if (PIPE == 1) {
            constexpr int WAIT1 = (R - 3) * RPPW;                   // stages s+2 .. s+R-2 may still be in flight
#define AC_RING_STEP(FC, FN)                                                                                      \
            do {                                                                                                  \
                waitcnt_vm(WAIT1);                                /* this wave's pieces of stage s + 1 landed */  \
                __builtin_amdgcn_sched_barrier(0);                                                                \
                __builtin_amdgcn_s_barrier();                     /* ... everyone's; reads of stage s - 1 done */ \
                __builtin_amdgcn_sched_barrier(0);                                                                \
                if (!(ABL & 2)) issue();                          /* stage s + R - 1 -> slot of stage s - 1 */    \
                if (!(ABL & 4)) read_frags(FN, slot);             /* fragments of stage s + 1 */                  \
                slot = slot + 1 == R ? 0 : slot + 1;                                                              \
                if (!(ABL & 1)) mfmas(FC);                        /* stage s */                                    \
                if (ABL & 16) {                                   /* pin the interleave: 2 MFMAs, 1 fragment read */ \
                    _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                           \
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                        \
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
                    }                                                                                             \
                }                                                                                                 \
                if (ABL & 8) asm volatile("" :: "v"(FN.a[0][0]), "v"(FN.a[1][1]), "v"(FN.b[0][2]), "v"(FN.b[1][0])); \
            } while (0)
            for (int q = 0; q < seg_n; q += 2) {                    // (segments are even: nk is even and blocks are cut at even units)
                AC_RING_STEP(F0, F1);
                AC_RING_STEP(F1, F0);
            }
#undef AC_RING_STEP
        } else {
            for (int q = 0; q < seg_n; ++q) {
                // ---------------- L phase: next stage's DMA, this stage's fragments ----------------
                if (!(ABL & 2)) issue();                            // stage + R - 1 -> the slot the previous stage used
                if (!(ABL & 4)) read_frags(F0, slot);
                slot = slot + 1 == R ? 0 : slot + 1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (grp == 1) waitcnt_vm(WAIT);                     // every wave's pieces of the NEXT stage have landed ...
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                       // ... and are visible to group 0's next L phase
                __builtin_amdgcn_sched_barrier(0);
                // ---------------- M phase: 24 MFMAs ----------------
                if (!(ABL & 1)) mfmas(F0);
                if (grp == 0) waitcnt_vm(WAIT);                     // this wave's pieces of the next stage have landed
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        s += seg_n;
        segment_end(cur_tile, ks0, seg_n);
        ks0 = 0;
        ++cur_tile;
    }
    if (PIPE == 0 && grp == 0) {                                    // group 1 executed one more barrier at the start
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // drain the over-issued tail stages before LDS is released

    if (deferred) {
        // contributions of the blocks (same XCD, lower ids) that hold the rest of first_tile, lowest k first, then our tail
        const int64_t tstart = (int64_t)first_tile * nk;            // first unit of the tile
        zero_acc();
        for (int j = 0; j < jb; ++j) {
            const int64_t ju0 = ubase + ((ux / 2) * j / nb) * 2, ju1 = ubase + ((ux / 2) * (j + 1) / nb) * 2;
            if (ju1 <= tstart || ju0 >= u0 || ju1 <= ju0) continue;  // no unit of this tile (the tile lies in [tstart, u0) before us)
            const int blk = 8 * j + xcd;
            wait_block(blk);
            add_partial(prm.partials + (size_t)blk * 2 * (RBM * RBN));
        }
        add_partial(my_part + RBM * RBN);
        epilogue(first_tile);
    }
}

struct RingWs { float* partials; int* flags; size_t bytes; int nblk; };
RingWs g_ring_ws = {nullptr, nullptr, 0, 0};      // grown on demand, per process (one device per process)

}  // namespace

namespace ac {

// true when launch_gemm should hand (M, N, K) with pre-split A and W to the ring kernel
bool ring_takes(int M, int N, int K, int cls, bool c_planes, bool allow_cuts) {
    if (M < 2048 || N < RBN || (N % 8) != 0 || (K % (2 * RSBK)) != 0 || K < 4 * RSBK) return false;   // nk even
    if (cls != EPI_BIAS && cls != EPI_BIAS_GELU && cls != EPI_BIAS_RES) return false;
    if (c_planes && cls == EPI_BIAS_RES) return false;
    if (!allow_cuts) {                       // default dispatch: only when the tiles divide evenly over the CUs
        const int tiles = ((N + RBN - 1) / RBN) * ((M + RBM - 1) / RBM), nblk = ac::dev_info().cus / 8 * 8;
        if (nblk < 8 || tiles % nblk != 0) return false;
    }
    return true;
}

int launch_gemm_ring(const uint16_t* Ap, int64_t a_rows, const uint16_t* Wp, int64_t w_rows, float* C, int64_t ldc,
                     uint16_t* Cp, int M, int N, int K, int cls, const acg::Epilogue& epi, hipStream_t stream) {
    RingParams p;
    p.Ap = Ap; p.a_rows = a_rows; p.Wp = Wp; p.w_rows = w_rows;
    p.C = Cp ? reinterpret_cast<float*>(Cp) : C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.tiles_n = (N + RBN - 1) / RBN;
    p.tiles = p.tiles_n * ((M + RBM - 1) / RBM);
    p.nk = K / RSBK;
    int nblk = ac::dev_info().cus / 8 * 8;
    if (nblk > p.tiles * 8) nblk = ((p.tiles + 7) / 8) * 8;       // never fewer than ~1/8 tile per block
    if (nblk < 8) nblk = 8;
    p.nblk = nblk;
    const size_t need = (size_t)nblk * 2 * RBM * RBN * sizeof(float) + (size_t)nblk * sizeof(int) + 256;
    if (g_ring_ws.bytes < need) {
        if (g_ring_ws.partials) (void)hipFree(g_ring_ws.partials);
        void* mem = nullptr;
        AC_HIP_CHECK(hipMalloc(&mem, need));
        g_ring_ws.partials = (float*)mem; g_ring_ws.bytes = need;
    }
    p.partials = g_ring_ws.partials;
    p.flags = reinterpret_cast<int*>(reinterpret_cast<char*>(g_ring_ws.partials) + (size_t)nblk * 2 * RBM * RBN * sizeof(float));
    p.epi = epi;
    if (p.tiles % nblk != 0)          // whole tiles per block: no partial hand-offs, the flags are never read
        AC_HIP_CHECK(hipMemsetAsync(p.flags, 0, (size_t)nblk * sizeof(int), stream));
    const dim3 grid(nblk), block(kRingThreads);
#define AC_RING(E, RR, CP)                                                                                         \
    do {                                                                                                          \
        const size_t lds = (size_t)RR * RNP * RNG * 64 * 16 + (CP ? 8 * kTrFloats * sizeof(float) : 0);           \
        (void)hipFuncSetAttribute((const void*)gemm_ring_nt<E, RR, CP, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_ring_nt<E, RR, CP, 16>), grid, block, lds, stream, p);                           \
    } while (0)
    static const int abl = getenv("AC_RING_ABLATE") ? atoi(getenv("AC_RING_ABLATE")) : 0;
    if ((abl || getenv("AC_RING_PIPE")) && !Cp) {
        const size_t lds = (size_t)4 * RNP * RNG * 64 * 16;
        static const int pipe = getenv("AC_RING_PIPE") ? atoi(getenv("AC_RING_PIPE")) : 1;
#define AC_ABL(A) do { if (pipe) { (void)hipFuncSetAttribute((const void*)gemm_ring_nt<EPI_BIAS, 4, false, A, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                       hipLaunchKernelGGL((gemm_ring_nt<EPI_BIAS, 4, false, A, 1>), grid, block, lds, stream, p); } else {                                              \
                       (void)hipFuncSetAttribute((const void*)gemm_ring_nt<EPI_BIAS, 4, false, A, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                       hipLaunchKernelGGL((gemm_ring_nt<EPI_BIAS, 4, false, A, 0>), grid, block, lds, stream, p); } } while (0)
        switch (abl) { case 0: AC_ABL(0); break; case 1: AC_ABL(1); break; case 2: AC_ABL(2); break; case 3: AC_ABL(3); break; case 4: AC_ABL(4); break;
                       case 5: AC_ABL(5); break; case 6: AC_ABL(6); break; case 9: AC_ABL(9); break; case 11: AC_ABL(11); break; case 16: AC_ABL(16); break;
                       default: AC_ABL(7); break; }
#undef AC_ABL
        AC_LAUNCH_CHECK();
        return AC_OK;
    }
    // (ABL = 16 is not an ablation: it pins the 2-MFMA / 1-fragment-read interleave, the best measured schedule)
    if (Cp) {
        if (cls == EPI_BIAS_GELU) AC_RING(EPI_BIAS_GELU, 3, true);
        else AC_RING(EPI_BIAS, 3, true);
    } else if (cls == EPI_BIAS) AC_RING(EPI_BIAS, 4, false);
    else if (cls == EPI_BIAS_GELU) AC_RING(EPI_BIAS_GELU, 4, false);
    else AC_RING(EPI_BIAS_RES, 4, false);
#undef AC_RING
    AC_LAUNCH_CHECK();
    return AC_OK;
}

}  // namespace ac
