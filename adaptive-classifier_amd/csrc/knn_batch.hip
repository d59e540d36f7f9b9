// Batched kNN proposals on the matrix pipe: the compute-bound regime of faiss.IndexFlatL2.search
// (/root/reference/src/adaptive_classifier/memory.py:113-114) -- many queries (predict_batch, BASELINE configs[2]:
// 4096 x 10M, configs[4]: 1024 x 2M) -- where the fp32-MFMA sweep of knn_l2.hip is bound by its 157 TFLOP/s pipe.
//
// Exactness does not come from the sweep: knn_merge_rerank recomputes the proposed candidates in fp64 and a
// certificate proves no other row can enter the top-k (knn_l2.hip).  The sweep only has to PROPOSE with a bounded
// error, so it runs on ONE fp16 plane per operand and ONE v_mfma_f32_32x32x16_f16 per tile and 16 k (round 2 used two
// bf16 planes and three products: 3x the matrix work and 2x the bytes for a bound only 1.75x tighter):
//   * operands are scaled by powers of two into fp16's normal range -- the store by 2^-e_p with 2^e_p > max |p| (one
//     exponent per store, from the maximum row norm), every query by its own 2^-e_q -- and rounded to nearest fp16:
//     |x^ - h| <= 2^-11 |x^| + 2^-25.  The store's plane and its row norms |p|^2 are prepared once per store
//     (ac_knn_prepare_store), the queries' plane per call;
//   * v = |p|^2 + f_q (h_p . h_q), f_q = -2 2^(e_p + e_q): products of fp16 values are exact in the fp32 accumulator, so
//     |v - exact| <= E = gamma (max|p| + |q|)^2 with gamma = 1.01 (2^-11 + (K + 18 + sqrt K) 2^-24): input rounding, the
//     subnormal tail, fp32 accumulation at 2 ulp per term (ac_knn_l2_topk_batch states the derivation);
//   * it is a GEMM: 256 store rows x 256 queries per 8-wave workgroup (waves of 64 x 128), 32-k stages in a ring of four
//     LDS slots filled by global_load_lds from the operand planes (the store's tile-major: a 256-row tile is one contiguous run,
//     k-slot-major inside; the queries' [k-slot][row][8]; the loop of gemm_pipe.hip: counted
//     vmcnt, raw barrier, fragment reads of stage s + 1 pinned under the MFMAs of stage s), ONE persistent workgroup per
//     CU that walks its row tiles without draining the ring;
//   * workgroup -> (query tile, row group): an XCD holds b <= 4 query tiles (their plane, <= 2 MB, stays in its L2) times
//     32 / b row groups; XCDs that share a query-tile set split the row groups.  A row tile is therefore fetched by
//     ceil(nqt / 4) XCDs and multiplied against b query tiles out of that XCD's L2.
// Selection without per-block running lists: an exact top-k' search over a strided SAMPLE of the store (the
// ordinary fp32 path, ~1.5 % of the rows) gives each query tau_q = its k'-th smallest sample distance -- a valid
// upper bound of the k'-th smallest distance over the whole store.  The GEMM's epilogue keeps every (row, v) with
// v < tau_q - |q|^2 + E (at least k' rows, a few thousand expected) in a per-query candidate buffer; the merge
// kernel radix-selects the k' best by v, re-ranks them in fp64 and certifies exactly as for the sweep (rows that
// were filtered out have v >= the k'-th kept v).  A candidate buffer that overflows sends its query to the exact
// fp64 fallback.
#include "common.h"
#include "gemm_common.h"
#include "grid_sync.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

namespace {
using namespace acg;

constexpr int BBM = 256, BBN = 256;          // store rows x queries per workgroup tile
constexpr int BGA = BBM / 32, BGW = BBN / 32, BRG = BGA + BGW;
constexpr int BSBK = 32;                      // k per stage: two MFMA chunks of 16
constexpr int BSLOT = 2 * BRG * 64;           // uint4 per ring slot: [chunk][group][lane] = 32 KB

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;                   // round to nearest even
    return (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
}

// 2^e with 2^e > x >= 0 (e = 0 for x = 0): the scale that brings a vector of norm x into the unit ball
__device__ __forceinline__ int unit_exponent(double x) { return x > 0.0 ? ilogb(x) + 1 : 0; }

// ---- store preparation ---------------------------------------------------------------------------------------------
// pass 1: |p|^2 per row (fp64 accumulate, rounded once) and the maximum
__global__ __launch_bounds__(256) void knn_norms_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int D,
                                                        float* __restrict__ norms, uint32_t* __restrict__ maxnorm_bits,
                                                        int64_t row_begin) {
    const int lane = threadIdx.x & 63;
    const int64_t row = row_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    double nn = 0.0;
    for (int c = lane; c < D; c += 64) { const double v = X[row * ldx + c]; nn = fma(v, v, nn); }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) nn += __shfl_xor(nn, o);
    if (lane == 0) {
        const float f = (float)nn;
        norms[row] = f;
        atomicMax(maxnorm_bits, __float_as_uint(f));                     // non-negative floats order like their bits
    }
}

// pass 2: fp16(x * 2^-e) in runs of 8 k: e from the maximum row norm (store; tile-major layout, see the store below) or from the
// row's own norm (queries: plane[k/8][row][8]; they also get their threshold and their epilogue factor here)
//   thr[q]  = (tau_q - |q|^2 + E_q) rounded up to fp32, tau_q = exact k'-th smallest sample distance (fp64)
//   qfac[q] = -2 2^(e_p + e_q)
__global__ __launch_bounds__(256) void knn_plane_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int64_t rows_pad,
                                                        int D, int Kp, const uint32_t* __restrict__ maxnorm_bits, int per_row_scale,
                                                        uint16_t* __restrict__ plane, const double* __restrict__ sampleD, int kp,
                                                        double gamma, float* __restrict__ thr, float* __restrict__ qfac,
                                                        float* __restrict__ pad_norms, int tile_major,
                                                        int32_t* __restrict__ zero_a, int64_t zero_na, int64_t row_begin) {
    // (query form) the counters the following launches append to start at zero: done here instead of by a memset launch
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < zero_na; i += (int64_t)gridDim.x * 256) zero_a[i] = 0;
    const int lane = threadIdx.x & 63;
    const int64_t row = row_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // (row_begin: incremental store updates)
    if (row >= rows_pad) return;
    const double pmax = sqrt((double)__uint_as_float(*maxnorm_bits) * 1.001);
    const int ep = unit_exponent(pmax);
    int e = ep;
    if (per_row_scale) {
        double a = 0.0;
        if (row < rows) for (int c = lane; c < D; c += 64) { const double v = X[row * ldx + c]; a = fma(v, v, a); }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o);
        const double qn = sqrt(a);
        e = unit_exponent(qn * 1.0000001);
        if (lane == 0) {
            if (row < rows) {
                const double E = gamma * (pmax + qn) * (pmax + qn) + 1e-30;
                const double tau = sampleD ? sampleD[(size_t)row * kp + kp - 1] : (double)INFINITY;
                float t = (float)(tau - a + E);
                if ((double)t < tau - a + E) t = nextafterf(t, INFINITY);
                thr[row] = isfinite(tau) ? t : INFINITY;                 // (no sample yet / sample smaller than k': keep everything)
                qfac[row] = -ldexpf(2.0f, ep + e);
            } else { thr[row] = -INFINITY; qfac[row] = 0.f; }            // padding queries keep nothing
        }
    }
    if (!per_row_scale && pad_norms && row >= rows && lane == 0) pad_norms[row] = INFINITY;   // tile padding never qualifies
    const float scale = ldexpf(1.0f, -e);                                // power of two: exact
    const int nslot = Kp >> 3;
    for (int q = lane; q < nslot; q += 64) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int c = 8 * q + t;
            v[t] = (row < rows && c < D) ? X[row * ldx + c] * scale : 0.f;
        }
        uint4 H;
        H.x = pack_f16(v[0], v[1]); H.y = pack_f16(v[2], v[3]); H.z = pack_f16(v[4], v[5]); H.w = pack_f16(v[6], v[7]);
        // queries: plane[k-slot][row][8].  store (tile_major): plane[row / 256][k-slot][row % 256][8] -- a 256-row tile is ONE
        // contiguous run of 256 * Kp * 2 bytes (k-slot-major inside), so a sweep streams the store front to back
        const int64_t at = tile_major ? (((row >> 8) * nslot + q) << 8) + (row & 255) : (int64_t)q * rows_pad + row;
        *reinterpret_cast<uint4*>(plane + at * 8) = H;
    }
}

struct BatchParams {
    const uint16_t* Pp; int64_t p_rows;      // store plane [p_rows / 256][Kp/8][256][8] fp16 (tile-major)
    const float* pnorm;                      // [p_rows]: |p|^2, +inf past N
    const uint16_t* Qp; int64_t q_rows;      // query plane, q_rows = round_up(nq, 256)
    const float* thr;                        // [q_rows]
    const float* qfac;                       // [q_rows]
    int64_t N;                               // LOGICAL rows swept: logical row i is store row (i >> 3) * 8 * row_stride + (i & 7)
    int64_t row_stride;                      // 1 = the whole store; > 1 = a sample (threshold stages): runs of 8 consecutive rows --
                                             // whole 128-byte lines of every k-slot of the plane -- every 8 * row_stride rows
    int best_only;                           // unfiltered sample stage: every lane offers only the best of its 32 rows per query column
    int Kp;
    int nqt;                                 // query tiles of 256 in this launch
    int b;                                   // query tiles per XCD (1, 2, 4)
    int sets;                                // query-tile sets (power of two <= 8): XCD x works on set x % sets
    int64_t ntiles;                          // row tiles of 256
    float* cand_d; int32_t* cand_i; int32_t* cand_cnt; int cap;
    int segs, segcap;                        // a query's list = segs segments of segcap = cap / segs entries, one counter each
    int32_t* clear_ctr;                      // (main sweep) [64] fallback slot counter and [4] caller's d_stats, zeroed by workgroup 0 here
    int32_t* clear_stats;                    //   instead of by two memset launches in front of the merge; NULL = leave alone
    // two_phase (round 6; one query tile, every workgroup owns at least one row tile, the whole grid resident): the thresholds come
    // from THIS launch -- after its first row tile every workgroup publishes, per query, the smallest sweep value of its 256 rows;
    // a grid barrier; workgroup g takes the k'-th smallest of the G minima of query g (G real rows' values: at least k' rows of
    // the store pass) one ulp up as thr[g]; a second barrier; then the first tile is filtered like every later one.  No sample
    // sweep, no threshold merge: two launches and ~40 us less for a 256 x 100k call.
    int two_phase;
    int kp, nq_real;
    float* thr_rw;                           // = thr
    unsigned* wgmin;                         // [gridDim.x][256] monotone keys of the per-workgroup minima
    acp::GridCtl* ctl;                       // zeroed before the launch
};

typedef const BatchParams __attribute__((address_space(4)))* KArgs;

template <int N> __device__ __forceinline__ void bwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// C/D layout of v_mfma_f32_32x32x16_f16: lane owns column (lane & 31); acc_row32(r, lane) gives its 16 rows.
//   NWV = 8: 4 x 2 waves of 64 x 128 (two per SIMD, <= 256 registers each) -- the shipped form.  Its fragment reads + DMA
//            writes need ~87 % of the LDS bandwidth at full matrix rate: the LDS, not the matrix pipe, paces this loop
//            (matrix pipe busy 49 % of the SIMD cycles, profiles/r03/knn_batch_sweep_pmc.json).
//   NWV = 4: 2 x 2 waves of 128 x 128, ONE per SIMD with the 512-register budget (a third fewer fragment bytes per MFMA):
//            builds, exact, 1.5x slower under hipcc's schedule -- kept as a template argument, not instantiated.
// BURST: the form for short launches -- sample stages (row_stride > 1, best_only) and sweeps of small stores (segmented
// lists); BURST = false is the plain whole-store sweep with one list per query that long launches run.
// (Measured and dropped in round 6, profiles/r06/knn_batch_nt_pmc_rejected.json, knn_batch_qplane_pmc_rejected.json: the store
//  plane's DMA with the non-temporal bit -- the workgroups that share a row tile then each fetch it themselves, 159 / 130 GB of
//  fabric reads at 4096 x 10M against 75 / 80 -- and a tile-major QUERY plane, 106 / 90 GB.)
template <int BNS, int NWV, bool BURST, bool TWO_PHASE = false>        // ring depth, waves; TWO_PHASE: BatchParams::two_phase (its own instantiation)
__global__ __launch_bounds__(64 * NWV, NWV / 4) void knn_batch_sweep(BatchParams prm) {
    constexpr int WMW = NWV / 2, TM = BBM / (32 * WMW), TN = 4;    // wave grid WMW x 2, wave tile (32 TM) x 128
    constexpr int GPW = BGA / NWV;                                  // store / query groups each wave stages
    constexpr int PPW = 4 * GPW;                                    // DMA pieces per wave and stage
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];    // [BNS][2 chunks][BRG groups][64 lanes]
    __shared__ unsigned tp_min[256], tp_vals[64 * NWV], tp_flag;   // (two_phase: per-query minima of this workgroup, the G minima of one query)
    auto tp_key = [](float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); };
    auto tp_key_inv = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); };
    const int tid = threadIdx.x, lane = tid & 63;
    if (BURST && TWO_PHASE && tid < 256) tp_min[tid] = 0xffffffffu;             // (visible long before its first use: the k-loop's barriers)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i32 = lane & 31, kg = lane >> 5;
    if (blockIdx.x == 0 && tid < 64) {                              // consumed by the merge / fallback kernels launched after this one
        KArgs ka0 = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
        int32_t* const cc = ka0->clear_ctr; int32_t* const cs = ka0->clear_stats;
        if (cc) cc[tid] = 0;
        if (cs && tid < 4) cs[tid] = 0;
    }
    // ---- workgroup -> (query tile, row group) ----
    const int per_xcd = (int)(gridDim.x >> 3), xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int xs = 8 / prm.sets, set = xcd % prm.sets, xr = xcd / prm.sets;
    const int qt = set * prm.b + j % prm.b;
    const int rgx = per_xcd / prm.b;                                // row groups per XCD
    const int g = xr * rgx + j / prm.b, G = xs * rgx;
    if (qt >= prm.nqt) return;                                      // (whole workgroup)
    const int ntiles = (int)prm.ntiles, nrows = (int)prm.N;       // (row and tile indices fit 32 bits: checked at launch)
    const int my_tiles = ntiles > g ? (ntiles - 1 - g) / G + 1 : 0;
    if (my_tiles == 0) return;
    const int nk = prm.Kp / BSBK;                                   // even (Kp % 64 == 0)

    // ---- DMA stream.  Wave w stages store groups w + NWV t and query groups w + NWV t, both chunks.
    //      A piece = 32 rows x 2 k-slots: lane (i32, kg) copies the 16 B of row i32, k-slot 4 s + 2 c + kg.
    //      Store plane (tile-major): k-slot s of row r sits at ((r / 256 * nslot + s) * 256 + r % 256) * 16 bytes.
    const int64_t a_step = 4 * 256 * 8, w_step = 4 * prm.q_rows * 8;
    const int64_t a_c1 = 2 * 256 * 8, w_c1 = 2 * prm.q_rows * 8;          // chunk 1 = two k-slots further
    const int nslot_p = prm.Kp >> 3;
    const uint16_t* pa[GPW];
    const uint16_t* pw[GPW];
    const int rs8i = 8 * (int)prm.row_stride;
    auto a_base = [&](int it) {
#pragma unroll
        for (int t = 0; t < GPW; ++t) {
            int row = (it * G + g) * BBM + 32 * (wave + NWV * t) + i32;
            if (row > nrows - 1) row = nrows - 1;
            const int srow = BURST ? (row >> 3) * rs8i + (row & 7) : row;     // store row (sample stages: runs of 8 rows)
            pa[t] = prm.Pp + ((((int64_t)(srow >> 8) * nslot_p + kg) << 8) + (srow & 255)) * 8;
        }
    };
    auto w_base = [&]() {
#pragma unroll
        for (int t = 0; t < GPW; ++t)
            pw[t] = prm.Qp + ((int64_t)kg * prm.q_rows + ((int64_t)qt * BBN + 32 * (wave + NWV * t) + i32)) * 8;
    };
    a_base(0); w_base();
    int st_it = 0, st_k = 0, st_slot = 0;                   // unit / ring slot the next issue() loads
    auto issue = [&]() {                                            // always PPW DMA instructions (exact vmcnt accounting)
        uint4* dst = lds + st_slot * BSLOT;
#pragma unroll
        for (int t = 0; t < GPW; ++t) {
            const int ga = wave + NWV * t;
            __builtin_amdgcn_global_load_lds((glb_void_t*)pa[t], (lds_void_t*)(dst + (0 * BRG + ga) * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)pw[t], (lds_void_t*)(dst + (0 * BRG + BGA + ga) * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(pa[t] + a_c1), (lds_void_t*)(dst + (1 * BRG + ga) * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(pw[t] + w_c1), (lds_void_t*)(dst + (1 * BRG + BGA + ga) * 64), 16, 0, 0);
        }
        st_slot = st_slot + 1 == BNS ? 0 : st_slot + 1;
        if (++st_k == nk) {
            st_k = 0;
            if (st_it + 1 < my_tiles) ++st_it;                      // past the end: the last tile again (never consumed)
            a_base(st_it); w_base();
        } else {
#pragma unroll
            for (int t = 0; t < GPW; ++t) { pa[t] += a_step; pw[t] += w_step; }
        }
    };

    // this lane's four query columns (their thresholds / epilogue factors are re-read per tile: registers are scarce)
    const int qcol0 = qt * BBN + wn * 128 + i32;                    // column ni = qcol0 + 32 ni

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    struct Frags { f16x8_t a[2][TM], b[2][TN]; };                  // [chunk][tile]
    auto read_frags = [&](Frags& F, int slot) {
        const uint4* base = lds + slot * BSLOT + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int a = 0; a < TM; ++a) F.a[c][a] = __builtin_bit_cast(f16x8_t, base[(c * BRG + TM * wm + a) * 64]);
#pragma unroll
            for (int b = 0; b < TN; ++b) F.b[c][b] = __builtin_bit_cast(f16x8_t, base[(c * BRG + BGA + TN * wn + b) * 64]);
        }
    };
    auto mfmas = [&](const Frags& F) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[c][a], F.b[c][b], acc[a][b], 0, 0, 0);
    };
    // tile done: v = |p|^2 + f_q acc; keep what beats the query's threshold.  Two passes, so that a wave waits for memory
    // ONCE per tile: (1) a bit mask of the qualifying elements per query column and their count -- no memory operations;
    // (2) one slot reservation (atomicAdd) per query column with candidates, issued back to back, then the stores.
    int it = 0;
    auto filter = [&]() {
        // (the filter's parameters are re-read from the kernel-argument segment here: kept live across the k-loop they exceed
        //  the scalar registers, and hipcc then parks them in vector lanes and pays ~40 v_readlane per stage to get them back)
        KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const float* const f_pnorm = ka->pnorm; const float* const f_thr = ka->thr; const float* const f_qfac = ka->qfac;
        float* const f_cand_d = ka->cand_d; int32_t* const f_cand_i = ka->cand_i; int32_t* const f_cand_cnt = ka->cand_cnt;
        const int f_cap = ka->cap, f_segs = ka->segs, f_segcap = ka->segcap, f_best_only = ka->best_only;
        const int row0 = (it * G + g) * BBM + wm * (32 * TM);
        // Appends reserve their slots with one atomic per lane and query column.  Device-scope atomics on ONE address from all
        // XCDs serialise at ~0.1 us each (they execute at the memory side): a single counter per query cost 84 of the 140 us
        // of a 256 x 100k sweep.  So a query's list is split into segments with a counter each, and this lane's (tile, wave
        // row, lane half) picks the segment -- consecutive 32-row blocks of the store go round the segments, which also keeps
        // a cluster of neighbouring rows from filling one of them.
        const int seg = BURST ? (((it * G + g) * (BBM / (32 * TM)) + wm) * 2 + kg) & (f_segs - 1) : 0;
        const int cidx0 = qcol0 * f_segs + seg, sbase = seg * f_segcap;       // (32-bit: q_rows * segs and cap are far below 2^31)
        // rows (r & 3) + 8 (r >> 2) + 4 kg of a 32-row tile: four runs of 4 norms (the array is padded to whole tiles with
        // +inf: rows past N never qualify)
        // (logical rows come in runs of 8 = one group of BatchParams::row_stride; a group past the end never qualifies --
        //  with row_stride == 1 the array is also padded with +inf to whole tiles)
        const int rs8 = 8 * (int)ka->row_stride, nr = (int)ka->N;
        auto load_pn = [&](int mi, float (&pn)[16]) {
            const int l0 = row0 + mi * 32;                          // multiple of 32
            if constexpr (!BURST) {                                 // the whole store: contiguous, padded with +inf -- nothing to check
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(f_pnorm + l0 + 8 * r4 + 4 * kg);
                    pn[4 * r4] = t.x; pn[4 * r4 + 1] = t.y; pn[4 * r4 + 2] = t.z; pn[4 * r4 + 3] = t.w;
                }
            } else {
                const float* base = f_pnorm + ((l0 >> 3) * rs8 + 4 * kg);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const bool in = l0 + 8 * r4 < nr;
                    const f32x4 t = *reinterpret_cast<const f32x4*>(in ? base + r4 * rs8 : f_pnorm);
                    pn[4 * r4] = in ? t.x : INFINITY; pn[4 * r4 + 1] = in ? t.y : INFINITY;
                    pn[4 * r4 + 2] = in ? t.z : INFINITY; pn[4 * r4 + 3] = in ? t.w : INFINITY;
                }
            }
        };
        float thr[4], qf[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) qf[ni] = f_qfac[qcol0 + 32 * ni];
        if constexpr (BURST && TWO_PHASE) {
            if (it == 0) {
                // (1) this lane's smallest value per query column over its 32 * TM rows -> the workgroup's, through LDS
                float best[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) best[ni] = INFINITY;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    float pn[16];
                    load_pn(mi, pn);
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) best[ni] = fminf(best[ni], fmaf(acc[mi][ni][r], qf[ni], pn[r]));
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) atomicMin(&tp_min[wn * 128 + i32 + 32 * ni], tp_key(best[ni]));
                __syncthreads();
                // (the thread id rebuilt from what the loop keeps live, the block id laundered: nothing of this once-per-launch
                //  block may hold a register -- or a spill slot -- across the k-loop)
                const int t = (((wm << 1) | wn) << 6) | (kg << 5) | i32;
                unsigned bx = blockIdx.x;
                asm volatile("" : "+s"(bx));
                unsigned* const wgmin = ka->wgmin;
                acp::GridCtl* const ctl = ka->ctl;
                const unsigned Gall = gridDim.x;
                if (t < 256) __hip_atomic_store(wgmin + (size_t)bx * 256 + t, tp_min[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = acp::grid_barrier<false>(ctl, 1u, Gall, &tp_flag);
                // (2) workgroup b: queries b, b + G, ...: the k'-th smallest of the G minima
                const int kp = ka->kp, nq_real = ka->nq_real;
                float* const thr_rw = ka->thr_rw;
                for (int q = (int)bx; q < 256; q += (int)Gall) {
                    __syncthreads();
                    if (t < (int)Gall) tp_vals[t] = __hip_atomic_load(wgmin + (size_t)t * 256 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    if (t < (int)Gall && q < nq_real) {
                        const unsigned my = tp_vals[t];
                        int rank = 0;
                        for (int j = 0; j < (int)Gall; ++j) { const unsigned v = tp_vals[j]; rank += (v < my || (v == my && j < t)) ? 1 : 0; }
                        if (rank == kp - 1) acp::st_sc1(thr_rw + q, ok ? nextafterf(tp_key_inv(my), INFINITY) : INFINITY);
                    }
                }
                ok = acp::grid_barrier<false>(ctl, 2u, Gall, &tp_flag) && ok;
                (void)ok;                                            // (a barrier that gave up leaves +inf thresholds: overflow -> exact fallback)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the ring's DMA accounting below only ever waits for FEWER operations)
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) thr[ni] = acp::ld_sc1(f_thr + qcol0 + 32 * ni);       // (written inside this launch)
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) thr[ni] = f_thr[qcol0 + 32 * ni];
        }
        if (BURST && !TWO_PHASE && f_best_only) {                    // (a two-phase launch is never a sample stage)
            // The first threshold stage has no threshold yet.  Any k' rows bound the k'-th smallest distance from above, so
            // instead of offering all of the sample (a million appends from the few workgroups a 4096-row sample occupies)
            // every lane offers the best of the 32 * TM rows it holds per query column: 8 offers per (query, 256-row tile).
            float best[4]; int bidx[4], slot[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) { best[ni] = INFINITY; bidx[ni] = 0; }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                float pn[16];
                load_pn(mi, pn);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float val = fmaf(acc[mi][ni][r], qf[ni], pn[r]);
                        if (val < best[ni]) { best[ni] = val; bidx[ni] = 16 * mi + r; }
                    }
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)                          // the four reservations in flight together
                slot[ni] = best[ni] < thr[ni] ? atomicAdd(&f_cand_cnt[cidx0 + 32 * ni * f_segs], 1) : f_segcap;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                if (slot[ni] < f_segcap) {
                    const size_t e = (size_t)(qcol0 + 32 * ni) * f_cap + (unsigned)(sbase + slot[ni]);
                    f_cand_d[e] = best[ni];
                    f_cand_i[e] = (int32_t)(row0 + (bidx[ni] >> 4) * 32 + acc_row32(bidx[ni] & 15, lane));
                }
            return;
        }
        unsigned mask[TM / 2][4];                                   // bit 16 (mi & 1) + r of query column ni, row tiles 2 h, 2 h + 1
        unsigned any = 0;
#pragma unroll
        for (int h = 0; h < TM / 2; ++h) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) mask[h][ni] = 0u;
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                float pn[16];
                load_pn(2 * h + m2, pn);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        mask[h][ni] |= (fmaf(acc[2 * h + m2][ni][r], qf[ni], pn[r]) < thr[ni] ? 1u : 0u) << (16 * m2 + r);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) any |= mask[h][ni];
        }
        if (__builtin_amdgcn_ballot_w64(any != 0) == 0) return;    // (whole wave: nothing kept)
        int base[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            int c = 0;
#pragma unroll
            for (int h = 0; h < TM / 2; ++h) c += __builtin_popcount(mask[h][ni]);
            base[ni] = c ? atomicAdd(&f_cand_cnt[cidx0 + 32 * ni * f_segs], c) : 0;
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            unsigned here = 0;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) here |= mask[mi / 2][ni] >> (16 * (mi & 1)) & 0xffffu;
            if (here == 0) continue;
            float pn[16];
            load_pn(mi, pn);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((mask[mi / 2][ni] >> (16 * (mi & 1) + r)) & 1u) {
                        const int slot = base[ni]++;
                        if (slot < f_segcap) {
                            const size_t e = (size_t)(qcol0 + 32 * ni) * f_cap + (unsigned)(sbase + slot);
                            f_cand_d[e] = fmaf(acc[mi][ni][r], qf[ni], pn[r]);
                            f_cand_i[e] = (int32_t)(row0 + mi * 32 + acc_row32(r, lane));
                        }
                    }
        }
    };

    // ---- prologue: BNS - 1 stages in flight, stage 0 landed and visible ----
    zero_acc();
#pragma unroll
    for (int s = 0; s < BNS - 1; ++s) issue();
    bwait_vm<(BNS - 2) * PPW>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Frags F0, F1;
    int slot = 0;                                                   // ring slot of the next stage to read
    // One step = one 32-k stage u: this wave's DMA pieces of stage u + 1 have landed -> barrier (everyone's have; the MFMAs
    // of stage u - 1 are issued) -> DMA of stage u + BNS - 1 into the slot stage u - 1 used -> fragments of stage u + 1 ->
    // the MFMAs of stage u with the fragment reads pinned between them.  The LAST stage of a row tile reads no
    // fragments: the filter runs with only the accumulators live (a spill reload pending at the loop's back edge made hipcc
    // wait vmcnt(0) -- the whole DMA ring -- inside the loop), and the next tile's first fragments are read after it.
    constexpr int NREAD = 2 * (TM + TN), NMFMA = 2 * TM * TN, MPR = NMFMA / NREAD;
#define AC_KNN_STEP(FC, FN, READ_NEXT)                                                                           \
    do {                                                                                                         \
        bwait_vm<(BNS - 3) * PPW>();                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_barrier();                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        issue();                                                                                                 \
        if (READ_NEXT) { read_frags(FN, slot); slot = slot + 1 == BNS ? 0 : slot + 1; }                          \
        mfmas(FC);                                                                                               \
        if (READ_NEXT) {                                                                                         \
            _Pragma("unroll") for (int g_ = 0; g_ < NREAD; ++g_) {                                               \
                __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);                                             \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
            }                                                                                                    \
        }                                                                                                        \
    } while (0)
    for (; it < my_tiles; ++it) {
        read_frags(F0, slot);                                       // stage 0 of this tile (landed: waited for one step ago)
        slot = slot + 1 == BNS ? 0 : slot + 1;
        for (int kt = 0; kt < nk - 2; kt += 2) {                    // (nk is even)
            AC_KNN_STEP(F0, F1, true);
            AC_KNN_STEP(F1, F0, true);
        }
        AC_KNN_STEP(F0, F1, true);
        AC_KNN_STEP(F1, F0, false);
        __builtin_amdgcn_sched_barrier(0);
        filter();
        zero_acc();
    }
#undef AC_KNN_STEP
    bwait_vm<0>();                                                  // the over-issued tail stages must land before LDS is released
}

}  // namespace

namespace ac {

static int knn_kp(int D) { return (D + 63) / 64 * 64; }

// rows of the sample with stride s: runs of 8 consecutive rows every 8 s rows (the incomplete last run is left out)
int64_t knn_sample_rows(int64_t N, int64_t s) { return s <= 1 ? N : N / (8 * s) * 8; }

size_t knn_planes_bytes(int64_t rows, int D) {
    const int64_t rp = (rows + 255) / 256 * 256;
    return (size_t)rp * knn_kp(D) * sizeof(uint16_t);
}

// |v - exact| <= gamma (max|p| + |q|)^2 for the fp16 one-product sweep over D columns (derivation: include/acamd.h at
// ac_knn_l2_topk_batch)
double knn_batch_gamma(int D) {
    const double Kp = knn_kp(D);
    return 1.01 * (4.8828125e-4 + (Kp + 18.0 + sqrt(Kp)) * 5.9604644775390625e-08);
}

int knn_prepare_store(const float* X, int64_t ldx, int64_t rows, int D, uint16_t* plane, float* norms, uint32_t* maxnorm_bits,
                      hipStream_t stream) {
    const int Kp = knn_kp(D);
    const int64_t rp = (rows + 255) / 256 * 256;
    if (rp == 0) return AC_OK;
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, X, ldx, rows, D, norms, maxnorm_bits, (int64_t)0);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_plane_kernel, dim3((unsigned)((rp + 3) / 4)), dim3(256), 0, stream, X, ldx, rows, rp, D, Kp, maxnorm_bits, 0,
                       plane, (const double*)nullptr, 0, 0.0, (float*)nullptr, (float*)nullptr, norms, 1, (int32_t*)nullptr, (int64_t)0, (int64_t)0);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// ---- incremental maintenance of a prepared store (ac_knn_update_store) ----
// head[0] = max |p|^2 (float bits) sits behind the tile padding, i.e. at norms[round_up(N, 256)]: it moves when N crosses a tile
// boundary.  begin: carry the maximum to its new slot, remember the old one, clear the verdict, pad the new last tile with +inf.
__global__ void knn_update_begin_kernel(float* norms, int64_t np_old, int64_t np_new, int64_t n_new) {
    const int t = threadIdx.x;
    __shared__ float old_max;
    if (t == 0) old_max = norms[np_old];
    __syncthreads();
    for (int64_t r = n_new + t; r < np_new; r += blockDim.x) norms[r] = INFINITY;     // (rows [n_old, n_new) get their norms next)
    __syncthreads();
    if (t == 0) { norms[np_new] = old_max; norms[np_new + 2] = old_max; norms[np_new + 1] = 0.f; }
}
// finish: did the store's power-of-two scale change?  (then every OTHER row's fp16 plane entry is stale: the caller prepares anew)
__global__ void knn_update_finish_kernel(const float* norms, int64_t np_new, int32_t* exponent_changed) {
    const double po = sqrt((double)norms[np_new + 2] * 1.001), pn = sqrt((double)norms[np_new] * 1.001);
    *exponent_changed = unit_exponent(po) != unit_exponent(pn) ? 1 : 0;
}

int knn_update_store(const float* X, int64_t ldx, int64_t n_old, int64_t n_new, int D, uint16_t* plane, float* norms, int64_t row0,
                     int64_t nrows, int32_t* exponent_changed, hipStream_t stream) {
    const int Kp = knn_kp(D);
    const int64_t np_old = (n_old + 255) / 256 * 256, np_new = (n_new + 255) / 256 * 256;
    uint32_t* maxbits = reinterpret_cast<uint32_t*>(norms + np_new);
    hipLaunchKernelGGL(knn_update_begin_kernel, dim3(1), dim3(256), 0, stream, norms, np_old, np_new, n_new);
    AC_LAUNCH_CHECK();
    // norms of the changed rows (atomicMax into the maximum), then their plane entries with the scale of the (possibly raised)
    // maximum; when rows were appended the rest of the last tile is rewritten too (zero plane entries behind the last row)
    const int64_t r_end = n_new > n_old ? np_new : row0 + nrows;
    const int64_t cnt_n = row0 + nrows - row0, cnt_p = r_end - row0;
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((cnt_n + 3) / 4)), dim3(256), 0, stream, X, ldx, row0 + nrows, D, norms, maxbits, row0);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_plane_kernel, dim3((unsigned)((cnt_p + 3) / 4)), dim3(256), 0, stream, X, ldx, n_new, r_end, D, Kp, maxbits, 0,
                       plane, (const double*)nullptr, 0, 0.0, (float*)nullptr, (float*)nullptr, (float*)nullptr, 1, (int32_t*)nullptr,
                       (int64_t)0, row0);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_update_finish_kernel, dim3(1), dim3(1), 0, stream, norms, np_new, exponent_changed);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

int knn_prepare_queries(const double* sampleD, int kp, const float* Q, int64_t ldQ, int D, int nq, const uint32_t* maxnorm_bits,
                        double gamma, uint16_t* qplane, float* thr, float* qfac, hipStream_t stream, int32_t* zero_ints,
                        int64_t zero_count) {
    const int Kp = knn_kp(D);
    const int64_t qp = ((int64_t)nq + 255) / 256 * 256;
    hipLaunchKernelGGL(knn_plane_kernel, dim3((unsigned)((qp + 3) / 4)), dim3(256), 0, stream, Q, ldQ, (int64_t)nq, qp, D, Kp,
                       maxnorm_bits, 1, qplane, sampleD, kp, gamma, thr, qfac, (float*)nullptr, 0, zero_ints, zero_count, (int64_t)0);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// N = rows of the prepared store, row_stride >= 1: sweep the knn_sample_rows(N, row_stride) logical rows (BatchParams::row_stride)
// true when a whole-store sweep of (N, nq) with `segs` segments and re-rank width kp can take its thresholds from its own first
// tile round (BatchParams::two_phase): one query tile, a row tile for every workgroup, k' minima available, the burst kernel
bool knn_batch_two_phase_applies(int64_t N, int nq, int kp, int segs) {
    const char* e = getenv("AC_KNN_TWO_PHASE");               // (per call: the equivalence test switches between the two forms)
    const bool off = e && atoi(e) == 0;
    int nblk = ac::dev_info().cus / 8 * 8;
    if (nblk < 8) nblk = 8;
    const int64_t ntiles = (N + BBM - 1) / BBM;
    // (a process that asked for "no in-launch exchanges" -- AC_LN_FUSION=0: a device shared between streams or processes, where a
    //  grid may not be resident at once -- gets the sample stage here too)
    return !off && ac::ln_fusion_enabled() && nq <= BBN && ntiles >= nblk && kp <= nblk && nblk <= 64 * 8 && segs > 1;
}
size_t knn_batch_two_phase_bytes() { return (size_t)64 * 8 * 256 * sizeof(unsigned); }       // wgmin for the largest grid

int knn_batch_launch(const uint16_t* Pp, const float* pnorm, int64_t N, int D, const uint16_t* Qp, int nq, const float* thr,
                     const float* qfac, float* cand_d, int32_t* cand_i, int32_t* cand_cnt, int cap, int segs, int64_t row_stride,
                     int best_only, hipStream_t stream, int32_t* clear_ctr, int32_t* clear_stats, int two_phase_kp, unsigned* wgmin,
                     void* ctl) {
    // (measured and dropped, profiles/r03/knn_batch_probe*.txt: a ring of 5 slots -- no change, the DMA depth is not the limit;
    //  2 x 2 waves of 128 x 128 with the 512-register budget, one wave per SIMD -- 120 vs 80 ms at 4096 x 10M: hipcc shuffles
    //  ~200 accumulator registers per iteration and a lone wave per SIMD hides nothing)
    constexpr int ns = 4, nwv = 8;
    const size_t lds = (size_t)ns * BSLOT * 16;
    // (per call, like the launch sites of knn_l2.hip: function attributes are per device, and a cached flag is neither)
    AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_batch_sweep<ns, nwv, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_batch_sweep<ns, nwv, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_batch_sweep<ns, nwv, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BatchParams p;
    p.Pp = Pp; p.p_rows = (N + 255) / 256 * 256; p.pnorm = pnorm;
    p.q_rows = ((int64_t)nq + 255) / 256 * 256;
    p.row_stride = row_stride < 1 ? 1 : row_stride;
    p.best_only = best_only;
    p.N = knn_sample_rows(N, p.row_stride); p.Kp = knn_kp(D);
    p.ntiles = (p.N + BBM - 1) / BBM;
    AC_REQUIRE(N < (int64_t)1 << 31, AC_EUNSUPPORTED, "knn batch: %lld rows (row indices are 32-bit here)", (long long)N);
    p.cap = cap; p.segs = segs; p.segcap = cap / segs;
    p.two_phase = two_phase_kp > 0 ? 1 : 0; p.kp = two_phase_kp; p.nq_real = nq; p.thr_rw = const_cast<float*>(thr); p.wgmin = wgmin;
    p.ctl = (acp::GridCtl*)ctl;
    AC_REQUIRE(!p.two_phase || (wgmin && ctl && row_stride <= 1 && !best_only && knn_batch_two_phase_applies(N, nq, two_phase_kp, segs)),
               AC_EINVAL, "knn batch: two-phase thresholds asked for a launch that cannot take them");
    // one persistent workgroup per CU (128 KB of LDS); the grid is a multiple of 8 so every XCD gets the same share
    int nblk = ac::dev_info().cus / 8 * 8;
    if (nblk < 8) nblk = 8;
    const int per_xcd = nblk / 8;
    const int nqt_all = (int)(p.q_rows / BBN);
    // up to 4 query tiles share an XCD (their plane stays in its L2); at most 8 such sets per launch
    static const int b_env = getenv("AC_KNN_BATCH_B") ? atoi(getenv("AC_KNN_BATCH_B")) : 0;      // A/B: query tiles per XCD
    int bdiv = b_env == 8 ? 8 : 4;                                  // query tiles one XCD's share of the grid can be split over:
    while (per_xcd % bdiv) bdiv >>= 1;                              // a power of two that divides it (an odd share -- a CU mask -- gives 1)
    const int chunk = 8 * bdiv;                                     // query tiles per launch (8 XCDs x bdiv)
    for (int t0 = 0; t0 < nqt_all; t0 += chunk) {
        const int nqt = nqt_all - t0 < chunk ? nqt_all - t0 : chunk;
        int b = bdiv;
        while (b > 1 && b > nqt) b >>= 1;
        int sets = 1;
        while (sets * b < nqt) sets <<= 1;
        AC_REQUIRE(sets <= 8, AC_EUNSUPPORTED, "knn batch: %d query tiles do not fit 8 XCDs x %d", nqt, b);
        p.nqt = nqt; p.b = b; p.sets = sets;
        const size_t qoff = (size_t)t0 * BBN;
        p.Qp = Qp + qoff * 8;                                       // plane[k/8][q_rows][8]: tile t0 starts at row t0 * 256 of every k-slot
        p.thr = thr + qoff; p.qfac = qfac + qoff;
        p.cand_d = cand_d + qoff * cap; p.cand_i = cand_i + qoff * cap; p.cand_cnt = cand_cnt + qoff * segs;
        p.clear_ctr = t0 == 0 ? clear_ctr : nullptr; p.clear_stats = t0 == 0 ? clear_stats : nullptr;
        const dim3 grid((unsigned)nblk), block(64 * nwv);
        if (p.two_phase) hipLaunchKernelGGL((knn_batch_sweep<ns, nwv, true, true>), grid, block, lds, stream, p);
        else if (segs > 1 || p.row_stride > 1 || best_only) hipLaunchKernelGGL((knn_batch_sweep<ns, nwv, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((knn_batch_sweep<ns, nwv, false>), grid, block, lds, stream, p);
        AC_LAUNCH_CHECK();
    }
    return AC_OK;
}

}  // namespace ac
