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
//     LDS slots filled by global_load_lds from the k-slot-major plane (the layout and the loop of gemm_pipe.hip: counted
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

#include <float.h>
#include <math.h>

namespace {
using namespace acg;

constexpr int BBM = 256, BBN = 256;          // store rows x queries per workgroup tile
constexpr int BGA = BBM / 32, BGW = BBN / 32, BRG = BGA + BGW;
constexpr int BSBK = 32;                      // k per stage: two MFMA chunks of 16
constexpr int BNS = 4;                        // ring depth
constexpr int BSLOT = 2 * BRG * 64;           // uint4 per ring slot: [chunk][group][lane] = 32 KB
constexpr int BPPW = 4;                       // DMA pieces per wave and stage (32 pieces of 1 KB, 8 waves)
constexpr int kBatchThreads = 512;
constexpr size_t kBatchLds = (size_t)BNS * BSLOT * 16;

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
                                                        float* __restrict__ norms, uint32_t* __restrict__ maxnorm_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
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

// pass 2: plane[k/8][row][8] = fp16(x * 2^-e): e from the maximum row norm (store) or from the row's own norm (queries,
// which also get their threshold and their epilogue factor here)
//   thr[q]  = (tau_q - |q|^2 + E_q) rounded up to fp32, tau_q = exact k'-th smallest sample distance (fp64)
//   qfac[q] = -2 2^(e_p + e_q)
__global__ __launch_bounds__(256) void knn_plane_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int64_t rows_pad,
                                                        int D, int Kp, const uint32_t* __restrict__ maxnorm_bits, int per_row_scale,
                                                        uint16_t* __restrict__ plane, const double* __restrict__ sampleD, int kp,
                                                        double gamma, float* __restrict__ thr, float* __restrict__ qfac,
                                                        float* __restrict__ pad_norms) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
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
                const double tau = sampleD[(size_t)row * kp + kp - 1];
                float t = (float)(tau - a + E);
                if ((double)t < tau - a + E) t = nextafterf(t, INFINITY);
                thr[row] = isfinite(tau) ? t : INFINITY;                 // (sample smaller than k': keep everything)
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
        *reinterpret_cast<uint4*>(plane + ((int64_t)q * rows_pad + row) * 8) = H;
    }
}

struct BatchParams {
    const uint16_t* Pp; int64_t p_rows;      // store plane [Kp/8][p_rows][8] fp16
    const float* pnorm;                      // [p_rows]: |p|^2, +inf past N
    const uint16_t* Qp; int64_t q_rows;      // query plane, q_rows = round_up(nq, 256)
    const float* thr;                        // [q_rows]
    const float* qfac;                       // [q_rows]
    int64_t N;
    int Kp;
    int nqt;                                 // query tiles of 256 in this launch
    int b;                                   // query tiles per XCD (1, 2, 4)
    int sets;                                // query-tile sets (power of two <= 8): XCD x works on set x % sets
    int64_t ntiles;                          // row tiles of 256
    float* cand_d; int32_t* cand_i; int32_t* cand_cnt; int cap;
};

template <int N> __device__ __forceinline__ void bwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// C/D layout of v_mfma_f32_32x32x16_f16: lane owns column (lane & 31); acc_row32(r, lane) gives its 16 rows.
__global__ __launch_bounds__(kBatchThreads, 2) void knn_batch_sweep(BatchParams prm) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];    // [BNS][2 chunks][BRG groups][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                        // 4 x 2 waves of 64 rows x 128 queries
    const int i32 = lane & 31, kg = lane >> 5;
    // ---- workgroup -> (query tile, row group) ----
    const int per_xcd = (int)(gridDim.x >> 3), xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int xs = 8 / prm.sets, set = xcd % prm.sets, xr = xcd / prm.sets;
    const int qt = set * prm.b + j % prm.b;
    const int rgx = per_xcd / prm.b;                                // row groups per XCD
    const int g = xr * rgx + j / prm.b, G = xs * rgx;
    if (qt >= prm.nqt) return;                                      // (whole workgroup)
    const int64_t my_tiles = prm.ntiles > g ? (prm.ntiles - 1 - g) / G + 1 : 0;
    if (my_tiles == 0) return;
    const int nk = prm.Kp / BSBK;                                   // even (Kp % 64 == 0)
    const int64_t units = my_tiles * nk;

    // ---- DMA stream.  Wave w stages store group w and query group w, both chunks: pieces (chunk c, A) and (chunk c, W).
    //      A piece = 32 rows x 2 k-slots: lane (i32, kg) copies the 16 B of row i32, k-slot 4 s + 2 c + kg.
    const int64_t a_step = 4 * prm.p_rows * 8, w_step = 4 * prm.q_rows * 8;
    auto a_base = [&](int64_t it) -> const uint16_t* {
        int64_t row = (it * G + g) * BBM + 32 * wave + i32;
        if (row > prm.N - 1) row = prm.N - 1;
        return prm.Pp + ((int64_t)kg * prm.p_rows + row) * 8;
    };
    const uint16_t* const pw0 = prm.Qp + ((int64_t)kg * prm.q_rows + ((int64_t)qt * BBN + 32 * wave + i32)) * 8;
    const uint16_t* pa = a_base(0);
    const uint16_t* pw = pw0;
    const int64_t a_c1 = 2 * prm.p_rows * 8, w_c1 = 2 * prm.q_rows * 8;   // chunk 1 = two k-slots further
    int64_t st_it = 0; int st_k = 0, st_slot = 0;                   // unit / ring slot the next issue() loads
    auto issue = [&]() {                                            // always BPPW DMA instructions (exact vmcnt accounting)
        uint4* dst = lds + st_slot * BSLOT;
        __builtin_amdgcn_global_load_lds((glb_void_t*)pa, (lds_void_t*)(dst + (0 * BRG + wave) * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)pw, (lds_void_t*)(dst + (0 * BRG + BGA + wave) * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pa + a_c1), (lds_void_t*)(dst + (1 * BRG + wave) * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pw + w_c1), (lds_void_t*)(dst + (1 * BRG + BGA + wave) * 64), 16, 0, 0);
        st_slot = st_slot + 1 == BNS ? 0 : st_slot + 1;
        if (++st_k == nk) {
            st_k = 0;
            if (st_it + 1 < my_tiles) ++st_it;                      // past the end: the last tile again (never consumed)
            pa = a_base(st_it); pw = pw0;
        } else { pa += a_step; pw += w_step; }
    };

    // thresholds / epilogue factors of this lane's four query columns (constant over the workgroup's row tiles)
    float thr[4], qf[4];
    int qcol[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        qcol[ni] = qt * BBN + wn * 128 + ni * 32 + i32;
        thr[ni] = prm.thr[qcol[ni]];
        qf[ni] = prm.qfac[qcol[ni]];
    }

    f32x16 acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    struct Frags { f16x8_t a[2][2], b[2][4]; };                    // [chunk][tile]
    auto read_frags = [&](Frags& F, int slot) {
        const uint4* base = lds + slot * BSLOT + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int a = 0; a < 2; ++a) F.a[c][a] = __builtin_bit_cast(f16x8_t, base[(c * BRG + 2 * wm + a) * 64]);
#pragma unroll
            for (int b = 0; b < 4; ++b) F.b[c][b] = __builtin_bit_cast(f16x8_t, base[(c * BRG + BGA + 4 * wn + b) * 64]);
        }
    };
    auto mfmas = [&](const Frags& F) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[c][a], F.b[c][b], acc[a][b], 0, 0, 0);
    };
    // tile done: v = |p|^2 + f_q acc; keep what beats the query's threshold
    int64_t it = 0;
    auto filter = [&]() {
        const int64_t row0 = (it * G + g) * BBM + wm * 64;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float pn[16];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {                        // rows (r & 3) + 8 (r >> 2) + 4 kg: four runs of 4
                // (the norms array is padded to whole tiles with +inf: rows past N never qualify)
                const f32x4 t = *reinterpret_cast<const f32x4*>(prm.pnorm + row0 + mi * 32 + 8 * r4 + 4 * kg);
                pn[4 * r4] = t.x; pn[4 * r4 + 1] = t.y; pn[4 * r4 + 2] = t.z; pn[4 * r4 + 3] = t.w;
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float val = fmaf(acc[mi][ni][r], qf[ni], pn[r]);
                    if (val < thr[ni]) {
                        const int64_t row = row0 + mi * 32 + acc_row32(r, lane);
                        const int slot = atomicAdd(&prm.cand_cnt[qcol[ni]], 1);
                        if (slot < prm.cap) {
                            prm.cand_d[(size_t)qcol[ni] * prm.cap + slot] = val;
                            prm.cand_i[(size_t)qcol[ni] * prm.cap + slot] = (int32_t)row;
                        }
                    }
                }
        }
    };

    // ---- prologue: BNS - 1 stages in flight, stage 0 landed and visible ----
    zero_acc();
#pragma unroll
    for (int s = 0; s < BNS - 1; ++s) issue();
    bwait_vm<(BNS - 2) * BPPW>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Frags F0, F1;
    read_frags(F0, 0);
    int slot = 1, kt = 0;                                           // ring slot of stage u + 1
#define AC_KNN_STEP(FC, FN)                                                                                      \
    do {                                                                                                         \
        bwait_vm<(BNS - 3) * BPPW>();                    /* this wave's pieces of stage u + 1 have landed */      \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_barrier();                    /* ... everyone's; the MFMAs of stage u - 1 are issued */ \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        issue();                                         /* stage u + BNS - 1 -> the slot stage u - 1 used */     \
        read_frags(FN, slot);                            /* fragments of stage u + 1 */                           \
        slot = slot + 1 == BNS ? 0 : slot + 1;                                                                   \
        mfmas(FC);                                       /* stage u */                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {   /* 16 MFMAs, 12 fragment reads: one read after each MFMA */ \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                   \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (++kt == nk) { filter(); zero_acc(); kt = 0; ++it; }                                                  \
    } while (0)
    for (int64_t u = 0; u < units; u += 2) {                        // (units is even: nk is)
        AC_KNN_STEP(F0, F1);
        AC_KNN_STEP(F1, F0);
    }
#undef AC_KNN_STEP
    bwait_vm<0>();                                                  // the over-issued tail stages must land before LDS is released
}

}  // namespace

namespace ac {

static int knn_kp(int D) { return (D + 63) / 64 * 64; }

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
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, X, ldx, rows, D, norms, maxnorm_bits);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_plane_kernel, dim3((unsigned)((rp + 3) / 4)), dim3(256), 0, stream, X, ldx, rows, rp, D, Kp, maxnorm_bits, 0,
                       plane, (const double*)nullptr, 0, 0.0, (float*)nullptr, (float*)nullptr, norms);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

int knn_prepare_queries(const double* sampleD, int kp, const float* Q, int64_t ldQ, int D, int nq, const uint32_t* maxnorm_bits,
                        double gamma, uint16_t* qplane, float* thr, float* qfac, hipStream_t stream) {
    const int Kp = knn_kp(D);
    const int64_t qp = ((int64_t)nq + 255) / 256 * 256;
    hipLaunchKernelGGL(knn_plane_kernel, dim3((unsigned)((qp + 3) / 4)), dim3(256), 0, stream, Q, ldQ, (int64_t)nq, qp, D, Kp,
                       maxnorm_bits, 1, qplane, sampleD, kp, gamma, thr, qfac, (float*)nullptr);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

int knn_batch_launch(const uint16_t* Pp, const float* pnorm, int64_t N, int D, const uint16_t* Qp, int nq, const float* thr,
                     const float* qfac, float* cand_d, int32_t* cand_i, int32_t* cand_cnt, int cap, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        AC_HIP_CHECK(hipFuncSetAttribute((const void*)knn_batch_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBatchLds));
        attr_set = true;
    }
    BatchParams p;
    p.Pp = Pp; p.p_rows = (N + 255) / 256 * 256; p.pnorm = pnorm;
    p.q_rows = ((int64_t)nq + 255) / 256 * 256;
    p.N = N; p.Kp = knn_kp(D);
    p.ntiles = (N + BBM - 1) / BBM;
    p.cap = cap;
    // one persistent workgroup per CU (128 KB of LDS); the grid is a multiple of 8 so every XCD gets the same share
    int nblk = ac::dev_info().cus / 8 * 8;
    if (nblk < 8) nblk = 8;
    const int per_xcd = nblk / 8;
    const int nqt_all = (int)(p.q_rows / BBN);
    // up to 4 query tiles share an XCD (their plane stays in its L2); at most 8 such sets per launch
    for (int t0 = 0; t0 < nqt_all; t0 += 32) {
        const int nqt = nqt_all - t0 < 32 ? nqt_all - t0 : 32;
        int b = nqt >= 4 ? 4 : (nqt >= 2 ? 2 : 1);
        while (per_xcd % b) b >>= 1;
        int sets = 1;
        while (sets * b < nqt) sets <<= 1;
        AC_REQUIRE(sets <= 8, AC_EUNSUPPORTED, "knn batch: %d query tiles do not fit 8 XCDs x %d", nqt, b);
        p.nqt = nqt; p.b = b; p.sets = sets;
        const size_t qoff = (size_t)t0 * BBN;
        p.Qp = Qp + qoff * 8;                                       // plane[k/8][q_rows][8]: tile t0 starts at row t0 * 256 of every k-slot
        p.thr = thr + qoff; p.qfac = qfac + qoff;
        p.cand_d = cand_d + qoff * cap; p.cand_i = cand_i + qoff * cap; p.cand_cnt = cand_cnt + qoff;
        hipLaunchKernelGGL(knn_batch_sweep, dim3((unsigned)nblk), dim3(kBatchThreads), kBatchLds, stream, p);
        AC_LAUNCH_CHECK();
    }
    return AC_OK;
}

}  // namespace ac
