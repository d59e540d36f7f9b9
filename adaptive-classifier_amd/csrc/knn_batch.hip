// Batched kNN proposals on the bf16 matrix pipe: the compute-bound regime of faiss.IndexFlatL2.search
// (/root/reference/src/adaptive_classifier/memory.py:113-114) -- many queries (predict_batch, BASELINE configs[2]:
// 4096 x 10M, configs[4]: 1024 x 2M) -- where the fp32-MFMA sweep of knn_l2.hip is bound by its 157 TFLOP/s pipe.
//
// Exactness does not come from the sweep: knn_merge_rerank recomputes the proposed candidates in fp64 and a
// certificate proves no other row can enter the top-k (knn_l2.hip).  The sweep only has to PROPOSE with a bounded
// error, so it can run on the bf16 pipe:
//   * every fp32 operand is split x = h + m + r, h = bf16(x), m = bf16(x - h) (|r| <= 2^-16 |x|); the store's planes
//     (h, m) and its row norms |p|^2 are prepared once per store (ac_knn_prepare_store), the queries' per call
//     (scaled by -2);
//   * v = |p|^2 - 2 q.p is evaluated as three v_mfma_f32_32x32x16_bf16 products (m.h, h.m, h.h; fp32 accumulate):
//     3/16 of the matrix time of the fp32-input form; |v - exact| <= E = gamma (|p| + |q|)^2 with gamma covering
//     the dropped m.m / r terms (3.02 * 2^-16 per product) and the fp32 accumulation (2 ulp per term assumed);
//   * it is a GEMM: 256 store rows x 128 queries per 8-wave workgroup, operands staged by global_load_lds from the
//     k-slot-major planes (the layout of gemm.hip), so every store row is read once per 128 queries instead of
//     once per 32.
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

constexpr int BBM = 256, BBN = 128;          // store rows x queries per workgroup tile
constexpr int BGA = BBM / 32, BGW = BBN / 32;
constexpr int BSBK = 16;                      // k per stage
constexpr int kBatchThreads = 512;

// ---- operand preparation: planes[p][k/8][row][8] (p = h, m), optional row norms --------------------------------
__global__ __launch_bounds__(256) void knn_split2_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int64_t rows_pad,
                                                         int D, int Kp, float scale, uint16_t* __restrict__ planes,
                                                         float* __restrict__ norms, uint32_t* __restrict__ maxnorm_bits) {
    // one wave per row: lanes own k-slots of 8
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_pad) return;
    const int nslot = Kp >> 3;
    const int64_t plane = rows_pad * (int64_t)Kp;
    double nn = 0.0;
    for (int q = lane; q < nslot; q += 64) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * q + e;
            v[e] = (row < rows && c < D) ? X[row * ldx + c] : 0.f;
            nn = fma((double)v[e], (double)v[e], nn);
            v[e] *= scale;                                     // power of two: exact
        }
        uint4 H, Mi;
        uint32_t l0, l1, l2, l3;
        ac::split2(v[0], v[1], H.x, Mi.x, l0);
        ac::split2(v[2], v[3], H.y, Mi.y, l1);
        ac::split2(v[4], v[5], H.z, Mi.z, l2);
        ac::split2(v[6], v[7], H.w, Mi.w, l3);
        uint16_t* dst = planes + ((int64_t)q * rows_pad + row) * 8;
        *reinterpret_cast<uint4*>(dst) = H;
        *reinterpret_cast<uint4*>(dst + plane) = Mi;
    }
    if (norms) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) nn += __shfl_xor(nn, o);
        if (lane == 0 && row < rows) {
            const float f = (float)nn;
            norms[row] = f;
            if (maxnorm_bits) atomicMax(maxnorm_bits, __float_as_uint(f));     // non-negative floats order like their bits
        }
    }
}

// thr[q] = (tau_q - |q|^2 + E_q) rounded up to fp32, tau_q = exact k'-th smallest sample distance (fp64)
__global__ __launch_bounds__(64) void knn_threshold_kernel(const double* __restrict__ sampleD, int kp, const float* __restrict__ Q,
                                                           int64_t ldQ, int D, int nq, int nq_pad, const uint32_t* maxnorm_bits,
                                                           double gamma, float* __restrict__ thr) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= nq) { if (lane == 0 && q < nq_pad) thr[q] = -INFINITY; return; }
    double a = 0;
    for (int c = lane; c < D; c += 64) { const double x = Q[(size_t)q * ldQ + c]; a = fma(x, x, a); }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o);
    if (lane == 0) {
        const double pn = sqrt((double)__uint_as_float(*maxnorm_bits) * 1.001), qn = sqrt(a);
        const double E = gamma * (pn + qn) * (pn + qn) + 1e-30;
        const double tau = sampleD[(size_t)q * kp + kp - 1];
        float t = (float)(tau - a + E);
        if ((double)t < tau - a + E) t = nextafterf(t, INFINITY);
        thr[q] = isfinite(tau) ? t : INFINITY;                 // (sample smaller than k': keep everything)
    }
}

struct BatchParams {
    const uint16_t* Pp; int64_t p_rows;      // store planes [2][Kp/8][p_rows][8]
    const float* pnorm;                      // [N]
    const uint16_t* Qp; int64_t q_rows;      // planes of -2 Q, q_rows = round_up(nq, 128)
    const float* thr;                        // [q_rows]
    int64_t N;
    int Kp;
    int G;                                   // row groups per query tile
    int nqt;                                 // query tiles of 128
    int map_by_xcd;                          // workgroup -> (query tile, row group) mapping, see the kernel
    int64_t ntiles;                          // row tiles of 256
    float* cand_d; int32_t* cand_i; int32_t* cand_cnt; int cap;
};

// C/D layout of v_mfma_f32_32x32x16_bf16: lane owns column (lane & 31); acc_row32(r, lane) gives its 16 rows.
__global__ __launch_bounds__(kBatchThreads, 4) void knn_batch_sweep(BatchParams prm) {
    __shared__ uint4 lds[2][2][(BGA + BGW) * 64];                  // [buffer][plane][A groups | W groups][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                        // 4 x 2 waves of 64 rows x 64 queries
    const int i32 = lane & 31, kg = lane >> 5;
    const int v = xcd_tile_id(blockIdx.x, gridDim.x);               // XCD-contiguous virtual id
    // Default: the nqt query-tile workgroups of one row group are neighbours on an XCD and share that row stream through
    // its L2.  Alternative (AC_KNN_BATCH_MAP=1, measured and rejected: 198 vs 187 ms at 4096 x 10M, 14.6 vs 13.3 ms at
    // 1024 x 2M): each XCD takes nqt / 8 query tiles against all row groups, which keeps its query tiles L2-resident and
    // reads the store 8 times in all, but leaves only nqt / 8 workgroups sharing a row stream.
    int qt, g;
    if (prm.map_by_xcd) {
        const int per_xcd = (int)(gridDim.x >> 3), qpx = prm.nqt >> 3;
        const int x = v / per_xcd, u = v - x * per_xcd;
        qt = x * qpx + u % qpx; g = u / qpx;
    } else { qt = v % prm.nqt; g = v / prm.nqt; }
    const int64_t my_tiles = prm.ntiles > g ? (prm.ntiles - 1 - g) / prm.G + 1 : 0;
    if (my_tiles == 0) return;
    const int nk = prm.Kp / BSBK;
    const int64_t units = my_tiles * nk;

    // staging sources.  W (queries): waves 0..3 stage group `wave`; A (store rows): all 8 waves stage group `wave`.
    const int64_t w_plane = prm.q_rows * (int64_t)prm.Kp, a_plane = prm.p_rows * (int64_t)prm.Kp;
    const int64_t w_step = 2 * prm.q_rows * 8, a_step = 2 * prm.p_rows * 8;
    const uint16_t* wsrc0 = prm.Qp + ((int64_t)kg * prm.q_rows + (qt * BBN + 32 * (wave & 3) + i32)) * 8;
    auto a_base = [&](int64_t it) -> const uint16_t* {
        int64_t row = (it * prm.G + g) * BBM + 32 * wave + i32;
        if (row > prm.N - 1) row = prm.N - 1;
        return prm.Pp + ((int64_t)kg * prm.p_rows + row) * 8;
    };
    const uint16_t* asrc = a_base(0);
    const uint16_t* wsrc = wsrc0;
    int64_t st_it = 0; int st_k = 0;                                // unit the next staging call loads
    auto stage = [&](int buf) {
        if (wave < BGW) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(wsrc + p * w_plane), (lds_void_t*)&lds[buf][p][(BGA + wave) * 64], 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(asrc + p * a_plane), (lds_void_t*)&lds[buf][p][wave * 64], 16, 0, 0);
        if (++st_k == nk) {
            st_k = 0;
            if (st_it + 1 < my_tiles) ++st_it;                      // past the end: the last tile again (never consumed)
            asrc = a_base(st_it); wsrc = wsrc0;
        } else { asrc += a_step; wsrc += w_step; }
    };

    // thresholds of this lane's two query columns (constant over the block's row tiles)
    float thr[2];
    int qcol[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        qcol[ni] = qt * BBN + wn * 64 + ni * 32 + i32;
        thr[ni] = prm.thr[qcol[ni]];
    }

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();
    stage(0);
    __syncthreads();
    int64_t it = 0; int kt = 0;
    for (int64_t u = 0; u < units; ++u) {
        const int cur = (int)(u & 1);
        stage(cur ^ 1);                                             // next unit (unconditional; the last one is a duplicate)
        bf16x8_t af[2][2], bf[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a][p] = __builtin_bit_cast(bf16x8_t, lds[cur][p][(2 * wm + a) * 64 + lane]);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b][p] = __builtin_bit_cast(bf16x8_t, lds[cur][p][(BGA + 2 * wn + b) * 64 + lane]);
        }
        constexpr int PAIRS[3][2] = {{1, 0}, {0, 1}, {0, 0}};       // m.h, h.m, h.h (smallest first)
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PAIRS[pr][0]], bf[b][PAIRS[pr][1]], acc[a][b], 0, 0, 0);
        if (++kt == nk) {
            // ---- tile done: v = |p|^2 + acc; keep what beats the query's threshold ----
            const int64_t row0 = (it * prm.G + g) * BBM + wm * 64;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                float pn[16];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {                    // rows (r & 3) + 8 (r >> 2) + 4 kg: four runs of 4
                    int64_t rr = row0 + mi * 32 + 8 * r4 + 4 * kg;
                    if (rr > prm.N - 4) {                            // ragged end: element loads
#pragma unroll
                        for (int e = 0; e < 4; ++e) pn[4 * r4 + e] = (rr + e < prm.N) ? prm.pnorm[rr + e] : INFINITY;
                    } else {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(prm.pnorm + rr);
                        pn[4 * r4] = t.x; pn[4 * r4 + 1] = t.y; pn[4 * r4 + 2] = t.z; pn[4 * r4 + 3] = t.w;
                    }
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float val = pn[r] + acc[mi][ni][r];
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
            zero_acc();
            kt = 0; ++it;
        }
        __syncthreads();                                            // drains the global_load_lds queue, ends every read of `cur`
    }
}

}  // namespace

namespace ac {

size_t knn_planes_bytes(int64_t rows, int D) {
    const int Kp = (D + 15) / 16 * 16;
    const int64_t rp = (rows + 127) / 128 * 128;
    return (size_t)2 * rp * Kp * sizeof(uint16_t);
}

int knn_split2(const float* X, int64_t ldx, int64_t rows, int D, float scale, uint16_t* planes, float* norms,
               uint32_t* maxnorm_bits, hipStream_t stream) {
    const int Kp = (D + 15) / 16 * 16;
    const int64_t rp = (rows + 127) / 128 * 128;
    if (rp == 0) return AC_OK;
    hipLaunchKernelGGL(knn_split2_kernel, dim3((unsigned)((rp + 3) / 4)), dim3(256), 0, stream, X, ldx, rows, rp, D, Kp, scale,
                       planes, norms, maxnorm_bits);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

int knn_thresholds(const double* sampleD, int kp, const float* Q, int64_t ldQ, int D, int nq, int nq_pad,
                   const uint32_t* maxnorm_bits, double gamma, float* thr, hipStream_t stream) {
    hipLaunchKernelGGL(knn_threshold_kernel, dim3(nq_pad), dim3(64), 0, stream, sampleD, kp, Q, ldQ, D, nq, nq_pad, maxnorm_bits,
                       gamma, thr);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

int knn_batch_launch(const uint16_t* Pp, const float* pnorm, int64_t N, int D, const uint16_t* Qp, int nq, const float* thr,
                     float* cand_d, int32_t* cand_i, int32_t* cand_cnt, int cap, hipStream_t stream) {
    BatchParams p;
    p.Pp = Pp; p.p_rows = (N + 127) / 128 * 128; p.pnorm = pnorm;
    p.Qp = Qp; p.q_rows = ((int64_t)nq + 127) / 128 * 128; p.thr = thr;
    p.N = N; p.Kp = (D + 15) / 16 * 16;
    p.nqt = (int)(p.q_rows / BBN);
    p.ntiles = (N + BBM - 1) / BBM;
    // two workgroups per CU stay resident; give every query tile enough row groups to fill the chip twice over,
    // but keep >= 2 row tiles per group (small stores: parallelism matters more than the pipeline ramp)
    int64_t G = ((int64_t)ac::dev_info().cus * 4 + p.nqt - 1) / p.nqt;
    if (G > p.ntiles / 2) G = p.ntiles / 2;
    if (G < 1) G = 1;
    p.G = (int)G;
    static const int map_env = getenv("AC_KNN_BATCH_MAP") ? atoi(getenv("AC_KNN_BATCH_MAP")) : -1;      // A/B switch
    p.map_by_xcd = (p.nqt % 8 == 0) && map_env == 1;
    p.cand_d = cand_d; p.cand_i = cand_i; p.cand_cnt = cand_cnt; p.cap = cap;
    hipLaunchKernelGGL(knn_batch_sweep, dim3((unsigned)(p.G * p.nqt)), dim3(kBatchThreads), 0, stream, p);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

}  // namespace ac
