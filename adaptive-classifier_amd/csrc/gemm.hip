// fp32 GEMMs on the f32-input MFMA pipe (v_mfma_f32_32x32x2_f32: exact fp32 fma chains,
// 157 TFLOP/s peak on MI355X).  Replaces the torch nn.Linear / autograd matmul sites of the
// hot path: AdaptiveHead (models.py:49-69,71-80), its backward (classifier.py:1499,345) and the
// encoder's QKV / output / FFN projections (transformers BertModel, called at classifier.py:1271).
//
//  gemm_tile128_nt : C[M,N] = epi(A[M,K] . W[N,K]^T), 128x128x32 block tile, 4 waves (2x2) of
//                    64x64, operands staged through LDS in MFMA *fragment order* (a lane's
//                    ds_read_b128 yields 4 consecutive k of its row; XOR-swizzled so both the
//                    staging writes and the fragment reads are bank-conflict free),
//                    register-staged double buffering, one barrier per k-tile.
//  gemm_direct     : one 32x32 output tile per wave, operands straight from global memory in
//                    fragment shape, K split over the block's 4 waves and reduced through LDS in
//                    a fixed order (deterministic).  Any operand layout; used for the small
//                    latency-bound head GEMMs (B = 32) and odd shapes.
#include "common.h"
#include "gemm_common.h"
#ifndef AC_XCD_REMAP
#define AC_XCD_REMAP 1
#endif

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

namespace {

using namespace acg;

// ---------------------------------------------------------------------------------------------
// LDS-tiled NT kernel: (64*TM) x 128 x 32 block tile, 4 waves as 2(M) x 2(N), wave tile (32*TM) x 64.
//   TM = 2: 128x128 tile, 64 KB LDS, 2 blocks/CU        TM = 1: 64x128 tile, 48 KB LDS, 3 blocks/CU
// ---------------------------------------------------------------------------------------------
constexpr int BN = 128, BK = 32;
constexpr int kTileThreads = 256;

// slot of (row, c4) inside an operand image laid out [row groups of 32][4 k-blocks of 8][64 float4]
__device__ __forceinline__ int tile_slot(int row, int c4) {   // c4 in [0,8)
    const int rg = row >> 5, i = row & 31, kb = c4 >> 1, h = c4 & 1;
    return ((rg * 4 + kb) << 6) + (((h << 5) + i) ^ c4);
}

template <int EPI, int TM>
__global__ __launch_bounds__(kTileThreads, TM == 2 ? 2 : 3) void gemm_tile_nt(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
    float* __restrict__ C, int64_t ldc, int M, int N, int K, Epilogue epi) {
    constexpr int BM = 64 * TM;
    constexpr int A_SLOTS = (BM / 32) * 4 * 64, W_SLOTS = (BN / 32) * 4 * 64;
    constexpr int PA = BM / 32, PW = BN / 32;                // staging passes (32 rows each)
    __shared__ __attribute__((aligned(16))) f32x4 lds[2][A_SLOTS + W_SLOTS];   // [stage][A | W]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // column tiles fastest: consecutive blocks share the same A rows (L2 reuse of A)
    const int ntn = (N + BN - 1) / BN;
    const int bn = blockIdx.x % ntn, bm = blockIdx.x / ntn;
    const int m0 = bm * BM, n0 = bn * BN;

    // staging assignment: thread -> (row = tid/8 + 32*p, c4 = tid%8)
    const int srow = tid >> 3, sc4 = tid & 7;
    const float* aptr[PA];
    const float* wptr[PW];
    int aslot[PA], wslot[PW];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        int ra = m0 + srow + 32 * p; if (ra > M - 1) ra = M - 1;
        aptr[p] = A + (int64_t)ra * lda + 4 * sc4;
        aslot[p] = tile_slot(srow + 32 * p, sc4);
    }
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        int rw = n0 + srow + 32 * p; if (rw > N - 1) rw = N - 1;
        wptr[p] = W + (int64_t)rw * ldw + 4 * sc4;
        wslot[p] = A_SLOTS + tile_slot(srow + 32 * p, sc4);
    }

    f32x16 acc[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / BK;
    f32x4 ra[PA], rw[PW];
#pragma unroll
    for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const f32x4*>(aptr[p]);
#pragma unroll
    for (int p = 0; p < PW; ++p) rw[p] = *reinterpret_cast<const f32x4*>(wptr[p]);
#pragma unroll
    for (int p = 0; p < PA; ++p) lds[0][aslot[p]] = ra[p];
#pragma unroll
    for (int p = 0; p < PW; ++p) lds[0][wslot[p]] = rw[p];
    __syncthreads();

    const int h = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // prefetch the next k-tile (the last iteration re-reads the current one: loads stay
        // unconditional so the compiler's vmcnt accounting is exact)
        const int knext = (kt + 1 < nk) ? (kt + 1) * BK : kt * BK;
#pragma unroll
        for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const f32x4*>(aptr[p] + knext);
#pragma unroll
        for (int p = 0; p < PW; ++p) rw[p] = *reinterpret_cast<const f32x4*>(wptr[p] + knext);
        // keep the loads above the MFMA section: hipcc otherwise sinks them to their first use
        // (the ds_write after the k-tile) and their HBM/L2 latency is exposed every k-tile
        __builtin_amdgcn_sched_barrier(0);
        const f32x4* As = lds[cur];
        const f32x4* Ws = lds[cur] + A_SLOTS;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int sl = lane ^ ((kb * 2 + h) & 7);
            f32x4 af[TM], bf[2];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = As[(((TM * wm + a) * 4 + kb) << 6) + sl];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = Ws[(((2 * wn + b) * 4 + kb) << 6) + sl];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // unconditional (the last k-tile stores a duplicate nobody reads): a conditional store lets
        // hipcc sink the loads into the branch, i.e. below the MFMA section again
#pragma unroll
        for (int p = 0; p < PA; ++p) lds[cur ^ 1][aslot[p]] = ra[p];
#pragma unroll
        for (int p = 0; p < PW; ++p) lds[cur ^ 1][wslot[p]] = rw[p];
        __syncthreads();
    }

    store_tile<EPI, TM>(acc, C, ldc, M, N, m0, n0, wm, wn, lane, epi);
}

// ---------------------------------------------------------------------------------------------
// Split-operand NT kernel: the same (64*TM) x 128 x 32 tile on the bf16 matrix pipe at fp32 accuracy.
//   Every fp32 operand element is split while it is staged into LDS:  x = h + m + l  with
//   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)  (round-to-nearest-even, v_cvt_pk_bf16_f32; the two
//   subtractions are exact, so h + m + l == x to 2^-27 |x|).  A tile product is then six
//   v_mfma_f32_32x32x16_bf16 (h.h, h.m, m.h, m.m, h.l, l.h), each product exact in the fp32 accumulator;
//   the dropped m.l + l.m + l.l terms are <= 2^-26 |a||b| per product -- below the 2^-24 rounding an
//   fp32 fma makes on the same product.  Six bf16 MFMAs cost 6/16 of the fp32-input MFMAs they replace.
//   Non-finite operands give NaN (inf - inf in the split), unlike a true fp32 fma.
//   LDS image per operand and plane: [row groups of 32][2 k-chunks of 16][64 x 16 B] in fragment order
//   (lane (i, kg) of chunk c reads 8 bf16 = k 16c + 8kg .. +7 of row i with one ds_read_b128), slots
//   XOR-swizzled by the k-slot so the 8-lane ds_write_b128 groups hit distinct banks.
// ---------------------------------------------------------------------------------------------
using ac::split8;

__device__ __forceinline__ int split_slot(int row, int q) {   // q = k-slot of 8 in [0,4)
    const int rg = row >> 5, i = row & 31, c = q >> 1, kg = q & 1;
    return ((rg * 2 + c) << 6) + (((kg << 5) + i) ^ (q << 2));
}

template <int EPI, int TM>
__global__ __launch_bounds__(kTileThreads, TM == 2 ? 3 : 4) void gemm_split_nt(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
    float* __restrict__ C, int64_t ldc, int M, int N, int K, Epilogue epi) {
    constexpr int BM = 64 * TM;
    constexpr int A_SLOTS = (BM / 32) * 2 * 64, W_SLOTS = (BN / 32) * 2 * 64;
    constexpr int PA = BM / 64, PW = BN / 64;                // staging passes (64 rows x 4 k-slots each)
    __shared__ uint4 lds[3][A_SLOTS + W_SLOTS];              // [plane h,m,l][A | W]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (N + BN - 1) / BN;
    const int bn = blockIdx.x % ntn, bm = blockIdx.x / ntn;
    const int m0 = bm * BM, n0 = bn * BN;

    // staging assignment: thread -> (row = tid/4 + 64*p, q = tid%4): 32 B of a row per thread
    const int srow = tid >> 2, sq = tid & 3;
    const float* aptr[PA];
    const float* wptr[PW];
    int aslot[PA], wslot[PW];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        int ra = m0 + srow + 64 * p; if (ra > M - 1) ra = M - 1;
        aptr[p] = A + (int64_t)ra * lda + 8 * sq;
        aslot[p] = split_slot(srow + 64 * p, sq);
    }
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        int rw = n0 + srow + 64 * p; if (rw > N - 1) rw = N - 1;
        wptr[p] = W + (int64_t)rw * ldw + 8 * sq;
        wslot[p] = A_SLOTS + split_slot(srow + 64 * p, sq);
    }

    f32x16 acc[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 ra[PA][2], rw[PW][2];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            ra[p][0] = *reinterpret_cast<const f32x4*>(aptr[p] + k0);
            ra[p][1] = *reinterpret_cast<const f32x4*>(aptr[p] + k0 + 4);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            rw[p][0] = *reinterpret_cast<const f32x4*>(wptr[p] + k0);
            rw[p][1] = *reinterpret_cast<const f32x4*>(wptr[p] + k0 + 4);
        }
    };
    auto store_tile_lds = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            uint4 H, Mi, L;
            split8(ra[p][0], ra[p][1], H, Mi, L);
            lds[0][aslot[p]] = H; lds[1][aslot[p]] = Mi; lds[2][aslot[p]] = L;
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            uint4 H, Mi, L;
            split8(rw[p][0], rw[p][1], H, Mi, L);
            lds[0][wslot[p]] = H; lds[1][wslot[p]] = Mi; lds[2][wslot[p]] = L;
        }
    };

    const int nk = K / BK;
    load_tile(0);
    store_tile_lds();
    __syncthreads();

    const int kg = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        // prefetch the next k-tile into registers (unconditional: see gemm_tile_nt)
        load_tile((kt + 1 < nk) ? (kt + 1) * BK : kt * BK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int sl = lane ^ ((2 * c + kg) << 2);
            bf16x8_t af[TM][3], bf[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    af[a][pl] = __builtin_bit_cast(bf16x8_t, lds[pl][(((TM * wm + a) * 2 + c) << 6) + sl]);
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    bf[b][pl] = __builtin_bit_cast(bf16x8_t, lds[pl][A_SLOTS + (((2 * wn + b) * 2 + c) << 6) + sl]);
            }
            // smallest products first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
            constexpr int PAIRS[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PAIRS[pr][0]], bf[b][PAIRS[pr][1]],
                                                                            acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();          // every wave is done reading this k-tile
        store_tile_lds();         // unconditional (the last one is a duplicate nobody reads)
        __syncthreads();
    }
    store_tile<EPI, TM>(acc, C, ldc, M, N, m0, n0, wm, wn, lane, epi);
}

// ---------------------------------------------------------------------------------------------
// Pre-split operands ("planes").  A matrix X[rows, K] (K % 8 == 0) is stored as three bf16 planes
//   P[p][q][row][e],  p = 0,1,2 (h, m, l),  q = k / 8,  e = k % 8          (bf16 units: 3 * rows * K)
// i.e. k-slot-major: the 16 bytes a lane feeds to v_mfma_f32_32x32x16_bf16 are contiguous, and the
// 32 rows x 2 k-slots a wave stages for one MFMA chunk are two contiguous 512-B runs -- so a
// global_load_lds dwordx4 per lane drops the fragment image into LDS in exactly the order the
// ds_read_b128 of lane (i, kg) wants it (slot = kg * 32 + i): no VGPR staging, no LDS stores, no swizzle.
// Weights are split once (HipBertEncoder init); activations by their producers or split in-kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows,
                                                           int K, uint16_t* __restrict__ P) {
    const int nq = K >> 3;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= rows * nq) return;
    const int64_t row = u / nq;
    const int q = (int)(u - row * nq);
    const float* src = X + row * ldx + 8 * q;
    uint4 H, Mi, L;
    split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), H, Mi, L);
    const int64_t plane = rows * (int64_t)K;
    uint16_t* dst = P + ((int64_t)q * rows + row) * 8;
    *reinterpret_cast<uint4*>(dst) = H;
    *reinterpret_cast<uint4*>(dst + plane) = Mi;
    *reinterpret_cast<uint4*>(dst + 2 * plane) = L;
}

// the two fp16 planes of X 2^s (AC_GEMM_F16X2; common.h split2h), same unit -> thread mapping
__global__ __launch_bounds__(256) void split_planes_f16_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows,
                                                               int K, float scale, uint16_t* __restrict__ P) {
    const int nq = K >> 3;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= rows * nq) return;
    const int64_t row = u / nq;
    const int q = (int)(u - row * nq);
    const float* src = X + row * ldx + 8 * q;
    uint4 H, L;
    ac::split8h(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), scale, H, L);
    uint16_t* dst = P + ((int64_t)q * rows + row) * 8;
    *reinterpret_cast<uint4*>(dst) = H;
    *reinterpret_cast<uint4*>(dst + rows * (int64_t)K) = L;
}

constexpr int SBK = 16;   // k per stage = one bf16 MFMA chunk

// C[M,N] = epi(A . W^T) with W given as planes; A as fp32 (split while staged) or as planes.
//   (64*TM) x 128 x 16 stages, two LDS buffers, ONE barrier per stage (24 MFMAs per wave at TM = 2).
//   (A three-buffer ring with loads spanning the barrier -- raw s_barrier + counted vmcnt -- measured 12 %
//   SLOWER at 8192^3: 72 KB of LDS leaves 2 blocks per CU instead of 3, and occupancy is what hides the
//   fragment-read latency here.)
//   WMW = wave rows of the block: 2 -> 4 waves, (64*TM) x 128 tile, 3 blocks/CU;  4 -> 8 waves, (128*TM) x 128
//   tile (25 % fewer staged bytes per MFMA), 2 blocks/CU -- used when the problem has >= 1.5 rounds of such tiles.
template <int EPI, int TM, bool A_PLANES, bool C_PLANES, int WMW = 2>
__global__ __launch_bounds__(128 * WMW, (WMW == 2 ? 3 : 4)) void gemm_planes_nt(
    const float* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Ap, int64_t a_rows,
    const uint16_t* __restrict__ Wp, int64_t w_rows, float* __restrict__ C, int64_t ldc, int M, int N, int K,
    Epilogue epi) {
    constexpr int BM = 32 * TM * WMW;
    constexpr int RA = BM / 32, RW = BN / 32;                       // 32-row groups per operand
    __shared__ uint4 lds[2][3][(RA + RW) * 64];                     // [buffer][plane][A groups | W groups][lane]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (N + BN - 1) / BN;
    const int tile = AC_XCD_REMAP ? xcd_tile_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int bn = tile % ntn, bm = tile / ntn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int i = lane & 31, kg = lane >> 5;

    // W: wave w < 4 stages row group w.  Source of (plane p, stage kt): wsrc + p * w_plane + kt * w_step
    int wrow = n0 + 32 * (wave & 3) + i; if (wrow > N - 1) wrow = N - 1;
    const uint16_t* wsrc = Wp + ((int64_t)kg * w_rows + wrow) * 8;
    const int64_t w_plane = w_rows * (int64_t)K, w_step = 2 * w_rows * 8;
    // A as planes: waves < RA stage row group `wave`
    int arow = m0 + 32 * wave + i; if (arow > M - 1) arow = M - 1;
    const uint16_t* asrc = A_PLANES ? Ap + ((int64_t)kg * a_rows + arow) * 8 : nullptr;
    const int64_t a_plane = a_rows * (int64_t)K, a_step = 2 * a_rows * 8;
    // A as fp32: thread -> (row = tid / 2, k-slot = tid % 2), 32 B of a row per thread
    static_assert(!(WMW == 4 && !A_PLANES), "the 8-wave tile is built for pre-split A only");
    const int frow = tid >> 1, fkg = tid & 1;
    int farow = m0 + frow; if (farow > M - 1) farow = M - 1;
    const float* fsrc = A_PLANES ? nullptr : A + (int64_t)farow * lda + 8 * fkg;
    const int fslot = (frow >> 5) * 64 + fkg * 32 + (frow & 31);
    const bool fact = frow < BM;                                    // TM = 1: half the threads stage A

    auto stage_glds = [&](int kt, int buf) {
        if (WMW == 2 || wave < RW) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(wsrc + p * w_plane + kt * w_step),
                                                 (lds_void_t*)&lds[buf][p][(RA + wave) * 64], 16, 0, 0);
        }
        if (A_PLANES && wave < RA) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(asrc + p * a_plane + kt * a_step),
                                                 (lds_void_t*)&lds[buf][p][wave * 64], 16, 0, 0);
        }
    };
    f32x4 fa0, fa1;
    auto load_a = [&](int kt) {
        fa0 = *reinterpret_cast<const f32x4*>(fsrc + kt * SBK);
        fa1 = *reinterpret_cast<const f32x4*>(fsrc + kt * SBK + 4);
    };
    auto store_a = [&](int buf) {
        uint4 H, Mi, L;
        split8(fa0, fa1, H, Mi, L);
        if (fact) { lds[buf][0][fslot] = H; lds[buf][1][fslot] = Mi; lds[buf][2][fslot] = L; }
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / SBK;
    stage_glds(0, 0);
    if (!A_PLANES) { load_a(0); store_a(0); }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int nxt = (kt + 1 < nk) ? kt + 1 : kt;     // unconditional staging (the last one is a duplicate)
        stage_glds(nxt, cur ^ 1);
        if (!A_PLANES) load_a(nxt);
        bf16x8_t af[TM][3], bf[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a][pl] = __builtin_bit_cast(bf16x8_t, lds[cur][pl][(TM * wm + a) * 64 + lane]);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b][pl] = __builtin_bit_cast(bf16x8_t, lds[cur][pl][(RA + 2 * wn + b) * 64 + lane]);
        }
        constexpr int PAIRS[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};   // smallest products first
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PAIRS[pr][0]], bf[b][PAIRS[pr][1]],
                                                                        acc[a][b], 0, 0, 0);
        if (!A_PLANES) store_a(cur ^ 1);
        __syncthreads();      // drains the global_load_lds queue (vmcnt(0)) and ends every read of `cur`
    }
    if (C_PLANES) {
        static_assert(sizeof(lds) >= 2 * WMW * kTrFloats * sizeof(float), "transpose scratch must fit the staging buffers");
        store_tile_planes<EPI, TM>(acc, reinterpret_cast<uint16_t*>(C), M, N, m0, n0, wm, wn, lane, epi,
                                   reinterpret_cast<float*>(&lds[0][0][0]) + wave * kTrFloats);
    } else store_tile<EPI, TM, BM>(acc, C, ldc, M, N, m0, n0, wm, wn, lane, epi);
}

// ---- split-K form for GEMMs with few output tiles (the CLS-only last encoder layer: 256 rows) ----
// 24 tiles of 64 x 128 on 256 CUs leave 90 % of the chip idle through a K = 3072 loop (185 us for 1.2 GFLOP).  Here the
// grid is tiles x ksplit: workgroup (tile, z) accumulates k in [z K / ksplit, (z + 1) K / ksplit) and stores its raw fp32
// tile into slice z of a scratch buffer; gemm_splitk_reduce adds the slices in order z = 0, 1, ... (deterministic) and
// applies the epilogue.  Staging and MFMA order inside a slice are those of gemm_planes_nt<., 1, false, false>.
__global__ __launch_bounds__(256, 3) void gemm_planes_splitk_nt(const float* __restrict__ A, int64_t lda,
                                                                const uint16_t* __restrict__ Wp, int64_t w_rows,
                                                                float* __restrict__ part, int M, int N, int K, int ksplit) {
    constexpr int BM = 64, RA = 2, RW = BN / 32;
    __shared__ uint4 lds[2][3][(RA + RW) * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (N + BN - 1) / BN;
    const int ntiles = (int)gridDim.x / ksplit, z = (int)blockIdx.x / ntiles, tile = (int)blockIdx.x - z * ntiles;
    const int bn = tile % ntn, bm = tile / ntn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int i = lane & 31, kg = lane >> 5;
    int wrow = n0 + 32 * (wave & 3) + i; if (wrow > N - 1) wrow = N - 1;
    const uint16_t* wsrc = Wp + ((int64_t)kg * w_rows + wrow) * 8;
    const int64_t w_plane = w_rows * (int64_t)K, w_step = 2 * w_rows * 8;
    const int frow = tid >> 1, fkg = tid & 1;
    int farow = m0 + frow; if (farow > M - 1) farow = M - 1;
    const float* fsrc = A + (int64_t)farow * lda + 8 * fkg;
    const int fslot = (frow >> 5) * 64 + fkg * 32 + (frow & 31);
    const bool fact = frow < BM;
    auto stage_glds = [&](int kt, int buf) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(wsrc + p * w_plane + kt * w_step),
                                             (lds_void_t*)&lds[buf][p][(RA + wave) * 64], 16, 0, 0);
    };
    f32x4 fa0, fa1;
    auto load_a = [&](int kt) {
        fa0 = *reinterpret_cast<const f32x4*>(fsrc + kt * SBK);
        fa1 = *reinterpret_cast<const f32x4*>(fsrc + kt * SBK + 4);
    };
    auto store_a = [&](int buf) {
        uint4 H, Mi, L;
        split8(fa0, fa1, H, Mi, L);
        if (fact) { lds[buf][0][fslot] = H; lds[buf][1][fslot] = Mi; lds[buf][2][fslot] = L; }
    };
    f32x16 acc[1][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;
    const int nks = K / SBK / ksplit, kt_lo = z * nks, kt_hi = kt_lo + nks;
    stage_glds(kt_lo, 0);
    load_a(kt_lo); store_a(0);
    __syncthreads();
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        const int cur = (kt - kt_lo) & 1;
        const int nxt = (kt + 1 < kt_hi) ? kt + 1 : kt;
        stage_glds(nxt, cur ^ 1);
        load_a(nxt);
        bf16x8_t af[3], bf[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            af[pl] = __builtin_bit_cast(bf16x8_t, lds[cur][pl][wm * 64 + lane]);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b][pl] = __builtin_bit_cast(bf16x8_t, lds[cur][pl][(RA + 2 * wn + b) * 64 + lane]);
        }
        constexpr int PAIRS[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};   // smallest products first
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PAIRS[pr][0]], bf[b][PAIRS[pr][1]], acc[0][b], 0, 0, 0);
        store_a(cur ^ 1);
        __syncthreads();
    }
    Epilogue none{};
    store_tile<EPI_IDENT, 1, BM>(acc, part + (size_t)z * M * N, N, M, N, m0, n0, wm, wn, lane, none);
}

// C = epi(sum_z part[z]): four outputs per thread, slices added in order
__global__ __launch_bounds__(256) void gemm_splitk_reduce(const float* __restrict__ part, int ksplit, int M, int N, float* __restrict__ C,
                                                          int64_t ldc, Epilogue epi) {
    const int64_t e4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N >> 2;
    if (e4 >= (int64_t)M * n4) return;
    const int row = (int)(e4 / n4), col = (int)(e4 % n4) * 4;
    const size_t slice = (size_t)M * N;
    f32x4 s = *reinterpret_cast<const f32x4*>(part + (size_t)row * N + col);
    for (int z = 1; z < ksplit; ++z) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(part + z * slice + (size_t)row * N + col);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    float* dst = C + (int64_t)row * ldc + col;
    dst[0] = apply_epilogue(epi, s.x, row, col, C, ldc, N);
    dst[1] = apply_epilogue(epi, s.y, row, col + 1, C, ldc, N);
    dst[2] = apply_epilogue(epi, s.z, row, col + 2, C, ldc, N);
    dst[3] = apply_epilogue(epi, s.w, row, col + 3, C, ldc, N);
}

// ---------------------------------------------------------------------------------------------
// direct kernel: one 32x32 tile per block, K split over 4 waves
//   AK: A element (m,k) at A[m*lda + k] (K-major) else A[k*lda + m]
//   BK_: B element (k,n) at B[n*ldb + k] (K-major, nn.Linear weight) else B[k*ldb + n]
// ---------------------------------------------------------------------------------------------
template <bool KMAJ>
__device__ __forceinline__ f32x4 load_frag(const float* base, int64_t ld, int idx, int idx_max, int k0,
                                           int K, bool vec_ok) {
    // idx = row (A) or column (B) owned by this lane (already clamped); k0 = first of 4 k's
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (KMAJ) {
        const float* p = base + (int64_t)idx * ld + k0;
        if (vec_ok && k0 + 3 < K) {
            v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) if (k0 + s < K) v[s] = p[s];
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) if (k0 + s < K) v[s] = base[(int64_t)(k0 + s) * ld + idx];
    }
    (void)idx_max;
    return v;
}

// one 32 x 32 output tile at (m0, n0); `red` = the block's [4][16][64] LDS reduction buffer.
// colsum != nullptr (requires M <= 32, m0 == 0): also colsum[col] = sum over the tile's rows of the stored values, rows
// in ascending order (the bias gradient of the layer below: column sums of dA).
template <bool AK, bool BKM>
__device__ __forceinline__ void gemm_direct_tile(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                 int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                 const Epilogue& epi, int m0, int n0, float (*red)[16][64],
                                                 float* __restrict__ colsum) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    int arow = m0 + i; if (arow > M - 1) arow = M - 1;
    int bcol = n0 + i; if (bcol > N - 1) bcol = N - 1;
    const bool avec = AK && ((lda & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
    const bool bvec = BKM && ((ldb & 3) == 0) && ((((uintptr_t)B) & 15) == 0);

    const int nkb = (K + 7) / 8;
    const int kb_lo = (int)(((int64_t)wave * nkb) / 4), kb_hi = (int)(((int64_t)(wave + 1) * nkb) / 4);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        const int k0 = kb * 8 + 4 * h;
        const f32x4 a = load_frag<AK>(A, lda, arow, M, k0, K, avec);
        const f32x4 b = load_frag<BKM>(B, ldb, bcol, N, k0, K, bvec);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // 1024 outputs, 4 per thread; fixed summation order over the 4 K-slices
    float outv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int o = tid + 256 * e;          // o = r * 64 + l
        const int r = o >> 6, l = o & 63;
        const float v = ((red[0][r][l] + red[1][r][l]) + red[2][r][l]) + red[3][r][l];
        const int64_t row = m0 + acc_row32(r, l);
        const int col = n0 + (l & 31);
        outv[e] = 0.f;
        if (row < M && col < N) { outv[e] = apply_epilogue(epi, v, row, col, C, ldc, N); C[row * ldc + col] = outv[e]; }
    }
    if (colsum) {
        __syncthreads();                      // every K-slice sum has been read
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int o = tid + 256 * e; red[0][o >> 6][o & 63] = outv[e]; }
        __syncthreads();
        if (tid < 32 && n0 + tid < N) {
            float sacc = 0.f;
            for (int row = 0; row < 32; ++row) {                     // row = (r & 3) + 8 (r >> 2) + 4 h
                const int hh = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
                sacc += red[0][r][tid + 32 * hh];                  // rows >= M hold 0
            }
            colsum[n0 + tid] = sacc;
        }
    }
}

template <bool AK, bool BKM>
__global__ __launch_bounds__(256) void gemm_direct(const float* __restrict__ A, int64_t lda,
                                                   const float* __restrict__ B, int64_t ldb,
                                                   float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                   Epilogue epi) {
    __shared__ float red[4][16][64];
    gemm_direct_tile<AK, BKM>(A, lda, B, ldb, C, ldc, M, N, K, epi, blockIdx.y * 32, blockIdx.x * 32, red, nullptr);
}

// Two independent small GEMMs in ONE launch (the head's backward through a hidden layer, head.hip):
//   problem 1  dW[M1,N1] = dY^T A      (A element (m,k) at dY[k*lda + m], B element (k,n) at Aact[k*ldb + n])
//   problem 2  dA[M2,N2] = (dY W) gated, M2 <= 32, + column sums of dA (the bias gradient of the layer below)
// Blocks [0, nblk1) take tiles of problem 1 (column tiles fastest), the rest tiles of problem 2.
struct DirectProblem {
    const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc; int M, N, K; Epilogue epi; float* colsum;
};
__global__ __launch_bounds__(256) void gemm_direct_pair(DirectProblem p1, DirectProblem p2, int nblk1) {
    __shared__ float red[4][16][64];
    const int b = blockIdx.x;
    if (b < nblk1) {
        const int tn = (p1.N + 31) / 32;
        gemm_direct_tile<false, false>(p1.A, p1.lda, p1.B, p1.ldb, p1.C, p1.ldc, p1.M, p1.N, p1.K, p1.epi, (b / tn) * 32, (b % tn) * 32,
                                       red, nullptr);
    } else {
        const int t = b - nblk1, tn = (p2.N + 31) / 32;
        gemm_direct_tile<true, false>(p2.A, p2.lda, p2.B, p2.ldb, p2.C, p2.ldc, p2.M, p2.N, p2.K, p2.epi, (t / tn) * 32, (t % tn) * 32,
                                      red, p2.colsum);
    }
}

// ---------------------------------------------------------------------------------------------
// few-tile NT kernel (65 <= M <= 512 with fewer 64 x 128 tiles than half the CUs: the head's three layers over a predict
// batch, the CLS-row GEMMs of the encoder's last layer).  Such a shape is latency-bound, not pipe-bound: the tiled kernels put
// 256 x 768 x 768 on 24 workgroups (30 us), split-K over operand planes needs a reduce launch (10 + 6 us).  Here one workgroup
// owns a 32 x 32 output tile (256 x 768 -> 192 workgroups), its 8 waves split K in interleaved 16-column slots (a lane reads
// whole 64-byte lines of its row: two float4 of A, two of W per slot, straight into v_mfma_f32_32x32x2_f32 operands -- exact
// fp32 products, fp32 accumulation), the next slot's loads are in flight under the current slot's 8 MFMAs, and the 8 partial
// tiles meet in LDS in a fixed order.  One launch, ~5 us for the head's layers.
// ---------------------------------------------------------------------------------------------
constexpr int kFtWaves = 8;
__global__ __launch_bounds__(kFtWaves * 64) void gemm_fewtiles_nt(const float* __restrict__ A, int64_t lda,
                                                                  const float* __restrict__ W, int64_t ldw,
                                                                  float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                                  Epilogue epi, int tiles_n) {
    __shared__ float red[kFtWaves][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
    const int i = lane & 31, h = lane >> 5;
    int arow = m0 + i; if (arow > M - 1) arow = M - 1;
    int wrow = n0 + i; if (wrow > N - 1) wrow = N - 1;
    const float* ap = A + (int64_t)arow * lda + 8 * h;
    const float* wp = W + (int64_t)wrow * ldw + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // slot s of this wave covers k in [16 (8 s + wave), +16): this lane's 8 of them start at 16 (8 s + wave) + 8 h  (K % 8 == 0)
    auto load = [&](int s, f32x4 (&a)[2], f32x4 (&b)[2]) {
        const int k = 16 * (kFtWaves * s + wave);
        if (k + 8 * h < K) {
            a[0] = *reinterpret_cast<const f32x4*>(ap + k); a[1] = *reinterpret_cast<const f32x4*>(ap + k + 4);
            b[0] = *reinterpret_cast<const f32x4*>(wp + k); b[1] = *reinterpret_cast<const f32x4*>(wp + k + 4);
        } else {
            a[0] = a[1] = b[0] = b[1] = zero;
        }
    };
    const int nslots = (K + 16 * kFtWaves - 1) / (16 * kFtWaves);
    f32x4 a0[2], b0[2], a1[2], b1[2];
    load(0, a0, b0);
    for (int s = 0; s < nslots; s += 2) {
        load(s + 1, a1, b1);                    // (beyond K: zeros)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e][c], b0[e][c], acc, 0, 0, 0);
        load(s + 2, a0, b0);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e][c], b1[e][c], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {                // 1024 outputs, 2 per thread; fixed summation order over the 8 K-slices
        const int o = tid + 512 * e;              // o = r * 64 + l
        const int r = o >> 6, l = o & 63;
        const float v = (((red[0][r][l] + red[1][r][l]) + (red[2][r][l] + red[3][r][l])) +
                         ((red[4][r][l] + red[5][r][l]) + (red[6][r][l] + red[7][r][l])));
        const int64_t row = m0 + acc_row32(r, l);
        const int col = n0 + (l & 31);
        if (row < M && col < N) C[row * ldc + col] = apply_epilogue(epi, v, row, col, C, ldc, N);
    }
}
// (the shapes it takes: see the comment above; `aligned` = 16-byte aligned bases and lda, ldw multiples of 4)
static bool fewtiles_takes(int M, int N, int K, bool aligned) {
    if (const char* e = getenv("AC_GEMM_FEWTILES")) { if (atoi(e) == 0) return false; }
    const int64_t t64 = (int64_t)((M + 63) / 64) * ((N + BN - 1) / BN);
    // (<= 512 rows and <= 2^30 multiply-adds: beyond, the fp32 matrix pipe -- 1/16 of the bf16 one -- is the bound, not latency)
    return aligned && M >= 65 && M <= 512 && (K % 8) == 0 && K >= 64 && 2 * t64 <= ac::dev_info().cus &&
           (int64_t)M * N * K <= ((int64_t)1 << 30);
}

// ---------------------------------------------------------------------------------------------
// small-M NT kernel (M <= 64): weight streaming.  out[m][n] = sum_k X[m][k] W[n][k] is computed as
// (W rows) x (X^T) on v_mfma_f32_16x16x4_f32: the A operand is 16 rows of W straight from HBM
// (float4 per lane, 16 rows x 64 B per wave load, exactly the kNN sweep's access shape), the B operand
// the <= 64 activation rows (L2-resident).  One block = 16 output columns, its 8 waves split K and are
// reduced through LDS in a fixed order.  With M this small the GEMM is bound by streaming W once, so
// parallelism over (N/16 blocks) x (8 waves) and loads-in-flight matter, not MFMA efficiency.
// ---------------------------------------------------------------------------------------------
constexpr int kSmWaves = 8;
// k-blocks (16 columns each) per register buffer; two buffers of (1 + J) float4 per k-block
template <int J> struct SmGroup { static constexpr int value = (J >= 4) ? 3 : 6; };

template <int J>
__global__ __launch_bounds__(kSmWaves * 64) void gemm_smallm_nt(const float* __restrict__ X, int64_t ldx,
                                                                 const float* __restrict__ W, int64_t ldw,
                                                                 float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                                 Epilogue epi) {
    __shared__ float red[kSmWaves][J][4][64];
    constexpr int kSmGroup = SmGroup<J>::value;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const int i = lane & 15, h = lane >> 4;
    int wrow = n0 + i; if (wrow > N - 1) wrow = N - 1;
    const float* wp = W + (int64_t)wrow * ldw + 4 * h;
    const float* xp[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
        int m = 16 * jj + i; if (m > M - 1) m = M - 1;
        xp[jj] = X + (int64_t)m * ldx + 4 * h;
    }
    const int nkb = (K + 15) / 16;
    const int kb_lo = (int)(((int64_t)wave * nkb) / kSmWaves), kb_hi = (int)(((int64_t)(wave + 1) * nkb) / kSmWaves);
    f32x4 acc[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 wa[2][kSmGroup], xb[2][kSmGroup][J];
    // loads are unconditional: k-blocks past the wave's range / past K re-read column K-4 and are
    // zeroed when consumed (keeps hipcc's vmcnt accounting exact, cf. knn_l2.hip)
#define AC_SM_LOAD(Bf, kb0)                                                              \
    _Pragma("unroll") for (int u = 0; u < kSmGroup; ++u) {                               \
        int col = ((kb0) + u) * 16;                                                      \
        if (col + 4 * h > K - 4) col = K - 4 - 4 * h;                                    \
        wa[Bf][u] = *reinterpret_cast<const f32x4*>(wp + col);                           \
        _Pragma("unroll") for (int jj = 0; jj < J; ++jj)                                 \
            xb[Bf][u][jj] = *reinterpret_cast<const f32x4*>(xp[jj] + col);               \
    }
#define AC_SM_COMPUTE(Bf, kb0)                                                           \
    _Pragma("unroll") for (int u = 0; u < kSmGroup; ++u) {                               \
        const bool ok = ((kb0) + u) < kb_hi && (((kb0) + u) * 16 + 4 * h) < K;           \
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};                                         \
        const f32x4 a = ok ? wa[Bf][u] : zero;                                           \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                    \
            _Pragma("unroll") for (int jj = 0; jj < J; ++jj)                             \
                acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], xb[Bf][u][jj][s], acc[jj], 0, 0, 0); \
    }
    if (kb_lo < kb_hi) {
        AC_SM_LOAD(0, kb_lo);
        for (int kb = kb_lo; kb < kb_hi; kb += 2 * kSmGroup) {
            AC_SM_LOAD(1, kb + kSmGroup);
            __builtin_amdgcn_sched_barrier(0);
            AC_SM_COMPUTE(0, kb);
            __builtin_amdgcn_sched_barrier(0);
            AC_SM_LOAD(0, kb + 2 * kSmGroup);
            __builtin_amdgcn_sched_barrier(0);
            AC_SM_COMPUTE(1, kb + kSmGroup);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef AC_SM_LOAD
#undef AC_SM_COMPUTE
#pragma unroll
    for (int jj = 0; jj < J; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][jj][r][lane] = acc[jj][r];
    __syncthreads();
    // J*256 outputs (16 n x 16*J m); C/D layout: n = n0 + 4*(l>>4) + r, m = 16*jj + (l & 15)
    for (int o = tid; o < J * 256; o += kSmWaves * 64) {
        const int jj = o >> 8, r = (o >> 6) & 3, l = o & 63;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kSmWaves; ++w) v += red[w][jj][r][l];
        const int m = 16 * jj + (l & 15), n = n0 + 4 * (l >> 4) + r;
        if (m < M && n < N) C[(int64_t)m * ldc + n] = apply_epilogue(epi, v, m, n, C, ldc, N);
    }
}

static int launch_gemm(bool a_kmaj, bool b_kmaj, const float* A, int64_t lda, const float* B, int64_t ldb,
                       float* C, int64_t ldc, int M, int N, int K, const Epilogue& epi, hipStream_t stream,
                       const uint16_t* Bp = nullptr, int64_t b_rows = 0, const uint16_t* Ap = nullptr,
                       int64_t a_rows = 0, uint16_t* Cp = nullptr) {
    if (M <= 0 || N <= 0) return AC_OK;
    const bool aligned = ((lda & 3) == 0) && ((ldb & 3) == 0) && ((((uintptr_t)A) & 15) == 0) &&
                         ((((uintptr_t)B) & 15) == 0);
    AC_REQUIRE(epi.act != ACT_GEGLU32 || (Cp && Ap), AC_EUNSUPPORTED, "gemm: fused GeGLU needs the pre-split kernel");
    AC_REQUIRE((!Ap && !Cp) || (a_kmaj && b_kmaj && aligned && ac::linear_takes_planes(M, N, K) && Bp),
               AC_EUNSUPPORTED, "gemm: operand / result planes given for a shape that does not take the pre-split kernel");
    if (a_kmaj && b_kmaj && aligned && M <= 64 && K >= 8 && (K % 4) == 0 && N >= 16) {
        const dim3 grid((N + 15) / 16), block(kSmWaves * 64);
        if (M <= 16) hipLaunchKernelGGL((gemm_smallm_nt<1>), grid, block, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);
        else if (M <= 32) hipLaunchKernelGGL((gemm_smallm_nt<2>), grid, block, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);
        else hipLaunchKernelGGL((gemm_smallm_nt<4>), grid, block, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);
    } else if (a_kmaj && b_kmaj && !Ap && !Cp && fewtiles_takes(M, N, K, aligned)) {
        const int tn = (N + 31) / 32;
        hipLaunchKernelGGL(gemm_fewtiles_nt, dim3((unsigned)(((M + 31) / 32) * tn)), dim3(kFtWaves * 64), 0, stream, A, lda, B, ldb, C, ldc,
                           M, N, K, epi, tn);
    } else if (a_kmaj && b_kmaj && aligned && M >= 192 && K >= BK && (K % BK) == 0) {
        // pick the M-tile that wastes fewer CU-rounds: cost = rounds * (tile rows) * (resident blocks)
        const int cus = ac::dev_info().cus;
        const int64_t ntn = (N + BN - 1) / BN;
        const int64_t b128 = (int64_t)((M + 127) / 128) * ntn, b64 = (int64_t)((M + 63) / 64) * ntn;
        const int r128 = ac::arith_split() ? 3 : 2, r64 = r128 + 1;   // resident blocks per CU
        const int64_t cost128 = ((b128 + r128 * cus - 1) / (r128 * cus)) * 128 * r128;
        const int64_t cost64 = ((b64 + r64 * cus - 1) / (r64 * cus)) * 64 * r64;
        // the 128-row tile does ~15 % more work per staged byte: take the 64-row one only for a clear win
        int tm = (double)cost64 * 1.15 < (double)cost128 ? 1 : 2;
        if (const char* e = getenv("AC_GEMM_TM")) { int v = atoi(e); if (v == 1 || v == 2) tm = v; }
        const int64_t nblk = tm == 2 ? b128 : b64;
        const bool plain = epi.alpha == 1.f && epi.beta == 0.f && epi.bias && !epi.mask && !epi.gate &&
                           epi.drop_p == 0.f;
        int cls = EPI_GENERIC;
        if (plain && !epi.residual && epi.act == ACT_NONE) cls = EPI_BIAS;
        else if (plain && !epi.residual && epi.act == ACT_GELU) cls = EPI_BIAS_GELU;
        else if (plain && !epi.residual && epi.act == ACT_RELU) cls = EPI_BIAS_RELU;
        else if (plain && epi.residual && epi.act == ACT_NONE) cls = EPI_BIAS_RES;
        else if (plain && !epi.residual && epi.act == ACT_GEGLU32) cls = EPI_GEGLU32;
        AC_REQUIRE(epi.act != ACT_GEGLU32 || (cls == EPI_GEGLU32 && Cp && (N % 64) == 0), AC_EUNSUPPORTED,
                   "gemm: the fused GeGLU epilogue needs planes output, a bias vector and N %% 64 == 0");
        const dim3 grid((unsigned)nblk), block(kTileThreads);
        const bool split = ac::arith_split();
        const bool planes = split && Bp != nullptr;      // (K % 32 == 0 here, so K % SBK == 0)
#define AC_LAUNCH_PLANES(E, AP)                                                                               \
    do {                                                                                                      \
        if (tm == 2) hipLaunchKernelGGL((gemm_planes_nt<E, 2, AP, false>), grid, block, 0, stream, A, lda, Ap, a_rows, Bp, b_rows, C, ldc, M, N, K, epi); \
        else hipLaunchKernelGGL((gemm_planes_nt<E, 1, AP, false>), grid, block, 0, stream, A, lda, Ap, a_rows, Bp, b_rows, C, ldc, M, N, K, epi);         \
    } while (0)
#define AC_LAUNCH_TILE(E)                                                                                     \
    do {                                                                                                      \
        if (split && tm == 2) hipLaunchKernelGGL((gemm_split_nt<E, 2>), grid, block, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi); \
        else if (split) hipLaunchKernelGGL((gemm_split_nt<E, 1>), grid, block, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);      \
        else if (tm == 2) hipLaunchKernelGGL((gemm_tile_nt<E, 2>), grid, block, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);     \
        else hipLaunchKernelGGL((gemm_tile_nt<E, 1>), grid, block, 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);                  \
    } while (0)
        // ring-staged kernel (gemm_pipe.hip): ac_gemm_set_variant(cfg >= 1000) forces one configuration (A/B harness);
        // variant 0 = the measured per-shape choice of pipe_choose(), variant 1 = never
        if (planes && Ap && ac::pipe_takes(M, N, K, cls, Cp != nullptr)) {
            const int v = ac::gemm_variant();
            const int cfg = v >= 1000 ? v : (v == 0 ? ac::pipe_choose(M, N, K, cls, Cp != nullptr) : 0);
            if (cfg) return ac::launch_gemm_pipe(cfg, Ap, a_rows, Bp, b_rows, C, ldc, Cp, M, N, K, cls, epi, stream);
        }
        // 8-wave 256 x 128 tile when both operands are pre-split and the grid has >= 1.5 rounds of such tiles
        const int64_t b256 = (int64_t)((M + 255) / 256) * ntn;
        static const int tile256_env = getenv("AC_GEMM_TILE256") ? atoi(getenv("AC_GEMM_TILE256")) : -1;
        // ... or when the 256-row tiles make (almost) exactly ONE residency round (2 workgroups per CU) while the 128-row tiles
        // would spill > 10 % into a second one (FFN1 at ~5000 packed token rows: 504 vs 984 tiles, 154 vs 173-196 us)
        const bool one_round256 = b256 <= 2 * (int64_t)cus && 10 * b256 >= 17 * (int64_t)cus && 10 * b128 > 33 * (int64_t)cus;
        const bool big = planes && Ap && (tile256_env >= 0 ? tile256_env != 0 : (b256 >= 3 * (int64_t)cus || one_round256)) &&
                         (cls == EPI_BIAS || cls == EPI_BIAS_GELU || cls == EPI_BIAS_RES || cls == EPI_GEGLU32) &&
                         !(cls == EPI_GEGLU32 && !Cp) && !(cls == EPI_BIAS_RES && Cp);
        if (big) {
            const dim3 grid8((unsigned)b256), block8(512);
            float* Cq = Cp ? reinterpret_cast<float*>(Cp) : C;
#define AC_L8(E, CP) hipLaunchKernelGGL((gemm_planes_nt<E, 2, true, CP, 4>), grid8, block8, 0, stream, A, lda, Ap, a_rows, Bp, b_rows, Cq, ldc, M, N, K, epi)
            if (Cp) {
                AC_REQUIRE((N % 8) == 0, AC_EUNSUPPORTED, "gemm: planes output needs N %% 8 == 0");
                if (cls == EPI_GEGLU32) AC_L8(EPI_GEGLU32, true);
                else if (cls == EPI_BIAS_GELU) AC_L8(EPI_BIAS_GELU, true);
                else AC_L8(EPI_BIAS, true);
            } else if (cls == EPI_BIAS) AC_L8(EPI_BIAS, false);
            else if (cls == EPI_BIAS_GELU) AC_L8(EPI_BIAS_GELU, false);
            else AC_L8(EPI_BIAS_RES, false);
#undef AC_L8
        } else
        if (Cp) {
            // result emitted as planes for the next GEMM: both operands pre-split, bias (+GELU) epilogues only
            AC_REQUIRE(planes && Ap && (cls == EPI_BIAS || cls == EPI_BIAS_GELU || cls == EPI_GEGLU32) && (N % 8) == 0, AC_EUNSUPPORTED,
                       "gemm: planes output needs pre-split operands, N %% 8 == 0 and a bias / bias+gelu epilogue");
            float* Cq = reinterpret_cast<float*>(Cp);
            if (cls == EPI_GEGLU32) {
                if (tm == 2) hipLaunchKernelGGL((gemm_planes_nt<EPI_GEGLU32, 2, true, true>), grid, block, 0, stream, A, lda, Ap, a_rows, Bp, b_rows, Cq, ldc, M, N, K, epi);
                else hipLaunchKernelGGL((gemm_planes_nt<EPI_GEGLU32, 1, true, true>), grid, block, 0, stream, A, lda, Ap, a_rows, Bp, b_rows, Cq, ldc, M, N, K, epi);
            } else if (cls == EPI_BIAS_GELU) {
                if (tm == 2) hipLaunchKernelGGL((gemm_planes_nt<EPI_BIAS_GELU, 2, true, true>), grid, block, 0, stream, A, lda, Ap, a_rows, Bp, b_rows, Cq, ldc, M, N, K, epi);
                else hipLaunchKernelGGL((gemm_planes_nt<EPI_BIAS_GELU, 1, true, true>), grid, block, 0, stream, A, lda, Ap, a_rows, Bp, b_rows, Cq, ldc, M, N, K, epi);
            } else {
                if (tm == 2) hipLaunchKernelGGL((gemm_planes_nt<EPI_BIAS, 2, true, true>), grid, block, 0, stream, A, lda, Ap, a_rows, Bp, b_rows, Cq, ldc, M, N, K, epi);
                else hipLaunchKernelGGL((gemm_planes_nt<EPI_BIAS, 1, true, true>), grid, block, 0, stream, A, lda, Ap, a_rows, Bp, b_rows, Cq, ldc, M, N, K, epi);
            }
        } else if (planes) {
            const bool ap = Ap != nullptr;
            switch (cls) {
                case EPI_BIAS: if (ap) AC_LAUNCH_PLANES(EPI_BIAS, true); else AC_LAUNCH_PLANES(EPI_BIAS, false); break;
                case EPI_BIAS_GELU: if (ap) AC_LAUNCH_PLANES(EPI_BIAS_GELU, true); else AC_LAUNCH_PLANES(EPI_BIAS_GELU, false); break;
                case EPI_BIAS_RELU: if (ap) AC_LAUNCH_PLANES(EPI_BIAS_RELU, true); else AC_LAUNCH_PLANES(EPI_BIAS_RELU, false); break;
                case EPI_BIAS_RES: if (ap) AC_LAUNCH_PLANES(EPI_BIAS_RES, true); else AC_LAUNCH_PLANES(EPI_BIAS_RES, false); break;
                default: if (ap) AC_LAUNCH_PLANES(EPI_GENERIC, true); else AC_LAUNCH_PLANES(EPI_GENERIC, false); break;
            }
        } else
        switch (cls) {
            case EPI_BIAS: AC_LAUNCH_TILE(EPI_BIAS); break;
            case EPI_BIAS_GELU: AC_LAUNCH_TILE(EPI_BIAS_GELU); break;
            case EPI_BIAS_RELU: AC_LAUNCH_TILE(EPI_BIAS_RELU); break;
            case EPI_BIAS_RES: AC_LAUNCH_TILE(EPI_BIAS_RES); break;
            default: AC_LAUNCH_TILE(EPI_GENERIC); break;
        }
#undef AC_LAUNCH_TILE
#undef AC_LAUNCH_PLANES
    } else {
        dim3 grid((N + 31) / 32, (M + 31) / 32);
        if (a_kmaj && b_kmaj)
            hipLaunchKernelGGL((gemm_direct<true, true>), grid, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);
        else if (a_kmaj && !b_kmaj)
            hipLaunchKernelGGL((gemm_direct<true, false>), grid, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);
        else if (!a_kmaj && b_kmaj)
            hipLaunchKernelGGL((gemm_direct<false, true>), grid, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);
        else
            hipLaunchKernelGGL((gemm_direct<false, false>), grid, dim3(256), 0, stream, A, lda, B, ldb, C, ldc, M, N, K, epi);
    }
    AC_LAUNCH_CHECK();
    return AC_OK;
}

}  // namespace

namespace ac {
static std::atomic<int> g_gemm_arith{-1};
int gemm_arith() {
    if (const int o = call_opts().arith; o >= 0) return o;          // the running call's own option (ac_bert_config.gemm_arith_opt)
    int v = g_gemm_arith.load(std::memory_order_relaxed);
    if (v < 0) {
        v = AC_GEMM_BF16X3;
        if (const char* e = getenv("AC_GEMM_ARITH"))
            v = (strcmp(e, "f32") == 0) ? AC_GEMM_F32 : (strcmp(e, "f16x2") == 0) ? AC_GEMM_F16X2 : AC_GEMM_BF16X3;
        g_gemm_arith.store(v, std::memory_order_relaxed);
    }
    return v;
}
void set_gemm_arith(int v) { g_gemm_arith.store(v, std::memory_order_relaxed); }
static std::atomic<int> g_gemm_variant{-1};
int gemm_variant() {
    int v = g_gemm_variant.load(std::memory_order_relaxed);
    if (v < 0) {
        v = 0;
        if (const char* e = getenv("AC_GEMM_VARIANT")) v = atoi(e);
        g_gemm_variant.store(v, std::memory_order_relaxed);
    }
    return v;
}
bool linear_takes_planes(int M, int N, int K) {
    // mirrors launch_gemm: not the small-M kernel, the LDS-tiled path, split arithmetic
    return arith_split() && !(M <= 64 && N >= 16) && M >= 192 && K >= 32 && (K % 32) == 0 && N >= 1;
}
// internal entry used by head.hip / bert.hip
int linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
               const float* residual, int64_t ldr, float* C, int64_t ldc, int M, int N, int K, int act,
               const uint8_t* mask, float mask_scale, hipStream_t stream, float drop_p, uint64_t drop_seed,
               const uint16_t* Wp, const uint16_t* Ap, uint16_t* Cp, int64_t w_plane_rows) {
    Epilogue e;
    e.bias = bias; e.residual = residual; e.ldr = ldr; e.act = act; e.alpha = 1.f; e.beta = 0.f;
    e.mask = mask; e.mask_scale = mask_scale; e.gate = nullptr; e.ldg = 0; e.gate_scale = 1.f;
    e.drop_p = mask ? 0.f : drop_p; e.drop_seed = drop_seed;
    return launch_gemm(true, true, A, lda, W, ldw, C, ldc, M, N, K, e, stream, Wp, w_plane_rows > 0 ? w_plane_rows : N, Ap, M, Cp);
}
// fp16x2 planes in, ring-staged kernels only (the BERT encoder under AC_GEMM_F16X2; ac_linear_f16x2)
bool linear_f16x2_takes(int M, int N, int K) {
    const int v = gemm_variant();
    return (v == 0 || v >= 1000) && M >= 192 && N >= 8 && (N % 8) == 0 && (K % 32) == 0 && K >= 64;
}
int linear_f16x2(const uint16_t* Ap, const uint16_t* Wp, const float* bias, const float* residual, int64_t ldr, float* C,
                 int64_t ldc, uint16_t* Cp, int M, int N, int K, int act, hipStream_t stream, int64_t w_plane_rows) {
    AC_REQUIRE(Ap && Wp && bias && (C || Cp), AC_EINVAL, "linear_f16x2: null pointer");
    AC_REQUIRE(linear_f16x2_takes(M, N, K), AC_EUNSUPPORTED, "linear_f16x2: %d x %d x %d does not take the ring-staged kernel", M, N, K);
    AC_REQUIRE((act == ACT_NONE || (act == ACT_GELU && Cp)) && !(Cp && residual), AC_EUNSUPPORTED,
               "linear_f16x2: act %d / residual / planes-out combination not built", act);
    Epilogue e;
    e.bias = bias; e.residual = residual; e.ldr = ldr; e.act = act; e.alpha = 1.f; e.beta = 0.f; e.mask = nullptr;
    e.mask_scale = 1.f; e.gate = nullptr; e.ldg = 0; e.gate_scale = 1.f; e.drop_p = 0.f; e.drop_seed = 0;
    const int cls = act == ACT_GELU ? EPI_BIAS_GELU : (residual ? EPI_BIAS_RES : EPI_BIAS);
    const int v = gemm_variant();
    const int cfg = v >= 1000 ? v : pipe_choose_f16(M, N, K);
    return launch_gemm_pipe(cfg, Ap, M, Wp, w_plane_rows > 0 ? w_plane_rows : N, C, ldc, Cp, M, N, K, cls, e, stream, 1);
}
// linear_f32 for shapes with few output tiles: split-K over `scratch` (see gemm_planes_splitk_nt); falls back to
// linear_f32 when the shape has enough tiles, the arithmetic is not bf16x3, or the scratch is too small.
int linear_f32_splitk(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* residual,
                      int64_t ldr, float* C, int64_t ldc, int M, int N, int K, int act, const uint16_t* Wp, float* scratch,
                      size_t scratch_bytes, hipStream_t stream, int64_t w_plane_rows, int* partials_only) {
    // partials_only: the caller finishes a split-K product itself (bert.hip splitk_ln_kernel: slices + bias + residual + LayerNorm in
    // one launch): on return *partials_only = the number of [M, N] slices left in `scratch` and NOTHING was written to C -- or 0, and
    // the whole linear ran as usual
    if (partials_only) *partials_only = 0;
    const int cus = dev_info().cus;
    const int64_t tiles = (int64_t)((M + 63) / 64) * ((N + BN - 1) / BN);
    int ksplit = 1;
    // (the few-tile fp32 kernel runs such a shape in ONE launch; its fp32 matrix pipe is 1/16 of the bf16 one, so only up to the
    //  size of 256 x 768 x 768 -- 8.6 us against 10 + 6 for split-K and its reduce; at the FFN shapes the two are level)
    const bool direct = W && (int64_t)M * N * K <= (int64_t)160 << 20 && fewtiles_takes(M, N, K, (lda % 4) == 0 && (ldw % 4) == 0 && ((((uintptr_t)A) | ((uintptr_t)W)) & 15) == 0);
    if (!direct && Wp && scratch && arith_split() && gemm_variant() == 0 && M >= 65 && M <= 512 && (N % 4) == 0 &&
        (K % 32) == 0 && (lda % 4) == 0 && ((((uintptr_t)A) & 15) == 0) && 2 * tiles <= cus) {
        const int nk = K / SBK;
        // as many slices as fill ~1.5 workgroups per CU, each at least 6 stages long, dividing the stage count
        for (int c = 2; c <= 32; ++c)
            if (nk % c == 0 && nk / c >= 6 && tiles * c <= (int64_t)cus * 3 / 2 && (size_t)c * M * N * sizeof(float) <= scratch_bytes) ksplit = c;
    }
    if (ksplit == 1)
        return linear_f32(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, nullptr, 1.f, stream, 0.f, 0, Wp, nullptr, nullptr, w_plane_rows);
    hipLaunchKernelGGL(gemm_planes_splitk_nt, dim3((unsigned)(tiles * ksplit)), dim3(256), 0, stream, A, lda, Wp,
                       w_plane_rows > 0 ? w_plane_rows : (int64_t)N, scratch, M, N, K, ksplit);
    AC_LAUNCH_CHECK();
    if (partials_only) { *partials_only = ksplit; return AC_OK; }
    Epilogue e;
    e.bias = bias; e.residual = residual; e.ldr = ldr; e.act = act; e.alpha = 1.f; e.beta = 0.f;
    e.mask = nullptr; e.mask_scale = 1.f; e.gate = nullptr; e.ldg = 0; e.gate_scale = 1.f; e.drop_p = 0.f; e.drop_seed = 0;
    const int64_t n4 = (int64_t)M * (N / 4);
    hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, scratch, ksplit, M, N, C, ldc, e);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
int head_backward_pair(const float* dY, int64_t ldy, const float* Aact, int64_t lda_act, const float* W, int64_t ldw, int B,
                       int Hout, int Hin, float gate_scale, float* gW, float* dA, float* gb_in, hipStream_t stream) {
    // problem 1: gW[Hout, Hin] = dY^T Aact (K = B);  problem 2: dA[B, Hin] = (dY W) gated by Aact != 0, K = Hout; gb_in = colsum(dA)
    AC_REQUIRE(B >= 1 && B <= 32, AC_EUNSUPPORTED, "head_backward_pair: B=%d (<= 32)", B);
    DirectProblem p1, p2;
    Epilogue e;
    e.bias = nullptr; e.residual = nullptr; e.ldr = 0; e.act = ACT_NONE; e.alpha = 1.f; e.beta = 0.f; e.mask = nullptr;
    e.mask_scale = 1.f; e.gate = nullptr; e.ldg = 0; e.gate_scale = 1.f; e.drop_p = 0.f; e.drop_seed = 0;
    p1.A = dY; p1.lda = ldy; p1.B = Aact; p1.ldb = lda_act; p1.C = gW; p1.ldc = Hin; p1.M = Hout; p1.N = Hin; p1.K = B; p1.epi = e;
    p1.colsum = nullptr;
    Epilogue g = e;
    g.gate = Aact; g.ldg = lda_act; g.gate_scale = gate_scale;
    p2.A = dY; p2.lda = ldy; p2.B = W; p2.ldb = ldw; p2.C = dA; p2.ldc = Hin; p2.M = B; p2.N = Hin; p2.K = Hout; p2.epi = g;
    p2.colsum = gb_in;
    const int nblk1 = ((Hout + 31) / 32) * ((Hin + 31) / 32), nblk2 = (Hin + 31) / 32;
    hipLaunchKernelGGL(gemm_direct_pair, dim3(nblk1 + nblk2), dim3(256), 0, stream, p1, p2, nblk1);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
int gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda,
             const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const float* gate, int64_t ldg,
             float gate_scale, hipStream_t stream) {
    Epilogue e;
    e.bias = nullptr; e.residual = nullptr; e.ldr = 0; e.act = ACT_NONE; e.alpha = alpha; e.beta = beta;
    e.mask = nullptr; e.mask_scale = 1.f; e.gate = gate; e.ldg = ldg; e.gate_scale = gate_scale;
    e.drop_p = 0.f; e.drop_seed = 0;
    return launch_gemm(transA == 0, transB != 0, A, lda, B, ldb, C, ldc, M, N, K, e, stream);
}
}  // namespace ac

extern "C" int ac_linear_f32(const float* d_A, int64_t lda, const float* d_W, int64_t ldw,
                             const float* d_bias, const float* d_residual, int64_t ldr, float* d_C,
                             int64_t ldc, int M, int N, int K, int act, ac_stream_t stream) {
    AC_REQUIRE(d_A && d_W && d_C, AC_EINVAL, "linear: null pointer");
    AC_REQUIRE(M >= 0 && N >= 0 && K >= 1 && lda >= K && ldw >= K && ldc >= N, AC_EINVAL,
               "linear: bad shape M=%d N=%d K=%d", M, N, K);
    AC_REQUIRE(act >= 0 && act <= 2, AC_EINVAL, "linear: bad activation %d", act);
    return ac::linear_f32(d_A, lda, d_W, ldw, d_bias, d_residual, ldr, d_C, ldc, M, N, K, act, nullptr, 1.f,
                          (hipStream_t)stream);
}

extern "C" int ac_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* d_A,
                           int64_t lda, const float* d_B, int64_t ldb, float beta, float* d_C,
                           int64_t ldc, ac_stream_t stream) {
    AC_REQUIRE(d_A && d_B && d_C, AC_EINVAL, "gemm: null pointer");
    AC_REQUIRE(M >= 0 && N >= 0 && K >= 1, AC_EINVAL, "gemm: bad shape");
    return ac::gemm_f32(transA, transB, M, N, K, alpha, d_A, lda, d_B, ldb, beta, d_C, ldc, nullptr, 0, 1.f,
                        (hipStream_t)stream);
}

extern "C" int ac_gemm_set_arith(int mode) {
    AC_TEST_HOOK_ONLY("ac_gemm_set_arith");
    AC_REQUIRE(mode == AC_GEMM_F32 || mode == AC_GEMM_BF16X3 || mode == AC_GEMM_F16X2, AC_EINVAL, "gemm arith: unknown mode %d", mode);
    ac::set_gemm_arith(mode);
    return AC_OK;
}
extern "C" int ac_gemm_get_arith(void) { return ac::gemm_arith(); }
extern "C" int ac_gemm_set_variant(int v) {
    AC_TEST_HOOK_ONLY("ac_gemm_set_variant");
    AC_REQUIRE(v == 0 || v == 1 || v >= 1000, AC_EINVAL, "gemm variant: unknown value %d", v);
    ac::g_gemm_variant.store(v, std::memory_order_relaxed);
    return AC_OK;
}

extern "C" int ac_split_bf16x3(const float* d_X, int64_t ldx, int64_t rows, int K, uint16_t* d_planes,
                               ac_stream_t stream) {
    AC_REQUIRE(d_X && d_planes, AC_EINVAL, "split: null pointer");
    AC_REQUIRE(rows >= 0 && K >= 8 && (K % 8) == 0 && ldx >= K && (ldx % 4) == 0 && (((uintptr_t)d_X) & 15) == 0 &&
                   (((uintptr_t)d_planes) & 15) == 0,
               AC_EINVAL, "split: K=%d must be a multiple of 8, rows 16-byte aligned", K);
    if (rows == 0) return AC_OK;
    const int64_t units = rows * (K / 8);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_X, ldx, rows, K, d_planes);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_split_f16x2(const float* d_X, int64_t ldx, int64_t rows, int K, int scale_log2, uint16_t* d_planes,
                              ac_stream_t stream) {
    AC_REQUIRE(d_X && d_planes, AC_EINVAL, "split: null pointer");
    AC_REQUIRE(rows >= 0 && K >= 8 && (K % 8) == 0 && ldx >= K && (ldx % 4) == 0 && (((uintptr_t)d_X) & 15) == 0 &&
                   (((uintptr_t)d_planes) & 15) == 0 && scale_log2 >= -14 && scale_log2 <= 15,
               AC_EINVAL, "split: K=%d must be a multiple of 8, rows 16-byte aligned, scale 2^%d", K, scale_log2);
    if (rows == 0) return AC_OK;
    const int64_t units = rows * (K / 8);
    hipLaunchKernelGGL(split_planes_f16_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_X, ldx, rows, K, ldexpf(1.0f, scale_log2), d_planes);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_linear_f16x2(const uint16_t* d_A_planes, const uint16_t* d_W_planes, const float* d_bias,
                               const float* d_residual, int64_t ldr, float* d_C, int64_t ldc, uint16_t* d_C_planes, int M,
                               int N, int K, int act, ac_stream_t stream) {
    AC_REQUIRE(M >= 1 && N >= 1 && K >= 1 && (d_C_planes || ldc >= N) && (!d_residual || ldr >= N), AC_EINVAL,
               "linear_f16x2: bad shape M=%d N=%d K=%d", M, N, K);
    return ac::linear_f16x2(d_A_planes, d_W_planes, d_bias, d_residual, ldr, d_C, ldc, d_C_planes, M, N, K, act,
                            (hipStream_t)stream);
}

extern "C" int ac_linear_bf16x3(const float* d_A, int64_t lda, const uint16_t* d_A_planes, const float* d_W,
                                int64_t ldw, const uint16_t* d_W_planes, const float* d_bias,
                                const float* d_residual, int64_t ldr, float* d_C, int64_t ldc,
                                uint16_t* d_C_planes, int M, int N, int K, int act, ac_stream_t stream) {
    AC_REQUIRE(d_A && d_W && (d_C || d_C_planes), AC_EINVAL, "linear: null pointer");
    AC_REQUIRE(M >= 0 && N >= 0 && K >= 1 && lda >= K && ldw >= K && (d_C_planes || ldc >= N), AC_EINVAL,
               "linear: bad shape M=%d N=%d K=%d", M, N, K);
    AC_REQUIRE(act >= 0 && act <= 3 && (act != 3 || d_C_planes), AC_EINVAL, "linear: bad activation %d", act);
    AC_REQUIRE(!d_A_planes || d_W_planes, AC_EINVAL, "linear: A planes need W planes");
    return ac::linear_f32(d_A, lda, d_W, ldw, d_bias, d_residual, ldr, d_C, ldc, M, N, K, act, nullptr, 1.f,
                          (hipStream_t)stream, 0.f, 0, d_W_planes,
                          (d_A_planes && ac::linear_takes_planes(M, N, K)) ? d_A_planes : nullptr, d_C_planes);
}

/* diagnostic: resident blocks per CU of the LDS-tiled kernels (bias epilogue), by hipOccupancy */
extern "C" int ac_gemm_occupancy(int kernel, int tm, int* blocks_per_cu) {
    AC_REQUIRE(blocks_per_cu && (tm == 1 || tm == 2) && kernel >= 0 && kernel <= 2, AC_EINVAL, "occupancy: bad arguments");
    const void* f = nullptr;
    if (kernel == 0) f = tm == 2 ? (const void*)gemm_tile_nt<EPI_BIAS, 2> : (const void*)gemm_tile_nt<EPI_BIAS, 1>;
    else if (kernel == 1) f = tm == 2 ? (const void*)gemm_split_nt<EPI_BIAS, 2> : (const void*)gemm_split_nt<EPI_BIAS, 1>;
    else f = tm == 2 ? (const void*)gemm_planes_nt<EPI_BIAS, 2, true, false> : (const void*)gemm_planes_nt<EPI_BIAS, 1, true, false>;
    AC_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, f, kTileThreads, 0));
    return AC_OK;
}
