// Shared pieces of the GEMM kernels (gemm.hip, gemm_pipe.hip): the runtime epilogue description, the compile-time
// epilogue classes and the C/D-layout tile stores (fp32 rows, or bf16x3 operand planes of the next GEMM).
// Device code only; include inside an anonymous namespace user.
#pragma once
#include "common.h"

#include <math.h>

namespace acg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_GEGLU32 = 3 };

struct Epilogue {
    const float* bias;      // [N] or null
    const float* residual;  // [M, ldr] or null  (added after the activation)
    int64_t ldr;
    int act;
    float alpha;            // C = alpha * acc (+ beta * Cold) before bias/act
    float beta;
    // inverted-dropout mask applied after the activation (train-mode head): 1 = keep
    const uint8_t* mask;    // [M, N] or null
    float mask_scale;
    // ... or generated in-kernel from a counter-based hash (no mask tensor, no torch RNG launch):
    // keep element (row, col) iff u(drop_seed, row * N + col) >= drop_p
    uint64_t drop_seed;
    float drop_p;
    // relu/dropout backward gate: out = gate[row,col] != 0 ? out * gate_scale : 0
    const float* gate;      // [M, ldg] or null
    int64_t ldg;
    float gate_scale;
};

// EPI_BIAS_RES_LN (gemm_pipe.hip): LayerNorm of the output rows fused into the epilogue.  The tiles of one row panel
// exchange per-row (mean, M2) of their columns through `part` and count themselves into `count[panel]` (zero before the
// launch); C receives the normalised rows (it may alias the residual), `planes` their operand planes for the next GEMM.
struct LnFuse {
    const float* gamma;
    const float* beta;
    float eps;
    float2* part;           // [row panels][column tiles][BM]
    unsigned* count;        // [row panels]
    unsigned* abort_;       // set when a row panel's tiles did not all arrive (the results are NaN then)
    uint16_t* planes;       // [3][N/8][M][8] or null
    int starve;             // test hook (ac_gemm_set_ln_fusion(2)): wait for one arrival more than will ever come
    // 0 (default): the fence-free hand-off -- sc1 (write-through) payload stores, s_waitcnt vmcnt(0), a RELAXED agent-scope arrival, a
    // relaxed poll, sc1 (L2-bypassing) payload loads: nothing of the payload ever sits in a non-coherent cache, so no cache write-back /
    // invalidate is needed (MI355X_MICROARCH.md "handoff-flag"; the grid barrier of grid_sync.h is built the same way).  1
    // (AC_EXCHANGE_FENCES=1, A/B): release on the arrival + acquire after the poll (buffer_wbl2 / buffer_inv: ~3.5 us per exchange).
    int fences;
};

// EPI_QKV_ATTN (gemm_pipe.hip): the [T, 3H] QKV projection with the self-attention of the packed sequences computed in the
// epilogue.  A 256 x 192 output tile is (256 token rows) x (one head's q | k | v): the W rows of tile `head` are q rows
// [64 head, 64 head + 64) of the fused [3H, H] weight, then the same rows of its k and v blocks.  Sequences that lie inside the
// tile's rows are finished there (context rows straight into the operand planes of the output projection); the q | k | v rows of
// a sequence that straddles a row-tile boundary go to `qkv` (fp32 [T, 3H]) for attention_mfma_kernel's boundary mode.
struct AttnFuse {
    const int32_t* cu;      // [b + 1] first row of every sequence, cu[b] = T (ac_bert_pack)
    const int32_t* tile_seq;// [ceil(T / 256)][264]: per row tile, the sequences that start in it and their first rows (qkv_attn_tile_seq)
    int b, H, smax;         // sequences, hidden size (= heads * 64), longest sequence (<= 64)
    float scale;            // 1 / sqrt(head dim)
    uint16_t* ctx_planes;   // [planes][H / 8][T][8]
    float* qkv;             // [T, 3H]: only rows of straddling sequences are written
    // exchange != null (one-round launches only: every tile co-resident, proven by the host before the launch): the tile BELOW a
    // boundary publishes its part of the straddling sequence's rows (sc1 stores into qkv) and sets exchange[tile * heads + head] =
    // epoch; the tile ABOVE waits for that word, pulls the rows into its staging area and finishes the sequence itself -- no
    // follow-up launch.  null: both tiles spill their parts and attention_mfma_kernel's boundary mode serves the sequence.
    unsigned* exchange;
    unsigned epoch;
    int fences;             // as LnFuse::fences
    unsigned* abort_;       // the encoder call's give-up word (shared with the LayerNorm exchange)
};

__device__ __forceinline__ float apply_epilogue(const Epilogue& e, float acc, int64_t row, int col,
                                                const float* C, int64_t ldc, int N) {
    float v = e.alpha * acc;
    if (e.beta != 0.f) v = fmaf(e.beta, C[row * ldc + col], v);
    if (e.bias) v += e.bias[col];
    if (e.act == ACT_RELU) v = v < 0.f ? 0.f : v;            // (torch.relu semantics: NaN stays NaN, unlike fmaxf)
    else if (e.act == ACT_GELU) v = ac::gelu_erf(v);
    if (e.mask) v = e.mask[row * (int64_t)N + col] ? v * e.mask_scale : 0.f;
    else if (e.drop_p > 0.f) v = ac::dropout_keep(e.drop_seed, (uint64_t)(row * (int64_t)N + col), e.drop_p) ? v * e.mask_scale : 0.f;
    if (e.residual) v += e.residual[row * e.ldr + col];
    if (e.gate) v = (e.gate[row * e.ldg + col] != 0.f) ? v * e.gate_scale : 0.f;
    return v;
}

// compile-time epilogue classes for the hot encoder/head shapes; EPI_GENERIC keeps the runtime flags
enum { EPI_GENERIC = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_RES = 3, EPI_BIAS_RELU = 4,
       // GeGLU over 32-column blocks: output columns [64t, 64t+32) are the inputs and [64t+32, 64t+64) the gates of
       // result columns [32t, 32t+32) -- a wave's two 32x32 tiles hold input_j and gate_j in the same lane/register
       EPI_GEGLU32 = 5,
       EPI_IDENT = 6,          // store the accumulators as they are (second half of the fused-LayerNorm epilogue)
       EPI_BIAS_RES_LN = 7,    // bias + residual, then LayerNorm over the whole row (gemm_pipe.hip)
       EPI_QKV_ATTN = 8 };     // the fused QKV projection's tile = one head's q | k | v: self-attention in the epilogue (gemm_pipe.hip)

template <int EPI>
__device__ __forceinline__ float fast_epilogue(float acc, float bias, float res) {
    if (EPI == EPI_IDENT) return acc;
    float v = acc + bias;
    if (EPI == EPI_BIAS_GELU) v = ac::gelu_erf(v);
    if (EPI == EPI_BIAS_RELU) v = v < 0.f ? 0.f : v;
    if (EPI == EPI_BIAS_RES) v += res;
    return v;
}

__device__ __forceinline__ int acc_row32(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// shared epilogue of the LDS-tiled kernels: the 32x32 C/D layout (lane owns column lane & 31, 16 rows)
// (a wave owns TM x TN tiles of 32 x 32: rows m0 + wm * 32 TM ..., columns n0 + wn * 32 TN ...)
template <int EPI, int TM, int BM = 64 * TM, int TN = 2>
__device__ __forceinline__ void store_tile(f32x16 (&acc)[TM][TN], float* __restrict__ C, int64_t ldc, int M, int N,
                                           int m0, int n0, int wm, int wn, int lane, const Epilogue& epi) {
    // epilogue: lane owns column (lane & 31) of each 32x32 tile, 16 rows
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = n0 + wn * (32 * TN) + ni * 32 + (lane & 31);
            if (col >= N) continue;
            const int64_t rbase = m0 + wm * (32 * TM) + mi * 32;
            if (EPI == EPI_GENERIC) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = rbase + acc_row32(r, lane);
                    if (row < M) C[row * ldc + col] = apply_epilogue(epi, acc[mi][ni][r], row, col, C, ldc, N);
                }
            } else {
                const float bias = EPI == EPI_IDENT ? 0.f : epi.bias[col];
                float res[16];
                if (EPI == EPI_BIAS_RES) {   // issue all residual loads first, then compute + store
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int64_t row = rbase + acc_row32(r, lane);
                        if (row > M - 1) row = M - 1;
                        res[r] = epi.residual[row * epi.ldr + col];
                    }
                }
                if (m0 + BM <= M) {          // block-uniform: interior tile, branch-free stores
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        C[(rbase + acc_row32(r, lane)) * ldc + col] =
                            fast_epilogue<EPI>(acc[mi][ni][r], bias, EPI == EPI_BIAS_RES ? res[r] : 0.f);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = rbase + acc_row32(r, lane);
                        const float v = fast_epilogue<EPI>(acc[mi][ni][r], bias, EPI == EPI_BIAS_RES ? res[r] : 0.f);
                        if (row < M) C[row * ldc + col] = v;
                    }
                }
            }
        }
}

// epilogue that emits the result as operand planes of the NEXT GEMM (C[M,N] -> planes[p][n/8][row][n%8]).
// In the C/D layout a lane owns one column, but a plane k-slot is 8 neighbouring columns of one row: each
// wave transposes its 32x32 tiles through a private LDS scratch (the staging buffers are idle by now), so
// every lane ends up with 8 consecutive columns of a row = one split8 and three 16-byte stores.
constexpr int kTrLd = 36;                                  // padded row of the 32x32 transpose scratch (floats)
constexpr int kTrFloats = 32 * kTrLd;                      // per wave

template <int EPI, int TM, int TN = 2, int AR = 3>
__device__ __forceinline__ void store_tile_planes(f32x16 (&acc)[TM][TN], uint16_t* __restrict__ Cp, int M, int N,
                                                  int m0, int n0, int wm, int wn, int lane, const Epilogue& epi,
                                                  float* scratch /* this wave's kTrFloats floats of LDS */) {
    constexpr bool GLU = EPI == EPI_GEGLU32;
    static_assert(!GLU || TN == 2, "the fused GeGLU epilogue pairs the two column tiles of a 64-column wave tile");
    const int NO = GLU ? N / 2 : N;                                  // result columns
    const int64_t plane = (int64_t)M * NO;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < (GLU ? 1 : TN); ++ni) {
            const int c0 = GLU ? n0 / 2 + wn * 32 : n0 + wn * (32 * TN) + ni * 32;      // first result column of the tile
            const int col = n0 + wn * (32 * TN) + ni * 32 + (lane & 31);                  // GEMM column of acc[mi][ni]
            const int64_t rbase = m0 + wm * (32 * TM) + mi * 32;
            const float bias = (EPI != EPI_IDENT && col < N) ? epi.bias[col] : 0.f;
            const float bias_g = (GLU && col + 32 < N) ? epi.bias[col + 32] : 0.f;
            if constexpr (EPI == EPI_BIAS_GELU) {                    // two accumulators per packed-fp32 evaluation (ac::gelu_erf2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const ac::gelu_f32x2 xv = {acc[mi][ni][r] + bias, acc[mi][ni][r + 1] + bias};
                    const ac::gelu_f32x2 gv = ac::gelu_erf2(xv);
                    scratch[acc_row32(r, lane) * kTrLd + (lane & 31)] = gv[0];
                    scratch[acc_row32(r + 1, lane) * kTrLd + (lane & 31)] = gv[1];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v;
                    if (GLU) {
                        const float x = acc[mi][0][r] + bias;
                        v = ac::gelu_erf(x) * (acc[mi][1][r] + bias_g);
                    } else {
                        v = fast_epilogue<EPI>(acc[mi][ni][r], bias, 0.f);
                    }
                    scratch[acc_row32(r, lane) * kTrLd + (lane & 31)] = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int rr = (lane >> 2) + 16 * u, q = lane & 3;       // row of the tile, k-slot of 8 columns
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(scratch + rr * kTrLd + 8 * q);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(scratch + rr * kTrLd + 8 * q + 4);
                const int64_t row = rbase + rr;
                const int cq = c0 + 8 * q;
                if (row < M && cq < NO)                              // (N % 8 == 0: checked at launch)
                    ac::emit_planes8(Cp + ac::plane_off(M, row, cq), plane, v0, v1, AR == 2);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
}


typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// Workgroups are dealt round-robin to the 8 XCDs (workgroup id % 8), each with a private 4 MB L2.  Map the
// hardware id to a logical tile id so that every XCD owns one CONTIGUOUS range of tiles (column tiles
// fastest): its resident blocks then share A row-panels and sweep the W panels together, instead of every
// XCD touching every A panel.  Bijective for any block count (the guide's q/r formula).
__device__ __forceinline__ int xcd_tile_id(int wg, int nwg) {
    const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
}

}  // namespace acg

namespace ac {
// gemm_pipe.hip: the planes GEMM with its operand stages in an LDS ring (counted vmcnt, raw barrier)
bool pipe_takes(int M, int N, int K, int cls, bool c_planes);
int pipe_choose(int M, int N, int K, int cls, bool c_planes);   // 0 = keep the two-buffer tile kernels of gemm.hip
int pipe_choose_f16(int M, int N, int K);                        // fp16x2 operands (never 0: there is no other kernel for them)
int launch_gemm_pipe(int cfg, const uint16_t* Ap, int64_t a_rows, const uint16_t* Wp, int64_t w_rows, float* C, int64_t ldc,
                     uint16_t* Cp, int M, int N, int K, int cls, const acg::Epilogue& epi, hipStream_t stream, int f16 = 0);
int gemm_variant();            // diagnostic switch (ac_gemm_set_variant): 0 = default dispatch, 1 = two-buffer kernels only, >= 1000 = one ring configuration
}  // namespace ac
