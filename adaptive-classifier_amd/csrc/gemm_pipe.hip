// gemm_pipe_nt: the planes GEMM of the encoder (transformers BertModel projections, called at
// /root/reference/src/adaptive_classifier/classifier.py:1271) with the operand stages held in an LDS RING instead of two
// buffers: the global_load_lds DMA of stage s + NS - 1 is issued while stage s is multiplied, the ring is drained with
// COUNTED s_waitcnt vmcnt(N) + a raw s_barrier (no vmcnt(0) in the loop), and (PIPE) every wave reads the fragments
// of stage s + 1 into a second register set under the MFMAs of stage s, the reads pinned between the MFMAs (PIPE = 2).
//
// Why (profiles/r03/gemm_sweep*.txt): per-workgroup clock stamps show the k-loop of these kernels keeping the matrix pipe
// 76 - 94 % busy; what the encoder's GEMMs at ~5000 packed token rows lose is ROUND QUANTISATION (738 tiles of 128 x 128 on
// 2 x 256 resident workgroups = 1.44 rounds) and the prologue / epilogue each round exposes.  So the tile is chosen per
// shape to make ONE full round of one workgroup per CU: 256 x 192 for the QKV projection (252 tiles), 256 x 256 for FFN1
// (252 tiles), 128 x 128 with 8 waves for the N = 768 GEMMs (246 tiles) -- pipe_choose().
//
// Tile family: (32 TM WMW) x (32 TN WNW), WMW x WNW waves of (32 TM) x (32 TN); 16-k stages of three bf16 planes per
// operand; six products smallest first, fp32 accumulate -- per output element the same products in the same order as
// gemm_planes_nt (gemm.hip), hence bit-identical results (tests/test_gemm_split_gpu.py).
//
// AR (round 4) = operand planes per stage: 3 = the bf16x3 split above; 2 = the opt-in fp16x2 form (AC_GEMM_F16X2: two fp16 terms of
// x 2^s per operand, three products l.h + h.l + h.h on v_mfma_f32_32x32x16_f16, accumulator x 2^-16 before the epilogue) -- same
// ring, same loop, two thirds of the bytes and half the matrix-pipe work per stage.
//
// Measured and dropped (profiles/r03/gemm_sweep2.txt): a wave-specialised form (4 loader waves issuing the DMA, the others
// only reading fragments and multiplying) -- equal or slower on every shape, i.e. the DMA issue cost inside the consumer
// waves is not what limits the loop.
#include "common.h"
#include "gemm_common.h"
#include "grid_sync.h"
#include "attention_core.h"

#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

namespace {
using namespace acg;

constexpr int PSBK = 16;                 // k per stage

struct PipeParams {
    const uint16_t* Ap; int64_t a_rows;
    const uint16_t* Wp; int64_t w_rows;
    float* C; int64_t ldc;              // fp32 result, or (C_PLANES) the planes of the next GEMM's operand
    int M, N, K;
    Epilogue epi;
    LnFuse ln;                          // EPI_BIAS_RES_LN only
    AttnFuse at;                        // EPI_QKV_ATTN only
    unsigned long long* stamps;         // diagnostic (ac_gemm_debug_stamps): 4 shader-clock stamps per workgroup, or null
    int krot;                           // experiment: XCD x starts its k-loop at stage x nk / 8 and wraps (ac_gemm_set_krot)
};

// shader-clock stamp `i` of this workgroup (wave 0 only; a wave-uniform branch on a kernel argument)
template <int STRIDE = 4>
__device__ __forceinline__ void stamp(const PipeParams& prm, int wave, int i) {
    if (prm.stamps && wave == 0) {
        const unsigned long long t = clock64();
        if ((threadIdx.x & 63) == 0) prm.stamps[(size_t)blockIdx.x * STRIDE + i] = t;
    }
}

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// AR = operand planes per stage: 3 = bf16x3 (six bf16 products), 2 = fp16x2 (three fp16 products, AC_GEMM_F16X2)
template <int TM, int TN, int WMW, int WNW, int NS, int AR = 3> struct PipeGeom {
    static constexpr int BM = 32 * TM * WMW, BN = 32 * TN * WNW;
    static constexpr int RA = BM / 32, RW = BN / 32, RG = RA + RW;      // 32-row groups per stage
    static constexpr int NP = AR * RG;                                   // 1 KB DMA pieces per stage
    static constexpr int NW = WMW * WNW;                                 // waves
    static constexpr int PPW = (NP + NW - 1) / NW;                       // pieces per wave and stage (excess = duplicates)
    static constexpr int SLOT = AR * RG * 64;                            // uint4 per ring slot
    static constexpr int LDS_BYTES = NS * SLOT * 16;
    static constexpr int TR_BYTES = NW * kTrFloats * 4;                  // transpose scratch of the planes epilogue
    static constexpr int BPC_LDS = (160 * 1024) / (LDS_BYTES > TR_BYTES ? LDS_BYTES : TR_BYTES);
    static constexpr int BPC = BPC_LDS < 1 ? 1 : (BPC_LDS * NW > 16 ? (16 / NW < 1 ? 1 : 16 / NW) : BPC_LDS);   // <= 4 waves per SIMD wanted
    static constexpr int WAVES_PER_SIMD = (BPC * NW + 3) / 4;
};

// the residual block of this wave's accumulators, in their layout (issued at kernel start, consumed by store_tile_ln)
template <int TM, int TN, int WMW, int WNW>
__device__ __forceinline__ void ln_prefetch_residual(float (&res)[TM * TN * 16], const PipeParams& prm, int m0, int n0, int wm,
                                                     int wn, int lane) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = n0 + wn * (32 * TN) + ni * 32 + (lane & 31);
            const int64_t rbase = m0 + wm * (32 * TM) + mi * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int64_t row = rbase + acc_row32(r, lane);
                if (row > prm.M - 1) row = prm.M - 1;
                res[(mi * TN + ni) * 16 + r] = prm.epi.residual[row * prm.epi.ldr + col];
            }
        }
}

// ---- EPI_BIAS_RES_LN: y = acc + bias + residual, LayerNorm over the whole row, result as fp32 rows AND operand planes ----
// A row of the output spans the N / BN tiles of its row panel; each tile reduces its BN columns to a per-row (mean, M2),
// publishes them (sc1 stores), counts itself into the panel's counter and waits for the others -- all tiles of a panel are
// co-resident because the launch is ONE round of one workgroup per CU (launch_gemm_pipe_ln refuses anything else; two such launches interleaved on one
// device by two processes could still starve each other: the wait is bounded and the rows turn NaN, which the host reports) -- then
// combines the partials (Chan et al.: equal counts), normalises its accumulators in place and stores them twice.
// Replaces a separate LayerNorm launch (read fp32 y, write fp32 x + planes: 20 us at 5141 x 768) by ~3 us of exchange.
template <int TM, int TN, int WMW, int WNW, int AR>
__device__ __forceinline__ void store_tile_ln(f32x16 (&acc)[TM][TN], const float (&res)[TM * TN * 16], const PipeParams& prm,
                                              int bm, int bn, int ntn, int wm, int wn, int lane, int tid, int wave, float* lds_f) {
    constexpr int BM = 32 * TM * WMW, BN = 32 * TN * WNW, NW = WMW * WNW, WC = 32 * TN;
    static_assert(BM <= 64 * NW, "one thread per tile row in the combine steps");
    const Epilogue& e = prm.epi;
    const LnFuse& ln = prm.ln;
    float2* wstat = reinterpret_cast<float2*>(lds_f + NW * kTrFloats);      // [WNW][BM]: (mean, M2) over a wave's WC columns
    float2* rstat = wstat + WNW * BM;                                       // [BM]: (mean, rstd) of the whole row
    unsigned* okf = reinterpret_cast<unsigned*>(rstat + BM);
    const int m0 = bm * BM, n0 = bn * BN, c = lane & 31;
    // 1. v = (acc + bias) + residual -- the operation order of the unfused epilogue; the residual was requested before the
    //    k-loop (ln_prefetch_residual), so its HBM latency is not part of this exposed epilogue
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const float bias = e.bias[n0 + wn * WC + ni * 32 + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = (acc[mi][ni][r] + bias) + res[(mi * TN + ni) * 16 + r];
        }
    // 2. per row: mean and M2 over this wave's WC columns (two passes over registers; 32 lanes of a half hold one row)
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.f;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) s += acc[mi][ni][r];
            s = acp::row16_sum(s);
            s += __shfl_xor(s, 16);
            const float mw = s * (1.0f / WC);
            float q = 0.f;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) { const float d = acc[mi][ni][r] - mw; q = fmaf(d, d, q); }
            q = acp::row16_sum(q);
            q += __shfl_xor(q, 16);
            if (c == r) wstat[wn * BM + wm * (32 * TM) + mi * 32 + acc_row32(r, lane)] = make_float2(mw, q);
        }
    __syncthreads();
    // 3. the tile's partial of every row -> global (sc1: visible to the other XCDs without cache maintenance); count in
    if (tid < BM) {
        float mean = 0.f;
#pragma unroll
        for (int w = 0; w < WNW; ++w) mean += wstat[w * BM + tid].x;
        mean *= 1.0f / WNW;
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < WNW; ++w) { const float2 pw = wstat[w * BM + tid]; const float d = pw.x - mean; m2 += fmaf((float)WC * d, d, pw.y); }
        const unsigned long long bits = (unsigned long long)__float_as_uint(mean) | ((unsigned long long)__float_as_uint(m2) << 32);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(ln.part + ((size_t)bm * ntn + bn) * BM + tid), bits,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the workgroup's partial stores are write-through and acknowledged (vmcnt(0) above, every wave; the barrier orders them before
    // thread 0's arrival): a relaxed arrival is enough -- see LnFuse::fences
    if (tid == 0) {
        if (ln.fences) __hip_atomic_fetch_add(ln.count + bm, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(ln.count + bm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // 4. wait for the panel's other tiles (bounded: a bug or a CU mask must not hang the GPU -- the rows become NaN instead)
    if (tid < 64) {
        // (~1 us per poll: gives up after about a second; once one panel has given up -- the flag is per encoder call -- the
        //  later launches of that call stop waiting at their first look at it instead of a second each)
        unsigned ok = 1;
        for (long spins = 0;; ++spins) {
            const unsigned v = __hip_atomic_load(ln.count + bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v >= (unsigned)(ntn + ln.starve)) {
                if (ln.fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (the partial loads below are sc1: never served from a cache)
                else asm volatile("" ::: "memory");                                    // (compiler: keep them after the poll)
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63) == 63 && (spins > (1l << 20) || __hip_atomic_load(ln.abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (lane == 0) __hip_atomic_store(ln.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        if (lane == 0) *okf = ok;
    }
    __syncthreads();
    // 5. whole-row statistics from the ntn partials (<= 8 column tiles: all loads in flight together)
    if (tid < BM) {
        float2 pj[8];
        const float* base = reinterpret_cast<const float*>(ln.part + (size_t)bm * ntn * BM + tid);
#pragma unroll
        for (int j = 0; j < 8; ++j) pj[j] = j < ntn ? acp::ld2_sc1(base + (size_t)j * BM * 2) : make_float2(0.f, 0.f);
        float mean = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) mean += pj[j].x;                         // (absent tiles add 0)
        mean /= (float)ntn;
        float m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < ntn) { const float d = pj[j].x - mean; m2 += fmaf((float)BN * d, d, pj[j].y); }
        const float var = m2 / (float)prm.N;
        const float rstd = 1.0f / sqrtf(var + ln.eps);
        rstat[tid] = make_float2(mean, *okf ? rstd : __uint_as_float(0x7fc00000u));
    }
    __syncthreads();
    // 6. normalise in place
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = n0 + wn * WC + ni * 32 + c;
            const float g = ln.gamma[col], b = ln.beta[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 st = rstat[wm * (32 * TM) + mi * 32 + acc_row32(r, lane)];
                acc[mi][ni][r] = (acc[mi][ni][r] - st.x) * st.y * g + b;
            }
        }
    // 7. fp32 rows (later residuals) and the operand planes of the next GEMM
    store_tile<EPI_IDENT, TM, BM, TN>(acc, prm.C, prm.ldc, prm.M, prm.N, m0, n0, wm, wn, lane, e);
    if (ln.planes)
        store_tile_planes<EPI_IDENT, TM, TN, AR>(acc, ln.planes, prm.M, prm.N, m0, n0, wm, wn, lane, e, lds_f + wave * kTrFloats);
}

// ---- EPI_QKV_ATTN: self-attention of the packed sequences in the epilogue of the QKV projection ----
// The tile is 256 token rows x (q | k | v of ONE head).  Staged as fp32 rows in LDS (the DMA ring is idle by now), a sequence's
// attention is exactly what attention_mfma_kernel does from HBM -- acattn::attention_tile, same instructions, same order, one wave
// per (sequence, 32-query tile) -- so the context rows are bit-identical to the two-launch route, and the 47 MB fp32 qkv round
// trip (written by this epilogue, read back by the attention launch) and the attention launch itself (22 us per layer at 5141
// rows, matrix pipe idle) are gone.  The whole tile is 196 KB of fp32, the LDS 160 KB: it is staged in 2 (longest sequence <= 32)
// or 3 (<= 64) overlapping passes of 160 rows, pass p serving the sequences that START in its first 128 / 96 rows (they end
// inside its 160).  A sequence that straddles a 256-row tile boundary cannot be finished by either tile: both write their part
// of its q | k | v rows to the fp32 qkv buffer and attention_mfma_kernel's boundary mode (one wave per boundary, head and query
// tile, <= 20 sequences at 5141 rows) serves it afterwards.
constexpr int kAtLd = 196;              // floats per staged row: 192 + 4 (row stride 784 B = 16 B mod 256: the 16 lanes of a b128 read hit distinct banks)
constexpr int kAtRows = 160;            // staged rows per pass
constexpr int kAtCu = ac::kQkvAttnCu;             // the tile's sequence offsets (<= 257: sequences of >= 1 row starting inside 256 rows, + the end)
constexpr int kAtBytes = kAtRows * kAtLd * 4 + kAtCu * 4;

template <int TM, int TN, int WMW, int WNW, int AR>
__device__ __forceinline__ void qkv_attention_epilogue(f32x16 (&acc)[TM][TN], const PipeParams& prm, int bm, int head, int wm, int wn,
                                                       int lane, int wave, float* T, int cu_pref, const float (&bias)[TN]) {
    static_assert(TM == 2 && TN == 3 && WMW == 4 && WNW == 2, "built for the 256 x 192 tile (8 waves of 64 x 96)");
    constexpr int BM = 256, NW = WMW * WNW;
    const AttnFuse& at = prm.at;
    const int m0 = bm * BM, H = at.H, c = lane & 31;
    const int stride = at.smax <= 32 ? 128 : 96;                        // + round_up(smax, 32) = 160 staged rows
    const int tile_end = m0 + BM < prm.M ? m0 + BM : prm.M;
    // the offsets of the sequences that start inside this tile (+ the one after them) -> LDS, ONE dependent global load for the
    // whole epilogue (a binary search over cu in HBM / L2 per pass and two loads per sequence cost ~5 k cycles per tile)
    // (the tile's row of the per-forward table -- [0] = sequences that start in the tile, [1 + i] = the first row of the i-th of them,
    //  one more = the end of the last -- was requested before the k-loop, one word per thread: no global latency in this epilogue)
    int* cuT = reinterpret_cast<int*>(T + kAtRows * kAtLd);
    if (wave * 64 + lane < kAtCu) cuT[wave * 64 + lane] = cu_pref;
    const int* cuL = cuT + 1;
    // rows [row_a, row_b) of the tile -> their place in the fp32 qkv buffer (q | k | v blocks H apart); one wave: lanes 0 .. 47 move one
    // 16-byte piece of a row each (part = lane / 16).  With the in-launch exchange the stores are sc1 (write-through: visible across
    // the XCDs' L2s) THROUGH A BUFFER DESCRIPTOR -- ordinary stores to the compiler, issued back to back; as relaxed atomics they were
    // kept in program order with a wait after each one, ~0.7 us per access on the critical path of the tile's last wave.
    const bool xchg = at.exchange != nullptr;
    const __amdgpu_buffer_rsrc_t rq = acp::make_rsrc(at.qkv, 0xffffffffu);
    const int xpart = lane >> 4, xc4 = lane & 15;
    auto qkv_byte_off = [&](int row) { return (unsigned)((((int64_t)row * 3 + xpart) * H + head * 64 + 4 * xc4) * 4); };
    auto spill_rows = [&](int row_a, int row_b, int lo_r) {
        if (lane < 48)
            for (int row = row_a; row < row_b; ++row) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(T + (row - m0 - lo_r) * kAtLd + xpart * 64 + 4 * xc4);
                if (xchg) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(acp::u32x4_t, v), rq, qkv_byte_off(row), 0, 16);
                else *reinterpret_cast<f32x4*>(at.qkv + (int64_t)row * (3 * H) + xpart * H + head * 64 + 4 * xc4) = v;
            }
    };
    // the rows [row_a, row_b) of the NEXT row tile (published there by its top-part spill) -> this pass's staging rows; one wave
    auto pull_rows = [&](int row_a, int row_b, int lo_r) -> bool {
        unsigned* flag = at.exchange + (size_t)(bm + 1) * (prm.N / 192) + head;
        unsigned ok = 1;
        for (long spins = 0;; ++spins) {
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == at.epoch) break;
            __builtin_amdgcn_s_sleep(1);
            // (the neighbour is resident -- the host proved it before choosing this form -- and publishes right after its k-loop,
            //  long before this point; the bound is an assertion against a device shared with another process, not a protocol step)
            if ((spins & 63) == 63 && (spins > (1l << 20) || __hip_atomic_load(at.abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (lane == 0) __hip_atomic_store(at.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        if (at.fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // (the row loads below are sc1: never served from a cache)
        else asm volatile("" ::: "memory");                                        // (compiler: keep them after the poll)
        if (lane < 48) {
            constexpr int RB = 8;                                       // rows in flight per batch of sc1 (L2-bypassing) loads
            for (int r0b = row_a; r0b < row_b; r0b += RB) {
                acp::u32x4_t v[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int row = r0b + u < row_b ? r0b + u : row_b - 1;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rq, qkv_byte_off(row), 0, 16);
                }
#pragma unroll
                for (int u = 0; u < RB; ++u)
                    if (r0b + u < row_b) {
                        const acp::u32x4_t w = ok ? v[u] : acp::u32x4_t{0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u};
                        *reinterpret_cast<acp::u32x4_t*>(T + (r0b + u - m0 - lo_r) * kAtLd + xpart * 64 + 4 * xc4) = w;
                    }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return ok != 0;
    };
    // diagnostic (ac_gemm_debug_stamps + AC_GEMM_STAMP_EPI=8): wave 0's clock at every pass boundary, slots 4 + 3 pass + {0, 1, 2} of
    // this workgroup's 16 (slots 0 .. 3 are the kernel's own: start, ring filled, k-loop done, end)
    // (compiled only into measurement builds, -DAC_QKV_ATTN_STAMPS: the kernel sits at its register limit and the extra live values
    //  of the stamps make hipcc spill -- tests/test_kernel_resources_cpu.py refuses scratch in the shipping build)
    auto xstamp = [&](int i) {
#ifdef AC_QKV_ATTN_STAMPS
        if (prm.stamps && wave == 0 && lane == 0 && i < 16) prm.stamps[(size_t)blockIdx.x * 16 + i] = clock64();
#else
        (void)i;
#endif
    };
    int pass_i = 0;
    for (int lo_r = 0; lo_r < BM; lo_r += stride, ++pass_i) {
        if (m0 + lo_r >= prm.M) break;                                  // (workgroup-uniform: the ragged last tile)
        // this pass's rows of the accumulators (+ bias: the operation order of the unfused epilogue) -> LDS, row-major
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int rb = wm * (32 * TM) + mi * 32;                    // first tile row of the 32-row block (wave-uniform)
            if (rb >= lo_r && rb < lo_r + kAtRows) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        T[(rb - lo_r + acc_row32(r, lane)) * kAtLd + wn * 96 + ni * 32 + c] = acc[mi][ni][r] + bias[ni];
            }
        }
        __syncthreads();
        xstamp(4 + 3 * pass_i);
        const int nseq = cuT[0];
        const int seq_lo = m0 + lo_r;
        const int seq_hi = seq_lo + stride < tile_end ? seq_lo + stride : tile_end;
        int s0 = 0;                                                     // first of the tile's sequences that starts at or after seq_lo
        {
            int hi = nseq;
            while (s0 < hi) { const int mid = (s0 + hi) >> 1; if (cuL[mid] < seq_lo) s0 = mid + 1; else hi = mid; }
        }
        if (lo_r == 0 && wave == NW - 1) {                              // the part of a sequence that began in the previous tile
            const int top_end = cuL[0] < tile_end ? cuL[0] : tile_end;  // (cuL[0] = cu[sb] >= m0; = T when no sequence starts at or after m0)
            spill_rows(m0, top_end, lo_r);
            if (xchg) {                                                 // publish: the rows above, then the word the tile above polls
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {                                        // (the rows are write-through and acknowledged: AttnFuse::fences)
                    if (at.fences) __hip_atomic_store(at.exchange + (size_t)bm * (prm.N / 192) + head, at.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    else __hip_atomic_store(at.exchange + (size_t)bm * (prm.N / 192) + head, at.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        for (int s = s0 + wave; s < nseq; s += NW) {
            const int r0 = cuL[s];
            if (r0 >= seq_hi) break;
            const int r1 = cuL[s + 1];
            bool whole = r1 <= m0 + BM;                                 // the whole sequence is in this tile (and in this pass's rows)
            if (!whole && xchg) {                                       // it continues in the next tile: fetch the rest and finish it here
                pull_rows(tile_end, r1, lo_r);
                whole = true;
            }
            if (whole) {
                const float* qb = T + (r0 - seq_lo) * kAtLd;
                const int S = r1 - r0;
                for (int qt = 0; qt * 32 < S; ++qt)
                    acattn::attention_tile<false, 64, false>(qb, qb + 64, qb + 128, kAtLd, S, qt, lane, at.scale, nullptr, nullptr, nullptr, -1,
                                                             nullptr, H, at.ctx_planes, prm.M, r0, head * 64, AR == 2);
            } else {
                spill_rows(r0, tile_end, lo_r);                         // it continues in the next tile: the boundary launch serves it
            }
        }
        xstamp(5 + 3 * pass_i);
        __syncthreads();                                                // (the next pass overwrites the staging rows)
        xstamp(6 + 3 * pass_i);
    }
}

template <int EPI, int TM, int TN, int WMW, int WNW, int NS, bool C_PLANES, int PIPE, int AR = 3>
__global__ __launch_bounds__(64 * WMW * WNW, (PipeGeom<TM, TN, WMW, WNW, NS, AR>::WAVES_PER_SIMD)) void gemm_pipe_nt(PipeParams prm) {
    using G = PipeGeom<TM, TN, WMW, WNW, NS, AR>;
    static_assert(AR == 3 || AR == 2, "operand planes: 3 = bf16x3, 2 = fp16x2");
    constexpr int BM = G::BM, BN = G::BN, RA = G::RA, RG = G::RG, NP = G::NP, NW = G::NW, PPW = G::PPW, SLOT = G::SLOT;
    static_assert(PIPE == 0 || NS >= 3, "the software-pipelined loop needs a ring of >= 3 stages");
    static_assert(NS >= 2, "ring depth");
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];       // [NS][AR planes][RG groups][64 lanes]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int ntn = (prm.N + BN - 1) / BN;
    const int tile = xcd_tile_id(blockIdx.x, gridDim.x);
    const int bn = tile % ntn, bm = tile / ntn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nk = prm.K / PSBK;
    stamp<(EPI == EPI_QKV_ATTN ? 16 : 4)>(prm, wave, 0);
    float lnres[EPI == EPI_BIAS_RES_LN ? TM * TN * 16 : 1];
    if constexpr (EPI == EPI_BIAS_RES_LN) ln_prefetch_residual<TM, TN, WMW, WNW>(lnres, prm, m0, n0, wm, wn, lane);
    // EPI_QKV_ATTN: what the epilogue needs from global memory -- this thread's word of the tile's sequence table and its three bias
    // values -- is requested here, ahead of every DMA (older than them on the vmcnt queue), so the epilogue starts without a miss
    int at_cu_pref = 0;
    float at_bias[EPI == EPI_QKV_ATTN ? TN : 1];
    if constexpr (EPI == EPI_QKV_ATTN) {
        if (tid < kAtCu) at_cu_pref = prm.at.tile_seq[(size_t)bm * kAtCu + tid];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = wn * (32 * TN) + ni * 32 + (lane & 31);          // column of the head's q | k | v block
            at_bias[ni] = prm.epi.bias[(col >> 6) * prm.at.H + bn * 64 + (col & 63)];
        }
    }

    // ---- DMA stream: piece j = wave + NW t -> (plane j / RG, group j % RG); lane (i, kg) copies 16 B of row i, k-slot kg
    const uint16_t* pp[PPW];
    const int64_t a_step = 2 * prm.a_rows * 8, w_step = 2 * prm.w_rows * 8;
    {
        const int i32 = lane & 31, kg = lane >> 5;
        const int64_t a_plane = prm.a_rows * (int64_t)prm.K, w_plane = prm.w_rows * (int64_t)prm.K;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const int j = (wave + NW * t) % NP, p = j / RG, g = j % RG;
            if (g < RA) {
                int row = m0 + 32 * g + i32; if (row > prm.M - 1) row = prm.M - 1;
                pp[t] = prm.Ap + p * a_plane + ((int64_t)kg * prm.a_rows + row) * 8;
            } else {
                int row;
                if constexpr (EPI == EPI_QKV_ATTN) {                    // column tile `bn` = head bn: its q rows, then its k rows, then its v rows
                    const int gg = g - RA;
                    row = (gg >> 1) * prm.at.H + bn * 64 + (gg & 1) * 32 + i32;
                } else {
                    row = n0 + 32 * (g - RA) + i32; if (row > prm.N - 1) row = prm.N - 1;
                }
                pp[t] = prm.Wp + p * w_plane + ((int64_t)kg * prm.w_rows + row) * 8;
            }
        }
    }
    int iss = 0;                                                        // next stage to issue
    // Rotated k order (experiment, ac_gemm_set_krot(1), default off): workgroups on XCD x start at stage x nk / 8 and wrap, so the
    // 8 XCDs fetch different slices of the operands at any moment.  Measured in the encoder (profiles/r04/krot_ab_base.txt):
    // 4.89 vs 4.91 ms bf16x3, 3.23 vs 3.24 ms fp16x2 -- nothing, once the ring really runs ahead (it had looked like -8 % on a
    // build whose ring was drained every stage by a compiler-inserted vmcnt(0): DESIGN 2.3g).  In-order keeps every output
    // element the same sum in the same order whatever the tile or the batch.
    int a_cur = prm.krot ? ((int)(blockIdx.x & 7) * nk / 8) : 0;         // the stage the pointers are at (wave-uniform)
    if (a_cur) {
#pragma unroll
        for (int t = 0; t < PPW; ++t) pp[t] += (int64_t)a_cur * (((wave + NW * t) % NP) % RG < RA ? a_step : w_step);
    }
    const int64_t a_back = (int64_t)(nk - 1) * a_step, w_back = (int64_t)(nk - 1) * w_step;
    auto issue = [&]() {                                                // always PPW DMA instructions (exact vmcnt accounting)
        const int slot = iss % NS;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const int j = (wave + NW * t) % NP, p = j / RG, g = j % RG;
            __builtin_amdgcn_global_load_lds((glb_void_t*)pp[t], (lds_void_t*)&lds[slot * SLOT + (p * RG + g) * 64], 16, 0, 0);
        }
        // next stage, wrapping to stage 0 after the last one -- one branch-free form for both k orders (scalar selects): in
        // order the wrap falls on the first over-issued tail stage, whose contents nobody reads
        const bool wrap = a_cur + 1 == nk;
        a_cur = wrap ? 0 : a_cur + 1;
        const int64_t da = wrap ? -a_back : a_step, dw = wrap ? -w_back : w_step;
#pragma unroll
        for (int t = 0; t < PPW; ++t) pp[t] += ((wave + NW * t) % NP) % RG < RA ? da : dw;
        ++iss;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // (the fragments are kept in the MFMA's own operand type, converted AT the LDS load: with them held as uint4 and converted at
    //  the MFMA, hipcc put an s_waitcnt vmcnt(0) in front of every stage's first ds_read -- it no longer proved the reads
    //  independent of the LDS-DMA writes in flight -- which drained the ring every stage: encode 4.96 -> 6.09 ms)
    using frag_t = std::conditional_t<AR == 3, bf16x8_t, f16x8_t>;
    struct Frags { frag_t a[TM][AR], b[TN][AR]; };
    auto read_frags = [&](Frags& F, int slot) {
        const uint4* base = lds + slot * SLOT + lane;
#pragma unroll
        for (int p = 0; p < AR; ++p) {
#pragma unroll
            for (int a = 0; a < TM; ++a) F.a[a][p] = __builtin_bit_cast(frag_t, base[(p * RG + TM * wm + a) * 64]);
#pragma unroll
            for (int b = 0; b < TN; ++b) F.b[b][p] = __builtin_bit_cast(frag_t, base[(p * RG + RA + TN * wn + b) * 64]);
        }
    };
    constexpr int NPROD = AR == 3 ? 6 : 3;
    auto mfmas = [&](const Frags& F) {
        // smallest products first: bf16x3 (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); fp16x2 (l,h) (h,l) (h,h)
        constexpr int PAIRS3[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
        constexpr int PAIRS2[3][2] = {{1, 0}, {0, 1}, {0, 0}};
#pragma unroll
        for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    if constexpr (AR == 3)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[a][PAIRS3[pr][0]], F.b[b][PAIRS3[pr][1]], acc[a][b], 0, 0, 0);
                    else
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[a][PAIRS2[pr][0]], F.b[b][PAIRS2[pr][1]], acc[a][b], 0, 0, 0);
                }
    };

    // ---- prologue: NS - 1 stages in flight, stage 0 landed and visible ----
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue();
    wait_vm<(NS - 2) * PPW>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp<(EPI == EPI_QKV_ATTN ? 16 : 4)>(prm, wave, 1);

    if constexpr (PIPE != 0) {
        Frags F0, F1;
        read_frags(F0, 0);
        int slot = 1 % NS;                                              // ring slot of stage s + 1
        constexpr int NREAD = AR * (TM + TN), NMFMA = NPROD * TM * TN, MPR = NMFMA / NREAD;   // MFMAs pinned in front of each read
        static_assert(MPR >= 1, "at least one MFMA per fragment read");
#define AC_PIPE_STEP(FC, FN)                                                                                     \
        do {                                                                                                     \
            wait_vm<(NS - 3) * PPW>();                   /* this wave's pieces of stage s + 1 have landed */      \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            __builtin_amdgcn_s_barrier();                /* ... everyone's; the MFMAs of stage s - 1 are issued */ \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            issue();                                     /* stage s + NS - 1 -> the slot stage s - 1 used */      \
            read_frags(FN, slot);                        /* fragments of stage s + 1 */                           \
            slot = slot + 1 == NS ? 0 : slot + 1;                                                                \
            mfmas(FC);                                   /* stage s */                                            \
            if (PIPE == 2) {                             /* pin the interleave: fragment reads spread under the MFMAs */ \
                _Pragma("unroll") for (int g_ = 0; g_ < NREAD; ++g_) {                                           \
                    __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);                                         \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                           \
                }                                                                                                \
            }                                                                                                    \
        } while (0)
        for (int s = 0; s < nk; s += 2) {                               // (nk is even: K % 32 == 0)
            AC_PIPE_STEP(F0, F1);
            AC_PIPE_STEP(F1, F0);
        }
#undef AC_PIPE_STEP
    } else {
        int slot = 0;
        for (int s = 0; s < nk; ++s) {
            if (s > 0) {
                wait_vm<(NS - 2) * PPW>();                              // this wave's pieces of stage s have landed
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                           // ... everyone's; every read of stage s - 1 is done
                __builtin_amdgcn_sched_barrier(0);
            }
            issue();                                                    // stage s + NS - 1 -> the slot stage s - 1 used
            Frags F;
            read_frags(F, slot);
            slot = slot + 1 == NS ? 0 : slot + 1;
            mfmas(F);
        }
    }
    wait_vm<0>();                                                       // the over-issued tail stages: LDS is about to be reused / released
    stamp<(EPI == EPI_QKV_ATTN ? 16 : 4)>(prm, wave, 2);
    if constexpr (AR == 2) {                                            // operands were x 2^6 and w 2^10
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] *= ac::kF16OutScale;
    }
    if constexpr (EPI == EPI_BIAS_RES_LN) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                   // nobody's DMA may land in the epilogue's LDS scratch
        __builtin_amdgcn_sched_barrier(0);
        store_tile_ln<TM, TN, WMW, WNW, AR>(acc, lnres, prm, bm, bn, ntn, wm, wn, lane, tid, wave, reinterpret_cast<float*>(lds));
    } else if constexpr (EPI == EPI_QKV_ATTN) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                   // the ring becomes the attention's staging area
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (EPI == EPI_QKV_ATTN)
            qkv_attention_epilogue<TM, TN, WMW, WNW, AR>(acc, prm, bm, bn, wm, wn, lane, wave, reinterpret_cast<float*>(lds), at_cu_pref, at_bias);
    } else if (C_PLANES) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                   // nobody's DMA may land in the transpose scratch
        __builtin_amdgcn_sched_barrier(0);
        store_tile_planes<EPI, TM, TN, AR>(acc, reinterpret_cast<uint16_t*>(prm.C), prm.M, prm.N, m0, n0, wm, wn, lane, prm.epi,
                                       reinterpret_cast<float*>(lds) + wave * kTrFloats);
    } else {
        store_tile<EPI, TM, BM, TN>(acc, prm.C, prm.ldc, prm.M, prm.N, m0, n0, wm, wn, lane, prm.epi);
    }
    if (prm.stamps) { wait_vm<0>(); stamp<(EPI == EPI_QKV_ATTN ? 16 : 4)>(prm, wave, 3); }
}

std::atomic<int> g_krot{-1};                  // ac_gemm_set_krot (-1 = environment AC_GEMM_KROT, default off)
int krot_enabled() {
    int v = g_krot.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("AC_GEMM_KROT");
        v = (e && atoi(e) != 0) ? 1 : 0;
        g_krot.store(v, std::memory_order_relaxed);
    }
    return v;
}
unsigned long long* g_stamps = nullptr;      // ac_gemm_debug_stamps
int64_t g_stamp_cap = 0;

template <int EPI, int TM, int TN, int WMW, int WNW, int NS, bool CP, int PIPE, int AR = 3>
int launch_one(PipeParams p, hipStream_t stream) {
    using G = PipeGeom<TM, TN, WMW, WNW, NS, AR>;
    const int64_t tiles = (int64_t)((p.M + G::BM - 1) / G::BM) * ((p.N + G::BN - 1) / G::BN);
    constexpr int LN_BYTES = G::TR_BYTES + (WNW + 1) * G::BM * 8 + 16;                       // EPI_BIAS_RES_LN scratch
    const size_t lds = EPI == EPI_BIAS_RES_LN ? (size_t)(G::LDS_BYTES > LN_BYTES ? G::LDS_BYTES : LN_BYTES)
                     : EPI == EPI_QKV_ATTN    ? (size_t)(G::LDS_BYTES > kAtBytes ? G::LDS_BYTES : kAtBytes)
                                              : (size_t)(G::LDS_BYTES > G::TR_BYTES || !CP ? G::LDS_BYTES : G::TR_BYTES);
    static std::atomic<unsigned long long> attr_set{0};                // (per instantiation and, inside, per device)
    if (ac::first_call_on_device(attr_set))
        AC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_pipe_nt<EPI, TM, TN, WMW, WNW, NS, CP, PIPE, AR>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // AC_GEMM_STAMP_EPI / AC_GEMM_STAMP_K (read at every launch: a probe switches them between forwards): only launches of this
    // epilogue class / inner dimension stamp, so one GEMM of a whole forward can be looked at in place
    const char* se = g_stamps ? getenv("AC_GEMM_STAMP_EPI") : nullptr;
    const char* sk = g_stamps ? getenv("AC_GEMM_STAMP_K") : nullptr;
    p.stamps = (g_stamps && tiles <= g_stamp_cap && (!se || atoi(se) == EPI) && (!sk || atoi(sk) == p.K)) ? g_stamps : nullptr;
    p.krot = krot_enabled();
    hipLaunchKernelGGL((gemm_pipe_nt<EPI, TM, TN, WMW, WNW, NS, CP, PIPE, AR>), dim3((unsigned)tiles), dim3(64 * WMW * WNW), lds, stream, p);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// fp16x2 operands: what the BERT layer needs -- QKV (bias), FFN1 (bias + GELU -> planes), AO / FFN2 (bias + residual [+ LN])
template <int TM, int TN, int WMW, int WNW, int NS, int PIPE>
int launch_cfg_f16(int cls, bool cp, const PipeParams& p, hipStream_t stream) {
    if (cp && cls == EPI_BIAS_GELU) return launch_one<EPI_BIAS_GELU, TM, TN, WMW, WNW, NS, true, PIPE, 2>(p, stream);
    if (cp && cls == EPI_BIAS) return launch_one<EPI_BIAS, TM, TN, WMW, WNW, NS, true, PIPE, 2>(p, stream);
    if (!cp && cls == EPI_BIAS) return launch_one<EPI_BIAS, TM, TN, WMW, WNW, NS, false, PIPE, 2>(p, stream);
    if (!cp && cls == EPI_BIAS_RES) return launch_one<EPI_BIAS_RES, TM, TN, WMW, WNW, NS, false, PIPE, 2>(p, stream);
    if constexpr (TM == 1 && TN == 2 && WMW == 4 && WNW == 2 && NS == 6 && PIPE == 2)
        if (!cp && cls == EPI_BIAS_RES_LN) return launch_one<EPI_BIAS_RES_LN, TM, TN, WMW, WNW, NS, false, PIPE, 2>(p, stream);
    ac::set_error("gemm_pipe: epilogue class %d (planes out %d) not built for fp16x2 operands", cls, (int)cp);
    return AC_EUNSUPPORTED;
}

// the (EPI, C_PLANES) combinations the encoders use
template <int TM, int TN, int WMW, int WNW, int NS, int PIPE>
int launch_cfg(int cls, bool cp, const PipeParams& p, hipStream_t stream) {
    if (cp) {
        if (cls == EPI_BIAS_GELU) return launch_one<EPI_BIAS_GELU, TM, TN, WMW, WNW, NS, true, PIPE>(p, stream);
        if (cls == EPI_BIAS) return launch_one<EPI_BIAS, TM, TN, WMW, WNW, NS, true, PIPE>(p, stream);
        if constexpr (TN == 2) if (cls == EPI_GEGLU32) return launch_one<EPI_GEGLU32, TM, TN, WMW, WNW, NS, true, PIPE>(p, stream);
    } else {
        if (cls == EPI_BIAS) return launch_one<EPI_BIAS, TM, TN, WMW, WNW, NS, false, PIPE>(p, stream);
        if (cls == EPI_BIAS_RES) return launch_one<EPI_BIAS_RES, TM, TN, WMW, WNW, NS, false, PIPE>(p, stream);
        if constexpr (TM == 1 && TN == 2 && WMW == 4 && WNW == 2 && NS == 6 && PIPE == 2)      // (built for the one tile that uses it)
            if (cls == EPI_BIAS_RES_LN) return launch_one<EPI_BIAS_RES_LN, TM, TN, WMW, WNW, NS, false, PIPE>(p, stream);
    }
    ac::set_error("gemm_pipe: epilogue class %d (planes out %d) not built for this tile", cls, (int)cp);
    return AC_EUNSUPPORTED;
}

}  // namespace

/* diagnostic: the ring-staged GEMM kernels write 4 shader-clock stamps per workgroup (start, ring filled, loop done, stores
 * drained) into d_buf[4 * workgroup] while d_buf is set and holds the grid (tools/gemm_bench.hip); null switches it off. */
/* experiment switch: 1 = the XCDs start their k-loops at different stages; 0 (default) = all at stage 0 */
extern "C" int ac_gemm_set_krot(int on) {
    AC_TEST_HOOK_ONLY("ac_gemm_set_krot");
    g_krot.store(on ? 1 : 0, std::memory_order_relaxed);
    return AC_OK;
}
extern "C" int ac_gemm_debug_stamps(unsigned long long* d_buf, int64_t capacity_workgroups) {
    AC_TEST_HOOK_ONLY("ac_gemm_debug_stamps");
    g_stamps = d_buf;
    g_stamp_cap = d_buf ? capacity_workgroups : 0;
    return AC_OK;
}

namespace ac {

bool pipe_takes(int M, int N, int K, int cls, bool c_planes) {
    if (M < 192 || N < 1 || (K % 32) != 0 || K < 64) return false;
    if (c_planes) return (cls == EPI_BIAS || cls == EPI_BIAS_GELU || cls == EPI_GEGLU32) && (N % 8) == 0;
    return cls == EPI_BIAS || cls == EPI_BIAS_RES;
}

// cfg = tm tn wmw wnw ns pipe, one decimal digit each (e.g. 234232: wave tile 64 x 96, 4 x 2 waves = 256 x 192, ring of 3, pinned)
// The set that survived the sweeps of profiles/r03/gemm_sweep*.txt (dropped there: unpinned / non-pipelined forms of the
// same tiles, 64 x 128, 256 x 128 with two buffers, 256 x 256 as 8 waves of 128 x 64 or 16 waves of 64 x 64, 128 x 256).
#define AC_PIPE_CONFIGS(X)                                                                         \
    X(2, 2, 2, 2, 3, 2) /* 128 x 128, 4 waves of 64 x 64, ring of 3: two workgroups per CU */       \
    X(1, 2, 4, 2, 6, 1) X(1, 2, 4, 2, 6, 2) /* 128 x 128, 8 waves of 32 x 64, ring of 6: one workgroup per CU */ \
    X(2, 2, 4, 2, 4, 2) /* 256 x 128, 8 waves of 64 x 64, ring of 4 */                              \
    X(2, 3, 4, 2, 3, 2) /* 256 x 192, 8 waves of 64 x 96, ring of 3 */                              \
    X(3, 2, 2, 4, 3, 2) /* 192 x 256, 8 waves of 96 x 64, ring of 3 */                              \
    X(2, 4, 4, 2, 3, 2) /* 256 x 256, 8 waves of 64 x 128, ring of 3 */

// fp16x2 operands only: a stage is 2/3 of the bytes and half the matrix-pipe time, so the same tiles with deeper rings
#define AC_PIPE_CONFIGS_F16(X)                                                                     \
    X(2, 4, 4, 2, 4, 2) /* 256 x 256, ring of 4 (128 KB) */                                          \
    X(2, 3, 4, 2, 4, 2) /* 256 x 192, ring of 4 */                                                   \
    X(2, 2, 4, 2, 6, 2) /* 256 x 128, ring of 6 */                                                   \
    X(1, 2, 4, 2, 8, 2) /* 128 x 128, 8 waves, ring of 8 */

// per-shape configuration of the default dispatch (0 = the two-buffer tile kernels).  A runtime table (ac_gemm_set_pipe_table,
// tuning / A-B runs) takes precedence over the built-in choice.
struct PipeRule { int N, K, cfg; };
static PipeRule g_rules[16];
static int g_nrules = -1;          // -1: no runtime table

// Built-in choice.  Every CU works through ceil(tiles / CUs) tiles whatever the residency, so a configuration's time goes as
//     ceil(tiles / CUs) * BM * BN / s(cfg)
// with s = its relative per-CU throughput once the chip is full (8192^3 and 20564-row sweeps of profiles/r03/gemm_sweep3.txt;
// the chip is power-limited at ~0.5 - 0.6 of the bf16x3 ceiling there, and bigger tiles move fewer bytes per MFMA).  The rule
// reproduces the measured best (or a configuration within ~2 % of it) on every bert-base / bert-large shape at ~5 k and ~20 k
// packed token rows.  The two-buffer kernels of gemm.hip are not candidates: no measured shape has them ahead of the best
// ring configuration (they stay for A given as fp32, for epilogues outside pipe_takes, and behind ac_gemm_set_variant(1)).
static PipeRule g_rules_f16[16];
static int g_nrules_f16 = -1;
static int builtin_choose(int M, int N, int K, int cls, bool f16 = false);
int pipe_choose(int M, int N, int K, int cls, bool c_planes) {
    (void)c_planes;
    if (g_nrules >= 0) {
        for (int i = 0; i < g_nrules; ++i) if (g_rules[i].N == N && g_rules[i].K == K) return g_rules[i].cfg;
        return 0;
    }
    return builtin_choose(M, N, K, cls);
}
// fp16x2 operands: its own runtime table (ac_gemm_set_pipe_table_f16), else the tile the bf16x3 rule picks -- the round
// quantisation argument is the same, and the fused-LayerNorm launches need the same 128 x 128 tile in both arithmetics
int pipe_choose_f16(int M, int N, int K) {
    if (g_nrules_f16 >= 0)
        for (int i = 0; i < g_nrules_f16; ++i) if (g_rules_f16[i].N == N && g_rules_f16[i].K == K && g_rules_f16[i].cfg) return g_rules_f16[i].cfg;
    return builtin_choose(M, N, K, EPI_BIAS, true);
}
// f16: the fp16x2 kernels' relative throughputs differ in one place (profiles/r04/f16x2_probe_base.txt: QKV at 5141 rows 256 x 192
// ring of 4 60 us, 192 x 256 68 us) -- their loop is paced by the operand fetch, and 256 x 192 with a ring of 4 fetches best
static int builtin_choose(int M, int N, int K, int cls, bool f16) {
    (void)K;
    const int64_t cus = dev_info().cus;
    struct Cand { int cfg, bm, bn; double s; };
    static const Cand cands[] = {
        {244232, 256, 256, 1.05}, {234232, 256, 192, 0.90}, {322432, 192, 256, 0.92},
        {224242, 256, 128, 0.90}, {124262, 128, 128, 0.82}, {222232, 128, 128, 0.87},
    };
    int best = 0;
    double best_cost = 0;
    for (const Cand& c : cands) {
        if (cls == EPI_GEGLU32 && (c.cfg / 10000) % 10 != 2) continue;       // (fused GeGLU pairs the two column tiles of a 64-column wave tile)
        const int64_t tiles = (int64_t)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        double sp = c.s;
        if (f16 && c.cfg == 234232) sp = 0.95;
        if (c.cfg == 222232 && 2 * tiles < 3 * cus) sp = 0.78;               // two-per-CU kernel with mostly one workgroup per CU
        const double t = (double)((tiles + cus - 1) / cus) * c.bm * c.bn / sp;
        if (best == 0 || t < best_cost) { best_cost = t; best = c.cfg; }
    }
    if (f16 && best == 234232) best = 234242;
    return best;
}

int launch_gemm_pipe(int cfg, const uint16_t* Ap, int64_t a_rows, const uint16_t* Wp, int64_t w_rows, float* C, int64_t ldc,
                     uint16_t* Cp, int M, int N, int K, int cls, const acg::Epilogue& epi, hipStream_t stream, int f16) {
    PipeParams p;
    p.Ap = Ap; p.a_rows = a_rows; p.Wp = Wp; p.w_rows = w_rows;
    p.C = Cp ? reinterpret_cast<float*>(Cp) : C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = epi;
    p.ln = LnFuse{};
    p.stamps = nullptr;
    const bool cp = Cp != nullptr;
#define AC_CASE(TM, TN, WMW, WNW, NS, PIPE) \
    if (!f16 && cfg == (((((TM * 10 + TN) * 10 + WMW) * 10 + WNW) * 10 + NS) * 10 + PIPE)) return launch_cfg<TM, TN, WMW, WNW, NS, PIPE>(cls, cp, p, stream);
    AC_PIPE_CONFIGS(AC_CASE)
#undef AC_CASE
#define AC_CASE(TM, TN, WMW, WNW, NS, PIPE) \
    if (f16 && cfg == (((((TM * 10 + TN) * 10 + WMW) * 10 + WNW) * 10 + NS) * 10 + PIPE)) return launch_cfg_f16<TM, TN, WMW, WNW, NS, PIPE>(cls, cp, p, stream);
    AC_PIPE_CONFIGS(AC_CASE)
    AC_PIPE_CONFIGS_F16(AC_CASE)
#undef AC_CASE
    set_error("gemm_pipe: configuration %d not built", cfg);
    return AC_EUNSUPPORTED;
}


bool ln_fusion_enabled();
// A/B switch of the in-launch exchanges' memory ordering (LnFuse::fences): AC_EXCHANGE_FENCES=1 = release / acquire fences
static int exchange_fences() {
    static const int v = [] { const char* e = getenv("AC_EXCHANGE_FENCES"); return e && atoi(e) != 0 ? 1 : 0; }();
    return v;
}
// ---- the QKV projection with the self-attention of the packed sequences in its epilogue (EPI_QKV_ATTN) ----
static std::atomic<long long> g_qkv_attn_launches{0};
bool qkv_attn_applies(int M, int H, int heads, int smax) {
    if (const char* e = getenv("AC_QKV_ATTN_FUSION"); e && atoi(e) == 0) return false;      // (A/B runs and the two-launch route's tests)
    return arith_split() && gemm_variant() == 0 && M >= 192 && heads >= 1 && H == heads * 64 && (H % 32) == 0 && H >= 64 &&
           smax >= 1 && smax <= 64 && pipe_choose(M, 3 * H, H, EPI_BIAS, false) != 0;      // (a table that switches the ring kernels off)
}
// The per-forward sequence table of the attention epilogue: row t (kAtCu words) describes row tile t = rows [256 t, 256 t + 256):
// [0] = n = the sequences that START in it, [1 + i] = the first row of the i-th of them (i < n), [1 + n] = the first row of the next
// sequence (= T after the last one).  Built once per forward (cu is the same for every layer); a tile reads its row with one load
// per thread.
__global__ __launch_bounds__(kAtCu) void qkv_attn_tile_seq_kernel(const int32_t* __restrict__ cu, int b, int ntiles, int32_t* __restrict__ out) {
    const int t = blockIdx.x, i = threadIdx.x;
    auto first_at_or_after = [&](int row) { int lo = 0, hi = b; while (lo < hi) { const int mid = (lo + hi) >> 1; if (cu[mid] < row) lo = mid + 1; else hi = mid; } return lo; };
    const int sb = first_at_or_after(t * kQkvAttnRows), se = first_at_or_after((t + 1) * kQkvAttnRows), n = se - sb;
    int v = 0;
    if (i == 0) v = n;
    else if (i - 1 <= n) v = cu[sb + i - 1];                          // (sb + n <= b: cu has b + 1 entries)
    out[(size_t)t * kAtCu + i] = v;
}
size_t qkv_attn_tile_seq_bytes(int M) { return ((size_t)(M + kQkvAttnRows - 1) / kQkvAttnRows) * kAtCu * sizeof(int32_t); }
int qkv_attn_tile_seq(const int32_t* cu, int b, int M, int32_t* tile_seq, hipStream_t stream) {
    const int ntiles = (M + kQkvAttnRows - 1) / kQkvAttnRows;
    hipLaunchKernelGGL(qkv_attn_tile_seq_kernel, dim3(ntiles), dim3(kAtCu), 0, stream, cu, b, ntiles, tile_seq);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
// Residency proof for the in-launch exchange of straddling sequences (as ln_resident_capacity above): the whole grid resident at once
static int64_t qkv_attn_resident_capacity(int f16) {
    static std::atomic<int> occ_cache[2][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    int occ = occ_cache[f16 ? 1 : 0][dev].load(std::memory_order_relaxed);
    if (occ <= 0) {
        int per_cu = 0;
        hipError_t e;
        if (f16) {
            using G = PipeGeom<2, 3, 4, 2, 4, 2>;
            const int lds = G::LDS_BYTES > kAtBytes ? G::LDS_BYTES : kAtBytes;
            const void* fn = (const void*)gemm_pipe_nt<EPI_QKV_ATTN, 2, 3, 4, 2, 4, false, 2, 2>;
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * G::NW, lds);
        } else {
            using G = PipeGeom<2, 3, 4, 2, 3, 3>;
            const int lds = G::LDS_BYTES > kAtBytes ? G::LDS_BYTES : kAtBytes;
            const void* fn = (const void*)gemm_pipe_nt<EPI_QKV_ATTN, 2, 3, 4, 2, 3, false, 2, 3>;
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * G::NW, lds);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        occ = per_cu > 0 ? per_cu : -1;
        occ_cache[f16 ? 1 : 0][dev].store(occ, std::memory_order_relaxed);
    }
    return occ > 0 ? (int64_t)occ * dev_info().cus : 0;
}
// true: one residency round (proven) and the LayerNorm-exchange option is on -- the straddling sequences are finished inside the launch
bool qkv_attn_exchange_applies(int M, int heads, int f16) {
    if (!ln_fusion_enabled()) return false;              // (the encoder's "no in-launch exchanges" switch covers this one too)
    if (const char* e = getenv("AC_QKV_ATTN_EXCHANGE"); e && atoi(e) == 0) return false;       // (A/B runs, the boundary launch's tests)
    const int64_t tiles = (int64_t)((M + kQkvAttnRows - 1) / kQkvAttnRows) * heads;
    return tiles <= qkv_attn_resident_capacity(f16);
}
int launch_gemm_pipe_qkv_attn(const uint16_t* Ap, int64_t a_rows, const uint16_t* Wp, int64_t w_rows, const float* bias, int M, int H,
                              int heads, const int32_t* cu, const int32_t* tile_seq, int b, int smax, float scale, uint16_t* ctx_planes,
                              float* qkv, hipStream_t stream, int f16, unsigned* exchange, unsigned epoch, unsigned* abort_flag) {
    AC_REQUIRE(qkv_attn_applies(M, H, heads, smax), AC_EUNSUPPORTED, "gemm_pipe: fused attention epilogue not applicable (M %d H %d heads %d longest %d)",
               M, H, heads, smax);
    AC_REQUIRE(Ap && Wp && bias && cu && tile_seq && ctx_planes && qkv && b >= 1, AC_EINVAL, "gemm_pipe_qkv_attn: null pointer");
    PipeParams p;
    p.Ap = Ap; p.a_rows = a_rows; p.Wp = Wp; p.w_rows = w_rows; p.C = qkv; p.ldc = 3 * (int64_t)H; p.M = M; p.N = 3 * H; p.K = H;
    Epilogue e;
    e.bias = bias; e.residual = nullptr; e.ldr = 0; e.act = ACT_NONE; e.alpha = 1.f; e.beta = 0.f; e.mask = nullptr;
    e.mask_scale = 1.f; e.gate = nullptr; e.ldg = 0; e.gate_scale = 1.f; e.drop_p = 0.f; e.drop_seed = 0;
    p.epi = e;
    p.ln = LnFuse{};
    p.at.cu = cu; p.at.tile_seq = tile_seq; p.at.b = b; p.at.H = H; p.at.smax = smax; p.at.scale = scale; p.at.ctx_planes = ctx_planes;
    p.at.qkv = qkv;
    AC_REQUIRE(!exchange || (abort_flag && qkv_attn_exchange_applies(M, heads, f16)), AC_EINVAL,
               "gemm_pipe_qkv_attn: the in-launch exchange needs every tile resident (%d rows x %d heads) and an abort word", M, heads);
    p.at.exchange = exchange; p.at.epoch = epoch; p.at.abort_ = abort_flag; p.at.fences = exchange_fences();
    p.stamps = nullptr;
    g_qkv_attn_launches.fetch_add(1, std::memory_order_relaxed);
    return f16 ? launch_one<EPI_QKV_ATTN, 2, 3, 4, 2, 4, false, 2, 2>(p, stream)
               : launch_one<EPI_QKV_ATTN, 2, 3, 4, 2, 3, false, 2, 3>(p, stream);
}

// ---- bias + residual + LayerNorm fused into the N-wide GEMMs of an encoder layer (EPI_BIAS_RES_LN) ----
constexpr int kLnCfg = 124262, kLnBM = 128, kLnBN = 128;     // the one tile the fused epilogue is built for
static std::atomic<int> g_ln_fusion{-1};
static std::atomic<long long> g_ln_launches{0};
bool ln_fusion_enabled() {
    if (const int o = call_opts().ln_fusion; o >= 0) return o != 0;  // the running call's own option (ac_bert_config.ln_fusion_opt)
    int v = g_ln_fusion.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("AC_LN_FUSION");
        v = (e && atoi(e) == 0) ? 0 : 1;
        g_ln_fusion.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
// The launch must be ONE round of one workgroup per CU (all tiles of a row panel co-resident) and the default dispatch must
// pick the 128 x 128 eight-wave tile for the shape anyway.
// Residency proof of the exchange, made at launch: workgroups of the fused kernel that fit one CU (occupancy query, once per
// device) x the CUs this process's workgroups can land on (dev_info().cus: MEASURED, so a CU mask counts) must hold the whole
// grid -- then every tile of every row panel is resident together and the panel counters fill.  When it does not hold the fused
// form is simply not chosen.  (What no launch-time check can see is another process's kernels on the same CUs: the bounded wait
// in store_tile_ln stays for that and for bugs, as an assertion that turns into NaN rows instead of a hung GPU.)
static int64_t ln_resident_capacity() {
    static std::atomic<int> occ_cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    int occ = occ_cache[dev].load(std::memory_order_relaxed);
    if (occ <= 0) {
        using G = PipeGeom<1, 2, 4, 2, 6, 3>;
        constexpr int LN_BYTES = G::TR_BYTES + (2 + 1) * G::BM * 8 + 16;
        const int lds = G::LDS_BYTES > LN_BYTES ? G::LDS_BYTES : LN_BYTES;
        const void* fn = (const void*)gemm_pipe_nt<EPI_BIAS_RES_LN, 1, 2, 4, 2, 6, false, 2, 3>;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * G::NW, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        occ = per_cu > 0 ? per_cu : -1;
        occ_cache[dev].store(occ, std::memory_order_relaxed);
    }
    return occ > 0 ? (int64_t)occ * dev_info().cus : 0;
}
bool pipe_ln_applies(int M, int N, int K) {
    if (!ln_fusion_enabled() || !arith_split() || gemm_variant() != 0) return false;
    if (M < 192 || (N % kLnBN) != 0 || N / kLnBN > 8 || (K % 32) != 0 || K < 64) return false;
    const int64_t tiles = (int64_t)((M + kLnBM - 1) / kLnBM) * (N / kLnBN);
    // one tile per CU is what the kernel is tuned for (tiles <= CUs); the proof is tiles <= resident capacity
    return tiles <= dev_info().cus && tiles <= ln_resident_capacity() && pipe_choose(M, N, K, EPI_BIAS_RES, false) == kLnCfg;
}
size_t pipe_ln_part_bytes(int M, int N) { return (size_t)((M + kLnBM - 1) / kLnBM) * (size_t)(N / kLnBN) * kLnBM * sizeof(float2); }
int pipe_ln_panels(int M) { return (M + kLnBM - 1) / kLnBM; }

int launch_gemm_pipe_ln(const uint16_t* Ap, int64_t a_rows, const uint16_t* Wp, int64_t w_rows, const float* bias,
                        const float* residual, int64_t ldr, float* C, int64_t ldc, int M, int N, int K, const float* gamma,
                        const float* beta, float eps, void* part, unsigned* count, unsigned* abort_flag, uint16_t* planes,
                        hipStream_t stream, int f16) {
    AC_REQUIRE(pipe_ln_applies(M, N, K), AC_EUNSUPPORTED, "gemm_pipe: fused LayerNorm epilogue not applicable to %d x %d x %d", M, N, K);
    AC_REQUIRE(Ap && Wp && bias && residual && C && gamma && beta && part && count && abort_flag, AC_EINVAL, "gemm_pipe_ln: null pointer");
    PipeParams p;
    p.Ap = Ap; p.a_rows = a_rows; p.Wp = Wp; p.w_rows = w_rows; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    Epilogue e;
    e.bias = bias; e.residual = residual; e.ldr = ldr; e.act = ACT_NONE; e.alpha = 1.f; e.beta = 0.f; e.mask = nullptr;
    e.mask_scale = 1.f; e.gate = nullptr; e.ldg = 0; e.gate_scale = 1.f; e.drop_p = 0.f; e.drop_seed = 0;
    p.epi = e;
    p.ln.gamma = gamma; p.ln.beta = beta; p.ln.eps = eps; p.ln.part = (float2*)part; p.ln.count = count; p.ln.abort_ = abort_flag;
    p.ln.fences = exchange_fences();
    p.ln.planes = planes; p.ln.starve = (call_opts().ln_fusion >= 0 ? call_opts().ln_fusion : g_ln_fusion.load(std::memory_order_relaxed)) == 2 ? 1 : 0;
    p.stamps = nullptr;
    g_ln_launches.fetch_add(1, std::memory_order_relaxed);
    return f16 ? launch_cfg_f16<1, 2, 4, 2, 6, 2>(EPI_BIAS_RES_LN, false, p, stream)
               : launch_cfg<1, 2, 4, 2, 6, 2>(EPI_BIAS_RES_LN, false, p, stream);
}
}  // namespace ac

/* tuning / A-B: "NxK=cfg;NxK=cfg;..." overrides the built-in per-shape choice of the ring-staged kernels for GEMMs with N output
 * columns and inner dimension K (cfg 0 = two-buffer tile kernels); an empty string = no ring kernel anywhere; NULL restores
 * the built-in table. */
/* process-wide switch (A/B runs, tests): 0 = the encoder keeps its LayerNorms as separate launches; default 1, or AC_LN_FUSION=0 */
extern "C" int ac_gemm_set_ln_fusion(int on) {
    AC_TEST_HOOK_ONLY("ac_gemm_set_ln_fusion");
    ac::g_ln_fusion.store(on == 2 ? 2 : (on ? 1 : 0), std::memory_order_relaxed);
    return AC_OK;
}

extern "C" int64_t ac_gemm_ln_fusion_launches(void) { return (int64_t)ac::g_ln_launches.load(std::memory_order_relaxed); }
/* diagnostic (tests assert that the fused route really ran): launches of the QKV GEMM with the attention epilogue so far */
extern "C" int64_t ac_gemm_qkv_attn_launches(void) { return (int64_t)ac::g_qkv_attn_launches.load(std::memory_order_relaxed); }

static int parse_pipe_table(const char* spec, ac::PipeRule* rules, int* nrules) {
    if (!spec) { *nrules = -1; return AC_OK; }
    int n = 0;
    const char* p = spec;
    while (*p && n < 16) {
        int N = 0, K = 0, cfg = 0, used = 0;
        if (sscanf(p, "%dx%d=%d%n", &N, &K, &cfg, &used) != 3) { ac::set_error("gemm pipe table: cannot parse '%s'", p); return AC_EINVAL; }
        rules[n++] = {N, K, cfg};
        p += used;
        if (*p == ';') ++p;
    }
    *nrules = n;
    return AC_OK;
}
extern "C" int ac_gemm_set_pipe_table(const char* spec) {
    AC_TEST_HOOK_ONLY("ac_gemm_set_pipe_table");
    return parse_pipe_table(spec, ac::g_rules, &ac::g_nrules);
}
/* the same for the fp16x2 kernels (AC_GEMM_F16X2); shapes the table does not name keep the built-in choice */
extern "C" int ac_gemm_set_pipe_table_f16(const char* spec) {
    AC_TEST_HOOK_ONLY("ac_gemm_set_pipe_table_f16");
    return parse_pipe_table(spec, ac::g_rules_f16, &ac::g_nrules_f16);
}
