// gemm_pipe_nt: the planes GEMM of the encoder (transformers BertModel projections, called at
// /root/reference/src/adaptive_classifier/classifier.py:1271) with the operand stages held in an LDS RING instead of two
// buffers: the global_load_lds DMA of stage s + NS - 1 is issued while stage s is multiplied, the ring is drained with
// COUNTED s_waitcnt vmcnt(N) + a raw s_barrier (no vmcnt(0) in the loop), and (PIPE) every wave reads the fragments
// of stage s + 1 into a second register set under the MFMAs of stage s.
//
// Why (profiles/r02/gemm_planes_pmc.json, gemm_shapes_isolated.txt): with two buffers and a draining barrier per 16-k
// stage the tile kernel of gemm.hip keeps one stage in flight per workgroup and leans on 3 resident workgroups per CU to
// hide the L2 -> LDS latency.  The encoder's N = 768 GEMMs (attention output, FFN2) at ~5000 packed token rows have
// only 1 - 2 workgroups per CU, so there the matrix pipe waits on every stage (0.33 - 0.35 of the bf16x3 ceiling).  A
// ring keeps NS - 1 stages in flight from ONE workgroup.
//
// Same tile family, arithmetic and epilogues as gemm_planes_nt: (32 TM WMW) x 128 tile, 2 WMW waves of (32 TM) x 64,
// 16-k stages of three bf16 planes per operand, six products smallest first, fp32 accumulate.
#include "common.h"
#include "gemm_common.h"

#include <stdio.h>
#include <stdlib.h>

namespace {
using namespace acg;

constexpr int PBN = 128;                 // tile columns
constexpr int PSBK = 16;                 // k per stage

struct PipeParams {
    const uint16_t* Ap; int64_t a_rows;
    const uint16_t* Wp; int64_t w_rows;
    float* C; int64_t ldc;              // fp32 result, or (C_PLANES) the planes of the next GEMM's operand
    int M, N, K;
    Epilogue epi;
    unsigned long long* stamps;         // diagnostic (ac_gemm_debug_stamps): 4 shader-clock stamps per workgroup, or null
};

// shader-clock stamp `i` of this workgroup (wave 0 only; a wave-uniform branch on a kernel argument)
__device__ __forceinline__ void stamp(const PipeParams& prm, int wave, int i) {
    if (prm.stamps && wave == 0) {
        const unsigned long long t = clock64();
        if ((threadIdx.x & 63) == 0) prm.stamps[(size_t)blockIdx.x * 4 + i] = t;
    }
}

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int TM, int WMW, int NS> struct PipeGeom {
    static constexpr int BM = 32 * TM * WMW;
    static constexpr int RA = BM / 32, RW = PBN / 32, RG = RA + RW;      // 32-row groups per stage
    static constexpr int NP = 3 * RG;                                    // 1 KB DMA pieces per stage
    static constexpr int NW = 2 * WMW;                                   // waves
    static constexpr int PPW = (NP + NW - 1) / NW;                       // pieces per wave and stage (excess = duplicates)
    static constexpr int SLOT = 3 * RG * 64;                             // uint4 per ring slot
    static constexpr int LDS_BYTES = NS * SLOT * 16;
    static constexpr int TR_BYTES = NW * kTrFloats * 4;                  // transpose scratch of the planes epilogue
    static constexpr int BPC_LDS = (160 * 1024) / (LDS_BYTES > TR_BYTES ? LDS_BYTES : TR_BYTES);
    static constexpr int BPC = BPC_LDS < 1 ? 1 : (BPC_LDS * NW > 16 ? 16 / NW : BPC_LDS);   // <= 4 waves per SIMD wanted
    static constexpr int WAVES_PER_SIMD = (BPC * NW + 3) / 4;
};

template <int EPI, int TM, int WMW, int NS, bool C_PLANES, int PIPE>
__global__ __launch_bounds__(128 * WMW, (PipeGeom<TM, WMW, NS>::WAVES_PER_SIMD)) void gemm_pipe_nt(PipeParams prm) {
    using G = PipeGeom<TM, WMW, NS>;
    constexpr int BM = G::BM, RA = G::RA, RG = G::RG, NP = G::NP, NW = G::NW, PPW = G::PPW, SLOT = G::SLOT;
    static_assert(!PIPE || NS >= 3, "the software-pipelined loop needs a ring of >= 3 stages");
    static_assert(NS >= 2, "ring depth");
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];       // [NS][3 planes][RG groups][64 lanes]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (prm.N + PBN - 1) / PBN;
    const int tile = xcd_tile_id(blockIdx.x, gridDim.x);
    const int bn = tile % ntn, bm = tile / ntn;
    const int m0 = bm * BM, n0 = bn * PBN;
    const int nk = prm.K / PSBK;
    stamp(prm, wave, 0);

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
                int row = n0 + 32 * (g - RA) + i32; if (row > prm.N - 1) row = prm.N - 1;
                pp[t] = prm.Wp + p * w_plane + ((int64_t)kg * prm.w_rows + row) * 8;
            }
        }
    }
    int iss = 0;                                                        // next stage to issue
    auto issue = [&]() {                                                // always PPW DMA instructions (exact vmcnt accounting)
        const int slot = iss % NS;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const int j = (wave + NW * t) % NP, p = j / RG, g = j % RG;
            __builtin_amdgcn_global_load_lds((glb_void_t*)pp[t], (lds_void_t*)&lds[slot * SLOT + (p * RG + g) * 64], 16, 0, 0);
        }
        if (iss + 1 < nk) {                                             // past the end: the last stage again (harmless duplicates)
#pragma unroll
            for (int t = 0; t < PPW; ++t) pp[t] += ((wave + NW * t) % NP) % RG < RA ? a_step : w_step;
        }
        ++iss;
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    struct Frags { bf16x8_t a[TM][3], b[2][3]; };
    auto read_frags = [&](Frags& F, int slot) {
        const uint4* base = lds + slot * SLOT + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < TM; ++a) F.a[a][p] = __builtin_bit_cast(bf16x8_t, base[(p * RG + TM * wm + a) * 64]);
#pragma unroll
            for (int b = 0; b < 2; ++b) F.b[b][p] = __builtin_bit_cast(bf16x8_t, base[(p * RG + RA + 2 * wn + b) * 64]);
        }
    };
    auto mfmas = [&](const Frags& F) {
        constexpr int PAIRS[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};   // smallest products first
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[a][PAIRS[pr][0]], F.b[b][PAIRS[pr][1]], acc[a][b], 0, 0, 0);
    };

    // ---- prologue: NS - 1 stages in flight, stage 0 landed and visible ----
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue();
    wait_vm<(NS - 2) * PPW>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp(prm, wave, 1);

    if constexpr (PIPE != 0) {
        Frags F0, F1;
        read_frags(F0, 0);
        int slot = 1 % NS;                                              // ring slot of stage s + 1
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
                _Pragma("unroll") for (int g_ = 0; g_ < 3 * (TM + 2); ++g_) {                                    \
                    __builtin_amdgcn_sched_group_barrier(0x008, TM == 2 ? 2 : 1, 0);                             \
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
    stamp(prm, wave, 2);
    if (C_PLANES) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                   // nobody's DMA may land in the transpose scratch
        __builtin_amdgcn_sched_barrier(0);
        store_tile_planes<EPI, TM>(acc, reinterpret_cast<uint16_t*>(prm.C), prm.M, prm.N, m0, n0, wm, wn, lane, prm.epi,
                                   reinterpret_cast<float*>(lds) + wave * kTrFloats);
    } else {
        store_tile<EPI, TM, BM>(acc, prm.C, prm.ldc, prm.M, prm.N, m0, n0, wm, wn, lane, prm.epi);
    }
    if (prm.stamps) { wait_vm<0>(); stamp(prm, wave, 3); }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised form: the DMA of a stage is issued by NLW LOADER waves, the 2 WMW CONSUMER waves only read fragments and
// feed the matrix pipe.  An LDS-DMA instruction costs its issuing wave 60 - 180 cycles (MI355X_MICROARCH.md, cycle
// constants): six of them per 24 MFMAs in the same in-order wave leave the matrix pipe idle for about as long as it
// runs; from a neighbouring wave of the same SIMD they issue beside the MFMAs.  Same ring, same counted waits: a loader
// waits for ITS pieces of stage s + 1, everyone meets at the one barrier of the step, the loader refills the slot of
// stage s - 1 while the consumers read stage s + 1 and multiply stage s.
// ---------------------------------------------------------------------------------------------------------------------
template <int TM, int WMW, int NS, int NLW> struct WsGeom {
    using G = PipeGeom<TM, WMW, NS>;
    static constexpr int NCW = 2 * WMW;
    static constexpr int PPL = (G::NP + NLW - 1) / NLW;                 // pieces per loader wave and stage
    static constexpr int WAVES = NCW + NLW;
    static constexpr int BPC = G::BPC_LDS < 1 ? 1 : (G::BPC_LDS * WAVES > 16 ? (16 / WAVES < 1 ? 1 : 16 / WAVES) : G::BPC_LDS);
    static constexpr int WAVES_PER_SIMD = (BPC * WAVES + 3) / 4;
};

template <int EPI, int TM, int WMW, int NS, bool C_PLANES, int PIPE, int NLW>
__global__ __launch_bounds__(64 * (2 * WMW + NLW), (WsGeom<TM, WMW, NS, NLW>::WAVES_PER_SIMD)) void gemm_ws_nt(PipeParams prm) {
    using G = PipeGeom<TM, WMW, NS>;
    using WG = WsGeom<TM, WMW, NS, NLW>;
    constexpr int BM = G::BM, RA = G::RA, RG = G::RG, NP = G::NP, SLOT = G::SLOT, NCW = WG::NCW, PPL = WG::PPL;
    static_assert(PIPE != 0 && NS >= 3, "the specialised kernel is software-pipelined");
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = (prm.N + PBN - 1) / PBN;
    const int tile = xcd_tile_id(blockIdx.x, gridDim.x);
    const int bn = tile % ntn, bm = tile / ntn;
    const int m0 = bm * BM, n0 = bn * PBN;
    const int nk = prm.K / PSBK;
    stamp(prm, wave, 0);

    if (wave >= NCW) {
        // ------------------------------------------------ loader ------------------------------------------------
        const int lw = wave - NCW;
        const uint16_t* pp[PPL];
        const int64_t a_step = 2 * prm.a_rows * 8, w_step = 2 * prm.w_rows * 8;
        {
            const int i32 = lane & 31, kg = lane >> 5;
            const int64_t a_plane = prm.a_rows * (int64_t)prm.K, w_plane = prm.w_rows * (int64_t)prm.K;
#pragma unroll
            for (int t = 0; t < PPL; ++t) {
                const int j = (lw + NLW * t) % NP, p = j / RG, g = j % RG;
                if (g < RA) {
                    int row = m0 + 32 * g + i32; if (row > prm.M - 1) row = prm.M - 1;
                    pp[t] = prm.Ap + p * a_plane + ((int64_t)kg * prm.a_rows + row) * 8;
                } else {
                    int row = n0 + 32 * (g - RA) + i32; if (row > prm.N - 1) row = prm.N - 1;
                    pp[t] = prm.Wp + p * w_plane + ((int64_t)kg * prm.w_rows + row) * 8;
                }
            }
        }
        int iss = 0;
        auto issue = [&]() {
            const int slot = iss % NS;
#pragma unroll
            for (int t = 0; t < PPL; ++t) {
                const int j = (lw + NLW * t) % NP, p = j / RG, g = j % RG;
                __builtin_amdgcn_global_load_lds((glb_void_t*)pp[t], (lds_void_t*)&lds[slot * SLOT + (p * RG + g) * 64], 16, 0, 0);
            }
            if (iss + 1 < nk) {
#pragma unroll
                for (int t = 0; t < PPL; ++t) pp[t] += ((lw + NLW * t) % NP) % RG < RA ? a_step : w_step;
            }
            ++iss;
        };
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue();
        wait_vm<(NS - 2) * PPL>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        for (int s = 0; s < nk; ++s) {
            wait_vm<(NS - 3) * PPL>();                                  // this wave's pieces of stage s + 1 have landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue();                                                    // stage s + NS - 1 -> the slot stage s - 1 used
        }
        wait_vm<0>();
        if (C_PLANES) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                               // (the consumers reuse the ring as transpose scratch)
        }
        return;
    }

    // -------------------------------------------------- consumer --------------------------------------------------
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    struct Frags { bf16x8_t a[TM][3], b[2][3]; };
    auto read_frags = [&](Frags& F, int slot) {
        const uint4* base = lds + slot * SLOT + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < TM; ++a) F.a[a][p] = __builtin_bit_cast(bf16x8_t, base[(p * RG + TM * wm + a) * 64]);
#pragma unroll
            for (int b = 0; b < 2; ++b) F.b[b][p] = __builtin_bit_cast(bf16x8_t, base[(p * RG + RA + 2 * wn + b) * 64]);
        }
    };
    auto mfmas = [&](const Frags& F) {
        constexpr int PAIRS[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[a][PAIRS[pr][0]], F.b[b][PAIRS[pr][1]], acc[a][b], 0, 0, 0);
    };
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                       // stage 0 has landed
    __builtin_amdgcn_sched_barrier(0);
    stamp(prm, wave, 1);
    Frags F0, F1;
    read_frags(F0, 0);
    int slot = 1 % NS;
#define AC_WS_STEP(FC, FN)                                                                                       \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_barrier();                    /* stage s + 1 has landed (the loaders waited for it) */ \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        read_frags(FN, slot);                                                                                    \
        slot = slot + 1 == NS ? 0 : slot + 1;                                                                    \
        mfmas(FC);                                                                                               \
        if (PIPE == 2) {                                                                                         \
            _Pragma("unroll") for (int g_ = 0; g_ < 3 * (TM + 2); ++g_) {                                        \
                __builtin_amdgcn_sched_group_barrier(0x008, TM == 2 ? 2 : 1, 0);                                 \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
            }                                                                                                    \
        }                                                                                                        \
    } while (0)
    for (int s = 0; s < nk; s += 2) {
        AC_WS_STEP(F0, F1);
        AC_WS_STEP(F1, F0);
    }
#undef AC_WS_STEP
    stamp(prm, wave, 2);
    if (C_PLANES) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                   // every loader has drained its DMA queue
        __builtin_amdgcn_sched_barrier(0);
        store_tile_planes<EPI, TM>(acc, reinterpret_cast<uint16_t*>(prm.C), prm.M, prm.N, m0, n0, wm, wn, lane, prm.epi,
                                   reinterpret_cast<float*>(lds) + wave * kTrFloats);
    } else {
        store_tile<EPI, TM, BM>(acc, prm.C, prm.ldc, prm.M, prm.N, m0, n0, wm, wn, lane, prm.epi);
    }
    if (prm.stamps) { wait_vm<0>(); stamp(prm, wave, 3); }
}

// one launchable configuration
unsigned long long* g_stamps = nullptr;      // ac_gemm_debug_stamps
int64_t g_stamp_cap = 0;

template <int EPI, int TM, int WMW, int NS, bool CP, int PIPE>
int launch_one(const PipeParams& p, hipStream_t stream) {
    using G = PipeGeom<TM, WMW, NS>;
    const int64_t tiles = (int64_t)((p.M + G::BM - 1) / G::BM) * ((p.N + PBN - 1) / PBN);
    const size_t lds = (size_t)(G::LDS_BYTES > G::TR_BYTES || !CP ? G::LDS_BYTES : G::TR_BYTES);
    static bool attr_set = false;                                      // (per instantiation)
    if (!attr_set) {
        AC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_pipe_nt<EPI, TM, WMW, NS, CP, PIPE>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_pipe_nt<EPI, TM, WMW, NS, CP, PIPE>), dim3((unsigned)tiles), dim3(128 * WMW), lds, stream, p);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

template <int EPI, int TM, int WMW, int NS, bool CP, int PIPE, int NLW>
int launch_one_ws(const PipeParams& p, hipStream_t stream) {
    using G = PipeGeom<TM, WMW, NS>;
    const int64_t tiles = (int64_t)((p.M + G::BM - 1) / G::BM) * ((p.N + PBN - 1) / PBN);
    const size_t lds = (size_t)(G::LDS_BYTES > G::TR_BYTES || !CP ? G::LDS_BYTES : G::TR_BYTES);
    static bool attr_set = false;
    if (!attr_set) {
        AC_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_ws_nt<EPI, TM, WMW, NS, CP, PIPE, NLW>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_ws_nt<EPI, TM, WMW, NS, CP, PIPE, NLW>), dim3((unsigned)tiles), dim3(64 * (2 * WMW + NLW)), lds, stream, p);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

template <int TM, int WMW, int NS, int PIPE, int NLW>
int launch_cfg_ws(int cls, bool cp, const PipeParams& p, hipStream_t stream) {
    if (cp) {
        if (cls == EPI_BIAS_GELU) return launch_one_ws<EPI_BIAS_GELU, TM, WMW, NS, true, PIPE, NLW>(p, stream);
        if (cls == EPI_BIAS) return launch_one_ws<EPI_BIAS, TM, WMW, NS, true, PIPE, NLW>(p, stream);
        if (cls == EPI_GEGLU32) return launch_one_ws<EPI_GEGLU32, TM, WMW, NS, true, PIPE, NLW>(p, stream);
    } else {
        if (cls == EPI_BIAS) return launch_one_ws<EPI_BIAS, TM, WMW, NS, false, PIPE, NLW>(p, stream);
        if (cls == EPI_BIAS_RES) return launch_one_ws<EPI_BIAS_RES, TM, WMW, NS, false, PIPE, NLW>(p, stream);
    }
    ac::set_error("gemm_ws: epilogue class %d (planes out %d) not built", cls, (int)cp);
    return AC_EUNSUPPORTED;
}

// the (EPI, C_PLANES) combinations the encoder uses
template <int TM, int WMW, int NS, int PIPE>
int launch_cfg(int cls, bool cp, const PipeParams& p, hipStream_t stream) {
    if (cp) {
        if (cls == EPI_BIAS_GELU) return launch_one<EPI_BIAS_GELU, TM, WMW, NS, true, PIPE>(p, stream);
        if (cls == EPI_BIAS) return launch_one<EPI_BIAS, TM, WMW, NS, true, PIPE>(p, stream);
        if (cls == EPI_GEGLU32) return launch_one<EPI_GEGLU32, TM, WMW, NS, true, PIPE>(p, stream);
    } else {
        if (cls == EPI_BIAS) return launch_one<EPI_BIAS, TM, WMW, NS, false, PIPE>(p, stream);
        if (cls == EPI_BIAS_RES) return launch_one<EPI_BIAS_RES, TM, WMW, NS, false, PIPE>(p, stream);
    }
    ac::set_error("gemm_pipe: epilogue class %d (planes out %d) not built", cls, (int)cp);
    return AC_EUNSUPPORTED;
}

}  // namespace

namespace ac {

bool pipe_takes(int M, int N, int K, int cls, bool c_planes) {
    if (M < 192 || N < 1 || (K % 32) != 0 || K < 64) return false;
    if (c_planes) return (cls == EPI_BIAS || cls == EPI_BIAS_GELU || cls == EPI_GEGLU32) && (N % 8) == 0;
    return cls == EPI_BIAS || cls == EPI_BIAS_RES;
}

}  // namespace ac

/* diagnostic: the ring-staged GEMM kernels write 4 shader-clock stamps per workgroup (start, ring filled, loop done, stores
 * drained) into d_buf[4 * workgroup] while d_buf is set and holds the grid (tools/gemm_bench.hip); null switches it off. */
extern "C" int ac_gemm_debug_stamps(unsigned long long* d_buf, int64_t capacity_workgroups) {
    g_stamps = d_buf;
    g_stamp_cap = d_buf ? capacity_workgroups : 0;
    return AC_OK;
}

namespace ac {

// per-shape configuration of the default dispatch (0 = the two-buffer tile kernels).  A runtime table (ac_gemm_set_pipe_table,
// tuning / A-B runs) takes precedence over the built-in choices, which come from tools/gemm_bench sweeps on MI355X.
struct PipeRule { int N, K, cfg; };
static PipeRule g_rules[16];
static int g_nrules = -1;          // -1: no runtime table

int pipe_choose(int M, int N, int K, int cls, bool c_planes) {
    (void)cls; (void)c_planes;
    if (g_nrules >= 0) {
        for (int i = 0; i < g_nrules; ++i) if (g_rules[i].N == N && g_rules[i].K == K) return g_rules[i].cfg;
        return 0;
    }
    (void)M;
    return 0;
}

}  // namespace ac

/* tuning / A-B: "NxK=cfg;NxK=cfg;..." overrides the built-in per-shape choice of the ring-staged kernels for GEMMs with N output
 * columns and inner dimension K (cfg 0 = two-buffer tile kernels); an empty string = no ring kernel anywhere; NULL restores
 * the built-in table. */
extern "C" int ac_gemm_set_pipe_table(const char* spec) {
    if (!spec) { ac::g_nrules = -1; return AC_OK; }
    int n = 0;
    const char* p = spec;
    while (*p && n < 16) {
        int N = 0, K = 0, cfg = 0, used = 0;
        if (sscanf(p, "%dx%d=%d%n", &N, &K, &cfg, &used) != 3) { ac::set_error("gemm pipe table: cannot parse '%s'", p); return AC_EINVAL; }
        ac::g_rules[n++] = {N, K, cfg};
        p += used;
        if (*p == ';') ++p;
    }
    ac::g_nrules = n;
    return AC_OK;
}

namespace ac {

// cfg = tm * 1000 + wmw * 100 + ns * 10 + pipe
int launch_gemm_pipe(int cfg, const uint16_t* Ap, int64_t a_rows, const uint16_t* Wp, int64_t w_rows, float* C, int64_t ldc,
                     uint16_t* Cp, int M, int N, int K, int cls, const acg::Epilogue& epi, hipStream_t stream) {
    PipeParams p;
    p.Ap = Ap; p.a_rows = a_rows; p.Wp = Wp; p.w_rows = w_rows;
    p.C = Cp ? reinterpret_cast<float*>(Cp) : C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.epi = epi;
    p.stamps = g_stamps;
    if (g_stamps) {
        const int bm = 32 * ((cfg / 1000) % 10) * ((cfg / 100) % 10);
        if ((int64_t)((M + bm - 1) / bm) * ((N + PBN - 1) / PBN) > g_stamp_cap) p.stamps = nullptr;
    }
    const bool cp = Cp != nullptr;
    switch (cfg) {
#define AC_WS(NLW, TM, WMW, NS, PIPE) case (NLW * 10000 + TM * 1000 + WMW * 100 + NS * 10 + PIPE): return launch_cfg_ws<TM, WMW, NS, PIPE, NLW>(cls, cp, p, stream);
        AC_WS(4, 2, 2, 6, 1)    // 128 x 128: 4 consumers of 64 x 64 + 4 loaders, one workgroup per CU
        AC_WS(4, 2, 2, 6, 2)
        AC_WS(2, 2, 2, 6, 2)    //            ... + 2 loaders
        AC_WS(2, 2, 2, 3, 2)    //            two workgroups per CU (6 waves each)
        AC_WS(4, 2, 4, 4, 1)    // 256 x 128: 8 consumers + 4 loaders
        AC_WS(4, 2, 4, 4, 2)
        AC_WS(4, 1, 4, 6, 2)    // 128 x 128: 8 consumers of 32 x 64 + 4 loaders
        AC_WS(4, 1, 4, 3, 2)
#undef AC_WS
#define AC_CFG(TM, WMW, NS, PIPE) case (TM * 1000 + WMW * 100 + NS * 10 + PIPE): return launch_cfg<TM, WMW, NS, PIPE>(cls, cp, p, stream);
        AC_CFG(2, 2, 2, 0)      // 128 x 128, 4 waves, two buffers (the gemm_planes_nt schedule with a raw barrier)
        AC_CFG(2, 2, 3, 0)
        AC_CFG(2, 2, 3, 1)
        AC_CFG(2, 2, 4, 1)
        AC_CFG(2, 2, 6, 1)
        AC_CFG(2, 2, 6, 2)
        AC_CFG(2, 2, 3, 2)
        AC_CFG(1, 2, 2, 0)      // 64 x 128, 4 waves
        AC_CFG(1, 2, 4, 0)
        AC_CFG(1, 2, 4, 1)
        AC_CFG(1, 2, 8, 1)
        AC_CFG(1, 4, 3, 0)      // 128 x 128, 8 waves of 32 x 64
        AC_CFG(1, 4, 3, 1)
        AC_CFG(1, 4, 3, 2)
        AC_CFG(1, 4, 6, 1)
        AC_CFG(1, 4, 6, 2)
        AC_CFG(2, 4, 2, 0)      // 256 x 128, 8 waves
        AC_CFG(2, 4, 3, 1)
        AC_CFG(2, 4, 4, 1)
        AC_CFG(2, 4, 4, 2)
#undef AC_CFG
        default: break;
    }
    set_error("gemm_pipe: configuration %d not built", cfg);
    return AC_EUNSUPPORTED;
}

}  // namespace ac
