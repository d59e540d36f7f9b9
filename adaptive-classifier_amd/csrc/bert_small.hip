// Small-batch BERT forward in ONE persistent launch (single-query predict(): classifier.py:1249-1282 with one short text).
// With <= 32 token rows every kernel of the layer-by-layer path is a few microseconds of work behind ~10 us of launch
// and dependency latency (86 launches, 1.0 ms); here the 12 layers run as phases of one kernel of 192 co-resident
// workgroups separated by fence-free grid barriers (1.7 us each), activations exchanged with sc1 (agent-coherent) stores / loads:
//
//   P0  embeddings: row t = word[id] + type[tt] + pos[p]                      (pre-LayerNorm) -> y0           | barrier
//   per layer:
//   PA  x = LN(y0) on the fly;  qkv = x Wqkv^T + b                                               -> xn, qkv    | barrier
//   PB  attention per head over the rows of each sequence (additive key mask), softmax, P V       -> ctx        | barrier
//   PC  y1 = ctx Wo^T + b + x                                                                                   | barrier
//   PD  x1 = LN(y1) on the fly;  ffn = gelu(x1 W1^T + b)                                          -> x1, ffn    | barrier
//   PE  y0 = ffn W2^T + b + x1                                                                                  | barrier
//   end x = LN(y0)[CLS rows], L2-normalised                                                       -> out
//
// GEMM phases: strict fp32 on v_mfma_f32_16x16x4_f32.  A workgroup owns one group of 16 output columns (FFN2: one group
// and a quarter / half of K, the parts added up by the consumers while they load them); its
// eight waves split K; lane (m, kk) = (lane & 15, lane >> 4) holds float4 A[m][k0 + 4 kk ..] and float4 W[n0 + m][k0 + 4 kk ..]
// -- four MFMA k-steps per load pair, 64-byte runs per row for both operands; the eight partial 32 x 16 tiles meet in
// LDS, where bias / GELU / residual are applied.  The LayerNorm in front of PA / PD is computed by every consumer from
// the fragments it already holds (two-pass statistics as ln_kernel); the rows of the normalised activations are written
// out by the phase's workgroups (row r by workgroup r mod nactive) for the residual connections.
#include "common.h"
#include "grid_sync.h"

#include <math.h>

#include <atomic>

namespace {

using namespace acp;

constexpr int kT = 512;              // threads per workgroup (8 waves)
constexpr int kTok = 32;             // token rows covered (two MFMA row tiles)
constexpr int kMaxLayers = 24;
constexpr int kDH = 64;
constexpr int kKB = 6;               // k-blocks of 16 per wave per operand chunk (K / 128 for K = 768)
constexpr int kKBmax = 12;           // ... 12 in the K = 3072 phase of a single row tile (all 24 k-blocks in flight at once)

struct LayerPtrs {
    const float *qkv_w, *qkv_b, *ao_w, *ao_b, *ln1_g, *ln1_b, *ff1_w, *ff1_b, *ff2_w, *ff2_b, *ln2_g, *ln2_b;
};

struct SmallParams {
    int H, I, L, heads, T, b, S, G;
    float eps;
    const int64_t* ids; const int64_t* type_ids; const int64_t* mask;
    const float *word, *pos, *type, *emb_g, *emb_b;
    float *y0, *xn, *qkv, *ctx, *y1, *x1, *ffn;      // [kTok, .] activations
    float* out; int64_t ldo;
    GridCtl* ctl;
    unsigned long long* dbg;       // AC_BERT_SMALL_DEBUG: s_memtime stamps, [workgroup < 4][layer][12]
    LayerPtrs layer[kMaxLayers];
};
typedef const SmallParams __attribute__((address_space(4))) * KArgs;

typedef float f32x4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
    u32x4_t u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 16);
}

// LDS layout (floats)
constexpr int kRed = 8 * 2 * 256;            // partial output tiles of the eight waves
constexpr int kStat = 8 * kTok;              // row statistics partials
constexpr int kAttnLd = kDH + 4;

struct Lds {
    float* red; float* stat; float* rowv; float* q; float* k; float* v; float* p; unsigned* flag;
};

// sum over the 8 waves x 4 k-slices of a per-lane partial of row (tile tt, m): every lane gets its two rows' totals
__device__ __forceinline__ void row_totals(float& s0, float& s1, const Lds& L, int wave, int lane) {
    // over kk (lanes m, m+16, m+32, m+48)
    s0 += __shfl_xor(s0, 16); s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
    if (lane < 16) { L.stat[wave * kTok + lane] = s0; L.stat[wave * kTok + 16 + lane] = s1; }
    __syncthreads();
    const int m = lane & 15;
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { t0 += L.stat[w * kTok + m]; t1 += L.stat[w * kTok + 16 + m]; }
    __syncthreads();
    s0 = t0; s1 = t1;
}

// One GEMM phase.  A [kTok, K] (global, written by other workgroups in an earlier phase), W [N, K], out [kTok, N].
// LN: K == H, nkb <= kKB: A is LayerNorm(g_ln, b_ln) of the loaded rows; the active workgroups also store the normalised
// rows (row r by workgroup r mod nactive).  Memory round trips are what a phase costs (~2 us each, HBM for W, the memory
// side for the sc1 reads), so: the first W chunk is requested BEFORE the grid barrier in front of the phase (WPre), the
// residual / bias of the epilogue at its start, and the work is a flat stream of (column group, K chunk) units whose
// operands are requested one unit ahead.
struct WPre { float4 w[kKBmax]; };

struct GemmGeom {
    int ngroups, g_lo, g_hi, nkb, kw, nchunks, nactive, rot, ks;
};
template <int KB>
__device__ __forceinline__ GemmGeom gemm_geom(int N, int K, int groups_per_block, int blk, int tid, int ksplit = 1) {
    GemmGeom g;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    g.ngroups = N / 16;
    // ksplit > 1 (one group per workgroup): workgroup blk takes column group blk % ngroups and the ks-th part of K; the
    // parts' partial results go to separate buffers that the CONSUMER adds up while it loads them
    g.ks = ksplit > 1 ? blk / g.ngroups : 0;
    const int bg = ksplit > 1 ? blk - g.ks * g.ngroups : blk;
    g.g_lo = (ksplit > 1 && g.ks >= ksplit) ? g.ngroups : bg * groups_per_block;
    g.g_hi = min(g.ngroups, g.g_lo + groups_per_block);
    const int Kb = K / ksplit;
    g.nkb = Kb / 128;                                  // k-blocks of 16 per wave
    // which eighth of K a wave takes, and (below) the order of its chunks, rotate with the workgroup: dozens of workgroups
    // read the SAME activation rows at the same moment, and in lock step they would queue on the same memory channels
    g.kw = g.ks * Kb + ((wave + blk) & 7) * (Kb / 8) + 4 * (lane >> 4);           // + 16 * kb
    g.rot = blk;
    g.nchunks = (g.nkb + KB - 1) / KB;
    g.nactive = (g.ngroups + groups_per_block - 1) / groups_per_block;
    return g;
}
template <int KB, int NW>
__device__ __forceinline__ void load_w(float4 (&w)[NW], __amdgpu_buffer_rsrc_t rW, const GemmGeom& g, int K, int unit, int m) {
    const int gi = g.g_lo + unit / g.nchunks, kb0 = ((unit % g.nchunks + g.rot) % g.nchunks) * KB;
    const bool live = gi < g.g_hi;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
        w[kb] = ld4_buf(rW, (live && kb0 + kb < g.nkb) ? (unsigned)(((size_t)(gi * 16 + m) * K + g.kw + 16 * (kb0 + kb)) * 4) : 0xffffff00u);
}
template <int KB>
__device__ __forceinline__ WPre prefetch_w(const float* W, int N, int K, int groups_per_block, int blk, int tid, int ksplit = 1) {
    WPre p;
    const GemmGeom g = gemm_geom<KB>(N, K, groups_per_block, blk, tid, ksplit);
    load_w<KB>(p.w, make_rsrc(W, (unsigned)((size_t)N * K * sizeof(float))), g, K, 0, tid & 15);
    return p;
}

template <bool LN, int ACT, bool RES, int KB, bool TWO, int NSUM = 1>
__device__ __forceinline__ void gemm_phase(const float* A, int K, const float* g_ln, const float* b_ln, float eps, float* xn_out,
                                           const float* W, const float* bias, const float* R, float* out, int N, int T,
                                           int groups_per_block, const Lds& L, int tid, int blk, const WPre& pre, unsigned long long* dbgp = nullptr,
                                           int ksplit = 1) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15;
    auto st = [&](int q) { if (dbgp && tid == 0) dbgp[q] = __builtin_readcyclecounter(); };
    st(0);
    const GemmGeom g = gemm_geom<KB>(N, K, groups_per_block, blk, tid, ksplit);
    if (g.g_lo >= g.ngroups) return;
    out += (size_t)g.ks * kTok * N;                 // (K split: this part's partial-result buffer)
    constexpr bool two = TWO;
    constexpr int KB1 = TWO ? KB : 1;       // (the second row tile's fragments exist only when it does)
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(A, (unsigned)((NSUM > 1 ? NSUM * kTok : two ? kTok : 16) * K * sizeof(float)));
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(W, (unsigned)((size_t)N * K * sizeof(float)));
    auto load_a = [&](float4 (&a0)[KB], float4 (&a1)[KB1], int unit) {
        const int kb0 = ((unit % g.nchunks + g.rot) % g.nchunks) * KB;
        const bool live = g.g_lo + unit / g.nchunks < g.g_hi;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const bool in = live && kb0 + kb < g.nkb;
            float4 p0[NSUM], p1[NSUM];
#pragma unroll
            for (int sp = 0; sp < NSUM; ++sp) {            // (NSUM > 1: the producer split K; its partial buffers are added here, in order)
                p0[sp] = ld4_sc1(rA, in ? (unsigned)(((sp * kTok + m) * K + g.kw + 16 * (kb0 + kb)) * 4) : 0xffffff00u);
                if (TWO) p1[sp] = ld4_sc1(rA, in ? (unsigned)(((sp * kTok + 16 + m) * K + g.kw + 16 * (kb0 + kb)) * 4) : 0xffffff00u);
            }
#pragma unroll
            for (int sp = 1; sp < NSUM; ++sp) {
                p0[0].x += p0[sp].x; p0[0].y += p0[sp].y; p0[0].z += p0[sp].z; p0[0].w += p0[sp].w;
                if (TWO) { p1[0].x += p1[sp].x; p1[0].y += p1[sp].y; p1[0].z += p1[sp].z; p1[0].w += p1[sp].w; }
            }
            a0[kb] = p0[0];
            if (TWO) a1[kb < KB1 ? kb : 0] = p1[0];
        }
    };
    // epilogue operands of the first group, requested now: thread (tt, l, i) owns token 16 tt + 4 (l >> 4) + i, column l & 15
    const int e_tt = tid >> 8, e_e = tid & 255, e_l = e_e >> 2, e_i = e_e & 3;
    const int token = 16 * e_tt + 4 * (e_l >> 4) + e_i, ecol = e_l & 15;
    float rv = 0.f, bv = 0.f;
    auto load_epi = [&](int gi) {
        const int n = gi * 16 + ecol;
        if (token < T && gi < g.g_hi && g.ks == 0) { bv = bias[n]; if (RES) rv = ld_sc1(R + (size_t)token * N + n); }
    };
    load_epi(g.g_lo);
    float4 a0A[KB], a1A[KB1], a0B[KB], a1B[KB1], wA[KB], wB[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) wA[kb] = pre.w[kb];
    load_a(a0A, a1A, 0);
    if (dbgp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st(1); }
    if (LN) {
        // ---- LayerNorm statistics over the full rows (the K slices of the eight waves meet in LDS) ----
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            s0 += (a0A[kb].x + a0A[kb].y) + (a0A[kb].z + a0A[kb].w);
            if (TWO) s1 += (a1A[kb < KB1 ? kb : 0].x + a1A[kb < KB1 ? kb : 0].y) + (a1A[kb < KB1 ? kb : 0].z + a1A[kb < KB1 ? kb : 0].w);
        }
        row_totals(s0, s1, L, wave, lane);
        const float mean0 = s0 / (float)K, mean1 = s1 / (float)K;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb < g.nkb) {
                const float d0 = a0A[kb].x - mean0, d1 = a0A[kb].y - mean0, d2 = a0A[kb].z - mean0, d3 = a0A[kb].w - mean0;
                q0 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                if (TWO) {
                    const float4 t1 = a1A[kb < KB1 ? kb : 0];
                    const float e0 = t1.x - mean1, e1 = t1.y - mean1, e2 = t1.z - mean1, e3 = t1.w - mean1;
                    q1 += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
                }
            }
        }
        row_totals(q0, q1, L, wave, lane);
        const float rstd0 = 1.0f / sqrtf(q0 / (float)K + eps), rstd1 = 1.0f / sqrtf(q1 / (float)K + eps);
        const __amdgpu_buffer_rsrc_t rX = make_rsrc(xn_out, (unsigned)(kTok * K * sizeof(float)));
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb < g.nkb) {
                const float4 gg = *reinterpret_cast<const float4*>(g_ln + g.kw + 16 * kb);
                const float4 bb = *reinterpret_cast<const float4*>(b_ln + g.kw + 16 * kb);
                a0A[kb].x = (a0A[kb].x - mean0) * rstd0 * gg.x + bb.x; a0A[kb].y = (a0A[kb].y - mean0) * rstd0 * gg.y + bb.y;
                a0A[kb].z = (a0A[kb].z - mean0) * rstd0 * gg.z + bb.z; a0A[kb].w = (a0A[kb].w - mean0) * rstd0 * gg.w + bb.w;
                if (TWO) {
                    float4& t1 = a1A[kb < KB1 ? kb : 0];
                    t1.x = (t1.x - mean1) * rstd1 * gg.x + bb.x; t1.y = (t1.y - mean1) * rstd1 * gg.y + bb.y;
                    t1.z = (t1.z - mean1) * rstd1 * gg.z + bb.z; t1.w = (t1.w - mean1) * rstd1 * gg.w + bb.w;
                }
                // the normalised rows, for the residual connection two phases on: row r by workgroup r mod nactive
                for (int r = blk; r < T; r += g.nactive)
                    if (m == (r & 15)) st4_sc1(rX, (unsigned)((r * K + g.kw + 16 * kb) * 4), (r < 16 || !TWO) ? a0A[kb] : a1A[kb < KB1 ? kb : 0]);
            }
        }
    }
    f32x4_ acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto mfmas = [&](const float4 (&a0)[KB], const float4 (&a1)[KB1], const float4 (&w)[KB]) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kb].x, w[kb].x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kb].y, w[kb].y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kb].z, w[kb].z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kb].w, w[kb].w, acc0, 0, 0, 0);
            if (TWO) {
                const float4 av = a1[kb < KB1 ? kb : 0];
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, w[kb].x, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, w[kb].y, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, w[kb].z, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, w[kb].w, acc1, 0, 0, 0);
            }
        }
    };
    // the eight K-partials of a finished column group meet in LDS; bias / GELU / residual; next group's epilogue operands
    auto finish_group = [&](int gi) {
        *reinterpret_cast<f32x4_*>(L.red + (wave * 2 + 0) * 256 + lane * 4) = acc0;
        *reinterpret_cast<f32x4_*>(L.red + (wave * 2 + 1) * 256 + lane * 4) = acc1;
        acc0 = f32x4_{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4_{0.f, 0.f, 0.f, 0.f};
        st(2);
        __syncthreads();
        st(3);
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += L.red[(w8 * 2 + e_tt) * 256 + e_e];
        if (token < T) {
            v += bv;
            if (ACT == 2) v = ac::gelu_erf(v);
            if (RES) v += rv;
            st_sc1(out + (size_t)token * N + gi * 16 + ecol, v);
        }
        load_epi(gi + 1);
        __syncthreads();
        st(4);
    };
    const int nunits = (g.g_hi - g.g_lo) * g.nchunks;
#pragma unroll 1
    for (int u = 0; u < nunits; u += 2) {
        // unit u in the A buffers, unit u + 1 requested into the B buffers (LN phases keep the one A chunk: nchunks == 1)
        load_w<KB>(wB, rW, g, K, u + 1, m);
        if (!LN) load_a(a0B, a1B, u + 1);
        mfmas(a0A, a1A, wA);
        if ((u + 1) % g.nchunks == 0) finish_group(g.g_lo + u / g.nchunks);
        if (u + 1 < nunits) {
            load_w<KB>(wA, rW, g, K, u + 2, m);
            if (!LN) load_a(a0A, a1A, u + 2);
            if (LN) mfmas(a0A, a1A, wB); else mfmas(a0B, a1B, wB);
            if ((u + 2) % g.nchunks == 0) finish_group(g.g_lo + (u + 1) / g.nchunks);
        }
    }
}

template <bool TWO>       // TWO: 17 .. 32 token rows (two MFMA row tiles); otherwise one
__global__ __launch_bounds__(kT) void bert_small_kernel(const SmallParams prm_) {
    constexpr int kSplitE = TWO ? 2 : 4;            // FFN2 (K = I): K split over 2 / 4 workgroups per column group; its consumers add the parts
    constexpr int kKBE = kKB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid0 = threadIdx.x, blk = blockIdx.x;
    Lds L;
    L.red = lds; L.stat = L.red + kRed; L.rowv = L.stat + kStat;
    L.q = L.rowv + 64; L.k = L.q + kTok * kAttnLd; L.v = L.k + kTok * kAttnLd; L.p = L.v + kTok * kAttnLd;
    L.flag = reinterpret_cast<unsigned*>(L.p + kTok * (kTok + 1));
    const int H = ka->H, I = ka->I, T = ka->T, S = ka->S, G = ka->G, nlayers = ka->L, heads = ka->heads;
    const float eps = ka->eps;
    unsigned bar = 0;
    // a barrier that gave up (a workgroup never arrived: the grid was not co-resident): poison the output, do not hang
    auto bail = [&]() { if (blk < ka->b) for (int c = tid0; c < ka->ldo; c += kT) ka->out[(size_t)blk * ka->ldo + c] = __builtin_nanf(""); };
    // ---- P0: embeddings (pre-LayerNorm), one workgroup per token row ----
    if (blk < T) {
        const int64_t id = ka->ids[blk];
        const int64_t tt = ka->type_ids ? ka->type_ids[blk] : 0;
        const int p = blk % S;
        const __amdgpu_buffer_rsrc_t rY = make_rsrc(ka->y0, (unsigned)(4 * kTok * H * sizeof(float)));
        for (int c4 = tid0; c4 < H / 4; c4 += kT) {
            const float4 w = *reinterpret_cast<const float4*>(ka->word + id * H + 4 * c4);
            const float4 ty = *reinterpret_cast<const float4*>(ka->type + tt * H + 4 * c4);
            const float4 po = *reinterpret_cast<const float4*>(ka->pos + (int64_t)p * H + 4 * c4);
            st4_sc1(rY, (unsigned)((blk * H + 4 * c4) * 4), make_float4((w.x + ty.x) + po.x, (w.y + ty.y) + po.y, (w.z + ty.z) + po.z, (w.w + ty.w) + po.w));
            for (int sp = 1; sp < kSplitE; ++sp) st4_sc1(rY, (unsigned)(((sp * kTok + blk) * H + 4 * c4) * 4), make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    grid_arrive(ka->ctl);
    WPre wpre = prefetch_w<kKB>(ka->layer[0].qkv_w, 3 * H, H, 1, blk, tid0);
    if (!grid_wait<false>(ka->ctl, ++bar, G, L.flag)) { bail(); return; }
    const float scale = 1.0f / sqrtf((float)kDH);
#pragma unroll 1
    for (int l = 0; l < nlayers; ++l) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        KArgs prm = ka;
        asm volatile("" : "+s"(prm));
        auto stamp = [&](int q) { if (prm->dbg && tid == 0 && (blk == 0 || blk == 40 || blk == 100)) prm->dbg[((blk == 0 ? 0 : blk == 40 ? 1 : 2) * kMaxLayers + l) * 12 + q] = __builtin_readcyclecounter(); };
        stamp(0);
        const float* g_in = l == 0 ? prm->emb_g : prm->layer[l - 1].ln2_g;
        const float* b_in = l == 0 ? prm->emb_b : prm->layer[l - 1].ln2_b;
        // ---- PA: x = LN(y0); qkv = x Wqkv^T + b ----
        gemm_phase<true, 0, false, kKB, TWO, kSplitE>(prm->y0, H, g_in, b_in, eps, prm->xn, prm->layer[l].qkv_w, prm->layer[l].qkv_b, nullptr, prm->qkv,
                                   3 * H, T, 1, L, tid, blk, wpre);
        stamp(1);
        grid_arrive(prm->ctl);
        wpre = prefetch_w<kKB>(prm->layer[l].ao_w, H, H, 1, blk, tid);                 // (PC's first W chunk: in flight across PB)
        if (!grid_wait<false>(prm->ctl, ++bar, G, L.flag)) { bail(); return; }
        stamp(2);
        // ---- PB: attention, one workgroup per head ----
        if (blk < heads) {
            const int h = blk;
            const __amdgpu_buffer_rsrc_t rQ = make_rsrc(prm->qkv, (unsigned)(kTok * 3 * H * sizeof(float)));
            for (int e = tid; e < 3 * T * (kDH / 4); e += kT) {            // q, k, v rows of this head -> LDS
                const int which = e / (T * (kDH / 4)), r = e - which * (T * (kDH / 4)), t = r / (kDH / 4), d4 = r - t * (kDH / 4);
                const float4 v = ld4_sc1(rQ, (unsigned)((t * 3 * H + which * H + h * kDH + 4 * d4) * 4));
                float* dst = (which == 0 ? L.q : which == 1 ? L.k : L.v) + t * kAttnLd + 4 * d4;
                *reinterpret_cast<float4*>(dst) = v;
            }
            __syncthreads();
            // scores + softmax: query row i = tid >> 4 by the 16 lanes of a DPP row, lane j16 takes keys j16 and j16 + 16
            {
                const int64_t* mask = prm->mask;
                const int i = tid >> 4, j16 = tid & 15;
                float sc2[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int j = j16 + 16 * r;
                    float sc = -INFINITY;
                    if (i < T && j < T && i / S == j / S && (!mask || mask[j] != 0)) {
                        float acc = 0.f;
#pragma unroll
                        for (int d = 0; d < kDH; d += 4) {
                            const float4 qv = *reinterpret_cast<const float4*>(L.q + i * kAttnLd + d);
                            const float4 kv = *reinterpret_cast<const float4*>(L.k + j * kAttnLd + d);
                            acc = fmaf(qv.x, kv.x, acc); acc = fmaf(qv.y, kv.y, acc); acc = fmaf(qv.z, kv.z, acc); acc = fmaf(qv.w, kv.w, acc);
                        }
                        sc = acc * scale;
                    }
                    sc2[r] = sc;
                }
                float mx = fmaxf(sc2[0], sc2[1]);
                mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, true)));
                mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x4E, 0xf, 0xf, true)));
                mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x141, 0xf, 0xf, true)));
                mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x140, 0xf, 0xf, true)));
                const float e0 = mx == -INFINITY ? 0.f : expf(sc2[0] - mx), e1 = mx == -INFINITY ? 0.f : expf(sc2[1] - mx);
                const float sum = row16_sum(e0 + e1);
                const float inv = sum > 0.f ? 1.f / sum : 0.f;
                if (i < T) { L.p[i * (kTok + 1) + j16] = e0 * inv; L.p[i * (kTok + 1) + j16 + 16] = e1 * inv; }
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t rC = make_rsrc(prm->ctx, (unsigned)(kTok * H * sizeof(float)));
            for (int e = tid; e < T * (kDH / 4); e += kT) {                 // ctx = P V
                const int i = e / (kDH / 4), d4 = e - i * (kDH / 4);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int j = 0; j < T; ++j) {
                    const float pv = L.p[i * (kTok + 1) + j];
                    const float4 vv = *reinterpret_cast<const float4*>(L.v + j * kAttnLd + 4 * d4);
                    acc.x = fmaf(pv, vv.x, acc.x); acc.y = fmaf(pv, vv.y, acc.y); acc.z = fmaf(pv, vv.z, acc.z); acc.w = fmaf(pv, vv.w, acc.w);
                }
                st4_sc1(rC, (unsigned)((i * H + h * kDH + 4 * d4) * 4), acc);
            }
        }
        stamp(3);
        if (!grid_barrier<false>(prm->ctl, ++bar, G, L.flag)) { bail(); return; }
        stamp(4);
        // ---- PC: y1 = ctx Wo^T + b + x ----
        gemm_phase<false, 0, true, kKB, TWO>(prm->ctx, H, nullptr, nullptr, eps, nullptr, prm->layer[l].ao_w, prm->layer[l].ao_b, prm->xn, prm->y1,
                                   H, T, 1, L, tid, blk, wpre, (prm->dbg && blk == 0 && l == 1) ? prm->dbg + 3 * kMaxLayers * 12 : nullptr);
        stamp(5);
        grid_arrive(prm->ctl);
        wpre = prefetch_w<kKB>(prm->layer[l].ff1_w, I, H, 1, blk, tid);
        if (!grid_wait<false>(prm->ctl, ++bar, G, L.flag)) { bail(); return; }
        stamp(6);
        // ---- PD: x1 = LN(y1); ffn = gelu(x1 W1^T + b) ----
        gemm_phase<true, 2, false, kKB, TWO>(prm->y1, H, prm->layer[l].ln1_g, prm->layer[l].ln1_b, eps, prm->x1, prm->layer[l].ff1_w,
                                   prm->layer[l].ff1_b, nullptr, prm->ffn, I, T, 1, L, tid, blk, wpre);
        stamp(7);
        grid_arrive(prm->ctl);
        wpre = prefetch_w<kKBE>(prm->layer[l].ff2_w, H, I, 1, blk, tid, kSplitE);
        if (!grid_wait<false>(prm->ctl, ++bar, G, L.flag)) { bail(); return; }
        stamp(8);
        // ---- PE: y0 = ffn W2^T + b + x1 ----
        gemm_phase<false, 0, true, kKBE, TWO>(prm->ffn, I, nullptr, nullptr, eps, nullptr, prm->layer[l].ff2_w, prm->layer[l].ff2_b, prm->x1, prm->y0,
                                   H, T, 1, L, tid, blk, wpre, nullptr, kSplitE);
        stamp(9);
        grid_arrive(prm->ctl);
        if (l + 1 < nlayers) wpre = prefetch_w<kKB>(prm->layer[l + 1].qkv_w, 3 * H, H, 1, blk, tid);
        if (!grid_wait<false>(prm->ctl, ++bar, G, L.flag)) { bail(); return; }
        stamp(10);
    }
    // ---- end: LN of the CLS rows, L2-normalise (F.normalize eps 1e-12), one workgroup (its first wave) per sequence ----
    if (blk < ka->b && tid0 < 64) {
        const int lane = tid0;
        const float* src = ka->y0 + (size_t)blk * S * H;
        const float* gl = ka->layer[nlayers - 1].ln2_g;
        const float* bl = ka->layer[nlayers - 1].ln2_b;
        float x[16];                                                       // H <= 1024
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = lane + 64 * e;
            x[e] = 0.f;
            if (c < H) for (int sp = 0; sp < kSplitE; ++sp) x[e] += ld_sc1(src + (size_t)sp * kTok * H + c);
            s += x[e];
        }
        const float mean = wave_sum(s) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const int c = lane + 64 * e; if (c < H) { const float d = x[e] - mean; q += d * d; } }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
        float n2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const int c = lane + 64 * e; if (c < H) { x[e] = (x[e] - mean) * rstd * gl[c] + bl[c]; n2 = fmaf(x[e], x[e], n2); } }
        const float nrm = fmaxf(sqrtf(wave_sum(n2)), 1e-12f);
        float* dst = ka->out + (size_t)blk * ka->ldo;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const int c = lane + 64 * e; if (c < H) dst[c] = x[e] / nrm; }
        for (int c = H + lane; c < ka->ldo; c += 64) dst[c] = 0.f;
    }
}

size_t small_lds_bytes() {
    return (size_t)(kRed + kStat + 64 + 3 * kTok * kAttnLd + kTok * (kTok + 1)) * sizeof(float) + 16;
}

size_t small_act_floats(int H, int I) { return (size_t)kTok * ((size_t)8 * H + 3 * (size_t)H + I); }       // y0 x 4 parts

}  // namespace

namespace ac {

static std::atomic<long long> g_small_launches{0};
long long bert_small_launches() { return g_small_launches.load(std::memory_order_relaxed); }


size_t bert_small_ws_bytes(int H, int I) {
    return align_up(small_act_floats(H, I) * sizeof(float), 256) + align_up(sizeof(GridCtl), 256) + (3 * kMaxLayers * 12 + 16) * sizeof(unsigned long long);
}

// AC_OK: encoded.  1: shape outside what the persistent kernel covers (caller runs the layer-by-layer path).
int bert_small_encode(const ac_bert_config& c, const ac_bert_weights& w, const int64_t* ids, const int64_t* type_ids,
                      const int64_t* mask, int b, int S, float* out, int64_t ldo, void* ws, hipStream_t stream) {
    const int T = b * S;
    if (!(persistent_mask() & 2) || T < 1 || T > kTok) return 1;
    if (c.hidden > 768 || (c.hidden % 128) || (c.intermediate % 128) || c.layers > kMaxLayers || c.hidden != c.heads * kDH) return 1;
    if (!w.type_emb || !w.pos_emb) return 1;
    const int G = 192 < dev_info().cus ? 192 : dev_info().cus;
    if (G < c.intermediate / 16 || G < 3 * c.hidden / 16 || G < c.heads || G < 4 * (c.hidden / 16) || (c.intermediate % 512)) return 1;      // one 16-column group per workgroup in every phase
    SmallParams p;
    p.H = c.hidden; p.I = c.intermediate; p.L = c.layers; p.heads = c.heads; p.T = T; p.b = b; p.S = S; p.G = G; p.eps = c.ln_eps;
    p.ids = ids; p.type_ids = type_ids; p.mask = mask;
    p.word = w.word_emb; p.pos = w.pos_emb; p.type = w.type_emb; p.emb_g = w.emb_ln_g; p.emb_b = w.emb_ln_b;
    float* f = (float*)ws;
    const size_t H = c.hidden, I = c.intermediate;
    p.y0 = f; f += 4 * kTok * H; p.xn = f; f += kTok * H; p.qkv = f; f += kTok * 3 * H; p.ctx = f; f += kTok * H;
    p.y1 = f; f += kTok * H; p.x1 = f; f += kTok * H; p.ffn = f; f += kTok * I;
    p.ctl = (GridCtl*)((char*)ws + align_up(small_act_floats(c.hidden, c.intermediate) * sizeof(float), 256));
    p.out = out; p.ldo = ldo;
    static const int debug = [] { const char* e = getenv("AC_BERT_SMALL_DEBUG"); return e ? atoi(e) : 0; }();
    p.dbg = debug ? (unsigned long long*)((char*)p.ctl + align_up(sizeof(GridCtl), 256)) : nullptr;
    for (int l = 0; l < c.layers; ++l) {
        LayerPtrs& q = p.layer[l];
        q.qkv_w = w.qkv_w[l]; q.qkv_b = w.qkv_b[l]; q.ao_w = w.ao_w[l]; q.ao_b = w.ao_b[l]; q.ln1_g = w.ln1_g[l]; q.ln1_b = w.ln1_b[l];
        q.ff1_w = w.ff1_w[l]; q.ff1_b = w.ff1_b[l]; q.ff2_w = w.ff2_w[l]; q.ff2_b = w.ff2_b[l]; q.ln2_g = w.ln2_g[l]; q.ln2_b = w.ln2_b[l];
    }
    AC_HIP_CHECK(hipMemsetAsync(p.ctl, 0, sizeof(GridCtl), stream));
    const size_t lds = small_lds_bytes();
    const void* fn = T > 16 ? (const void*)bert_small_kernel<true> : (const void*)bert_small_kernel<false>;
    static std::atomic<unsigned long long> attr_set[2] = {{0}, {0}};  // (function attributes are per device)
    if (first_call_on_device(attr_set[T > 16]))
        AC_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // 192 workgroups of one per CU (LDS / registers) on 256 CUs: co-resident by construction whenever the device is not shared
    // with another compute process; a plain launch then has the same residency as a cooperative one and saves its ~30 us of
    // launch overhead -- 5 % of a single-query predict().  AC_BERT_SMALL_COOP=1 asks for the checked cooperative launch; a
    // barrier that cannot complete gives up after a bounded spin and poisons the output with NaNs.
    // residency proof at launch (replaces "discover it by a barrier that gives up"): workgroups per CU (occupancy query) x the
    // CUs this process's workgroups reach (dev_info().cus, measured: a CU mask counts) must hold the grid; otherwise the
    // layer-by-layer path runs, chosen up front
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kT, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        if ((int64_t)per_cu * dev_info().cus < G) return 1;
    }
    g_small_launches.fetch_add(1, std::memory_order_relaxed);
    static const int coop = [] { const char* e = getenv("AC_BERT_SMALL_COOP"); return e ? atoi(e) : 0; }();
    if (coop) {
        void* args[] = {&p};
        AC_HIP_CHECK(hipLaunchCooperativeKernel(fn, dim3(G), dim3(kT), args, (unsigned)lds, stream));
    } else if (T > 16) {
        hipLaunchKernelGGL(bert_small_kernel<true>, dim3(G), dim3(kT), lds, stream, p);
        AC_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(bert_small_kernel<false>, dim3(G), dim3(kT), lds, stream, p);
        AC_LAUNCH_CHECK();
    }
    if (debug) {
        static unsigned long long h[3 * kMaxLayers * 12 + 16];
        AC_HIP_CHECK(hipStreamSynchronize(stream));
        AC_HIP_CHECK(hipMemcpy(h, p.dbg, sizeof(h), hipMemcpyDeviceToHost));
        static const char* names[10] = {"PA", "B", "PB", "B", "PC", "B", "PD", "B", "PE", "B"};
        for (int wg = 0; wg < 3; ++wg) {
            fprintf(stderr, "bert_small wg%d (cycles):", wg == 0 ? 0 : wg == 1 ? 40 : 100);
            for (int q = 0; q < 10; ++q) {
                double acc = 0;
                for (int l = 1; l < c.layers; ++l) acc += (double)(h[(wg * kMaxLayers + l) * 12 + q + 1] - h[(wg * kMaxLayers + l) * 12 + q]);
                fprintf(stderr, " %s %.0f", names[q], acc / (c.layers - 1));
            }
            fprintf(stderr, "\n");
        }
        { const unsigned long long* q = h + 3 * kMaxLayers * 12; fprintf(stderr, "  PC wg0 layer1: A-wait %llu  mfma %llu  sync %llu  epilogue %llu\n", q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3]); }
    }
    return AC_OK;
}

// abort flag of the last bert_small_encode launch in this workspace (1 = a grid barrier gave up and the output rows are NaN).
// Synchronises the stream (4-byte D2H).
int bert_small_aborted(int H, int I, const void* ws, hipStream_t stream, int* aborted) {
    const GridCtl* ctl = (const GridCtl*)((const char*)ws + align_up(small_act_floats(H, I) * sizeof(float), 256));
    unsigned v = 0;
    AC_HIP_CHECK(hipMemcpyAsync(&v, &ctl->abort_, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    AC_HIP_CHECK(hipStreamSynchronize(stream));
    *aborted = v != 0;
    return AC_OK;
}

}  // namespace ac
