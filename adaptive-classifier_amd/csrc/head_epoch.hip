// Weights-stationary training epoch of the AdaptiveHead (classifier.py:1483-1507 / :327-353): ONE persistent launch
// runs every step of the epoch.  The 0.9 M parameters and their AdamW moments (p, m, v = 10.6 MB) are partitioned over
// the 256 CUs and live in LDS for the whole epoch; the only data that crosses CUs per step are the two hidden
// activations (a1 [32, H1], a2 [32, H2]), the tiny output layer and one pair of norm partials per workgroup.
//
//   ownership (workgroup g of G):  rows  [g r1, g r1 + r1) of W1 / b1         (r1 = ceil(H1 / G))
//                                  rows  [g r2, g r2 + r2) of W2 / b2, the same columns of W3   (r2 = ceil(H2 / G))
//                                  and a DUPLICATE of columns [g r1, ...) of W2 with their own m, v: the backward pass
//                                  needs W2 by column (d1 = d2 W2) while the forward needs it by row; both owners
//                                  compute the same gradient with the same fma chain and apply the same update, so the
//                                  two copies stay bit-identical and W2 never travels.
//   per step, three grid barriers:
//     P1  a1[:, own1] = drop(relu(X W1o^T + b1))                                   -> a1 to global      | B1
//     P2  a2[:, own2] = drop(relu(a1 W2r^T + b2))                                  -> a2 to global      | B2
//     P3  every workgroup: logits, loss, dz (B x C, redundantly), d2 = (dz W3) gated, in LDS
//     P4  gradients of everything it owns (registers), EWC term, partial sums of |g|^2              | B3
//     P5  clip coefficient from the G partials (fixed order), AdamW on the owned elements in LDS, W3 / b3 to global
//   Exchanged data travels as sc1 stores / loads (agent scope: coherent across the 8 XCD L2s), so the barriers need no
//   L2 write-back / invalidate: 1.8 us per barrier (eight counters polled by eight lanes) instead of 7.8 us with
//   release/acquire fences (tools/gridbar_probe.hip, grid_sync.h).  X, labels, Fisher and the EWC anchor are read-only for the launch: plain loads.
//
// Same arithmetic contract as the step-by-step path in head.hip (fp32, formulas of ewc_adamw_kernel and
// head_top_kernel); summation orders differ (fma chains over the batch / wave-strided dot products instead of MFMA
// tiles), which is inside the tolerance the head tests hold against torch.  ac_head_train_step and
// ac_head_train_epoch both route here when the shape fits, so "epoch == step loop" stays bit-identical.
#include "common.h"

#include <atomic>
#include "grid_sync.h"

#include <math.h>

#ifndef AC_EPOCH_ACQUIRE_B1
#define AC_EPOCH_ACQUIRE_B1 0
#endif

namespace {

using namespace acp;

constexpr int kT = 512;               // threads per workgroup
constexpr int kR1 = 4, kR2 = 2;       // most rows of layer 1 / layer 2 one workgroup owns
constexpr int kKU = 2;                // D, H1 <= kT * kKU (four columns per thread of a four-wave team)
constexpr int kMaxC = 16, kMaxB = 32;
constexpr int kMaxG = 512;

struct EpochParams {
    int D, H1, H2, C, G, r1, r2;
    float *P, *M, *V, *Gout;
    const float *F, *Old;
    const float* X; int64_t ldx; const int64_t* y; const float* T; int64_t ldt; const int64_t* order;
    int64_t n_total; int batch, loss_kind;
    float dropout_p; uint64_t seed0;
    float lambda_B, lam_direct, max_norm, lr, beta1, beta2, eps, wd; int step0;     // lam_direct >= 0: EWC weight given per step

    float* out; float* loss_accum;
    float* a1g; float* a2g; float* partials; acp::GridCtl* ctl;
    unsigned long long* dbg;          // AC_HEAD_EPOCH_DEBUG: s_memtime stamps of workgroup 0, [step < 16][16]
    int64_t o_w1, o_b1, o_w2, o_b2, o_w3, o_b3;
};

// kAcquireB1: barrier B1 ends with an acquire and a1 / W3 are read with plain (L2-cached) loads; otherwise every exchanged
// word is read with sc1 loads and no barrier fences (A/B on MI355X: see DESIGN)
constexpr bool kAcquireB1 = AC_EPOCH_ACQUIRE_B1;
template <bool PLAIN>
__device__ __forceinline__ float4 ld4_x(__amdgpu_buffer_rsrc_t r, unsigned byte_off) { return PLAIN ? ld4_buf(r, byte_off) : ld4_sc1(r, byte_off); }

// Sums of N <= 8 per-lane values over the wave through a wave-private LDS tile [64][9]: lane writes its N partials, lane
// (v, q) = (lane >> 3, lane & 7) adds the partials of value v from lanes 8q .. 8q+7 in order, three DPP butterflies finish.
// Lanes 8v .. 8v+7 return the sum of value v.  ~30 instructions per call instead of ~60 per VALUE with wave_sum(); both
// the write and the read pattern are bank-conflict free (strides 9 and 72 + 1).
template <int N>
__device__ __forceinline__ float wave_reduce_n(const float (&vals)[N], float* scr, int lane) {
#pragma unroll
    for (int i = 0; i < N; ++i) scr[lane * 9 + i] = vals[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int v = lane >> 3, q = lane & 7;
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) sum += scr[(q * 8 + l) * 9 + v];
    sum = dpp_add<0xB1>(sum); sum = dpp_add<0x4E>(sum); sum = dpp_add<0x141>(sum);
    __builtin_amdgcn_wave_barrier();
    return sum;
}
__device__ __forceinline__ float quad_sum(float v) { v = dpp_add<0xB1>(v); return dpp_add<0x4E>(v); }
// deterministic sums of two values over the workgroup (8 waves); every thread gets both results
__device__ __forceinline__ void block_sum8x2(float& a, float& b, float* sh) {
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = a; sh[8 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    a = ((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7]));
    b = ((sh[8] + sh[9]) + (sh[10] + sh[11])) + ((sh[12] + sh[13]) + (sh[14] + sh[15]));
    __syncthreads();
}

// beta^n in double by square-and-multiply (a few ulp of double from libm's pow -- invisible after the float conversion of
// lr / (1 - beta^n); generic pow() costs ~400 instructions per call site, which this kernel's instruction cache cannot spare)
__device__ __forceinline__ double ipow(double b, int n) {
    double r = 1.0;
#pragma unroll 1
    while (n > 0) { if (n & 1) r *= b; b *= b; n >>= 1; }
    return r;
}

struct StepScalars {
    float lam, two_lam, lr_wd, one_m_b1, one_m_b2, step_size, bc2_sqrt, s1, s2;
};

// one element's AdamW update on register values (formulas of ewc_adamw_kernel, head.hip).  The W2 duplicate must stay
// bit-identical to the row copy although the two are updated at different places in the code, so the contraction of every
// multiply-add is written out (fmaf) and the compiler may not choose its own.
__device__ __forceinline__ void adamw_vals(float& pi, float& mi, float& vi, float g_tot, float coef, const StepScalars& sc, float beta2,
                                           float eps) {
#pragma clang fp contract(off)
    const float gi = g_tot * coef;
    pi = pi * (1.f - sc.lr_wd);
    mi = fmaf(gi - mi, sc.one_m_b1, mi);
    vi = fmaf(sc.one_m_b2 * gi, gi, vi * beta2);
    const float denom = sqrtf(vi) / sc.bc2_sqrt + eps;
    pi = fmaf(-sc.step_size, mi / denom, pi);
}
__device__ __forceinline__ void adamw_elem(float* Ps, float* Ms, float* Vs, int e, float g_tot, float coef, const StepScalars& sc,
                                           float beta2, float eps) {
    float pi = Ps[e], mi = Ms[e], vi = Vs[e];
    adamw_vals(pi, mi, vi, g_tot, coef, sc, beta2, eps);
    Ps[e] = pi; Ms[e] = mi; Vs[e] = vi;
}

typedef const EpochParams __attribute__((address_space(4))) * KArgs;

// KC: bound on C for the per-thread class loops (4 or kMaxC).  R1 / R2: row capacity of a workgroup in layer 1 / 2
// (r1 <= R1, r2 <= R2 rows are real; slots past them hold zeros, take part in the arithmetic and never reach memory).
// Code size matters here: every step runs the whole kernel body once, so it has to stay inside the instruction cache
// (64 KB per CU pair) -- few guards, rolled class loops, and the kernel arguments re-read from the kernarg segment
// where they are used instead of sitting in ~60 scalar registers for the whole launch.
template <int KC, int R1, int R2>
__global__ __launch_bounds__(kT) void head_epoch_kernel(const EpochParams prm_) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid0 = threadIdx.x;
    const int g = blockIdx.x;
    const int D = ka->D, H1 = ka->H1, H2 = ka->H2, C = ka->C, G = ka->G;
    const int i0_ = g * ka->r1, j0_ = g * ka->r2;
    const int n1 = max(0, min(ka->r1, H1 - i0_)), n2 = max(0, min(ka->r2, H2 - j0_));
    // ---- state regions (element offsets inside each of the three LDS arrays), 16-byte aligned ----
    const int oW1 = 0, oB1 = R1 * D, oW2r = oB1 + 4, oB2 = oW2r + R2 * H1, oW2c = oB2 + 4, oW3 = oW2c + R1 * H2,
              oB3 = oW3 + kMaxC * R2, nstate = oB3 + kMaxC;
    float* Ps = lds;
    float* Ms = Ps + nstate;
    float* Vs = Ms + nstate;
    const int lda2 = H2 + 4;
    float* A2 = Vs + nstate;                       // [32][lda2]  a2, then d2 in place
    float* W3s = A2 + kMaxB * lda2;                // [kMaxC][H2] (rows >= C zero)
    float* a1own = W3s + kMaxC * H2;               // [32][4]
    float* a2own = a1own + kMaxB * 4;              // [32][2]
    float* d1s = a2own + kMaxB * 2;                // [32][4]
    float* zs = d1s + kMaxB * 4;                   // [32][16]
    float* dzs = zs + kMaxB * kMaxC;               // [32][16]
    float* rl = dzs + kMaxB * kMaxC;               // [32]
    float* b3s = rl + kMaxB;                       // [16]
    float* red = b3s + kMaxC;                      // [16]
    float* gsml = red + 16;                        // [kMaxC][2] gW3[:, own2] hand-over
    float* wscr = gsml + kMaxC * 2;                // [8 waves][64][9] wave_reduce_n tiles
    float* scal = wscr + 8 * 64 * 9;               // [2][8] step scalars, double-buffered
    int64_t* rowoff = reinterpret_cast<int64_t*>(scal + 16);    // [32] element offset of batch row b in X
    int64_t* rowidx = rowoff + kMaxB;                           // [32] row index (labels / targets)
    unsigned* flag = reinterpret_cast<unsigned*>(rowidx + kMaxB);

    // global index of state element e (-1: slot not backed by a parameter), primary = owned (not the W2 duplicate)
    auto gidx = [&](int e, bool& primary) -> int64_t {
        primary = true;
        if (e < oB1) { const int ii = e / D, k = e - ii * D; return ii < n1 ? ka->o_w1 + (int64_t)(i0_ + ii) * D + k : -1; }
        if (e < oW2r) { const int ii = e - oB1; return ii < n1 ? ka->o_b1 + i0_ + ii : -1; }
        if (e < oB2) { const int q = e - oW2r, jj = q / H1, i = q - jj * H1; return jj < n2 ? ka->o_w2 + (int64_t)(j0_ + jj) * H1 + i : -1; }
        if (e < oW2c) { const int jj = e - oB2; return jj < n2 ? ka->o_b2 + j0_ + jj : -1; }
        if (e < oW3) { const int q = e - oW2c, ii = q / H2, j = q - ii * H2; primary = false; return ii < n1 ? ka->o_w2 + (int64_t)j * H1 + i0_ + ii : -1; }
        if (e < oB3) { const int q = e - oW3, c = q / R2, jj = q - c * R2; return (c < C && jj < n2) ? ka->o_w3 + (int64_t)c * H2 + j0_ + jj : -1; }
        { const int c = e - oB3; primary = g == 0; return c < C ? ka->o_b3 + c : -1; }
    };
    {
        const float* P = ka->P; const float* M = ka->M; const float* V = ka->V;
#pragma unroll 1
        for (int e = tid0; e < nstate; e += kT) {
            bool prim;
            const int64_t gi = gidx(e, prim);
            Ps[e] = gi >= 0 ? P[gi] : 0.f;
            Ms[e] = gi >= 0 ? M[gi] : 0.f;
            Vs[e] = gi >= 0 ? V[gi] : 0.f;
        }
        for (int e = tid0; e < kMaxC * H2; e += kT) W3s[e] = 0.f;
        for (int e = tid0; e < kMaxB * kMaxC; e += kT) { zs[e] = 0.f; dzs[e] = 0.f; }
        if (tid0 < kMaxC) b3s[tid0] = 0.f;
        if (tid0 < kMaxB * 4) { a1own[tid0] = 0.f; d1s[tid0] = 0.f; }
        if (tid0 < kMaxB * 2) a2own[tid0] = 0.f;
    }
    // AdamW bias corrections in double like torch's Python floats; EWC weight lambda_B / rows of that batch
    auto write_scalars = [&](int si) {
        const int64_t offs = (int64_t)si * ka->batch;
        if (offs >= ka->n_total) return;
        const int nbs = (int)min((int64_t)ka->batch, ka->n_total - offs);
        const int step = ka->step0 + si;
        const double bc1 = 1.0 - ipow((double)ka->beta1, step);
        const double bc2 = 1.0 - ipow((double)ka->beta2, step);
        const float lam = ka->F == nullptr ? 0.f : (ka->lam_direct >= 0.f ? ka->lam_direct : (float)((double)ka->lambda_B / nbs));
        float* sq = scal + 8 * (si & 1);
        sq[0] = lam; sq[1] = 2.f * lam; sq[2] = (float)((double)ka->lr * (double)ka->wd);
        sq[3] = 1.f - ka->beta1; sq[4] = 1.f - ka->beta2; sq[5] = (float)((double)ka->lr / bc1);
        sq[6] = (float)sqrt(bc2);
    };
    if (tid0 == kT - 1) write_scalars(0);
    __syncthreads();

    const int64_t n_total = ka->n_total;
    const int batch = ka->batch;
    unsigned bar = 0;
    // a barrier that gave up (the grid was not co-resident -- a device shared with another compute process): nothing is
    // written back; the epoch loss and the step outputs are poisoned with NaNs so that the host notices and re-runs the epoch
    // through the step-by-step launches (training.py / classifier.py)
    auto bail = [&]() {
        if (g == 0 && tid0 == 0) {
            const float nanv = __builtin_nanf("");
            ka->out[0] = nanv; ka->out[1] = nanv; ka->out[2] = nanv;
            if (ka->loss_accum) *ka->loss_accum = nanv;
        }
    };
    int step_i = 0;
#pragma unroll 1
    for (int64_t off = 0; off < n_total; off += batch, ++step_i) {
        // fresh (opaque) copies per step of the thread id, the kernarg pointer and the ownership bases: keeps the compiler
        // from hoisting every address of the step out of the loop and holding hundreds of them live across it
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        KArgs prm = ka;
        int i0 = i0_, j0 = j0_;
        asm volatile("" : "+s"(prm), "+s"(i0), "+s"(j0));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: wave-level branches stay uniform)
        const int nb = (int)min((int64_t)batch, n_total - off);
        const bool last_step = off + batch >= n_total;
        if (nb < kMaxB && tid >= nb * 2 && tid < kMaxB * 2) a2own[tid] = 0.f;      // (short last batch: stale rows out)
        if (nb < kMaxB && tid >= nb * 4 && tid < kMaxB * 4) a1own[tid] = 0.f;
        if (tid < kMaxB) {                 // (rows past the batch alias its first row: loaded unconditionally, weighted by zero)
            const int64_t at = off + (tid < nb ? tid : 0);
            const int64_t r = prm->order ? prm->order[at] : at;
            rowidx[tid] = r;
            rowoff[tid] = r * prm->ldx;
        }
        __syncthreads();
        StepScalars sc;
        {
            const float* sq = scal + 8 * (step_i & 1);          // written one step ahead (see the loss phase)
            sc.lam = sq[0]; sc.two_lam = sq[1]; sc.lr_wd = sq[2]; sc.one_m_b1 = sq[3]; sc.one_m_b2 = sq[4];
            sc.step_size = sq[5]; sc.bc2_sqrt = sq[6];
        }
        const float drop_p = prm->dropout_p;
        const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        sc.s1 = keep_scale; sc.s2 = keep_scale;
        auto stamp = [&](int ph) { if (prm->dbg && g == 0 && tid == 0 && step_i < 16) prm->dbg[step_i * 16 + ph] = __builtin_readcyclecounter(); };
        stamp(0);

        // ================= P1: a1[:, own1] = drop(relu(X W1o^T + b1)) =================
        // wave w: rows w, w+8, w+16, w+24; lanes: 16-byte chunks of the D sum
        if (n1 > 0) {
            float acc[4][R1];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ii = 0; ii < R1; ++ii) acc[q][ii] = 0.f;
            const float* X = prm->X;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 4 * (lane + 64 * t);
                if (k < D) {
                    float4 x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const float4*>(X + rowoff[wave + 8 * q] + k);
#pragma unroll
                    for (int ii = 0; ii < R1; ++ii) {
                        const float4 w = *reinterpret_cast<const float4*>(Ps + oW1 + ii * D + k);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[q][ii] = fmaf(x[q].x, w.x, acc[q][ii]); acc[q][ii] = fmaf(x[q].y, w.y, acc[q][ii]);
                            acc[q][ii] = fmaf(x[q].z, w.z, acc[q][ii]); acc[q][ii] = fmaf(x[q].w, w.w, acc[q][ii]);
                        }
                    }
                }
            }
            // sums over the wave: two rounds of (2 rows x 4 slots); lanes 8v .. 8v+7 of round h hold (row q = 2h + (v >> 2), slot v & 3)
            float* scr = wscr + wave * 576;
            float mine[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float vals[8];
#pragma unroll
                for (int v = 0; v < 8; ++v) vals[v] = (v & 3) < R1 ? acc[2 * h + (v >> 2)][v & 3] : 0.f;
                mine[h] = wave_reduce_n<8>(vals, scr, lane);
            }
            if ((lane & 7) < 2) {
                const int h = lane & 7, v = lane >> 3, q = 2 * h + (v >> 2), ii = v & 3, b = wave + 8 * q;
                if (b < nb && ii < n1) {
                    float val = (h == 0 ? mine[0] : mine[1]) + Ps[oB1 + ii];
                    val = val < 0.f ? 0.f : val;
                    if (drop_p > 0.f)
                        val = ac::dropout_keep(prm->seed0 + (uint64_t)step_i, (uint64_t)((int64_t)b * H1 + i0 + ii), drop_p) ? val * sc.s1 : 0.f;
                    a1own[b * 4 + ii] = val;
                    st_sc1(prm->a1g + (size_t)b * H1 + i0 + ii, val);
                }
            }
        }
        stamp(1);
        if (!grid_barrier<kAcquireB1>(prm->ctl, ++bar, G, flag)) { bail(); return; }
        stamp(2);

        // ================= P2: a2[:, own2] = drop(relu(a1 W2r^T + b2)) =================
        if (n2 > 0) {
            const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(prm->a1g, (unsigned)(nb * H1 * sizeof(float)));      // rows past the batch were never written (workspace garbage, NaN x 0): bounded -> they read as zero
            float acc[4][R2];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int jj = 0; jj < R2; ++jj) acc[q][jj] = 0.f;
            float4 x[4][4];                                   // rows wave + 8q, columns 4 (lane + 64 t): all 16 loads in flight
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    x[t][q] = ld4_x<kAcquireB1>(ra1, 4 * (lane + 64 * t) < H1 ? (unsigned)(((wave + 8 * q) * H1 + 4 * (lane + 64 * t)) * 4) : 0xffffff00u);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 4 * (lane + 64 * t);
                if (k < H1) {
#pragma unroll
                    for (int jj = 0; jj < R2; ++jj) {
                        const float4 w = *reinterpret_cast<const float4*>(Ps + oW2r + jj * H1 + k);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[q][jj] = fmaf(x[t][q].x, w.x, acc[q][jj]); acc[q][jj] = fmaf(x[t][q].y, w.y, acc[q][jj]);
                            acc[q][jj] = fmaf(x[t][q].z, w.z, acc[q][jj]); acc[q][jj] = fmaf(x[t][q].w, w.w, acc[q][jj]);
                        }
                    }
                }
            }
            float vals[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) vals[v] = (v & 1) < R2 ? acc[v >> 1][v & 1] : 0.f;
            const float mine = wave_reduce_n<8>(vals, wscr + wave * 576, lane);
            if ((lane & 7) == 0) {
                const int v = lane >> 3, q = v >> 1, jj = v & 1, b = wave + 8 * q;
                if (b < nb && jj < n2) {
                    float val = mine + Ps[oB2 + jj];
                    val = val < 0.f ? 0.f : val;
                    if (drop_p > 0.f)
                        val = ac::dropout_keep((prm->seed0 + (uint64_t)step_i) ^ 0xA5A5A5A5A5A5A5A5ull, (uint64_t)((int64_t)b * H2 + j0 + jj), drop_p) ? val * sc.s2 : 0.f;
                    a2own[b * 2 + jj] = val;
                    st_sc1(prm->a2g + (size_t)b * H2 + j0 + jj, val);
                }
            }
        }
        {   // the output layer (final since the previous step's P5, visible since B1's acquire): staged ahead of the barrier
            const __amdgpu_buffer_rsrc_t rw3 = make_rsrc(prm->P + prm->o_w3, (unsigned)(C * H2 * sizeof(float)));
#pragma unroll 1
            for (int e4 = tid; e4 < C * H2 / 4; e4 += kT) *reinterpret_cast<float4*>(W3s + 4 * e4) = ld4_x<kAcquireB1>(rw3, (unsigned)(e4 * 16));
            if (tid < C) b3s[tid] = kAcquireB1 ? prm->P[prm->o_b3 + tid] : ld_sc1(prm->P + prm->o_b3 + tid);
        }
        stamp(3);
        if (!grid_barrier<false>(prm->ctl, ++bar, G, flag)) { bail(); return; }
        stamp(4);

        // ================= P3: logits, loss, dz, d2 (every workgroup, identically) =================
        {
            const __amdgpu_buffer_rsrc_t ra2 = make_rsrc(prm->a2g, (unsigned)(nb * H2 * sizeof(float)));
            constexpr int kStage = kMaxB / 4;                       // 16-byte pieces of a2 per thread (H2 <= kT): 8
            float4 va[kStage];
#pragma unroll
            for (int u = 0; u < kStage; ++u) va[u] = ld4_sc1(ra2, (unsigned)((tid + kT * u) * 16));      // (sc1: straight from the
            // memory side, faster here than an acquire + L2 misses; past the end: zeros)
#pragma unroll
            for (int u = 0; u < kStage; ++u) {
                const int e4 = tid + kT * u, b = (4 * e4) / H2, j = 4 * e4 - b * H2;
                if (b < kMaxB) *reinterpret_cast<float4*>(A2 + b * lda2 + j) = va[u];
            }
        }
        __syncthreads();
        stamp(5);
        // logits: wave w owns rows 4w .. 4w+3, lanes own 16-byte chunks of the H2 sum; four classes at a time
#pragma unroll 1
        for (int c0 = 0; c0 < C; c0 += 4) {
            float acc[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc[q][cc] = 0.f;
#pragma unroll 1
            for (int k = 4 * lane; k < H2; k += 256) {
                float4 w[4], x[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) w[cc] = *reinterpret_cast<const float4*>(W3s + (c0 + cc) * H2 + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const float4*>(A2 + (4 * wave + q) * lda2 + k);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        acc[q][cc] = fmaf(x[q].x, w[cc].x, acc[q][cc]); acc[q][cc] = fmaf(x[q].y, w[cc].y, acc[q][cc]);
                        acc[q][cc] = fmaf(x[q].z, w[cc].z, acc[q][cc]); acc[q][cc] = fmaf(x[q].w, w[cc].w, acc[q][cc]);
                    }
            }
            float* scr = wscr + wave * 576;
            float mine[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float vals[8];
#pragma unroll
                for (int v = 0; v < 8; ++v) vals[v] = acc[2 * h + (v >> 2)][v & 3];
                mine[h] = wave_reduce_n<8>(vals, scr, lane);
            }
            if ((lane & 7) < 2) {
                const int h = lane & 7, v = lane >> 3, q = 2 * h + (v >> 2), cc = v & 3, b = 4 * wave + q;
                if (b < nb && c0 + cc < C) zs[b * kMaxC + c0 + cc] = (h == 0 ? mine[0] : mine[1]) + b3s[c0 + cc];
            }
        }
        __syncthreads();
        stamp(6);
        if (tid < nb) {
            const int b = tid;
            const float* zr = zs + b * kMaxC;
            float* dzr = dzs + b * kMaxC;
            const int kind = prm->loss_kind;
            if (kind == AC_LOSS_BCE_SIGMOID) {
                const float inv = 1.f / ((float)nb * (float)C);
                const float* Trow = prm->T + rowidx[b] * prm->ldt;
                float sum = 0.f;
#pragma unroll 1
                for (int c = 0; c < C; ++c) {
                    const float p = 1.f / (1.f + expf(-zr[c]));
                    const float t = Trow[c];
                    sum -= t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f);
                    const float pq = p * (1.f - p);
                    dzr[c] = (p - t) / fmaxf(pq, 1e-12f) * inv * pq;
                }
                rl[b] = sum / (float)C;
            } else {
                const bool sig = kind == AC_LOSS_CE_SIGMOID;
                float v[KC];
                float mx = -INFINITY;
#pragma unroll
                for (int c = 0; c < KC; ++c) { v[c] = c < C ? zr[c] : -INFINITY; }
                if (sig) {
#pragma unroll 1
                    for (int c = 0; c < C; ++c) zs[b * kMaxC + c] = 1.f / (1.f + expf(-zr[c]));
#pragma unroll
                    for (int c = 0; c < KC; ++c) { v[c] = c < C ? zr[c] : -INFINITY; }
                }
#pragma unroll
                for (int c = 0; c < KC; ++c) mx = fmaxf(mx, v[c]);
                float ex[KC];
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < KC; ++c) { ex[c] = expf(v[c] - mx); sum += ex[c]; }        // (exp(-inf) = 0 for the slots past C)
                const int yb = (int)prm->y[rowidx[b]];
                const float invB = 1.f / (float)nb;
                float vy = 0.f;
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    float gz = (ex[c] / sum - (c == yb ? 1.f : 0.f)) * invB;
                    if (sig) gz *= v[c] * (1.f - v[c]);
                    if (c < C) dzr[c] = gz;
                    if (c == yb) vy = v[c];
                }
                rl[b] = (mx + logf(sum)) - vy;
            }
        } else if (tid < kMaxB) {
#pragma unroll
            for (int c = 0; c < kMaxC; ++c) dzs[tid * kMaxC + c] = 0.f;             // rows past the batch contribute nothing below
        } else if (tid == kT - 1) {
            write_scalars(step_i + 1);                                               // an idle wave: the next step's scalars
        }
        __syncthreads();
        stamp(7);
        float ce_loss = 0.f;
        if (wave == 0) {
            float sum = lane < nb ? rl[lane] : 0.f;
            sum = wave_sum(sum);
            ce_loss = sum / (float)nb;          // (used by workgroup 0, thread 0)
        }
        // thread j < H2: column j of a2 / W3 -> d2[:, j] (written over a2), the W2 duplicate's gradient
        // gW2[j, own1] = sum_b d2[b, j] a1[b, own1], and for the owner of column j: gb2[j], gW3[:, j]
        float gwc[4] = {0.f, 0.f, 0.f, 0.f}, gsm = 0.f;      // gsm: this thread's small-tensor gradient, if any
        int sm_e = -1;                                       // ... and its state slot
        if (tid < H2) {
            const int j = tid;
            float w3[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) w3[c] = W3s[c * H2 + j];
            const bool own_col = j >= j0 && j < j0 + n2;
            float gb = 0.f;
#pragma unroll 4
            for (int b = 0; b < kMaxB; ++b) {
                const float a = A2[b * lda2 + j];
                float t = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < KC; c4 += 4) {
                    const float4 dz4 = *reinterpret_cast<const float4*>(dzs + b * kMaxC + c4);
                    t = fmaf(dz4.x, w3[c4], t); t = fmaf(dz4.y, w3[c4 + 1], t); t = fmaf(dz4.z, w3[c4 + 2], t); t = fmaf(dz4.w, w3[c4 + 3], t);
                }
                const float d = (a != 0.f) ? t * sc.s2 : 0.f;                  // (dz is zero for rows past the batch)
                A2[b * lda2 + j] = d;
                gb += d;
                const float4 ao = *reinterpret_cast<const float4*>(a1own + b * 4);
                gwc[0] = fmaf(d, ao.x, gwc[0]); gwc[1] = fmaf(d, ao.y, gwc[1]); gwc[2] = fmaf(d, ao.z, gwc[2]); gwc[3] = fmaf(d, ao.w, gwc[3]);
            }
            if (own_col) { gsm = gb; sm_e = oB2 + (j - j0); }
        }
        __syncthreads();
        stamp(8);

        // ================= P4: the remaining gradients of the owned elements =================
        // d1[:, own1] = (d2 W2c) gated by a1own.  Thread (bp, q) = (tid >> 5, tid & 31): rows 2bp, 2bp+1, 16-byte chunks q, q+32,
        // ... of the j sum against all R1 columns (one W2c chunk serves two rows, one d2 chunk serves R1 columns: 3x less
        // LDS traffic than a thread per output); the 32 partials of a row pair meet through the wave's reduction tile.
        {
            const int bp = tid >> 5, q = tid & 31;
            float acc[2][R1];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int ii = 0; ii < R1; ++ii) acc[r][ii] = 0.f;
#pragma unroll 1
            for (int j = 4 * q; j < H2; j += 128) {
                const float4 d0 = *reinterpret_cast<const float4*>(A2 + (2 * bp) * lda2 + j);
                const float4 d1v = *reinterpret_cast<const float4*>(A2 + (2 * bp + 1) * lda2 + j);
#pragma unroll
                for (int ii = 0; ii < R1; ++ii) {
                    const float4 wv = *reinterpret_cast<const float4*>(Ps + oW2c + ii * H2 + j);
                    acc[0][ii] = fmaf(d0.x, wv.x, acc[0][ii]); acc[0][ii] = fmaf(d0.y, wv.y, acc[0][ii]);
                    acc[0][ii] = fmaf(d0.z, wv.z, acc[0][ii]); acc[0][ii] = fmaf(d0.w, wv.w, acc[0][ii]);
                    acc[1][ii] = fmaf(d1v.x, wv.x, acc[1][ii]); acc[1][ii] = fmaf(d1v.y, wv.y, acc[1][ii]);
                    acc[1][ii] = fmaf(d1v.z, wv.z, acc[1][ii]); acc[1][ii] = fmaf(d1v.w, wv.w, acc[1][ii]);
                }
            }
            // the wave holds two row pairs (lanes 0-31: pair 2 wave, lanes 32-63: pair 2 wave + 1): one round per pair
            float* scr = wscr + wave * 576;
            float mine[2];
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
                float vals[8];
                const bool in = (lane >> 5) == hp;
#pragma unroll
                for (int v = 0; v < 8; ++v) vals[v] = (in && (v & 3) < R1) ? acc[v >> 2][v & 3] : 0.f;
                mine[hp] = wave_reduce_n<8>(vals, scr, lane);
            }
            if ((lane & 7) < 2) {
                const int hp = lane & 7, v = lane >> 3, b = 2 * (2 * wave + hp) + (v >> 2), ii = v & 3;
                d1s[b * 4 + ii] = (a1own[b * 4 + ii] != 0.f) ? (hp == 0 ? mine[0] : mine[1]) * sc.s1 : 0.f;   // (a1own is zero past the batch / the owned rows)
            }
        }
        __syncthreads();
        stamp(9);
        // team A (waves 0-3): gW1[own1, 4t .. 4t+3] = d1[:, own1]^T X;  team B (waves 4-7): gW2[own2, 4t ..] = d2[:, own2]^T a1
        // -- fma chains ascending in b, sixteen 16-byte loads in flight.  (Requesting the operands ahead of d2 / d1 was tried:
        // slower -- every phase between request and use ran ~1.6x longer.)
        const bool teamA = wave < kT / 128;
        const int tcol = 4 * (teamA ? tid : tid - kT / 2);
        float4 gw[R1];                              // team B uses the first R2
#pragma unroll
        for (int ii = 0; ii < R1; ++ii) gw[ii] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool gw_active = teamA ? (n1 > 0 && tcol < D) : (n2 > 0 && tcol < H1);
        if (teamA) {                               // (wave-uniform branch: the two teams' registers overlap)
            if (gw_active) {
                const float* X = prm->X;
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    float4 x[kMaxB / 2];
#pragma unroll
                    for (int bb = 0; bb < kMaxB / 2; ++bb) x[bb] = *reinterpret_cast<const float4*>(X + rowoff[16 * h + bb] + tcol);
#pragma unroll
                    for (int bb = 0; bb < kMaxB / 2; ++bb) {
                        const float4 dv = *reinterpret_cast<const float4*>(d1s + (16 * h + bb) * 4);      // (zero for rows past the batch)
                        const float dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                        for (int ii = 0; ii < R1; ++ii) {
                            gw[ii].x = fmaf(dd[ii], x[bb].x, gw[ii].x); gw[ii].y = fmaf(dd[ii], x[bb].y, gw[ii].y);
                            gw[ii].z = fmaf(dd[ii], x[bb].z, gw[ii].z); gw[ii].w = fmaf(dd[ii], x[bb].w, gw[ii].w);
                        }
                    }
                }
            }
        } else {
            if (gw_active) {
                const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(prm->a1g, (unsigned)(nb * H1 * sizeof(float)));      // rows past the batch were never written (workspace garbage, NaN x 0): bounded -> they read as zero
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    float4 x[kMaxB / 2];
#pragma unroll
                    for (int bb = 0; bb < kMaxB / 2; ++bb) x[bb] = ld4_x<kAcquireB1>(ra1, (unsigned)(((16 * h + bb) * H1 + tcol) * 4));
#pragma unroll
                    for (int bb = 0; bb < kMaxB / 2; ++bb) {
#pragma unroll
                        for (int jj = 0; jj < R2; ++jj) {
                            const float d = jj < n2 ? A2[(16 * h + bb) * lda2 + j0 + jj] : 0.f;  // (zero for rows past the batch)
                            gw[jj].x = fmaf(d, x[bb].x, gw[jj].x); gw[jj].y = fmaf(d, x[bb].y, gw[jj].y);
                            gw[jj].z = fmaf(d, x[bb].z, gw[jj].z); gw[jj].w = fmaf(d, x[bb].w, gw[jj].w);
                        }
                    }
                }
            }
        }
        stamp(10);
        // small tensors.  gb2: the column owner above (thread j0 + jj).  gW3[:, own2]: threads kT-32 .., gb1: threads kT-64 .., gb3: threads kT-128 .. of workgroup 0 -- all outside [0, H2) (host: H2 <= kT - 128)
        if (tid >= kT - 2 * kMaxC) {
            const int q = tid - (kT - 2 * kMaxC), c = q >> 1, jj = q & 1;
            if (c < C && jj < n2) {                                      // gW3[c, j0 + jj] = sum_b dz[b, c] a2[b, j0 + jj]
#pragma unroll 8
                for (int b = 0; b < kMaxB; ++b) gsm = fmaf(dzs[b * kMaxC + c], a2own[b * 2 + jj], gsm);
                sm_e = oW3 + c * R2 + jj;
            }
        } else if (tid >= kT - 64 && tid < kT - 64 + R1) {
            const int ii = tid - (kT - 64);
            if (ii < n1) {
#pragma unroll 8
                for (int b = 0; b < kMaxB; ++b) gsm += d1s[b * 4 + ii];
                sm_e = oB1 + ii;
            }
        } else if (tid >= kT - 128 && tid < kT - 128 + C && g == 0) {
            const int c = tid - (kT - 128);
#pragma unroll 8
            for (int b = 0; b < kMaxB; ++b) gsm += dzs[b * kMaxC + c];
            sm_e = oB3 + c;
        }
        stamp(11);
        int64_t sm_gi = -1;
        if (sm_e >= 0) {
            if (sm_e >= oB3) sm_gi = prm->o_b3 + (sm_e - oB3);
            else if (sm_e >= oW3) { const int c = (sm_e - oW3) / R2, jj = (sm_e - oW3) - c * R2; sm_gi = prm->o_w3 + (int64_t)c * H2 + j0 + jj; }
            else if (sm_e >= oB2 && sm_e < oW2c) sm_gi = prm->o_b2 + j0 + (sm_e - oB2);
            else sm_gi = prm->o_b1 + i0 + (sm_e - oB1);
        }
        // every element this thread owns: f(state slot, global index, real parameter?, primary?, gradient register)
        const int nrow = teamA ? n1 : n2;
        const int e_row0 = teamA ? oW1 + tcol : oW2r + tcol;
        const int e_stride = teamA ? D : H1;
        const int64_t gi_row0 = teamA ? prm->o_w1 + (int64_t)i0 * D + tcol : prm->o_w2 + (int64_t)j0 * H1 + tcol;
        // rows of the big matrices as 16-byte vectors (4 consecutive columns per thread: float4 LDS / global accesses, no bank
        // conflicts -- scalar accesses at a 16-byte lane stride were 4-way conflicted, 23 % of the kernel's LDS cycles), the
        // W2 duplicate and the small tensors element by element
        auto for_rows = [&](auto&& f4) {
            if (gw_active) {
#pragma unroll
                for (int r = 0; r < R1; ++r)
                    if (r < (teamA ? R1 : R2)) f4(e_row0 + r * e_stride, gi_row0 + (int64_t)r * e_stride, r < nrow, gw[r]);
            }
        };
        auto for_scalars = [&](auto&& f) {
            if (tid < H2) {
#pragma unroll
                for (int ii = 0; ii < R1; ++ii) f(oW2c + ii * H2 + tid, prm->o_w2 + (int64_t)tid * H1 + i0 + ii, ii < n1, false, gwc[ii]);
            }
            if (sm_e >= 0) f(sm_e, sm_gi, true, true, gsm);
        };
        // raw gradients of the last step -> d_grads (what the step-by-step path leaves there)
        if (last_step && prm->Gout) {
            float* Gout = prm->Gout;
            for_rows([&](int, int64_t gi, bool real, float4& g4) { if (real) *reinterpret_cast<float4*>(Gout + gi) = g4; });
            for_scalars([&](int, int64_t gi, bool real, bool primary, float& gval) { if (real && primary) Gout[gi] = gval; });
        }
        // EWC term and the partial sums of |g_tot|^2 and F (p - p*)^2 over the PRIMARY elements
        float sg = 0.f, se = 0.f;
        if (prm->F != nullptr) {
            const unsigned nparam_bytes = (unsigned)((prm->o_b3 + C) * sizeof(float));
            const __amdgpu_buffer_rsrc_t rold = make_rsrc(prm->Old, nparam_bytes);
            const __amdgpu_buffer_rsrc_t rfis = make_rsrc(prm->F, nparam_bytes);
            for_rows([&](int e, int64_t gi, bool real, float4& g4) {
                const unsigned o = real ? (unsigned)gi * 4u : 0xffffff00u;            // (slots without a parameter read zeros)
                const float4 pv = *reinterpret_cast<const float4*>(Ps + e), ov = ld4_buf(rold, o), fv = ld4_buf(rfis, o);
                const float d0 = pv.x - ov.x, d1 = pv.y - ov.y, d2 = pv.z - ov.z, d3 = pv.w - ov.w;
                const float f0 = fv.x * d0, f1 = fv.y * d1, f2 = fv.z * d2, f3 = fv.w * d3;
                se = fmaf(f0, d0, se); se = fmaf(f1, d1, se); se = fmaf(f2, d2, se); se = fmaf(f3, d3, se);
                g4.x = fmaf(sc.two_lam, f0, g4.x); g4.y = fmaf(sc.two_lam, f1, g4.y);
                g4.z = fmaf(sc.two_lam, f2, g4.z); g4.w = fmaf(sc.two_lam, f3, g4.w);
            });
            for_scalars([&](int e, int64_t gi, bool real, bool primary, float& gval) {
                const unsigned o = real ? (unsigned)gi * 4u : 0xffffff00u;
                const float dlt = Ps[e] - __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rold, o, 0, 0));
                const float fd = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rfis, o, 0, 0)) * dlt;
                if (primary) se = fmaf(fd, dlt, se);
                gval = fmaf(sc.two_lam, fd, gval);
            });
        }
        for_rows([&](int, int64_t, bool, float4& g4) {
            sg = fmaf(g4.x, g4.x, sg); sg = fmaf(g4.y, g4.y, sg); sg = fmaf(g4.z, g4.z, sg); sg = fmaf(g4.w, g4.w, sg);
        });
        for_scalars([&](int, int64_t, bool, bool primary, float& gval) { if (primary) sg = fmaf(gval, gval, sg); });
        block_sum8x2(sg, se, red);
        if (tid == 0) { st_sc1(prm->partials + g, sg); st_sc1(prm->partials + kMaxG + g, se); }
        stamp(12);
        if (!grid_barrier<false>(prm->ctl, ++bar, G, flag)) { bail(); return; }
        stamp(13);

        // ================= P5: clip + AdamW on the owned elements =================
        float tg = tid < G ? ld_sc1(prm->partials + tid) : 0.f, te = tid < G ? ld_sc1(prm->partials + kMaxG + tid) : 0.f;
        block_sum8x2(tg, te, red);
        const float norm = sqrtf(tg);
        const float max_norm = prm->max_norm;
        float coef = max_norm / (norm + 1e-6f);
        if (coef > 1.f) coef = 1.f;
        if (max_norm <= 0.f) coef = 1.f;
        if (g == 0 && tid == 0) {
            float* out = prm->out;
            out[0] = ce_loss; out[1] = sc.lam * te; out[2] = norm;
            if (prm->loss_accum) *prm->loss_accum += ce_loss + sc.lam * te;
        }
        {
            const float beta2 = prm->beta2, eps = prm->eps;
            for_rows([&](int e, int64_t, bool, float4& g4) {
                float4 pv = *reinterpret_cast<const float4*>(Ps + e), mv = *reinterpret_cast<const float4*>(Ms + e),
                       vv = *reinterpret_cast<const float4*>(Vs + e);
                adamw_vals(pv.x, mv.x, vv.x, g4.x, coef, sc, beta2, eps); adamw_vals(pv.y, mv.y, vv.y, g4.y, coef, sc, beta2, eps);
                adamw_vals(pv.z, mv.z, vv.z, g4.z, coef, sc, beta2, eps); adamw_vals(pv.w, mv.w, vv.w, g4.w, coef, sc, beta2, eps);
                *reinterpret_cast<float4*>(Ps + e) = pv; *reinterpret_cast<float4*>(Ms + e) = mv; *reinterpret_cast<float4*>(Vs + e) = vv;
            });
            for_scalars([&](int e, int64_t, bool, bool, float& gval) { adamw_elem(Ps, Ms, Vs, e, gval, coef, sc, beta2, eps); });
        }
        if (sm_e >= oW3) st_sc1(prm->P + sm_gi, Ps[sm_e]);        // W3 / b3 travel: every workgroup reads them in P3
        stamp(14);
        __syncthreads();
    }
    // ---- write the owned state back ----
    {
        float* P = ka->P; float* M = ka->M; float* V = ka->V;
#pragma unroll 1
        for (int e = tid0; e < nstate; e += kT) {
            bool prim;
            const int64_t gi = gidx(e, prim);
            if (gi >= 0 && prim) {
                if (e < oW3) P[gi] = Ps[e];          // (W3 / b3 were already written, coherently, step by step)
                M[gi] = Ms[e]; V[gi] = Vs[e];
            }
        }
    }
}

size_t epoch_lds_floats(int D, int H1, int H2, int R1, int R2) {
    const size_t nstate = (size_t)R1 * D + 4 + (size_t)R2 * H1 + 4 + (size_t)R1 * H2 + kMaxC * R2 + kMaxC;
    return 3 * nstate + (size_t)kMaxB * (H2 + 4) + (size_t)kMaxC * H2 + kMaxB * 4 + kMaxB * 2 + kMaxB * 4 + 2 * kMaxB * kMaxC + kMaxB +
           kMaxC + 16 + kMaxC * 2 + 8 * 64 * 9 + 16;
}

size_t epoch_lds_bytes(int D, int H1, int H2, int R1, int R2) {
    return epoch_lds_floats(D, H1, H2, R1, R2) * sizeof(float) + 2 * kMaxB * sizeof(int64_t) + 16;
}

}  // namespace

namespace ac {

static std::atomic<long long> g_epoch_launches{0};
long long head_epoch_launches() { return g_epoch_launches.load(std::memory_order_relaxed); }

size_t head_epoch_ws_bytes(int H1, int H2) {
    return align_up((size_t)kMaxB * H1 * sizeof(float), 256) + align_up((size_t)kMaxB * H2 * sizeof(float), 256) +
           align_up(2 * kMaxG * sizeof(float), 256) + align_up(sizeof(acp::GridCtl), 256) + 16 * 16 * sizeof(unsigned long long);
}

// AC_OK: the epoch ran.  1: shape / alignment outside what the persistent kernel covers (caller falls back to the
// step-by-step launches).  Anything else: error.
int head_epoch_persistent(const ac_head_dims& d, float* P, float* M, float* V, float* Gout, const float* X, int64_t ldx,
                          const int64_t* y, const float* T, int64_t ldt, int loss_kind, const int64_t* order, int64_t n_total,
                          int batch, float dropout_p, uint64_t seed0, const float* F, const float* Old, float lambda_B,
                          float lam_direct, float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step0, float* out,
                          float* loss_accum, void* ws, hipStream_t stream) {
    if (!(persistent_mask() & 1) || n_total <= 0) return 1;
    const int G = dev_info().cus < kMaxG ? dev_info().cus : kMaxG;
    if (G < 8) return 1;
    const int r1 = (d.H1 + G - 1) / G, r2 = (d.H2 + G - 1) / G;
    if (batch > kMaxB || d.C > kMaxC || r1 > kR1 || r2 > kR2 || d.D > kT * kKU || d.H1 > kT * kKU || d.H2 > kT - 128) return 1;
    if ((d.D & 3) || (d.H1 & 3) || (d.H2 & 3) || (ldx & 3) || (((uintptr_t)X) & 15) || (((uintptr_t)P) & 15) ||
        (((uintptr_t)Gout) & 15) || (((uintptr_t)F) & 15) || (((uintptr_t)Old) & 15)) return 1;
    const int64_t o_w1 = 0, o_b1 = o_w1 + (int64_t)d.H1 * d.D, o_w2 = o_b1 + d.H1, o_b2 = o_w2 + (int64_t)d.H2 * d.H1, o_w3 = o_b2 + d.H2,
                  o_b3 = o_w3 + (int64_t)d.C * d.H2;
    if (o_w3 & 3) return 1;
    const bool r32 = r1 <= 3;                         // instantiations: (R1, R2) = (3, 2) -- the 768/768/384 head -- and (4, 2)
    const size_t lds = epoch_lds_bytes(d.D, d.H1, d.H2, r32 ? 3 : 4, 2);
    if (lds > (size_t)dev_info().lds_per_block) return 1;
    EpochParams p;
    p.D = d.D; p.H1 = d.H1; p.H2 = d.H2; p.C = d.C; p.G = G; p.r1 = r1; p.r2 = r2;
    p.P = P; p.M = M; p.V = V; p.Gout = Gout; p.F = F; p.Old = Old;
    p.X = X; p.ldx = ldx; p.y = y; p.T = T; p.ldt = ldt; p.order = order; p.n_total = n_total; p.batch = batch; p.loss_kind = loss_kind;
    p.dropout_p = dropout_p; p.seed0 = seed0; p.lambda_B = lambda_B; p.lam_direct = lam_direct; p.max_norm = max_norm; p.lr = lr; p.beta1 = beta1; p.beta2 = beta2;
    p.eps = eps; p.wd = wd; p.step0 = step0; p.out = out; p.loss_accum = loss_accum;
    char* w = (char*)ws;
    p.a1g = (float*)w; w += align_up((size_t)kMaxB * d.H1 * sizeof(float), 256);
    p.a2g = (float*)w; w += align_up((size_t)kMaxB * d.H2 * sizeof(float), 256);
    p.partials = (float*)w; w += align_up(2 * kMaxG * sizeof(float), 256);
    p.ctl = (acp::GridCtl*)w; w += align_up(sizeof(acp::GridCtl), 256);
    static const int debug = [] { const char* e = getenv("AC_HEAD_EPOCH_DEBUG"); return e ? atoi(e) : 0; }();
    p.dbg = debug ? (unsigned long long*)w : nullptr;
    if (debug) AC_HIP_CHECK(hipMemsetAsync(p.dbg, 0, 16 * 16 * sizeof(unsigned long long), stream));
    p.o_w1 = o_w1; p.o_b1 = o_b1; p.o_w2 = o_w2; p.o_b2 = o_b2; p.o_w3 = o_w3; p.o_b3 = o_b3;
    AC_HIP_CHECK(hipMemsetAsync(p.ctl, 0, sizeof(acp::GridCtl), stream));
    const int variant = (d.C <= 4 ? 0 : 1) + (r32 ? 0 : 2);
    const void* fns[4] = {(const void*)head_epoch_kernel<4, 3, 2>, (const void*)head_epoch_kernel<kMaxC, 3, 2>,
                          (const void*)head_epoch_kernel<4, 4, 2>, (const void*)head_epoch_kernel<kMaxC, 4, 2>};
    const void* fn = fns[variant];
    static size_t lds_set[4] = {0, 0, 0, 0};
    if (lds > lds_set[variant]) {
        AC_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set[variant] = lds;
    }
    // residency proof at launch: workgroups per CU (occupancy query) x CUs this process reaches (dev_info().cus is MEASURED, a CU
    // mask counts) must hold the grid -- otherwise the step-by-step launches run, chosen up front.  hipLaunchCooperativeKernel
    // repeats the check against the chip's CU count.
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kT, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    if ((int64_t)per_cu * dev_info().cus < G) return 1;
    void* args[] = {&p};
    AC_HIP_CHECK(hipLaunchCooperativeKernel(fn, dim3(G), dim3(kT), args, (unsigned)lds, stream));
    g_epoch_launches.fetch_add(1, std::memory_order_relaxed);
    if (debug) {        // phase timings of workgroup 0 (shader cycles), steps 1..15 averaged
        unsigned long long h[16 * 16];
        AC_HIP_CHECK(hipStreamSynchronize(stream));
        AC_HIP_CHECK(hipMemcpy(h, p.dbg, sizeof(h), hipMemcpyDeviceToHost));
        const int nsteps = (int)((n_total + batch - 1) / batch), ns = nsteps < 16 ? nsteps : 16;
        if (ns >= 3) {
            static const char* names[15] = {"P1", "B1", "P2", "B2", "stage", "z", "loss", "d2", "d1", "gw", "gwc+small", "ewc+norm", "B3", "P5", "next"};
            fprintf(stderr, "head_epoch (cycles, wg 0):");
            for (int q = 0; q < 15; ++q) {
                double acc = 0;
                for (int st = 1; st < ns - 1; ++st) acc += q < 14 ? (double)(h[st * 16 + q + 1] - h[st * 16 + q]) : (double)(h[(st + 1) * 16] - h[st * 16 + 14]);
                fprintf(stderr, " %s %.0f", names[q], acc / (ns - 2));
            }
            fprintf(stderr, " | LDS %zu B\n", lds);
        }
    }
    return AC_OK;
}

}  // namespace ac
