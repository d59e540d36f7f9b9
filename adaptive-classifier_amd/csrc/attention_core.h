// The fused fp32 attention of one (sequence, head, 32-query tile) by ONE wave -- shared by attention_mfma_kernel (bert.hip: operands
// from the fp32 qkv rows in HBM) and the attention epilogue of the QKV GEMM (gemm_pipe.hip, EPI_QKV_ATTN: operands from the GEMM's
// own output tile staged in LDS).  Same instructions in the same order either way, so the two routes give bit-identical context rows.
// transformers/models/bert/modeling_bert.py:111-203 (BertSelfAttention, eager); modeling_modernbert.py:188-219 for ROPE.
#pragma once
#include "common.h"

#include <math.h>

namespace acattn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- fused attention on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32), head dim 64 (or 32) ----
//   S^T tile = K_tile . Q^T  : A = K rows (lane (key j, k-slice h) holds float4 K[j][8kb+4h..]),
//                              B = Q rows (same shape, pre-scaled, resident in registers)
//                              -> C layout: lane & 31 = QUERY, registers = the tile's 32 keys
//   so the online softmax (max / sum over keys) is per-lane register work plus ONE exchange with the
//   partner lane (lane ^ 32) -- no LDS, no row reductions across the wave.
//   O^T += V_tile^T . P^T    : B = P, which is ALREADY in B-operand layout (lane = query, k = lane >> 5:
//                              MFMA step r consumes keys row(r,0) and row(r,1)); A = V^T read as
//                              V[key][32t + (lane & 31)] (128-B coalesced rows).  C layout again has
//                              lane & 31 = query, so the rescale by exp(m_old - m_new) is per lane.
// The k-order inside a tile is whatever the C layout dictates -- a dot product does not care.
__device__ __forceinline__ int crow32(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void rope_rotate(f32x4 (&x)[8], const float* cs, const float* sn, int pos, int h) {
#pragma clang fp contract(off)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(cs + (int64_t)pos * 32 + 8 * kb + 4 * h);
        const f32x4 s = *reinterpret_cast<const f32x4*>(sn + (int64_t)pos * 32 + 8 * kb + 4 * h);
        const f32x4 lo = x[kb], hi = x[kb + 4];
        x[kb] = lo * c + (-hi) * s;          // q * cos + rotate_half(q) * sin, two roundings per term like torch
        x[kb + 4] = hi * c + lo * s;
    }
}

// qb / kb / vb: row 0 of the sequence at this head's Q / K / V columns, `ld` floats between consecutive token rows; S = the
// sequence's length, qt = which 32-query tile.  mask_row (optional): the sequence's key mask (int64, 0 = masked).
// Output: ctx_rows (fp32, row 0 of the sequence at this head's columns, H floats between rows) or, when ctx_planes is given, the
// operand planes of the [rows, H] output-projection input (row0 = the sequence's first row there, headcol = head * DHT).
// ROPE (ModernBERT): q, k rotated by the position's angle before the scores,
//   x'[d] = x[d] cos[d] - x[d+32] sin[d],  x'[d+32] = x[d+32] cos[d] + x[d] sin[d]   (d < 32; cos/sin tables
//   [position][32] computed on the host exactly as transformers does).  Both halves of a pair sit in the same
//   lane (fragment k-blocks kb and kb + 4), so the rotation is register-local.
// window >= 0 (sliding-window layers, masking_utils.py:141-151): key k is visible to query q iff |q - k| <= window.
// FP32_OUT = false: the caller always passes ctx_planes (the fp32-row store path is not compiled: in a kernel whose operands sit
// in LDS its generic destination pointer would become flat stores).
template <bool ROPE, int DHT, bool FP32_OUT = true>
__device__ __forceinline__ void attention_tile(const float* qb, const float* kb_, const float* vb, int64_t ld, int S, int qt, int lane,
                                               float scale, const int64_t* mask_row, const float* rope_cos, const float* rope_sin,
                                               int window, float* ctx_rows, int H, uint16_t* ctx_planes, int64_t rows, int64_t row0,
                                               int headcol, int f16) {
    // No implicit mul + add fusion in here: whether hipcc contracts `l * corr + psum` or packs two of them into v_pk_fma_f32
    // depends on the surrounding kernel, and the two routes must round alike (the one intended fma is written out).
#pragma clang fp contract(off)
    static_assert(DHT == 64 || (DHT == 32 && !ROPE), "head dim 64, or 32 without RoPE");
    constexpr int NKB = DHT / 8;                                     // 8-dim k-blocks of the QK^T product
    const int j = lane & 31, h = lane >> 5;
    const int qi = qt * 32 + j;
    const bool qvalid = qi < S;

    f32x4 Qf[NKB];
    {
        const float* qp = qb + (int64_t)(qvalid ? qi : S - 1) * ld + 4 * h;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) Qf[kb] = *reinterpret_cast<const f32x4*>(qp + 8 * kb);
        if constexpr (ROPE) rope_rotate(Qf, rope_cos, rope_sin, qvalid ? qi : S - 1, h);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) Qf[kb] = Qf[kb] * scale;
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -INFINITY, l = 0.f;

    // K fragment of a tile (keys past S are clamped; they are masked below).  The next tile's fragment is
    // requested right after the QK^T MFMAs have consumed this one, so its latency hides under softmax + PV.
    f32x4 Kf[NKB];
    auto load_k = [&](int k0) {
        int kr = k0 + j; if (kr > S - 1) kr = S - 1;
        const float* kp = kb_ + (int64_t)kr * ld + 4 * h;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) Kf[kb] = *reinterpret_cast<const f32x4*>(kp + 8 * kb);
        if constexpr (ROPE) rope_rotate(Kf, rope_cos, rope_sin, kr, h);
    };
    // key tiles that can hold a visible key for any of this tile's 32 queries (wave-uniform bounds)
    int kbeg = 0, kend = S;
    if (window >= 0) {
        kbeg = qt * 32 - window; kbeg = kbeg < 0 ? 0 : (kbeg / 32) * 32;
        const int last = qt * 32 + 31 + window;
        if (last + 1 < kend) kend = last + 1;
    }
    load_k(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        // key validity as a 32-bit mask shared by the wave
        const int kj = k0 + j;
        const bool kv = kj < S && (!mask_row || mask_row[kj] != 0);
        const unsigned vmask = (unsigned)(__ballot(kv && h == 0) & 0xffffffffull);
        if (vmask == 0u) { load_k(k0 + 32); continue; }              // wave-uniform
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(Kf[kb][s4], Qf[kb][s4], st, 0, 0, 0);
        load_k(k0 + 32);                                             // clamped past the end: harmless re-read
        // online softmax for query `j` (this lane + partner lane hold its 32 keys)
        float cmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            bool ok = (vmask >> crow32(r, h)) & 1u;
            if (window >= 0) { const int dk = qi - (k0 + crow32(r, h)); ok = ok && dk <= window && -dk <= window; }
            st[r] = ok ? st[r] : -INFINITY;
            cmax = fmaxf(cmax, st[r]);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float m_new = fmaxf(m, cmax);
        // with a window a tile may hold no visible key for THIS query: nothing accumulated yet -> keep zeros
        const bool none = m_new == -INFINITY;
        // e^x as 2^(x log2 e) on v_exp_f32 (one multiply + one transcendental instead of ocml's ~25-instruction expf, 17 times per
        // key tile and lane): x <= 0 here, the result's relative error is ~2^-22 (1 + |x|) -- far inside the softmax's own rounding
        constexpr float kLog2e = 1.4426950408889634f;
        const float corr = none ? 1.f : __builtin_amdgcn_exp2f((m - m_new) * kLog2e);            // m = -inf -> 0
        float psum = 0.f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = none ? 0.f : __builtin_amdgcn_exp2f((st[r] - m_new) * kLog2e); psum += p[r]; }
        psum += __shfl_xor(psum, 32);
        l = __builtin_fmaf(l, corr, psum);
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
        // O^T += V^T P^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int vr = k0 + crow32(r, h); if (vr > S - 1) vr = S - 1;
            const float* vp = vb + (int64_t)vr * ld + j;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], p[r], o0, 0, 0, 0);
            if constexpr (DHT == 64) o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], p[r], o1, 0, 0, 0);
        }
    }
    if (qvalid) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        // C layout of O^T: lane & 31 = query (this lane), register r = output dim crow32(r, h) (+32 for o1)
        if (!FP32_OUT || ctx_planes) {
            // registers 4g .. 4g+3 are dims 8g + 4h .. +3: half of k-slot (head * 8 + 4t + g); the partner lane
            // (h ^ 1) writes the other half
            const int64_t row = row0 + qi, plane = rows * H;
#pragma unroll
            for (int t = 0; t < DHT / 32; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (t ? o1[4 * g + e] : o0[4 * g + e]) * inv;
                    ac::emit_planes4(ctx_planes + ac::plane_off(rows, row, headcol + 32 * t + 8 * g + 4 * h), plane, v, f16);
                }
        } else if constexpr (FP32_OUT) {
            float* dst = ctx_rows + (int64_t)qi * H;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dst[crow32(r, h)] = o0[r] * inv;
                if constexpr (DHT == 64) dst[32 + crow32(r, h)] = o1[r] * inv;
            }
        }
    }
}

}  // namespace acattn
