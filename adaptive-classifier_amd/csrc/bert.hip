// BERT-family encoder forward -> unit-norm CLS embedding, the "E" site of the hot path:
//   classifier.py:1271  outputs = self.model(**inputs)            (transformers BertModel, eval)
//   classifier.py:1272  outputs.last_hidden_state[:, 0, :]
//   classifier.py:1275  F.normalize(embeddings, p=2, dim=1)
// Arithmetic follows transformers/models/bert/modeling_bert.py (v5.15.0): embeddings :53-108,
// self-attention :111-203, self-output :282-293, intermediate/output :325-351.
//
// All dense projections run on the fp32 MFMA pipe (gemm.hip) with fused epilogues
// (bias | bias+GELU(erf) | bias+residual).  This file holds the memory-bound glue as single-pass
// wave-per-token kernels (embedding gather+LayerNorm, LayerNorm, CLS gather+L2 normalise) and a
// fused fp32 attention kernel (one wave per (sequence, head, 64-query tile); K/V tiles staged
// through LDS, online softmax, scores never touch HBM).
#include "common.h"
#include "attention_core.h"

#include <math.h>
#include <atomic>
#include <chrono>
#include <string.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxVec = 8;    // float4 per lane held in registers: H <= 64 * 4 * 8 = 2048

// ---- wave-per-token LayerNorm helpers ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// dst: the fp32 row; planes (optional): the same row as bf16x3 operand planes of a [rows, H] matrix
template <bool F16 = false>      // (compile time: as a run-time flag the two plane forms cost embed_ln_kernel 18 -> 23 us)
__device__ __forceinline__ void layernorm_store(f32x4 (&x)[kMaxVec], int H, int lane, const float* g,
                                                const float* b, float eps, float* dst, uint16_t* planes = nullptr,
                                                int64_t rows = 0, int64_t row = 0) {
    const int nv = H >> 2;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e)
        if (lane + 64 * e < nv) s += (x[e].x + x[e].y) + (x[e].z + x[e].w);
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e)
        if (lane + 64 * e < nv) {
            const float d0 = x[e].x - mean, d1 = x[e].y - mean, d2 = x[e].z - mean, d3 = x[e].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    const float var = wave_sum(q) / (float)H;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e) {
        const int c4 = lane + 64 * e;
        if (c4 < nv) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * c4);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 4 * c4);
            f32x4 y;
            y.x = (x[e].x - mean) * rstd * gg.x + bb.x;
            y.y = (x[e].y - mean) * rstd * gg.y + bb.y;
            y.z = (x[e].z - mean) * rstd * gg.z + bb.z;
            y.w = (x[e].w - mean) * rstd * gg.w + bb.w;
            *reinterpret_cast<f32x4*>(dst + 4 * c4) = y;
            if (planes)       // lanes 2j, 2j+1 fill the two halves of k-slot j
                ac::emit_planes4(planes + ac::plane_off(rows, row, 4 * c4), rows * (int64_t)H, y, F16);
        }
    }
}

// word + position + token_type embeddings -> LayerNorm (modeling_bert.py:85-107)
// tok_src (packed / padding-free mode): row t of the output is token tok_src[t] = seq * S + pos of the [b, S] inputs
template <bool F16>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* ids, const int64_t* type_ids, int T, int S,
                                                       int H, const float* word, const float* pos,
                                                       const float* type, const float* g, const float* b,
                                                       float eps, float* out, uint16_t* planes,
                                                       const int32_t* __restrict__ tok_src = nullptr,
                                                       const int32_t* __restrict__ t_dev = nullptr) {
    // t_dev: the token-row count is still on its way to the host (ac_bert_encode_cls_unpad): the launch covers b * S rows and
    // the real count -- the row stride of the operand planes too -- is read here
    if (t_dev) T = t_dev[0];
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int src = tok_src ? tok_src[t] : t;
    const int64_t id = ids[src];
    const int64_t tt = type_ids ? type_ids[src] : 0;
    const int p = src % S;
    const int nv = H >> 2;
    f32x4 x[kMaxVec];
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e) {
        const int c4 = lane + 64 * e;
        if (c4 < nv) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(word + id * H + 4 * c4);
            if (type) {               // BERT: inputs_embeds + token_type_embeddings, then + position
                const f32x4 ty = *reinterpret_cast<const f32x4*>(type + tt * H + 4 * c4);
                const f32x4 po = *reinterpret_cast<const f32x4*>(pos + (int64_t)p * H + 4 * c4);
                x[e] = (w + ty) + po;
            } else {
                x[e] = w;             // ModernBERT: token embeddings only (positions enter through RoPE)
            }
        }
    }
    layernorm_store<F16>(x, H, lane, g, b, eps, out + (int64_t)t * H, planes, T, t);
}

template <bool F16>
__global__ __launch_bounds__(256) void ln_kernel_t(const float* in, int T, int H, const float* g, const float* b,
                                                   float eps, float* out, uint16_t* planes, int64_t in_stride) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int nv = H >> 2;
    f32x4 x[kMaxVec];
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e) {
        const int c4 = lane + 64 * e;
        if (c4 < nv) x[e] = *reinterpret_cast<const f32x4*>(in + (int64_t)t * in_stride + 4 * c4);
    }
    layernorm_store<F16>(x, H, lane, g, b, eps, out + (int64_t)t * H, planes, T, t);
}

inline void launch_ln(bool f16, int blocks, hipStream_t stream, const float* in, int T, int H, const float* g, const float* b,
                      float eps, float* out, uint16_t* planes, int64_t in_stride) {
    if (f16) hipLaunchKernelGGL(ln_kernel_t<true>, dim3(blocks), dim3(256), 0, stream, in, T, H, g, b, eps, out, planes, in_stride);
    else hipLaunchKernelGGL(ln_kernel_t<false>, dim3(blocks), dim3(256), 0, stream, in, T, H, g, b, eps, out, planes, in_stride);
}

// The end of the CLS-only last layer in ONE launch (instead of gemm_splitk_reduce -> ln_kernel_t -> cls_normalize_kernel): the K
// slices of the FFN-down product summed in order + bias + residual, LayerNorm over the row, F.normalize of the result.  One wave
// per CLS row; same expressions as the three kernels (the norm's sum runs over this kernel's lane layout: equal to fp32 rounding).
__global__ __launch_bounds__(256) void splitk_ln_normalize_kernel(const float* __restrict__ part, int ksplit, int M, int H,
                                                                  const float* __restrict__ bias, const float* __restrict__ residual,
                                                                  int64_t ldr, const float* __restrict__ g, const float* __restrict__ be,
                                                                  float eps, float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= M) return;
    const int nv = H >> 2;
    const size_t slice = (size_t)M * H;
    f32x4 x[kMaxVec];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e) x[e] = zero4;
    // the slices in order, four at a time: the loads of a group (x kMaxVec vectors) are in flight together -- one slice per
    // round trip made this kernel 21 us for 256 rows
    for (int z0 = 0; z0 < ksplit; z0 += 4) {
        f32x4 t[4][kMaxVec];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < kMaxVec; ++e) {
                const int c4 = lane + 64 * e;
                t[u][e] = (z0 + u < ksplit && c4 < nv) ? *reinterpret_cast<const f32x4*>(part + (size_t)(z0 + u) * slice + (size_t)i * H + 4 * c4) : zero4;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (z0 + u < ksplit) {
#pragma unroll
                for (int e = 0; e < kMaxVec; ++e)
                    if (z0 + u == 0) x[e] = t[u][e];
                    else { x[e].x += t[u][e].x; x[e].y += t[u][e].y; x[e].z += t[u][e].z; x[e].w += t[u][e].w; }
            }
    }
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e) {
        const int c4 = lane + 64 * e;
        if (c4 < nv) {
            const f32x4 s = x[e];
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + 4 * c4);
            const f32x4 rr = *reinterpret_cast<const f32x4*>(residual + (int64_t)i * ldr + 4 * c4);
            x[e].x = (s.x + bb.x) + rr.x; x[e].y = (s.y + bb.y) + rr.y; x[e].z = (s.z + bb.z) + rr.z; x[e].w = (s.w + bb.w) + rr.w;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e)
        if (lane + 64 * e < nv) s += (x[e].x + x[e].y) + (x[e].z + x[e].w);
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e)
        if (lane + 64 * e < nv) {
            const float d0 = x[e].x - mean, d1 = x[e].y - mean, d2 = x[e].z - mean, d3 = x[e].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    const float var = wave_sum(q) / (float)H;
    const float rstd = 1.0f / sqrtf(var + eps);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e) {
        const int c4 = lane + 64 * e;
        if (c4 < nv) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * c4);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(be + 4 * c4);
            f32x4 y;
            y.x = (x[e].x - mean) * rstd * gg.x + bb.x;
            y.y = (x[e].y - mean) * rstd * gg.y + bb.y;
            y.z = (x[e].z - mean) * rstd * gg.z + bb.z;
            y.w = (x[e].w - mean) * rstd * gg.w + bb.w;
            x[e] = y;
            ss = fmaf(y.x, y.x, ss); ss = fmaf(y.y, y.y, ss); ss = fmaf(y.z, y.z, ss); ss = fmaf(y.w, y.w, ss);
        }
    }
    const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e) {
        const int c4 = lane + 64 * e;
        if (c4 < nv) {
            f32x4 y = x[e];
            y.x = y.x / nrm; y.y = y.y / nrm; y.z = y.z / nrm; y.w = y.w / nrm;
            float* dst = out + (int64_t)i * ldo + 4 * c4;
            dst[0] = y.x; dst[1] = y.y; dst[2] = y.z; dst[3] = y.w;                   // (ldo need not be a multiple of 4)
        }
    }
    for (int c = H + lane; c < ldo; c += 64) out[(int64_t)i * ldo + c] = 0.f;
}

// last_hidden_state[:, 0, :] -> F.normalize(p=2, dim=1, eps=1e-12)
__global__ __launch_bounds__(256) void cls_normalize_kernel(const float* x, int b, int S, int H, float* out,
                                                            int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= b) return;
    const float* src = x + (int64_t)i * S * H;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s = fmaf(src[c], src[c], s);
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = lane; c < H; c += 64) out[(int64_t)i * ldo + c] = src[c] / nrm;
    for (int c = H + lane; c < ldo; c += 64) out[(int64_t)i * ldo + c] = 0.f;
}

constexpr int DH = 64;        // head dim (checked in ac_bert_encode_cls)
constexpr int KT = 64;        // keys per LDS tile of the CLS-only kernel

using acattn::f32x16;
using acattn::rope_rotate;

// One wave per (sequence, head, 32-query tile); grid = (ceil(S/32), heads, batch); the arithmetic is acattn::attention_tile
// (attention_core.h).  ctx_planes (optional): emit the context as bf16x3 operand planes of the [batch*S, H] output-projection
// input instead of fp32 rows.
// cu (packed / padding-free mode): sequence bi owns rows [cu[bi], cu[bi+1]) of qkv / ctx, all of them real tokens
// (no mask); total_rows = cu[batch] is the row count of the planes output.  cu == nullptr: rows bi*S .. bi*S+S-1.
// DHT = head dimension: 64 (BERT-base / large, DistilBERT, RoBERTa, ModernBERT) or 32 (the MiniLM family: 384 = 12 x 32).
// boundary_stride > 0 (packed mode, after the QKV GEMM with the fused attention epilogue, gemm_pipe.hip EPI_QKV_ATTN): grid z
// walks the row-tile boundaries boundary_stride * (z + 1) instead of the sequences; the wave serves the sequence that STRADDLES
// its boundary (whose rows two GEMM tiles share, so neither could finish it) and leaves when a sequence starts exactly there.
template <bool ROPE, int DHT = 64>
__global__ __launch_bounds__(64) void attention_mfma_kernel(const float* qkv, const int64_t* mask, int S_, int H,
                                                            float scale, float* ctx, uint16_t* ctx_planes,
                                                            const float* rope_cos, const float* rope_sin,
                                                            int window, const int32_t* __restrict__ cu = nullptr,
                                                            int64_t total_rows = 0, int f16 = 0, int boundary_stride = 0,
                                                            int nseq = 0) {
    const int lane = threadIdx.x;
    const int qt = blockIdx.x, head = blockIdx.y;
    int bi = blockIdx.z;
    if (boundary_stride > 0) {
        const int brow = boundary_stride * ((int)blockIdx.z + 1);
        int lo = 0, hi = nseq;                                       // the sequence holding row `brow`: largest s with cu[s] <= brow
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cu[mid] <= brow) lo = mid; else hi = mid; }
        if (cu[lo] == brow || brow >= cu[nseq]) return;              // a sequence starts at the boundary: nothing straddles it
        bi = lo;
    }
    const int64_t ld = 3 * (int64_t)H;
    const int64_t row0 = cu ? (int64_t)cu[bi] : (int64_t)bi * S_;
    const int S = cu ? cu[bi + 1] - cu[bi] : S_;
    if (qt * 32 >= S) return;                                        // (packed mode: short sequence)
    const float* base = qkv + row0 * ld + head * DHT;
    const int64_t rows = cu ? total_rows : (int64_t)gridDim.z * S_;
    acattn::attention_tile<ROPE, DHT>(base, base + H, base + 2 * H, ld, S, qt, lane, scale, mask ? mask + (int64_t)bi * S_ : nullptr,
                                      rope_cos, rope_sin, window, ctx ? ctx + row0 * H + head * DHT : nullptr, H, ctx_planes, rows,
                                      row0, head * DHT, f16);
}

// Last layer: only the CLS query of every sequence feeds the output (classifier.py:1272), so its
// attention is a single-query problem.  One wave per (sequence, head): lane = key for the scores,
// lane = output dim for the P.V reduction; K tile rows padded to 65 floats (conflict-free column reads).
// ROPE (ModernBERT): the CLS query is at position 0, whose rotation is the identity, so only the keys rotate;
// window >= 0: only keys at positions <= window are visible to it.
template <bool ROPE, int DHT = 64, int KTT = KT>     // KTT: keys per LDS tile (32 when no sequence is longer: half the LDS, twice the waves per CU)
__global__ __launch_bounds__(64) void attention_cls_kernel(const float* qkv, const int64_t* mask, int S_, int H,
                                                           float scale, float* ctx_cls, const float* rope_cos,
                                                           const float* rope_sin, int window,
                                                           const int32_t* __restrict__ cu = nullptr,
                                                           const float* __restrict__ q_cls = nullptr) {
    // q_cls: the b CLS queries as compact [b, H] rows (BERT's last layer projects Q for those rows only); else column block 0 of qkv
    static_assert(DHT == 64 || (DHT == 32 && !ROPE), "head dim 64, or 32 without RoPE");
    __shared__ float Ks[KTT][DHT + 1];
    __shared__ __attribute__((aligned(16))) float Vs[KTT][DHT];
    __shared__ float qs[DHT];
    __shared__ float ps[KTT];
    const int lane = threadIdx.x;
    const int head = blockIdx.x, bi = blockIdx.y;
    const int64_t ld = 3 * (int64_t)H;
    const int64_t row0 = cu ? (int64_t)cu[bi] : (int64_t)bi * S_;
    const int S = cu ? cu[bi + 1] - cu[bi] : S_;
    const float* base = qkv + row0 * ld + head * DHT;
    if (lane < DHT) qs[lane] = (q_cls ? q_cls[(int64_t)bi * H + head * DHT + lane] : base[lane]) * scale;     // CLS token = row 0 of the sequence
    float m = -INFINITY, l = 0.f, o = 0.f;      // o: output dim `lane`
    const int Svis = (window >= 0 && window + 1 < S) ? window + 1 : S;     // keys the CLS query can see
    for (int k0 = 0; k0 < Svis; k0 += KTT) {
        const int nk = (Svis - k0) < KTT ? (Svis - k0) : KTT;
        const int kl = lane < KTT ? lane : 0;          // (lanes beyond the tile: a valid row, their score is masked)
        __syncthreads();
        constexpr int LPR = DHT / 4;                     // lanes per key row (a float4 each)
        for (int r = lane / LPR; r < KTT; r += 64 / LPR) {
            const int c = (lane % LPR) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (r < nk) {
                const float* kp = base + (int64_t)(k0 + r) * ld + H;
                kv = *reinterpret_cast<const f32x4*>(kp + c);
                vv = *reinterpret_cast<const f32x4*>(kp + H + c);
            }
            Ks[r][c] = kv.x; Ks[r][c + 1] = kv.y; Ks[r][c + 2] = kv.z; Ks[r][c + 3] = kv.w;
            *reinterpret_cast<f32x4*>(&Vs[r][c]) = vv;
        }
        __syncthreads();
        const bool valid = lane < nk && (!mask || mask[(int64_t)bi * S_ + k0 + lane] != 0);
        float sc = 0.f;
        if constexpr (ROPE) {
            const int pos = (k0 + lane < S) ? k0 + lane : S - 1;
            const float* cs = rope_cos + (int64_t)pos * 32;
            const float* sn = rope_sin + (int64_t)pos * 32;
#pragma unroll 8
            for (int d = 0; d < 32; ++d) {
                const float lo = Ks[kl][d], hi = Ks[kl][d + 32], c = cs[d], sv = sn[d];
                sc = fmaf(qs[d], lo * c + (-hi) * sv, sc);
                sc = fmaf(qs[d + 32], hi * c + lo * sv, sc);
            }
        } else {
#pragma unroll 16
            for (int d = 0; d < DHT; ++d) sc = fmaf(qs[d], Ks[kl][d], sc);
        }
        sc = valid ? sc : -INFINITY;
        float cmax = sc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, off));
        const float m_new = fmaxf(m, cmax);
        if (m_new == -INFINITY) continue;        // wave-uniform
        const float corr = expf(m - m_new);
        const float p = expf(sc - m_new);        // masked -> 0
        if (lane < KTT) ps[lane] = p;
        float psum = p;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) psum += __shfl_xor(psum, off);
        l = l * corr + psum;
        o *= corr;
        __syncthreads();
        if (lane < DHT)
            for (int j = 0; j < nk; ++j) o = fmaf(ps[j], Vs[j][lane], o);
        m = m_new;
    }
    if (lane < DHT) ctx_cls[(int64_t)bi * H + head * DHT + lane] = l > 0.f ? o / l : 0.f;
}

// ModernBERT MLP gate (modeling_modernbert.py:89-91): u = Wi x is [T, 2I]; g = gelu(u[:, :I]) * u[:, I:]
// (erf GELU), written as fp32 rows and, optionally, as the operand planes of the following Wo GEMM.
// interleaved32: the columns of u come in blocks of 64 = 32 inputs then their 32 gates (the weight rows were
// permuted at load for the fused GEMM epilogue EPI_GEGLU32); otherwise all inputs then all gates.
__global__ __launch_bounds__(256) void geglu_kernel(const float* __restrict__ u, int64_t T, int I, float* __restrict__ g,
                                                    uint16_t* __restrict__ planes, int interleaved32) {
    const int nq = I >> 3;                                     // 8 outputs per thread = one k-slot
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * nq) return;
    const int64_t row = idx / nq;
    const int q = (int)(idx - row * nq);
    const int c = 8 * q;                                       // first output column
    const float* a = u + row * 2 * (int64_t)I + (interleaved32 ? (c >> 5) * 64 + (c & 31) : c);
    const int goff = interleaved32 ? 32 : I;                   // distance from an input to its gate
    f32x4 o[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + 4 * e);
        const f32x4 gate = *reinterpret_cast<const f32x4*>(a + goff + 4 * e);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[e][c] = ac::gelu_erf(x[c]) * gate[c];
        *reinterpret_cast<f32x4*>(g + row * I + 8 * q + 4 * e) = o[e];
    }
    if (planes) {
        uint4 Hh, Mm, Ll;
        ac::split8(o[0], o[1], Hh, Mm, Ll);
        uint16_t* p = planes + ac::plane_off(T, row, 8 * q);
        const int64_t plane = T * (int64_t)I;
        *reinterpret_cast<uint4*>(p) = Hh;
        *reinterpret_cast<uint4*>(p + plane) = Mm;
        *reinterpret_cast<uint4*>(p + 2 * plane) = Ll;
    }
}

// ---- padding-free ("packed") mode ---------------------------------------------------------------------------
// The reference pads every text to the longest of the batch and runs the encoder over the padding too
// (classifier.py:1259-1271); padded positions never influence real ones (additive -inf mask, row-wise LayerNorm /
// FFN) and only the CLS rows are consumed, so the padding rows can simply be left out: tokens are laid out
// sequence after sequence, [sum(len), H], and attention works per sequence on its own rows.
// pack_lens_kernel: len[s] = sum(mask[s, :]); flags a row whose ones are not a prefix (then the caller keeps the
// padded path).  pack_scan_kernel: cu = exclusive scan of len (one workgroup), info = {total rows, not-prefix flag,
// longest sequence}.  pack_fill_kernel: tok_src[cu[s] + p] = s * S + p.
__global__ __launch_bounds__(256) void pack_lens_kernel(const int64_t* __restrict__ mask, int b, int S, int32_t* __restrict__ lens,
                                                        int32_t* __restrict__ info) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= b) return;
    int n = 0, bad = 0;
    for (int p0 = 0; p0 < S; p0 += 64) {
        const int p = p0 + lane;
        const bool on = p < S && mask[(int64_t)s * S + p] != 0;
        const uint64_t m = __ballot(on);
        // a prefix inside this chunk: ones then zeros, and no ones after an earlier zero
        const int ones = __popcll(m);
        const uint64_t want = ones == 64 ? ~0ull : ((1ull << ones) - 1);
        if (m != want || (ones > 0 && n != p0)) bad = 1;
        n += ones;
    }
    if (lane == 0) { lens[s] = n; if (bad || n == 0) atomicOr(&info[1], 1); }
}
__global__ __launch_bounds__(1024) void pack_scan_kernel(const int32_t* __restrict__ lens, int b, int32_t* __restrict__ cu,
                                                         int32_t* __restrict__ info) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (b + 1023) / 1024;
    int sum = 0, mx = 0;
    for (int i = tid * per; i < (tid + 1) * per && i < b; ++i) { sum += lens[i]; mx = lens[i] > mx ? lens[i] : mx; }
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                           // inclusive Hillis-Steele scan of the partial sums
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int i = tid * per; i < (tid + 1) * per && i < b; ++i) { cu[i] = run; run += lens[i]; }
    if (tid == 1023) { cu[b] = part[1023]; info[0] = part[1023]; }
    atomicMax(&info[2], mx);
}
__global__ __launch_bounds__(256) void pack_fill_kernel(const int32_t* __restrict__ cu, int b, int S, int32_t* __restrict__ tok_src) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= b) return;
    const int r0 = cu[s], n = cu[s + 1] - r0;
    for (int p = lane; p < n; p += 64) tok_src[r0 + p] = s * S + p;
}
// out[i, :] = in[rows[i], :]   (the CLS rows of the packed layout)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ rows, int n, int H,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(in + (int64_t)rows[i] * H);
    f32x4* dst = reinterpret_cast<f32x4*>(out + (int64_t)i * H);
    for (int c = lane; c < (H >> 2); c += 64) dst[c] = src[c];
}

struct BertWs {
    size_t x, qkv, ctx, y, ffn, xp, ctxp, ffnp, small, lnctl, lnpart, cu, tile_seq, attn_xchg, pack_src, pack_info, total;     // *p: bf16x3 operand planes (3 * rows * K uint16)
    size_t lnctl_bytes, attn_xchg_bytes;
};
// cu of an UNPACKED batch without a mask (every sequence has S real tokens): what ac_bert_pack would have produced
__global__ __launch_bounds__(256) void iota_cu_kernel(int32_t* cu, int b, int S) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i <= b) cu[i] = i * S;
}
// Everything of a padding-free forward that comes before its first GEMM and needs no token count on the HOST, in ONE workgroup
// (ac_bert_encode_cls_unpad; the separate form is ac_bert_pack + 2 memsets + the verdict roll + the tile table = 9 stream
// operations and a D2H round trip of ~27 us with the GPU idle):
//   lens / prefix check / exclusive scan -> cu, info = {rows, not-prefix flag, longest, -}   (pack_lens / pack_scan)
//   tok_src[cu[s] + p] = s * S + p                                                            (pack_fill)
//   the fused attention epilogue's row-tile table                                             (ac::qkv_attn_tile_seq_kernel)
//   the LayerNorm exchange's counters and the attention exchange's words := 0, the verdict words rolled (or cleared)
//   info -> a host-mapped slot, `epoch` last: the host spins on it while the embedding kernel (launched over b * S rows, reading
//   the count from `info`) already runs
__global__ __launch_bounds__(1024) void pack_prologue_kernel(const int64_t* __restrict__ mask, int b, int S, int32_t* __restrict__ cu,
                                                             int32_t* __restrict__ tok_src, int32_t* __restrict__ info,
                                                             int32_t* __restrict__ tile_seq, unsigned* __restrict__ zero_a, int zero_a_words,
                                                             unsigned* __restrict__ zero_b, int zero_b_words, unsigned* __restrict__ verdict,
                                                             int clear_verdict, int32_t* host_slot, int epoch) {
    extern __shared__ int32_t pk_lds[];                 // lens[b] | cu[b + 1]
#ifdef AC_PROLOGUE_STAMPS
#define PK_STAMP(i) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(info + 16)[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PK_STAMP(i) do {} while (0)
#endif
    PK_STAMP(0);
    __shared__ int part[16];
    __shared__ int s_bad, s_longest;
    int32_t* lens = pk_lds;
    int32_t* cus = pk_lds + b;
    const int tid = threadIdx.x, n = b * S;
    for (int i = tid; i < b; i += 1024) lens[i] = 0;
    if (tid == 0) { s_bad = 0; s_longest = 0; }
    __syncthreads();
    for (int i = tid; i < zero_a_words; i += 1024) zero_a[i] = 0u;
    for (int i = tid; i < zero_b_words; i += 1024) zero_b[i] = 0u;
    if (tid == 0) {
        if (clear_verdict) { verdict[0] = 0u; verdict[1] = 0u; }
        else { verdict[1] |= verdict[0]; verdict[0] = 0u; }
    }
    PK_STAMP(1);
    int bad = 0;
    for (int i0 = 8 * tid; i0 < n; i0 += 8 * 1024) {    // ones form a prefix of the row <=> no one right after a zero
        int64_t m[9];                                   // this thread's eight consecutive elements and the one before them:
        m[0] = i0 > 0 ? mask[i0 - 1] : 1;               // nine independent loads in flight (8.7 -> 3 us against two dependent
#pragma unroll                                          //  rounds of strided single elements)
        for (int u = 0; u < 8; ++u) m[u + 1] = i0 + u < n ? mask[i0 + u] : 0;
        int row = i0 / S, p = i0 - row * S, ones = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u < n && m[u + 1] != 0) {
                ++ones;
                if (p > 0 && m[u] == 0) bad = 1;
            }
            if (++p == S) {                             // (one LDS atomic per thread and row, not per element)
                if (ones) atomicAdd(&lens[row], ones);
                p = 0; ++row; ones = 0;
            }
        }
        if (ones) atomicAdd(&lens[row], ones);
    }
    if (bad) s_bad = 1;
    __syncthreads();
    PK_STAMP(2);
    const int per = (b + 1023) / 1024;
    int sum = 0, mx = 0;
    for (int i = tid * per; i < (tid + 1) * per && i < b; ++i) {
        const int l = lens[i];
        sum += l; mx = l > mx ? l : mx;
        if (l == 0) bad = 1;
    }
    if (bad) s_bad = 1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_xor(mx, o); mx = v > mx ? v : mx; }
    if ((tid & 63) == 0 && mx) atomicMax(&s_longest, mx);        // (one atomic per wave: 256 on one LDS word cost 3 us)
    // inclusive scan of the 1024 partial sums: inside each wave on shuffles, then the 16 wave totals (two barriers instead of the
    // twenty of a Hillis-Steele pass over LDS)
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) part[tid >> 6] = incl;
    __syncthreads();
    int wave_base = 0, grand = 0;
#pragma unroll
    for (int wv = 0; wv < 16; ++wv) {
        const int t = part[wv];
        if (wv < (tid >> 6)) wave_base += t;
        grand += t;
    }
    incl += wave_base;
    int run = incl - sum;
    for (int i = tid * per; i < (tid + 1) * per && i < b; ++i) { cus[i] = run; cu[i] = run; run += lens[i]; }
    const int total = grand;
    PK_STAMP(3);
    if (tid == 0) {
        cus[b] = total; cu[b] = total; info[0] = total; info[1] = s_bad; info[2] = s_longest; info[3] = 0;
        if (host_slot) {                                // the host's copy leaves now: it sizes the GEMM launches while the rest runs
            host_slot[0] = total; host_slot[1] = s_bad; host_slot[2] = s_longest;
            __threadfence_system();
            __hip_atomic_store(&host_slot[3], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    PK_STAMP(4);
    for (int i = tid; i < n; i += 1024) {
        const int q = i / S, p = i - q * S;
        if (p < lens[q]) tok_src[cus[q] + p] = i;
    }
    PK_STAMP(5);
    // the row-tile table (gemm_pipe.hip qkv_attn_tile_seq_kernel): a wave per tile
    const int ntiles = (total + ac::kQkvAttnRows - 1) / ac::kQkvAttnRows, lane = tid & 63;
    for (int t = tid >> 6; t < ntiles; t += 16) {
        auto first_at_or_after = [&](int row) { int lo = 0, hi = b; while (lo < hi) { const int mid = (lo + hi) >> 1; if (cus[mid] < row) lo = mid + 1; else hi = mid; } return lo; };
        const int sb = first_at_or_after(t * ac::kQkvAttnRows), se = first_at_or_after((t + 1) * ac::kQkvAttnRows), ns = se - sb;
        for (int i = lane; i < ac::kQkvAttnCu; i += 64) {
            int v = 0;
            if (i == 0) v = ns;
            else if (i - 1 <= ns) v = cus[sb + i - 1];
            tile_seq[(size_t)t * ac::kQkvAttnCu + i] = v;
        }
    }
    PK_STAMP(6);
}
// The verdict of the fused-LayerNorm GEMM epilogues sits at offset 0 of the workspace WHATEVER (b, S) the workspace is used
// with: word 0 = "a panel of the CURRENT call gave up" (set by the kernels; later launches of the call stop waiting at their first
// look at it), word 1 = the same for EARLIER calls since the last ac_bert_ln_fusion_clear.  Every call starts by rolling word 0
// into word 1 (so a C caller that never clears still starts every call with a clean word 0, and the verdict of a multi-chunk
// encode -- chunks of different row counts sharing one workspace -- survives the later chunks); ac_bert_ln_fusion_status
// reports word 0 | word 1.
constexpr size_t kLnAbortHead = 256;
__global__ void ln_verdict_roll_kernel(unsigned* w) {
    w[1] |= w[0];
    w[0] = 0u;
}
BertWs bert_ws(const ac_bert_config& c, int b, int S) {
    BertWs w;
    const size_t T = (size_t)b * S;
    size_t off = kLnAbortHead;
    auto take = [&](size_t n) { size_t o = off; off += ac::align_up(n * sizeof(float), 256); return o; };
    w.x = take(T * c.hidden);
    w.qkv = take(T * 3 * c.hidden);
    w.ctx = take(T * c.hidden);
    w.y = take(T * c.hidden);
    w.ffn = take(T * c.intermediate);
    auto take16 = [&](size_t n) { size_t o = off; off += ac::align_up(n * sizeof(uint16_t), 256); return o; };
    w.xp = take16(3 * T * c.hidden);
    w.ctxp = take16(3 * T * c.hidden);
    w.ffnp = take16(3 * T * c.intermediate);
    // the one-launch small-batch path (bert_small.hip) works on 32 padded rows of its own
    w.small = off;
    if (T <= 32) off += ac::bert_small_ws_bytes(c.hidden, c.intermediate);
    // fused LayerNorm epilogues (gemm_pipe.hip): one counter per (layer, LayerNorm, 128-row panel), partials
    w.lnctl = off;
    w.lnctl_bytes = ac::align_up((size_t)c.layers * 2 * ac::pipe_ln_panels((int)T) * sizeof(unsigned), 256);
    off += w.lnctl_bytes;
    w.lnpart = off;
    off += ac::align_up(ac::pipe_ln_part_bytes((int)T, c.hidden), 256);
    w.cu = off;                       // sequence offsets of an unpacked, unmasked batch + the row tiles' first sequences (fused attention epilogue)
    off += ac::align_up((size_t)(b + 1) * sizeof(int32_t), 256);
    w.tile_seq = off;
    off += ac::align_up(ac::qkv_attn_tile_seq_bytes((int)T), 256);
    w.attn_xchg = off;                // one word per (256-row tile, head): the in-launch exchange of straddling sequences
    w.attn_xchg_bytes = ac::align_up(((T + ac::kQkvAttnRows - 1) / ac::kQkvAttnRows) * (size_t)c.heads * sizeof(unsigned), 256);
    off += w.attn_xchg_bytes;
    w.pack_src = off;                 // ac_bert_encode_cls_unpad: token sources of the packed rows, {rows, flag, longest, -}
    off += ac::align_up(T * sizeof(int32_t), 256);
    w.pack_info = off;
    off += 256;
    w.total = off;
    return w;
}

int check_cfg(const ac_bert_config* c) {
    AC_REQUIRE(c != nullptr, AC_EINVAL, "bert: config is NULL");
    AC_REQUIRE(c->hidden >= 64 && c->layers >= 1 && c->heads >= 1 && c->intermediate >= 4, AC_EINVAL,
               "bert: bad config");
    AC_REQUIRE(c->hidden % 4 == 0 && c->hidden <= 64 * 4 * kMaxVec, AC_EUNSUPPORTED,
               "bert: hidden=%d unsupported (must be a multiple of 4 and <= %d)", c->hidden, 64 * 4 * kMaxVec);
    AC_REQUIRE(c->hidden == c->heads * 64 || c->hidden == c->heads * 32, AC_EUNSUPPORTED,
               "bert: head dim %d unsupported (64 or 32)", c->hidden / c->heads);
    AC_REQUIRE(c->gemm_arith_opt >= 0 && c->gemm_arith_opt <= 3 && c->ln_fusion_opt >= 0 && c->ln_fusion_opt <= 3 &&
                   c->one_launch_opt >= 0 && c->one_launch_opt <= 2, AC_EINVAL,
               "bert: per-call options out of range (gemm_arith_opt %d, ln_fusion_opt %d, one_launch_opt %d)", c->gemm_arith_opt,
               c->ln_fusion_opt, c->one_launch_opt);
    return AC_OK;
}

}  // namespace

extern "C" int ac_bert_workspace(const ac_bert_config* cfg, int b, int S, size_t* bytes) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    AC_REQUIRE(bytes && b >= 0 && S >= 1, AC_EINVAL, "bert workspace: bad arguments");
    *bytes = bert_ws(*cfg, b > 0 ? b : 1, S).total;
    return AC_OK;
}

namespace {

// shared body of ac_bert_encode_cls (cu == nullptr: the [b, S] rows incl. padding, T = b * S) and
// ac_bert_encode_cls_packed (cu / tok_src from ac_bert_pack: T = cu[b] real-token rows, no mask)
// AC_BERT_TAIL_FUSED=0: the last layer ends with gemm_splitk_reduce -> ln_kernel_t -> cls_normalize_kernel (A/B runs, the equivalence test)
static bool tail_fused() {
    const char* e = getenv("AC_BERT_TAIL_FUSED");
    return !(e && atoi(e) == 0);
}
// What of a forward is decided by the token-row count alone: operand planes between the GEMMs, fp16x2 planes
struct EncodePlan { bool wplanes, pl, f16; };
EncodePlan encode_plan(const ac_bert_config& c, const ac_bert_weights* w, int T) {
    const int H = c.hidden, I = c.intermediate;
    EncodePlan p;
    p.wplanes = w->qkv_w3 && w->ao_w3 && w->ff1_w3 && w->ff2_w3;
    p.pl = p.wplanes && ac::linear_takes_planes(T, H, H) && ac::linear_takes_planes(T, H, I) && (H % 8) == 0;
    p.f16 = p.pl && ac::gemm_arith() == AC_GEMM_F16X2 && w->qkv_wh && w->ao_wh && w->ff1_wh && w->ff2_wh && c.layers > 1 &&
            ac::linear_f16x2_takes(T, 3 * H, H) && ac::linear_f16x2_takes(T, H, H) && ac::linear_f16x2_takes(T, I, H) &&
            ac::linear_f16x2_takes(T, H, I);
    return p;
}
// the embedding launch over `rows` token rows (t_dev: the real count is read on the device, see embed_ln_kernel)
void launch_embed(const ac_bert_config& c, const ac_bert_weights* w, const EncodePlan& pn, const int64_t* d_ids, const int64_t* d_type_ids,
                  int rows, int S, float* x, uint16_t* xp, const int32_t* tok_src, const int32_t* t_dev, hipStream_t stream) {
    const int H = c.hidden, tok_blocks = (rows + 3) / 4;
    if (pn.f16)
        hipLaunchKernelGGL(embed_ln_kernel<true>, dim3(tok_blocks), dim3(256), 0, stream, d_ids, d_type_ids, rows, S, H,
                           w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g, w->emb_ln_b, c.ln_eps, x, xp, tok_src, t_dev);
    else
        hipLaunchKernelGGL(embed_ln_kernel<false>, dim3(tok_blocks), dim3(256), 0, stream, d_ids, d_type_ids, rows, S, H,
                           w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g, w->emb_ln_b, c.ln_eps, x,
                           pn.pl ? xp : nullptr, tok_src, t_dev);
}
// pre (ac_bert_encode_cls_unpad's pack_prologue_kernel ran on this stream): kPreCtl = the verdict roll, the zeroing of the exchange
// words and the row-tile table are done; kPreEmbed = the embedding kernel too
constexpr int kPreCtl = 1, kPreEmbed = 2;
int bert_encode_impl(const ac_bert_config* cfg, const ac_bert_weights* w, const int64_t* d_ids, const int64_t* d_type_ids,
                     const int64_t* d_mask, int b, int S, const int32_t* cu, const int32_t* tok_src, int T, int Smax,
                     float* d_out, int64_t ldo, void* d_ws, size_t ws_bytes, hipStream_t stream, int pre = 0) {
    int rc;
    const ac_bert_config& c = *cfg;
    const BertWs ws = bert_ws(c, b, S);
    AC_REQUIRE(d_ws && ws_bytes >= ws.total, AC_EWORKSPACE, "bert_encode_cls: workspace %zu < %zu", ws_bytes, ws.total);
    char* base = (char*)d_ws;
    float* x = (float*)(base + ws.x);
    float* qkv = (float*)(base + ws.qkv);
    float* ctx = (float*)(base + ws.ctx);
    float* y = (float*)(base + ws.y);
    float* ffn = (float*)(base + ws.ffn);
    const int H = c.hidden, I = c.intermediate;

    uint16_t* xp = (uint16_t*)(base + ws.xp);
    uint16_t* ctxp = (uint16_t*)(base + ws.ctxp);
    uint16_t* ffnp = (uint16_t*)(base + ws.ffnp);
    // Pre-split operand planes (AC_GEMM_BF16X3 with weight planes present): every producer of a GEMM input
    // -- the LayerNorms, the attention kernel, the GELU epilogue of FFN1 -- emits the bf16x3 planes the next
    // GEMM stages with direct global->LDS loads, so no GEMM splits its operands again.  The T-row GEMMs of
    // layers 0 .. L-2 qualify; the CLS-only last layer (b rows) keeps fp32 activations.
    const EncodePlan pn = encode_plan(c, w, T);
    const bool wplanes = pn.wplanes, pl = pn.pl;
    // bias + residual + LayerNorm in the epilogue of the attention-output and FFN2 GEMMs (one-round launches only)
    // AC_GEMM_F16X2 (opt-in): the same flow on fp16x2 planes when the fp16 weight planes are there and all four token-row GEMMs
    // take the ring-staged kernel; the CLS-only tail of the last layer (fp32 activations, b rows) stays bf16x3
    const bool f16 = pn.f16;
    const bool fuse_ln = pl && c.layers > 1 && ac::pipe_ln_applies(T, H, H) && ac::pipe_ln_applies(T, H, I);
    unsigned* ln_abort = (unsigned*)base;
    if (!(pre & kPreCtl)) {
        hipLaunchKernelGGL(ln_verdict_roll_kernel, dim3(1), dim3(1), 0, stream, ln_abort);
        AC_LAUNCH_CHECK();
    }
    unsigned* ln_count = (unsigned*)(base + ws.lnctl);
    const int ln_panels = ac::pipe_ln_panels(T);
    if (fuse_ln && !(pre & kPreCtl)) AC_HIP_CHECK(hipMemsetAsync(base + ws.lnctl, 0, ws.lnctl_bytes, stream));

    if (!(pre & kPreEmbed)) {
        launch_embed(c, w, pn, d_ids, d_type_ids, T, S, x, xp, tok_src, nullptr, stream);
        AC_LAUNCH_CHECK();
    }
    const int dh = c.hidden / c.heads;                // 64, or 32 (MiniLM family)
    const float scale = 1.0f / sqrtf((float)dh);
    // Self-attention inside the QKV GEMM's epilogue (gemm_pipe.hip EPI_QKV_ATTN): sequences laid out row after row (packed, or
    // unpacked without a mask = every sequence S real tokens), head dim 64, longest sequence <= 64, layers 0 .. L-2 on planes
    const int32_t* cu_at = cu;
    const bool fuse_attn = pl && c.layers > 1 && dh == 64 && (cu || !d_mask) && ac::qkv_attn_applies(T, H, c.heads, Smax);
    if (fuse_attn && !cu) {
        int32_t* cuw = (int32_t*)(base + ws.cu);
        hipLaunchKernelGGL(iota_cu_kernel, dim3((b + 256) / 256), dim3(256), 0, stream, cuw, b, S);
        AC_LAUNCH_CHECK();
        cu_at = cuw;
    }
    int32_t* tile_seq = (int32_t*)(base + ws.tile_seq);
    unsigned* attn_xchg = nullptr;
    if (fuse_attn) {
        if (!(pre & kPreCtl) || !cu) {
            rc = ac::qkv_attn_tile_seq(cu_at, b, T, tile_seq, stream);
            if (rc) return rc;
        }
        if (ac::qkv_attn_exchange_applies(T, c.heads, (int)f16)) {        // every tile resident (proven): no boundary launches
            attn_xchg = (unsigned*)(base + ws.attn_xchg);
            if (!(pre & kPreCtl)) AC_HIP_CHECK(hipMemsetAsync(attn_xchg, 0, ws.attn_xchg_bytes, stream));
        }
    }
    for (int l = 0; l < c.layers; ++l) {
        const uint16_t* qkv_w3 = wplanes ? w->qkv_w3[l] : nullptr;
        const uint16_t* ao_w3 = wplanes ? w->ao_w3[l] : nullptr;
        const uint16_t* ff1_w3 = wplanes ? w->ff1_w3[l] : nullptr;
        const uint16_t* ff2_w3 = wplanes ? w->ff2_w3[l] : nullptr;
        const bool last = (l == c.layers - 1);
        // After the last layer's attention only the CLS row of each sequence is consumed, so the
        // output projection, both LayerNorms and the FFN run on b rows instead of b*S.
        const int Ml = last ? b : T;
        const bool lp = pl && !last;                   // this layer's post-attention GEMMs run on planes
        const float* resid = x;                        // residual = layer input
        int64_t ldres = last ? (int64_t)S * H : H;     // CLS rows of x are S*H apart (padded layout)
        // ... and BEFORE it only K and V of every token and the Q of the CLS rows (round 5): the last QKV projection becomes a
        // [T, 2H] GEMM over the K | V rows of the fused weight (a third fewer flops of that GEMM) + a [b, H] one for Q
        const bool q_cls_only = last && T >= 4 * b;    // (short sequences: the split saves nothing worth two launches)
        if (last && cu) {                              // packed layout: the CLS rows sit at cu[s]; gather them
            hipLaunchKernelGGL(gather_rows_kernel, dim3((b + 3) / 4), dim3(256), 0, stream, x, cu, b, H, ffn);
            AC_LAUNCH_CHECK();
            resid = ffn;                               // compact CLS rows, staged in ffn (free until FFN1 writes it)
            ldres = H;
        }
        if (fuse_attn && !last) {
            rc = ac::launch_gemm_pipe_qkv_attn(xp, T, f16 ? w->qkv_wh[l] : qkv_w3, 3 * H, w->qkv_b[l], T, H, c.heads, cu_at, tile_seq, b, Smax,
                                               scale, ctxp, qkv, stream, (int)f16, attn_xchg, (unsigned)(l + 1), ln_abort);
            if (rc) return rc;
            // the sequences that straddle a 256-row tile boundary (their q | k | v rows are in qkv): one wave per boundary --
            // unless the tiles exchanged the rows among themselves inside the launch
            const int nbound = (T - 1) / ac::kQkvAttnRows;
            if (nbound > 0 && !attn_xchg) {
                hipLaunchKernelGGL((attention_mfma_kernel<false, 64>), dim3((Smax + 31) / 32, c.heads, nbound), dim3(64), 0, stream, qkv,
                                   nullptr, S, H, scale, ctx, ctxp, nullptr, nullptr, -1, cu_at, (int64_t)T, (int)f16, ac::kQkvAttnRows, b);
                AC_LAUNCH_CHECK();
            }
        } else if (q_cls_only) {
            const int64_t HH = (int64_t)H * H;
            rc = f16 ? ac::linear_f16x2(xp, w->qkv_wh[l] + (int64_t)H * 8, w->qkv_b[l] + H, nullptr, 0, qkv + H, 3 * H, nullptr, T, 2 * H, H, 0,
                                        stream, 3 * H)
                     : ac::linear_f32(x, H, w->qkv_w[l] + HH, H, w->qkv_b[l] + H, nullptr, 0, qkv + H, 3 * H, T, 2 * H, H, 0, nullptr, 1.f,
                                      stream, 0.f, 0, qkv_w3 ? qkv_w3 + (int64_t)H * 8 : nullptr, pl ? xp : nullptr, nullptr, 3 * H);
            if (rc) return rc;
            // Q of the CLS rows -> y (compact [b, H]; free until the output projection writes it)
            rc = ac::linear_f32_splitk(resid, ldres, w->qkv_w[l], H, w->qkv_b[l], nullptr, 0, y, H, b, H, H, 0, qkv_w3, ctx,
                                       (size_t)T * H * sizeof(float), stream, 3 * H);
        } else {
            rc = f16 ? ac::linear_f16x2(xp, w->qkv_wh[l], w->qkv_b[l], nullptr, 0, qkv, 3 * H, nullptr, T, 3 * H, H, 0, stream)
                     : ac::linear_f32(x, H, w->qkv_w[l], H, w->qkv_b[l], nullptr, 0, qkv, 3 * H, T, 3 * H, H, 0, nullptr, 1.f,
                                      stream, 0.f, 0, qkv_w3, pl ? xp : nullptr);
        }
        if (rc) return rc;
        if (fuse_attn && !last) {
            // (context rows are already in ctxp)
        } else if (last) {
            const float* qc = q_cls_only ? y : nullptr;
            if (dh == 64 && Smax <= 32)
                hipLaunchKernelGGL((attention_cls_kernel<false, 64, 32>), dim3(c.heads, b), dim3(64), 0, stream, qkv, d_mask, S, H, scale,
                                   ctx, nullptr, nullptr, -1, cu, qc);
            else if (dh == 64)
                hipLaunchKernelGGL((attention_cls_kernel<false, 64>), dim3(c.heads, b), dim3(64), 0, stream, qkv, d_mask, S, H, scale,
                                   ctx, nullptr, nullptr, -1, cu, qc);
            else
                hipLaunchKernelGGL((attention_cls_kernel<false, 32>), dim3(c.heads, b), dim3(64), 0, stream, qkv, d_mask, S, H, scale,
                                   ctx, nullptr, nullptr, -1, cu, qc);
        } else {
            if (dh == 64)
                hipLaunchKernelGGL((attention_mfma_kernel<false, 64>), dim3((Smax + 31) / 32, c.heads, b), dim3(64), 0, stream, qkv,
                                   d_mask, S, H, scale, ctx, lp ? ctxp : nullptr, nullptr, nullptr, -1, cu, (int64_t)T, (int)f16);
            else
                hipLaunchKernelGGL((attention_mfma_kernel<false, 32>), dim3((Smax + 31) / 32, c.heads, b), dim3(64), 0, stream, qkv,
                                   d_mask, S, H, scale, ctx, lp ? ctxp : nullptr, nullptr, nullptr, -1, cu, (int64_t)T, (int)f16);
        }
        AC_LAUNCH_CHECK();
        const int lblocks = (Ml + 3) / 4;
        const bool fl = fuse_ln && lp;                 // x <- LayerNorm(x + ctx Wo^T + b) in ONE launch, in place
        if (fl) {
            rc = ac::launch_gemm_pipe_ln(ctxp, Ml, f16 ? w->ao_wh[l] : ao_w3, H, w->ao_b[l], x, H, x, H, Ml, H, H, w->ln1_g[l], w->ln1_b[l], c.ln_eps,
                                         base + ws.lnpart, ln_count + (size_t)(2 * l) * ln_panels, ln_abort, xp, stream, (int)f16);
            if (rc) return rc;
        } else {
            // (last layer: b CLS rows = a handful of output tiles -> split-K over the qkv buffer, which is dead by now)
            rc = last ? ac::linear_f32_splitk(ctx, H, w->ao_w[l], H, w->ao_b[l], resid, ldres, y, H, Ml, H, H, 0, ao_w3, qkv,
                                              (size_t)T * 3 * H * sizeof(float), stream)
                      : (f16 ? ac::linear_f16x2(ctxp, w->ao_wh[l], w->ao_b[l], resid, ldres, y, H, nullptr, Ml, H, H, 0, stream)
                             : ac::linear_f32(ctx, H, w->ao_w[l], H, w->ao_b[l], resid, ldres, y, H, Ml, H, H, 0, nullptr, 1.f, stream,
                                              0.f, 0, ao_w3, lp ? ctxp : nullptr));
            if (rc) return rc;
            // (last layer: x is overwritten with b compact rows; its old contents are no longer needed)
            launch_ln(f16, lblocks, stream, y, Ml, H, w->ln1_g[l], w->ln1_b[l], c.ln_eps, last ? ctx : x, lp ? xp : nullptr, (int64_t)H);
            AC_LAUNCH_CHECK();
        }
        float* x1 = last ? ctx : x;                    // ctx is free again after the AO projection
        rc = last ? ac::linear_f32_splitk(x1, H, w->ff1_w[l], H, w->ff1_b[l], nullptr, 0, ffn, I, Ml, I, H, 2, ff1_w3, qkv,
                                          (size_t)T * 3 * H * sizeof(float), stream)
                  : (f16 ? ac::linear_f16x2(xp, w->ff1_wh[l], w->ff1_b[l], nullptr, 0, nullptr, I, ffnp, Ml, I, H, 2, stream)
                         : ac::linear_f32(x1, H, w->ff1_w[l], H, w->ff1_b[l], nullptr, 0, ffn, I, Ml, I, H, 2, nullptr, 1.f, stream,
                                          0.f, 0, ff1_w3, lp ? xp : nullptr, lp ? ffnp : nullptr));
        if (rc) return rc;
        if (fl) {                                      // x <- LayerNorm(x + ffn W2^T + b), planes for the next layer's QKV GEMM
            rc = ac::launch_gemm_pipe_ln(ffnp, Ml, f16 ? w->ff2_wh[l] : ff2_w3, H, w->ff2_b[l], x, H, x, H, Ml, H, I, w->ln2_g[l], w->ln2_b[l], c.ln_eps,
                                         base + ws.lnpart, ln_count + (size_t)(2 * l + 1) * ln_panels, ln_abort, xp, stream, (int)f16);
            if (rc) return rc;
            continue;
        }
        if (last && tail_fused()) {
            // the forward's last three launches as one: K slices + bias + residual -> LayerNorm -> F.normalize -> d_out
            int ks = 0;
            rc = ac::linear_f32_splitk(ffn, I, w->ff2_w[l], I, w->ff2_b[l], x1, H, y, H, Ml, H, I, 0, ff2_w3, qkv,
                                       (size_t)T * 3 * H * sizeof(float), stream, 0, &ks);
            if (rc) return rc;
            if (ks > 0) {
                // (a wave per CLS row, a workgroup each: spread over the CUs)
                hipLaunchKernelGGL(splitk_ln_normalize_kernel, dim3(b), dim3(64), 0, stream, qkv, ks, b, H, w->ff2_b[l], x1, (int64_t)H,
                                   w->ln2_g[l], w->ln2_b[l], c.ln_eps, d_out, ldo);
                AC_LAUNCH_CHECK();
                return AC_OK;
            }
            launch_ln(f16, lblocks, stream, y, Ml, H, w->ln2_g[l], w->ln2_b[l], c.ln_eps, x, nullptr, (int64_t)H);
            AC_LAUNCH_CHECK();
            break;                                     // (the product did not take the split-K form: the separate kernels)
        }
        rc = last ? ac::linear_f32_splitk(ffn, I, w->ff2_w[l], I, w->ff2_b[l], x1, H, y, H, Ml, H, I, 0, ff2_w3, qkv,
                                          (size_t)T * 3 * H * sizeof(float), stream)
                  : (f16 ? ac::linear_f16x2(ffnp, w->ff2_wh[l], w->ff2_b[l], x1, H, y, H, nullptr, Ml, H, I, 0, stream)
                         : ac::linear_f32(ffn, I, w->ff2_w[l], I, w->ff2_b[l], x1, H, y, H, Ml, H, I, 0, nullptr, 1.f, stream, 0.f, 0,
                                          ff2_w3, lp ? ffnp : nullptr));
        if (rc) return rc;
        // the next layer's QKV GEMM reads x as planes; the last layer's output (b compact rows) stays fp32
        launch_ln(f16, lblocks, stream, y, Ml, H, w->ln2_g[l], w->ln2_b[l], c.ln_eps, x, lp ? xp : nullptr, (int64_t)H);
        AC_LAUNCH_CHECK();
    }
    // after the CLS-only last layer x holds b compact rows (sequence stride 1)
    hipLaunchKernelGGL(cls_normalize_kernel, dim3((b + 3) / 4), dim3(256), 0, stream, x, b, 1, H, d_out, ldo);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

}  // namespace

extern "C" int ac_bert_encode_cls_opts(const ac_bert_config* cfg, const ac_bert_weights* w, const int64_t* d_ids,
                                       const int64_t* d_type_ids, const int64_t* d_mask, int b, int S,
                                       float* d_out, int64_t ldo, void* d_ws, size_t ws_bytes, int opts, int* used_one_launch,
                                       ac_stream_t stream_) {
    if (used_one_launch) *used_one_launch = 0;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (b == 0) return AC_OK;
    const ac::CallScope scope(cfg->gemm_arith_opt, cfg->ln_fusion_opt, cfg->one_launch_opt);
    AC_REQUIRE(w && d_ids && d_out && b > 0 && S >= 1 && S <= cfg->max_pos && ldo >= cfg->hidden, AC_EINVAL,
               "bert_encode_cls: bad arguments (b=%d S=%d max_pos=%d)", b, S, cfg->max_pos);
    if (b * S <= 32 && !(opts & AC_BERT_LAYERED)) {   // a handful of token rows (single-query predict): every layer in ONE persistent launch
        const BertWs ws = bert_ws(*cfg, b, S);
        AC_REQUIRE(d_ws && ws_bytes >= ws.total, AC_EWORKSPACE, "bert_encode_cls: workspace %zu < %zu", ws_bytes, ws.total);
        rc = ac::bert_small_encode(*cfg, *w, d_ids, d_type_ids, d_mask, b, S, d_out, ldo, (char*)d_ws + ws.small, (hipStream_t)stream_);
        if (rc == AC_OK && used_one_launch) *used_one_launch = 1;
        if (rc != 1) return rc;
    }
    return bert_encode_impl(cfg, w, d_ids, d_type_ids, d_mask, b, S, nullptr, nullptr, b * S, S, d_out, ldo, d_ws, ws_bytes,
                            (hipStream_t)stream_);
}

extern "C" int ac_bert_encode_cls(const ac_bert_config* cfg, const ac_bert_weights* w, const int64_t* d_ids,
                                  const int64_t* d_type_ids, const int64_t* d_mask, int b, int S,
                                  float* d_out, int64_t ldo, void* d_ws, size_t ws_bytes, ac_stream_t stream_) {
    return ac_bert_encode_cls_opts(cfg, w, d_ids, d_type_ids, d_mask, b, S, d_out, ldo, d_ws, ws_bytes, 0, nullptr, stream_);
}

extern "C" int ac_bert_one_launch_status(const ac_bert_config* cfg, int b, int S, const void* d_ws, size_t ws_bytes,
                                         int* aborted, ac_stream_t stream_) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    AC_REQUIRE(aborted && d_ws && b > 0 && S >= 1 && b * S <= 32, AC_EINVAL, "bert_one_launch_status: bad arguments");
    const BertWs ws = bert_ws(*cfg, b, S);
    AC_REQUIRE(ws_bytes >= ws.total, AC_EWORKSPACE, "bert_one_launch_status: workspace %zu < %zu", ws_bytes, ws.total);
    return ac::bert_small_aborted(cfg->hidden, cfg->intermediate, (const char*)d_ws + ws.small, (hipStream_t)stream_, aborted);
}

extern "C" int ac_bert_ln_fusion_status(const ac_bert_config* cfg, int b, int S, const void* d_ws, size_t ws_bytes,
                                       int* aborted, ac_stream_t stream_) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    (void)b; (void)S;                                  // (kept in the signature: the word no longer depends on the chunk shape)
    AC_REQUIRE(aborted && d_ws && ws_bytes >= kLnAbortHead, AC_EINVAL, "bert_ln_fusion_status: bad arguments");
    unsigned flag[2] = {0, 0};
    AC_HIP_CHECK(hipMemcpyAsync(flag, (const char*)d_ws, sizeof(flag), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    AC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
    *aborted = (flag[0] | flag[1]) != 0;
    return AC_OK;
}

extern "C" int ac_bert_ln_fusion_clear(void* d_ws, size_t ws_bytes, ac_stream_t stream_) {
    AC_REQUIRE(d_ws && ws_bytes >= kLnAbortHead, AC_EINVAL, "bert_ln_fusion_clear: bad arguments");
    AC_HIP_CHECK(hipMemsetAsync(d_ws, 0, 2 * sizeof(unsigned), (hipStream_t)stream_));
    return AC_OK;
}

extern "C" int ac_bert_pack(const int64_t* d_mask, int b, int S, int32_t* d_cu, int32_t* d_tok_src, int32_t* d_info,
                            ac_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(d_mask && d_cu && d_tok_src && d_info && b >= 1 && S >= 1, AC_EINVAL, "bert_pack: bad arguments");
    AC_HIP_CHECK(hipMemsetAsync(d_info, 0, 4 * sizeof(int32_t), stream));
    int32_t* lens = d_tok_src;                         // scratch: the lengths live in tok_src until the fill pass
    hipLaunchKernelGGL(pack_lens_kernel, dim3((b + 3) / 4), dim3(256), 0, stream, d_mask, b, S, lens, d_info);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, stream, lens, b, d_cu, d_info);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(pack_fill_kernel, dim3((b + 3) / 4), dim3(256), 0, stream, d_cu, b, S, d_tok_src);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

namespace {
// Host-mapped slots the prologue kernel reports {rows, not-prefix flag, longest, epoch} into (fine-grained pinned memory: the host
// sees the stores while later kernels of the stream run).  64 slots of 64 bytes, handed out round-robin: concurrent callers
// (threads / streams) never share a slot in flight unless 64 calls overtake one.
int32_t* info_slot(int* epoch_out) {
    static int32_t* base = [] {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64 * 64, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return (int32_t*)nullptr; }
        memset(p, 0, 64 * 64);
        return (int32_t*)p;
    }();
    static std::atomic<unsigned> next{0};
    const unsigned n = next.fetch_add(1, std::memory_order_relaxed) + 1;
    *epoch_out = (int)(n & 0x3fffffff) + 1;               // never 0 (the slots' initial value)
    return base ? base + 16 * (n & 63) : nullptr;
}
}  // namespace

// (measurement: how long the last ac_bert_encode_cls_unpad call waited for the packing kernel's report, launch latency included)
static std::atomic<long long> g_unpad_wait_ns{0};
extern "C" long long ac_bert_unpad_last_wait_ns(void) { return g_unpad_wait_ns.load(std::memory_order_relaxed); }

// ac_bert_pack + ac_bert_encode_cls_packed as ONE call without a stream synchronisation (include/acamd.h)
extern "C" int ac_bert_encode_cls_unpad(const ac_bert_config* cfg, const ac_bert_weights* w, const int64_t* d_ids,
                                        const int64_t* d_type_ids, const int64_t* d_mask, int b, int S, float* d_out, int64_t ldo,
                                        void* d_ws, size_t ws_bytes, int clear_verdict, int* total_tokens, int* path, ac_stream_t stream_) {
    if (total_tokens) *total_tokens = 0;
    if (path) *path = AC_BERT_PATH_PACKED;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (b == 0) return AC_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const ac::CallScope scope(cfg->gemm_arith_opt, cfg->ln_fusion_opt, cfg->one_launch_opt);
    AC_REQUIRE(w && d_ids && d_mask && d_out && b > 0 && S >= 2 && S <= cfg->max_pos && ldo >= cfg->hidden && (int64_t)b * S > 32 &&
                   (int64_t)b * S < ((int64_t)1 << 30),
               AC_EINVAL, "bert_encode_cls_unpad: bad arguments (b=%d S=%d max_pos=%d; more than 32 token rows)", b, S, cfg->max_pos);
    const ac_bert_config& c = *cfg;
    const BertWs ws = bert_ws(c, b, S);
    AC_REQUIRE(d_ws && ws_bytes >= ws.total, AC_EWORKSPACE, "bert_encode_cls_unpad: workspace %zu < %zu", ws_bytes, ws.total);
    char* base = (char*)d_ws;
    int32_t* cu = (int32_t*)(base + ws.cu);
    int32_t* src = (int32_t*)(base + ws.pack_src);
    int32_t* info = (int32_t*)(base + ws.pack_info);
    int epoch = 0;
    int32_t* slot = info_slot(&epoch);
    const size_t lds = (size_t)(2 * b + 1) * sizeof(int32_t);
    AC_REQUIRE(slot && lds <= 96 * 1024, AC_EUNSUPPORTED, "bert_encode_cls_unpad: %d sequences in one call (or no pinned host memory): "
               "use ac_bert_pack + ac_bert_encode_cls_packed", b);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)pack_prologue_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(pack_prologue_kernel, dim3(1), dim3(1024), lds, stream, d_mask, b, S, cu, src, info, (int32_t*)(base + ws.tile_seq),
                       (unsigned*)(base + ws.lnctl), (int)(ws.lnctl_bytes / 4), (unsigned*)(base + ws.attn_xchg), (int)(ws.attn_xchg_bytes / 4),
                       (unsigned*)base, clear_verdict, slot, epoch);
    AC_LAUNCH_CHECK();
    // the embedding kernel does not wait for the host: same plan at the fewest (one per sequence) and the most (b * S) token rows
    // a batch can have <=> the launch is the one the count would have chosen
    const EncodePlan lo = encode_plan(c, w, b), hi = encode_plan(c, w, b * S);
    const bool early = lo.pl == hi.pl && lo.f16 == hi.f16;
    if (early) {
        launch_embed(c, w, hi, d_ids, d_type_ids, b * S, S, (float*)(base + ws.x), (uint16_t*)(base + ws.xp), src, info, stream);
        AC_LAUNCH_CHECK();
    }
    // the host's copy of {rows, flag, longest}: a short spin on the slot (the kernel above is ~10 us of a stream that is normally
    // empty), then a blocking wait for whatever was queued ahead of this call
    volatile int32_t* vs = slot;
    bool seen = false;
    const auto wait_t0 = std::chrono::steady_clock::now();
    for (int spin = 0; spin < 200000 && !seen; ++spin) {
        seen = __atomic_load_n(&vs[3], __ATOMIC_ACQUIRE) == epoch;
        if (!seen) __builtin_ia32_pause();
    }
    if (!seen) {
        AC_HIP_CHECK(hipStreamSynchronize(stream));
        seen = __atomic_load_n(&vs[3], __ATOMIC_ACQUIRE) == epoch;
        AC_REQUIRE(seen, AC_EHIP, "bert_encode_cls_unpad: the packing kernel's report never arrived");
    }
    g_unpad_wait_ns.store(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - wait_t0).count(),
                          std::memory_order_relaxed);
    const int total = vs[0], not_prefix = vs[1], longest = vs[2];
    if (total_tokens) *total_tokens = total;
    if (!not_prefix && total < b * S) {
        AC_REQUIRE(total >= b && longest >= 1 && longest <= S, AC_EHIP, "bert_encode_cls_unpad: bad report (%d rows, longest %d)", total, longest);
        return bert_encode_impl(cfg, w, d_ids, d_type_ids, nullptr, b, S, cu, src, total, longest, d_out, ldo, d_ws, ws_bytes, stream,
                                kPreCtl | (early ? kPreEmbed : 0));
    }
    // nothing to leave out (every row full: no mask needed), or a mask whose ones are not a prefix of its row: the [b, S] forward
    if (path) *path = not_prefix ? AC_BERT_PATH_PADDED_MASK : AC_BERT_PATH_PADDED;
    if (total_tokens) *total_tokens = b * S;
    return bert_encode_impl(cfg, w, d_ids, d_type_ids, not_prefix ? d_mask : nullptr, b, S, nullptr, nullptr, b * S, S, d_out, ldo, d_ws,
                            ws_bytes, stream);
}

extern "C" int ac_bert_encode_cls_packed(const ac_bert_config* cfg, const ac_bert_weights* w, const int64_t* d_ids,
                                         const int64_t* d_type_ids, int b, int S, const int32_t* d_cu,
                                         const int32_t* d_tok_src, int total_tokens, int longest, float* d_out, int64_t ldo,
                                         void* d_ws, size_t ws_bytes, ac_stream_t stream_) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (b == 0) return AC_OK;
    const ac::CallScope scope(cfg->gemm_arith_opt, cfg->ln_fusion_opt, cfg->one_launch_opt);
    AC_REQUIRE(w && d_ids && d_out && d_cu && d_tok_src && b > 0 && S >= 1 && S <= cfg->max_pos && ldo >= cfg->hidden &&
                   total_tokens >= b && total_tokens <= b * S && longest >= 1 && longest <= S,
               AC_EINVAL, "bert_encode_cls_packed: bad arguments (b=%d S=%d tokens=%d longest=%d)", b, S, total_tokens, longest);
    return bert_encode_impl(cfg, w, d_ids, d_type_ids, nullptr, b, S, d_cu, d_tok_src, total_tokens, longest, d_out, ldo, d_ws,
                            ws_bytes, (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------------------------
// ModernBERT
// ------------------------------------------------------------------------------------------------
namespace {

struct MbWs {
    size_t x, y, xn, qkv, ctx, u, g, xnp, ctxp, gp, total;
};
MbWs mb_ws(const ac_modernbert_config& c, int b, int S) {
    MbWs w;
    const size_t T = (size_t)b * S, H = c.hidden, I = c.intermediate;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += ac::align_up(n * sizeof(float), 256); return o; };
    auto take16 = [&](size_t n) { size_t o = off; off += ac::align_up(n * sizeof(uint16_t), 256); return o; };
    w.x = take(T * H); w.y = take(T * H); w.xn = take(T * H);
    w.qkv = take(T * 3 * H); w.ctx = take(T * H);
    w.u = take(T * 2 * I); w.g = take(T * I);
    w.xnp = take16(3 * T * H); w.ctxp = take16(3 * T * H); w.gp = take16(3 * T * I);
    w.total = off;
    return w;
}

int check_mb(const ac_modernbert_config* c) {
    AC_REQUIRE(c != nullptr, AC_EINVAL, "modernbert: config is NULL");
    AC_REQUIRE(c->hidden >= 64 && c->layers >= 1 && c->heads >= 1 && c->intermediate >= 8 && c->global_every >= 1 &&
                   c->local_window >= 0 && c->max_pos >= 1,
               AC_EINVAL, "modernbert: bad config");
    AC_REQUIRE(c->hidden % 4 == 0 && c->hidden <= 64 * 4 * kMaxVec && c->intermediate % 8 == 0, AC_EUNSUPPORTED,
               "modernbert: hidden=%d / intermediate=%d unsupported", c->hidden, c->intermediate);
    AC_REQUIRE(c->hidden == c->heads * DH, AC_EUNSUPPORTED, "modernbert: head dim %d unsupported (only %d)",
               c->hidden / c->heads, DH);
    AC_REQUIRE(c->gemm_arith_opt >= 0 && c->gemm_arith_opt <= 3, AC_EINVAL, "modernbert: gemm_arith_opt %d out of range", c->gemm_arith_opt);
    return AC_OK;
}

}  // namespace

extern "C" int ac_modernbert_workspace(const ac_modernbert_config* cfg, int b, int S, size_t* bytes) {
    int rc = check_mb(cfg);
    if (rc) return rc;
    AC_REQUIRE(bytes && b >= 0 && S >= 1, AC_EINVAL, "modernbert workspace: bad arguments");
    *bytes = mb_ws(*cfg, b > 0 ? b : 1, S).total;
    return AC_OK;
}

namespace {

// shared body of ac_modernbert_encode_cls (cu == nullptr: the [b, S] rows incl. padding, T = b * S) and
// ac_modernbert_encode_cls_packed (cu / tok_src from ac_bert_pack: T real-token rows, no mask)
int modernbert_encode_impl(const ac_modernbert_config* cfg, const ac_modernbert_weights* w, const int64_t* d_ids,
                           const int64_t* d_mask, int b, int S, const int32_t* cu, const int32_t* tok_src, int T, int Smax,
                           float* d_out, int64_t ldo, void* d_ws, size_t ws_bytes, hipStream_t stream) {
    int rc;
    AC_REQUIRE(w->tok_emb && w->emb_norm_g && w->final_norm_g && w->zero_bias && w->rope_cos_global &&
                   w->rope_sin_global && w->rope_cos_local && w->rope_sin_local && w->attn_norm_g && w->wqkv && w->wo &&
                   w->mlp_norm_g && w->wi && w->wo2,
               AC_EINVAL, "modernbert_encode_cls: missing weights");
    const ac_modernbert_config& c = *cfg;
    const MbWs ws = mb_ws(c, b, S);
    AC_REQUIRE(d_ws && ws_bytes >= ws.total, AC_EWORKSPACE, "modernbert_encode_cls: workspace %zu < %zu", ws_bytes, ws.total);
    char* base = (char*)d_ws;
    float* x = (float*)(base + ws.x);      // residual stream (ping-pongs with y)
    float* y = (float*)(base + ws.y);
    float* xn = (float*)(base + ws.xn);    // normed input of the current GEMM
    float* qkv = (float*)(base + ws.qkv);
    float* ctx = (float*)(base + ws.ctx);
    float* u = (float*)(base + ws.u);
    float* g = (float*)(base + ws.g);
    uint16_t* xnp = (uint16_t*)(base + ws.xnp);
    uint16_t* ctxp = (uint16_t*)(base + ws.ctxp);
    uint16_t* gp = (uint16_t*)(base + ws.gp);
    const int H = c.hidden, I = c.intermediate;
    const int tok_blocks = (T + 3) / 4;
    const float* zb = w->zero_bias;
    auto opt = [&](const float* const* arr, int l) -> const float* { return (arr && arr[l]) ? arr[l] : zb; };
    const bool wplanes = w->wqkv3 && w->wo3 && w->wi3 && w->wo23;
    const bool pl = wplanes && ac::linear_takes_planes(T, H, H) && ac::linear_takes_planes(T, H, I) && (H % 8) == 0;

    // embeddings -> LayerNorm: this IS the input of layer 0's attention (attn_norm = Identity there)
    hipLaunchKernelGGL(embed_ln_kernel<false>, dim3(tok_blocks), dim3(256), 0, stream, d_ids, (const int64_t*)nullptr, T, S, H,
                       w->tok_emb, (const float*)nullptr, (const float*)nullptr, w->emb_norm_g,
                       w->emb_norm_b ? w->emb_norm_b : zb, c.norm_eps, x, pl ? xnp : nullptr, tok_src);
    AC_LAUNCH_CHECK();
    const float scale = 1.0f / sqrtf((float)DH);
    for (int l = 0; l < c.layers; ++l) {
        const float* a_in = x;             // attention input rows (fp32) -- x itself for layer 0
        if (l > 0) {
            AC_REQUIRE(w->attn_norm_g[l] != nullptr, AC_EINVAL, "modernbert: attn_norm weight of layer %d is NULL", l);
            hipLaunchKernelGGL(ln_kernel_t<false>, dim3(tok_blocks), dim3(256), 0, stream, x, T, H, w->attn_norm_g[l],
                               opt(w->attn_norm_b, l), c.norm_eps, xn, pl ? xnp : nullptr, (int64_t)H);
            AC_LAUNCH_CHECK();
            a_in = xn;
        }
        rc = ac::linear_f32(a_in, H, w->wqkv[l], H, opt(w->wqkv_b, l), nullptr, 0, qkv, 3 * H, T, 3 * H, H, 0, nullptr,
                            1.f, stream, 0.f, 0, wplanes ? w->wqkv3[l] : nullptr, pl ? xnp : nullptr);
        if (rc) return rc;
        const bool global = (l % c.global_every) == 0;
        const float* rc_ = global ? w->rope_cos_global : w->rope_cos_local;
        const float* rs_ = global ? w->rope_sin_global : w->rope_sin_local;
        const int win = global ? -1 : c.local_window;
        const bool last = (l == c.layers - 1);
        // After the last layer's attention only the CLS row of each sequence is consumed (classifier.py:1272):
        // the CLS-query attention, the output projection, the MLP and the final norm run on b rows, not b*S.
        const int Ml = last ? b : T;
        const bool lp = last ? (wplanes && ac::linear_takes_planes(b, H, H) && ac::linear_takes_planes(b, H, I) && (H % 8) == 0) : pl;
        const float* resid = x;
        int64_t ldres = last ? (int64_t)S * H : H;
        if (last) {
            hipLaunchKernelGGL((attention_cls_kernel<true, 64>), dim3(c.heads, b), dim3(64), 0, stream, qkv, d_mask, S, H, scale, ctx,
                               rc_, rs_, win, cu);
            if (cu) {                                  // packed layout: the CLS rows sit at cu[s]; gather them (g is free until GeGLU)
                AC_LAUNCH_CHECK();
                hipLaunchKernelGGL(gather_rows_kernel, dim3((b + 3) / 4), dim3(256), 0, stream, x, cu, b, H, g);
                resid = g;
                ldres = H;
            }
        } else {
            hipLaunchKernelGGL((attention_mfma_kernel<true, 64>), dim3((Smax + 31) / 32, c.heads, b), dim3(64), 0, stream, qkv, d_mask,
                               S, H, scale, ctx, pl ? ctxp : nullptr, rc_, rs_, win, cu, (int64_t)T);
        }
        AC_LAUNCH_CHECK();
        // y = x + ctx Wo^T   (last layer: ctx is b compact rows, the residual rows of x are S*H apart / gathered)
        rc = ac::linear_f32(ctx, H, w->wo[l], H, opt(w->wo_b, l), resid, ldres, y, H, Ml, H, H, 0, nullptr, 1.f,
                            stream, 0.f, 0, wplanes ? w->wo3[l] : nullptr, (pl && !last) ? ctxp : nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL(ln_kernel_t<false>, dim3((Ml + 3) / 4), dim3(256), 0, stream, y, Ml, H, w->mlp_norm_g[l],
                           opt(w->mlp_norm_b, l), c.norm_eps, xn, lp ? xnp : nullptr, (int64_t)H);
        AC_LAUNCH_CHECK();
        if (lp && w->wi_interleaved32) {
            // GeGLU fused into the Wi GEMM's epilogue, result straight into the operand planes of the Wo2 GEMM
            rc = ac::linear_f32(xn, H, w->wi[l], H, opt(w->wi_b, l), nullptr, 0, nullptr, I, Ml, 2 * I, H, 3, nullptr, 1.f,
                                stream, 0.f, 0, w->wi3[l], xnp, gp);
            if (rc) return rc;
        } else {
            rc = ac::linear_f32(xn, H, w->wi[l], H, opt(w->wi_b, l), nullptr, 0, u, 2 * I, Ml, 2 * I, H, 0, nullptr, 1.f,
                                stream, 0.f, 0, wplanes ? w->wi3[l] : nullptr, lp ? xnp : nullptr);
            if (rc) return rc;
            const int64_t units = (int64_t)Ml * (I / 8);
            hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, stream, u, (int64_t)Ml, I, g,
                               lp ? gp : nullptr, w->wi_interleaved32);
            AC_LAUNCH_CHECK();
        }
        // x = y + g Wo2^T   (last layer: b compact rows)
        rc = ac::linear_f32(g, I, w->wo2[l], I, opt(w->wo2_b, l), y, H, x, H, Ml, H, I, 0, nullptr, 1.f, stream, 0.f, 0,
                            wplanes ? w->wo23[l] : nullptr, lp ? gp : nullptr);
        if (rc) return rc;
    }
    // after the CLS-only last layer x holds b compact rows: final LayerNorm, then L2-normalise
    hipLaunchKernelGGL(ln_kernel_t<false>, dim3((b + 3) / 4), dim3(256), 0, stream, x, b, H, w->final_norm_g,
                       w->final_norm_b ? w->final_norm_b : zb, c.norm_eps, xn, (uint16_t*)nullptr, (int64_t)H);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(cls_normalize_kernel, dim3((b + 3) / 4), dim3(256), 0, stream, xn, b, 1, H, d_out, ldo);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

}  // namespace

extern "C" int ac_modernbert_encode_cls(const ac_modernbert_config* cfg, const ac_modernbert_weights* w,
                                        const int64_t* d_ids, const int64_t* d_mask, int b, int S, float* d_out,
                                        int64_t ldo, void* d_ws, size_t ws_bytes, ac_stream_t stream_) {
    int rc = check_mb(cfg);
    if (rc) return rc;
    if (b == 0) return AC_OK;
    const ac::CallScope scope(cfg->gemm_arith_opt, 0, 0);
    AC_REQUIRE(w && d_ids && d_out && b > 0 && S >= 1 && S <= cfg->max_pos && ldo >= cfg->hidden, AC_EINVAL,
               "modernbert_encode_cls: bad arguments (b=%d S=%d max_pos=%d)", b, S, cfg->max_pos);
    return modernbert_encode_impl(cfg, w, d_ids, d_mask, b, S, nullptr, nullptr, b * S, S, d_out, ldo, d_ws, ws_bytes,
                                  (hipStream_t)stream_);
}

extern "C" int ac_modernbert_encode_cls_packed(const ac_modernbert_config* cfg, const ac_modernbert_weights* w,
                                               const int64_t* d_ids, int b, int S, const int32_t* d_cu,
                                               const int32_t* d_tok_src, int total_tokens, int longest, float* d_out,
                                               int64_t ldo, void* d_ws, size_t ws_bytes, ac_stream_t stream_) {
    int rc = check_mb(cfg);
    if (rc) return rc;
    if (b == 0) return AC_OK;
    const ac::CallScope scope(cfg->gemm_arith_opt, 0, 0);
    AC_REQUIRE(w && d_ids && d_out && d_cu && d_tok_src && b > 0 && S >= 1 && S <= cfg->max_pos && ldo >= cfg->hidden &&
                   total_tokens >= b && total_tokens <= b * S && longest >= 1 && longest <= S,
               AC_EINVAL, "modernbert_encode_cls_packed: bad arguments (b=%d S=%d tokens=%d longest=%d)", b, S, total_tokens, longest);
    return modernbert_encode_impl(cfg, w, d_ids, nullptr, b, S, d_cu, d_tok_src, total_tokens, longest, d_out, ldo, d_ws, ws_bytes,
                                  (hipStream_t)stream_);
}
