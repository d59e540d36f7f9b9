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

#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxVec = 8;    // float4 per lane held in registers: H <= 64 * 4 * 8 = 2048

// ---- wave-per-token LayerNorm helpers ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void layernorm_store(f32x4 (&x)[kMaxVec], int H, int lane, const float* g,
                                                const float* b, float eps, float* dst) {
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
        }
    }
}

// word + position + token_type embeddings -> LayerNorm (modeling_bert.py:85-107)
__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* ids, const int64_t* type_ids, int T, int S,
                                                       int H, const float* word, const float* pos,
                                                       const float* type, const float* g, const float* b,
                                                       float eps, float* out) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int64_t id = ids[t];
    const int64_t tt = type_ids ? type_ids[t] : 0;
    const int p = t % S;
    const int nv = H >> 2;
    f32x4 x[kMaxVec];
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e) {
        const int c4 = lane + 64 * e;
        if (c4 < nv) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(word + id * H + 4 * c4);
            const f32x4 ty = *reinterpret_cast<const f32x4*>(type + tt * H + 4 * c4);
            const f32x4 po = *reinterpret_cast<const f32x4*>(pos + (int64_t)p * H + 4 * c4);
            x[e] = (w + ty) + po;     // inputs_embeds + token_type_embeddings, then + position
        }
    }
    layernorm_store(x, H, lane, g, b, eps, out + (int64_t)t * H);
}

__global__ __launch_bounds__(256) void ln_kernel(const float* in, int T, int H, const float* g, const float* b,
                                                 float eps, float* out) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const int nv = H >> 2;
    f32x4 x[kMaxVec];
#pragma unroll
    for (int e = 0; e < kMaxVec; ++e) {
        const int c4 = lane + 64 * e;
        if (c4 < nv) x[e] = *reinterpret_cast<const f32x4*>(in + (int64_t)t * H + 4 * c4);
    }
    layernorm_store(x, H, lane, g, b, eps, out + (int64_t)t * H);
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

// ---- fused attention, head dim 64, fp32 ----
// grid = (ceil(S/64), heads, batch), block = one wave.  Lane = one query row.
constexpr int DH = 64;
constexpr int KT = 64;        // keys per LDS tile
constexpr int CH = 16;        // keys per online-softmax chunk

__global__ __launch_bounds__(64) void attention_kernel(const float* qkv, const int64_t* mask, int S, int H,
                                                       float scale, float* ctx) {
    __shared__ __attribute__((aligned(16))) float Ks[KT][DH];
    __shared__ __attribute__((aligned(16))) float Vs[KT][DH];
    __shared__ int valid_s[KT];
    const int lane = threadIdx.x;
    const int qt = blockIdx.x, head = blockIdx.y, bi = blockIdx.z;
    const int64_t ld = 3 * (int64_t)H;
    const float* base = qkv + (int64_t)bi * S * ld + head * DH;
    const int qi = qt * 64 + lane;
    const bool qvalid = qi < S;

    f32x4 q[DH / 4], o[DH / 4];
    {
        const float* qp = base + (int64_t)(qvalid ? qi : S - 1) * ld;
#pragma unroll
        for (int d = 0; d < DH / 4; ++d) {
            q[d] = *reinterpret_cast<const f32x4*>(qp + 4 * d) * scale;
            o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    float m = -INFINITY, l = 0.f;

    for (int k0 = 0; k0 < S; k0 += KT) {
        const int nk = (S - k0) < KT ? (S - k0) : KT;
        __syncthreads();
        // stage K/V tile: 16 lanes per key row (256 B contiguous), 4 keys per pass
        for (int r = lane >> 4; r < KT; r += 4) {
            const int c = (lane & 15) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (r < nk) {
                const float* kp = base + (int64_t)(k0 + r) * ld + H;
                kv = *reinterpret_cast<const f32x4*>(kp + c);
                vv = *reinterpret_cast<const f32x4*>(kp + H + c);
            }
            *reinterpret_cast<f32x4*>(&Ks[r][c]) = kv;
            *reinterpret_cast<f32x4*>(&Vs[r][c]) = vv;
        }
        if (lane < KT) valid_s[lane] = (lane < nk) && (!mask || mask[(int64_t)bi * S + k0 + lane] != 0);
        __syncthreads();

        for (int c0 = 0; c0 < nk; c0 += CH) {
            float s[CH];
            float cmax = -INFINITY;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                float acc = 0.f;
                const f32x4* kr = reinterpret_cast<const f32x4*>(&Ks[c0 + j][0]);   // uniform address: broadcast
#pragma unroll
                for (int d = 0; d < DH / 4; ++d) {
                    const f32x4 kk = kr[d];
                    acc = fmaf(q[d].x, kk.x, acc); acc = fmaf(q[d].y, kk.y, acc);
                    acc = fmaf(q[d].z, kk.z, acc); acc = fmaf(q[d].w, kk.w, acc);
                }
                s[j] = (c0 + j < KT && valid_s[c0 + j]) ? acc : -INFINITY;   // additive -inf mask
                cmax = fmaxf(cmax, s[j]);
            }
            const float m_new = fmaxf(m, cmax);
            if (m_new == -INFINITY) continue;      // wave-uniform in practice (mask is per key)
            const float corr = expf(m - m_new);    // m = -inf -> 0
            l *= corr;
#pragma unroll
            for (int d = 0; d < DH / 4; ++d) o[d] *= corr;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float p = expf(s[j] - m_new);
                l += p;
                const f32x4* vr = reinterpret_cast<const f32x4*>(&Vs[c0 + j][0]);
#pragma unroll
                for (int d = 0; d < DH / 4; ++d) {
                    const f32x4 vv = vr[d];
                    o[d].x = fmaf(p, vv.x, o[d].x); o[d].y = fmaf(p, vv.y, o[d].y);
                    o[d].z = fmaf(p, vv.z, o[d].z); o[d].w = fmaf(p, vv.w, o[d].w);
                }
            }
            m = m_new;
        }
    }
    if (qvalid) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        float* dst = ctx + ((int64_t)bi * S + qi) * H + head * DH;
#pragma unroll
        for (int d = 0; d < DH / 4; ++d) *reinterpret_cast<f32x4*>(dst + 4 * d) = o[d] * inv;
    }
}

// Last layer: only the CLS query of every sequence feeds the output (classifier.py:1272), so its
// attention is a single-query problem.  One wave per (sequence, head): lane = key for the scores,
// lane = output dim for the P.V reduction; K tile rows padded to 65 floats (conflict-free column reads).
__global__ __launch_bounds__(64) void attention_cls_kernel(const float* qkv, const int64_t* mask, int S, int H,
                                                           float scale, float* ctx_cls) {
    __shared__ float Ks[KT][DH + 1];
    __shared__ __attribute__((aligned(16))) float Vs[KT][DH];
    __shared__ float qs[DH];
    __shared__ float ps[KT];
    const int lane = threadIdx.x;
    const int head = blockIdx.x, bi = blockIdx.y;
    const int64_t ld = 3 * (int64_t)H;
    const float* base = qkv + (int64_t)bi * S * ld + head * DH;
    qs[lane] = base[lane] * scale;              // CLS token = row 0 of the sequence
    float m = -INFINITY, l = 0.f, o = 0.f;      // o: output dim `lane`
    for (int k0 = 0; k0 < S; k0 += KT) {
        const int nk = (S - k0) < KT ? (S - k0) : KT;
        __syncthreads();
        for (int r = lane >> 4; r < KT; r += 4) {
            const int c = (lane & 15) * 4;
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
        const bool valid = lane < nk && (!mask || mask[(int64_t)bi * S + k0 + lane] != 0);
        float sc = 0.f;
#pragma unroll 16
        for (int d = 0; d < DH; ++d) sc = fmaf(qs[d], Ks[lane][d], sc);
        sc = valid ? sc : -INFINITY;
        float cmax = sc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, off));
        const float m_new = fmaxf(m, cmax);
        if (m_new == -INFINITY) continue;        // wave-uniform
        const float corr = expf(m - m_new);
        const float p = expf(sc - m_new);        // masked -> 0
        ps[lane] = p;
        float psum = p;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) psum += __shfl_xor(psum, off);
        l = l * corr + psum;
        o *= corr;
        __syncthreads();
        for (int j = 0; j < nk; ++j) o = fmaf(ps[j], Vs[j][lane], o);
        m = m_new;
    }
    ctx_cls[(int64_t)bi * H + head * DH + lane] = l > 0.f ? o / l : 0.f;
}

struct BertWs {
    size_t x, qkv, ctx, y, ffn, total;
};
BertWs bert_ws(const ac_bert_config& c, int b, int S) {
    BertWs w;
    const size_t T = (size_t)b * S;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += ac::align_up(n * sizeof(float), 256); return o; };
    w.x = take(T * c.hidden);
    w.qkv = take(T * 3 * c.hidden);
    w.ctx = take(T * c.hidden);
    w.y = take(T * c.hidden);
    w.ffn = take(T * c.intermediate);
    w.total = off;
    return w;
}

int check_cfg(const ac_bert_config* c) {
    AC_REQUIRE(c != nullptr, AC_EINVAL, "bert: config is NULL");
    AC_REQUIRE(c->hidden >= 64 && c->layers >= 1 && c->heads >= 1 && c->intermediate >= 4, AC_EINVAL,
               "bert: bad config");
    AC_REQUIRE(c->hidden % 4 == 0 && c->hidden <= 64 * 4 * kMaxVec, AC_EUNSUPPORTED,
               "bert: hidden=%d unsupported (must be a multiple of 4 and <= %d)", c->hidden, 64 * 4 * kMaxVec);
    AC_REQUIRE(c->hidden == c->heads * DH, AC_EUNSUPPORTED, "bert: head dim %d unsupported (only %d)",
               c->hidden / c->heads, DH);
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

extern "C" int ac_bert_encode_cls(const ac_bert_config* cfg, const ac_bert_weights* w, const int64_t* d_ids,
                                  const int64_t* d_type_ids, const int64_t* d_mask, int b, int S,
                                  float* d_out, int64_t ldo, void* d_ws, size_t ws_bytes, ac_stream_t stream_) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (b == 0) return AC_OK;
    AC_REQUIRE(w && d_ids && d_out && b > 0 && S >= 1 && S <= cfg->max_pos && ldo >= cfg->hidden, AC_EINVAL,
               "bert_encode_cls: bad arguments (b=%d S=%d max_pos=%d)", b, S, cfg->max_pos);
    const ac_bert_config& c = *cfg;
    const BertWs ws = bert_ws(c, b, S);
    AC_REQUIRE(d_ws && ws_bytes >= ws.total, AC_EWORKSPACE, "bert_encode_cls: workspace %zu < %zu", ws_bytes, ws.total);
    char* base = (char*)d_ws;
    float* x = (float*)(base + ws.x);
    float* qkv = (float*)(base + ws.qkv);
    float* ctx = (float*)(base + ws.ctx);
    float* y = (float*)(base + ws.y);
    float* ffn = (float*)(base + ws.ffn);
    const int T = b * S, H = c.hidden, I = c.intermediate;
    const int tok_blocks = (T + 3) / 4;

    hipLaunchKernelGGL(embed_ln_kernel, dim3(tok_blocks), dim3(256), 0, stream, d_ids, d_type_ids, T, S, H,
                       w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g, w->emb_ln_b, c.ln_eps, x);
    AC_LAUNCH_CHECK();
    const float scale = 1.0f / sqrtf((float)DH);
    for (int l = 0; l < c.layers; ++l) {
        rc = ac::linear_f32(x, H, w->qkv_w[l], H, w->qkv_b[l], nullptr, 0, qkv, 3 * H, T, 3 * H, H, 0, nullptr, 1.f, stream);
        if (rc) return rc;
        const bool last = (l == c.layers - 1);
        // After the last layer's attention only the CLS row of each sequence is consumed, so the
        // output projection, both LayerNorms and the FFN run on b rows instead of b*S.
        const int Ml = last ? b : T;
        const float* resid = x;                        // residual = layer input
        const int64_t ldres = last ? (int64_t)S * H : H;   // CLS rows of x are S*H apart
        if (last) {
            hipLaunchKernelGGL(attention_cls_kernel, dim3(c.heads, b), dim3(64), 0, stream, qkv, d_mask, S, H, scale, ctx);
        } else {
            hipLaunchKernelGGL(attention_kernel, dim3((S + 63) / 64, c.heads, b), dim3(64), 0, stream, qkv, d_mask, S, H,
                               scale, ctx);
        }
        AC_LAUNCH_CHECK();
        const int lblocks = (Ml + 3) / 4;
        rc = ac::linear_f32(ctx, H, w->ao_w[l], H, w->ao_b[l], resid, ldres, y, H, Ml, H, H, 0, nullptr, 1.f, stream);
        if (rc) return rc;
        // (last layer: x is overwritten with b compact rows; its old contents are no longer needed)
        hipLaunchKernelGGL(ln_kernel, dim3(lblocks), dim3(256), 0, stream, y, Ml, H, w->ln1_g[l], w->ln1_b[l],
                           c.ln_eps, last ? ctx : x);
        AC_LAUNCH_CHECK();
        float* x1 = last ? ctx : x;                    // ctx is free again after the AO projection
        rc = ac::linear_f32(x1, H, w->ff1_w[l], H, w->ff1_b[l], nullptr, 0, ffn, I, Ml, I, H, 2, nullptr, 1.f, stream);
        if (rc) return rc;
        rc = ac::linear_f32(ffn, I, w->ff2_w[l], I, w->ff2_b[l], x1, H, y, H, Ml, H, I, 0, nullptr, 1.f, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(ln_kernel, dim3(lblocks), dim3(256), 0, stream, y, Ml, H, w->ln2_g[l], w->ln2_b[l],
                           c.ln_eps, x);
        AC_LAUNCH_CHECK();
    }
    // after the CLS-only last layer x holds b compact rows (sequence stride 1)
    hipLaunchKernelGGL(cls_normalize_kernel, dim3((b + 3) / 4), dim3(256), 0, stream, x, b, 1, H, d_out, ldo);
    AC_LAUNCH_CHECK();
    return AC_OK;
}
