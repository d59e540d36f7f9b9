// AdaptiveHead (models.py:30-98), EWC (ewc.py:39-116) and the AdamW training step
// (classifier.py:1463-1505 and :307-351) as HIP kernels.
//
// The head's six tensors live in one flat fp32 block (see acamd.h) so the optimizer step,
// the Fisher accumulation and the EWC penalty are single launches over P = ~0.9 M elements.
// The GEMMs run on the fp32 MFMA pipe (gemm.hip); with B = 32 the step is latency bound, so
// everything else is fused into GEMM epilogues: bias+ReLU+dropout (forward), ReLU/dropout
// backward gate (backward), softmax+CE+dlogits in one small kernel, bias gradients in one.
#include "common.h"

#include <math.h>

namespace {

struct HeadOffsets {
    int64_t w1, b1, w2, b2, w3, b3, total;
};

HeadOffsets head_offsets(const ac_head_dims& d) {
    HeadOffsets o;
    o.w1 = 0;
    o.b1 = o.w1 + (int64_t)d.H1 * d.D;
    o.w2 = o.b1 + d.H1;
    o.b2 = o.w2 + (int64_t)d.H2 * d.H1;
    o.w3 = o.b2 + d.H2;
    o.b3 = o.w3 + (int64_t)d.C * d.H2;
    o.total = o.b3 + d.C;
    return o;
}

struct HeadWs {
    size_t a1, a2, z, dz, d2, d1, rowloss, xg, yg, tg, scratch, epoch, total;
};

HeadWs head_ws(const ac_head_dims& d, int B) {
    HeadWs w;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += ac::align_up(n * sizeof(float), 256); return o; };
    w.a1 = take((size_t)B * d.H1);
    w.a2 = take((size_t)B * d.H2);
    w.z = take((size_t)B * d.C);
    w.dz = take((size_t)B * d.C);
    w.d2 = take((size_t)B * d.H2);
    w.d1 = take((size_t)B * d.H1);
    w.rowloss = take((size_t)B);
    w.xg = take((size_t)B * d.D);          // gathered batch (ac_head_train_step with an index)
    w.yg = take((size_t)B * 2);            // int64 labels
    w.tg = take((size_t)B * d.C);          // gathered multi-hot targets (BCE)
    w.scratch = take(AC_REDUCE_SCRATCH_BYTES / sizeof(float));
    w.epoch = take(ac::head_epoch_ws_bytes(d.H1, d.H2) / sizeof(float));      // persistent epoch: a1 / a2 exchange, partials, barrier
    w.total = off;
    return w;
}

// loss + dlogits, one wave per row, one block.  kind:
//   AC_LOSS_CE          nn.CrossEntropyLoss on logits (classifier.py:1463,1498):
//                       loss = mean_b(logsumexp(z_b) - z_b[y_b]);  dz = (softmax(z) - onehot(y)) / B
//   AC_LOSS_BCE_SIGMOID nn.BCELoss on sigmoid(z) against multi-hot targets T (multilabel.py:41-43,361,380):
//                       loss = mean_{b,c} -(t log p + (1-t) log(1-p)), logs clamped at -100 like torch;
//                       dz = (p - t) / max(p(1-p), 1e-12) / (B C) * p(1-p)
//   AC_LOSS_CE_SIGMOID  the reference's new-class path run on a multi-label head (classifier.py:337-339 with
//                       multilabel.py:41-43): CrossEntropyLoss applied to sigmoid(z);
//                       dz = (softmax(p) - onehot(y)) / B * p(1-p)
__global__ __launch_bounds__(256) void loss_fwd_bwd_kernel(const float* z, const int64_t* y, const float* T, int64_t ldt,
                                                           int B, int C, int kind, float* dz, float* rowloss,
                                                           float* loss) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < B; b += 4) {
        const float* zr = z + (size_t)b * C;
        if (kind == AC_LOSS_BCE_SIGMOID) {
            const float inv = 1.f / ((float)B * (float)C);
            float s = 0.f;
            for (int c = lane; c < C; c += 64) {
                const float p = 1.f / (1.f + expf(-zr[c]));
                const float t = T[(size_t)b * ldt + c];
                s -= t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f);
                const float pq = p * (1.f - p);
                dz[(size_t)b * C + c] = (p - t) / fmaxf(pq, 1e-12f) * inv * pq;
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
            if (lane == 0) rowloss[b] = s / (float)C;
            continue;
        }
        const bool sig = kind == AC_LOSS_CE_SIGMOID;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) {
            const float v = sig ? 1.f / (1.f + expf(-zr[c])) : zr[c];
            mx = fmaxf(mx, v);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = sig ? 1.f / (1.f + expf(-zr[c])) : zr[c];
            sum += expf(v - mx);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
        const int64_t yb = y[b];
        const float invB = 1.f / (float)B;
        for (int c = lane; c < C; c += 64) {
            const float v = sig ? 1.f / (1.f + expf(-zr[c])) : zr[c];
            float g = (expf(v - mx) / sum - (c == yb ? 1.f : 0.f)) * invB;
            if (sig) g *= v * (1.f - v);
            dz[(size_t)b * C + c] = g;
        }
        if (lane == 0) {
            const float vy = sig ? 1.f / (1.f + expf(-zr[yb])) : zr[yb];
            rowloss[b] = (mx + logf(sum)) - vy;
        }
    }
    __syncthreads();
    if (wave == 0) {
        float s = 0.f;
        for (int b = lane; b < B; b += 64) s += rowloss[b];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
        if (lane == 0) *loss = s / (float)B;
    }
}

// torch.sigmoid over n elements (multilabel.py:43)
__global__ __launch_bounds__(256) void sigmoid_kernel(const float* in, int64_t n, float* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = 1.f / (1.f + expf(-in[i]));
}

// bias gradients: column sums of dz [B,C], d2 [B,H2], d1 [B,H1] into the flat grad block
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* dz, const float* d2, const float* d1,
                                                        int B, int C, int H2, int H1, float* gb3, float* gb2,
                                                        float* gb1) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float* src; float* dst; int n, c;
    if (t < H1) { src = d1; dst = gb1; n = H1; c = t; }
    else if (t < H1 + H2) { src = d2; dst = gb2; n = H2; c = t - H1; }
    else if (t < H1 + H2 + C) { src = dz; dst = gb3; n = C; c = t - H1 - H2; }
    else return;
    if (!dst) return;                            // already produced by head_top_kernel
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += src[(size_t)b * n + c];
    dst[c] = s;
}

// The top of the network in ONE launch when there are few classes (C <= kTopMaxC): logits z = a2 W3^T + b3,
// the loss and dz (same formulas as loss_fwd_bwd_kernel), gW3 = dz^T a2, gb3, d2 = (dz W3) gated by layer 2's
// relu/dropout, gb2 -- everything above the second hidden layer is O(B * H2 * C) work, i.e. five launches'
// worth of latency for a few microseconds of arithmetic.  One workgroup; thread k owns column k of a2 / W3.
constexpr int kTopMaxC = 16;
constexpr int kTopMaxB = 256;

// STAGE: a2 [B, H2] and W3 [C, H2] are first copied into LDS with coalesced 16-byte loads (the training batch: 32 x 384
// + 4 x 384 floats = 54 KB); the dot products of steps 1 and 3 then read LDS instead of chasing L2 latencies.  Same
// arithmetic, same order: bit-identical to the unstaged form.
template <bool STAGE>
__global__ __launch_bounds__(512) void head_top_kernel(const float* __restrict__ a2g, int H2, const float* __restrict__ W3g,
                                                       const float* __restrict__ b3, const int64_t* __restrict__ y,
                                                       const float* __restrict__ T, int64_t ldt, int B, int C, int kind,
                                                       float gate_scale, float* __restrict__ d2, float* __restrict__ gW3,
                                                       float* __restrict__ gb3, float* __restrict__ gb2,
                                                       float* __restrict__ loss) {
    __shared__ float zs[kTopMaxB][kTopMaxC];
    __shared__ float dzs[kTopMaxB][kTopMaxC];
    __shared__ float rl[kTopMaxB];
    extern __shared__ __attribute__((aligned(16))) float stage_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* a2 = a2g;
    const float* W3 = W3g;
    if (STAGE) {                                           // (H2 % 4 == 0 and 16-byte aligned sources: checked at launch)
        float* sa = stage_lds;
        float* sw = stage_lds + (size_t)B * H2;
        const int na4 = B * H2 / 4, nw4 = C * H2 / 4;
        for (int t = tid; t < na4; t += 512) reinterpret_cast<float4*>(sa)[t] = reinterpret_cast<const float4*>(a2g)[t];
        for (int t = tid; t < nw4; t += 512) reinterpret_cast<float4*>(sw)[t] = reinterpret_cast<const float4*>(W3g)[t];
        __syncthreads();
        a2 = sa; W3 = sw;
    }
    // 1. logits: four threads per (row, class) pair, each a strided quarter of the dot product, so the usual
    //    B * C = 128 pairs finish in one pass of the 512 threads with every load independent
    for (int p0 = 0; p0 < B * C; p0 += 128) {
        const int p = p0 + (tid >> 2), q = tid & 3;
        float acc = 0.f;
        if (p < B * C) {
            const int b = p / C, c = p - b * C;
            const float* ar = a2 + (size_t)b * H2;
            const float* wr = W3 + (size_t)c * H2;
#pragma unroll 8
            for (int k = q; k < H2; k += 4) acc = fmaf(ar[k], wr[k], acc);
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (p < B * C && q == 0) { const int b = p / C, c = p - b * C; zs[b][c] = acc + b3[c]; }
    }
    __syncthreads();
    // 2. loss and dz, one thread per row
    if (tid < B) {
        const int b = tid;
        if (kind == AC_LOSS_BCE_SIGMOID) {
            const float inv = 1.f / ((float)B * (float)C);
            float s = 0.f;
            for (int c = 0; c < C; ++c) {
                const float p = 1.f / (1.f + expf(-zs[b][c]));
                const float t = T[(size_t)b * ldt + c];
                s -= t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f);
                const float pq = p * (1.f - p);
                dzs[b][c] = (p - t) / fmaxf(pq, 1e-12f) * inv * pq;
            }
            rl[b] = s / (float)C;
        } else {
            const bool sig = kind == AC_LOSS_CE_SIGMOID;
            float v[kTopMaxC];
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < kTopMaxC; ++c)
                if (c < C) { v[c] = sig ? 1.f / (1.f + expf(-zs[b][c])) : zs[b][c]; mx = fmaxf(mx, v[c]); }
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < kTopMaxC; ++c) if (c < C) sum += expf(v[c] - mx);
            const int yb = (int)y[b];
            const float invB = 1.f / (float)B;
            float vy = 0.f;
#pragma unroll
            for (int c = 0; c < kTopMaxC; ++c)
                if (c < C) {
                    float g = (expf(v[c] - mx) / sum - (c == yb ? 1.f : 0.f)) * invB;
                    if (sig) g *= v[c] * (1.f - v[c]);
                    dzs[b][c] = g;
                    if (c == yb) vy = v[c];
                }
            rl[b] = (mx + logf(sum)) - vy;
        }
    }
    __syncthreads();
    if (wave == 0) {
        float s = 0.f;
        for (int b = lane; b < B; b += 64) s += rl[b];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
        if (lane == 0) *loss = s / (float)B;
    }
    if (tid >= 64 && tid < 64 + C) {                       // gb3 = column sums of dz, rows in order
        const int c = tid - 64;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dzs[b][c];
        gb3[c] = s;
    }
    // 3. thread k: column k of a2 / W3 -> gW3[:, k], d2[:, k], gb2[k]
    for (int k = tid; k < H2; k += 512) {
        float w3[kTopMaxC], dw[kTopMaxC];
#pragma unroll
        for (int c = 0; c < kTopMaxC; ++c) { w3[c] = c < C ? W3[(size_t)c * H2 + k] : 0.f; dw[c] = 0.f; }
        float db = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) {
            const float a = a2[(size_t)b * H2 + k];
            float t = 0.f;
#pragma unroll
            for (int c = 0; c < kTopMaxC; ++c)
                if (c < C) { const float g = dzs[b][c]; t = fmaf(g, w3[c], t); dw[c] = fmaf(g, a, dw[c]); }
            const float d = (a != 0.f) ? t * gate_scale : 0.f;
            d2[(size_t)b * H2 + k] = d;
            db += d;
        }
#pragma unroll
        for (int c = 0; c < kTopMaxC; ++c) if (c < C) gW3[(size_t)c * H2 + k] = dw[c];
        gb2[k] = db;
    }
}

__global__ __launch_bounds__(256) void fisher_acc_kernel(const float* g, float inv, float* F, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        F[i] += gi * gi * inv;      // ewc.py:92  fisher += grad**2 / len(loader)
    }
}

// batch gather: xg[b,:] = X[idx[b],:], yg[b] = y[idx[b]]  (DataLoader batch assembly, classifier.py:1485-1487)
__global__ __launch_bounds__(256) void gather_batch_kernel(const float* X, int64_t ldx, const int64_t* y,
                                                           const float* T, int64_t ldt, int C, const int64_t* idx,
                                                           int B, int D, float* xg, int64_t* yg, float* tg) {
    const int b = blockIdx.x;
    const int64_t r = idx[b];
    for (int c = threadIdx.x; c < D; c += 256) xg[(size_t)b * D + c] = X[r * ldx + c];
    if (T) for (int c = threadIdx.x; c < C; c += 256) tg[(size_t)b * C + c] = T[r * ldt + c];
    if (threadIdx.x == 0 && y) yg[b] = y[r];
}

// F.softmax(logits, dim=1) (classifier.py:435,1345), one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* in, int B, int C, float* out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* zr = in + (size_t)b * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, zr[c]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(zr[c] - mx);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
    for (int c = lane; c < C; c += 64) out[(size_t)b * C + c] = expf(zr[c] - mx) / sum;
}

// F.normalize(x, p=2, dim=1) with eps 1e-12 (classifier.py:1450), one wave per row
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* in, int64_t ldi, int B, int D, float* out,
                                                                int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* src = in + (size_t)b * ldi;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s = fmaf(src[c], src[c], s);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    const float nrm = fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < D; c += 64) out[(size_t)b * ldo + c] = src[c] / nrm;
}

// ---- deterministic two-pass reductions over the flat parameter block ----
constexpr int kRedBlocks = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    const float r = ((sh[0] + sh[1]) + (sh[2] + sh[3]));
    __syncthreads();
    return r;
}

// pass A: per-block partial sums of  sum g_tot^2  and  sum F (p - p*)^2
__global__ __launch_bounds__(256) void ewc_partials_kernel(const float* p, const float* g, const float* F,
                                                           const float* pold, int64_t n, float two_lam,
                                                           float* partials) {
    __shared__ float sh[4];
    float sg = 0.f, se = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)kRedBlocks * 256) {
        float gi = g ? g[i] : 0.f;
        if (F) {
            const float d = p[i] - pold[i];
            const float fd = F[i] * d;
            se = fmaf(fd, d, se);
            gi = fmaf(two_lam, fd, gi);       // d/dp [ lam * F (p-p*)^2 ] = 2 lam F (p-p*)
        }
        sg = fmaf(gi, gi, sg);
    }
    const float tg = block_sum(sg, sh);
    const float te = block_sum(se, sh);
    if (threadIdx.x == 0) { partials[blockIdx.x] = tg; partials[kRedBlocks + blockIdx.x] = te; }
}

__device__ __forceinline__ void reduce_partials(const float* partials, float* sh, float* tot_g, float* tot_e) {
    const float a = threadIdx.x < kRedBlocks ? partials[threadIdx.x] : 0.f;
    const float b = threadIdx.x < kRedBlocks ? partials[kRedBlocks + threadIdx.x] : 0.f;
    *tot_g = block_sum(a, sh);
    *tot_e = block_sum(b, sh);
}

__global__ __launch_bounds__(256) void ewc_loss_final_kernel(const float* partials, float lam, float* out) {
    __shared__ float sh[4];
    float tg, te;
    reduce_partials(partials, sh, &tg, &te);
    if (threadIdx.x == 0) *out = lam * te;
}

// pass B: clip + AdamW  (torch.optim.AdamW single-tensor path; clip_grad_norm_)
__global__ __launch_bounds__(256) void ewc_adamw_kernel(float* p, const float* g, float* m, float* v,
                                                        const float* F, const float* pold, int64_t n,
                                                        float lam, float two_lam, float max_norm, float lr_wd,
                                                        float beta1, float beta2, float one_m_b1, float one_m_b2,
                                                        float eps, float step_size, float bc2_sqrt,
                                                        const float* partials, float* out, const float* ce_loss,
                                                        float* loss_accum) {
    __shared__ float sh[4];
    float tg, te;
    reduce_partials(partials, sh, &tg, &te);
    const float norm = sqrtf(tg);
    float coef = max_norm / (norm + 1e-6f);
    if (coef > 1.f) coef = 1.f;
    if (max_norm <= 0.f) coef = 1.f;          // max_norm <= 0 disables clipping
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (out) { out[0] = lam * te; out[1] = norm; }
        if (loss_accum) *loss_accum += (ce_loss ? *ce_loss : 0.f) + lam * te;   // total_loss += loss.item()
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float pi = p[i];
        float gi = g[i];
        if (F) gi = fmaf(two_lam, F[i] * (pi - pold[i]), gi);
        gi *= coef;
        pi *= (1.f - lr_wd);                                  // decoupled weight decay
        float mi = m[i];
        mi = mi + (gi - mi) * one_m_b1;                       // exp_avg.lerp_(grad, 1 - beta1)
        float vi = v[i] * beta2;
        vi = vi + one_m_b2 * gi * gi;                         // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);                   // addcdiv_(exp_avg, denom, value=-step_size)
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

int check_dims(const ac_head_dims* d) {
    AC_REQUIRE(d != nullptr, AC_EINVAL, "head: dims is NULL");
    AC_REQUIRE(d->D >= 1 && d->H1 >= 1 && d->H2 >= 1 && d->C >= 1, AC_EINVAL, "head: bad dims %d/%d/%d/%d",
               d->D, d->H1, d->H2, d->C);
    return AC_OK;
}

}  // namespace

extern "C" int64_t ac_head_param_count(const ac_head_dims* dims) {
    if (!dims) return -1;
    return head_offsets(*dims).total;
}

extern "C" int ac_head_workspace(const ac_head_dims* dims, int B, size_t* bytes) {
    int rc = check_dims(dims);
    if (rc) return rc;
    AC_REQUIRE(bytes && B >= 0, AC_EINVAL, "head workspace: bad arguments");
    *bytes = head_ws(*dims, B > 0 ? B : 1).total;
    return AC_OK;
}

extern "C" int ac_head_forward(const ac_head_dims* dims, const float* d_params, const float* d_X,
                               int64_t ldx, int B, float* d_logits, void* d_ws, size_t ws_bytes,
                               ac_stream_t stream_) {
    int rc = check_dims(dims);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (B == 0) return AC_OK;
    AC_REQUIRE(d_params && d_X && d_logits && B > 0 && ldx >= dims->D, AC_EINVAL, "head_forward: bad arguments");
    const ac_head_dims& d = *dims;
    const HeadOffsets o = head_offsets(d);
    const HeadWs w = head_ws(d, B);
    AC_REQUIRE(d_ws && ws_bytes >= w.total, AC_EWORKSPACE, "head_forward: workspace %zu < %zu", ws_bytes, w.total);
    char* ws = (char*)d_ws;
    float* a1 = (float*)(ws + w.a1);
    float* a2 = (float*)(ws + w.a2);
    const float* P = d_params;
    rc = ac::linear_f32(d_X, ldx, P + o.w1, d.D, P + o.b1, nullptr, 0, a1, d.H1, B, d.H1, d.D, 1, nullptr, 1.f, stream);
    if (rc) return rc;
    rc = ac::linear_f32(a1, d.H1, P + o.w2, d.H1, P + o.b2, nullptr, 0, a2, d.H2, B, d.H2, d.H1, 1, nullptr, 1.f, stream);
    if (rc) return rc;
    return ac::linear_f32(a2, d.H2, P + o.w3, d.H2, P + o.b3, nullptr, 0, d_logits, d.C, B, d.C, d.H2, 0, nullptr, 1.f, stream);
}

namespace {

// forward (train mode) + CE + backward into G; masks either explicit (uint8, 1 = keep) or generated
// in-kernel from `seed` when dropout_p > 0 and no mask is given with use_seed.
int head_fwd_bwd(const ac_head_dims& d, const float* P, const float* X, int64_t ldx, const int64_t* y,
                 const uint8_t* mask1, const uint8_t* mask2, float dropout_p, bool use_seed, uint64_t seed, int B,
                 float* d_loss, float* G, char* ws, const HeadWs& w, hipStream_t stream, int loss_kind = AC_LOSS_CE,
                 const float* targets = nullptr, int64_t ldt = 0) {
    const HeadOffsets o = head_offsets(d);
    float* a1 = (float*)(ws + w.a1);
    float* a2 = (float*)(ws + w.a2);
    float* z = (float*)(ws + w.z);
    float* dz = (float*)(ws + w.dz);
    float* d2 = (float*)(ws + w.d2);
    float* d1 = (float*)(ws + w.d1);
    float* rowloss = (float*)(ws + w.rowloss);
    const bool drop = dropout_p > 0.f && (use_seed || mask1 || mask2);
    const float scale = 1.f / (1.f - dropout_p);       // nn.Dropout(0.1), models.py:58
    const float s1 = (drop && (use_seed || mask1)) ? scale : 1.f;
    const float s2 = (drop && (use_seed || mask2)) ? scale : 1.f;
    const float p1 = (use_seed && !mask1) ? dropout_p : 0.f, p2 = (use_seed && !mask2) ? dropout_p : 0.f;
    int rc;
    // forward (train mode): a = dropout(relu(x W^T + b))
    rc = ac::linear_f32(X, ldx, P + o.w1, d.D, P + o.b1, nullptr, 0, a1, d.H1, B, d.H1, d.D, 1, mask1, s1, stream, p1, seed);
    if (rc) return rc;
    rc = ac::linear_f32(a1, d.H1, P + o.w2, d.H1, P + o.b2, nullptr, 0, a2, d.H2, B, d.H2, d.H1, 1, mask2, s2, stream, p2,
                        seed ^ 0xA5A5A5A5A5A5A5A5ull);
    if (rc) return rc;
    const bool top = d.C <= kTopMaxC && B <= kTopMaxB;
    if (top) {
        // logits + loss + dz + gW3 + gb3 + d2 + gb2 in one launch (see head_top_kernel)
        const size_t stage_bytes = ((size_t)B + d.C) * d.H2 * sizeof(float);
        const bool stage = (d.H2 % 4) == 0 && stage_bytes <= 96 * 1024 && ((((uintptr_t)a2) | ((uintptr_t)(P + o.w3))) & 15) == 0;
        if (stage) {
            (void)hipFuncSetAttribute((const void*)head_top_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes);
            hipLaunchKernelGGL(head_top_kernel<true>, dim3(1), dim3(512), stage_bytes, stream, a2, d.H2, P + o.w3, P + o.b3, y,
                               targets, ldt, B, d.C, loss_kind, s2, d2, G + o.w3, G + o.b3, G + o.b2, d_loss);
        } else {
            hipLaunchKernelGGL(head_top_kernel<false>, dim3(1), dim3(512), 0, stream, a2, d.H2, P + o.w3, P + o.b3, y, targets,
                               ldt, B, d.C, loss_kind, s2, d2, G + o.w3, G + o.b3, G + o.b2, d_loss);
        }
        AC_LAUNCH_CHECK();
    } else {
        rc = ac::linear_f32(a2, d.H2, P + o.w3, d.H2, P + o.b3, nullptr, 0, z, d.C, B, d.C, d.H2, 0, nullptr, 1.f, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(loss_fwd_bwd_kernel, dim3(1), dim3(256), 0, stream, z, y, targets, ldt, B, d.C, loss_kind, dz,
                           rowloss, d_loss);
        AC_LAUNCH_CHECK();
        // backward.  dW = dY^T A  (transA=1: dY stored [B,out] is the [K,M] layout), dA = dY W gated
        // by relu'/dropout (a != 0 ? scale : 0).
        rc = ac::gemm_f32(1, 0, d.C, d.H2, B, 1.f, dz, d.C, a2, d.H2, 0.f, G + o.w3, d.H2, nullptr, 0, 1.f, stream);
        if (rc) return rc;
        rc = ac::gemm_f32(0, 0, B, d.H2, d.C, 1.f, dz, d.C, P + o.w3, d.H2, 0.f, d2, d.H2, a2, d.H2, s2, stream);
        if (rc) return rc;
    }
    if (top && B <= 32) {
        // the training batch (32 rows): gW2, d1 (gated) and gb1 in ONE launch, then gW1 -- no bias-gradient launch
        rc = ac::head_backward_pair(d2, d.H2, a1, d.H1, P + o.w2, d.H1, B, d.H2, d.H1, s1, G + o.w2, d1, G + o.b1, stream);
        if (rc) return rc;
        return ac::gemm_f32(1, 0, d.H1, d.D, B, 1.f, d1, d.H1, X, ldx, 0.f, G + o.w1, d.D, nullptr, 0, 1.f, stream);
    }
    rc = ac::gemm_f32(1, 0, d.H2, d.H1, B, 1.f, d2, d.H2, a1, d.H1, 0.f, G + o.w2, d.H1, nullptr, 0, 1.f, stream);
    if (rc) return rc;
    rc = ac::gemm_f32(0, 0, B, d.H1, d.H2, 1.f, d2, d.H2, P + o.w2, d.H1, 0.f, d1, d.H1, a1, d.H1, s1, stream);
    if (rc) return rc;
    rc = ac::gemm_f32(1, 0, d.H1, d.D, B, 1.f, d1, d.H1, X, ldx, 0.f, G + o.w1, d.D, nullptr, 0, 1.f, stream);
    if (rc) return rc;
    const int nb = d.H1 + d.H2 + d.C;
    hipLaunchKernelGGL(bias_grad_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, dz, d2, d1, B, d.C, d.H2,
                       d.H1, top ? nullptr : G + o.b3, top ? nullptr : G + o.b2, G + o.b1);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

int adamw_launch(float* p, const float* g, float* m, float* v, const float* F, const float* pold, int64_t n,
                 float lam, float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step,
                 float* d_out, float* partials, const float* ce_loss, float* loss_accum, hipStream_t stream) {
    const float two_lam = 2.f * lam;
    hipLaunchKernelGGL(ewc_partials_kernel, dim3(kRedBlocks), dim3(256), 0, stream, p, g, F, pold, n, two_lam, partials);
    AC_LAUNCH_CHECK();
    // bias corrections in double like torch's Python floats (adamw single-tensor path)
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float lr_wd = (float)((double)lr * (double)wd);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ewc_adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, F, pold, n, lam,
                       two_lam, max_norm, lr_wd, beta1, beta2, 1.f - beta1, 1.f - beta2, eps, step_size, bc2_sqrt,
                       partials, d_out, ce_loss, loss_accum);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

}  // namespace

extern "C" int ac_head_fwd_bwd_ce(const ac_head_dims* dims, const float* d_params, const float* d_X,
                                  int64_t ldx, const int64_t* d_y, const uint8_t* d_mask1,
                                  const uint8_t* d_mask2, float dropout_p, int B, float* d_loss,
                                  float* d_grads, void* d_ws, size_t ws_bytes, ac_stream_t stream_) {
    int rc = check_dims(dims);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(d_params && d_X && d_y && d_loss && d_grads && B > 0 && ldx >= dims->D, AC_EINVAL,
               "head_fwd_bwd_ce: bad arguments");
    AC_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, AC_EINVAL, "head_fwd_bwd_ce: dropout_p=%f", dropout_p);
    const HeadWs w = head_ws(*dims, B);
    AC_REQUIRE(d_ws && ws_bytes >= w.total, AC_EWORKSPACE, "head_fwd_bwd_ce: workspace %zu < %zu", ws_bytes, w.total);
    return head_fwd_bwd(*dims, d_params, d_X, ldx, d_y, d_mask1, d_mask2, dropout_p, false, 0, B, d_loss, d_grads,
                        (char*)d_ws, w, stream);
}

extern "C" int ac_head_fwd_bwd_loss(const ac_head_dims* dims, const float* d_params, const float* d_X, int64_t ldx,
                                    const int64_t* d_y, const float* d_targets, int64_t ldt, int loss_kind,
                                    const uint8_t* d_mask1, const uint8_t* d_mask2, float dropout_p, int B,
                                    float* d_loss, float* d_grads, void* d_ws, size_t ws_bytes, ac_stream_t stream_) {
    int rc = check_dims(dims);
    if (rc) return rc;
    AC_REQUIRE(d_params && d_X && d_loss && d_grads && B > 0 && ldx >= dims->D, AC_EINVAL, "head_fwd_bwd_loss: bad arguments");
    AC_REQUIRE(loss_kind >= AC_LOSS_CE && loss_kind <= AC_LOSS_CE_SIGMOID, AC_EINVAL, "head_fwd_bwd_loss: loss_kind=%d", loss_kind);
    AC_REQUIRE(loss_kind == AC_LOSS_BCE_SIGMOID ? (d_targets && ldt >= dims->C) : (d_y != nullptr), AC_EINVAL,
               "head_fwd_bwd_loss: BCE needs float targets [B, C]; CE needs int64 labels");
    AC_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, AC_EINVAL, "head_fwd_bwd_loss: dropout_p=%f", dropout_p);
    const HeadWs w = head_ws(*dims, B);
    AC_REQUIRE(d_ws && ws_bytes >= w.total, AC_EWORKSPACE, "head_fwd_bwd_loss: workspace %zu < %zu", ws_bytes, w.total);
    return head_fwd_bwd(*dims, d_params, d_X, ldx, d_y, d_mask1, d_mask2, dropout_p, false, 0, B, d_loss, d_grads,
                        (char*)d_ws, w, (hipStream_t)stream_, loss_kind, d_targets, ldt);
}

extern "C" int ac_sigmoid(const float* d_in, int64_t n, float* d_out, ac_stream_t stream) {
    AC_REQUIRE(d_in && d_out && n >= 0, AC_EINVAL, "sigmoid: bad arguments");
    if (n == 0) return AC_OK;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in, n, d_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_head_train_step(const ac_head_dims* dims, float* d_params, float* d_m, float* d_v, float* d_grads,
                                  const float* d_X, int64_t ldx, const int64_t* d_y, const float* d_targets,
                                  int64_t ldt, int loss_kind, const int64_t* d_index, int B,
                                  float dropout_p, uint64_t dropout_seed, const float* d_fisher, const float* d_old,
                                  float lambda_over_B, float max_grad_norm, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int step, float* d_out, float* d_loss_accum,
                                  void* d_ws, size_t ws_bytes, ac_stream_t stream_) {
    int rc = check_dims(dims);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(d_params && d_m && d_v && d_grads && d_X && d_out && B > 0 && ldx >= dims->D && step >= 1,
               AC_EINVAL, "head_train_step: bad arguments");
    const bool stepwise = (loss_kind & AC_LOSS_STEPWISE) != 0;       // per-call opt-out of the persistent kernel
    loss_kind &= ~AC_LOSS_STEPWISE;
    AC_REQUIRE(loss_kind >= AC_LOSS_CE && loss_kind <= AC_LOSS_CE_SIGMOID, AC_EINVAL, "head_train_step: loss_kind=%d", loss_kind);
    AC_REQUIRE(loss_kind == AC_LOSS_BCE_SIGMOID ? (d_targets && ldt >= dims->C) : (d_y != nullptr), AC_EINVAL,
               "head_train_step: BCE needs float targets [rows, C]; CE needs int64 labels");
    AC_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, AC_EINVAL, "head_train_step: dropout_p=%f", dropout_p);
    AC_REQUIRE((d_fisher == nullptr) == (d_old == nullptr), AC_EINVAL,
               "head_train_step: fisher and old params must be given together");
    const ac_head_dims& d = *dims;
    const HeadWs w = head_ws(d, B);
    AC_REQUIRE(d_ws && ws_bytes >= w.total, AC_EWORKSPACE, "head_train_step: workspace %zu < %zu", ws_bytes, w.total);
    char* ws = (char*)d_ws;
    if (!stepwise) {   // one step of the persistent epoch kernel when the shape fits (the same code path as ac_head_train_epoch)
        const int prc = ac::head_epoch_persistent(d, d_params, d_m, d_v, d_grads, d_X, ldx, d_y, d_targets, ldt, loss_kind, d_index, B, B,
                                                  dropout_p, dropout_seed, d_fisher, d_old, 0.f, d_fisher ? lambda_over_B : 0.f,
                                                  max_grad_norm, lr, beta1, beta2, eps, weight_decay, step, d_out, d_loss_accum,
                                                  ws + w.epoch, stream);
        if (prc != 1) return prc;
    }
    const float* X = d_X;
    const int64_t* y = d_y;
    const float* T = d_targets;
    int64_t ld = ldx, ldT = ldt;
    if (d_index) {       // assemble the batch on device: rows d_index[0..B) of the stored examples
        float* xg = (float*)(ws + w.xg);
        int64_t* yg = (int64_t*)(ws + w.yg);
        float* tg = (float*)(ws + w.tg);
        hipLaunchKernelGGL(gather_batch_kernel, dim3(B), dim3(256), 0, stream, d_X, ldx, d_y, d_targets, ldt, d.C,
                           d_index, B, d.D, xg, yg, tg);
        AC_LAUNCH_CHECK();
        X = xg; y = d_y ? yg : nullptr; ld = d.D;
        if (d_targets) { T = tg; ldT = d.C; }
    }
    rc = head_fwd_bwd(d, d_params, X, ld, y, nullptr, nullptr, dropout_p, dropout_p > 0.f, dropout_seed, B, d_out, d_grads,
                      ws, w, stream, loss_kind, T, ldT);
    if (rc) return rc;
    // d_out[0] = CE loss (written above); [1] = EWC penalty, [2] = grad norm
    return adamw_launch(d_params, d_grads, d_m, d_v, d_fisher, d_old, head_offsets(d).total, lambda_over_B,
                        max_grad_norm, lr, beta1, beta2, eps, weight_decay, step, d_out + 1, (float*)(ws + w.scratch),
                        d_out, d_loss_accum, stream);
}

extern "C" int ac_softmax_rows(const float* d_in, int B, int C, float* d_out, ac_stream_t stream) {
    AC_REQUIRE(d_in && d_out && B >= 0 && C >= 1, AC_EINVAL, "softmax_rows: bad arguments");
    if (B == 0) return AC_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_in, B, C, d_out);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_l2_normalize_rows(const float* d_in, int64_t ldi, int B, int D, float* d_out, int64_t ldo,
                                    ac_stream_t stream) {
    AC_REQUIRE(d_in && d_out && B >= 0 && D >= 1 && ldi >= D && ldo >= D, AC_EINVAL, "l2_normalize_rows: bad arguments");
    if (B == 0) return AC_OK;
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_in, ldi, B, D,
                       d_out, ldo);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_fisher_accumulate(const float* d_grads, float inv_num_batches, float* d_fisher, int64_t n,
                                    ac_stream_t stream) {
    AC_REQUIRE(d_grads && d_fisher && n >= 0, AC_EINVAL, "fisher_accumulate: bad arguments");
    if (n == 0) return AC_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fisher_acc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_grads,
                       inv_num_batches, d_fisher, n);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_ewc_loss(const float* d_params, const float* d_fisher, const float* d_old, int64_t n,
                           float lambda_over_B, float* d_out_loss, void* d_scratch, ac_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AC_REQUIRE(d_params && d_fisher && d_old && d_out_loss && d_scratch && n >= 0, AC_EINVAL,
               "ewc_loss: bad arguments");
    float* partials = (float*)d_scratch;
    hipLaunchKernelGGL(ewc_partials_kernel, dim3(kRedBlocks), dim3(256), 0, stream, d_params, (const float*)nullptr,
                       d_fisher, d_old, n, 0.f, partials);
    AC_LAUNCH_CHECK();
    hipLaunchKernelGGL(ewc_loss_final_kernel, dim3(1), dim3(256), 0, stream, partials, lambda_over_B, d_out_loss);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

extern "C" int ac_ewc_adamw_step(float* d_params, const float* d_grads, float* d_m, float* d_v,
                                 const float* d_fisher, const float* d_old, int64_t n, float lambda_over_B,
                                 float max_grad_norm, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int step, float* d_out, void* d_scratch,
                                 ac_stream_t stream_) {
    AC_REQUIRE(d_params && d_grads && d_m && d_v && d_scratch && n >= 0 && step >= 1, AC_EINVAL,
               "ewc_adamw_step: bad arguments");
    AC_REQUIRE((d_fisher == nullptr) == (d_old == nullptr), AC_EINVAL,
               "ewc_adamw_step: fisher and old params must be given together");
    return adamw_launch(d_params, d_grads, d_m, d_v, d_fisher, d_old, n, lambda_over_B, max_grad_norm, lr, beta1, beta2,
                        eps, weight_decay, step, d_out, (float*)d_scratch, nullptr, nullptr, (hipStream_t)stream_);
}

// One whole epoch of the loop at classifier.py:1485-1507 / :329-353 in ONE call: the batches are consecutive
// slices of d_order (the seeded DataLoader's order for this epoch, uploaded once), step i uses dropout seed
// seed0 + i and AdamW step step0 + i, the EWC weight is lambda_B / (rows of that batch) as `ewc_loss / batch`
// in the reference.  Only launches -- the host-side per-step cost is the kernel enqueues themselves.
extern "C" int ac_head_train_epoch(const ac_head_dims* dims, float* d_params, float* d_m, float* d_v, float* d_grads,
                                   const float* d_X, int64_t ldx, const int64_t* d_y, const float* d_targets,
                                   int64_t ldt, int loss_kind, const int64_t* d_order, int64_t n_total, int batch,
                                   float dropout_p, uint64_t seed0, const float* d_fisher, const float* d_old,
                                   float lambda_B, float max_grad_norm, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int step0, float* d_out, float* d_loss_accum, void* d_ws,
                                   size_t ws_bytes, int* steps_done, ac_stream_t stream) {
    AC_REQUIRE(n_total >= 0 && batch >= 1 && step0 >= 1, AC_EINVAL, "head_train_epoch: bad arguments");
    int n = 0;
    if (n_total > 0 && dims && d_params && d_m && d_v && d_grads && d_X && d_out && d_ws) {
        int rc = check_dims(dims);
        if (rc) return rc;
        const int kind = loss_kind & ~AC_LOSS_STEPWISE;
        AC_REQUIRE(kind >= AC_LOSS_CE && kind <= AC_LOSS_CE_SIGMOID, AC_EINVAL, "head_train_epoch: loss_kind=%d", loss_kind);
        AC_REQUIRE(kind == AC_LOSS_BCE_SIGMOID ? (d_targets && ldt >= dims->C) : (d_y != nullptr), AC_EINVAL,
                   "head_train_epoch: BCE needs float targets [rows, C]; CE needs int64 labels");
        AC_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && ldx >= dims->D, AC_EINVAL, "head_train_epoch: bad dropout_p / ldx");
        AC_REQUIRE((d_fisher == nullptr) == (d_old == nullptr), AC_EINVAL, "head_train_epoch: fisher and old params must be given together");
        const HeadWs w = head_ws(*dims, (int)(n_total < batch ? n_total : batch));
        AC_REQUIRE(ws_bytes >= w.total, AC_EWORKSPACE, "head_train_epoch: workspace %zu < %zu", ws_bytes, w.total);
        rc = (loss_kind & AC_LOSS_STEPWISE) ? 1 :
             ac::head_epoch_persistent(*dims, d_params, d_m, d_v, d_grads, d_X, ldx, d_y, d_targets, ldt, kind, d_order, n_total,
                                       batch, dropout_p, seed0, d_fisher, d_old, lambda_B, -1.f, max_grad_norm, lr, beta1, beta2, eps,
                                       weight_decay, step0, d_out, d_loss_accum, (char*)d_ws + w.epoch, (hipStream_t)stream);
        if (rc == AC_OK) { if (steps_done) *steps_done = (int)((n_total + batch - 1) / batch); return AC_OK; }
        if (rc != 1) return rc;
    }
    for (int64_t off = 0; off < n_total; off += batch, ++n) {
        const int nb = (int)((n_total - off) < batch ? (n_total - off) : batch);
        // d_order == NULL: the caller already laid the rows out in epoch order -> batches are consecutive row slices
        // (no per-step gather launch); identical arithmetic either way
        const int rc = ac_head_train_step(dims, d_params, d_m, d_v, d_grads, d_order ? d_X : d_X + off * ldx, ldx,
                                          (d_order || !d_y) ? d_y : d_y + off,
                                          (d_order || !d_targets) ? d_targets : d_targets + off * ldt, ldt, loss_kind,
                                          d_order ? d_order + off : nullptr, nb, dropout_p, seed0 + (uint64_t)n, d_fisher, d_old,
                                          d_fisher ? (float)((double)lambda_B / nb) : 0.f, max_grad_norm, lr, beta1, beta2, eps,
                                          weight_decay, step0 + n, d_out, d_loss_accum, d_ws, ws_bytes, stream);
        if (rc) return rc;
    }
    if (steps_done) *steps_done = n;
    return AC_OK;
}
