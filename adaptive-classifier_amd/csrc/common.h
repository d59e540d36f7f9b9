// Shared host-side helpers for libacamd.so (gfx950 only; no portability layer).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/acamd.h"

namespace ac {

void set_error(const char* fmt, ...);

struct DevInfo {
    int cus;             // CUs this process's workgroups can land on (measured once per device: common.hip; <= hw_cus under a CU mask)
    int lds_per_block;   // max dynamic+static LDS per workgroup (bytes)
    size_t hbm_bytes;
    int hw_cus;          // the chip's CU count (hipDeviceProp.multiProcessorCount)
};
const DevInfo& dev_info();   // lazily queried for the current device

// The process-wide switches of the ABI (ac_gemm_set_arith / _variant / _pipe_table / _krot / _ln_fusion, ac_set_persistent_kernels,
// ac_gemm_debug_stamps) are TEST HOOKS: they exist for tests, A/B runs and tools, and do nothing in a process that has not asked
// for them with AC_TEST_HOOKS=1 in its environment at load time (they return AC_EUNSUPPORTED).  A production process therefore has
// no mutable global state besides the per-thread error string; what a call computes is decided by its arguments
// (ac_bert_config.*_opt) and by the read-once configuration variables (AC_GEMM_ARITH, AC_LN_FUSION, ...).
bool test_hooks_enabled();
#define AC_TEST_HOOK_ONLY(name)                                                                                                   \
    do {                                                                                                                          \
        if (!ac::test_hooks_enabled()) {                                                                                          \
            ac::set_error(name ": process-wide switches are test hooks; this process did not enable them (AC_TEST_HOOKS=1)");    \
            return AC_EUNSUPPORTED;                                                                                               \
        }                                                                                                                         \
    } while (0)

// One-time-per-DEVICE work on a latency-critical launch path (function attributes such as the dynamic-LDS opt-in are per
// device; a plain `static bool` would skip them on a process's second GPU).  `done` = a static std::atomic<uint64_t> of the call
// site; returns true the first time the current device asks (racing threads may both see true: the work must be idempotent).
bool first_call_on_device(std::atomic<unsigned long long>& done);

#define AC_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ac::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                          __FILE__, __LINE__);                                          \
            return AC_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define AC_REQUIRE(cond, code, ...)                                                     \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ac::set_error(__VA_ARGS__);                                                 \
            return (code);                                                              \
        }                                                                               \
    } while (0)

#define AC_LAUNCH_CHECK()                                                               \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            ac::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),    \
                          __FILE__, __LINE__);                                          \
            return AC_EHIP;                                                             \
        }                                                                               \
    } while (0)

__host__ __device__ static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// gemm.hip: fp32 MFMA GEMMs shared by head.hip and bert.hip
//   C[M,N] = epi(A[M,K] . W[N,K]^T): bias, act (0 none / 1 relu / 2 gelu-erf), optional inverted
//   dropout mask (uint8 [M,N], kept values * mask_scale), optional residual added last.
int linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
               const float* residual, int64_t ldr, float* C, int64_t ldc, int M, int N, int K, int act,
               const uint8_t* mask, float mask_scale, hipStream_t stream, float drop_p = 0.f,
               uint64_t drop_seed = 0, const uint16_t* W_planes = nullptr, const uint16_t* A_planes = nullptr,
               uint16_t* C_planes = nullptr, int64_t w_plane_rows = 0);
// w_plane_rows (here and below): rows of the [rows, K] matrix whose planes W_planes points INTO (0 = N): a GEMM over a block
// of N consecutive output rows of a larger weight (the K | V rows of a fused QKV matrix) passes W_planes + 8 * first_row
// split-K form of linear_f32 for GEMMs with few output tiles (W planes required, scratch = ksplit * M * N floats)
int linear_f32_splitk(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* residual,
                      int64_t ldr, float* C, int64_t ldc, int M, int N, int K, int act, const uint16_t* W_planes, float* scratch,
                      size_t scratch_bytes, hipStream_t stream, int64_t w_plane_rows = 0, int* partials_only = nullptr);
// gemm_pipe.hip:
// bias + residual + LayerNorm fused into the epilogue (one-round launches of the 128 x 128 tile only: see gemm_pipe.hip);
// `count` = one zeroed word per 128-row panel, `part` = pipe_ln_part_bytes() of scratch, C may alias the residual
bool pipe_ln_applies(int M, int N, int K);
size_t pipe_ln_part_bytes(int M, int N);
int pipe_ln_panels(int M);
int launch_gemm_pipe_ln(const uint16_t* Ap, int64_t a_rows, const uint16_t* Wp, int64_t w_rows, const float* bias,
                        const float* residual, int64_t ldr, float* C, int64_t ldc, int M, int N, int K, const float* gamma,
                        const float* beta, float eps, void* part, unsigned* count, unsigned* abort_flag, uint16_t* planes,
                        hipStream_t stream, int f16 = 0);      // f16: operands AND emitted planes are fp16x2
// the [T, 3H] QKV projection of a PACKED batch with the self-attention in its epilogue (gemm_pipe.hip EPI_QKV_ATTN): context rows
// go straight to `ctx_planes`; `qkv` (fp32 [T, 3H]) only receives the rows of sequences that straddle a 256-row tile boundary,
// which the caller then serves with attention_mfma_kernel's boundary mode (boundary_stride = kQkvAttnRows)
constexpr int kQkvAttnRows = 256;
constexpr int kQkvAttnCu = 264;            // words of one row tile's entry in the sequence table (gemm_pipe.hip kAtCu)
bool qkv_attn_applies(int M, int H, int heads, int smax);
size_t qkv_attn_tile_seq_bytes(int M);
int qkv_attn_tile_seq(const int32_t* cu, int b, int M, int32_t* tile_seq, hipStream_t stream);     // once per forward (cu is the same for every layer)
// exchange (optional; qkv_attn_exchange_applies must hold: one proven residency round): ceil(M / 256) * heads words, zero before the
// forward's first launch, `epoch` distinct per launch (layer + 1) -- straddling sequences are then finished inside the launch and
// no boundary launch is needed; abort_flag = the encoder call's give-up word
bool qkv_attn_exchange_applies(int M, int heads, int f16);
int launch_gemm_pipe_qkv_attn(const uint16_t* Ap, int64_t a_rows, const uint16_t* Wp, int64_t w_rows, const float* bias, int M, int H,
                              int heads, const int32_t* cu, const int32_t* tile_seq, int b, int smax, float scale, uint16_t* ctx_planes,
                              float* qkv, hipStream_t stream, int f16 = 0, unsigned* exchange = nullptr, unsigned epoch = 0,
                              unsigned* abort_flag = nullptr);
// true when linear_f32(M, N, K) with W planes takes the pre-split kernel (only then may A / C planes be passed)
bool linear_takes_planes(int M, int N, int K);

// counter-based Bernoulli(1-p) keep decision for in-kernel dropout (stateless: seed + element index)
__host__ __device__ static inline bool dropout_keep(uint64_t seed, uint64_t idx, float p) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f) >= p;
}
#ifdef __HIPCC__
// ---- bf16x3 split helpers (AC_GEMM_BF16X3): x = h + m + l, each term a bf16, RNE via v_cvt_pk_bf16_f32 ----
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const f32x2_t f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
// two fp32 -> one dword per plane (a in the low half)
__device__ __forceinline__ void split2(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = pack_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pack_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = pack_bf16(sa, sb);
}
__device__ __forceinline__ void split4(const f32x4_t& x, uint2& H, uint2& M, uint2& L) {
    split2(x[0], x[1], H.x, M.x, L.x);
    split2(x[2], x[3], H.y, M.y, L.y);
}
// 8 consecutive fp32 (two float4) -> three planes of 8 bf16
__device__ __forceinline__ void split8(const f32x4_t& x0, const f32x4_t& x1, uint4& H, uint4& M, uint4& L) {
    split2(x0[0], x0[1], H.x, M.x, L.x);
    split2(x0[2], x0[3], H.y, M.y, L.y);
    split2(x1[0], x1[1], H.z, M.z, L.z);
    split2(x1[2], x1[3], H.w, M.w, L.w);
}
// ---- fp16x2 operand planes (AC_GEMM_F16X2): x * 2^s = h + l, two fp16 terms (RNE), same plane layout, planes 0 and 1 ----
// h = fp16(x 2^s), l = fp16(x 2^s - h): the subtraction is exact, so |x 2^s - h - l| <= max(2^-22 |x 2^s|, 2^-25) (the
// second term: l in fp16's subnormal range).  The scale is a fixed power of two per operand KIND -- activations 2^6 (|x| <
// 1023.5), weights 2^10 (|w| < 63.97) -- so a GEMM's result is the fp32 accumulator times 2^-16.  An operand element beyond
// that range makes h = inf, l = -inf, hence NaN in every output it feeds: the caller sees non-finite rows and repeats the
// call in bf16x3 (adaptive_classifier/encoder.py), it does not get a silently wrong number.
constexpr int kF16ActLog2 = 6, kF16WLog2 = 10;
constexpr float kF16ActScale = (float)(1 << kF16ActLog2), kF16OutScale = 1.0f / (float)(1 << (kF16ActLog2 + kF16WLog2));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h(float a, float b, float s, uint32_t& h, uint32_t& l) {
    a *= s; b *= s;
    const f16x2_t hv = {(_Float16)a, (_Float16)b};
    const f16x2_t lv = {(_Float16)(a - (float)hv[0]), (_Float16)(b - (float)hv[1])};
    h = __builtin_bit_cast(uint32_t, hv);
    l = __builtin_bit_cast(uint32_t, lv);
}
__device__ __forceinline__ void split4h(const f32x4_t& x, float s, uint2& H, uint2& L) {
    split2h(x[0], x[1], s, H.x, L.x);
    split2h(x[2], x[3], s, H.y, L.y);
}
__device__ __forceinline__ void split8h(const f32x4_t& x0, const f32x4_t& x1, float s, uint4& H, uint4& L) {
    split2h(x0[0], x0[1], s, H.x, L.x);
    split2h(x0[2], x0[3], s, H.y, L.y);
    split2h(x1[0], x1[1], s, H.z, L.z);
    split2h(x1[2], x1[3], s, H.w, L.w);
}
// what a producer of ACTIVATION planes stores for 4 / 8 consecutive k of one row: three bf16 planes, or (f16 != 0, wave
// uniform) the two fp16 planes; p = the element's address in plane 0, `plane` = the plane stride
__device__ __forceinline__ void emit_planes4(uint16_t* p, int64_t plane, const f32x4_t& v, int f16) {
    if (f16) {
        uint2 H, L;
        split4h(v, kF16ActScale, H, L);
        *reinterpret_cast<uint2*>(p) = H;
        *reinterpret_cast<uint2*>(p + plane) = L;
    } else {
        uint2 H, M, L;
        split4(v, H, M, L);
        *reinterpret_cast<uint2*>(p) = H;
        *reinterpret_cast<uint2*>(p + plane) = M;
        *reinterpret_cast<uint2*>(p + 2 * plane) = L;
    }
}
__device__ __forceinline__ void emit_planes8(uint16_t* p, int64_t plane, const f32x4_t& v0, const f32x4_t& v1, int f16) {
    if (f16) {
        uint4 H, L;
        split8h(v0, v1, kF16ActScale, H, L);
        *reinterpret_cast<uint4*>(p) = H;
        *reinterpret_cast<uint4*>(p + plane) = L;
    } else {
        uint4 H, M, L;
        split8(v0, v1, H, M, L);
        *reinterpret_cast<uint4*>(p) = H;
        *reinterpret_cast<uint4*>(p + plane) = M;
        *reinterpret_cast<uint4*>(p + 2 * plane) = L;
    }
}
// ---- GELU (erf form, transformers' "gelu": modeling_bert.py:325-337 via ACT2FN) ----
// gelu(x) = x/2 (1 + erf(x / sqrt 2)) with a BRANCH-FREE erf: erf(|t|) = 1 - 2^(-|t| P(|t|)), P a degree-8 polynomial fitted
// (weighted minimax on the error of erf, fp64 fit error 2.2e-9) to -log2(erfc(t)) / t on (0, 4]; beyond 4 the result rounds to 1
// in fp32 anyway.  17 VALU instructions with one v_exp_f32 instead of the ~45 of ocml's two-branch erff (both branches execute in
// a divergent wave) -- the FFN1 epilogue evaluates it 128 times per lane with the matrix pipe idle.  Accuracy in
// fp32 against the exact function (6M points on [-8, 8], tools/gelu_check.py): |erf| error <= 8.5e-8, |gelu| error <= 4.6e-7 --
// torch's own fp32 CPU gelu sits at 1.2e-6 on the same points.  NaN in, NaN out; gelu(+inf) = +inf.
__device__ __forceinline__ float gelu_erf(float x) {
    const float t = x * 0.70710678118654752440f;
    const float a = fminf(__builtin_fabsf(t), 4.0f);
    float p = -1.160484225692926e-05f;
    p = __builtin_fmaf(p, a, 0.00015296465426217765f);
    p = __builtin_fmaf(p, a, -0.0008482354460284114f);
    p = __builtin_fmaf(p, a, 0.002274787751957774f);
    p = __builtin_fmaf(p, a, -8.480761607643217e-05f);
    p = __builtin_fmaf(p, a, -0.027724474668502808f);
    p = __builtin_fmaf(p, a, 0.1483079046010971f);
    p = __builtin_fmaf(p, a, 0.9184429049491882f);
    p = __builtin_fmaf(p, a, 1.6279072761535645f);
    const float e = __builtin_amdgcn_exp2f(-(p * a));          // v_exp_f32: 2^-E, E in [0, 26.2]
    const float r = __builtin_copysignf(1.0f - e, t);
    return 0.5f * x * (1.0f + r);
}

// Two at a time on the packed fp32 ALU (v_pk_mul / v_pk_fma / v_pk_add_f32: two IEEE fp32 operations per lane and issue slot --
// the same operations in the same order as gelu_erf, hence the same bits; only v_exp_f32 and the sign transfer stay scalar).  The
// FFN1 epilogue is VALU-bound (128 evaluations per lane under an idle matrix pipe): 34 -> 21 issue slots per pair.
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu_erf2(gelu_f32x2 x) {
    const gelu_f32x2 t = x * 0.70710678118654752440f;
    const gelu_f32x2 a = __builtin_elementwise_min(__builtin_elementwise_abs(t), (gelu_f32x2){4.0f, 4.0f});
    gelu_f32x2 p = {-1.160484225692926e-05f, -1.160484225692926e-05f};
    p = __builtin_elementwise_fma(p, a, (gelu_f32x2){0.00015296465426217765f, 0.00015296465426217765f});
    p = __builtin_elementwise_fma(p, a, (gelu_f32x2){-0.0008482354460284114f, -0.0008482354460284114f});
    p = __builtin_elementwise_fma(p, a, (gelu_f32x2){0.002274787751957774f, 0.002274787751957774f});
    p = __builtin_elementwise_fma(p, a, (gelu_f32x2){-8.480761607643217e-05f, -8.480761607643217e-05f});
    p = __builtin_elementwise_fma(p, a, (gelu_f32x2){-0.027724474668502808f, -0.027724474668502808f});
    p = __builtin_elementwise_fma(p, a, (gelu_f32x2){0.1483079046010971f, 0.1483079046010971f});
    p = __builtin_elementwise_fma(p, a, (gelu_f32x2){0.9184429049491882f, 0.9184429049491882f});
    p = __builtin_elementwise_fma(p, a, (gelu_f32x2){1.6279072761535645f, 1.6279072761535645f});
    const gelu_f32x2 pa = p * a;
    const gelu_f32x2 e = {__builtin_amdgcn_exp2f(-pa[0]), __builtin_amdgcn_exp2f(-pa[1])};
    const gelu_f32x2 om = (gelu_f32x2){1.0f, 1.0f} - e;
    const gelu_f32x2 r = {__builtin_copysignf(om[0], t[0]), __builtin_copysignf(om[1], t[1])};
    return (x * 0.5f) * ((gelu_f32x2){1.0f, 1.0f} + r);
}

// element offset (uint16 units) of (row, k) inside one plane of a [rows, K] operand: planes[p][k/8][row][k%8]
__device__ __forceinline__ int64_t plane_off(int64_t rows, int64_t row, int k) {
    return ((int64_t)(k >> 3) * rows + row) * 8 + (k & 7);
}
#endif

// knn_batch.hip: fp16 operand plane + GEMM-form proposal sweep for batched kNN (used by knn_l2.hip)
size_t knn_planes_bytes(int64_t rows, int D);      // bytes of the fp16 plane of a [rows, D] operand (rows padded to 256, D to 64)
double knn_batch_gamma(int D);                     // |sweep value - exact| <= gamma (max|p| + |q|)^2
int knn_prepare_store(const float* X, int64_t ldx, int64_t rows, int D, uint16_t* plane, float* norms, uint32_t* maxnorm_bits,
                      hipStream_t stream);
int knn_update_store(const float* X, int64_t ldx, int64_t n_old, int64_t n_new, int D, uint16_t* plane, float* norms, int64_t row0,
                     int64_t nrows, int32_t* exponent_changed, hipStream_t stream);
int knn_prepare_queries(const double* sampleD, int kp, const float* Q, int64_t ldQ, int D, int nq, const uint32_t* maxnorm_bits,
                        double gamma, uint16_t* qplane, float* thr, float* qfac, hipStream_t stream, int32_t* zero_ints = nullptr,
                        int64_t zero_count = 0);             // zero_ints: device ints the same launch clears (the candidate counters)
int64_t knn_sample_rows(int64_t N, int64_t stride);
int knn_batch_launch(const uint16_t* Pp, const float* pnorm, int64_t N, int D, const uint16_t* Qp, int nq, const float* thr,
                     const float* qfac, float* cand_d, int32_t* cand_i, int32_t* cand_cnt, int cap, int segs, int64_t row_stride,
                     int best_only, hipStream_t stream, int32_t* clear_ctr = nullptr, int32_t* clear_stats = nullptr,
                     int two_phase_kp = 0, unsigned* wgmin = nullptr, void* ctl = nullptr);
bool ln_fusion_enabled();        // gemm_pipe.hip: false = this call / process runs no in-launch exchange between workgroups
bool knn_batch_two_phase_applies(int64_t N, int nq, int kp, int segs);
size_t knn_batch_two_phase_bytes();

// Per-call options (ac_bert_config.gemm_arith_opt / ln_fusion_opt / one_launch_opt): for the duration of ONE native call on the
// calling thread they take precedence over the process-wide switches (ac_gemm_set_arith, ac_gemm_set_ln_fusion,
// ac_set_persistent_kernels), which remain as test / A-B hooks and as the default of calls that do not say.  Thread-local and
// scoped (RAII): no state outlives the call, two objects with different options interleave freely on one or on several threads.
struct CallOpts { int arith = -1, ln_fusion = -1, one_launch = -1; };     // -1 = not given: the process-wide value
CallOpts& call_opts();
struct CallScope {
    CallOpts saved;
    CallScope(int arith_opt, int ln_fusion_opt, int one_launch_opt);       // *_opt encoding of acamd.h: 0 = default, value + 1
    ~CallScope();
    CallScope(const CallScope&) = delete;
    CallScope& operator=(const CallScope&) = delete;
};

// arithmetic of the LDS-tiled GEMMs (per-call option, else ac_gemm_set_arith / env AC_GEMM_ARITH = f32 | bf16x3 | f16x2)
int gemm_arith();
void set_gemm_arith(int mode);
inline bool arith_split() { return gemm_arith() != AC_GEMM_F32; }   // bf16x3, or f16x2 (= bf16x3 wherever no fp16 planes exist)
// C = epi(A W^T) on fp16x2 operand planes (gemm_pipe.hip only): A planes at scale 2^kF16ActLog2, W planes at 2^kF16WLog2.
// C fp32 rows, or (Cp) the fp16x2 activation planes of the next GEMM.  act: 0 none, 2 gelu (planes output only).
bool linear_f16x2_takes(int M, int N, int K);
int linear_f16x2(const uint16_t* Ap, const uint16_t* Wp, const float* bias, const float* residual, int64_t ldr, float* C,
                 int64_t ldc, uint16_t* Cp, int M, int N, int K, int act, hipStream_t stream, int64_t w_plane_rows = 0);
//   C = alpha * op(A) op(B) + beta * C; if gate != null: C = gate[m,n] != 0 ? C * gate_scale : 0
int gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int64_t lda,
             const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const float* gate,
             int64_t ldg, float gate_scale, hipStream_t stream);

// The head's backward through one hidden layer in ONE launch (B <= 32): gW[Hout,Hin] = dY^T Aact, dA[B,Hin] = (dY W) gated by
// Aact != 0 (x gate_scale), gb_in[Hin] = column sums of dA (rows ascending).  gemm.hip.
int head_backward_pair(const float* dY, int64_t ldy, const float* Aact, int64_t lda_act, const float* W, int64_t ldw, int B,
                       int Hout, int Hin, float gate_scale, float* gW, float* dA, float* gb_in, hipStream_t stream);

// bit 0: head_epoch.hip, bit 1: bert_small.hip (ac_set_persistent_kernels; common.hip)
int persistent_mask();
int set_persistent_mask(int m);

// head_epoch.hip: the weights-stationary persistent training epoch (see there).  Returns AC_OK, 1 (shape not covered:
// fall back to the step-by-step launches) or an error code.  lam_direct >= 0: EWC weight of every step; < 0: lambda_B / rows.
size_t head_epoch_ws_bytes(int H1, int H2);
int head_epoch_persistent(const ac_head_dims& d, float* P, float* M, float* V, float* Gout, const float* X, int64_t ldx,
                          const int64_t* y, const float* T, int64_t ldt, int loss_kind, const int64_t* order, int64_t n_total,
                          int batch, float dropout_p, uint64_t seed0, const float* F, const float* Old, float lambda_B,
                          float lam_direct, float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step0,
                          float* out, float* loss_accum, void* ws, hipStream_t stream);

long long head_epoch_launches();       // persistent launches so far (diagnostic: ac_persistent_launches)
long long bert_small_launches();
// bert_small.hip: BERT forward of <= 32 token rows in one persistent launch.  AC_OK, 1 (shape not covered) or an error.
size_t bert_small_ws_bytes(int H, int I);
int bert_small_encode(const ac_bert_config& c, const ac_bert_weights& w, const int64_t* ids, const int64_t* type_ids,
                      const int64_t* mask, int b, int S, float* out, int64_t ldo, void* ws, hipStream_t stream);
int bert_small_aborted(int H, int I, const void* ws, hipStream_t stream, int* aborted);

}  // namespace ac
