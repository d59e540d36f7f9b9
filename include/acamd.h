/*
 * acamd.h -- C ABI of libacamd.so: the MI355X (gfx950) hot path of
 * codelion/adaptive-classifier's predict()/add_examples().
 *
 * The reference has no FFI of its own: it is Python calling three third-party
 * native libraries (faiss, torch, transformers).  Each entry point below names
 * the reference call site (file:line under /root/reference) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative AC_E* code on failure;
 *     ac_last_error() returns a per-thread message for the last failure.
 *   - all pointers named d_* are DEVICE pointers (HBM) owned by the caller
 *     (e.g. torch.Tensor.data_ptr()); nothing is allocated behind the caller's
 *     back: scratch space is passed in and sized by the *_workspace() queries.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Calls
 *     only enqueue work; they never synchronise the stream.
 *   - the library keeps no global mutable state except the per-thread error
 *     string and a lazily cached device-properties struct.
 *   - fp32 arithmetic unless stated; ids are int64 like faiss::idx_t.
 */
#ifndef ACAMD_H
#define ACAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AC_OK            0
#define AC_EINVAL       -1   /* bad argument (shape, alignment, null) */
#define AC_EUNSUPPORTED -2   /* valid request outside what the kernels cover */
#define AC_EWORKSPACE   -3   /* workspace too small */
#define AC_EHIP         -4   /* HIP runtime error (message has hipGetErrorString) */

typedef void* ac_stream_t;

const char* ac_last_error(void);
int ac_version(void);                 /* 100*major + minor */
int ac_device_info(int* cu_count, int* lds_bytes_per_block, size_t* hbm_bytes);
/* chip_cus = the device's CU count; active_cus = the CUs this process's workgroups can actually land on: chip_cus, unless the
 * environment carries HSA_CU_MASK / ROC_GLOBAL_CU_MASK -- then it is MEASURED once per device by a probe launch (after a device
 * synchronise); env AC_ACTIVE_CUS overrides, for callers whose own stream carries a CU mask.  cu_count of ac_device_info is active_cus: every co-residency decision of the library (persistent
 * kernels' grids, the fused-LayerNorm exchange, one-round tile choice, sweep grids) is made against it, i.e. a fused / persistent
 * path whose workgroups could not all be resident is NOT SELECTED, instead of selected and abandoned by a bounded wait. */
int ac_device_cus(int* chip_cus, int* active_cus);
/* Diagnostic (tests: "was the persistent path SELECTED"): launches so far of the persistent training epoch (head_epoch.hip) and of
 * the one-launch small-batch encoder (bert_small.hip) by this process. */
int ac_persistent_launches(int64_t* head_epoch, int64_t* bert_small);

/* ------------------------------------------------------------------------- *
 *  M: PrototypeMemory kNN  (faiss.IndexFlatL2.search, memory.py:114)
 * ------------------------------------------------------------------------- */

/* Limit of the fused sweep: k <= AC_KNN_MAX_K (and a 16-query tile of width D plus its
 * candidate lists must fit the 160 KB LDS, D <~ 2000).  Outside it ac_knn_l2_topk uses an exact
 * small-store path (any k, any D) when N <= 8192 -- the reference's own regime, one prototype per
 * class searched with k = #classes (classifier.py:424-425) -- and returns AC_EUNSUPPORTED otherwise. */
#define AC_KNN_MAX_K 248

/* Bytes of scratch ac_knn_l2_topk needs for this problem size. */
int ac_knn_l2_topk_workspace(int64_t N, int D, int nq, int k, size_t* bytes);

/*
 * Exact squared-L2 k-nearest rows, the semantics of faiss.IndexFlatL2.search
 * (memory.py:114): for each of the nq queries the k smallest
 * ||P[n,:] - Q[q,:]||^2 in ascending order, ties broken by the lower row id.
 *
 * Exactness contract (what "bit-exact top-k" means here):
 * the returned ids are the top-k under the *exactly computed* distance
 * (fp64 accumulation of (p-q)^2, which is exact to 1e-16 relative for fp32
 * inputs), and d_outD is that distance rounded once to fp32.  Internally an
 * fp32-MFMA sweep proposes k+pad candidates per query, the candidates are
 * re-ranked in fp64, and a per-query error-bound certificate decides whether
 * the proposal provably contains the true top-k; queries that fail the
 * certificate are redone by an exact fp64 sweep inside the same call.
 *
 *   d_P   [N, ldP] row-major fp32 prototype rows of this shard, 16-byte
 *         aligned, ldP % 4 == 0, columns D..round_up(D,4) zero
 *   d_Q   [nq, ldQ] row-major fp32 queries
 *   row_offset  added to local row ids (global id of row 0 of this shard)
 *   d_outD [nq, k] fp32, d_outI [nq, k] int64; if k > N the tail is padded
 *         with (FLT_MAX, -1) like faiss
 *   d_stats  optional int32[4] (may be NULL): {queries that took the exact
 *         fallback, 1 if the sweep streamed the rows through the LDS ring
 *         (knn_sweep_ring: nq <= 16, D % 32 == 0, D <= 1024) else 0, 0, 0}
 */
int ac_knn_l2_topk(const float* d_P, int64_t N, int64_t ldP, int D,
                   const float* d_Q, int nq, int64_t ldQ, int k,
                   int64_t row_offset,
                   float* d_outD, int64_t* d_outI,
                   void* d_ws, size_t ws_bytes, int32_t* d_stats,
                   ac_stream_t stream);

/*
 * ac_knn_l2_topk with one more output: d_outD64 [nq, k] (may be NULL) receives the exact fp64 distances whose
 * fp32 roundings go to d_outD (padding: +inf).  A row-sharded search must merge the shards' lists by
 * THESE: two candidates whose exact distances differ but round to the same fp32 value would otherwise be ordered
 * by id instead of by distance, and the sharded result could differ from the unsharded one (memory.py:114 has one
 * index, so the reference never faces this).
 */
int ac_knn_l2_topk_x(const float* d_P, int64_t N, int64_t ldP, int D,
                     const float* d_Q, int nq, int64_t ldQ, int k,
                     int64_t row_offset,
                     float* d_outD, double* d_outD64, int64_t* d_outI,
                     void* d_ws, size_t ws_bytes, int32_t* d_stats,
                     ac_stream_t stream);

/*
 * Search over a PREPARED store (BASELINE configs[2] / [4], predict_batch; replaces the same faiss call, memory.py:114).  Same
 * result contract as ac_knn_l2_topk_x -- the ids are the exact top-k, bit for bit -- but candidates are PROPOSED from ONE fp16
 * plane per operand (one v_mfma_f32_32x32x16_f16 per 32 x 32 x 16 block) and only the re-rank / certificate / fallback stay in
 * fp64:
 *   nq <= 64   knn_plane_sweep (knn_l2.hip): ONE bandwidth-bound pass over the store's plane (2 B per element: half the bytes of
 *              the fp32 sweep) with the query tile resident in LDS; d_stats[1] = 2 reports that it ran;
 *   nq  > 64   knn_batch_sweep (knn_batch.hip): a GEMM on the matrix pipe, thresholds from strided samples.
 *   ac_knn_store_bytes       sizes of the two auxiliary buffers of a store of N rows
 *   ac_knn_prepare_store     fills them from the fp32 rows: d_planes = fp16(p 2^-e_p), 2^e_p > max |p|, TILE-MAJOR
 *                            [rows / 256][K / 8][256][8] (rows padded to 256, K to 64: a 256-row tile is one contiguous run,
 *                            k-slot-major inside; an opaque layout -- only these entry points read it); d_norms = |p|^2 per
 *                            row, +inf on the padding rows, the maximum at [round_up(N, 256)].  Redo after any row changes.
 *   ac_knn_l2_topk_batch     N >= 65536, k <= 100 (AC_EUNSUPPORTED otherwise: use ac_knn_l2_topk_x); d_stats as above
 * Error bound of the proposal value v = |p|^2 - 2 p.q used by the filter and the certificate (unit = (max|p| + |q|)^2):
 *   operands scaled into the unit ball (p^ = p 2^-e_p, q^ = q 2^-e_q, per-query e_q) and rounded to nearest fp16:
 *   |x^ - h| <= 2^-11 |x^| + 2^-25, so |p^.q^ - h_p.h_q| <= 2^-10 (1 + 2^-10) |p^||q^| + 2^-25 sqrt(K) (|p^| + |q^|);
 *   fp16 products are exact in the fp32 accumulator, whose K + 16 additions are charged 2 ulp each; times 2 2^(e_p+e_q),
 *   plus the roundings of |p|^2 and of the final fma:  gamma = 1.01 (2^-11 + (K + 18 + sqrt K) 2^-24).
 */
int ac_knn_store_bytes(int64_t N, int D, size_t* planes_bytes, size_t* norms_bytes);
int ac_knn_prepare_store(const float* d_P, int64_t N, int64_t ldP, int D,
                         uint16_t* d_planes, float* d_norms, ac_stream_t stream);
/* Incremental maintenance of a prepared store after `index.add` / an in-place row update (memory.py:159,172,190; round 4):
 * rows [row0, row0 + nrows) of d_P changed or were appended (N_new >= N_old; an append passes row0 <= N_old and
 * row0 + nrows == N_new).  d_planes / d_norms must hold a store of N_new rows (ac_knn_store_bytes(N_new); the tile-major plane and
 * the norms of the first N_old rows are a PREFIX of the larger buffers, so growing them is a plain copy).  Recomputes the norms
 * and fp16 plane entries of those rows, carries the maximum |p|^2 (it only ever grows: an upper bound is what the certificate
 * needs), pads the new last tile.  *d_exponent_changed (device int) = 1 when the new maximum moved the store's power-of-two
 * scale: every OTHER row's plane entry is stale then and the caller must ac_knn_prepare_store again; 0 = the store is ready.
 * Cost: four small launches + nrows rows, instead of two passes over all N rows. */
int ac_knn_update_store(const float* d_P, int64_t N_old, int64_t N_new, int64_t ldP, int D,
                        uint16_t* d_planes, float* d_norms, int64_t row0, int64_t nrows,
                        int32_t* d_exponent_changed, ac_stream_t stream);
int ac_knn_l2_topk_batch_workspace(int64_t N, int D, int nq, int k, size_t* bytes);
int ac_knn_l2_topk_batch(const float* d_P, int64_t N, int64_t ldP, int D,
                         const uint16_t* d_planes, const float* d_norms,
                         const float* d_Q, int nq, int64_t ldQ, int k,
                         int64_t row_offset,
                         float* d_outD, double* d_outD64, int64_t* d_outI,
                         void* d_ws, size_t ws_bytes, int32_t* d_stats,
                         ac_stream_t stream);

/*
 * Optional profiling hook: when both events are non-NULL, every following
 * ac_knn_l2_topk call of THIS thread records `start` immediately before and
 * `stop` immediately after its sweep kernel (the HBM-bound kernel) on the
 * call's stream, so the caller can read that kernel's duration with
 * hipEventElapsedTime.  Pass NULLs to switch it off.  (hipEvent_t as void*.)
 */
int ac_knn_set_profile_events(void* start_event, void* stop_event);

/*
 * (No reference counterpart: the reference searches one CPU index, memory.py:114.)
 * Merge per-shard results: in [shards, nq, k] ascending
 * lists -> global ascending top-k by (distance, id).  Entries with id < 0 are
 * padding.  Pure selection: no arithmetic on the distances.
 */
int ac_topk_merge(const float* d_D_in, const int64_t* d_I_in, int shards,
                  int nq, int k, float* d_outD, int64_t* d_outI,
                  ac_stream_t stream);
/* The same over exact fp64 per-shard distances (ac_knn_l2_topk_x's d_outD64): order by (exact distance, id),
 * emit the fp32 rounding -- the merge a row-sharded search uses so that it equals the unsharded search bit for bit. */
int ac_topk_merge_f64(const double* d_D_in, const int64_t* d_I_in, int shards,
                      int nq, int k, float* d_outD, int64_t* d_outI,
                      ac_stream_t stream);

/*
 * memory.py:117,129-130: s = exp(-d) per hit, then softmax over the k hits of
 * each query.  Hits with id < 0 get score 0 and are left out of the softmax.
 */
int ac_proto_scores(const float* d_D, const int64_t* d_I, int nq, int k,
                    float* d_out, ac_stream_t stream);

/*
 * Hit row ids -> class ids (index_to_label lookup, memory.py:121-125, batched):
 *   c = d_row_class ? d_row_class[id] : id ;  out = d_class_lut ? d_class_lut[c] : c
 * ids < 0 (padding) or >= nrows give -1.  d_row_class int32 [nrows] (generalised
 * store, several rows per class) may be NULL (one row per class, row == prototype
 * index); d_class_lut int64 [nlut] maps memory label index -> classifier class id.
 */
int ac_rows_to_class(const int64_t* d_I, int64_t n, const int32_t* d_row_class,
                     int64_t nrows, const int64_t* d_class_lut, int nlut,
                     int64_t* d_out, ac_stream_t stream);

/* Deterministic synthetic rows: row r, col c = unit-normalised
 * N(0,1) from a counter-based generator keyed on (seed, row_offset + r, c).
 * Bit-identical to oracle/synth.py.  d_out [n, ld] fp32. */
int ac_synth_unit_rows(float* d_out, int64_t n, int64_t ld, int D,
                       uint64_t seed, int64_t row_offset, ac_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  H: AdaptiveHead  (models.py:30-98) + EWC (ewc.py) + AdamW
 * ------------------------------------------------------------------------- */

/* nn.Linear (models.py:49-69 in the head; the transformers Linear layers behind classifier.py:1271 in the
 * encoder) with a fused epilogue:
 *   C[M,N] = act(A[M,K] @ W[N,K]^T + bias[N]) (+ residual[M,N])
 * W is the torch nn.Linear layout.  act: 0 none, 1 ReLU, 2 GELU(erf).  fp32 in, fp32 out; products on the
 * fp32-input MFMA (exact fma chains) or, for M >= 192 under AC_GEMM_BF16X3 (the default, see
 * ac_gemm_set_arith), as the exact three-term bf16 split -- fp32-grade either way.  lda/ldw/ldc in elements. */
int ac_linear_f32(const float* d_A, int64_t lda, const float* d_W, int64_t ldw,
                  const float* d_bias, const float* d_residual, int64_t ldr,
                  float* d_C, int64_t ldc, int M, int N, int K, int act,
                  ac_stream_t stream);

/* General fp32 GEMM C = alpha * op(A) @ op(B) + beta * C: what autograd runs for the head's Linear layers in
 * loss.backward() (classifier.py:1499, :345): dW = dY^T X, dX = dY W.  transA/transB: 0 = as stored, 1 = T. */
int ac_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                const float* d_A, int64_t lda, const float* d_B, int64_t ldb,
                float beta, float* d_C, int64_t ldc, ac_stream_t stream);

/* Arithmetic of the large-M (M >= 192) NT GEMMs behind ac_linear_f32 / ac_bert_encode_cls / ac_head_*:
 *   AC_GEMM_F32     fp32-input MFMA (v_mfma_f32_32x32x2_f32): bitwise an fp32 fma chain.
 *   AC_GEMM_BF16X3  every fp32 operand split exactly into three bf16 terms (x = h + m + l), six bf16 MFMA
 *                   products (hh, hm, mh, mm, hl, lh) accumulated in fp32: per-product error <= 2^-26,
 *                   i.e. fp32-grade results (tests/test_gemm_split_gpu.py bounds it against fp64), at
 *                   6/16 of the matrix-pipe time.  Non-finite operands yield NaN.
 *   AC_GEMM_F16X2   OPT-IN, not fp32-exact on its inputs: in the BERT encoder's token-row GEMMs (ac_bert_encode_cls* with the
 *                   fp16 weight planes of ac_bert_weights present) every operand is rounded to TWO fp16 terms of x 2^s
 *                   (22 significant bits; s = 6 for activations, 10 for weights) and a product is three fp16 MFMAs
 *                   (lh, hl, hh) in fp32: per-product error <= 3 * 2^-22 |a||b| -- inside the a-priori bound K 2^-24
 *                   sum|a||b| of an fp32 dot product for K >= 12, but several times the error the fp32 MFMA or the bf16x3
 *                   split actually make (tests/test_gemm_f16x2_gpu.py measures all three against fp64) -- at half the
 *                   matrix-pipe time of bf16x3.  An operand beyond the fp16 range at its scale (|activation| >= 1023.5,
 *                   |weight| >= 63.97) turns the rows it feeds into NaN, never into a wrong finite number; the Python
 *                   encoder repeats such a call in bf16x3.  Everywhere else (head, ModernBERT, small shapes, ac_linear_*)
 *                   this mode IS bf16x3.
 * The reference computes these products with torch fp32 matmuls (transformers BertModel called at
 * classifier.py:1271; nn.Linear in models.py:49-80).  Process-wide; the initial value comes from the
 * environment variable AC_GEMM_ARITH ("f32" | "bf16x3" | "f16x2"; default bf16x3). */
#define AC_GEMM_F32 0
#define AC_GEMM_BF16X3 1
#define AC_GEMM_F16X2 2
/* ---- TEST HOOKS -------------------------------------------------------------------------------------------------------------
 * ac_gemm_set_arith, ac_gemm_set_variant, ac_gemm_debug_stamps, ac_gemm_set_pipe_table(_f16), ac_gemm_set_krot,
 * ac_gemm_set_ln_fusion and ac_set_persistent_kernels(mask >= 0) change PROCESS-WIDE state.  They exist for tests, A/B runs and
 * the scripts under tools/, and are inert in a process that did not set AC_TEST_HOOKS=1 in its environment before loading the
 * library: they return AC_EUNSUPPORTED (ac_set_persistent_kernels: the unchanged mask) and change nothing.  The product never
 * calls them: what a call computes is decided by its arguments (ac_bert_config / ac_modernbert_config *_opt words, AC_LOSS_STEPWISE,
 * AC_BERT_LAYERED) and by configuration variables read once at load (AC_GEMM_ARITH, AC_LN_FUSION, AC_GEMM_VARIANT,
 * AC_HEAD_PERSISTENT, AC_BERT_SMALL, AC_ACTIVE_CUS, AC_EXCHANGE_FENCES, AC_KNN_THR_EXACT) or at the call (the A/B switches of
 * equivalent forms, all default ON: AC_QKV_ATTN_FUSION, AC_QKV_ATTN_EXCHANGE, AC_BERT_TAIL_FUSED, AC_GEMM_FEWTILES; in the Python
 * host layer AC_BERT_UNPAD_ONE_CALL, AC_PREDICT_POST) -- SURVEY 8b: no global mutable state besides the per-thread error
 * string.  Queries (ac_gemm_get_arith, ac_set_persistent_kernels(-1), the *_launches counters) always answer. */
int ac_gemm_set_arith(int mode);
int ac_gemm_get_arith(void);
/* Diagnostic / A-B switch for the large-M pre-split GEMM: 0 = default dispatch (per-shape choice between the two-buffer
 * tile kernels of gemm.hip and the ring-staged kernels of gemm_pipe.hip), 1 = two-buffer tile kernels only,
 * >= 1000 = one ring configuration tm * 1000 + wmw * 100 + ring depth * 10 + pipelining (tools/gemm_bench.hip).
 * Env AC_GEMM_VARIANT sets the initial value. */
int ac_gemm_set_variant(int variant);
/* Diagnostic: while d_buf is non-null the ring-staged GEMM kernels (gemm_pipe.hip) write 4 shader-clock stamps per
 * workgroup -- start, ring filled, k-loop done, stores drained -- to d_buf[4 * workgroup] (launches whose grid exceeds
 * capacity_workgroups skip it).  Used by tools/gemm_bench.hip to see where a launch's time goes. */
int ac_gemm_debug_stamps(unsigned long long* d_buf, int64_t capacity_workgroups);
/* Tuning / A-B: "NxK=cfg;NxK=cfg;..." overrides the built-in per-shape choice between the two-buffer tile kernels (cfg 0) and
 * one ring configuration (cfg as in ac_gemm_set_variant) for planes GEMMs with N output columns and inner dimension K;
 * "" = two-buffer kernels everywhere; NULL restores the built-in table (tools/encode_ab.py). */
int ac_gemm_set_pipe_table(const char* spec);
/* Experiment switch of the ring-staged kernels (default 0).  1: workgroups on XCD x start their k-loop at stage x * nk / 8 and
 * wrap around, so the 8 XCDs fetch 8 different slices of the operands at any moment (one-round launches run in lockstep).  Measured
 * in the encoder at BASELINE configs[1]: no difference (4.89 vs 4.91 ms) -- the ring already hides those misses; kept for the
 * A/B (tools/encode_krot_ab.py).  With it a tile's fp32 accumulation order depends on its XCD (rounding-level differences
 * between shapes); 0 keeps every output element the same sum in the same order.  Env AC_GEMM_KROT sets the initial value. */
int ac_gemm_set_krot(int on);
/* The same for the fp16x2 kernels of AC_GEMM_F16X2 (a shape the table does not name, or names with cfg 0, keeps the built-in
 * choice -- there is no two-buffer kernel for fp16x2 operands); the configurations built for fp16x2 only are listed in
 * gemm_pipe.hip (AC_PIPE_CONFIGS_F16). */
int ac_gemm_set_pipe_table_f16(const char* spec);
/* The BERT encoder folds `x = LayerNorm(x + A W^T + b)` (BertSelfOutput / BertOutput, transformers modeling_bert.py) into
 * the epilogue of the attention-output and FFN2 GEMMs when the launch is one round of 128 x 128 tiles, one per CU: the
 * tiles of a 128-row panel exchange per-row (mean, M2) partials and each normalises its own block (gemm_pipe.hip).
 * 0 keeps the LayerNorms as separate launches (A/B, tests); default 1; env AC_LN_FUSION=0 sets the initial value.
 * 2 = on, with a starved exchange: every tile waits for an arrival that never comes (tests of the give-up path only). */
int ac_gemm_set_ln_fusion(int on);
int64_t ac_gemm_ln_fusion_launches(void);      /* fused launches of this process so far (tests: "did the fused path run") */
/* The BERT encoder also folds the self-attention (modeling_bert.py:111-203) into the epilogue of the fused QKV projection when
 * the batch's sequences lie row after row (packed, or unpacked without a mask), the head dimension is 64 and no sequence is
 * longer than 64 tokens: an output tile is 256 token rows x one head's q | k | v, the sequences inside it are finished from LDS
 * with the instructions of the stand-alone attention kernel (bit-identical context rows), and the fp32 [T, 3H] round trip plus
 * the attention launch disappear (gemm_pipe.hip EPI_QKV_ATTN).  Env AC_QKV_ATTN_FUSION=0 keeps the two launches (A/B, tests).
 * Diagnostic: launches of the fused form by this process so far. */
int64_t ac_gemm_qkv_attn_launches(void);

/* The persistent one-launch kernels of the latency-bound ends of the path are chosen automatically when the shape fits;
 * this switch (A/B tests, diagnosis) turns them off or on process-wide.  mask bit 0: ac_head_train_step / _epoch through
 * head_epoch.hip (otherwise the step-by-step launches); bit 1: ac_bert_encode_cls with <= 32 token rows through
 * bert_small.hip (otherwise the layer-by-layer kernels).  Returns the previous mask; mask < 0 only queries.
 * Initial value 3, or env AC_HEAD_PERSISTENT / AC_BERT_SMALL = 0 to clear a bit. */
int ac_set_persistent_kernels(int mask);

/* Measurement aid (bench.py `value_sustained`): one stamp per XCD of (shader clock counter s_memtime, 100 MHz real-time counter
 * s_memrealtime) into d_out16[2 * xcd + {0, 1}] (16 x uint64, zeroed first; an XCD no workgroup landed on stays 0).  Two stamps
 * on one stream bracket a region: its average shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz -- what the DVFS of
 * MI355X_MICROARCH.md ("give-back") left of the nominal 2.4 GHz while the region ran.  Nothing of the hot path calls it. */
int ac_clock_stamp(unsigned long long* d_out16, ac_stream_t stream);

/* Diagnostic: resident workgroups per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor) of the LDS-tiled
 * GEMM kernels.  kernel: 0 = fp32-MFMA tile, 1 = bf16x3 split-in-kernel, 2 = bf16x3 planes; tm: 1 = 64-row,
 * 2 = 128-row tile. */
int ac_gemm_occupancy(int kernel, int tm, int* blocks_per_cu);

/* Pre-split operands for AC_GEMM_BF16X3.  X[rows, K] fp32 (K % 8 == 0, 16-byte aligned rows) -> three bf16
 * planes h, m, l with X == h + m + l (to 2^-27 |x|), stored k-slot-major
 *     planes[p][k / 8][row][k % 8]        (uint16 units; 3 * rows * K in total)
 * so the GEMM stages MFMA fragments with direct global->LDS loads.  Weights are split once at load
 * (adaptive_classifier/encoder.py); activations are split inside the GEMM unless a producer already
 * emitted planes. */
int ac_split_bf16x3(const float* d_X, int64_t ldx, int64_t rows, int K,
                    uint16_t* d_planes, ac_stream_t stream);

/* Operand planes for AC_GEMM_F16X2: X[rows, K] fp32 -> two fp16 planes h, l of X 2^scale_log2 (h = fp16(x 2^s), l = fp16(x 2^s
 * - h), round to nearest even) in the layout of ac_split_bf16x3's first two planes: planes[p][k / 8][row][k % 8], p = 0, 1,
 * plane stride rows * K (2 * rows * K uint16 in total).  scale_log2: 10 for weights (what ac_bert_weights.*_wh hold), 6 for
 * activations (what the encoder's producers emit). */
int ac_split_f16x2(const float* d_X, int64_t ldx, int64_t rows, int K, int scale_log2,
                   uint16_t* d_planes, ac_stream_t stream);

/* C[M,N] = act(A W^T + bias) (+ residual) on fp16x2 planes (A at scale 2^6, W at 2^10), the ring-staged kernels of
 * gemm_pipe.hip only: M >= 192, K % 32 == 0, K >= 64 (AC_EUNSUPPORTED otherwise).  d_C fp32 rows, or -- d_C_planes non-NULL,
 * N % 8 == 0, no residual, act 0 or 2 -- the fp16x2 activation planes of the next GEMM instead.  Test / tuning entry: the
 * encoder calls the same internal function. */
int ac_linear_f16x2(const uint16_t* d_A_planes, const uint16_t* d_W_planes, const float* d_bias,
                    const float* d_residual, int64_t ldr, float* d_C, int64_t ldc, uint16_t* d_C_planes,
                    int M, int N, int K, int act, ac_stream_t stream);

/* ac_linear_f32 with optional pre-split operands (either may be NULL; A planes require W planes).  The
 * planes are used when the arithmetic mode is AC_GEMM_BF16X3 and the shape takes the LDS-tiled path
 * (M >= 192, K % 32 == 0); otherwise the fp32 operands are read, so both must always be valid.
 * d_W_planes is ac_split_bf16x3 of W[N, K]; d_A_planes of A[M, K].
 * d_C_planes (optional): emit the result as the operand planes of a following GEMM (layout of
 * ac_split_bf16x3 of C[M, N]) INSTEAD of d_C; needs both operand planes, the pre-split path (else
 * AC_EUNSUPPORTED), N % 8 == 0, no residual.
 * act: 0 none, 1 relu, 2 erf-gelu, 3 = GeGLU over 32-column blocks (planes output only): GEMM columns
 * [64t, 64t+32) are inputs, [64t+32, 64t+64) their gates; the result has N / 2 columns,
 * out[:, 32t + j] = gelu(in_j) * gate_j. */
int ac_linear_bf16x3(const float* d_A, int64_t lda, const uint16_t* d_A_planes,
                     const float* d_W, int64_t ldw, const uint16_t* d_W_planes,
                     const float* d_bias, const float* d_residual, int64_t ldr,
                     float* d_C, int64_t ldc, uint16_t* d_C_planes,
                     int M, int N, int K, int act, ac_stream_t stream);

/* Flat parameter block of an AdaptiveHead with hidden dims [H1, H2]
 * (classifier.py:1241: H1 = D, H2 = D/2).  All six tensors live in ONE
 * contiguous fp32 buffer in state_dict order
 *   model.0.weight [H1,D] | model.0.bias [H1] | model.3.weight [H2,H1] |
 *   model.3.bias [H2] | model.6.weight [C,H2] | model.6.bias [C]
 * so that the optimizer step is one launch over P = ac_head_param_count(). */
typedef struct {
    int D, H1, H2, C;
} ac_head_dims;

int64_t ac_head_param_count(const ac_head_dims* dims);

int ac_head_workspace(const ac_head_dims* dims, int B, size_t* bytes);

/* models.py:71-80 in eval mode (dropout off): logits[B,C]. */
int ac_head_forward(const ac_head_dims* dims, const float* d_params,
                    const float* d_X, int64_t ldx, int B, float* d_logits,
                    void* d_ws, size_t ws_bytes, ac_stream_t stream);

/* One training forward+backward (classifier.py:1489-1499 / :332-345):
 * train-mode forward with inverted dropout p (mask bytes supplied by the
 * caller, 1 = keep, NULL = no dropout), mean cross-entropy against int64
 * labels, backward into d_grads (same flat layout as params).  d_loss gets
 * the scalar CE loss. */
int ac_head_fwd_bwd_ce(const ac_head_dims* dims, const float* d_params,
                       const float* d_X, int64_t ldx, const int64_t* d_y,
                       const uint8_t* d_mask1, const uint8_t* d_mask2,
                       float dropout_p, int B, float* d_loss, float* d_grads,
                       void* d_ws, size_t ws_bytes, ac_stream_t stream);

#define AC_REDUCE_SCRATCH_BYTES 8192

/* Loss applied to the head's output layer in the training entry points. */
#define AC_LOSS_CE          0  /* nn.CrossEntropyLoss on logits, int64 labels (classifier.py:1463) */
#define AC_LOSS_BCE_SIGMOID 1  /* nn.BCELoss on sigmoid(logits), float multi-hot targets (multilabel.py:361) */
#define AC_LOSS_CE_SIGMOID  2  /* CrossEntropyLoss on sigmoid(logits): what the reference's new-class loop
                                  computes when the head is a MultiLabelAdaptiveHead (classifier.py:337-339) */
/* OR-ed into loss_kind of ac_head_train_step / ac_head_train_epoch: this call takes the step-by-step launches even where the
 * persistent epoch kernel (head_epoch.hip) would apply -- a per-call, thread-safe form of ac_set_persistent_kernels(~1),
 * used to repeat an epoch whose persistent launch came back void (NaN loss: a grid barrier gave up on a shared device). */
#define AC_LOSS_STEPWISE 0x100

/* ac_head_fwd_bwd_ce generalised to the three losses (explicit dropout masks, parity path).
 * d_y int64 [B] for the CE kinds, d_targets float [B, ldt] for BCE. */
int ac_head_fwd_bwd_loss(const ac_head_dims* dims, const float* d_params,
                         const float* d_X, int64_t ldx, const int64_t* d_y,
                         const float* d_targets, int64_t ldt, int loss_kind,
                         const uint8_t* d_mask1, const uint8_t* d_mask2,
                         float dropout_p, int B, float* d_loss, float* d_grads,
                         void* d_ws, size_t ws_bytes, ac_stream_t stream);

/* torch.sigmoid over n floats (MultiLabelAdaptiveHead.forward, multilabel.py:41-43). */
int ac_sigmoid(const float* d_in, int64_t n, float* d_out, ac_stream_t stream);

/*
 * One whole training step of classifier.py:1485-1507 / :329-353 in a single call:
 * (optional) batch gather X[index], y[index] -> train-mode forward with in-kernel
 * counter-based dropout (seed; no mask tensors) -> CE -> backward -> EWC gradient
 * + clip_grad_norm_ + AdamW.  d_out[0] = CE loss, [1] = EWC penalty, [2] = grad
 * norm; if d_loss_accum != NULL, *d_loss_accum += CE + penalty (the epoch's
 * `total_loss += loss.item()` without a host sync).  d_grads: flat scratch for
 * the gradients.  loss_kind: AC_LOSS_* (d_y for the CE kinds, d_targets [rows, ldt]
 * for BCE; both indexed through d_index when given).  Workspace: ac_head_workspace.
 */
int ac_head_train_step(const ac_head_dims* dims, float* d_params, float* d_m,
                       float* d_v, float* d_grads, const float* d_X, int64_t ldx,
                       const int64_t* d_y, const float* d_targets, int64_t ldt,
                       int loss_kind, const int64_t* d_index, int B,
                       float dropout_p, uint64_t dropout_seed,
                       const float* d_fisher, const float* d_old,
                       float lambda_over_B, float max_grad_norm, float lr,
                       float beta1, float beta2, float eps, float weight_decay,
                       int step, float* d_out, float* d_loss_accum,
                       void* d_ws, size_t ws_bytes, ac_stream_t stream);

/*
 * One EPOCH of that loop in one call (classifier.py:1485-1507 / :329-353): batches are consecutive
 * `batch`-row slices of d_order [n_total] (the epoch's DataLoader order; the last one may be short), step i
 * uses dropout seed seed0 + i, AdamW step step0 + i (step0 >= 1) and EWC weight lambda_B / rows_i.  Same
 * kernels as ac_head_train_step; the workspace must cover `batch` rows.  *steps_done = ceil(n_total/batch).
 */
int ac_head_train_epoch(const ac_head_dims* dims, float* d_params, float* d_m,
                        float* d_v, float* d_grads, const float* d_X, int64_t ldx,
                        const int64_t* d_y, const float* d_targets, int64_t ldt,
                        int loss_kind, const int64_t* d_order, int64_t n_total, int batch,
                        float dropout_p, uint64_t seed0,
                        const float* d_fisher, const float* d_old, float lambda_B,
                        float max_grad_norm, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step0,
                        float* d_out, float* d_loss_accum,
                        void* d_ws, size_t ws_bytes, int* steps_done, ac_stream_t stream);

/* F.softmax(logits, dim=1) over [B, C] rows (classifier.py:435,1345). */
int ac_softmax_rows(const float* d_in, int B, int C, float* d_out,
                    ac_stream_t stream);

/* F.normalize(x, p=2, dim=1, eps=1e-12) over [B, D] rows (classifier.py:1450). */
int ac_l2_normalize_rows(const float* d_in, int64_t ldi, int B, int D,
                         float* d_out, int64_t ldo, ac_stream_t stream);

/* ewc.py:90-92: fisher += grad^2 * inv_num_batches over the flat block. */
int ac_fisher_accumulate(const float* d_grads, float inv_num_batches,
                         float* d_fisher, int64_t n, ac_stream_t stream);

/* ewc.py:96-116: lambda * sum(F * (p - p_old)^2) [/ batch_size]. */
int ac_ewc_loss(const float* d_params, const float* d_fisher,
                const float* d_old, int64_t n, float lambda_over_B,
                float* d_out_loss, void* d_scratch, ac_stream_t stream);

/*
 * The fused EWC-regularised AdamW step (classifier.py:337-351 +
 * torch.optim.AdamW + clip_grad_norm_):
 *   g_i  = grad_i + 2*lambda_over_B * F_i * (p_i - pold_i)     (EWC gradient; skipped if F NULL)
 *   coef = min(1, max_norm / (||g||_2 + 1e-6))                 (clip_grad_norm_)
 *   p   *= 1 - lr*wd ; m = b1 m + (1-b1) coef g ; v = b2 v + (1-b2)(coef g)^2
 *   p   -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * d_out[0] = EWC penalty lambda_over_B*sum F (p-pold)^2 (before the update),
 * d_out[1] = ||g||_2 (before clipping).  `d_scratch` >= AC_REDUCE_SCRATCH_BYTES
 * (also for ac_ewc_loss); reductions use a fixed order, so results are deterministic.
 */
int ac_ewc_adamw_step(float* d_params, const float* d_grads, float* d_m,
                      float* d_v, const float* d_fisher, const float* d_old,
                      int64_t n, float lambda_over_B, float max_grad_norm,
                      float lr, float beta1, float beta2, float eps,
                      float weight_decay, int step, float* d_out,
                      void* d_scratch, ac_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  E: encoder  (classifier.py:1271-1275 -> transformers BertModel.forward)
 * ------------------------------------------------------------------------- */

typedef struct {
    int hidden;        /* H  (768 bert-base, 1024 bert-large / e5-large-v2) */
    int layers;        /* L */
    int heads;         /* A, head dim = H / A */
    int intermediate;  /* I */
    int vocab;
    int max_pos;
    int type_vocab;
    float ln_eps;      /* 1e-12 */
    /* Per-call options: 0 = the process-wide default (ac_gemm_set_arith / ac_gemm_set_ln_fusion / ac_set_persistent_kernels
     * and their environment variables), otherwise the value + 1.  They hold for the calls this config is passed to and for
     * nothing else -- two encoders (or two classifiers on one encoder) with different options interleave freely in one
     * process (tests/test_encoder_gpu.py::test_per_call_options_do_not_leak_between_objects); the process-wide setters remain
     * as test / A-B hooks.  A zero-initialised struct means "defaults". */
    int gemm_arith_opt;   /* 1 + AC_GEMM_F32 | 1 + AC_GEMM_BF16X3 | 1 + AC_GEMM_F16X2 (fp16x2 additionally needs the *_wh planes) */
    int ln_fusion_opt;    /* 1 = LayerNorms as separate launches, 2 = fused into the GEMM epilogues where the launch allows it
                           * (3 = fused with a starved exchange: tests of the give-up path only) */
    int one_launch_opt;   /* 1 = never the one persistent launch for <= 32 token rows (like AC_BERT_LAYERED), 2 = allowed */
} ac_bert_config;

/* Device pointers to the weights, nn.Linear [out,in] layout, fp32.
 * Per-layer arrays have `layers` entries (host arrays of device pointers). */
typedef struct {
    const float* word_emb;      /* [vocab, H] */
    const float* pos_emb;       /* [max_pos, H] */
    const float* type_emb;      /* [type_vocab, H] */
    const float* emb_ln_g;      /* [H] */
    const float* emb_ln_b;
    const float* const* qkv_w;  /* [3H, H]  rows: Wq | Wk | Wv */
    const float* const* qkv_b;  /* [3H] */
    const float* const* ao_w;   /* [H, H] attention output dense */
    const float* const* ao_b;
    const float* const* ln1_g;  /* attention output LayerNorm */
    const float* const* ln1_b;
    const float* const* ff1_w;  /* [I, H] */
    const float* const* ff1_b;
    const float* const* ff2_w;  /* [H, I] */
    const float* const* ff2_b;
    const float* const* ln2_g;  /* output LayerNorm */
    const float* const* ln2_b;
    /* Optional (all four or none; NULL = absent): ac_split_bf16x3 of the four weight matrices of every
     * layer.  With them and AC_GEMM_BF16X3 the token-row GEMMs run entirely on pre-split operands (the
     * LayerNorm / attention / GELU producers emit activation planes); without them operands are split
     * inside the GEMMs.  Same results to fp32 rounding either way. */
    const uint16_t* const* qkv_w3;
    const uint16_t* const* ao_w3;
    const uint16_t* const* ff1_w3;
    const uint16_t* const* ff2_w3;
    /* Optional (all four or none, and only next to the four above): ac_split_f16x2(scale_log2 = 10) of the same matrices.
     * Used when the arithmetic mode is AC_GEMM_F16X2 and every token-row GEMM of the call takes the ring-staged kernel
     * (>= 192 token rows); otherwise the call runs as under AC_GEMM_BF16X3. */
    const uint16_t* const* qkv_wh;
    const uint16_t* const* ao_wh;
    const uint16_t* const* ff1_wh;
    const uint16_t* const* ff2_wh;
} ac_bert_weights;

/*
 * memory.py:41-83 + :196-217 for ALL the examples one add_examples() call adds, one workgroup per class job,
 * ONE launch.  Per job: rows [n_old + n_new, ld] = the class's stored embeddings in list order followed by the
 * new ones in arrival order.  For t = 0 .. n_new-1: row n_old+t joins; if the class then holds more than `cap`
 * rows, the row farthest (L2) from the class mean (fp64 running sum, mean rounded to fp32, fp32 differences,
 * fp64 accumulation) is dropped -- exactly the sequential loop of the reference, whose cost there is O(n D)
 * host work per example.  sum [D] fp64: in = sum of the n_old rows, out = sum of the survivors.
 * alive [n_old + n_new]: 1 = kept.  dist [n_old + n_new] fp64: survivors' distance to the mean at the last
 * prune (the reference leaves the list sorted ascending by it).  dropped [n_new]: row dropped at step t or -1.
 * All job pointers are device pointers; the job array itself is passed twice: h_jobs (host copy, validated)
 * and d_jobs (the same bytes in device memory, read by the kernel).  n_old + n_new <= 8192, D <= 4096
 * (AC_EUNSUPPORTED beyond: callers fall back to per-example host logic).
 */
typedef struct {
    const float* rows; int64_t ld;
    int n_old, n_new, cap, reserved;
    double* sum; uint8_t* alive; double* dist; int32_t* dropped;
} ac_prune_job;

int ac_memory_add_prune(const ac_prune_job* h_jobs, const ac_prune_job* d_jobs, int njobs, int D,
                        ac_stream_t stream);

/*
 * classifier.py:447-480 (predict) and :1359-1384 (predict_batch): blend prototype scores with head
 * probabilities, normalise, stable-sort, keep the top k -- on the device, in fp64 like the reference's
 * Python floats.  d_scores / d_hit_class [b, kp]: ac_proto_scores output and the class id of every hit
 * (-1 = padding / unknown label; both NULL when there is no prototype store); d_head_probs [b, C] or NULL;
 * d_w_proto / d_w_head [C] fp64 per-class weights (0.7/0.3 for predict_batch; by training history for
 * predict); ncls_head = how many of the head's top classes vote (C for predict, min(k, C) for
 * predict_batch).  Output per query: n = min(k, classes present), then class ids and normalised scores in
 * descending order (ties: hits in distance order first, then head classes by probability).
 * C <= 2048 (AC_EUNSUPPORTED beyond: the host formula in classifier.py::_blend covers that).
 */
int ac_blend_topk(const float* d_scores, const int64_t* d_hit_class, int kp,
                  const float* d_head_probs, int C,
                  const double* d_w_proto, const double* d_w_head,
                  int ncls_head, int k, int b,
                  int32_t* d_out_n, int32_t* d_out_class, double* d_out_score,
                  ac_stream_t stream);

/*
 * The whole tail of a predict batch in ONE launch: ac_proto_scores + ac_rows_to_class + ac_softmax_rows + ac_blend_topk (the
 * same arithmetic in the same order; results bit for bit those of the four calls), and -- wait_host = 1 -- without a copy or
 * a stream synchronisation: the workgroup that finishes last copies the packed result from `out` (device memory) into h_out
 * (host-mapped memory from ac_host_alloc, at least as large) and publishes a completion flag the call spins on (falling
 * back to a blocking wait); the call returns when h_out can be read by the host.  h_out = NULL: asynchronous, result in `out`.
 *   d_dist / d_ids [b, kp]   the search's fp32 distances and int64 row ids (both NULL: no prototype hits); kp <= 1024
 *   d_row_class / nrows / d_class_lut / nlut   as ac_rows_to_class
 *   d_head [b, C]            the head's outputs (NULL: none); head_softmax = 1 applies F.softmax over them first
 *   d_w_proto / d_w_head / ncls_head / k       as ac_blend_topk;  C <= 2048
 *   out                      n[b] int32 | class[b, k] int32 | score[b, k] float64 at byte offsets 0, 4 b, round_up(4 b + 4 b k, 8)
 *                            (device memory, 16-byte aligned, out_bytes >= that size rounded up to 16; h_out receives the same bytes)
 * Replaces classifier.py:1347-1384 (prototype scores, label lookup, F.softmax, the blend and its sort) for a whole batch.
 * ac_host_alloc: page-locked, device-mapped, fine-grained host memory (hipHostMalloc coherent | mapped).
 */
int ac_predict_post(const float* d_dist, const int64_t* d_ids, int kp, const int32_t* d_row_class, int64_t nrows,
                    const int64_t* d_class_lut, int nlut, const float* d_head, int C, int head_softmax,
                    const double* d_w_proto, const double* d_w_head, int ncls_head, int k, int b, void* out,
                    size_t out_bytes, void* h_out, ac_stream_t stream);
int ac_host_alloc(size_t bytes, void** p);
int ac_host_free(void* p);

int ac_bert_workspace(const ac_bert_config* cfg, int b, int S, size_t* bytes);

/*
 * classifier.py:1271-1275: BertModel forward in eval mode on
 * (input_ids, token_type_ids, attention_mask) int64 [b,S], take
 * last_hidden_state[:,0,:], L2-normalise (F.normalize eps 1e-12) and write
 * d_out_unit_cls [b, ldo] fp32 -- directly usable as the kNN query block.
 * d_type_ids may be NULL (all zero); d_mask may be NULL (all ones).
 */
int ac_bert_encode_cls(const ac_bert_config* cfg, const ac_bert_weights* w,
                       const int64_t* d_ids, const int64_t* d_type_ids,
                       const int64_t* d_mask, int b, int S,
                       float* d_out_unit_cls, int64_t ldo,
                       void* d_ws, size_t ws_bytes, ac_stream_t stream);

/* ac_bert_encode_cls with per-call options and a report of the path taken.
 *   opts & AC_BERT_LAYERED   never take the one-launch path of bert_small.hip (<= 32 token rows), whatever
 *                            ac_set_persistent_kernels says: the layer-by-layer kernels run.
 *   used_one_launch          (nullable, host) 1 if this call ran as the one persistent launch.
 * The one-launch kernel is launched without the cooperative-launch residency check (it saves ~30 us per single query);
 * if the device is shared with another compute process one of its grid barriers can give up after a bounded spin, in
 * which case the output rows are NaN.  ac_bert_one_launch_status reads that verdict for the LAST one-launch call that
 * used this workspace (same cfg, b, S): *aborted = 1 -> repeat the call with AC_BERT_LAYERED.  It synchronises the
 * stream (a 4-byte D2H); callers that inspect their final result for NaN instead need not call it. */
#define AC_BERT_LAYERED 1
int ac_bert_encode_cls_opts(const ac_bert_config* cfg, const ac_bert_weights* w,
                            const int64_t* d_ids, const int64_t* d_type_ids,
                            const int64_t* d_mask, int b, int S,
                            float* d_out_unit_cls, int64_t ldo,
                            void* d_ws, size_t ws_bytes, int opts, int* used_one_launch, ac_stream_t stream);
int ac_bert_one_launch_status(const ac_bert_config* cfg, int b, int S, const void* d_ws, size_t ws_bytes,
                              int* aborted, ac_stream_t stream);
/* Verdict of the fused-LayerNorm GEMM epilogues (ac_gemm_set_ln_fusion) of every layer-by-layer ac_bert_encode_cls[_opts|
 * _packed] call that used this workspace SINCE THE LAST ac_bert_ln_fusion_clear.  The verdict lives at offset 0 of the
 * workspace whatever (b, S) it is used with (so one read covers all the row chunks of a big batch): word 0 is set by the
 * kernels of the current call (its later fused launches then stop waiting at their first look at it), every call starts by
 * folding word 0 into the sticky word 1, _status reports word 0 | word 1.  *aborted = 1 when the tiles of a row panel did not
 * all arrive within the bounded wait (possible only if the device cannot hold one workgroup per CU at once, e.g. under a CU
 * mask, or shared with another compute process); the output rows of the affected call(s) are NaN then -- clear, then repeat
 * after ac_gemm_set_ln_fusion(0).  The one-launch path of <= 32 token rows has no such epilogue and never touches the words.
 * _status synchronises the stream (an 8-byte D2H); _clear is asynchronous.  A caller that wants verdicts clears a fresh
 * workspace once before its first use (uninitialised memory could read as a stale verdict; the encode calls themselves are
 * unaffected by that).  b, S of _status are unused (kept for source compatibility).
 * Replaces nothing in the reference (classifier.py:1271 has no failure mode of this kind). */
int ac_bert_ln_fusion_clear(void* d_ws, size_t ws_bytes, ac_stream_t stream);
int ac_bert_ln_fusion_status(const ac_bert_config* cfg, int b, int S, const void* d_ws, size_t ws_bytes,
                             int* aborted, ac_stream_t stream);

/*
 * Padding-free ("packed") form of the same forward.  The reference pads every text to the longest of the batch and
 * encodes the padding too (classifier.py:1259-1271); padded positions cannot influence real ones (additive -inf mask,
 * row-wise LayerNorm / FFN) and only the CLS rows are consumed, so they can be left out: the same unit-norm CLS vectors
 * from sum(len) token rows instead of b * S.
 *   ac_bert_pack   from the attention mask [b, S]: d_cu int32 [b + 1] (row offset of every sequence), d_tok_src int32
 *                  [b * S] (token seq * S + pos of every packed row) and d_info int32 [4] = {total rows, 1 if some row of
 *                  the mask is not a non-empty prefix of ones (then use ac_bert_encode_cls), longest sequence, 0}.
 *                  Asynchronous; the caller reads d_info back to size the encoder launch.
 *   ac_bert_encode_cls_packed   total_tokens / longest = d_info[0] / d_info[2]; workspace as ac_bert_workspace(b, S).
 */
int ac_bert_pack(const int64_t* d_mask, int b, int S, int32_t* d_cu, int32_t* d_tok_src, int32_t* d_info,
                 ac_stream_t stream);
int ac_bert_encode_cls_packed(const ac_bert_config* cfg, const ac_bert_weights* w,
                              const int64_t* d_ids, const int64_t* d_type_ids, int b, int S,
                              const int32_t* d_cu, const int32_t* d_tok_src, int total_tokens, int longest,
                              float* d_out_unit_cls, int64_t ldo,
                              void* d_ws, size_t ws_bytes, ac_stream_t stream);

/*
 * ac_bert_pack + ac_bert_encode_cls_packed as ONE call with no stream synchronisation in it (the predict paths' form).  One
 * workgroup derives the packing, the fused attention epilogue's row-tile table and the zeroed exchange words from d_mask and
 * reports {token rows, not-prefix flag, longest} into a host-mapped slot; the embedding kernel is launched at once over b * S
 * rows (the real count is read on the device), the host picks the report up while that kernel runs and sizes the GEMM launches
 * -- the separate form leaves the GPU idle for the D2H round trip (~27 us) and 9 small stream operations in front of the first
 * GEMM.  More than 32 token rows (fewer: ac_bert_encode_cls, the one persistent launch).
 *   clear_verdict     1 = the fused-LayerNorm verdict words start clean (what ac_bert_ln_fusion_clear does, without its memset);
 *                     0 = rolled like every encode call
 *   *total_tokens     the token rows the forward ran (b * S on the two padded paths)
 *   *path             AC_BERT_PATH_PACKED; AC_BERT_PATH_PADDED = every row of the mask is all ones: the [b, S] forward without a
 *                     mask; AC_BERT_PATH_PADDED_MASK = some row's ones are not a non-empty prefix: the [b, S] forward with d_mask
 * Same results as the separate calls, bit for bit.  Workspace as ac_bert_workspace(b, S).  Replaces classifier.py:1259-1275
 * (tokenizer output -> model forward -> CLS rows) like ac_bert_encode_cls.
 */
#define AC_BERT_PATH_PACKED      0
#define AC_BERT_PATH_PADDED      1
#define AC_BERT_PATH_PADDED_MASK 2
int ac_bert_encode_cls_unpad(const ac_bert_config* cfg, const ac_bert_weights* w, const int64_t* d_ids, const int64_t* d_type_ids,
                             const int64_t* d_mask, int b, int S, float* d_out_unit_cls, int64_t ldo, void* d_ws, size_t ws_bytes,
                             int clear_verdict, int* total_tokens, int* path, ac_stream_t stream);
/* measurement: nanoseconds the last ac_bert_encode_cls_unpad call spent waiting for the packing kernel's report (its launch
 * latency + run time + the host-mapped store); tools/r06_gap_probe.py */
long long ac_bert_unpad_last_wait_ns(void);

/* ---- ModernBERT encoder ("answerdotai/ModernBERT-base", the reference's other default) ----
 * transformers modeling_modernbert.py: token embeddings -> LayerNorm; `layers` pre-norm blocks
 *   x += Wo attn(rope(Wqkv norm(x)))        (layer 0 has no attn norm; layer l attends globally when
 *                                            l % global_every == 0, else within |q - k| <= local_window)
 *   x += Wo2 (gelu(u[:, :I]) * u[:, I:]),  u = Wi norm(x)                       (GeGLU, erf GELU)
 * then final LayerNorm; classifier.py:1272-1275 takes [:, 0, :] and L2-normalises.  Head dim 64.
 * Biases are optional everywhere (NULL = absent, ModernBERT's default). */
typedef struct {
    int hidden, layers, heads, intermediate;
    int vocab, max_pos;
    int global_every;     /* global_attn_every_n_layers (3) */
    int local_window;     /* config.sliding_window = local_attention / 2 (64): |q - k| <= local_window */
    float norm_eps;       /* 1e-5 */
    int gemm_arith_opt;   /* per-call GEMM arithmetic as in ac_bert_config: 0 = process default, else AC_GEMM_* + 1 (AC_GEMM_F16X2
                           * runs as bf16x3 here: this encoder builds no fp16 weight planes) */
} ac_modernbert_config;

typedef struct {
    const float* tok_emb;            /* [vocab, H] */
    const float* emb_norm_g;         /* [H] */
    const float* emb_norm_b;         /* [H] or NULL */
    const float* final_norm_g;
    const float* final_norm_b;
    /* RoPE tables [max_pos, 32] fp32: cos / sin of position * inv_freq, computed by the host exactly as
     * ModernBertRotaryEmbedding does (theta 160000 for global layers, 10000 for local ones) */
    const float* rope_cos_global; const float* rope_sin_global;
    const float* rope_cos_local;  const float* rope_sin_local;
    /* per-layer host arrays of device pointers (entries may be NULL where noted) */
    const float* const* attn_norm_g; /* [H]; NULL for layer 0 (Identity) */
    const float* const* attn_norm_b; /* NULL array or NULL entries = no bias */
    const float* const* wqkv;        /* [3H, H] rows q | k | v */
    const float* const* wqkv_b;
    const float* const* wo;          /* [H, H] */
    const float* const* wo_b;
    const float* const* mlp_norm_g;
    const float* const* mlp_norm_b;
    const float* const* wi;          /* [2I, H] rows input | gate -- or, when wi_interleaved32 != 0, permuted in
                                      * blocks of 64 = 32 input rows then their 32 gate rows (I % 32 == 0), which
                                      * lets the GeGLU fuse into the GEMM epilogue; wi_b permuted alike */
    const float* const* wi_b;
    const float* const* wo2;         /* [H, I] */
    const float* const* wo2_b;
    const float* zero_bias;          /* max(3H, 2I) zeros: stands in for absent biases */
    /* optional bf16x3 planes of the four weight matrices (ac_split_bf16x3), all or none */
    const uint16_t* const* wqkv3;
    const uint16_t* const* wo3;
    const uint16_t* const* wi3;
    const uint16_t* const* wo23;
    int wi_interleaved32;
} ac_modernbert_weights;

int ac_modernbert_workspace(const ac_modernbert_config* cfg, int b, int S, size_t* bytes);

/* ids / mask int64 [b, S] (mask NULL = all ones) -> unit-norm CLS embeddings d_out_unit_cls [b, ldo]. */
int ac_modernbert_encode_cls(const ac_modernbert_config* cfg, const ac_modernbert_weights* w,
                             const int64_t* d_ids, const int64_t* d_mask, int b, int S,
                             float* d_out_unit_cls, int64_t ldo,
                             void* d_ws, size_t ws_bytes, ac_stream_t stream);

/* Padding-free form (see ac_bert_encode_cls_packed): d_cu / d_tok_src / total_tokens / longest from ac_bert_pack on the
 * attention mask; RoPE positions are positions inside each sequence; identical CLS vectors.  Workspace as
 * ac_modernbert_workspace(b, S). */
int ac_modernbert_encode_cls_packed(const ac_modernbert_config* cfg, const ac_modernbert_weights* w,
                                    const int64_t* d_ids, int b, int S,
                                    const int32_t* d_cu, const int32_t* d_tok_src, int total_tokens, int longest,
                                    float* d_out_unit_cls, int64_t ldo,
                                    void* d_ws, size_t ws_bytes, ac_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  T: tokenizer  (self.tokenizer(texts, max_length, truncation=True, padding=True), classifier.py:1259-1265)
 * ------------------------------------------------------------------------- */

/* A BERT WordPiece vocabulary as an open-addressing hash table in device memory (built by the host wrapper):
 * slot of a piece = (h ^ (h >> 32)) & (slots - 1), linear probing, h = ac_wordpiece_hash(piece bytes, continuation). */
typedef struct ac_wordpiece_vocab {
    const uint64_t* keys;       /* [slots] stored hash, 0 = empty slot */
    const uint32_t* offs;       /* [slots] offset of the piece's bytes in blob ("##" prefix stripped) */
    const int32_t* ids;         /* [slots] token id, bit 30 set for continuation ("##...") pieces */
    const uint16_t* lens;       /* [slots] byte length ("##" prefix stripped) */
    const uint8_t* blob;        /* concatenated piece bytes */
    int slots;                  /* power of two */
    int max_piece_bytes;        /* longest piece */
    int unk_id, cls_id, sep_id, pad_id;
    int lower_case;             /* 1 = lower-case A-Z (uncased vocabularies) */
} ac_wordpiece_vocab;

/* FNV-1a 64 over the piece bytes ("##" folded into the initial state when continuation != 0); never returns 0. */
uint64_t ac_wordpiece_hash(const uint8_t* bytes, int len, int continuation);

/*
 * Tokenise b ASCII texts on the device exactly as transformers' BertTokenizer does (BertNormalizer clean_text +
 * lowercase, BertPreTokenizer, greedy longest-match WordPiece with [UNK] for unmatched / over-long words,
 * [CLS] ... [SEP], truncation to max_length, padding with pad_id):
 *   d_text / d_offsets  the texts' bytes back to back, int32 offsets [b + 1]; each text <= 4096 bytes, ASCII only,
 *                       free of literal special-token strings (the caller routes other texts to the host tokenizer)
 *   d_ids, d_mask       int64 [b, max_length] token ids (padded) and attention mask
 *   d_lens              int32 [b] sequence lengths incl. [CLS] / [SEP]
 */
int ac_wordpiece_encode(const uint8_t* d_text, const int32_t* d_offsets, int b,
                        const ac_wordpiece_vocab* vocab, int max_length,
                        int64_t* d_ids, int64_t* d_mask, int32_t* d_lens, ac_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ACAMD_H */
