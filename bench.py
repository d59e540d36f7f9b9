"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 either under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment) or plain `python bench.py --gpus N`, which spawns its own N ranks (launch_ranks).

metric   predict() queries/sec + kNN GB/s vs HBM roofline, 768-d
N = 1    step = one predict() pass over one batch: BERT-base encoder (random init, fp32) on 256 pre-tokenised
         synthetic texts (<= 32 tokens, ragged) -> exact L2 top-16 over the 100k x 768 prototype store -> AdaptiveHead
         (768-768-384-4) -> the reference's predict_batch blend -> Python result lists.  This is
         BASELINE.json configs[1]; token ids and prototypes are resident in HBM before the timed region.
N > 1    one process per GPU.  step = one 4096-query batch of BASELINE configs[2]: the 10M x 768 store row-sharded over the
         ranks, k = 32, every rank owns 4096 / N queries: all-gather(queries) + local exact search + all-to-all(exact fp64
         distance, id) over RCCL + merge of the rank's own block.  value = 4096 * K / max-over-ranks time
         ("scaling": "strong"; rank 0 also measures the one-GPU form of the same job inside the run).  configs[1] data
         parallel (every rank encodes its own 256 texts, 100k rows sharded) is carried as `configs1_weak`, and configs[4] (e5-large-v2
         architecture, 2M x 1024 store row-sharded, 1024 texts per step data parallel: the BASELINE quotes it on 4 GPUs) as
         `configs4_data_parallel_sharded` (--no-extras skips it).
roofline the kNN distance sweep in its HBM-bound regime (the north star's roofline target): knn_sweep
         over 10M x 768 fp32 rows (30.7 GB) with 16 resident queries, timed with HIP events recorded
         around that kernel on its own stream (ac_knn_set_profile_events).  algorithmic bytes = N*D*4.
         At N > 1 the 10M rows are sharded over the ranks (BASELINE configs[2]): every rank sweeps its shard,
         achieved = total bytes / slowest rank's kernel time, peak = N x 8 TB/s.
         The encoder GEMM chain and the batched kNN of the timed step are MFMA-bound; their achieved
         TFLOP/s are reported next to it.  The encoder's large GEMMs default to the bf16x3 split arithmetic
         (every fp32 operand = h + m + l exactly, six bf16 MFMA products, fp32 accumulate: fp32-grade, see
         tests/test_gemm_split_gpu.py), whose matrix-pipe ceiling is 2500 / 6 = 416.7 fp32-equivalent
         TFLOP/s; the same timed loop is repeated with the fp32-input MFMA arithmetic (AC_GEMM_F32,
         157.3 TFLOP/s peak) and reported as config.value_f32_mfma.
cpu_baseline  kind "reference": the UNMODIFIED reference's predict_batch (classifier.py:1308-1388; staged byte for byte under
         oracle/_ref/ref_ac, Hub -> oracle/hub_standin.py, faiss -> a one-thread-per-query fp32 scan as faiss does for nq = 1) on
         the host cores, one or two passes over the same 256-text batch, rank 0, N = 1 only.  `cpu_baseline_port` keeps the
         oracle port of the step (transformers BertModel fp32 + C brute-force kNN over all cores + torch head) beside it.
value_sustained  (config) the timed loop run for >= 2.5 s with the observed shader clock (ac_clock_stamp), next to the
         20-step `value`.
extras   (N = 1, outside the timed region of `value`; --no-extras skips them) the other BASELINE configs in reduced form,
         so that one default run measures every config: `predict_from_text` (predict_batch on raw strings, tokenisation
         included, device WordPiece vs host tokenizer), `latency_ms_b1` (one predict() of one 16-token text + its CPU port),
         `cfg4` (configs[4] on one GPU: e5-large-v2 architecture, 2M x 1024 store, batch 1024, 5 steps, parity on 8 queries),
         `add_examples` (configs[3] at 6000 examples, as-wired EWC mode), `add_examples_with_encoder` (the same loop fed TEXTS:
         tokenizer + encoder in the loop, 6000 examples).  `--config latency | cfg4 | add_examples [--with-encoder]` run them at
         full size as their own JSON lines.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "adaptive-classifier_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
F32_MFMA_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TF = 2500.0     # v_mfma_f32_32x32x16_bf16 dense peak; the bf16x3 split spends 6 of them per product

BATCH, SEQ, NPROTO, DIM, KNN_K, NCLASS, VOCAB = 256, 32, 100_000, 768, 16, 4, 30522


def make_classifier(dev, rank, world):
    from adaptive_classifier import AdaptiveClassifier, AdaptiveHead
    from adaptive_classifier import index as ix
    from adaptive_classifier.encoder import HipBertEncoder
    from adaptive_classifier.sharded import ShardedSearch, shard_bounds
    from transformers import BertConfig, BertModel
    cfg = BertConfig()                                  # bert-base-uncased architecture
    torch.manual_seed(0)
    hf = BertModel(cfg, add_pooling_layer=False).eval()
    enc = HipBertEncoder(hf, device=dev)
    clf = AdaptiveClassifier("bert-base-uncased(random-init)", device=str(dev), encoder=enc, tokenizer=None)
    labels = [f"c{i}" for i in range(NCLASS)]
    clf.label_to_id = {l: i for i, l in enumerate(labels)}
    clf.id_to_label = {i: l for i, l in enumerate(labels)}
    clf.training_history = {l: 25 for l in labels}
    clf.adaptive_head = AdaptiveHead(DIM, NCLASS, [DIM, DIM // 2]).to(dev).eval()
    lo, hi = shard_bounds(NPROTO, world, rank)
    rows = ix.synth_unit_rows(hi - lo, DIM, 1, row_offset=lo, device=dev)          # this rank's row shard
    row_labels = torch.arange(NPROTO, dtype=torch.int32) % NCLASS                  # replicated row->class map
    sharded = ShardedSearch(rows, hi - lo, DIM, lo, block_rows=BATCH) if world > 1 else None      # 256 texts on every rank: the fixed-batch path
    clf.memory.load_rows(rows, row_labels, labels, sharded=sharded)
    return clf, hf


def synthetic_tokens(dev, rank, full_length=False):
    """SURVEY 8d configs[1]: token ids ~U[1000, vocab), lengths ~U[8, 32] (full_length: every text 32 tokens)."""
    g = torch.Generator().manual_seed(1234 + rank)
    ids = torch.randint(1000, VOCAB, (BATCH, SEQ), generator=g)
    ids[:, 0] = 101
    lens = torch.randint(8, SEQ + 1, (BATCH,), generator=g)
    lens[0] = SEQ
    if full_length:
        lens[:] = SEQ
    mask = (torch.arange(SEQ)[None, :] < lens[:, None]).to(torch.int64)
    ids = ids * mask
    return ids.to(dev), torch.zeros_like(ids).to(dev), mask.to(dev)


def predict_step(clf, ids, types, mask):
    """predict_batch() after the tokenizer: encoder -> kNN -> head -> blend -> result lists (AdaptiveClassifier.predict_tokens)"""
    return clf.predict_tokens(ids, types, mask, k=KNN_K)


def time_stages(clf, ids, types, mask, reps=5):
    """HIP-event timing of the device stages of one step (current stream), launched as predict_tokens launches them: the encoder
    (ac_bert_encode_cls_unpad: packing + forward, no stream synchronisation), the search (distances + row ids), the head's forward,
    and ac_predict_post (scores, hit classes, softmax, blend, top-k: here in its asynchronous form, result left on the device)."""
    ev = lambda: torch.cuda.Event(enable_timing=True)
    out = {}
    e = [ev() for _ in range(5)]
    tot = np.zeros(4)
    for _ in range(reps):
        e[0].record()
        emb = clf.model.encode_cls(ids, types, mask, verify=False)
        e[1].record()
        D, I = clf.memory.search_raw(emb, KNN_K)
        e[2].record()
        head = clf._head_outputs(emb)
        e[3].record()
        clf._post_launch(D.contiguous(), I.contiguous(), clf.memory.class_map(clf.label_to_id, emb.device), head.contiguous(), KNN_K, False, host=False)
        e[4].record()
        torch.cuda.synchronize()
        tot += [e[i].elapsed_time(e[i + 1]) for i in range(4)]
    tot /= reps
    out["encode_ms"], out["knn_ms"], out["head_ms"], out["post_ms"] = [float(x) for x in tot]
    # same box, same process, same loop (2 x reps forwards back to back): the encoder as it ships, and with the self-attention as
    # its own launch (the round-5 form; AC_QKV_ATTN_FUSION is read at every call) -- what the attention epilogue of the QKV GEMM
    # (gemm_pipe.hip EPI_QKV_ATTN) is worth on THIS box
    def back_to_back():
        clf.model.encode_cls(ids, types, mask, verify=False)
        e[0].record()
        for _ in range(2 * reps):
            clf.model.encode_cls(ids, types, mask, verify=False)
        e[1].record()
        torch.cuda.synchronize()
        return float(e[0].elapsed_time(e[1]) / (2 * reps))
    out["encode_ms_back_to_back"] = back_to_back()
    os.environ["AC_QKV_ATTN_FUSION"] = "0"
    try:
        out["encode_ms_back_to_back_attention_as_its_own_launch"] = back_to_back()
    finally:
        del os.environ["AC_QKV_ATTN_FUSION"]
    return out


def step_parity(clf, hf, ids, types, mask, n_check=64, n_enc=8):
    """Correctness evidence carried by the bench line itself: the timed step's kNN ids for `n_check` of its queries vs
    the exact oracle on the same store (bit-exact bar), its encoder output vs transformers fp32 on CPU for `n_enc`
    sequences and its head logits vs torch CPU (1e-4 bar).  The oracle is the CHECKER here, never the thing timed."""
    from oracle import c_oracle
    with torch.no_grad():
        emb = clf.model.encode_cls(ids, types, mask)
        S, I, D = clf.memory.search_batch(emb, KNN_K)
        logits = clf.adaptive_head.forward_native(emb)
        torch.cuda.synchronize()
        n = clf.memory.index.ntotal
        P = clf.memory.index._store[:n, :DIM].cpu().numpy()
        Q = emb[:n_check].cpu().numpy()
        oD, oI = c_oracle.knn_l2_topk_batch(P, Q, KNN_K)
        got = I[:n_check].cpu().numpy()
        want_emb = torch.nn.functional.normalize(
            hf(input_ids=ids[:n_enc].cpu(), token_type_ids=types[:n_enc].cpu(), attention_mask=mask[:n_enc].cpu())
            .last_hidden_state[:, 0, :], dim=1)
        sd = {k: v.detach().cpu() for k, v in clf.adaptive_head.state_dict().items()}
        x = emb[:n_check].cpu()
        h = torch.relu(x @ sd["model.0.weight"].T + sd["model.0.bias"])
        h = torch.relu(h @ sd["model.3.weight"].T + sd["model.3.bias"])
        want_logits = h @ sd["model.6.weight"].T + sd["model.6.bias"]
    return {"checked_queries": int(n_check), "id_mismatches": int((got != oI).sum()),
            "dist_max_ulp": float(np.max(np.abs(D[:n_check].cpu().numpy() - oD) / np.spacing(np.maximum(oD, np.float32(1e-30))))),
            "oracle": "oracle/knn_faiss_forms.c exact fp64 (p-q)^2, ties to the lower id, over the full 100k-row store",
            "encoder_checked_sequences": int(n_enc),
            "encoder_max_abs_diff_vs_transformers_fp32": float((emb[:n_enc].cpu() - want_emb).abs().max()),
            "head_max_abs_diff_vs_torch_fp32": float((logits[:n_check].cpu() - want_logits).abs().max())}


def sweep_parity(P, n_rows, Q, out_ids, k):
    """ids of the roofline launch's resident queries vs the chunked exact oracle over the whole store
    (device -> host in 1M-row chunks, per-chunk batched oracle + merge)."""
    from oracle import c_oracle
    Qh = Q[:, :DIM].cpu().numpy()
    chunks = ((s, P[s:min(n_rows, s + 1_000_000), :DIM].cpu().numpy()) for s in range(0, n_rows, 1_000_000))
    oD, oI = c_oracle.knn_l2_topk_chunked(chunks, Qh, k)
    return {"checked_queries": int(Qh.shape[0]), "id_mismatches": int((out_ids.cpu().numpy() != oI).sum()), "rows": int(n_rows), "k": int(k)}


def sweep_roofline(dev, n_rows, full=True, parity=False):
    """knn_sweep alone over n_rows x 768: algorithmic bytes / kernel time (HIP events around the kernel).
    The headline entry uses 16 resident queries; `by_resident_queries` lists 1 / 8 / 16 / 32.
    full=False (the per-rank shard sweep at N > 1): only the 16-query measurement."""
    from adaptive_classifier import _native as nv
    from adaptive_classifier import index as ix
    k = 32
    P = ix.synth_unit_rows(n_rows, DIM, 1, device=dev)
    bytes_alg = n_rows * DIM * 4
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize()          # materialise the hipEvent handles

    def measure(nq, reps, prepared=None):
        Q = ix.synth_unit_rows(nq, DIM, 2, device=dev)
        ws = torch.empty(max(ix.knn_workspace_bytes(n_rows, DIM, nq, k), 0 if prepared is None else ix.knn_batch_workspace_bytes(n_rows, DIM, nq, k)),
                         dtype=torch.uint8, device=dev)
        stats = torch.zeros(4, dtype=torch.int32, device=dev)
        out = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
        for _ in range(2):
            ix.knn_l2_topk(P, n_rows, DIM, Q, k, out=out, workspace=ws, stats=stats, prepared=prepared)
        torch.cuda.synchronize()
        times, calls = [], []
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nv.lib().ac_knn_set_profile_events(e0.cuda_event, e1.cuda_event)
        try:
            for _ in range(reps):
                c0.record()
                ix.knn_l2_topk(P, n_rows, DIM, Q, k, out=out, workspace=ws, stats=stats, prepared=prepared)
                c1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
                calls.append(c0.elapsed_time(c1))
        finally:
            nv.lib().ac_knn_set_profile_events(None, None)
        if prepared is not None:
            keep["plane_ids"] = out[1].clone()
            keep["plane_form"] = int(stats[1].item())
        elif nq in (16, 32):
            keep["ids%d" % nq] = out[1].clone()
        if nq == 16 and parity and prepared is None:
            par["v"] = sweep_parity(P, n_rows, Q, out[1], k)
        return float(np.mean(times)), float(np.min(times)), float(np.mean(calls)), int(stats[0].item())

    par = {"v": None}
    keep = {"ids16": None, "ids32": None}
    ms, ms_min, call_ms, nfb = measure(16, 8)
    table = {}
    for nq in ((1, 8, 16, 32) if full else (16,)):
        t, _, c, _ = (ms, ms_min, call_ms, nfb) if nq == 16 else measure(nq, 4)
        table[str(nq)] = {"kernel_ms": t, "GBps": bytes_alg / t / 1e6, "frac": bytes_alg / t / 1e6 / HBM_PEAK_GBS,
                          "whole_call_ms": c}
    # BASELINE configs[2] on this one GPU: 4096 queries x the whole store, k = 32 (compute-bound regime, whole call)
    import glob
    batch = plane = None
    out16_ids = keep["ids16"]           # ids of the 16 roofline queries (= the first 16 of the 4096-query batch, same seed)
    if full:
        nqb = 4096
        Qb = ix.synth_unit_rows(nqb, DIM, 2, device=dev)
        prep = ix.prepare_store(P, n_rows, DIM)        # fp16 plane + row norms, once per store
        wsb = torch.empty(ix.knn_batch_workspace_bytes(n_rows, DIM, nqb, k), dtype=torch.uint8, device=dev)
        stb = torch.zeros(4, dtype=torch.int32, device=dev)
        outb = (torch.empty((nqb, k), dtype=torch.float32, device=dev), torch.empty((nqb, k), dtype=torch.int64, device=dev))
        ix.knn_l2_topk(P, n_rows, DIM, Qb, k, out=outb, workspace=wsb, stats=stb, prepared=prep)
        torch.cuda.synchronize()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        for _ in range(2):
            ix.knn_l2_topk(P, n_rows, DIM, Qb, k, out=outb, workspace=wsb, stats=stb, prepared=prep)
        b1.record(); torch.cuda.synchronize()
        bms = b0.elapsed_time(b1) / 2
        batch = {"workload": "BASELINE configs[2] on ONE GPU: %d x %d store, k=%d, batch %d" % (n_rows, DIM, k, nqb),
                 "path": "prepared store: fp16 GEMM-form proposals (one MFMA product) + fp64 re-rank / certificate (ac_knn_l2_topk_batch)",
                 "ms_per_batch": bms, "queries_per_s": nqb / bms * 1e3, "TFLOPs_fp32_equiv": 2.0 * nqb * n_rows * DIM / bms / 1e9,
                 "exact_fallback_queries": int(stb[0].item()),
                 "ids_equal_fp32_sweep_subset": bool(torch.equal(outb[1][:16], out16_ids)) if out16_ids is not None else None}
        del Qb, wsb, outb
        # the same store's fp16 plane swept ONCE for 1 .. 64 resident queries (knn_plane_sweep, round 4): its own entry, priced on
        # ITS algorithmic bytes N * D * 2 -- never against the fp32 sweep's N * D * 4
        plane_bytes = n_rows * DIM * 2
        ptab, ids_ok = {}, None
        for nq in (1, 16, 32, 48, 64):
            t, tmin, c, fb = measure(nq, 4, prepared=prep)
            ptab[str(nq)] = {"kernel_ms": t, "GBps": plane_bytes / t / 1e6, "frac": plane_bytes / t / 1e6 / HBM_PEAK_GBS, "whole_call_ms": c,
                             "exact_fallback_queries": fb, "form": keep["plane_form"]}
            if nq in (16, 32) and keep["ids%d" % nq] is not None:
                same = bool(torch.equal(keep["plane_ids"], keep["ids%d" % nq]))
                ids_ok = same if ids_ok is None else (ids_ok and same)
        t16 = ptab["16"]["kernel_ms"]
        plane = {"bound": "hbm", "achieved": plane_bytes / t16 / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": plane_bytes / t16 / 1e6 / HBM_PEAK_GBS,
                 "traffic": None, "kernel": "knn_plane_sweep (prepared store: tile-major fp16 plane by non-temporal whole-line loads straight into "
                                            "MFMA fragments, query tile resident in LDS; exact result through the fp64 re-rank + certificate)",
                 "rows": n_rows, "dim": DIM, "resident_queries": 16, "algorithmic_bytes_per_launch": plane_bytes,
                 "note": "bytes = N * D * 2 (the fp16 plane; + 4 B of |p|^2 per row = 0.26 %%); the fp32 rows are read only for the k' re-ranked "
                         "candidates per query.  Speed-up of the sweep over the fp32 form at 16 queries: %.2fx" % (ms / t16),
                 "avg_kernel_ms": t16, "by_resident_queries": ptab, "ids_equal_fp32_sweep": ids_ok}
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "knn_plane_sweep_pmc.json"))):
            pj = json.load(open(f))
            if pj.get("rows") == n_rows and pj.get("dim") == DIM:
                plane["traffic"] = pj["hbm_read_bytes_per_launch_corrected"] + pj["hbm_write_bytes_per_launch"]
                plane["traffic_source"] = os.path.relpath(f, ROOT)
        del prep
    # HBM traffic per launch from the committed rocprofv3 PMC pass (FETCH_SIZE x2 gfx950 correction,
    # profiles/<round>/knn_sweep_pmc.json); null when no pass exists for this problem size.
    traffic, traffic_src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "knn_sweep_pmc.json"))):
        pj = json.load(open(f))
        if pj.get("rows") == n_rows and pj.get("dim") == DIM:
            traffic = pj["hbm_read_bytes_per_launch_corrected"] + pj["hbm_write_bytes_per_launch"]
            traffic_src = os.path.relpath(f, ROOT)
    del P
    torch.cuda.empty_cache()
    return {"bound": "hbm", "achieved": bytes_alg / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_alg / ms / 1e6 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "knn_sweep_ring (16 resident queries: rows through wave-private LDS rings by non-temporal DMA, query "
                      "fragments in registers; knn_sweep<1> with AC_KNN_RING=0)", "rows": n_rows, "dim": DIM, "resident_queries": 16,
            "algorithmic_bytes_per_launch": bytes_alg, "avg_kernel_ms": ms, "min_kernel_ms": ms_min,
            "whole_call_ms": call_ms, "exact_fallback_queries": nfb, "parity": par["v"], "by_resident_queries": table,
            "batch4096": batch, "fp16_plane": plane}


def cpu_baseline(hf, clf, rows_dev, sample=2048, chunk=64):
    """Oracle port of the same step on the host cores (bounded sample)."""
    from oracle import c_oracle, head_oracle
    cores = c_oracle.usable_cores()           # the box's cgroup quota, not os.cpu_count()
    torch.set_num_threads(cores)
    c_oracle.set_threads(cores)
    g = torch.Generator().manual_seed(99)
    ids = torch.randint(1000, VOCAB, (sample, SEQ), generator=g)
    mask = torch.ones_like(ids)
    P = rows_dev[:, :DIM].cpu().numpy()
    head = head_oracle.make_head(DIM, NCLASS).eval()
    t0 = time.perf_counter()
    with torch.no_grad():
        emb = torch.cat([torch.nn.functional.normalize(
            hf(input_ids=ids[i:i + chunk], attention_mask=mask[i:i + chunk]).last_hidden_state[:, 0, :], dim=1)
            for i in range(0, sample, chunk)])
        t1 = time.perf_counter()
        D, I = c_oracle.knn_l2_topk_f32(P, emb.numpy(), KNN_K)       # what faiss's nq<20 path computes, all cores
        t2 = time.perf_counter()
        s = np.exp(-D)
        torch.softmax(torch.from_numpy(s), dim=1)
        torch.softmax(head(emb), dim=1)
    t3 = time.perf_counter()
    # What the reference's own loop would add: predict_batch searches ONE query at a time (classifier.py:1329-1334 ->
    # memory.py:114), and faiss's IndexFlat parallelises over queries -- a single-query search scans the store on ONE thread
    # (faiss utils/distances.cpp, exhaustive_L2sqr_seq).  Timed here on a few queries with the port's scan pinned to one thread.
    nt = c_oracle.num_threads()
    c_oracle.set_threads(1)
    nseq = 8
    t4 = time.perf_counter()
    for i in range(nseq):
        c_oracle.knn_l2_topk_f32(P, emb[i:i + 1].numpy(), KNN_K)
    per_query = (time.perf_counter() - t4) / nseq
    c_oracle.set_threads(nt)
    enc_per_text = (t1 - t0) / sample
    return {"value": sample / (t3 - t0), "unit": "queries/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{sample} texts (batches of {chunk}) x S={SEQ}: transformers BertModel fp32 (torch CPU) + C fp32 brute-force kNN "
                      f"over {NPROTO}x{DIM} (OpenMP, {nt} threads) + torch head",
            "encode_s": t1 - t0, "knn_s": t2 - t1, "head_s": t3 - t2,
            "with_the_references_per_query_search": {
                "value": 1.0 / (enc_per_text + per_query), "unit": "queries/s", "knn_ms_per_query_one_thread": per_query * 1e3,
                "note": "the port above searches the whole batch with all cores, which flatters the CPU: the reference searches one "
                        "query at a time and faiss scans the store for a single query on one thread; this figure = the port's "
                        f"encoder rate + {nseq} single-thread single-query scans (a model of the reference's loop, not a run of it: "
                        "faiss is not installable here)"}}


import contextlib


@contextlib.contextmanager
def staged_reference():
    """The UNMODIFIED reference package (oracle/_ref/ref_ac, staged byte for byte by oracle/stage_ref.py, sha256-checked here on
    every use) importable as `ref_ac`, with the two things that cannot exist offline stood in for: the Hub (oracle/hub_standin.py)
    and faiss -- an IndexFlatL2 with the protocol memory.py uses, on the C oracle's fp32 scan, ONE thread per single-query search
    (what faiss's IndexFlat does for nq = 1: exhaustive_L2sqr_seq parallelises over queries only, and nq = 1 is what the reference's
    loops ask for, classifier.py:1329-1334 -> memory.py:114).  Yields (ref_ac, cores) or None when oracle/_ref is not staged."""
    import hashlib
    import types
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    man_path = os.path.join(ref_dir, "MANIFEST.json")
    if not os.path.exists(os.path.join(ref_dir, "ref_ac", "classifier.py")) or not os.path.exists(man_path):
        yield None
        return
    man = json.load(open(man_path))
    for rel, ent in man.items():                       # byte-identical to the reference tree it was staged from
        if hashlib.sha256(open(os.path.join(ref_dir, rel), "rb").read()).hexdigest() != ent["sha256"]:
            raise SystemExit("bench.py: oracle/_ref/%s is not the staged reference file (sha256 mismatch)" % rel)
    from oracle import c_oracle, hub_standin
    cores = c_oracle.usable_cores()
    torch.set_num_threads(cores)

    class FlatL2:                                       # the faiss protocol memory.py uses, on the C oracle's fp32 scan
        def __init__(self, d):
            self.d, self._x = int(d), np.zeros((0, int(d)), np.float32)
        ntotal = property(lambda self: self._x.shape[0])

        def add(self, x):
            self._x = np.ascontiguousarray(np.concatenate([self._x, np.asarray(x, np.float32).reshape(-1, self.d)]))

        def search(self, x, k):
            # ONE thread for this scan only: liboracle and torch share the OpenMP runtime, so a lasting set_threads(1) would
            # also pin the reference's BertModel forward to one core
            c_oracle.set_threads(1)
            try:
                return c_oracle.knn_l2_topk_f32(self._x, np.ascontiguousarray(x, np.float32).reshape(-1, self.d), int(k))
            finally:
                c_oracle.set_threads(cores)

        def remove_ids(self, ids):
            keep = np.ones(self.ntotal, bool); keep[np.asarray(ids).reshape(-1)] = False
            self._x = self._x[keep]
    saved_faiss = sys.modules.get("faiss")
    shim = types.ModuleType("faiss")
    shim.IndexFlatL2 = FlatL2
    sys.modules["faiss"] = shim
    hub_standin.install()
    sys.path.insert(0, ref_dir)
    try:
        import ref_ac
        assert os.path.realpath(ref_ac.__file__).startswith(os.path.realpath(ref_dir))
        yield ref_ac, cores
    finally:
        sys.path.remove(ref_dir)
        c_oracle.set_threads(cores)
        if saved_faiss is None:
            sys.modules.pop("faiss", None)
        else:
            sys.modules["faiss"] = saved_faiss
        hub_standin.uninstall()


REF_NOTES = {"hub": "oracle/hub_standin.py: bert-base-uncased ARCHITECTURE, seeded random init, synthetic WordPiece vocabulary "
                    "(texts tokenise to the bench's token ids)",
             "faiss": "NOT real faiss (not installable offline): IndexFlatL2 stand-in on oracle/knn_oracle.c's fp32 scan, one "
                      "thread per single-query search as faiss's IndexFlat does for nq = 1"}


def cpu_baseline_reference(ids, mask, rows_dev, seconds_hint=20.0):
    """`cpu_baseline` with kind "reference": the UNMODIFIED reference package (staged_reference above) running ITS `predict_batch`
    (classifier.py:1308-1388) on the host cores over the configs[1] workload: bert-base architecture (random init, through the
    offline Hub stand-in), the same 100k x 768 rows in its faiss index (row -> class map = its own `index_to_label`), its own
    AdaptiveHead, k = 16, batch_size = 256, texts whose tokenisation is exactly the bench's synthetic token ids.
    Returns None when oracle/_ref is not staged."""
    from oracle import hub_standin
    with staged_reference() as env:
        if env is None:
            return None
        ref_ac, cores = env
        clf = ref_ac.AdaptiveClassifier("bert-base-uncased", device="cpu", use_onnx=False)
        labels = [f"c{i}" for i in range(NCLASS)]
        clf.label_to_id = {l: i for i, l in enumerate(labels)}
        clf.id_to_label = {i: l for i, l in enumerate(labels)}
        clf.training_history = {l: 25 for l in labels}
        clf.adaptive_head = ref_ac.AdaptiveHead(DIM, NCLASS, [DIM, DIM // 2]).eval()
        P = rows_dev[:, :DIM].cpu().numpy()
        clf.memory.index.add(P)
        clf.memory.index_to_label = {i: labels[i % NCLASS] for i in range(P.shape[0])}
        clf.memory.updates_since_rebuild = 0
        ids_h, mask_h = ids.cpu(), mask.cpu()
        lens = mask_h.sum(1).tolist()
        # the tokenizer adds [CLS] / [SEP]: a text of len - 2 filler words tokenises to `len` ids (synthetic ids >= 1000 are fillers)
        texts = [hub_standin.text_for_ids(ids_h[i, 1:max(2, int(n) - 1)].tolist()) for i, n in enumerate(lens)]
        tok = clf.tokenizer(texts[:4], max_length=512, truncation=True, padding=True, return_tensors="pt")
        assert int(tok["attention_mask"][0].sum()) == int(lens[0]), (tok["attention_mask"].sum(1), lens[:4])
        clf.predict_batch(texts[:8], k=KNN_K, batch_size=8)                 # warm-up (allocator, thread pools)
        n, t0, out = 0, time.perf_counter(), None
        while n == 0 or (time.perf_counter() - t0 < seconds_hint * 0.5 and n < 4):
            out = clf.predict_batch(texts, k=KNN_K, batch_size=BATCH)
            n += 1
        dt = (time.perf_counter() - t0) / n
        assert len(out) == BATCH and all(len(p) >= 1 for p in out)
        # where the time goes (the reference's own methods, timed separately on the same batch)
        t1 = time.perf_counter(); emb = clf._get_embeddings(texts); t_enc = time.perf_counter() - t1
        t1 = time.perf_counter()
        for e in emb[:32]:
            clf.memory.get_nearest_prototypes(e, k=KNN_K)
        t_knn = (time.perf_counter() - t1) / 32
        return {"value": BATCH / dt, "unit": "queries/s", "cores": int(cores), "kind": "reference", "gc_frozen": __import__("gc").get_freeze_count() > 0,
                "sample": "%d x predict_batch(256 texts, k=%d, batch_size=256) of the unmodified reference (oracle/_ref/ref_ac = "
                          "/root/reference/src/adaptive_classifier, sha256-checked): tokenizer -> BertModel fp32 on torch CPU (%d "
                          "threads) -> per query get_nearest_prototypes over %d x %d rows -> head -> blend" % (n, KNN_K, cores, P.shape[0], DIM),
                "seconds_per_batch": dt, "encode_s_per_batch": t_enc, "knn_ms_per_query": t_knn * 1e3, **REF_NOTES}


def cpu_baseline_add_examples_reference(E, n_classes, cap=1000, seconds_hint=15.0):
    """configs[3]'s `cpu_baseline` with kind "reference": the UNMODIFIED reference's `add_examples` (classifier.py:132-200) on the
    host cores, fed the workload's pre-computed embeddings -- its `_get_embeddings` (the encoder call the product's
    `add_embeddings` loop does not contain either) is replaced ON THE INSTANCE by a lookup into the same embedding table, nothing
    else: label maps, `memory.add_example` per example (prototype update, prune at the cap), `_train_adaptive_head`
    (:1428-1522: DataLoader, dropout, CE, backward, clip, AdamW, ReduceLROnPlateau, early stopping) on everything stored, index
    rebuild.  BOUNDED SAMPLE: a 50 000-example run spends 92 % of its calls with the memory at the cap, where one reference call
    retrains on 4000 examples (seconds); so the memory is first filled to the cap through the reference's own
    `memory.add_example` (untimed) + one untimed call, then whole `add_examples` calls of 32 are timed for ~seconds_hint."""
    with staged_reference() as env:
        if env is None:
            return None
        ref_ac, cores = env
        torch.manual_seed(0)
        clf = ref_ac.AdaptiveClassifier("bert-base-uncased", device="cpu", use_onnx=False, config={"max_examples_per_class": cap})
        clf.model = None                                                    # (the encoder is not part of this loop)
        table = E.float().cpu()
        clf._get_embeddings = lambda texts: [table[int(t[1:])] for t in texts]
        labels = [f"c{i}" for i in range(n_classes)]
        for i, l in enumerate(labels):
            clf.label_to_id[l] = i
            clf.id_to_label[i] = l
        n_fill = cap * n_classes
        for i in range(n_fill):                                             # untimed: the reference's own bookkeeping up to the cap
            clf.memory.add_example(ref_ac.Example(f"t{i}", labels[i % n_classes], table[i]), labels[i % n_classes])
            clf.training_history[labels[i % n_classes]] = clf.training_history.get(labels[i % n_classes], 0) + 1
        pos = n_fill

        def one_call():
            nonlocal pos
            idx = list(range(pos, pos + 32))
            pos += 32
            t = time.perf_counter()
            clf.add_examples([f"t{i}" for i in idx], [labels[i % n_classes] for i in idx])
            return time.perf_counter() - t
        one_call()                                                           # untimed warm-up (builds the head, thread pools)
        times = []
        while not times or (sum(times) < seconds_hint and len(times) < 8):
            times.append(one_call())
        dt = sum(times) / len(times)
        stored = sum(len(v) for v in clf.memory.examples.values())
        return {"value": 32.0 / dt, "unit": "examples/s", "cores": int(cores), "kind": "reference", "gc_frozen": __import__("gc").get_freeze_count() > 0,
                "seconds_per_call": dt, "calls_timed": len(times), "stored_examples": stored,
                "sample": "%d x add_examples(32 examples) of the unmodified reference (oracle/_ref/ref_ac, sha256-checked; its "
                          "_get_embeddings replaced on the instance by a lookup into the workload's embedding table, as the product's "
                          "add_embeddings loop has no encoder call either) with its memory at the %d-per-class cap (%d stored): "
                          "memory.add_example x 32 -> _train_adaptive_head on everything stored (torch CPU, %d threads) -> "
                          "_rebuild_index" % (len(times), cap, stored, cores),
                "faiss": REF_NOTES["faiss"]}


def shader_clock_between(a, b):
    """Average shader clock (MHz) between two ac_clock_stamp buffers (host lists of 16 ints): per XCD d(s_memtime) /
    d(s_memrealtime) x 100 MHz, averaged over the XCDs stamped both times."""
    f = [(b[2 * x] - a[2 * x]) / (b[2 * x + 1] - a[2 * x + 1]) * 100.0 for x in range(8)
         if a[2 * x + 1] and b[2 * x + 1] and b[2 * x + 1] > a[2 * x + 1]]
    return (float(np.mean(f)), len(f)) if f else (None, 0)


def sustained_predict(step, batch, seconds=2.5, min_steps=50):
    """The timed loop of `value` run for >= `seconds` (the 20-step headline lasts ~0.1 s, a boost-clock number: VERDICT r04
    weak #4), bracketed by ac_clock_stamp on the same stream.  Returns queries/s, steps, ms/step and the observed shader clock."""
    from adaptive_classifier import _native as nv
    dev = torch.device("cuda", torch.cuda.current_device())
    st = torch.zeros((2, 16), dtype=torch.int64, device=dev)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); torch.cuda.synchronize()
    n = max(min_steps, int(seconds / max(1e-4, time.perf_counter() - t0)))
    nv.check(nv.lib().ac_clock_stamp(nv.ptr(st[0]), nv.stream_ptr(dev)), "ac_clock_stamp")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nv.check(nv.lib().ac_clock_stamp(nv.ptr(st[1]), nv.stream_ptr(dev)), "ac_clock_stamp")
    torch.cuda.synchronize()
    h = st.cpu().tolist()
    mhz, nx = shader_clock_between(h[0], h[1])
    return {"value": batch * n / dt, "unit": "queries/s", "steps": n, "seconds": dt, "ms_per_step": dt / n * 1e3,
            "shader_clock_mhz": mhz, "xcds_stamped": nx,
            "note": "same step as `value`, run back to back for %.1f s; shader clock = d(s_memtime) / d(s_memrealtime) over the region "
                    "(ac_clock_stamp), nominal 2400 MHz" % dt}


def _guarded(what, fn, *a, **k):
    """An EXTRA of the line (anything but `value` / `roofline`) must never cost the line itself: on an exception the key carries
    the error text instead of a number."""
    try:
        return fn(*a, **k)
    except Exception as e:                      # noqa: BLE001 -- reported, not swallowed
        import traceback
        print("bench.py: %s failed: %r" % (what, e), file=sys.stderr)
        traceback.print_exc()
        return {"error": "%s: %r" % (what, e)}


def _max_over_ranks(x, dev):
    t = torch.tensor([x], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _job_barrier():
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def sharded_cfg2(dev, rank, world, total_rows, steps, warmup, batch=4096, k=32, parity_queries=16):
    """BASELINE configs[2] on N GPUs through ShardedSearch (the production exchange): the 10M x 768 store row-sharded over
    the ranks, 4096 queries per step (4096 / N per rank, data parallel): all-gather(queries) -> local exact search of all
    4096 against the shard -> all-to-all of (exact fp64 distance, id) -> every rank merges its own block.  EXACTLY `steps`
    timed steps between barriers, max over ranks.  Rank 0 then repeats the same 4096-query batch against the WHOLE store on
    its own GPU (the one-GPU form of the same workload) -- the strong-scaling reference and, on its first `parity_queries`
    queries, the check that the sharded ids equal the unsharded ones."""
    from adaptive_classifier import index as ix
    from adaptive_classifier.sharded import ShardedSearch, shard_bounds
    lo, hi = shard_bounds(total_rows, world, rank)
    rows = ix.synth_unit_rows(hi - lo, DIM, 1, row_offset=lo, device=dev)
    qlo, qhi = shard_bounds(batch, world, rank)
    q_local = ix.synth_unit_rows(qhi - qlo, DIM, 2, row_offset=qlo, device=dev)
    sizes = [shard_bounds(batch, world, r)[1] - shard_bounds(batch, world, r)[0] for r in range(world)]   # known by construction
    # the fixed-batch path: blocks padded to the largest per-rank block, messages in buffers allocated once, nothing read back to
    # the host between the collectives, and the query all-gather of step i + 1 issued before step i is searched (search_blocks)
    ss = ShardedSearch(rows, hi - lo, DIM, lo, block_rows=max(sizes))
    for _ in range(max(1, warmup)):
        ss.search_block(q_local, k)
    _job_barrier()
    t0 = time.perf_counter()
    for Dg, Ig in ss.search_blocks((q_local for _ in range(steps)), k):
        pass
    _job_barrier()
    dt = _max_over_ranks(time.perf_counter() - t0, dev)
    # self-check of the job's shape, gathered THROUGH the job's own communicator: which ranks took part, what each one holds
    who = ss._all_gather(torch.tensor([rank, hi - lo, qhi - qlo, torch.cuda.current_device()], dtype=torch.int64, device=dev)).tolist()
    self_check = {"rccl_ranks_seen": len({w[0] for w in who}), "ranks": [w[0] for w in who], "shard_rows_per_rank": [w[1] for w in who],
                  "queries_per_rank": [w[2] for w in who], "device_index_per_rank": [w[3] for w in who],
                  "rows_total": sum(w[1] for w in who), "size_exchanges": ss.stats["size_exchanges"],
                  "message_buffers_allocated": ss.stats["buffer_allocations"], "query_gathers_prefetched": ss.stats["prefetched_gathers"]}
    # where the step goes on this rank: local search vs the rest (exchange + merge), HIP events on the current stream
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    q_all = ss.gather_queries(q_local, sizes)
    torch.cuda.synchronize()
    e[0].record()
    ss._search(rows, hi - lo, DIM, q_all, k, lo)
    e[1].record(); torch.cuda.synchronize()
    local_ms = _max_over_ranks(e[0].elapsed_time(e[1]), dev)
    ids_rank0 = Ig[:parity_queries].clone() if rank == 0 else None
    del rows, ss
    torch.cuda.empty_cache()
    one = None
    if rank == 0:
        P = ix.synth_unit_rows(total_rows, DIM, 1, device=dev)
        Q = ix.synth_unit_rows(batch, DIM, 2, device=dev)
        prep = ix.prepare_store(P, total_rows, DIM)
        ws = torch.empty(ix.knn_batch_workspace_bytes(total_rows, DIM, batch, k), dtype=torch.uint8, device=dev)
        out = (torch.empty((batch, k), dtype=torch.float32, device=dev), torch.empty((batch, k), dtype=torch.int64, device=dev))
        ix.knn_l2_topk(P, total_rows, DIM, Q, k, out=out, workspace=ws, prepared=prep)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(3):
            ix.knn_l2_topk(P, total_rows, DIM, Q, k, out=out, workspace=ws, prepared=prep)
        torch.cuda.synchronize(); d1 = (time.perf_counter() - t1) / 3
        one = {"ms_per_batch": d1 * 1e3, "queries_per_s": batch / d1,
               "sharded_ids_equal_unsharded": bool(torch.equal(out[1][:parity_queries], ids_rank0)),
               "checked_queries": int(parity_queries),
               "note": "the same 4096-query batch against the whole store on rank 0's GPU alone, measured in this job while the "
                       "other ranks wait (on a shared-device gloo run the other ranks' memory is still resident)"}
        del P, Q, prep, ws, out
        torch.cuda.empty_cache()
    dist.barrier()
    return {"ms_per_step": dt / steps * 1e3, "queries_per_s": batch * steps / dt, "rows_per_gpu": hi - lo, "self_check": self_check,
            "queries_per_gpu": qhi - qlo, "local_search_ms_max_over_ranks": local_ms,
            "exchange_and_merge_ms": max(0.0, dt / steps * 1e3 - local_ms), "one_gpu_same_workload": one,
            "speedup_vs_one_gpu": None if one is None else one["ms_per_batch"] / (dt / steps * 1e3)}


def shard_sweep_roofline(dev, rank, world, total_rows):
    """N > 1 roofline entry: the 10M-row store row-sharded, every rank sweeps its own shard with 16 resident queries (no
    collective in the sweep); achieved = total algorithmic bytes / the slowest rank's kernel time, peak = N x 8 TB/s."""
    from adaptive_classifier.sharded import shard_bounds
    lo, hi = shard_bounds(total_rows, world, rank)
    r = sweep_roofline(dev, hi - lo, full=False)
    slow = _max_over_ranks(r["avg_kernel_ms"], dev)
    total_bytes = total_rows * DIM * 4
    return {"bound": "hbm", "achieved": total_bytes / slow / 1e6, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
            "frac": total_bytes / slow / 1e6 / (HBM_PEAK_GBS * world), "traffic": None,
            "kernel": r["kernel"] + "; one shard per GPU", "rows": total_rows, "rows_per_gpu": hi - lo, "dim": DIM,
            "resident_queries": 16, "algorithmic_bytes_per_launch": (hi - lo) * DIM * 4, "avg_kernel_ms": slow,
            "rank0_avg_kernel_ms": r["avg_kernel_ms"],
            "aggregation": "sum of the shards' algorithmic bytes / max over ranks of the per-launch kernel time (HIP events on the kernel's stream)"}


def measure_cfg4(dev, args, steps=None, warmup=None, parity_queries=16):
    """BASELINE configs[4] end to end on ONE GPU: e5-large-v2 architecture (= BERT-large: 24 layers, 1024 hidden, 16 heads,
    4096 intermediate; random init, no weights offline), 2M x 1024 prototype store, 64 classes (row % 64), batch 1024,
    k = 32, S = 32.  Same step as the headline: encode -> kNN -> head -> blend -> Python result lists."""
    from adaptive_classifier import AdaptiveClassifier, AdaptiveHead
    from adaptive_classifier import index as ix
    from adaptive_classifier.encoder import HipBertEncoder
    from transformers import BertConfig, BertModel
    D, NP_, C, B, K_, S = 1024, 2_000_000, 64, 1024, 32, 32
    cfg = BertConfig(hidden_size=D, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    torch.manual_seed(0)
    hf = BertModel(cfg, add_pooling_layer=False).eval()
    enc = HipBertEncoder(hf, device=dev)
    clf = AdaptiveClassifier("e5-large-v2(random-init)", device=str(dev), encoder=enc, tokenizer=None)
    labels = [f"c{i}" for i in range(C)]
    clf.label_to_id = {l: i for i, l in enumerate(labels)}
    clf.id_to_label = {i: l for i, l in enumerate(labels)}
    clf.training_history = {l: 25 for l in labels}
    clf.adaptive_head = AdaptiveHead(D, C, [D, D // 2]).to(dev).eval()
    rows = ix.synth_unit_rows(NP_, D, 1, device=dev)
    clf.memory.load_rows(rows, torch.arange(NP_, dtype=torch.int32) % C, labels)
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(1000, VOCAB, (B, S), generator=g); ids[:, 0] = 101
    lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
    ids = (ids * mask).to(dev); mask = mask.to(dev); types = torch.zeros_like(ids)
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    step = lambda: clf.predict_tokens(ids, types, mask, k=K_)
    for _ in range(warmup):
        preds = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        preds = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert len(preds) == B
    sus = _guarded("cfg4 value_sustained", sustained_predict, step, B, seconds=2.5, min_steps=20)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record(); emb = clf.model.encode_cls(ids, types, mask, verify=False); ev[1].record()
    S_, I_, D_ = clf.memory.search_batch(emb, K_); ev[2].record(); torch.cuda.synchronize()
    parity = None
    if not args.no_parity:
        from oracle import c_oracle
        sel = np.arange(0, B, max(1, B // parity_queries))
        chunks = ((s, rows[s:min(NP_, s + 500_000), :D].cpu().numpy()) for s in range(0, NP_, 500_000))
        oD, oI = c_oracle.knn_l2_topk_chunked(chunks, emb[:, :D].cpu().numpy()[sel], K_)
        want = torch.nn.functional.normalize(hf(input_ids=ids[:4].cpu(), token_type_ids=types[:4].cpu(),
                                                attention_mask=mask[:4].cpu()).last_hidden_state[:, 0, :], dim=1)
        parity = {"checked_queries": int(len(sel)), "id_mismatches": int((I_.cpu().numpy()[sel] != oI).sum()),
                  "encoder_max_abs_diff_vs_transformers_fp32": float((emb[:4].cpu() - want.detach()).abs().max())}
    lens_h = mask.sum(1).double().cpu()
    enc_flops = enc.flops(B, S, tokens=float(lens_h.sum()), sum_len_sq=float((lens_h ** 2).sum())) if enc.last_tokens < B * S else enc.flops(B, S)
    f16 = None
    from adaptive_classifier import _native as nv
    if nv.lib().ac_gemm_get_arith() == 1:          # the opt-in fp16x2 arithmetic on the same batch (see main(): value_f16x2_opt_in)
        enc.enable_f16x2()
        clf._gemm_arith = nv.AC_GEMM_F16X2          # per-object option
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(2, steps // 2)):
            step()
        torch.cuda.synchronize()
        dt16 = (time.perf_counter() - t0) / max(2, steps // 2)
        emb16 = clf.model.encode_cls(ids, types, mask, arith="f16x2")
        clf._gemm_arith = None
        enc.disable_f16x2()
        f16 = {"value": B / dt16, "unit": "queries/s", "ms_per_step": dt16 * 1e3, "overflow_fallbacks": int(enc.f16x2_overflows),
               "max_abs_embedding_diff_vs_bf16x3": float((emb16 - emb).abs().max()), "note": "OPT-IN arithmetic, not `value`"}
    del rows, clf, enc
    torch.cuda.empty_cache()
    return {
        "metric": "predict() queries/sec + kNN GB/s vs HBM roofline, 768-d", "value": B * steps / dt, "unit": "queries/s",
        "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4] on ONE GPU (the config names 4): e5-large-v2 architecture (BERT-large, random "
                               "init), 1024-d, 2M prototypes, 64 classes, batch=1024, k=32, S=32, end-to-end predict()",
                   "batch_per_gpu": B, "seq_len": S, "prototypes": NP_, "dim": D, "k": K_, "classes": C, "parallelism": "dp1"},
        "stages_ms": {"encode_ms": ev[0].elapsed_time(ev[1]), "knn_ms": ev[1].elapsed_time(ev[2])},
        "roofline_encoder": {"bound": "mfma", "achieved": enc_flops / ev[0].elapsed_time(ev[1]) / 1e9,
                             "peak": BF16_MFMA_PEAK_TF / 6.0, "unit": "TFLOP/s",
                             "frac": enc_flops / ev[0].elapsed_time(ev[1]) / 1e9 / (BF16_MFMA_PEAK_TF / 6.0),
                             "note": "executed FLOPs (padding tokens left out, last layer on the CLS rows) against the fp32-equivalent "
                                     "bf16x3 ceiling 2500 / 6"},
        "parity": parity, "value_f16x2_opt_in": f16, "value_sustained": sus}


def measure_add_examples(dev, args, n=None, modes=("as_wired", "intended"), with_cpu=None, with_cpu_reference=False):
    """BASELINE configs[3]: the add_examples() continuous-learning loop on one GPU.  50 000 pre-computed unit-norm 768-d
    embeddings (class centroid + 0.5 noise, 4 classes) fed in chunks of 32 through add_embeddings (= add_examples after
    the encoder call), max_examples_per_class = 1000; every call updates the memory (device prune), retrains the head on
    everything stored (<= 10 epochs, early stopping) and rebuilds the index -- what the reference does per call.  Then a
    5th class in both EWC modes.  `--examples N` shrinks the run."""
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier import index as ix
    from adaptive_classifier.encoder import _Cfg

    class _NoEncoder:                      # the loop is fed embeddings; the encoder is bypassed
        config = _Cfg(DIM, "precomputed-embeddings")

    n, C = (args.examples if n is None else n), 4
    with_cpu = (not getattr(args, "no_cpu_baseline", False)) if with_cpu is None else with_cpu
    cent = ix.synth_unit_rows(C + 1, DIM, 3, device=dev)[:, :DIM]
    noise = ix.synth_unit_rows(n + 64, DIM, 4, device=dev)[:, :DIM]
    cls = torch.arange(n + 64, device=dev) % C
    cls[n:] = C
    E = torch.nn.functional.normalize(cent[cls] + 0.5 * noise, dim=1).cpu()
    out = {}
    # untimed warm-up (the contract's W): a 256-example loop on a throwaway classifier loads every kernel once
    wclf = AdaptiveClassifier("precomputed", device=str(dev), encoder=_NoEncoder(), tokenizer=None)
    for s in range(0, 256, 32):
        wclf.add_embeddings([f"w{i}" for i in range(s, s + 32)], [E[i] for i in range(s, s + 32)], [f"c{i % C}" for i in range(s, s + 32)])
    wclf.add_embeddings([f"wn{i}" for i in range(8)], [E[n + i] for i in range(8)], ["znew"] * 8)
    del wclf
    for mode in modes:
        clf = AdaptiveClassifier("precomputed", device=str(dev), config={"ewc_mode": mode}, encoder=_NoEncoder(), tokenizer=None)
        T = {"memory": 0.0, "train": 0.0, "rebuild": 0.0}

        def timed(obj, name, key):
            f = getattr(obj, name)

            def g(*a, **k):
                t = time.perf_counter(); r = f(*a, **k); T[key] += time.perf_counter() - t; return r
            setattr(obj, name, g)
        timed(clf.memory, "add_examples_batch", "memory"); timed(clf, "_train_adaptive_head", "train")
        timed(clf.memory, "_rebuild_index", "rebuild")
        steps = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(0, n, 32):
            e = min(n, s + 32)
            clf.add_embeddings([f"t{i}" for i in range(s, e)], [E[i] for i in range(s, e)], [f"c{i % C}" for i in range(s, e)])
            steps += clf.last_train_info["steps"]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        np.random.seed(0)
        t1 = time.perf_counter()
        clf.add_embeddings([f"n{i}" for i in range(32)], [E[n + i] for i in range(32)], ["znew"] * 32)
        torch.cuda.synchronize(); dt_new = time.perf_counter() - t1
        test = torch.nn.functional.normalize(cent[torch.arange(200, device=dev) % (C + 1)] + 0.5 * ix.synth_unit_rows(200, DIM, 9, device=dev)[:, :DIM], dim=1)
        preds = clf.predict_embeddings(test, k=1)
        names = [f"c{i}" for i in range(C)] + ["znew"]
        acc = float(np.mean([p[0][0] == names[i % (C + 1)] for i, p in enumerate(preds)]))
        out[mode] = {"examples": n, "seconds": dt, "examples_per_s": n / dt, "train_steps": steps, "steps_per_s": steps / dt,
                     "host_seconds_by_phase": T, "stored": clf.get_memory_stats()["total_examples"],
                     "new_class_seconds": dt_new, "new_class_info": clf.last_train_info, "accuracy_5way": acc}
        if mode == "as_wired":
            headline = out[mode]
    # cpu_baseline: the reference's training step (torch CPU: forward with dropout, CE, backward, clip, AdamW -- oracle/head_oracle.py)
    # on the host cores, a bounded sample of the same steps; the loop's examples/s would be steps/s x (examples per step of
    # the GPU run) if nothing else cost anything
    cpu = None
    if with_cpu:
        from oracle import c_oracle, head_oracle
        cores = c_oracle.usable_cores()
        torch.set_num_threads(cores)
        ref = head_oracle.make_head(DIM, C).train()
        opt = torch.optim.AdamW(ref.parameters(), lr=0.001, weight_decay=0.01)
        nbc = max(1, min(64, E.shape[0] // 32))
        Xc = E[:32 * nbc].float().cpu(); yc = (torch.arange(32 * nbc) % C)
        for i in range(5):
            head_oracle.train_step(ref, opt, Xc[(i % nbc) * 32:(i % nbc + 1) * 32], yc[(i % nbc) * 32:(i % nbc + 1) * 32])
        t0 = time.perf_counter(); nst = 0
        while time.perf_counter() - t0 < 10.0:
            j = nst % nbc
            head_oracle.train_step(ref, opt, Xc[j * 32:(j + 1) * 32], yc[j * 32:(j + 1) * 32]); nst += 1
        dtc = time.perf_counter() - t0
        cpu = {"value": nst / dtc * (n / max(1, headline["train_steps"])), "unit": "examples/s", "cores": int(torch.get_num_threads()), "kind": "port",
               "steps_per_s": nst / dtc, "sample": f"{nst} reference training steps (batch 32, torch CPU) in {dtc:.1f} s; value = steps/s x "
               f"examples per training step of the GPU run ({n}/{headline['train_steps']}), memory bookkeeping not charged"}
    # ... and the unmodified reference's own add_examples loop (kind "reference"), when oracle/_ref is staged: that is `cpu_baseline`,
    # the port above stays beside it
    cpu_port = None
    if with_cpu or with_cpu_reference:
        ref = _guarded("add_examples cpu_baseline (reference)", cpu_baseline_add_examples_reference, E, C)
        if ref is not None and "error" not in ref:
            cpu, cpu_port = ref, cpu
    return {
        "metric": "add_examples() examples/sec (continuous-learning loop)", "value": headline["examples_per_s"], "unit": "examples/s",
        "n_gpus": 1, "steps": headline["train_steps"], "warmup": 0, "ms_per_step": headline["seconds"] / max(1, headline["train_steps"]) * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: add_examples() loop, %d pre-computed 768-d examples in chunks of 32, 4 classes, "
                               "cap 1000/class, head retrained per call (<= 10 epochs), then a 5th class (EWC path)" % n,
                   "dim": DIM, "chunk": 32, "max_examples_per_class": 1000,
                   "training": "ac_head_train_epoch: ONE persistent launch per epoch (head_epoch.hip: weights + AdamW moments "
                               "stationary in LDS, 3 grid barriers per step) + 1 memset; one host sync per epoch (early stopping)",
                   "launches_per_epoch": 2},
        "roofline": {"bound": "latency", "achieved": headline["steps_per_s"], "unit": "training steps/s", "peak": None, "frac": None,
                     "note": "the step is a chain of dependent phases (3 grid barriers + 3 dependent cross-CU reads per step, ~28 us); "
                             "its algorithmic traffic (36 B/param = 32 MB/step) would take 4 us at the HBM peak and never leaves LDS here"},
        "cpu_baseline": cpu, "cpu_baseline_port": cpu_port,
        "modes": out}


def measure_add_examples_text(dev, args, n=None, enc=None):
    """BASELINE configs[3] WITH THE ENCODER IN THE LOOP (SURVEY 8d cfg3 "and once with the encoder"; classifier.py:132-200 from
    text): n synthetic texts of 6..28 words (<= 32 tokens; a class's texts draw 80 % of their words from that class's word pool)
    fed in chunks of 32 through `add_examples(texts, labels)`: device WordPiece -> bert-base encoder (random init) -> D2H of
    the chunk's embeddings (the reference's list-of-CPU-tensors contract) -> memory update (device prune at the 1000 cap) -> head
    retrained on everything stored (<= 10 epochs) -> index rebuild; then a 5th class (`_train_new_classes`, as-wired EWC)."""
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier.encoder import HipBertEncoder
    from transformers import BertConfig, BertModel, BertTokenizer
    n, C = (args.examples if n is None else n), 4
    vocab, words = synthetic_wordpiece()
    tok = BertTokenizer(vocab=vocab, do_lower_case=True)
    if enc is None:
        torch.manual_seed(0)
        enc = HipBertEncoder(BertModel(BertConfig(), add_pooling_layer=False).eval(), device=dev)
    rng = np.random.default_rng(21)
    pools = [words[c::C + 1] for c in range(C + 1)]

    def text(c):
        k = int(rng.integers(6, 29))
        own = rng.random(k) < 0.8
        return " ".join(str(rng.choice(pools[c])) if o else str(rng.choice(words)) for o in own)
    labels = [f"c{i % C}" for i in range(n)]
    texts = [text(i % C) for i in range(n)]
    new_texts = [text(C) for _ in range(64)]
    test_texts = [text(i % (C + 1)) for i in range(200)]
    cfg = {"max_length": SEQ}
    wclf = AdaptiveClassifier("bert-base-uncased(random-init)", device=str(dev), config=cfg, encoder=enc, tokenizer=tok)
    for s0 in range(0, 128, 32):                        # untimed warm-up on a throwaway classifier: every kernel loaded once
        wclf.add_examples(texts[s0:s0 + 32], labels[s0:s0 + 32])
    wclf.add_examples(new_texts[:8], ["znew"] * 8)
    del wclf
    clf = AdaptiveClassifier("bert-base-uncased(random-init)", device=str(dev), config=cfg, encoder=enc, tokenizer=tok)
    T = {"encode": 0.0, "memory": 0.0, "train": 0.0, "rebuild": 0.0}

    def timed(obj, name, key):
        f = getattr(obj, name)

        def g(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); T[key] += time.perf_counter() - t; return r
        setattr(obj, name, g)
    timed(clf, "_get_embeddings", "encode"); timed(clf.memory, "add_examples_batch", "memory")
    timed(clf, "_train_adaptive_head", "train"); timed(clf.memory, "_rebuild_index", "rebuild")
    steps = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s0 in range(0, n, 32):
        clf.add_examples(texts[s0:s0 + 32], labels[s0:s0 + 32])
        steps += clf.last_train_info.get("steps", 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    clf.add_examples(new_texts, ["znew"] * 64)
    torch.cuda.synchronize(); dt_new = time.perf_counter() - t1
    preds = clf.predict_batch(test_texts, k=1)
    names = [f"c{i}" for i in range(C)] + ["znew"]
    acc = float(np.mean([p[0][0] == names[i % (C + 1)] for i, p in enumerate(preds)]))
    return {"metric": "add_examples() examples/sec (continuous-learning loop, encoder in the loop)", "value": n / dt, "unit": "examples/s",
            "examples": n, "seconds": dt, "train_steps": steps, "steps_per_s": steps / dt, "host_seconds_by_phase": T,
            "tokenizer": type(clf.tokenizer).__name__, "stored": clf.get_memory_stats()["total_examples"],
            "new_class_seconds": dt_new, "accuracy_5way": acc,
            "config": {"workload": "BASELINE configs[3] with the encoder: add_examples(texts, labels) loop, %d synthetic texts (6..28 words, "
                                   "<= 32 tokens) in chunks of 32, bert-base-uncased arch (random init), 4 classes, cap 1000/class, head "
                                   "retrained per call (<= 10 epochs), then a 5th class" % n,
                       "dim": DIM, "chunk": 32, "max_examples_per_class": 1000, "seq_len": SEQ}}


def measure_latency(dev, args, S=16, reps=200, made=None, with_cpu=None, cpu_seconds=10.0):
    """Single predict() latency: one text of S tokens (ids given: tokenisation excluded like everywhere in this file) ->
    encoder (one persistent launch, bert_small.hip) -> kNN over the configs[1] store (100k x 768) -> head -> blend -> the
    reference's [(label, score)] list on the host.  cpu_baseline: the same single query through the CPU port."""
    clf, hf = made if made is not None else make_classifier(dev, 0, 1)
    with_cpu = (not args.no_cpu_baseline) if with_cpu is None else with_cpu
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(1000, VOCAB, (1, S), generator=g); ids[:, 0] = 101
    ids_d = ids.to(dev)
    ncls = len(clf.id_to_label)

    def one():                                       # the chain of AdaptiveClassifier._predict_regular (verify=False: no mid-chain sync)
        emb = clf.model.encode_cls(ids_d, verify=False)
        S_, I_, P_ = clf._device_stage(emb, ncls)
        return clf._finish(S_, I_, P_, 3, True, b=1)
    for _ in range(10):
        res = one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        res = one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        clf.model.encode_cls(ids_d, verify=False)
    e1.record(); torch.cuda.synchronize()
    enc_ms = e0.elapsed_time(e1) / reps
    cpu = None
    if with_cpu:
        from oracle import c_oracle, head_oracle
        cores = c_oracle.usable_cores()
        torch.set_num_threads(cores); c_oracle.set_threads(cores)
        from adaptive_classifier import index as ix
        P = ix.synth_unit_rows(NPROTO, DIM, 1, device=dev)[:, :DIM].cpu().numpy()          # the same store, on the host
        head = head_oracle.make_head(DIM, NCLASS).eval()
        mask = torch.ones_like(ids)

        def cpu_one():
            with torch.no_grad():
                e = torch.nn.functional.normalize(hf(input_ids=ids, attention_mask=mask).last_hidden_state[:, 0, :], dim=1)
                D_, I_ = c_oracle.knn_l2_topk_f32(P, e.numpy(), KNN_K)
                torch.softmax(torch.from_numpy(np.exp(-D_)), dim=1); torch.softmax(head(e), dim=1)
        for _ in range(3):
            cpu_one()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < cpu_seconds:
            cpu_one(); n += 1
        cpu = {"value": (time.perf_counter() - t0) / n * 1e3, "unit": "ms", "cores": int(torch.get_num_threads()), "kind": "port",
               "sample": f"{n} single queries: transformers BertModel fp32 (torch CPU) + C fp32 brute-force kNN over {NPROTO}x{DIM} + torch head"}
    return {
        "metric": "single predict() latency (one text, %d tokens)" % S, "value": dt * 1e3, "unit": "ms", "n_gpus": 1, "steps": reps,
        "warmup": 10, "ms_per_step": dt * 1e3, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "one query of %d tokens, bert-base-uncased arch (random init), %d prototypes x %d-d, %d classes, "
                               "predict() = encoder + kNN + head + blend + host list" % (S, NPROTO, DIM, NCLASS),
                   "seq_len": S, "prototypes": NPROTO, "dim": DIM, "classes": NCLASS,
                   "encoder": "one persistent launch (bert_small.hip), strict fp32 MFMA"},
        "stages_ms": {"encode_ms": enc_ms, "rest_ms": dt * 1e3 - enc_ms},
        "roofline": {"bound": "latency", "achieved": dt * 1e3, "unit": "ms", "peak": None, "frac": None,
                     "note": "60 dependent phase boundaries of ~3.5 us inside the encoder launch + ~10 launches after it; the 340 MB of "
                             "fp32 encoder weights stream once (43 us at the HBM peak), the 307 MB store once (38 us)"},
        "cpu_baseline": cpu, "reference_published": {"pytorch_cpu_ms": 8.3, "onnx_cpu_ms": 2.1, "source": "reference README.md:256-261, hardware unspecified"},
        "result": [[l, float(s_)] for l, s_ in res[0]]}


def synthetic_wordpiece(vocab_size=VOCAB, seed=7):
    """A synthetic BERT-style WordPiece vocabulary of `vocab_size` entries (no pretrained vocabulary is available offline):
    specials, every printable ASCII character and its ## form, then random lower-case words and ## suffixes.  Returns
    (vocab dict, list of whole words it contains)."""
    rng = np.random.default_rng(seed)
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars = [chr(c) for c in range(33, 127) if not ("A" <= chr(c) <= "Z")]
    toks += chars + ["##" + c for c in chars if c.isalnum()]
    seen, words = set(toks), []
    letters = list("abcdefghijklmnopqrstuvwxyz")
    while len(toks) < vocab_size:
        w = "".join(rng.choice(letters, int(rng.integers(2, 10))))
        for piece in ((w, "##" + w[-3:]) if len(toks) % 3 else (w,)):
            if piece not in seen and len(toks) < vocab_size:
                seen.add(piece); toks.append(piece)
                if not piece.startswith("##"):
                    words.append(piece)
    return {t: i for i, t in enumerate(toks)}, words


def measure_predict_from_text(dev, clf, reps=5):
    """predict_batch(raw strings) queries/s, tokenisation INCLUDED ("with and without host tokenisation"):
    256 synthetic texts of 6..28 words (<= 32 tokens after truncation) through the classifier's own entry point, once with
    the on-device WordPiece (ac_wordpiece_encode) and once with the wrapped transformers tokenizer on the host."""
    from adaptive_classifier import AdaptiveClassifier
    from transformers import BertTokenizer
    vocab, words = synthetic_wordpiece()
    tok = BertTokenizer(vocab=vocab, do_lower_case=True)
    rng = np.random.default_rng(11)
    punct = list(",.;:!?")
    texts = []
    for _ in range(BATCH):
        ws = [str(rng.choice(words)) for _ in range(int(rng.integers(6, 29)))]
        if rng.random() < 0.5:
            ws[int(rng.integers(0, len(ws)))] += str(rng.choice(punct))
        texts.append(" ".join(ws).capitalize())
    out = {"texts": BATCH, "max_length": SEQ, "vocabulary": "synthetic WordPiece, %d entries (no pretrained vocabulary offline)" % len(vocab)}
    for name, extra in (("device_tokenizer", {}), ("host_tokenizer", {"device_tokenizer": False})):
        c = AdaptiveClassifier("bert-base-uncased(random-init)", device=str(dev), config={"max_length": SEQ, **extra},
                               encoder=clf.model, tokenizer=tok)
        c.label_to_id, c.id_to_label, c.training_history = clf.label_to_id, clf.id_to_label, clf.training_history
        c.adaptive_head, c.memory = clf.adaptive_head, clf.memory
        for _ in range(2):
            preds = c.predict_batch(texts, k=KNN_K, batch_size=BATCH)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            preds = c.predict_batch(texts, k=KNN_K, batch_size=BATCH)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            enc_in = c.tokenizer(texts, max_length=SEQ, truncation=True, padding=True, return_tensors="pt")
        torch.cuda.synchronize(); dtt = (time.perf_counter() - t0) / reps
        assert len(preds) == BATCH
        out[name] = {"queries_per_s": BATCH / dt, "ms_per_batch": dt * 1e3, "tokenizer_call_ms": dtt * 1e3,
                     "tokens_padded": list(enc_in["input_ids"].shape), "tokenizer": type(c.tokenizer).__name__}
    return out


def timed_predict(clf, ids, types, mask, steps, warmup):
    """`warmup` untimed steps, then EXACTLY `steps` predict() steps between barrier + synchronize on both sides;
    returns the elapsed seconds, max over ranks."""
    for _ in range(warmup):
        preds = predict_step(clf, ids, types, mask)
    _job_barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        preds = predict_step(clf, ids, types, mask)
    _job_barrier()
    dt = time.perf_counter() - t0
    if dist.is_initialized() and dist.get_world_size() > 1:
        dt = _max_over_ranks(dt, ids.device)
    assert len(preds) == ids.shape[0] and all(len(p) >= 1 for p in preds)
    return dt


def encoder_roofline(clf, stages, enc_flops, enc_peak, arith):
    ref_flops = clf.model.flops(BATCH, SEQ, executed=False)
    return {"bound": "mfma", "achieved": enc_flops / stages["encode_ms"] / 1e9,
            "peak": enc_peak, "unit": "TFLOP/s",
            "frac": enc_flops / stages["encode_ms"] / 1e9 / enc_peak,
            "flops_per_step": enc_flops,
            # the same FLOPs over the encoder's time in a back-to-back loop (its kernels' own time: no per-step launch ramp after the
            # step's result copy); `frac` stays on the stage time of the timed step, as in earlier rounds
            "frac_back_to_back": (enc_flops / stages["encode_ms_back_to_back"] / 1e9 / enc_peak) if "encode_ms_back_to_back" in stages else None,
            "peak_note": ("fp32-equivalent: bf16 MFMA dense peak 2500 / 6 products" if arith == 1
                          else "fp16 MFMA dense peak 2500 / 3 products (opt-in fp16x2 arithmetic)" if arith == 2
                          else "fp32-input MFMA dense peak"),
            "tokens_per_step": int(getattr(clf.model, "last_tokens", BATCH * SEQ)),
            "tokens_per_step_padded": BATCH * SEQ,
            "note": "executed FLOPs: the padding tokens of the ragged batch (lengths ~U[8,32], SURVEY 8d) are left "
                    "out of the forward (ac_bert_encode_cls_packed: identical CLS vectors), and the last layer runs "
                    "its post-attention part on the CLS rows only; the padded BertModel.forward the reference runs "
                    "would be %.4g" % ref_flops,
            "reference_flops_rate": {"TFLOPs": ref_flops / stages["encode_ms"] / 1e9,
                                     "of_peak": ref_flops / stages["encode_ms"] / 1e9 / enc_peak,
                                     "note": "NOT a roofline fraction: the FLOPs the reference's padded forward spends on this "
                                             "batch divided by the time this path takes for the same outputs"}}


def cfg4_multi(dev, rank, world, steps, warmup):
    """BASELINE configs[4] on N GPUs (it is quoted on 4): e5-large-v2 architecture (BERT-large, random init) replicated, 1024 texts
    per step data parallel (1024 / N per rank, ragged <= 32 tokens), the 2M x 1024 store row-sharded, 64 classes, k = 32: encoder ->
    all-gather(queries) -> local exact search -> all-to-all -> merge -> head -> blend -> result lists on every rank."""
    from adaptive_classifier import AdaptiveClassifier, AdaptiveHead
    from adaptive_classifier import index as ix
    from adaptive_classifier.encoder import HipBertEncoder
    from adaptive_classifier.sharded import ShardedSearch, shard_bounds
    from transformers import BertConfig, BertModel
    D, NP_, C, B, K_, S = 1024, 2_000_000, 64, 1024, 32, 32
    cfg = BertConfig(hidden_size=D, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    torch.manual_seed(0)
    enc = HipBertEncoder(BertModel(cfg, add_pooling_layer=False).eval(), device=dev)
    clf = AdaptiveClassifier("e5-large-v2(random-init)", device=str(dev), encoder=enc, tokenizer=None)
    labels = [f"c{i}" for i in range(C)]
    clf.label_to_id = {l: i for i, l in enumerate(labels)}
    clf.id_to_label = {i: l for i, l in enumerate(labels)}
    clf.training_history = {l: 25 for l in labels}
    clf.adaptive_head = AdaptiveHead(D, C, [D, D // 2]).to(dev).eval()
    lo, hi = shard_bounds(NP_, world, rank)
    rows = ix.synth_unit_rows(hi - lo, D, 1, row_offset=lo, device=dev)
    clf.memory.load_rows(rows, torch.arange(NP_, dtype=torch.int32) % C, labels,
                         sharded=ShardedSearch(rows, hi - lo, D, lo, block_rows=B // world))
    b = B // world
    g = torch.Generator().manual_seed(1234 + rank)
    ids = torch.randint(1000, VOCAB, (b, S), generator=g); ids[:, 0] = 101
    lens = torch.randint(8, S + 1, (b,), generator=g); lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
    ids = (ids * mask).to(dev); mask = mask.to(dev); types = torch.zeros_like(ids)
    for _ in range(warmup):
        preds = clf.predict_tokens(ids, types, mask, k=K_)
    _job_barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        preds = clf.predict_tokens(ids, types, mask, k=K_)
    _job_barrier()
    dt = _max_over_ranks(time.perf_counter() - t0, dev)
    assert len(preds) == b
    del clf, enc, rows
    torch.cuda.empty_cache()
    return {"workload": "BASELINE configs[4]: e5-large-v2 architecture (random init), %d x %d store row-sharded over %d GPUs, %d classes, "
                        "k=%d, %d texts per step data parallel (%d per rank, <= %d tokens, ragged)" % (NP_, D, world, C, K_, b * world, b, S),
            "value": b * world * steps / dt, "unit": "queries/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "rows_per_gpu": hi - lo}


def main_multi(args, dev, rank, world):
    """The N > 1 line.  `value` = BASELINE configs[2] -- the 10M x 768 store row-sharded over the N ranks, k = 32, 4096
    queries per step, through the production exchange (strong scaling: the job is the same at every N; the one-GPU form of
    the same job is measured by rank 0 inside this run, `config.one_gpu_same_workload`).  `roofline` = every rank's shard
    sweep against N x 8 TB/s.  `configs1_weak` = the N = 1 headline workload (configs[1], predict() end to end) run data
    parallel with the 100k-row store row-sharded, 256 texts per rank per step."""
    cfg2 = sharded_cfg2(dev, rank, world, args.sweep_rows, args.steps, args.warmup)
    roof = None if args.no_sweep else shard_sweep_roofline(dev, rank, world, args.sweep_rows)
    clf, hf = make_classifier(dev, rank, world)
    ids, types, mask = synthetic_tokens(dev, rank)
    dt = timed_predict(clf, ids, types, mask, args.steps, args.warmup)
    stages = time_stages(clf, ids, types, mask)
    backend = dist.get_backend()
    line = {
        "metric": "predict() queries/sec + kNN GB/s vs HBM roofline, 768-d",
        "value": cfg2["queries_per_s"], "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": cfg2["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: %d x %d synthetic prototypes row-sharded over %d GPUs, k=32, batch 4096 "
                               "(%d queries per rank): all-gather(queries) -> local exact search (fp16-plane proposals + fp64 "
                               "re-rank) -> all-to-all of (exact fp64 distance, id) -> every rank merges its own block"
                               % (args.sweep_rows, DIM, world, cfg2["queries_per_gpu"]),
                   "rows": args.sweep_rows, "dim": DIM, "k": 32, "batch": 4096, "rows_per_gpu": cfg2["rows_per_gpu"],
                   "parallelism": f"rowshard{world}+dp{world}", "backend": backend,
                   "collectives": "all_gather_into_tensor(queries, async: issued one step ahead) + all_to_all_single(candidates) per step; "
                                  "fixed shapes, pre-allocated messages, no host read-back between them",
                   "local_search_ms_max_over_ranks": cfg2["local_search_ms_max_over_ranks"],
                   "exchange_and_merge_ms": cfg2["exchange_and_merge_ms"],
                   "one_gpu_same_workload": cfg2["one_gpu_same_workload"],
                   "speedup_vs_one_gpu": cfg2["speedup_vs_one_gpu"],
                   "value_n1_equivalent": None if cfg2["one_gpu_same_workload"] is None else cfg2["one_gpu_same_workload"]["queries_per_s"],
                   "self_check": cfg2["self_check"],
                   "note": ("N = 1 of this file reports configs[1] (predict() end to end); the N > 1 line leads with the sharded "
                            "kNN of configs[2] because at configs[1] the kNN is 4 % of a step -- its data-parallel weak scaling is "
                            "carried as configs1_weak")},
        "roofline": roof,
        "configs1_weak": {"workload": "BASELINE configs[1] data parallel: 256 texts per rank per step, 100k x 768 store row-sharded "
                                      "over the ranks, k=16 (all-gather queries, local search, all-to-all, merge), head, blend",
                          "value": BATCH * world * args.steps / dt, "unit": "queries/s", "scaling": "weak",
                          "ms_per_step": dt / args.steps * 1e3, "stages_ms": stages},
    }
    if not args.no_extras:
        del clf
        torch.cuda.empty_cache()
        line["configs4_data_parallel_sharded"] = cfg4_multi(dev, rank, world, steps=max(2, min(args.steps, 5)), warmup=1)
    if backend != "nccl":
        line["not_a_measurement"] = ("backend %s: ranks may share one GPU and device tensors are staged through the host -- this run "
                                     "exercises the N-rank code path only" % backend)
    return line


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks ourselves (what `torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1` does, minus the elastic agent): rank r = LOCAL_RANK r -> cuda:r, rendezvous
    on a free 127.0.0.1 port, rank 0's stdout (the ONE JSON line) is ours, the other ranks' output goes to stderr.
    Returns the exit code (0 only if every rank exited 0); a failing rank takes the others down (exact PIDs)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    backend = os.environ.get("AC_BENCH_BACKEND", "nccl")
    if ndev < n and backend == "nccl":
        print(f"bench.py: --gpus {n} but {ndev} GPU(s) visible; RCCL needs one device per rank "
              f"(AC_BENCH_BACKEND=gloo runs the N-rank code path on shared devices: a code-path exercise, not a measurement)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=r == 0))

    def pump():          # our stdout carries rank 0's JSON line and nothing else (gloo, for one, prints its connection banner on stdout)
        for ln in procs[0].stdout:
            (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln)
        sys.stdout.flush()
    import threading
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    th.join(timeout=10)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sweep-rows", type=int, default=10_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the on-box oracle checks (parity fields become null)")
    ap.add_argument("--no-extras", action="store_true",
                    help="default run only: skip the other BASELINE configs carried as extra keys (latency_ms_b1, cfg4, add_examples, predict_from_text)")
    ap.add_argument("--config", default="predict", choices=["predict", "cfg4", "add_examples", "latency"],
                    help="predict = BASELINE configs[1] (the headline, default); cfg4 = configs[4] end to end on one GPU; "
                         "add_examples = configs[3] continuous-learning loop; latency = one predict() of one short text "
                         "(the only number the reference publishes: README.md:256-261)")
    ap.add_argument("--examples", type=int, default=50_000, help="--config add_examples: number of examples fed")
    ap.add_argument("--with-encoder", action="store_true",
                    help="--config add_examples: feed TEXTS through add_examples() (tokenizer + encoder in the loop) instead of "
                         "pre-computed embeddings")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks of itself (one per GPU)
        raise SystemExit(launch_ranks(args.gpus))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or unset WORLD_SIZE and let bench.py spawn them)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  AC_BENCH_BACKEND=gloo exists only to exercise the N>1 code path with
        # several ranks on ONE GPU (RCCL refuses duplicate devices); it is never used for reported numbers.
        backend = os.environ.get("AC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.config != "predict":
        if world > 1:
            raise SystemExit("--config %s is a single-GPU measurement" % args.config)
        out = (measure_latency(dev, args) if args.config == "latency" else
               measure_cfg4(dev, args) if args.config == "cfg4" else
               measure_add_examples_text(dev, args) if args.with_encoder else measure_add_examples(dev, args))
        print(json.dumps(out), flush=True)
        return

    if world > 1:
        line = main_multi(args, dev, rank, world)
        if rank == 0:
            print(json.dumps(line), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    clf, hf = make_classifier(dev, rank, world)
    ids, types, mask = synthetic_tokens(dev, rank)
    # Serving hygiene, not a kernel trick: a step returns 256 lists of 16 (label, score) tuples, i.e. ~4 k tracked objects, so
    # CPython starts a FULL (generation-2) collection every ~16 steps -- and with transformers + two BERT models imported that
    # walk over ~10^6 long-lived objects takes 30 - 40 ms, longer than six steps (seen as sporadic 37 k / 36 k queries/s lines
    # among 48 k ones in round 5).  gc.freeze() moves everything allocated so far into the permanent generation (what serving
    # stacks do after model load); the per-step garbage is still collected.
    import gc
    dt_unfrozen = timed_predict(clf, ids, types, mask, args.steps, args.warmup)     # the same loop BEFORE the freeze, reported beside `value`
    gc.collect()
    gc.freeze()                                   # (stays in force for the rest of the process: the cpu_baseline legs run frozen too)
    dt = timed_predict(clf, ids, types, mask, args.steps, args.warmup)
    stages = time_stages(clf, ids, types, mask)

    # the same loop with the fp32-input MFMA arithmetic for the large GEMMs (reported next to `value`)
    from adaptive_classifier import _native as nv
    arith = nv.lib().ac_gemm_get_arith()
    clf._gemm_arith = nv.AC_GEMM_F32              # per-object option (config["gemm_arith"]): no process-wide switch is touched
    dt32 = timed_predict(clf, ids, types, mask, args.steps, 2)
    clf._gemm_arith = None
    # ... and with the OPT-IN fp16x2 arithmetic (include/acamd.h AC_GEMM_F16X2: operands rounded to two fp16 terms = 22 bits,
    # three fp16 MFMA products; NOT the arithmetic of `value`): the same loop, its embeddings measured against the bf16x3 ones
    # of the same batch and against transformers fp32, the predicted labels against the headline's
    f16 = None
    if arith == 1 and hasattr(clf.model, "enable_f16x2"):
        with torch.no_grad():
            emb3 = clf.model.encode_cls(ids, types, mask).clone()
            res3 = predict_step(clf, ids, types, mask)
            clf.model.enable_f16x2()
            clf._gemm_arith = nv.AC_GEMM_F16X2
            dt16 = timed_predict(clf, ids, types, mask, args.steps, 2)
            active = bool(clf.model.f16x2_active("f16x2"))
            emb16 = clf.model.encode_cls(ids, types, mask, arith="f16x2").clone()
            res16 = predict_step(clf, ids, types, mask)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                clf.model.encode_cls(ids, types, mask, verify=False, arith="f16x2")
            e1.record(); torch.cuda.synchronize()
            want = torch.nn.functional.normalize(
                hf(input_ids=ids[:8].cpu(), token_type_ids=types[:8].cpu(), attention_mask=mask[:8].cpu()).last_hidden_state[:, 0, :], dim=1)
            clf._gemm_arith = None
            clf.model.disable_f16x2()
        f16 = {"value": BATCH * args.steps / dt16, "unit": "queries/s", "ms_per_step": dt16 / args.steps * 1e3,
               "encode_ms": e0.elapsed_time(e1) / 5, "ran_fp16x2": active, "overflow_fallbacks": int(clf.model.f16x2_overflows),
               "max_abs_embedding_diff_vs_bf16x3": float((emb16 - emb3).abs().max()),
               "encoder_max_abs_diff_vs_transformers_fp32": float((emb16[:8].cpu() - want).abs().max()),
               "same_labels_as_value": [[l for l, _ in p] for p in res16] == [[l for l, _ in p] for p in res3],
               "queries_with_the_same_neighbour_ids": float((clf.memory.search_batch(emb16, KNN_K)[1] ==
                                                              clf.memory.search_batch(emb3, KNN_K)[1]).all(1).float().mean()),
               "score_note": "the synthetic store is 100k random unit vectors: the k-th neighbours of a query are near-ties, so "
                             "embeddings 2e-7 apart can swap one and move a blended score by a few percent -- the same happens "
                             "between any two fp32 evaluation orders",
               "max_score_diff_vs_value": float(max(abs(x[1] - y[1]) for p, q in zip(res16, res3) for x, y in zip(p, q))),
               "note": "OPT-IN, not `value`: config gemm_arith='f16x2' / AC_GEMM_ARITH=f16x2.  Operands of the token-row GEMMs rounded "
                       "to two fp16 terms of x 2^s (22 bits), 3 fp16 MFMA products instead of 6 bf16 ones; error per product <= 3 * "
                       "2^-22 |a||b| (tests/test_gemm_f16x2_gpu.py measures it next to fp32-MFMA and bf16x3 against fp64); an operand "
                       "out of fp16 range gives NaN and the call is repeated in bf16x3"}
    # ... and with every text at the full 32 tokens (no padding to leave out): same timed loop, same arithmetic as `value`
    ids_f, types_f, mask_f = synthetic_tokens(dev, rank, full_length=True)
    dt_full = timed_predict(clf, ids_f, types_f, mask_f, args.steps, 2)
    tokens_full = int(getattr(clf.model, "last_tokens", BATCH * SEQ))
    predict_step(clf, ids, types, mask)          # (restores last_tokens of the headline batch for the accounting below)

    enc_peak = BF16_MFMA_PEAK_TF / 6.0 if arith == 1 else BF16_MFMA_PEAK_TF / 3.0 if arith == 2 else F32_MFMA_PEAK_TF
    lens_h = mask.sum(1).double().cpu()
    tokens = int(getattr(clf.model, "last_tokens", BATCH * SEQ))
    unpadded = tokens < BATCH * SEQ
    enc_flops = (clf.model.flops(BATCH, SEQ, tokens=float(lens_h.sum()), sum_len_sq=float((lens_h ** 2).sum()))
                 if unpadded else clf.model.flops(BATCH, SEQ))
    knn_flops = 2.0 * BATCH * clf.memory.index.ntotal * DIM
    line = {
        "metric": "predict() queries/sec + kNN GB/s vs HBM roofline, 768-d",
        "value": BATCH * args.steps / dt, "unit": "queries/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: bert-base-uncased arch (random init), 768-d, 100k prototypes, k=16, "
                               "batch=256, max 32 tokens per text (RAGGED: lengths ~U[8,32], SURVEY 8d; the forward is "
                               "padding-free, so a step runs tokens_per_step token rows, not 256 x 32), 4 classes",
                   "batch_per_gpu": BATCH, "seq_len": SEQ, "tokens_per_step": tokens, "tokens_per_step_padded": BATCH * SEQ,
                   "length_distribution": {"kind": "uniform integer [8, 32], first text 32", "min": int(lens_h.min()),
                                           "mean": float(lens_h.mean()), "max": int(lens_h.max())},
                   "padding_free": bool(unpadded),
                   "step_pipeline": ("predict_tokens: ac_bert_encode_cls_unpad (packing + forward, the token count reaches the host through a "
                                     "mapped slot while the embedding kernel runs: no stream synchronisation) -> search (distances, row ids) -> "
                                     "head forward (3 launches) -> ac_predict_post (scores, hit classes, softmax, blend, top-k in one launch; the "
                                     "packed result is copied to host-mapped memory by the last workgroup and the host waits on its flag) -> "
                                     "_hostfast.unpack (the reference's lists, one C pass)"),
                   "gc_frozen": True,
                   "value_gc_unfrozen": BATCH * args.steps / dt_unfrozen,
                   "prototypes": NPROTO, "dim": DIM, "k": KNN_K, "classes": NCLASS, "parallelism": "dp1",
                   "gemm_arith": ("bf16x3 split: fp32 operands = h+m+l exactly, 6 bf16 MFMA products, fp32 "
                                  "accumulate (fp32-grade; tests/test_gemm_split_gpu.py)" if arith == 1
                                  else "fp16x2 (AC_GEMM_ARITH=f16x2 in the environment: OPT-IN, operands rounded to 22 bits, "
                                       "3 fp16 MFMA products; NOT fp32-grade -- see include/acamd.h)" if arith == 2
                                  else "fp32-input MFMA"),
                   "value_f16x2_opt_in": f16,
                   "value_f32_mfma": BATCH * args.steps / dt32,
                   "ms_per_step_f32_mfma": dt32 / args.steps * 1e3,
                   "value_full_length": BATCH * args.steps / dt_full,
                   "ms_per_step_full_length": dt_full / args.steps * 1e3,
                   "tokens_per_step_full_length": tokens_full},
        "stages_ms": stages,
        "roofline_encoder": encoder_roofline(clf, stages, enc_flops, enc_peak, arith),
        "roofline_knn_batch": {"bound": "mfma", "achieved": knn_flops / stages["knn_ms"] / 1e9,
                               "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                               "frac": knn_flops / stages["knn_ms"] / 1e9 / BF16_MFMA_PEAK_TF,
                               "note": "the WHOLE kNN stage of the timed step (query plane, fp16 one-product sweep with its thresholds taken from "
                                       "its own first tile round, merge / exact re-rank, the two exact-fallback launches) against the fp16 "
                                       "dense MFMA peak: at 256 queries x 100k rows the stage is "
                                       "launch- and append-bound, not MFMA-bound (the sweep kernel alone reaches 0.38 of that peak "
                                       "at 4096 x 10M: profiles/r03/knn_batch_sweep_pmc.json)"},
    }
    if not args.no_sweep:
        free, _ = torch.cuda.mem_get_info(dev)
        n_rows = args.sweep_rows
        while n_rows * DIM * 4 > 0.8 * free and n_rows > 100_000:
            n_rows //= 2
        line["roofline"] = sweep_roofline(dev, n_rows, parity=not args.no_parity)
        line["roofline_fp16_plane"] = line["roofline"].pop("fp16_plane")
    if not args.no_parity:
        line["parity"] = step_parity(clf, hf, ids, types, mask)
    # the same loop for >= 2.5 s (the headline's 20 steps are ~0.1 s at boost clock), with the observed shader clock
    line["config"]["value_sustained"] = _guarded("value_sustained", sustained_predict, lambda: predict_step(clf, ids, types, mask), BATCH)
    # (the 20-step number again AFTER the sustained run, i.e. on a warm chip: what a caller in steady state sees)
    if not args.no_cpu_baseline:
        port = cpu_baseline(hf, clf, clf.memory.index._store[:NPROTO])
        ref = _guarded("cpu_baseline (reference)", cpu_baseline_reference, ids, mask, clf.memory.index._store[:NPROTO])
        if ref is not None and "error" not in ref:
            line["cpu_baseline"] = ref
            line["cpu_baseline_port"] = port
        else:
            port["note"] = ("oracle/_ref is not staged on this box" if ref is None else ref["error"]) + \
                           ": the oracle port stands in for the reference run"
            line["cpu_baseline"] = port
    if not args.no_extras:
        # the other BASELINE configs, measured by the same run (outside the timed region of `value`), reduced so the whole
        # command stays within ~90 s: `--config latency | cfg4 | add_examples` run them at full size on their own
        torch.cuda.empty_cache()
        line["predict_from_text"] = measure_predict_from_text(dev, clf)
        lat = measure_latency(dev, args, reps=100, made=(clf, hf), with_cpu=not args.no_cpu_baseline, cpu_seconds=3.0)
        line["latency_ms_b1"] = {k: lat[k] for k in ("value", "unit", "stages_ms", "cpu_baseline", "config", "reference_published")}
        del clf
        torch.cuda.empty_cache()
        c4 = measure_cfg4(dev, args, steps=5, warmup=2, parity_queries=8)
        line["cfg4"] = {k: c4[k] for k in ("value", "unit", "ms_per_step", "steps", "stages_ms", "roofline_encoder", "parity", "config",
                                               "value_f16x2_opt_in", "value_sustained")}
        # BASELINE configs[3] at its stated size (50 000 examples, ~25 s) with the unmodified reference's loop timed beside it
        ae = measure_add_examples(dev, args, n=50_000, modes=("as_wired",), with_cpu=False, with_cpu_reference=not args.no_cpu_baseline)
        m = ae["modes"]["as_wired"]
        line["add_examples"] = {"value": ae["value"], "unit": ae["unit"], "examples": m["examples"], "train_steps": m["train_steps"],
                                "steps_per_s": m["steps_per_s"], "host_seconds_by_phase": m["host_seconds_by_phase"],
                                "accuracy_5way": m["accuracy_5way"], "cpu_baseline": ae["cpu_baseline"], "config": ae["config"]}
        line["add_examples_with_encoder"] = _guarded("add_examples_with_encoder", measure_add_examples_text, dev, args, n=6000)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
