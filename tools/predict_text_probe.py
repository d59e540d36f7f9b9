"""Scratch probe: clf.predict("one text") end to end (tokenizer included), device WordPiece vs host tokenizer, bert-base shape."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from transformers import BertConfig, BertModel, BertTokenizer, BertTokenizerFast
from adaptive_classifier import AdaptiveClassifier
from adaptive_classifier.encoder import HipBertEncoder
from test_tokenizer_gpu import _vocab
from oracle import synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
vocab = _vocab()
hf = BertTokenizer(vocab=vocab, do_lower_case=True)
model = BertModel(BertConfig(vocab_size=len(vocab) + 8), add_pooling_layer=False).eval()
enc = HipBertEncoder(model, device=dev)
X = synth.synth_unit_rows(100, 768, 3)
text = "please help me reset the password of my account"
for name, cfg in (("device tokenizer", {}), ("host tokenizer", {"device_tokenizer": False})):
    clf = AdaptiveClassifier("x", device="cuda:0", config=cfg, encoder=enc, tokenizer=hf)
    clf.add_embeddings([f"t{i}" for i in range(100)], [torch.from_numpy(x) for x in X], [f"c{i % 4}" for i in range(100)])
    for _ in range(10): r = clf.predict(text, k=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n): r = clf.predict(text, k=3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): enc_in = clf.tokenizer([text], max_length=512, truncation=True, padding=True, return_tensors="pt")
    torch.cuda.synchronize(); dtt = (time.perf_counter() - t0) / n
    print(f"{name}: predict(text) {dt*1e3:.3f} ms end to end; tokenizer call alone {dtt*1e3:.3f} ms; tokens {enc_in['input_ids'].shape}", r[:2])
