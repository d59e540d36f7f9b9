"""Recipe for oracle/_ref/: stage the UNMODIFIED reference files the boundary tests execute.

TEST INFRASTRUCTURE.  Nothing under oracle/ is imported by the product.  oracle/_ref/ is git-ignored (no reference source
enters this repository's history) but NOT gpurun-ignored, so -- like the built .so files -- it travels to the GPU box, where
/root/reference does not exist.  `__graft_entry__.build()` runs this whenever /root/reference is present.

What is staged (byte for byte; sha256 of every file recorded in oracle/_ref/MANIFEST.json):
  tests/test_memory.py, test_ewc.py, test_multilabel.py, test_classifier.py, test_new_class_accuracy_preservation.py,
  test_order_independence.py, test_single_example_confidence.py, test_confidence_consistency.py,
  test_reported_confidence_drop.py                                       -> oracle/_ref/tests/
        the reference's own tests of this path; tests/test_reference_suite_gpu.py runs them, unmodified, against
        the PRODUCT package (`adaptive_classifier` resolves to adaptive-classifier_amd/adaptive_classifier), the Hub being
        replaced by oracle/hub_standin.py (seeded random-init models of the named architectures, synthetic WordPiece vocabulary).
  src/adaptive_classifier/*.py (the whole package: __init__, classifier, ewc, memory, models, multilabel, strategic)
                                                                          -> oracle/_ref/ref_ac/
        the UNMODIFIED reference as package `ref_ac` (relative imports only, so the name does not matter).  Two users:
        (i) INTEGRATION.md Option B -- its PrototypeMemory with `HipFlatL2Index` installed as `faiss.IndexFlatL2`;
        (ii) bench.py's `cpu_baseline` (kind "reference"): its `AdaptiveClassifier.predict_batch` (classifier.py:1308-1388)
        timed on the GPU box's host cores with oracle/faiss_shim.py as `faiss` (real faiss is not installable offline).
        `import ref_ac` needs a `faiss` module in sys.modules (the reference's memory.py:5 imports it).
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "_ref")
TESTS = ["test_memory", "test_ewc", "test_multilabel", "test_classifier", "test_new_class_accuracy_preservation",
         "test_order_independence", "test_single_example_confidence", "test_confidence_consistency",
         "test_reported_confidence_drop"]
PACKAGE = ["__init__", "classifier", "ewc", "memory", "models", "multilabel", "strategic"]
FILES = [(f"tests/{t}.py", f"tests/{t}.py") for t in TESTS] + \
        [(f"src/adaptive_classifier/{m}.py", f"ref_ac/{m}.py") for m in PACKAGE]


def stage(reference="/root/reference", dest=DEST):
    if not os.path.isdir(reference):
        return False
    manifest = {}
    for src, dst in FILES:
        s, d = os.path.join(reference, src), os.path.join(dest, dst)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(s, d)
        manifest[dst] = {"from": src, "sha256": hashlib.sha256(open(d, "rb").read()).hexdigest()}
    json.dump(manifest, open(os.path.join(dest, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    return True


if __name__ == "__main__":
    ok = stage(*(sys.argv[1:2] or ["/root/reference"]))
    print("staged into", DEST if ok else "(nothing: reference tree not present)")
