"""Recipe for oracle/_ref/: stage the UNMODIFIED reference files the boundary tests execute.

TEST INFRASTRUCTURE.  Nothing under oracle/ is imported by the product.  oracle/_ref/ is git-ignored (no reference source
enters this repository's history) but NOT gpurun-ignored, so -- like the built .so files -- it travels to the GPU box, where
/root/reference does not exist.  `__graft_entry__.build()` runs this whenever /root/reference is present.

What is staged (byte for byte; sha256 of every file recorded in oracle/_ref/MANIFEST.json):
  tests/test_memory.py, tests/test_ewc.py, tests/test_multilabel.py   -> oracle/_ref/tests/
        the reference's own tests of this path; tests/test_reference_suite_gpu.py runs them, unmodified, against
        the PRODUCT package (`adaptive_classifier` resolves to adaptive-classifier_amd/adaptive_classifier).
  src/adaptive_classifier/memory.py, models.py                         -> oracle/_ref/ref_ac/
        the reference's PrototypeMemory, imported as package `ref_ac` with `HipFlatL2Index` installed as `faiss.IndexFlatL2`
        (INTEGRATION.md Option B): the reference's own host logic on the product's index.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "_ref")
FILES = [
    ("tests/test_memory.py", "tests/test_memory.py"),
    ("tests/test_ewc.py", "tests/test_ewc.py"),
    ("tests/test_multilabel.py", "tests/test_multilabel.py"),
    ("src/adaptive_classifier/memory.py", "ref_ac/memory.py"),
    ("src/adaptive_classifier/models.py", "ref_ac/models.py"),
]


def stage(reference="/root/reference", dest=DEST):
    if not os.path.isdir(reference):
        return False
    manifest = {}
    for src, dst in FILES:
        s, d = os.path.join(reference, src), os.path.join(dest, dst)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(s, d)
        manifest[dst] = {"from": src, "sha256": hashlib.sha256(open(d, "rb").read()).hexdigest()}
    # the package marker is ours (the reference's __init__ imports classifier.py and with it transformers + faiss)
    open(os.path.join(dest, "ref_ac", "__init__.py"), "w").write(
        "# staged by oracle/stage_ref.py: the reference's memory.py / models.py as package `ref_ac` (test infrastructure)\n")
    json.dump(manifest, open(os.path.join(dest, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    return True


if __name__ == "__main__":
    ok = stage(*(sys.argv[1:2] or ["/root/reference"]))
    print("staged into", DEST if ok else "(nothing: reference tree not present)")
