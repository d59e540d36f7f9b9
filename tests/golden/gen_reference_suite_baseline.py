"""Which of the reference's own tests pass ON THE REFERENCE ITSELF when the Hub is replaced by oracle/hub_standin.py?

    python tests/golden/gen_reference_suite_baseline.py [--logs DIR]       # build container, CPU, ~25 min without --logs

The reference's hot-path test files name Hub checkpoints; offline they run against seeded random-init models of the named
architectures (oracle/hub_standin.py) and `faiss` = oracle/faiss_shim.py.  Tests that assert SEMANTIC quality (a confidence above
0.7 after 100 examples, accuracy above chance) need pretrained weights and fail on the unmodified reference under the same
stand-in; one (`test_save_load_multilabel`) fails on the reference for a reason of its own (its `load()` passes `use_onnx` to a
constructor that does not take it).  This script records the outcome of every test of those files on the REFERENCE
(PYTHONPATH=/root/reference/src, CPU) in tests/golden/reference_suite_on_reference.json.  tests/test_reference_suite_gpu.py then
requires the PRODUCT to pass every test the reference passes, minus exclusions it names one by one.

--logs DIR: parse `<file>.log` files of an earlier `pytest -rA` run instead of running again.
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FILES = ["test_classifier", "test_order_independence", "test_single_example_confidence", "test_confidence_consistency",
         "test_reported_confidence_drop", "test_new_class_accuracy_preservation", "test_multilabel", "test_ewc", "test_memory"]


def run(name, logdir):
    if logdir and os.path.exists(os.path.join(logdir, name + ".log")):
        return open(os.path.join(logdir, name + ".log")).read()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=ROOT + os.pathsep + "/root/reference/src",
               AC_STANDIN_FAISS_SHIM="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "oracle.hub_standin", f"/root/reference/tests/{name}.py", "-q",
                        "-p", "no:cacheprovider", "-W", "ignore", "-rA", "--tb=short"], env=env, capture_output=True, text=True,
                       cwd="/tmp")
    return r.stdout + r.stderr


def main():
    logdir = sys.argv[sys.argv.index("--logs") + 1] if "--logs" in sys.argv else None
    out = {}
    for name in FILES:
        log = run(name, logdir)
        res = {}
        for m in re.finditer(r"^(PASSED|FAILED|ERROR) \S*?%s\.py::(\S+)" % name, log, re.M):
            res[m.group(2)] = m.group(1).lower()
        why = {}
        for m in re.finditer(r"^_+ (\S+) _+$\n(.*?)(?=^_+ \S+ _+$|^=+ )", log, re.M | re.S):
            errs = re.findall(r"^E\s+(.*)$", m.group(2), re.M)
            if m.group(1) in res and res[m.group(1)] != "passed" and errs:
                why[m.group(1)] = errs[0][:160]
        out[name + ".py"] = {t: ({"outcome": o} if o == "passed" else {"outcome": o, "why": why.get(t, "")}) for t, o in sorted(res.items())}
        print(name, {o: list(res.values()).count(o) for o in set(res.values())})
    json.dump({"how": "unmodified reference (/root/reference/src) on CPU, Hub -> oracle/hub_standin.py, faiss -> oracle/faiss_shim.py",
               "files": out}, open(os.path.join(HERE, "reference_suite_on_reference.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
