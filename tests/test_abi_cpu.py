"""CPU suite: libacamd.so loads and exports exactly the symbols include/acamd.h declares;
host-only entry points (workspace planners, argument validation) behave without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "acamd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ac_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from adaptive_classifier import _native as nv
    L = nv.lib()
    declared = _header_functions()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(L, name), f"{name} declared in acamd.h but not exported by libacamd.so"
    assert sorted(nv.exported_symbols()) == declared      # the ctypes table covers the whole header
    assert L.ac_version() >= 1


def test_workspace_planners_and_validation_without_gpu():
    from adaptive_classifier import _native as nv
    L = nv.lib()
    b = ctypes.c_size_t(0)
    assert L.ac_knn_l2_topk_workspace(10_000_000, 768, 4096, 32, ctypes.byref(b)) == 0 and b.value > 0
    assert L.ac_knn_l2_topk_workspace(100, 768, 8, 4, ctypes.byref(b)) == 0
    assert L.ac_knn_l2_topk_workspace(100, 768, 8, 1000, ctypes.byref(b)) == 0           # small store: exact path, any k
    assert L.ac_knn_l2_topk_workspace(100, 4096, 8, 8, ctypes.byref(b)) == 0              # ... and any D
    assert L.ac_knn_l2_topk_workspace(100000, 768, 8, 1000, ctypes.byref(b)) == -2        # big store + k beyond the sweep
    assert b"k=1000" in L.ac_last_error()
    assert L.ac_knn_l2_topk_workspace(100000, 4096, 8, 8, ctypes.byref(b)) == -2          # big store + D too wide for LDS
    dims = nv.ac_head_dims(768, 768, 384, 4)
    assert L.ac_head_param_count(ctypes.byref(dims)) == 887428                            # P at C = 4
    assert L.ac_head_workspace(ctypes.byref(dims), 32, ctypes.byref(b)) == 0 and b.value > 0
    cfg = nv.ac_bert_config(768, 12, 12, 3072, 30522, 512, 2, 1e-12)
    assert L.ac_bert_workspace(ctypes.byref(cfg), 256, 32, ctypes.byref(b)) == 0 and b.value > 256 * 32 * 768 * 4
    bad = nv.ac_bert_config(768, 12, 8, 3072, 30522, 512, 2, 1e-12)                       # head dim 96
    assert L.ac_bert_workspace(ctypes.byref(bad), 1, 8, ctypes.byref(b)) == -2


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from adaptive_classifier import _native as nv
    from adaptive_classifier.index import HipFlatL2Index
    idx = HipFlatL2Index(8)
    idx.add(torch.zeros(3, 8))
    assert idx.ntotal == 3                        # host bookkeeping works
    with pytest.raises(nv.NativeError):
        idx.search(torch.zeros(1, 8).numpy(), 1)  # no CPU search path exists
    from adaptive_classifier import AdaptiveClassifier
    with pytest.raises(nv.NativeError):
        AdaptiveClassifier("bert-base-uncased", device="cpu")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "adaptive-classifier_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_process_wide_switches_are_inert_without_the_test_hook_flag():
    """include/acamd.h "TEST HOOKS": in a process that did not set AC_TEST_HOOKS=1 the process-wide setters refuse
    (AC_EUNSUPPORTED = -2) and change nothing; with the flag (this suite's conftest sets it) they work.  No GPU needed."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from adaptive_classifier import _native as nv; L = nv.lib(); "
            "a = L.ac_gemm_get_arith(); r = [L.ac_gemm_set_arith(0), L.ac_gemm_set_variant(1), L.ac_gemm_set_ln_fusion(0), "
            "L.ac_gemm_set_krot(1), L.ac_gemm_set_pipe_table(b''), L.ac_gemm_set_pipe_table_f16(b'')]; "
            "m = L.ac_set_persistent_kernels(-1); m2 = L.ac_set_persistent_kernels(0); m3 = L.ac_set_persistent_kernels(-1); "
            "print(r, a == L.ac_gemm_get_arith(), m == m2 == m3, b'AC_TEST_HOOKS' in L.ac_last_error())" % os.path.join(ROOT, "adaptive-classifier_amd"))
    env = {k: v for k, v in os.environ.items() if k != "AC_TEST_HOOKS"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip()
    assert out == "[-2, -2, -2, -2, -2, -2] True True True", out
    env["AC_TEST_HOOKS"] = "1"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip()
    assert out.startswith("[0, 0, 0, 0, 0, 0] False False"), out
