"""Build hygiene of the two MFMA pipelines (no GPU needed: hipcc cross-compiles): the ring-staged kernels run at the register
limit of their occupancy (256 VGPRs, two waves per SIMD), and a few more live values make hipcc spill -- a scratch reload with
its s_waitcnt vmcnt(0) inside the k-loop drains the whole LDS-DMA ring every stage (seen twice while tuning: +8 .. 14 % time).
The device listing must show no scratch for them."""
import functools
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adaptive-classifier_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@functools.lru_cache(maxsize=None)
def listing(src):
    """device assembly of one source file (compiled once per test session)"""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-S",
                               "--cuda-device-only", os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read()


def kernel_resources(src):
    text = listing(src)
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        res[name] = (scratch, vgpr)
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,pattern", [("knn_batch.hip", "knn_batch_sweep"), ("gemm_pipe.hip", "gemm_pipe_nt"),
                                         ("knn_l2.hip", "knn_sweep_ring"), ("knn_l2.hip", "knn_plane_sweep")])
def test_ring_staged_kernels_do_not_spill(src, pattern):
    res = {k: v for k, v in kernel_resources(src).items() if pattern in k}
    assert res, "no %s kernels found in the listing" % pattern
    spilled = {k: v for k, v in res.items() if v[0] != 0}
    # One exception, bounded: the two-phase instantiation of the batch sweep (knn_batch_sweep<.., true, true>: its thresholds
    # come from its own first tile round through two grid barriers, knn_batch.hip) carries a once-per-launch exchange block that
    # costs <= 16 bytes of spill slots -- and NONE of the spill traffic may sit in the k-loop: no scratch instruction between the
    # first and the last MFMA of the listing (the loop the rule above protects).
    for name in [k for k in spilled if "knn_batch_sweep" in k and "Lb1ELb1E" in k]:
        assert spilled[name][0] <= 16, (name, spilled[name])
        body = re.search(r"^%s:[^\n]*\n(.*?)\.Lfunc_end" % re.escape(name), listing(src), re.S | re.M).group(1).split("\n")
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        sc = [i for i, l in enumerate(body) if "scratch_" in l]
        assert mf and sc and not [i for i in sc if mf[0] <= i <= mf[-1]], (name, "scratch traffic inside the k-loop", sc, mf[0], mf[-1])
        del spilled[name]
    assert not spilled, "kernels with scratch (scratch bytes, vgprs): %r" % spilled
    assert all(v[1] <= 512 for v in res.values())


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_streaming_sweeps_keep_their_prefetch_queue():
    """The two bandwidth-bound sweeps wait on COUNTED vmcnt values in their k-loops.  A flat load there (a volatile access to
    LDS through a generic pointer compiles to one) counts on vmcnt as well and forces s_waitcnt vmcnt(0) -- the whole prefetch
    queue -- once per row tile: the listing of these kernels must hold no flat memory instruction at all."""
    text = listing("knn_l2.hip")
    seen = 0
    for m in re.finditer(r"^(_ZN\S*(knn_sweep_ring|knn_plane_sweep)\S*):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M):
        seen += 1
        body = m.group(3)
        assert "flat_load" not in body and "flat_store" not in body and "flat_atomic" not in body, m.group(1)
    assert seen >= 4


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_ring_gemm_loops_wait_on_counted_vmcnt_only():
    """gemm_pipe_nt drains its LDS ring with COUNTED waits: (NS - 2) PPW before the loop, (NS - 3) PPW in each of the loop's two
    steps.  hipcc inserts an s_waitcnt vmcnt(0) of its own in front of an LDS read it cannot prove independent of the LDS-DMA
    writes in flight -- it did when the fragments were loaded as uint4 and converted at the MFMA (round 4), right after every
    counted wait, and the encoder went from 4.96 to 6.09 ms.  For every instantiation with a ring of >= 4 slots the first three
    vmcnt waits of the listing must therefore be the counted, non-zero ones."""
    text = listing("gemm_pipe.hip")
    seen = 0
    for m in re.finditer(r"^(_ZN\S*gemm_pipe_ntI(\S+?)EEvNS\S*):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M):
        args = [int(x) for x in re.findall(r"L[ib](\d+)", m.group(2) + "E")]  # EPI TM TN WMW WNW NS CP PIPE AR
        epi, tm, tn, wmw, wnw, ns, cp, pipe, ar = args
        if ns < 4 or pipe == 0:
            continue
        seen += 1
        rg = tm * wmw + tn * wnw
        ppw = -(-ar * rg // (wmw * wnw))
        waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", m.group(3))]
        assert waits[:3] == [(ns - 2) * ppw, (ns - 3) * ppw, (ns - 3) * ppw], (m.group(1), waits[:6])
    assert seen >= 10
