"""TEST INFRASTRUCTURE -- not part of the product path.

Deterministic synthetic unit-norm rows, bit-identical to the
HIP generator `ac_synth_unit_rows` (adaptive-classifier_amd/csrc/synth.hip).

Only integer arithmetic plus correctly-rounded IEEE fp64 sqrt / divide / f64->f32
conversion is used, so numpy and the GPU produce the same bits:

    h      = splitmix64(splitmix64(seed ^ (row * 0xD1342543DE82EF95)) + col)
    x(r,c) = (sum of the four 16-bit fields of h) - 131070          (integer, ~bell shaped)
    out    = float32( float64(x) / sqrt(float64(sum_c x^2)) )

The reference feeds unit-norm embeddings into the memory (classifier.py:1275), which is
why the rows are normalised.
"""
import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)
_ROWK = np.uint64(0xD1342543DE82EF95)


def _splitmix64(z):
    z = z + _M1
    z = (z ^ (z >> np.uint64(30))) * _M2
    z = (z ^ (z >> np.uint64(27))) * _M3
    return z ^ (z >> np.uint64(31))


def synth_int_rows(n, D, seed, row_offset=0):
    """Integer pre-normalisation values x(r,c), int64 [n, D]."""
    with np.errstate(over="ignore"):
        rows = (np.arange(n, dtype=np.uint64) + np.uint64(row_offset))[:, None]
        cols = np.arange(D, dtype=np.uint64)[None, :]
        h = _splitmix64(_splitmix64(np.uint64(seed) ^ (rows * _ROWK)) + cols)
        s = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
             + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48)))
    return s.astype(np.int64) - 131070


def synth_unit_rows(n, D, seed, row_offset=0, chunk=65536):
    """float32 [n, D] unit-norm rows; same bits as the HIP generator."""
    out = np.empty((n, D), dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = synth_int_rows(e - s, D, seed, row_offset + s)
        ss = (x * x).sum(axis=1)                      # exact in int64
        nrm = np.sqrt(ss.astype(np.float64))          # correctly rounded
        nrm[nrm == 0] = 1.0
        out[s:e] = (x.astype(np.float64) / nrm[:, None]).astype(np.float32)
    return out
