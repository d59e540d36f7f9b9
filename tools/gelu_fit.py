import numpy as np
from scipy.special import erf, erfc
np.set_printoptions(precision=17)
TMAX = 4.0
def target_h(t):
    return -np.log2(erfc(t)) / t
def fit(deg, iters=60):
    # weighted minimax-ish fit of h(t) ~ poly(t) on (0, TMAX], error measured in erf
    t = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * (TMAX / 2) + TMAX / 2
    t = np.sort(t)
    h = target_h(t)
    E = t * h
    w_erf = np.log(2) * 2.0 ** (-E) * t          # d erf / d h
    V = np.vander(t, deg + 1, increasing=True)
    w = w_erf.copy()
    best = None
    for it in range(iters):
        c, *_ = np.linalg.lstsq(V * w[:, None], h * w, rcond=None)
        err = (V @ c - h) * w_erf
        m = np.abs(err).max()
        if best is None or m < best[0]:
            best = (m, c.copy())
        # Lawson reweighting
        w = w * (np.abs(err) / m + 1e-3) ** 0.5
        w = w / w.max() * w_erf.max()
    return best
for deg in range(6, 13):
    m, c = fit(deg)
    print(deg, m)
