import numpy as np, torch
from scipy.special import erf, erfc
from gelu_fit import fit, TMAX
m, c = fit(8)
c32 = c.astype(np.float32)
print("coeffs (low->high):", [float(x).hex() for x in c32], [float(x) for x in c32])
def f32(x): return np.asarray(x, np.float64).astype(np.float32).astype(np.float64)
def erf_fast(t):            # t float32 array
    a = np.minimum(np.abs(t.astype(np.float64)), TMAX)
    p = np.full_like(a, float(c32[8]))
    for k in range(7, -1, -1):
        p = f32(p * a + float(c32[k]))       # fma
    E = f32(p * a)
    e = f32(np.exp2(-E))
    r = f32(1.0 - e)
    return np.copysign(r, t).astype(np.float32)
x = np.concatenate([np.linspace(-8, 8, 4_000_001), np.random.default_rng(0).standard_normal(2_000_000) * 1.5, np.array([0.0, 1e-10, -1e-10, 1e-5, 30.0, -30.0])]).astype(np.float32)
t = (x * np.float32(0.70710678118654752440)).astype(np.float32)
ef = erf_fast(t)
et = erf(t.astype(np.float64))
print("max |erf_fast - erf|:", np.abs(ef - et).max(), "at t=", t[np.abs(ef - et).argmax()])
g_fast = (np.float32(0.5) * x * (np.float32(1.0) + ef)).astype(np.float32)
g_true = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
g_torch = torch.nn.functional.gelu(torch.from_numpy(x)).numpy()
print("gelu: fast vs exact max abs", np.abs(g_fast - g_true).max(), " torch f32 vs exact", np.abs(g_torch - g_true).max())
rel = np.abs(g_fast - g_true) / np.maximum(np.abs(g_true), 1e-30)
relt = np.abs(g_torch - g_true) / np.maximum(np.abs(g_true), 1e-30)
sel = np.abs(x) > 1e-6
print("gelu rel err (|x|>1e-6): fast max", rel[sel].max(), "at", x[sel][rel[sel].argmax()], " torch max", relt[sel].max(), "at", x[sel][relt[sel].argmax()])
for lo, hi in [(-8,-4),(-4,-2),(-2,-1),(-1,0),(0,1),(1,2),(2,4),(4,8)]:
    s = (x >= lo) & (x < hi)
    print(lo, hi, "fast abs", np.abs(g_fast - g_true)[s].max(), "torch abs", np.abs(g_torch - g_true)[s].max())
