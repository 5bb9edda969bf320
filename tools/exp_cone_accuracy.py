"""How accurate is the reference's exponential-cone projection (SCS-style bisection on the dual variable with a Newton solve inside, as restated in
oracle/fos_oracle.py) near the cone?  The Feasibility instance of fuzz seed 20439 (alternating projections onto an affine set and two exponential cones): the
oracle's own sensitivity to a one-ulp change of b, then -- per iteration -- the distance of the projected point to the cone and the distance of the RESTATED
projection to a 60-digit one (mpmath).  CPU only: `python tools/exp_cone_accuracy.py`."""
import sys, math; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np
import fos_oracle as orc
import fuzz_parity as fz
seed = 20439
rng = np.random.default_rng([seed, 577])
n = int(rng.choice([3, 7, 30, 100, 257, 400]))
use_box = rng.random() < 0.25
cones = None if use_box else fz.random_cones(rng, n, 1)
m = max(1, min(n - 1, int(n * rng.uniform(0.05, 0.6))))
A = rng.standard_normal((m, n)) / math.sqrt(n)
algname = str(rng.choice(["DR", "AP", "GAP", "GAPA", "FISTA", "Dykstra"]))
a, a1, a2 = float(rng.uniform(0.3, 1.0)), float(rng.uniform(1.0, 1.9)), float(rng.uniform(1.0, 1.9))
beta = float(rng.uniform(0.0, 0.9))
wrap = str(rng.choice(["none", "none", "ls", "long"]))
K = orc.ConeProduct.from_lengths([(orc.CONE_CODES[k], l) for k, l in cones])
x0 = np.empty(n); K.prox(x0, rng.standard_normal(n))
b = A @ x0
def run(bvec):
    oalg = orc.AP()
    omodel = orc.FeasibilityModel(orc.Feasibility(orc.IndAffine(A, bvec), K, n), oalg)
    ost = orc.FeasibilityStatus(omodel, 10 ** 9, 1e-30, 0, 1)
    xo = np.zeros(n); seq = []
    for i in range(1, 26):
        ost.i = i; oalg.step(xo, i, ost); seq.append(xo.copy())
    return seq
r0 = run(b); r1 = run(b * (1 + 2.220446049250313e-16)); r2 = run(b * (1 - 1.1102230246251565e-16))
for i in range(25):
    print(i + 1, "%.3e %.3e" % (np.abs(r0[i] - r1[i]).max(), np.abs(r0[i] - r2[i]).max()), "step %.3e" % (np.abs(r0[i] - r0[i - 1]).max() if i else 0))

# ---- how accurate is the restated (SCS-style) projection near the boundary?  truth by mpmath (60 digits): minimise |p - v|^2 over the boundary s exp(r/s) = t
import mpmath as mp
mp.mp.dps = 60
def true_proj(v):
    r0_, s0_, t0_ = [mp.mpf(float(a)) for a in v]
    # parametrise the boundary by (r, s): t = s exp(r/s); stationarity in r and s (Newton from the float solution)
    def F(r, s):
        e = mp.e ** (r / s); t = s * e
        dtr = e; dts = e * (1 - r / s)
        return [(r - r0_) + (t - t0_) * dtr, (s - s0_) + (t - t0_) * dts]
    sol = mp.findroot(F, (r0_, s0_))
    r, s = sol[0], sol[1]
    return np.array([float(r), float(s), float(s * mp.e ** (r / s))])
oalg = orc.AP()
omodel = orc.FeasibilityModel(orc.Feasibility(orc.IndAffine(A, b), K, n), oalg)
S1 = orc.IndAffine(A, b)
x = np.zeros(n)
for i in range(1, 24):
    y = np.empty(n); S1.prox(y, x)                   # AP: x <- P_K(P_aff(x))
    xn = np.empty(n); K.prox(xn, y)
    for c0 in (0, 4):
        v = y[c0:c0 + 3]
        s_, r_, t_ = v[1], v[0], v[2]
        inc = s_ > 0 and s_ * math.exp(r_ / s_) <= t_
        if not inc and i >= 10:
            tp = true_proj(v)
            print(i, c0, "dist to cone %.2e  restated projection off the true one by %.2e" % (np.linalg.norm(tp - v), np.abs(xn[c0:c0 + 3] - tp).max()))
    x = xn
