"""PSD kernel evidence on C4 in steady state: sweeps-per-projection histogram and a per-phase time split.
`python tools/psd_phases.py [warmup=250] [nblocks=512]`.  The phase split is by truncation: the kernel is launched with
phase_limit = 1..4 (ends after load+shift / warm-start product G0 = M'V_prev / Jacobi sweeps / weights+basis store) on the SAME
input, timed by HIP events (profiling class PSD); consecutive differences are the phases.  Truncated launches leave garbage in
the output buffer, so they run on fos_prox_cones (test entry) with the solver's iterate untouched."""
import sys; sys.path.insert(0, '.')
import json
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 250
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
prob = pkg.workloads.c4_block_sdp(nblocks=512, block_range=(0, nb))
d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
d.set_alg(pkg.DR()); d.set_iterate(None)
d.psd_debug(True, 0)
hist = {}
d.step(1, warm, 10 ** 12, 1e-8)
for i in range(20):                                  # 20 steady-state iterations: sweeps of every (cone, copy) projection
    d.step(warm + 1 + i, 1, 10 ** 12, 1e-8)
    for s in d.psd_sweeps(): hist[int(s)] = hist.get(int(s), 0) + 1
# two consecutive steady-state iterates: the basis is warmed on z_a (a full, untimed projection), the timed launch projects z_b
# -- one outer iteration of drift, as inside the solver
z_a = d.get_iterate()
d.step(warm + 21, 1, 10 ** 12, 1e-8)
z_b = d.get_iterate()
res, sweeps_b = {}, None
for lim in (1, 2, 3, 4, 0):
    tot, n = 0.0, 0
    for _ in range(10):
        d.profile(0); d.psd_debug(True, 0)
        d.prox_cones(z_a)
        d.profile(1); d.psd_debug(True, lim)
        d.profile_read_classes()
        d.prox_cones(z_b)
        k, ms = d.profile_read_classes()["psd"]
        tot += ms; n += k
        if lim == 0: sweeps_b = d.psd_sweeps()
    res[lim] = 1e3 * tot / max(1, n)
d.psd_debug(True, 0)
out = {"workload": "C4, %d PSD(64) cones, DR, after %d outer iterations" % (nb, warm),
       "sweeps_histogram_20_iterations": dict(sorted(hist.items())),
       "sweeps_of_the_timed_projection": {int(k): int(v) for k, v in zip(*np.unique(sweeps_b, return_counts=True))},
       "kernel_us_truncated_after": {"load+shift": res[1], "warm-start product": res[2], "jacobi sweeps": res[3], "weights+basis": res[4], "full": res[0]},
       "phase_us": {"load+shift": res[1], "G0 = M'V_prev": res[2] - res[1], "jacobi sweeps": res[3] - res[2],
                    "weights + basis store": res[4] - res[3], "rebuild + store": res[0] - res[4]}}
print(json.dumps(out, indent=1))
