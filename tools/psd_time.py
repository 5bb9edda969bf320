"""Time of ONE batched PSD(64) projection in the solver's steady state, per kernel choice and batch size.
`python tools/psd_time.py [warmup=250] [nb ...]` -- C4 restricted to nb blocks (2 nb matrices per projection); for each of
FOS_PSD_REFINE=0 (Jacobi: workgroup / wavefront kernel by batch size) and =1 (refinement by matrix products + Jacobi for the
flagged matrices) a handle is built, warmed up, and the projection of the NEXT iterate is timed by HIP events on the solver's
stream (profiling class PSD) behind an untimed projection of the current one -- one outer iteration of drift, as in the solver."""
import sys; sys.path.insert(0, '.')
import os, json
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 250
nbs = [int(a) for a in sys.argv[2:]] or [64, 128, 256, 512]
out = {}
for nb in nbs:
    prob = pkg.workloads.c4_block_sdp(nblocks=512, block_range=(0, nb))
    row = {}
    for mode in ("0", "1"):
        os.environ["FOS_PSD_REFINE"] = mode
        d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
        d.set_alg(pkg.DR()); d.set_iterate(None)
        d.psd_debug(True, 0)
        d.step(1, warm, 10 ** 12, 1e-8)
        zs = [d.get_iterate()]
        for q in range(2):
            d.step(warm + 1 + q, 1, 10 ** 12, 1e-8)
            zs.append(d.get_iterate())
        res = {}
        for lim in ((0,) if mode == "0" else (0, 11, 16, 17)):
            tot, n, st = 0.0, 0, None
            for _ in range(6):
                d.profile(0); d.psd_debug(True, 0)
                d.prox_cones(zs[0]); d.prox_cones(zs[1])          # two untimed projections: the bases the timed one extrapolates from
                d.profile(1); d.profile_read_classes(); d.psd_debug(True, lim)
                d.prox_cones(zs[2])
                k, ms = d.profile_read_classes()["psd"]
                tot += ms; n += k
                if lim == 0: st = d.psd_sweeps()
            res[lim] = round(1e3 * tot / max(1, n), 2)
            if lim == 0:
                vals, cnt = np.unique(st, return_counts=True)
                hist = {int(a): int(b) for a, b in zip(vals, cnt)}
        d.psd_debug(True, 0)
        row["refine" if mode == "1" else "jacobi"] = {"us": res[0], "record_histogram": hist}
        if mode == "1":
            row["refine"]["us_truncated_after"] = {"load + start basis": res[11], "all iterations": res[16], "Newton-Schulz step": res[17], "full": res[0]}
        d.close()
    out["%d matrices" % (2 * nb)] = row
print(json.dumps({"workload": "C4 blocks, DR, projection of iterate %d behind iterate %d" % (warm + 1, warm),
                  "record": "refinement: 100 (+ 1000: extrapolated start accepted) + 16 rotations + iterations (both starts); Jacobi: sweeps", "per_batch": out}, indent=1))
