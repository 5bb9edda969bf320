"""Copies the round-6 evidence pass (tools/gpu_profile_r06.sh -> gpurun_out/r06final/) into profiles/r06_* and builds
profiles/r06_kkt_traffic.json (HBM-side bytes per KKT sweep from the --pmc passes: what bench.py's `roofline.traffic` reads).
`python tools/collect_profiles_r05.py [src=gpurun_out/r06final]`"""
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06final"
dst = "profiles"
for f in sorted(os.listdir(src)):
    p = os.path.join(src, f)
    if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(p) > 10:
        shutil.copy(p, os.path.join(dst, "r06_" + f))
    elif f.startswith("trace_") and f.endswith(".md"):
        shutil.copy(p, os.path.join(dst, "r06_" + f[:-3] + "_kernel_stats.md"))
    elif f.startswith("pmc_") and f.endswith(".md"):
        shutil.copy(p, os.path.join(dst, "r06_" + f.lower()))
    elif f == "pmc_c4_stream.json":
        v = json.load(open(p))
        k = list(v)[0]
        json.dump({"C4": {"kernel": k, "sweeps_per_launch": 21, "bytes_per_launch": v[k]["bytes"], "bytes_per_sweep": v[k]["bytes"] / 21,
                          "kernel_trace_us_per_launch": v[k]["kernel_trace_us"], "kernel_trace_us_per_sweep": v[k]["kernel_trace_us"] / 21,
                          "source": "profiles/r06_pmc_c4_stream.md",
                          "note": "HBM-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes; launches of exactly 20 CG iterations = 21 sweeps: the start residual and one per iteration, the cap found by an exchange of r.r alone (tools/stream_only.py)"}},
                  open(os.path.join(dst, "r06_stream_traffic.json"), "w"), indent=1)
    elif f in ("psd_time.json", "psd_orders_final.txt", "stream_stamps.txt"):
        shutil.copy(p, os.path.join(dst, "r06_" + f))
traffic = {}
for wl in ("C4", "C2", "C3", "C5"):
    p = os.path.join(src, "pmc_%s.json" % wl)
    if not os.path.exists(p):
        continue
    ker = json.load(open(p))
    main = max((k for k in ker if "deferred" not in k), key=lambda k: ker[k]["bytes"])
    traffic[wl] = {"traffic_bytes": ker[main]["bytes"], "kernels": ker, "source": "profiles/r06_pmc_%s.md" % wl.lower(), "kernel": main,
                   "kernel_trace_us_sweep_alone": ker[main]["kernel_trace_us"],
                   "note": "KKT sweep alone (tools/kkt_only.py = fos_bench_kkt: the stand-alone apply; the CG sweep streams the same bytes), HBM-side "
                           "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes (tools/pmc_sweep.sh)"}
json.dump(traffic, open(os.path.join(dst, "r06_kkt_traffic.json"), "w"), indent=1)
print(sorted(f for f in os.listdir(dst) if f.startswith("r06_")))
