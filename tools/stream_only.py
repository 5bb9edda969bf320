"""The streamed resident CG solve alone: `python tools/stream_only.py C4 0 [solves]` -- `solves` launches of cg_stream_kernel of exactly K = 20
iterations each (21 sweeps: the start residual and one per iteration; the cap is found by an exchange of r.r alone) on C4 from a random point; used with
PROG=tools/stream_only.py KPAT=cg_stream bash tools/pmc_sweep.sh <tag> C4  (HBM-side bytes per LAUNCH = 21 sweeps)."""
import sys; sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prob = pkg.workloads.c4_block_sdp()
d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
d.set_cg_variant("resident")
assert d.resident_stats()["form"] == "streamed"
rng = np.random.default_rng(0)
x0, rhs = rng.standard_normal(d.N), rng.standard_normal(d.N)
K = 20
d.profile(1); d.profile_read_classes()
for _ in range(reps):
    _, it = d.cg_kkt(x0, rhs, 1e-300, K)
    assert it == K
n, ms = d.profile_read_classes()["resident"]
nnz = prob.nnz
print("C4 streamed resident solve: %d launches of %d iterations, %.1f us per launch = %.2f us per sweep; algorithmic bytes per sweep %.1f MB" %
      (n, K, 1e3 * ms / n, 1e3 * ms / n / (K + 1), (8.0 * nnz + 8.0 * prob.m) / 1e6))
