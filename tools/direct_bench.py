"""direct=true vs CG on a workload: set-up time and ms per outer iteration.  `python tools/direct_bench.py C2|mid|small`"""
import sys, time; sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "mid"
prob = {"C2": pkg.workloads.c2_lp, "mid": pkg.workloads.mid_mixed, "small": pkg.workloads.small_mixed,
        "lp2k": lambda: pkg.workloads.c2_lp(m=1000, n=2000, scale=45.0)}[wl]()
d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
d.set_alg(pkg.DR()); d.set_iterate(None)
d.step(1, 300, 10 ** 9, 1e-8); d.sync()
t = time.perf_counter(); d.step(301, 50, 10 ** 9, 1e-8); d.sync(); t_cg = (time.perf_counter() - t) / 50
t = time.perf_counter(); d.enable_direct(prob.A); d.sync(); t_setup = time.perf_counter() - t
d.set_iterate(None); d.reset_affine()
d.step(1, 20, 10 ** 9, 1e-8); d.sync()
t = time.perf_counter(); d.step(21, 50, 10 ** 9, 1e-8); d.sync(); t_dir = (time.perf_counter() - t) / 50
print("%s l=%d: CG path %.3f ms per outer iteration (%d CG its); direct set-up %.2f s, %.3f ms per outer iteration" %
      (wl, d.l, 1e3 * t_cg, d.cgiter(), t_setup, 1e3 * t_dir))
