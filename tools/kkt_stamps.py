"""Start / end stamps of every workgroup of the row-block KKT sweep (library built with -DFOS_KKT_STAMPS): `python tools/kkt_stamps.py [C4]`
prints the sweep's span, the workgroups' durations and how many are resident over time (the tail of the launch)."""
import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
prob = {"C4": pkg.workloads.c4_block_sdp, "C2": pkg.workloads.c2_lp, "C3": pkg.workloads.c3_socp, "C5": pkg.workloads.c5_mixed,
        "C4S": lambda: pkg.workloads.c4_block_sdp(nblocks=512, block_range=(0, 64))}[wl]()          # C4S: the 64 blocks one of eight ranks holds
d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
d.set_iterate(np.random.default_rng(0).standard_normal(d.N))
ms = d.bench_kkt(50)
print("kkt avg us %.2f" % (1e3 * ms / 50), d.operator_stats())
lib = pkg.lib.load()
buf = (C.c_longlong * (2 * 16384))()
if lib.fos_debug_kkt_stamps(buf, 2 * 16384) != 0:
    sys.exit("library built without -DFOS_KKT_STAMPS")
st = np.array(buf[:], dtype=np.int64).reshape(16384, 2)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
s, e = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0
print("workgroups %d  span %.2f us  duration mean %.2f  min %.2f  max %.2f" % (len(st), e.max(), (e - s).mean(), (e - s).min(), (e - s).max()))
T = e.max()
for a in np.arange(0.0, T, T / 12):
    b = a + T / 12
    res = ((s < b) & (e > a)).sum()
    print("  %5.1f - %5.1f us: resident workgroups (any overlap) %5d, starting %5d, ending %5d" % (a, b, res, ((s >= a) & (s < b)).sum(), ((e >= a) & (e < b)).sum()))
allst = np.array(buf[:], dtype=np.int64).reshape(16384, 2)
widx = np.nonzero(allst[:, 0] > 0)[0]
order = np.argsort(-(e - s))[:24]
print("slowest workgroups (blockIdx: start, duration us):", ", ".join("%d: %.1f %.1f" % (widx[k], s[k], (e - s)[k]) for k in order))
if len(sys.argv) > 2:                                       # end times by XCD (blockIdx % 8) and by dispatch round (blockIdx // 256)
    idx = np.nonzero(np.array(buf[:], dtype=np.int64).reshape(16384, 2)[:, 0] > 0)[0]
    long_ = (e - s) > 0.5 * (e - s).max()
    for x in range(8):
        m = long_ & (idx % 8 == x)
        print("  XCD %d: %4d workgroups, end mean %.1f  min %.1f  max %.1f" % (x, m.sum(), e[m].mean(), e[m].min(), e[m].max()))
    for r in range(int(idx.max()) // 256 + 1):
        m = long_ & (idx // 256 == r)
        if m.sum():
            print("  blockIdx %4d..%4d: %4d workgroups, end mean %.1f  min %.1f  max %.1f" % (256 * r, 256 * r + 255, m.sum(), e[m].mean(), e[m].min(), e[m].max()))
