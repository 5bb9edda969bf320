"""In-kernel time stamps of the window-panel sweep (library built with -DFOS_WIN_STAMPS: `make -C firstordersolvers.jl_amd/csrc clean all
EXTRA=-DFOS_WIN_STAMPS`): `python tools/win_stamps.py [C5]` prints, per phase of a segment, the mean / max time over the segments of
workgroup 100 for every wavefront (microseconds of the 100 MHz clock)."""
import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "C5"
prob = {"C4": pkg.workloads.c4_block_sdp, "C2": pkg.workloads.c2_lp, "C3": pkg.workloads.c3_socp, "C5": pkg.workloads.c5_mixed}[wl]()
d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
d.set_iterate(np.random.default_rng(0).standard_normal(d.N))
ms = d.bench_kkt(50)
print("kkt avg us %.2f" % (1e3 * ms / 50))
lib = pkg.lib.load()
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # wavefronts per workgroup (16: -DFOS_WIN_TALL)
buf = (C.c_longlong * (NW * 64 * 8))()
rc = lib.fos_debug_win_stamps(buf, NW * 64 * 8)
if rc != 0:
    sys.exit("library built without -DFOS_WIN_STAMPS (rc %d)" % rc)
st = np.array(buf[:], dtype=np.int64).reshape(NW, 64, 8)
nseg = int((st[0, :63, 0] > 0).sum())                       # (slot 63 holds the kernel-level stamps)
print("segments stamped:", nseg)
names = ["issue loads", "barrier 1", "wait + window -> LDS", "barrier 2", "multiply + row sums", "loop tail -> next"]
for w in range(NW):
    t = st[w, :nseg, :6].astype(np.float64) / 100.0
    d_ = np.diff(t, axis=1)
    nxt = t[1:, 0] - t[:-1, 5]
    row = ["%5.2f/%5.2f" % (d_[:, i].mean(), d_[:, i].max()) for i in range(5)] + ["%5.2f" % nxt.mean()]
    print("wave %d  segment %.2f us:  " % (w, (t[-1, 5] - t[0, 0]) / nseg) + "  ".join("%s %s" % (n, r) for n, r in zip(names, row)))
g = st[:, 63, :6].astype(np.float64) / 100.0
for w in (0, NW - 1):
    print("wave %d: kernel entry -> segment loop %.2f us, loop %.2f, -> barrier %.2f, epilogue %.2f, -> reduction stored %.2f" % ((w,) + tuple(np.diff(g[w]))))
print("first stamp -> last stamp of wave 0: %.2f us" % ((st[0, nseg - 1, 5] - st[0, 0, 0]) / 100.0))
