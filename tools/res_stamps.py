"""In-kernel time stamps of the resident CG solve (library built with -DFOS_RES_STAMPS:
    make -C firstordersolvers.jl_amd/csrc VARIANT=res_stamps EXTRA=-DFOS_RES_STAMPS
    FOSHIP_LIB=firstordersolvers.jl_amd/csrc/libfoship_res_stamps.so python tools/res_stamps.py [nblocks]).
Runs the 64-block shard of C4 (what one of eight ranks holds) to its steady state on the resident solve and prints, per recorded
workgroup and wavefront, where an iteration's time goes: sweep / publish / wait for the records / barrier / totals / ranks / update."""
import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prob = pkg.workloads.c4_block_sdp(nblocks=512, block_range=(0, nb))
d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
print(d.resident_stats())
d.set_cg_variant("resident")
d.set_alg(pkg.DR())
d.set_iterate(None)
d.step(1, 220, 10 ** 12, 1e-8)
print("cg iterations of the last solve:", d.cgiter())
lib = pkg.lib.load()
n = 4 * 16 * 64 * 16
buf = (C.c_longlong * n)()
lib.fos_debug_res_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int]
if lib.fos_debug_res_stamps(buf, n) != 0:
    sys.exit("library built without -DFOS_RES_STAMPS")
st = np.array(buf[:], dtype=np.int64).reshape(4, 16, 64, 16)
plan = d.resident_stats()
ncomp = plan['waves_per_workgroup']
cnames = ["sweep + wave sums", "barrier A (slowest wavefront)", "exchange (barrier B)", "row update"]
mnames = ["barrier A (sweep)", "reduce + publish", "poll own words", "wait other comm waves", "totals (+ ranks)", "scalars, columns, tau", "barrier B"]
# clock rate of the stamps (s_memtime ticks per microsecond) from the two calibration pairs
rates = []
for w in range(4):
    for who in range(16):
        a, b = st[w, who, 63, 6:8], st[w, who, 62, 6:8]
        if a[1] > 0 and b[1] > a[1]:
            rates.append((b[0] - a[0]) / ((b[1] - a[1]) / 100.0))
rate = float(np.median(rates)) if rates else 100.0
print("stamp clock: %.1f ticks per us (from %d pairs)" % (rate, len(rates)))
st = st.copy().astype(np.float64)
cal = st[:, :, 62:, :].copy()
# the launch's own head and tail, on compute wavefront 0 of workgroup 0 (shader clock): entry -> first loop top, last stamp -> exit
e0, e1 = cal[0, 0, 1, 6], cal[0, 0, 0, 6]
tops = [st[0, 0, k, 0] for k in range(62) if st[0, 0, k, 0] > 0]
if e0 > 0 and e1 > e0 and tops:
    print("workgroup 0, compute wavefront 0: kernel entry -> first sweep %.2f us (tiles and vectors -> registers / LDS), %d exchanges in %.2f us, last loop top -> exit %.2f us; entry -> exit %.2f us"
          % ((tops[0] - e0) / rate, len(tops), (tops[-1] - tops[0]) / rate, (e1 - tops[-1]) / rate, (e1 - e0) / rate))
st[:, :, 62:, :] = 0
st = st / rate * 100.0                      # -> the unit the code below divides by 100
t00 = st[st > 0].min()
for w, wn in enumerate(("wg 0", "wg 1", "wg G/2", "wg G-1")):
    # per compute wavefront: when it has swept, relative to the workgroup's earliest loop top of that iteration (iteration 5)
    it5 = 5
    tops = [st[w, v, it5, 0] for v in range(ncomp) if st[w, v, it5, 0] > 0]
    if tops:
        t0w = min(tops)
        print("%s, iteration %d, per compute wavefront (loop top -> swept, us after the earliest top): " % (wn, it5) +
              "  ".join("w%d %.2f->%.2f" % (v, (st[w, v, it5, 0] - t0w) / 100.0, (st[w, v, it5, 1] - t0w) / 100.0) for v in range(ncomp) if st[w, v, it5, 0] > 0))
    for who, whn, names, last in ((0, "compute wave 0", cnames, 4), (ncomp, "comm wave 0", mnames, 7)):
        s = st[w, who]
        its = [k for k in range(64) if s[k, 0] > 0 and s[k, last] > 0]
        if not its:
            continue
        print("%s %s: solve start +%.2f us, %d exchanges recorded (0 = the start sweep)" % (wn, whn, (s[its[0], 0] - t00) / 100.0, len(its)))
        ph = np.array([[(s[k, j + 1] - s[k, j]) / 100.0 for j in range(last)] for k in its])
        tot = np.array([(s[k, last] - s[k, 0]) / 100.0 for k in its])
        for k in its[:3] + its[-1:]:
            print("   it %2d: total %6.2f us | " % (k, (s[k, last] - s[k, 0]) / 100.0) + "  ".join("%s %.2f" % (nm, v) for nm, v in zip(names, ph[its.index(k)])))
        print("   mean over iterations 1..: total %.2f us | " % tot[1:].mean() + "  ".join("%s %.2f" % (nm, v) for nm, v in zip(names, ph[1:].mean(axis=0))))
        if who == ncomp:           # finer stamps inside the first communication wavefront: offsets from barrier A / from the arrival of everybody's words
            f = lambda a, b: np.mean([(s[k, a] - s[k, b]) / 100.0 for k in its[1:] if s[k, a] > 0 and s[k, b] > 0])
            print("   finer: after A: column sums added +%.2f, wavefront sums added +%.2f, four sums formed +%.2f, published +%.2f | after all words: totals +%.2f, +ranks %.2f, w rows +%.2f, scalars + updates +%.2f, LDS written +%.2f"
                  % (f(8, 1), f(9, 1), f(10, 1), f(2, 1), f(11, 4), f(5, 4), f(12, 4), f(13, 4), f(6, 4)))
d.close()
