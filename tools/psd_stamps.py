"""In-kernel time stamps of psd64_refine_kernel (library built with -DFOS_PSD_STAMPS: `touch firstordersolvers.jl_amd/csrc/psd.hip; make -C
firstordersolvers.jl_amd/csrc EXTRA=-DFOS_PSD_STAMPS`): `python tools/psd_stamps.py [warmup=250] [nb ...]` -- C4 restricted to nb blocks, the projection of the
next iterate in the solver's steady state (as tools/psd_time.py); prints, for workgroup 0, what every wavefront spent between consecutive stamps
(microseconds of the 100 MHz clock)."""
import sys, ctypes as C; sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 250
nbs = [int(a) for a in sys.argv[2:]] or [64, 512]
NAMES = {0: "entry", 1: "loads requested + M, V -> LDS", 2: "|M|_F barrier", 3: "(iteration) Vb / loop head", 4: "G = M V", 5: "Rayleigh, G'", 6: "barrier (d, V visible)",
         7: "d_i requested + N = V'G'", 8: "E", 10: "|E| reduce", 11: "V + V E", 12: "barrier (V read, sums)", 13: "V -> registers, LDS",
         14: "barrier + accept", 15: "NS: V'V", 16: "NS: V (I + R/2) + barriers", 17: "P = V D+ V' -> LDS", 18: "basis out + barrier", 19: "output"}
lib = pkg.lib.load()
for nb in nbs:
    prob = pkg.workloads.c4_block_sdp(nblocks=512, block_range=(0, nb))
    d = pkg.HipHSDE(prob.A, prob.b, prob.c, prob.K1, prob.K2)
    d.set_alg(pkg.DR()); d.set_iterate(None)
    d.psd_debug(True, 0)
    d.step(1, warm, 10 ** 12, 1e-8)
    zs = [d.get_iterate()]
    for q in range(2):
        d.step(warm + 1 + q, 1, 10 ** 12, 1e-8)
        zs.append(d.get_iterate())
    d.prox_cones(zs[0]); d.prox_cones(zs[1]); d.prox_cones(zs[2])
    d.sync()
    rec = d.psd_sweeps()
    buf = (C.c_longlong * (4 * 128))()
    rc = lib.fos_debug_psd_stamps(buf, 4 * 128)
    if rc != 0:
        sys.exit("library built without -DFOS_PSD_STAMPS (rc %d)" % rc)
    st = np.array(buf[:], dtype=np.int64).reshape(4, 128)
    print("== %d matrices; record of matrix 0: %d" % (2 * nb, int(rec[0])))
    rows = []
    for w in range(4):
        ids, t = st[w] % 64, (st[w] // 64).astype(np.float64) / 100.0
        n = int(np.argmax(ids == 19)) + 1
        rows.append((ids[:n], t[:n]))
    n = min(len(r[0]) for r in rows)
    t00 = min(r[1][0] for r in rows)
    for k in range(1, n):
        i = int(rows[0][0][k])
        print("  %-34s " % NAMES.get(i, str(i)) + "  ".join("w%d %5.2f" % (w, rows[w][1][k] - rows[w][1][k - 1]) for w in range(4))
              + "   | at %6.2f" % (max(rows[w][1][k] for w in range(4)) - t00))
    d.close()
