"""Writes tests/golden/reference_test_inputs.npz: the inputs of the reference's test/testDRandGAPA.jl
(`Random.seed!(2); A = randn(40, 50); b = randn(40, 1)`) as Julia < 1.5 and Julia 1.5/1.6 drew them, regenerated
by oracle/julia_random.py (Julia's generator restated; no Julia in this image), with the optima that file holds
(:11-17), and the inputs of test/testfeasibility.jl:2-7 (Julia >= 1.5 draw) -- reference-held known answers for whole solves.   python tests/golden/make_reference_test_inputs.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import julia_random as jr  # noqa: E402

out = {}
for tag, gen in (("pre15", "pre1.5"), ("v15", "1.5")):
    A, b, opt = jr.readme_nnls_data(gen)
    out["A_" + tag], out["b_" + tag], out["opt_" + tag] = A, b, np.float64(opt)
out["feas_xsol"], out["feas_A"] = jr.feasibility_test_data()
np.savez(os.path.join(ROOT, "tests", "golden", "reference_test_inputs.npz"), **out)
print({k: getattr(v, "shape", v) for k, v in out.items()})
