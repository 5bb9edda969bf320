"""bench.py pieces that need no GPU: the decoding of the PSD kernels' per-matrix records, the tolerance-floor iteration, and that the
measured region never touches the oracle (the product path must not route through it: only `cpu_baseline` may)."""
import ast
import importlib.util
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_psd_record_decoding():
    b = _bench()
    rec = np.array([3, 4, 102, 1103, 1104 + 16, 120, 9, 1000000 + 230000 + 1102])     # Jacobi sweeps, refinement records (+ debug digits)
    refined, it, rot, extrap = b.decode_psd_records(rec)
    assert refined.tolist() == [False, False, True, True, True, True, False, True]
    assert it[refined].tolist() == [2, 3, 4, 4, 2]
    assert rot[refined].tolist() == [0, 0, 1, 1, 0]
    assert extrap.tolist() == [False, False, False, True, True, False, False, True]


def test_tolerance_floor_iteration_matches_the_schedule():
    """affinepluslinear.jl:108-112: tol = max(0.2^sqrt(i), l eps); the bench warms up to the first i at which the floor holds."""
    b = _bench()
    for l in (150, 15001, 70001, 1081345):
        i = b.tolerance_floor_iteration(l)
        floor = l * 2.220446049250313e-16
        assert 0.2 ** np.sqrt(i) <= floor < 0.2 ** np.sqrt(i - 1)


def test_only_the_cpu_baseline_touches_the_oracle():
    """Every import of oracle/ modules in bench.py sits inside the function `cpu_baseline`, which is defined AFTER the timed region (run_case) and called
    only under `not args.no_cpu_baseline` on one rank (the headline's baseline and the sample for `value_as_specified`): the oracle is the checker and the
    timed CPU baseline, never the thing measured."""
    src = (ROOT / "bench.py").read_text()
    tree = ast.parse(src)
    lines = src.splitlines()
    funcs = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "cpu_baseline"]
    assert len(funcs) == 1
    f = funcs[0]
    run_case = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "run_case"][0]
    assert f.lineno > run_case.end_lineno                        # (not inside, not in front of, the timed region)
    seen = 0
    for node in ast.walk(tree):
        names = []
        if isinstance(node, ast.Import):
            names = [a.name for a in node.names]
        elif isinstance(node, ast.ImportFrom):
            names = [node.module or ""]
        if any(n.split(".")[0] in ("fos_oracle", "fos_cport") for n in names):
            assert f.lineno < node.lineno <= f.end_lineno, (node.lineno, names)
            seen += 1
    assert seen >= 2
    calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "cpu_baseline"]
    assert len(calls) == 2
    for c in calls:                                              # each call sits under a guard on --no-cpu-baseline, in a world == 1 branch
        head = "\n".join(lines[max(0, c.lineno - 12):c.lineno])
        assert "not args.no_cpu_baseline" in head, c.lineno
        assert "world == 1" in "\n".join(lines[max(0, c.lineno - 14):c.lineno]), c.lineno
