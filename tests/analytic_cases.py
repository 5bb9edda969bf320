"""Textbook conic programs with CLOSED-FORM optima, in the conic form the reference's `loadproblem!` receives
(FOSSolverInterface.jl:17-36: min c'x s.t. b - A x in K1, x in K2) with MathProgBase's cone conventions, which `conemap`
(src/cones.jl:4-14) hands to ProximalOperators.jl:
    :SOC        (t, x): ||x|| <= t                       :SOCRotated (p, q, x): ||x||^2 <= 2 p q, p, q >= 0
    :ExpPrimal  (x, y, z): y e^(x/y) <= z, y > 0         :ExpDual    (u, v, w): -u e^(v/u) <= e w, u < 0
    :SDP        svec, off-diagonals times sqrt 2
No reference test touches :SOCRotated / :ExpPrimal / :ExpDual ("parity unpinned"); these problems pin what CAN be pinned
without the third-party source: the entry order and the constants of each cone's definition at whole-solve level, independent
of the oracle -- a projection onto a differently ordered or differently scaled cone ends at a different optimum (each case lists the
value a plausible wrong convention would give).  Each case: (name, A, b, c, K1, K2, optimum, x*, wrong_optima)."""
import math

import numpy as np
import scipy.sparse as sp


def _case(name, A, b, c, K1, K2, opt, x, wrong=()):
    return dict(name=name, A=sp.csc_matrix(np.atleast_2d(np.asarray(A, dtype=float))), b=np.asarray(b, dtype=float), c=np.asarray(c, dtype=float),
                K1=K1, K2=K2, opt=float(opt), x=np.asarray(x, dtype=float), wrong=tuple(wrong))


def cases():
    out = []
    e = math.e
    # 1. min t  s.t. (x, 1, t) in ExpPrimal, x >= 1           -> t = e at x = 1     (vars x, t)
    out.append(_case("exp-primal-rows", [[-1, 0], [0, 0], [0, -1], [-1, 0]], [0, 1, 0, -1], [0, 1],
                     [("ExpPrimal", 3), ("NonNeg", 1)], [("Free", 2)], e, [1, e],
                     wrong=(1.0, 0.0)))                    # (z, y, x) order: e^(t) <= x unbounded below / other orders give 0 or 1
    # 2. max x  s.t. (x, y, z) in ExpPrimal (a VARIABLE cone), y = 1, z = 2     -> x = log 2
    out.append(_case("exp-primal-vars-log", [[0, 1, 0], [0, 0, 1]], [1, 2], [-1, 0, 0],
                     [("Zero", 2)], [("ExpPrimal", 3)], -math.log(2.0), [math.log(2.0), 1, 2], wrong=(-2.0, -math.log(0.5))))
    # 3. min w  s.t. (u, v, w) in ExpDual (variable cone), u = -1, v = 2        -> w = e^(-3)      (-u e^(v/u) <= e w)
    out.append(_case("exp-dual-vars", [[1, 0, 0], [0, 1, 0]], [-1, 2], [0, 0, 1],
                     [("Zero", 2)], [("ExpDual", 3)], math.exp(-3.0), [-1, 2, math.exp(-3.0)], wrong=(math.exp(-2.0), 0.0)))
    # 4. min w  s.t. (-1, v, w) in ExpDual as ROWS, v >= 0.5 free otherwise: -u e^(v/u) = e^(-v) <= e w -> w = e^(-v-1) minimised by the
    #    largest v allowed: v <= 3                                             -> w = e^(-4)      (vars v, w)
    out.append(_case("exp-dual-rows", [[0, 0], [-1, 0], [0, -1], [1, 0]], [-1, 0, 0, 3], [0, 1],
                     [("ExpDual", 3), ("NonNeg", 1)], [("Free", 2)], math.exp(-4.0), [3, math.exp(-4.0)], wrong=(math.exp(-3.0),)))
    # 5. min p + q  s.t. (p, q, z) in SOCRotated, z = 1                         -> p = q = 1/sqrt 2, value sqrt 2   (pq >= z^2 would give 2)
    r = 1.0 / math.sqrt(2.0)
    out.append(_case("soc-rotated-vars", [[0, 0, 1]], [1], [1, 1, 0], [("Zero", 1)], [("SOCRotated", 3)], math.sqrt(2.0), [r, r, 1], wrong=(2.0, 1.0)))
    # 6. min p  s.t. (p, 2, (3, 4)) in SOCRotated as ROWS                      -> 25 <= 4 p, p = 6.25             (pq >= ||x||^2: 12.5)
    out.append(_case("soc-rotated-rows", [[-1], [0], [0], [0]], [0, 2, 3, 4], [1], [("SOCRotated", 4)], [("Free", 1)], 6.25, [6.25], wrong=(12.5, 5.0)))
    # 7. min t  s.t. (t, (3, 4) - x (1, 1)) in SOC  -> distance of (3, 4) from the line span(1, 1): 1/sqrt 2 at x = 3.5   (vars t, x)
    out.append(_case("soc-rows", [[-1, 0], [0, 1], [0, 1]], [0, 3, 4], [1, 0], [("SOC", 3)], [("Free", 2)], r, [r, 3.5], wrong=(5.0,)))
    # 8. min X11 + X22  s.t. X PSD (variable cone, svec), X12 = 1               -> X = [[1, 1], [1, 1]], value 2; svec = (1, sqrt 2, 1)
    #    (without the sqrt 2 scaling the constraint reads X12 = 1/sqrt 2 ... value sqrt 2)
    out.append(_case("sdp-vars", [[0, 1, 0]], [math.sqrt(2.0)], [1, 0, 1], [("Zero", 1)], [("SDP", 3)], 2.0, [1, math.sqrt(2.0), 1], wrong=(2 * math.sqrt(2.0), math.sqrt(2.0))))
    # 9. three exponential cones coupled by an equality: min sum t_i  s.t. (x_i, 1, t_i) in ExpPrimal, sum x_i = 0
    #    -> x = 0 (convexity + symmetry), value 3                                  (vars x1..x3, t1..t3)
    A = np.zeros((10, 6))
    b = np.zeros(10)
    for i in range(3):
        A[3 * i, i] = -1.0
        b[3 * i + 1] = 1.0
        A[3 * i + 2, 3 + i] = -1.0
    A[9, 0:3] = 1.0
    out.append(_case("exp-sum", A, b, [0, 0, 0, 1, 1, 1], [("ExpPrimal", 3)] * 3 + [("Zero", 1)], [("Free", 6)], 3.0, [0, 0, 0, 1, 1, 1]))
    return out
