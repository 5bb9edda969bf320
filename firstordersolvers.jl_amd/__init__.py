"""
firstordersolvers.jl_amd -- MI355X-native hot path of FirstOrderSolvers.jl.

Host-side mirror of the reference's solver interface (GAP/DR/AP/GAPA/FISTA constructors,
FOSMathProgModel with loadproblem/optimize/status/getobjval/getsolution, model.history) over the
C-ABI shared library `csrc/libfoship.so` (hand-written HIP kernels for gfx950).  There is no CPU
fallback: every compute call goes through the library and fails loudly if it is missing.
"""
from . import _lib as lib          # noqa: F401
from . import workloads            # noqa: F401
from .interface import (AP, DR, FISTA, GAP, GAPA, GAPP, Dykstra, FOSAlgorithm, FOSMathProgModel, HipHSDE, HSDEStatus,  # noqa: F401
                        LineSearchWrapper, LongstepWrapper, Solution, solve, HEADER_CG, HEADER_DIRECT,
                        ConeProduct, Feasibility, FeasibilityModel, FeasibilitySolution, HipFeasibility, IndAffine, IndBox, solve_feasibility)
from . import sharding             # noqa: F401
