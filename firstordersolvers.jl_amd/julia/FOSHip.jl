# FOSHip.jl -- thin Julia shim that routes FirstOrderSolvers.jl's HSDE hot path to libfoship.so (MI355X / HIP).
#
# STATUS: written against the reference sources but NOT EXECUTED -- the build container has no Julia binary
# (SURVEY.md section 8(c)).  The same C ABI is exercised end to end by the Python mirror
# (firstordersolvers.jl_amd/interface.py, tests/test_gpu_parity.py); this file is the binding a maintainer adds.
#
# How it plugs in (reference files in brackets):
#   * `include("FOSHip.jl")` at the end of src/FirstOrderSolvers.jl (after solvers/*.jl are loaded).
#   * The user passes `gpu=true` with any solver:  solve!(problem, DR(eps=1e-8, gpu=true)).  Keyword arguments are
#     captured in `alg.options` and splatted into `model.options` [FOSSolverInterface.jl:5, types.jl:57]; unknown keys
#     are ignored by the reference, so nothing else changes.
#   * `init_algorithm!` [gap.jl:23, gapa.jl:27, fista.jl:20, dykstra.jl:19] is intercepted for FOSMathProgModel when
#     `gpu=true`: it returns a `HipData` (a FOSSolverData) instead of GAPData/GAPAData/...; everything downstream
#     dispatches on that data type:  `iterate` [solverwrapper.jl:20-41], `Base.step`, `getsol`, `getcgiter`.
#   * FOSMathProgModel, loadproblem!, optimize!, status/getobjval/getsolution, model.history, the printed table and
#     HSDE_populatesolution stay the reference's own code.
module FOSHip

using ..FirstOrderSolvers
import ..FirstOrderSolvers: FOSAlgorithm, FOSSolverData, FOSMathProgModel, HSDEStatus, GAP, GAPA, FISTA, Dykstra,
                            init_algorithm!, getsol, getcgiter, iterate, printstatusheader, printstatusiter,
                            savedata, ConeProduct
import ProximalOperators
using SparseArrays, Printf

const libfoship = get(ENV, "FOSHIP_LIB", "libfoship.so")

# ---- mirror of include/foship.h ------------------------------------------------------------------------------
const FOS_ALG_GAP, FOS_ALG_GAPA, FOS_ALG_FISTA, FOS_ALG_DYKSTRA = Cint(0), Cint(1), Cint(2), Cint(3)
const STATUS_SYMBOLS = (:Continue, :Optimal, :Unbounded, :Infeasible)        # FOS_STATUS_*

struct CheckResult            # struct fos_check_result
    p::Cdouble; d::Cdouble; g::Cdouble; ctx::Cdouble; bty::Cdouble
    kappa::Cdouble; tau::Cdouble; norm_axs::Cdouble; norm_aty::Cdouble; norm_b::Cdouble; norm_c::Cdouble
    cgiter::Int64; status::Int32; cg_maxiter_hit::Int32
end

function check(code::Cint)
    code == 0 && return
    msg = unsafe_string(ccall((:fos_last_error, libfoship), Cstring, ()))
    error("libfoship error $code: $msg")
end

# cone codes: keys of conemap [cones.jl:4-14]
conecode(::ProximalOperators.IndFree) = Int32(0)
conecode(::ProximalOperators.IndZero) = Int32(1)
conecode(::ProximalOperators.IndNonnegative) = Int32(2)
conecode(::ProximalOperators.IndNonpositive) = Int32(3)
conecode(::ProximalOperators.IndSOC) = Int32(4)
conecode(::ProximalOperators.IndRotatedSOC) = Int32(5)
conecode(::ProximalOperators.IndPSD) = Int32(6)
conecode(::ProximalOperators.IndExpPrimal) = Int32(7)
conecode(::ProximalOperators.IndExpDual) = Int32(8)

function conearrays(K::ConeProduct)
    N = length(K.cones)
    types = Int32[conecode(K.cones[i]) for i in 1:N]
    starts = Int64[first(K.ranges[i]) for i in 1:N]
    lens = Int64[length(K.ranges[i]) for i in 1:N]
    return types, starts, lens
end

# ---- the device-resident solver data ---------------------------------------------------------------------------
mutable struct HipData <: FOSSolverData
    handle::Ptr{Cvoid}
    m::Int
    n::Int
    cgiter::Int64
    lsinterval::Int64                     # > 0: LineSearchWrapper around the algorithm [wrappers/linesearch.jl]
    gappinterval::Int64                   # > 0: GAPP, its search interval [solvers/gapproj.jl]
    function HipData(model::FOSMathProgModel, device::Integer)
        A = model.A                       # SparseMatrixCSC{Float64,Int}: colptr/rowval are Int64 and 1-based, as the ABI wants
        m, n = size(A)
        b = convert(Vector{Float64}, vec(model.b))
        c = convert(Vector{Float64}, vec(model.c))
        t1, s1, l1 = conearrays(model.K1)
        t2, s2, l2 = conearrays(model.K2)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve A b c t1 s1 l1 t2 s2 l2 begin
            check(ccall((:fos_create, libfoship), Cint,
                        (Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                         Int64, Ptr{Int32}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int32}, Ptr{Int64}, Ptr{Int64},
                         Cint, Ref{Ptr{Cvoid}}),
                        m, n, A.colptr, A.rowval, A.nzval, b, c,
                        length(t1), t1, s1, l1, length(t2), t2, s2, l2, Cint(device), h))
        end
        d = new(h[], m, n, 0, 0, 0)
        finalizer(x -> (x.handle != C_NULL && ccall((:fos_destroy, libfoship), Cint, (Ptr{Cvoid},), x.handle); x.handle = C_NULL), d)
        return d
    end
end

algargs(a::GAP) = (FOS_ALG_GAP, a.α, a.α1, a.α2, 0.0)              # gap.jl:6-13
algargs(a::GAPA) = (FOS_ALG_GAPA, a.α, 0.0, 0.0, a.β)              # gapa.jl:9-15
algargs(a::FISTA) = (FOS_ALG_FISTA, a.α, 0.0, 0.0, 0.0)            # fista.jl:6-11
algargs(a::Dykstra) = (FOS_ALG_DYKSTRA, 0.0, 0.0, 0.0, 0.0)        # dykstra.jl:5-9
algargs(a::FirstOrderSolvers.GAPP) = (FOS_ALG_GAP, a.α, a.α1, a.α2, 0.0)   # gapproj.jl:5-13: GAP + fos_set_gapp(iproj)

set_alg!(d::HipData, alg) = check(ccall((:fos_set_alg, libfoship), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cdouble),
                                        d.handle, algargs(alg)...))

usegpu(model::FOSMathProgModel) = get(model.options, :gpu, false) === true

# ---- init_algorithm!: return HipData when gpu=true, otherwise fall through to the reference method --------------
for T in (:GAP, :GAPA, :FISTA, :Dykstra, :(FirstOrderSolvers.GAPP))
    @eval function init_algorithm!(alg::$T, model::FOSMathProgModel)
        if usegpu(model)
            # the reference's closure [HSDE.jl:24-27], built directly: get_sets_and_status would also construct the host-side
            # AffinePlusLinear (5 N-vectors of CG state) and DualConeProduct, which the device path never touches.
            # m, n as DualConeProduct's constructor takes them [cones.jl:121]
            sm, sn = model.K1.ranges[end][end], model.K2.ranges[end][end]
            dmode = Ref{Int32}(0)          # fos_get_direct_mode, read below: 1 dense inverse, 2 block form (no CG: the table drops its cg column), 3 CG at its floor
            status_generator = (mo, checki, eps, verbose, debug) ->
                HSDEStatus(sm, sn, 0, mo, :Continue, checki, eps, verbose, false, alg.direct && dmode[] in (1, 2), time_ns(), model.init_duration, debug)
            data = HipData(model, get(model.options, :device, 0))
            set_alg!(data, alg)
            if haskey(model.options, :cg_variant)      # device-side key: which CG recurrence the affine projection runs (FOS_CG_* of foship.h)
                check(ccall((:fos_set_cg_variant, libfoship), Cint, (Ptr{Cvoid}, Int32), data.handle, Int32(model.options[:cg_variant])))
            end
            if alg isa FirstOrderSolvers.GAPP            # "projected GAP": every iproj-th iteration is a 21-point search on the device
                check(ccall((:fos_set_gapp, libfoship), Cint, (Ptr{Cvoid}, Int64), data.handle, Int64(alg.iproj)))
                data.gappinterval = alg.iproj
            end
            if alg.direct            # HSDE.jl:12-15: S1 = IndAffine([Q -I], 0) -> exact projection, (I + Q Q')^-1 formed once on the device
                A = model.A
                GC.@preserve A check(ccall((:fos_enable_direct, libfoship), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Cdouble}),
                                           data.handle, A.colptr, A.rowval, A.nzval))
                check(ccall((:fos_get_direct_mode, libfoship), Cint, (Ptr{Cvoid}, Ref{Int32}), data.handle, dmode))
            end
            return data, status_generator
        end
        return invoke(init_algorithm!, Tuple{$T,FirstOrderSolvers.AbstractFOSModel}, alg, model)
    end
end

# ---- LineSearchWrapper(GAP / GAPA) [wrappers/linesearch.jl]: the wrapped algorithm's handle with the search switched on; the
#      search itself (tmp1 = x; x = S2!(S1!(x)); 31 trial step lengths; x = tmp1 + alpha_best res) runs on the device.
function init_algorithm!(ls::LineSearchWrapper, model::FOSMathProgModel)
    if usegpu(model)
        data, status_generator = init_algorithm!(ls.alg, model)          # errors for algorithms other than GAP / GAPA below
        check(ccall((:fos_set_linesearch, libfoship), Cint, (Ptr{Cvoid}, Int64), data.handle, ls.lsinterval))
        data.lsinterval = ls.lsinterval
        return data, status_generator
    end
    return invoke(init_algorithm!, Tuple{LineSearchWrapper,FirstOrderSolvers.AbstractFOSModel}, ls, model)
end

# ---- LongstepWrapper(GAP / GAPA / FISTA / Dykstra) [wrappers/longstep.jl, saveplanes.jl]: the wrapped algorithm's handle with the plane saving and the
#      projection onto the saved planes switched on (fos_set_longstep); planes and projection stay on the device, `iterate` is the wrapped algorithm's.
import ..FirstOrderSolvers: LongstepWrapper
function init_algorithm!(long::LongstepWrapper, model::FOSMathProgModel)
    if usegpu(model)
        !FirstOrderSolvers.support_longstep(long.alg) && @error "Algorithm alg does not support longstep"        # longstep.jl:28
        data, status_generator = init_algorithm!(long.alg, model)
        check(ccall((:fos_set_longstep, libfoship), Cint, (Ptr{Cvoid}, Int64, Int64), data.handle, Int64(long.longinterval), Int64(long.nsave)))
        return data, status_generator
    end
    return invoke(init_algorithm!, Tuple{LongstepWrapper,FirstOrderSolvers.AbstractFOSModel}, long, model)
end
iterate(long::LongstepWrapper, data::HipData, status::HSDEStatus, x, max_iters) = iterate(long.alg, data, status, x, max_iters)
getsol(long::LongstepWrapper, data::HipData, x) = getsol(long.alg, data, x)

# what linesearch.jl:51,63,69 print during a search, from the device's record of the last one
function print_gapp(data::HipData)                       # gapproj.jl:51,57
    log = Vector{Float64}(undef, 23)
    check(ccall((:fos_gapp_log, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, log))
    for k = 1:21
        println("normtest: $(log[k])")
    end
    println("αbest: $(log[22])")
end
function print_linesearch(data::HipData)
    log = Vector{Float64}(undef, 34)
    check(ccall((:fos_linesearch_log, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, log))
    println("test, $(log[1])")
    α = 0.1
    for k = 0:30
        α = α * 1.8
        println("α: $α, $(log[2+k])")
    end
    println("α: $(log[33])")
end

getcgiter(d::HipData) = d.cgiter                                              # defaults.jl:25-30

# ---- what checkstatus does once the device has produced the scalars [HSDEStatus.jl:39-65] ----------------------
function record!(stat::HSDEStatus, data::HipData, r::CheckResult)
    t = time_ns() - stat.init_time
    i, model = stat.i, stat.model
    data.cgiter = r.cgiter
    if stat.debug > 0           # savedata [HSDEStatus.jl:125-139]; debug=2 vectors need fos_get_checked (below)
        x = y = s = Float64[]
        if stat.debug > 1
            z = Vector{Float64}(undef, 2 * (data.m + data.n + 1))
            check(ccall((:fos_get_checked, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, z))
            nu = data.n + data.m + 1
            x, y, s = z[1:data.n], z[data.n+1:data.n+data.m], z[nu+data.n+1:nu+data.n+data.m]
        end
        savedata(i, r.p, r.d, r.g, r.ctx, r.bty, r.kappa, r.tau, x, y, s, t, model, stat.debug)
    end
    if stat.verbose > 0 && !stat.direct                                        # HSDEStatus.jl:43-47
        push!(model.history, :cgiter, i, r.cgiter)
        printstatusiter(i, r.p, r.d, r.g, r.ctx, r.bty, r.kappa / r.tau, r.cgiter, t)
    elseif stat.verbose > 0                                                    # :48-50
        printstatusiter(i, r.p, r.d, r.g, r.ctx, r.bty, r.kappa / r.tau, t)
    end
    r.cg_maxiter_hit != 0 && @warn "CG reached max iterations, result may be inaccurate"   # conjugategradients.jl:53
    stat.status = STATUS_SYMBOLS[r.status+1]
    if stat.status == :Optimal && stat.verbose > 0
        println("Found solution i=$i")
    end
    stat.checked = true
    return
end

# ---- iterate [solverwrapper.jl:20-41], device resident: the host sees one ccall per check interval ---------------
function iterate(alg::FOSAlgorithm, data::HipData, status::HSDEStatus, x, max_iters)
    t1 = time()
    printstatusheader(status)
    check(ccall((:fos_set_iterate, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, x))     # initx / z0
    i = 0
    done, checked, res = Ref{Int64}(0), Ref{Int32}(0), Ref{CheckResult}()
    while i < max_iters
        count = min(max_iters - i, status.checki - (i % status.checki))
        ls = data.lsinterval
        ls > 0 && (count = min(count, ls - (i % ls)))                 # stop at every line-search iteration: its output is printed
        gp = data.gappinterval
        gp > 0 && (count = min(count, gp - (i % gp)))
        check(ccall((:fos_step, libfoship), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Cdouble, Ref{Int64}, Ref{Int32}, Ref{CheckResult}),
                    data.handle, i + 1, count, status.checki, status.eps, done, checked, res))
        i += done[]
        status.i = i
        ls > 0 && i % ls == 0 && print_linesearch(data)
        gp > 0 && i % gp == 0 && print_gapp(data)
        if checked[] != 0
            record!(status, data, res[])
            status.status != :Continue && break
        else
            status.checked = false
        end
    end
    guess = Vector{Float64}(undef, length(x))
    force = status.checked ? Int32(0) : Int32(1)                                   # solverwrapper.jl:31-34
    check(ccall((:fos_getsol, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int32, Cdouble, Ref{CheckResult}),
                data.handle, guess, force, status.eps, res))
    force != 0 && record!(status, data, res[])
    if status.verbose > 0
        println("Time for iterations: ")
        println("$(time() - t1) s")
    end
    return guess
end

# ---- single-step entry points for callers written against step/getsol (one upload, one ccall, one download per
#      iteration: correct but PCIe bound -- `iterate` above is the path to use).  The host vector is uploaded on EVERY call, so
#      a caller that edits x between steps is honoured.  LineSearchWrapper and LongstepWrapper run on the device through `iterate` (above).
function Base.step(alg::FOSAlgorithm, data::HipData, x, i, status::HSDEStatus, longstep = nothing)
    longstep === nothing || error("longstep/linesearch wrappers are not available with gpu=true")
    check(ccall((:fos_set_iterate, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, x))
    done, checked, res = Ref{Int64}(0), Ref{Int32}(0), Ref{CheckResult}()
    check(ccall((:fos_step, libfoship), Cint,
                (Ptr{Cvoid}, Int64, Int64, Int64, Cdouble, Ref{Int64}, Ref{Int32}, Ref{CheckResult}),
                data.handle, i, 1, status.checki, status.eps, done, checked, res))
    if checked[] != 0
        record!(status, data, res[])
    else
        status.checked = false
    end
    check(ccall((:fos_get_iterate, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, x))
    return
end

function getsol(alg::FOSAlgorithm, data::HipData, x)
    guess = Vector{Float64}(undef, length(x))
    res = Ref{CheckResult}()
    check(ccall((:fos_getsol, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int32, Cdouble, Ref{CheckResult}),
                data.handle, guess, Int32(0), 0.0, res))
    return guess
end

# ---- checkpoint / resume of a device-resident run: the iterate, S1's persistent state (CGdata.xinit, AffinePlusLinear.i:
#      affinepluslinear.jl:58-69,114) and the algorithm's *Data struct (FISTAData.y/.xold/.t fista.jl:15-25, DykstraData.p/.q
#      dykstra.jl:12-23, GAPAData.alpha12 gapa.jl:29) -- everything `iterate` would need to go on from iteration i + 1 elsewhere
function checkpoint(data::HipData, N::Integer)
    x, xinit, a, b = (Vector{Float64}(undef, N) for _ in 1:4)
    i, firstrun, scal = Ref{Int64}(0), Ref{Int32}(0), zeros(2)
    check(ccall((:fos_get_iterate, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, x))
    check(ccall((:fos_get_affine_state, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ref{Int64}, Ref{Int32}), data.handle, xinit, i, firstrun))
    check(ccall((:fos_get_alg_state, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), data.handle, a, b, scal))
    return (x = x, xinit = xinit, i = i[], firstrun = firstrun[] != 0, a = a, b = b, t = scal[1], alpha12 = scal[2])
end
function restore!(data::HipData, cp)
    check(ccall((:fos_set_iterate, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, cp.x))
    cp.firstrun || check(ccall((:fos_set_affine_state, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64), data.handle, cp.xinit, Int64(cp.i)))
    check(ccall((:fos_set_alg_state, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), data.handle, cp.a, cp.b, [cp.t, cp.alpha12]))
    return data
end

# ---- Feasibility form [problemforms/Feasibility/Feasibility.jl, FeasibilityStatus.jl]: solve!(Feasibility(S1, S2, n), alg; gpu=true)
#      with S1, S2 among ProximalOperators.IndAffine (dense A), IndBox -- the sets of test/testfeasibility.jl -- and
#      FirstOrderSolvers.ConeProduct on the device; any other ProximableFunction through a host callback (fos_feas_set_callback).
#      init_algorithm! returns a HipFeasData; iterate dispatches on it; FeasibilityModel, populate_solution and the printed table
#      stay the reference's own code.
import ..FirstOrderSolvers: FeasibilityModel, FeasibilityStatus

# fos_prox_fn: int32 fn(void* ctx, int64 n, const double* x, double* y) -- ctx is the set (boxed in a Ref), y = prox_S(x)
function prox_trampoline(ctx::Ptr{Cvoid}, n::Int64, x::Ptr{Cdouble}, y::Ptr{Cdouble})::Int32
    try
        S = unsafe_pointer_to_objref(ctx)[]
        ProximalOperators.prox!(unsafe_wrap(Array, y, n), S, unsafe_wrap(Array, x, n))
        return Int32(0)
    catch
        return Int32(1)                                     # (an exception must not unwind through the C frames)
    end
end
const PROX_TRAMPOLINE = Ref{Ptr{Cvoid}}(C_NULL)             # @cfunction pointers are made at run time (__init__)

mutable struct HipFeasData <: FOSSolverData
    handle::Ptr{Cvoid}
    lsinterval::Int64                     # > 0: LineSearchWrapper around the algorithm
    gappinterval::Int64                   # > 0: GAPP, its search interval
    sets::Vector{Base.RefValue{Any}}      # callback sets, kept alive as long as the handle
    function HipFeasData(model::FeasibilityModel, device::Integer)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:fos_feas_create, libfoship), Cint, (Int64, Int32, Ref{Ptr{Cvoid}}), Int64(model.n), Int32(device), h))
        d = new(h[], 0, 0, Base.RefValue{Any}[])
        finalizer(x -> ccall((:fos_feas_destroy, libfoship), Cint, (Ptr{Cvoid},), x.handle), d)
        for (which, S) in ((Int32(1), model.S1), (Int32(2), model.S2))
            if S isa ProximalOperators.IndBox && (S.lb isa AbstractArray || S.ub isa AbstractArray)
                lo = S.lb isa AbstractArray ? Vector{Float64}(vec(S.lb)) : fill(Float64(S.lb), model.n)
                hi = S.ub isa AbstractArray ? Vector{Float64}(vec(S.ub)) : fill(Float64(S.ub), model.n)
                GC.@preserve lo hi check(ccall((:fos_feas_set_box_arrays, libfoship), Cint, (Ptr{Cvoid}, Int32, Ptr{Cdouble}, Ptr{Cdouble}), d.handle, which, lo, hi))
            elseif S isa ProximalOperators.IndBox
                check(ccall((:fos_feas_set_box, libfoship), Cint, (Ptr{Cvoid}, Int32, Cdouble, Cdouble), d.handle, which, Float64(S.lb), Float64(S.ub)))
            elseif S isa ConeProduct                        # the reference's own cone stack [cones.jl:31-94]
                types, _, lens = conearrays(S)
                GC.@preserve types lens check(ccall((:fos_feas_set_cones, libfoship), Cint, (Ptr{Cvoid}, Int32, Int64, Ptr{Int32}, Ptr{Int64}),
                                                    d.handle, which, Int64(length(types)), types, lens))
            elseif S isa ProximalOperators.IndAffine && S.A isa AbstractMatrix && !(S.A isa SparseArrays.AbstractSparseMatrix)
                At = Matrix{Float64}(transpose(S.A))        # IndAffine(A, b), dense: the C ABI takes A row-major = column-major A'
                b = Vector{Float64}(S.b)
                GC.@preserve At b check(ccall((:fos_feas_set_affine, libfoship), Cint, (Ptr{Cvoid}, Int32, Int64, Ptr{Cdouble}, Ptr{Cdouble}),
                                              d.handle, which, Int64(size(S.A, 1)), At, b))
            elseif S isa ProximalOperators.IndAffine && S.A isa SparseArrays.SparseMatrixCSC
                # IndAffine(A, b) over a sparse A stays sparse on the device (any n): the arrays of the SparseMatrixCSC as they are (1-based)
                As = SparseArrays.SparseMatrixCSC{Float64,Int64}(S.A)
                b = Vector{Float64}(S.b)
                GC.@preserve As b check(ccall((:fos_feas_set_affine_sparse, libfoship), Cint,
                                              (Ptr{Cvoid}, Int32, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cdouble}, Ptr{Cdouble}),
                                              d.handle, which, Int64(size(As, 1)), As.colptr, As.rowval, As.nzval, b))
            else                                            # any other ProximableFunction: prox!(y, S, x) on host vectors [Feasibility.jl:2-6]
                push!(d.sets, Ref{Any}(S))                  # (rooted: the library keeps a pointer to it)
                check(ccall((:fos_feas_set_callback, libfoship), Cint, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
                            d.handle, which, PROX_TRAMPOLINE[], pointer_from_objref(d.sets[end])))
            end
        end
        return d
    end
end

getcgiter(::HipFeasData) = 0

for T in (:GAP, :GAPA, :FISTA, :Dykstra)
    @eval function init_algorithm!(alg::$T, model::FeasibilityModel)
        if get(model.options, :gpu, false) === true
            data = HipFeasData(model, get(model.options, :device, 0))
            check(ccall((:fos_feas_set_alg, libfoship), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cdouble), data.handle, algargs(alg)...))
            status_generator = (mo, checki, eps, verbose, debug) ->            # Feasibility.jl:76-78
                FeasibilityStatus(mo.n, 0, mo, fill(NaN, mo.n), Array{Array{Float64,1},1}(), :Continue, checki, eps, verbose, false, true,
                                  time_ns(), mo.init_duration, debug)
            return data, status_generator
        end
        return invoke(init_algorithm!, Tuple{$T,FirstOrderSolvers.AbstractFOSModel}, alg, model)
    end
end

# LineSearchWrapper(GAP / GAPA) on the Feasibility form (test/testfeasibility.jl:36-44): the wrapped algorithm's handle with the
# search switched on; what linesearch.jl prints during a search comes from fos_feas_linesearch_log
function init_algorithm!(long::LongstepWrapper, model::FeasibilityModel)      # [wrappers/longstep.jl] on the Feasibility form: fos_feas_set_longstep
    if usegpu(model)
        !FirstOrderSolvers.support_longstep(long.alg) && @error "Algorithm alg does not support longstep"
        data, status_generator = init_algorithm!(long.alg, model)
        check(ccall((:fos_feas_set_longstep, libfoship), Cint, (Ptr{Cvoid}, Int64, Int64), data.handle, Int64(long.longinterval), Int64(long.nsave)))
        return data, status_generator
    end
    return invoke(init_algorithm!, Tuple{LongstepWrapper,FirstOrderSolvers.AbstractFOSModel}, long, model)
end
iterate(long::LongstepWrapper, data::HipFeasData, status::FeasibilityStatus, x, max_iters) = iterate(long.alg, data, status, x, max_iters)
getsol(long::LongstepWrapper, data::HipFeasData, x) = getsol(long.alg, data, x)
function init_algorithm!(ls::LineSearchWrapper, model::FeasibilityModel)
    if get(model.options, :gpu, false) === true
        data, status_generator = init_algorithm!(ls.alg, model)
        check(ccall((:fos_feas_set_linesearch, libfoship), Cint, (Ptr{Cvoid}, Int64), data.handle, ls.lsinterval))
        data.lsinterval = ls.lsinterval
        return data, status_generator
    end
    return invoke(init_algorithm!, Tuple{LineSearchWrapper,FirstOrderSolvers.AbstractFOSModel}, ls, model)
end
function print_linesearch(data::HipFeasData)
    log = Vector{Float64}(undef, 34)
    check(ccall((:fos_feas_linesearch_log, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, log))
    println("test, $(log[1])")
    α = 0.1
    for k = 0:30
        α = α * 1.8
        println("α: $α, $(log[2+k])")
    end
    println("α: $(log[33])")
end

# GAPP ("projected GAP", solvers/gapproj.jl) on the Feasibility form: every iproj-th iteration is a 21-point search on the device
import ..FirstOrderSolvers: GAPP
function init_algorithm!(alg::GAPP, model::FeasibilityModel)
    if get(model.options, :gpu, false) === true
        data = HipFeasData(model, get(model.options, :device, 0))
        check(ccall((:fos_feas_set_gapp, libfoship), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Int64), data.handle, alg.α, alg.α1, alg.α2, Int64(alg.iproj)))
        data.gappinterval = alg.iproj
        status_generator = (mo, checki, eps, verbose, debug) ->
            FeasibilityStatus(mo.n, 0, mo, fill(NaN, mo.n), Array{Array{Float64,1},1}(), :Continue, checki, eps, verbose, false, true,
                              time_ns(), mo.init_duration, debug)
        return data, status_generator
    end
    return invoke(init_algorithm!, Tuple{GAPP,FirstOrderSolvers.AbstractFOSModel}, alg, model)
end
function print_gapp(data::HipFeasData)                       # gapproj.jl:51,57
    log = Vector{Float64}(undef, 23)
    check(ccall((:fos_feas_gapp_log, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, log))
    for k = 1:21
        println("normtest: $(log[k])")
    end
    println("αbest: $(log[22])")
end

function iterate(alg::FOSAlgorithm, data::HipFeasData, status::FeasibilityStatus, x, max_iters)
    t1 = time()
    printstatusheader(status)
    check(ccall((:fos_feas_set_iterate, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), data.handle, x))
    i = 0
    done, st, err, checked = Ref{Int64}(0), Ref{Int32}(0), Ref{Cdouble}(NaN), Ref{Int32}(0)
    report = function ()
        t = time_ns() - status.init_time
        status.debug > 0 && savedata(status.i, err[], x, t, status.model, status.extra, min(status.debug, 1))     # FeasibilityStatus.jl:42-45
        status.verbose > 0 && printstatusiter(status.i, err[], t)                                              # :46-54 (direct: no cg column)
        status.status = STATUS_SYMBOLS[st[] + 1]
        status.verbose > 0 && status.status == :Optimal && println("Found solution i=$(status.i)")
        status.checked = true
    end
    while i < max_iters
        count = min(max_iters - i, status.checki - (i % status.checki))
        ls = data.lsinterval
        ls > 0 && (count = min(count, ls - (i % ls)))
        gp = data.gappinterval
        gp > 0 && (count = min(count, gp - (i % gp)))
        check(ccall((:fos_feas_step, libfoship), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Cdouble, Ref{Int64}, Ref{Int32}, Ref{Cdouble}, Ref{Int32}),
                    data.handle, i + 1, count, status.checki, status.eps, done, st, err, checked))
        i += done[]
        status.i = i
        ls > 0 && i % ls == 0 && print_linesearch(data)
        gp > 0 && i % gp == 0 && print_gapp(data)
        if checked[] != 0
            report()
            status.status != :Continue && break
        else
            status.checked = false
        end
    end
    guess = Vector{Float64}(undef, length(x))
    force = status.checked ? Int32(0) : Int32(1)                                   # solverwrapper.jl:31-34
    check(ccall((:fos_feas_getsol, libfoship), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int32, Cdouble, Ref{Int32}, Ref{Cdouble}),
                data.handle, guess, force, status.eps, st, err))
    force != 0 && report()
    if status.verbose > 0
        println("Time for iterations: ")
        println("$(time() - t1) s")
    end
    return guess
end

function __init__()
    PROX_TRAMPOLINE[] = @cfunction(prox_trampoline, Int32, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}))
end

end # module
