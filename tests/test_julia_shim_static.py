"""Static cross-check of the Julia `ccall` shim (firstordersolvers.jl_amd/julia/FOSHip.jl) against include/foship.h: there is no
Julia in the build image, so the shim has never run -- this test at least keeps every ccall's symbol, return type, argument
types and argument count in step with the header (which grew from 40 to 81 entries over the rounds), and the CheckResult
struct in step with fos_check_result."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
JL = (ROOT / "firstordersolvers.jl_amd" / "julia" / "FOSHip.jl").read_text()
HDR = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "foship.h").read_text(), flags=re.S)

# C parameter type -> the Julia types a ccall may declare for it
C2JL = {
    "int": {"Cint"}, "int32_t": {"Int32", "Cint"}, "int64_t": {"Int64"}, "double": {"Cdouble", "Float64"},
    "fos_handle": {"Ptr{Cvoid}"}, "fos_handle*": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "fos_feas_handle": {"Ptr{Cvoid}"}, "fos_feas_handle*": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "int*": {"Ref{Cint}", "Ptr{Cint}"}, "int32_t*": {"Ref{Int32}", "Ptr{Int32}", "Ref{Cint}", "Ptr{Cint}"},
    "int64_t*": {"Ref{Int64}", "Ptr{Int64}"}, "double*": {"Ref{Cdouble}", "Ptr{Cdouble}", "Ptr{Float64}"},
    "void*": {"Ptr{Cvoid}"}, "char*": {"Ptr{UInt8}", "Cstring"},
    "fos_check_result*": {"Ref{CheckResult}", "Ptr{CheckResult}"}, "fos_allreduce_fn": {"Ptr{Cvoid}"}, "fos_prox_fn": {"Ptr{Cvoid}"},
}


def header_prototypes():
    protos = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(fos_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", HDR, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if params and params != "void":
            for prm in params.split(","):
                prm = re.sub(r"\bconst\b", "", prm).strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", prm)          # type then parameter name
                t = re.sub(r"\s+", "", mm.group(1))
                types.append(t)
        protos[name] = ("Cint" if ret == "int" else "Cstring", types)
    return protos


def split_top(s):
    """split at top-level commas"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls():
    calls = []
    for m in re.finditer(r"ccall\(\(:(fos_[a-z0-9_]+),\s*libfoship\),", JL):
        i = m.end()
        depth, j = 1, i                     # find the ccall's closing parenthesis
        while depth:
            depth += {"(": 1, ")": -1}.get(JL[j], 0)
            j += 1
        parts = split_top(re.sub(r"#[^\n]*", "", JL[i:j - 1]))
        ret, argt = parts[0], parts[1]
        assert argt.startswith("(") and argt.endswith(")"), (m.group(1), argt)
        types = split_top(argt[1:-1].rstrip(","))
        calls.append((m.group(1), ret, types, parts[2:]))
    return calls


def test_every_ccall_matches_the_header():
    protos = header_prototypes()
    assert len(protos) >= 55 and "fos_step" in protos and "fos_set_cg_variant" in protos
    calls = julia_ccalls()
    assert len(calls) >= 15
    for name, ret, types, args in calls:
        assert name in protos, "FOSHip.jl calls %s, which include/foship.h does not declare" % name
        cret, ctypes_ = protos[name]
        assert ret == cret, (name, ret, cret)
        assert len(types) == len(ctypes_), "%s: %d argument types in the ccall, %d parameters in the header" % (name, len(types), len(ctypes_))
        for k, (jt, ct) in enumerate(zip(types, ctypes_)):
            assert jt in C2JL[ct], "%s argument %d: Julia %s for C %s" % (name, k + 1, jt, ct)
        splat = any(a.endswith("...") for a in args)
        if not splat:
            assert len(args) == len(types), "%s: %d values passed for %d declared types" % (name, len(args), len(types))
    # the calls the drop-in path cannot work without
    used = {c[0] for c in calls}
    for need in ("fos_create", "fos_destroy", "fos_set_alg", "fos_set_iterate", "fos_step", "fos_getsol", "fos_last_error",
                 "fos_feas_create", "fos_feas_destroy", "fos_feas_set_affine", "fos_feas_set_box", "fos_feas_set_cones", "fos_feas_set_alg", "fos_feas_set_iterate",
                 "fos_feas_step", "fos_feas_getsol"):
        assert need in used, need


def test_checkresult_struct_matches_fos_check_result():
    cs = re.search(r"typedef struct fos_check_result \{(.*?)\} fos_check_result;", HDR, flags=re.S).group(1)
    cfields = [(re.sub(r"\s+", "", t), n) for t, n in re.findall(r"(double|int64_t|int32_t)\s+([a-z_]+)\s*;", cs)]
    js = re.search(r"struct CheckResult[^\n]*\n(.*?)\nend", JL, flags=re.S).group(1)
    jfields = [(n, t) for n, t in re.findall(r"([a-z_]+)::([A-Za-z0-9]+)", js)]
    assert len(cfields) == len(jfields) == 14
    tmap = {"double": "Cdouble", "int64_t": "Int64", "int32_t": "Int32"}
    for (ct, cn), (jn, jt) in zip(cfields, jfields):
        assert cn == jn and tmap[ct] == jt, (cn, ct, jn, jt)


def test_shim_constants_match_the_header():
    m = re.search(r"const\s+(FOS_ALG_[A-Z_, ]+?)\s*=\s*([^\n]+)", JL)
    names = [x.strip() for x in m.group(1).split(",")]
    values = [int(v) for v in re.findall(r"\((-?\d+)\)", m.group(2))]
    assert names == ["FOS_ALG_GAP", "FOS_ALG_GAPA", "FOS_ALG_FISTA", "FOS_ALG_DYKSTRA"] and len(values) == 4
    for name, jv in zip(names, values):
        assert int(re.search(r"#define\s+%s\s+(-?\d+)" % name, HDR).group(1)) == jv, name
    syms = re.search(r"const STATUS_SYMBOLS = \(([^)]*)\)", JL).group(1).replace(":", "").replace(" ", "").split(",")
    for code, sym in enumerate(syms):
        assert int(re.search(r"#define\s+FOS_STATUS_%s\s+(\d+)" % sym.upper(), HDR).group(1)) == code
