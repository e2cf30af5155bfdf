"""The Julia binding (julia/BOHip.jl) cannot be executed here (no Julia toolchain), so its contact surface with the
library is checked mechanically, three ways: C prototypes in include/bohip.h  <->  ctypes signatures in
bayesianoptimization.jl_amd/_lib.py (exercised on the GPU)  <->  the `ccall` tuples in julia/BOHip.jl."""
import ctypes as C
import os
import re

from conftest import ROOT

JL = os.path.join(ROOT, "julia", "BOHip.jl")
HDR = os.path.join(ROOT, "include", "bohip.h")

JL2CANON = {"Cint": "int", "Int64": "int64", "UInt64": "uint64", "Float64": "double", "Cdouble": "double",
            "Cstring": "cstr", "Cvoid": "void", "Ptr{Float64}": "ptr(double)", "Ptr{Int64}": "ptr(int64)",
            "Ptr{Cint}": "ptr(int)", "Ptr{Cvoid}": "ptr(void)", "Ptr{Ptr{Cvoid}}": "ptr(ptr)", "Ptr{Best}": "ptr(best)",
            "Ptr{Cstring}": "ptr(cstr)"}


def canon_ctypes(t):
    from bohip import _lib

    if t is None:
        return "void"
    table = {C.c_int: "int", C.c_int64: "int64", C.c_uint64: "uint64", C.c_double: "double", C.c_char_p: "cstr",
             C.c_void_p: "ptr(void)"}
    if t in table:
        return table[t]
    if t is C.POINTER(C.c_double):
        return "ptr(double)"
    if t is C.POINTER(C.c_int64):
        return "ptr(int64)"
    if t is C.POINTER(C.c_int):
        return "ptr(int)"
    if t is C.POINTER(C.c_void_p):
        return "ptr(ptr)"
    if t is C.POINTER(_lib.Best):
        return "ptr(best)"
    if t is C.POINTER(C.c_char_p):
        return "ptr(cstr)"
    raise AssertionError(f"unmapped ctypes type {t}")


def canon_c(t):
    t = re.sub(r"\bconst\b", "", t).replace(" ", "")
    table = {"int": "int", "int64_t": "int64", "uint64_t": "uint64", "double": "double", "void": "void",
             "double*": "ptr(double)", "int64_t*": "ptr(int64)", "int*": "ptr(int)", "void*": "ptr(void)",
             "bohip_gp*": "ptr(void)", "bohip_mgp*": "ptr(void)", "bohip_gp**": "ptr(ptr)", "bohip_mgp**": "ptr(ptr)",
             "bohip_direct*": "ptr(void)", "bohip_direct**": "ptr(ptr)",
             "bohip_best*": "ptr(best)", "char*": "cstr", "char**": "ptr(cstr)"}
    assert t in table, f"unmapped C type {t!r}"
    return table[t]


def compatible(a, b):
    """equal, or an untyped pointer (device addresses, opaque handles) against any data pointer"""
    return a == b or ("ptr(void)" in (a, b) and a.startswith("ptr") and b.startswith("ptr"))


def julia_ccalls():
    src = open(JL).read()
    out = {}
    for m in re.finditer(r"ccall\(\(:(\w+), libbohip\),\s*([\w{}]+),\s*\(([^()]*)\)", src):
        name, ret, args = m.group(1), m.group(2), m.group(3)
        types = [a.strip() for a in args.split(",") if a.strip()]
        assert name not in out, f"{name} bound twice in BOHip.jl"
        out[name] = (JL2CANON[ret], [JL2CANON[t] for t in types])
    return out


def header_prototypes():
    txt = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    txt = re.sub(r"#.*", "", txt)
    out = {}
    for m in re.finditer(r"([\w \*]+?)\b(bohip_\w+)\s*\(([^()]*)\)\s*;", txt):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)(\w+)$", a)          # strip the parameter name
                types.append(canon_c(mm.group(1)))
        out[name] = (canon_c(ret), types)
    return out


def test_header_ctypes_and_julia_agree_on_every_symbol():
    from bohip import _lib

    jl, hd = julia_ccalls(), header_prototypes()
    ct = {k: (canon_ctypes(r), [canon_ctypes(a) for a in args]) for k, (r, args) in _lib.SIGNATURES.items()}
    assert set(hd) == set(ct), set(hd) ^ set(ct)
    assert set(jl) == set(ct), ("BOHip.jl must bind exactly the header's symbols", set(jl) ^ set(ct))
    for name in sorted(ct):
        for other, label in ((hd, "include/bohip.h"), (jl, "julia/BOHip.jl")):
            r0, a0 = ct[name]
            r1, a1 = other[name]
            assert compatible(r0, r1), (name, label, "return", r0, r1)
            assert len(a0) == len(a1), (name, label, "argument count", a0, a1)
            for i, (x, y) in enumerate(zip(a0, a1)):
                assert compatible(x, y), (name, label, f"argument {i}", x, y)
        # Julia against the header directly, with NO untyped-pointer slack except where the header itself says void* / handle
        for i, (x, y) in enumerate(zip(hd[name][1], jl[name][1])):
            assert compatible(x, y), (name, "header vs Julia", i, x, y)


def test_julia_module_owns_the_loop_for_device_models():
    """the reference's BOpt hard-types opt::NLopt.Opt, so the binding must own BOpt/boptimize! for its
    model types, and every generic function the loop touches (src/models/gp.jl:2-18,42-47) needs a device method."""
    src = open(JL).read()
    for needle in ("using LinearAlgebra", "function BOpt(func, model::AbstractBOHipModel",
                   "function boptimize!(o::DeviceBOpt)", "acquire_max_device(o.acquisition, o.model",
                   "function mean_var(m::AbstractBOHipModel, X::AbstractMatrix)",
                   "function mean_var(m::AbstractBOHipModel, x::AbstractVector)",
                   "function myrand(m::AbstractBOHipModel, X::AbstractMatrix)", "dims(m::AbstractBOHipModel)",
                   "maxy(m::AbstractBOHipModel)", "function update!(m::AbstractBOHipModel, x, y)",
                   "function optimizemodel!(o::MAPGPOptimizer, model::AbstractBOHipModel)",
                   "function defaultoptions(::Type{<:AbstractBOHipModel}, ::Type{ThompsonSamplingSimple})",
                   "function show(io::IO, ::MIME\"text/plain\", m::AbstractBOHipModel)", "BO.initialise_model!(o)",
                   "BO._evaluate_function(o, x)"):
        assert needle in src, needle
    assert "opt::NLopt.Opt" not in re.sub(r"#.*", "", src)                 # no NLopt.Opt field, no ForwardDiff in the search
    assert "ForwardDiff" not in re.sub(r"#.*", "", src)
    # balanced block structure (a cheap syntax smoke test in the absence of a Julia parser)
    code = re.sub(r'"""(.|\n)*?"""', '""', src)
    code = re.sub(r'"(?:[^"\\\n]|\\.)*"', '""', re.sub(r"#.*", "", code))
    code = re.sub(r"\[[^\[\]]*\bfor\b[^\[\]]*\]", "[]", code)         # comprehensions open no block
    opens = len(re.findall(r"\b(function|if|for|while|begin|struct|module|let|do|try|abstract type)\b", code))
    ends = len(re.findall(r"\bend\b", code))
    assert opens == ends, (opens, ends)
    for a, b in ("()", "[]", "{}"):
        assert code.count(a) == code.count(b), (a, code.count(a), code.count(b))


def test_both_hosts_take_the_same_route_for_direct_methods():
    """:GN_DIRECT* (the reference's default for ThompsonSamplingSimple, src/acquisition.jl:7-9) runs the SAME batched dividing-rectangles
    search on both hosts: the Python mirror (acquisition._batched_direct_l) and the reference-side binding (julia/BOHip.jl).  Julia cannot
    run here, so the rules are compared as text: every rule statement of one has its counterpart in the other, both dispatch on "DIRECT"
    in the method name in the generic AND the ThompsonSamplingSimple method, and both honour maxtime / stopval."""
    jl = open(JL).read()
    py = open(os.path.join(ROOT, "bayesianoptimization.jl_amd", "acquisition.py")).read()
    jfun = jl[jl.index("function _batched_direct_l("):jl.index("function acquire_max_device(::ThompsonSamplingSimple")]
    pfun = py[py.index("def _batched_direct_l("):py.index("# NLopt.Opt properties the reference forwards")]
    pairs = [
        ("np.full((d, 1), 0.5)", "fill(0.5, d, 1)"),                                             # start: the centre of the unit cube
        ("size = Lv.min(axis=0)", "minimum(Lv, dims = 1)"),                                      # a rectangle's size = its longest side
        ("3.0 ** (-k)", "3.0^(-k)"),                                                             # hull abscissa
        ("(y2 - y1) * (pt[0] - x1) <= (pt[1] - y1) * (x2 - x1)", "(y2 - y1) * (pt[1] - x1) <= (pt[2] - y1) * (x2 - x1)"),   # upper-hull test
        ("longest if longest.size == d else longest[:1]", "length(longest) == d ? longest : longest[1:1]"),   # cube: all sides; else the first longest
        ("3.0 ** (-(kmin + 1))", "3.0^(-(kmin + 1))"),                                           # trisection step
        ("(maxeval - evals - len(cols)) // 2", "(maxeval - evals - length(cols)) ÷ 2"),          # evaluation budget
        ("np.argsort(-w, kind=\"stable\")", "sortperm(-w, alg = MergeSort)"),                    # best sampled value first, stable
        ("F.max() >= stopval", "maximum(F) >= stopval"),
        ("time.monotonic() < deadline", "time() < deadline"),
        ("np.where(np.isnan(F), -math.inf, F)", "isnan(x) ? -Inf : x"),                          # NaN never wins
    ]
    for a, b in pairs:
        assert a in pfun, ("python", a)
        assert b in jfun, ("julia", b)
    assert jl.count('occursin("DIRECT", uppercase(string(options.method)))') == 2           # generic + ThompsonSamplingSimple
    assert jl.count("_batched_direct_l(f_batch, lb, ub, max(1, options.maxeval)") == 2
    assert py.count("direct_l_search(f_batch, lb, ub, max(1, maxeval), sval, maxtime)") == 1 and '"DIRECT" in method.upper()' in py
    # a single-device model takes the search as ONE library call on both hosts (bohip_gp_direct_max; csrc/direct_l.h is the same
    # bookkeeping, tests/test_direct_l.py holds it to the NumPy twin bit for bit); acq 5 = one posterior draw per point
    assert "model.direct_max(" in py and ":bohip_gp_direct_max" in jl and jl.count("_direct_max_device(m, ") == 3   # the definition + two call sites
    assert "const ACQ_THOMPSON_DRAW = Cint(5)" in jl and "#define BOHIP_ACQ_THOMPSON_DRAW 5" in open(os.path.join(ROOT, "include", "bohip.h")).read()
    # one posterior draw per new point in both (mu + sigma z, src/models/gp.jl:6)
    assert "mu .+ sqrt.(max.(var, 0.0)) .* randn(length(mu))" in jl
    assert "np.sqrt(np.maximum(np.asarray(var), 0.0)) * gen.standard_normal(np.size(mu))" in py
