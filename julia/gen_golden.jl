# gen_golden.jl -- pins libbohip's oracle against the REAL reference.
#
# Run on any machine with Julia and the reference's dependencies (none is needed from this repository but the data file):
#
#     julia --project=/path/to/BayesianOptimization.jl julia/gen_golden.jl tests/golden/julia_inputs.txt tests/golden/julia_outputs.txt
#
# Reads the inputs of the committed golden cases (tests/golden/julia_inputs.txt, written by
# tests/golden/export_inputs_for_julia.py: the six golden cases + a C2-shaped sample at N = 3000), pushes every case
# through GaussianProcesses.ElasticGPE exactly as the reference's loop does (append!, src/models/gp.jl:11) and through the
# reference's own functors (src/acquisitionfunctions.jl), and writes what the oracle restates: the posterior mean and
# variance of predict_f, alpha, the Cholesky factor's diagonal and last row, the noise actually added to the diagonal
# (settles the `+eps()` question), the marginal likelihood, every acquisition's scores and the arg-max under the
# reference's rule (strict `>`, first maximum wins, src/acquisition.jl:62).
# tests/test_julia_golden.py consumes the output file when present: oracle-vs-Julia on the CPU, device-vs-Julia on the GPU.
# Until someone runs this, DESIGN.md and the oracle header say PARITY UNPINNED.
#
# The file format (bohip-golden-v1, tests/golden/gio.py) is line-oriented text so that no JSON/NPZ package is required.
# NOT EXECUTED in this repository (no Julia toolchain in the build image).
using GaussianProcesses, LinearAlgebra
import BayesianOptimization
const BO = BayesianOptimization

# ---- bohip-golden-v1 reader / writer ------------------------------------------------------------------------------------
function read_cases(path)
    cases = Vector{Pair{String, Dict{String, Any}}}()
    open(path) do f
        strip(readline(f)) == "bohip-golden-v1" || error("not a bohip-golden-v1 file")
        cur = nothing
        while !eof(f)
            t = split(readline(f))
            isempty(t) && continue
            if t[1] == "case"
                cur = Dict{String, Any}()
                push!(cases, String(t[2]) => cur)
            elseif t[1] == "str"
                cur[String(t[2])] = join(t[3:end], " ")
            elseif t[1] == "array"
                nd = parse(Int, t[3])
                shape = [parse(Int, s) for s in t[4:3+nd]]
                vals = [parse(Float64, s) for s in split(readline(f))]
                # the file is ROW-major: a (rows, cols) array becomes a cols x rows Julia matrix by reshape, i.e. an
                # N x d block of observations arrives directly as the d x N matrix GaussianProcesses wants
                cur[String(t[2])] = nd == 0 ? vals[1] : nd == 1 ? vals : reshape(vals, reverse(shape)...)
            end
        end
    end
    cases
end
fmt(x::Float64) = isnan(x) ? "nan" : isinf(x) ? (x > 0 ? "inf" : "-inf") : repr(x)
function write_array(io, key, a::AbstractVector)
    println(io, "array $key 1 $(length(a))")
    println(io, join((fmt(Float64(v)) for v in a), " "))
end
function write_scalar(io, key, v)
    println(io, "array $key 0")
    println(io, fmt(Float64(v)))
end

# ---- one case -------------------------------------------------------------------------------------------------------------
function make_kernel(kern, ll, ls)
    kern == "SEArd" ? SEArd(collect(ll), ls) : kern == "SEIso" ? SEIso(ll[1], ls) : kern == "Mat52Ard" ? Mat52Ard(collect(ll), ls) :
    error("unknown kernel $kern")
end
function make_acq(name, p)
    name == "EI" ? BO.ExpectedImprovement(p[1]) :
    name == "PI" ? BO.ProbabilityOfImprovement(p[1]) :
    name == "UCB" ? BO.UpperConfidenceBound(BO.NoBetaScaling(), p[1]) :
    name == "MI" ? BO.MutualInformation(p[1], p[2]) :            # fields (sqrtα, γ̂): the golden params are exactly those
    name == "MaxMean" ? BO.MaxMean() : error("unknown acquisition $name")
end
function argmax_first(f)                                        # src/acquisition.jl:55-66: maxf = -Inf; f > maxf
    maxf = -Inf; idx = -1
    for (j, v) in enumerate(f)
        if v > maxf
            maxf = v; idx = j - 1                                # 0-based like include/bohip.h
        end
    end
    maxf, idx
end

function run_case(io, name, c)
    X = c["X"]; y = c["y"]; Xs = c["Xs"]                        # d x N, N, d x R (see read_cases)
    d = size(X, 1)
    ll = c["loglen"] isa Number ? [c["loglen"]] : c["loglen"]
    mean = c["mean"] == "MeanZero" ? MeanZero() : MeanConst(c["beta"])
    gp = ElasticGPE(d; mean = mean, kernel = make_kernel(c["kern"], ll, c["logsig"]), logNoise = c["lognoise"],
                    capacity = max(size(X, 2), 1))
    append!(gp, X, y)                                           # update! (src/models/gp.jl:11), what initialise_model! does
    μ, σ² = predict_f(gp, Xs)
    cK = Matrix(gp.cK)                                          # covariance incl. noise as the package holds it
    U = cholesky(Symmetric(cK)).U                               # L' ; the package's own factor agrees to rounding
    σf² = exp(2 * c["logsig"])
    println(io, "case $name")
    println(io, "str generator GaussianProcesses.jl $(pkgversion(GaussianProcesses)) BayesianOptimization.jl $(pkgversion(BO)) julia $(VERSION)")
    write_array(io, "mu", μ)
    write_array(io, "var", σ²)
    write_array(io, "alpha", gp.alpha)
    write_array(io, "Ldiag", diag(U))
    write_array(io, "Lrow_last", U[:, end])
    write_scalar(io, "mll", gp.mll)
    write_scalar(io, "noise_on_diagonal", cK[1, 1] - σf²)       # exp(2 logNoise) [+ eps()]: the first UPSTREAM-UNVERIFIED switch
    write_scalar(io, "var_min", minimum(σ²))                    # < 0 would mean predict_f does not clamp (second switch)
    for key in sort(collect(keys(c)))
        endswith(key, "_params") || continue
        acq = String(split(key, "_")[1])
        p = c[key] isa Number ? [c[key]] : collect(c[key])
        a = make_acq(acq, p)
        f = BO.acquisitionfunction(a, gp)(Xs)                   # batched form, src/acquisitionfunctions.jl:4-9
        f1 = [BO.acquisitionfunction(a, gp)(Xs[:, j]) for j in 1:min(size(Xs, 2), 4)]   # single-column form (test/acquisitionfunctions.jl:8-11)
        all(f[1:length(f1)] .== f1) || @warn("batched != single in case $name, $acq")
        maxf, idx = argmax_first(f)
        write_array(io, "$(acq)_score", f)
        write_array(io, "$(acq)_best", [maxf])
        write_array(io, "$(acq)_best_idx", [Float64(idx)])
    end
    println(io, "end")
end

function main(args)
    length(args) == 2 || error("usage: julia gen_golden.jl <julia_inputs.txt> <julia_outputs.txt>")
    cases = read_cases(args[1])
    open(args[2], "w") do io
        println(io, "bohip-golden-v1")
        for (name, c) in cases
            @info "case $name: N = $(size(c["X"], 2)), d = $(size(c["X"], 1)), R = $(size(c["Xs"], 2))"
            run_case(io, name, c)
        end
    end
    @info "wrote $(args[2]); commit it next to julia_inputs.txt and run pytest: tests/test_julia_golden.py picks it up"
end

main(ARGS)
