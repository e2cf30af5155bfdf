# BOHip.jl -- the reference-side binding: run jbrea/BayesianOptimization.jl's GP posterior + acquisition search on an
# MI355X through libbohip.so (include/bohip.h).
#
#     using BayesianOptimization, BOHip
#     model = BOHipGPE(2; mean = 0.0, kernel = :SEArd, loglen = [0., 0.], logsig = 5., logNoise = 0., capacity = 3000)
#     opt   = BOpt(f, model, UpperConfidenceBound(), MAPGPOptimizer(every = 50, ...), [-5., 0.], [10., 15.]; ...)
#     boptimize!(opt)                      # README.md:20-48 unchanged, except for the model constructor
#
# How it plugs in (SURVEY.md 8-B1).  The reference's seam is dispatch on the model type, but two things in it are
# hard-wired to NLopt + ForwardDiff and cannot be reached by dispatch alone: `BOpt.opt::NLopt.Opt`
# (src/BayesianOptimization.jl:74, built by nlopt_setup :134) and the per-candidate `Dual` evaluation it drives
# (src/acquisition.jl:11-17,59).  So this module OWNS the optimisation object for its model types:
#   * `BOpt(func, model::AbstractBOHipModel, ...)` is a more specific method of the reference's constructor (same
#     positional arguments, same keywords, same validation, src/BayesianOptimization.jl:89-136) returning a `DeviceBOpt`
#     -- the reference's struct without the `opt::NLopt.Opt` field;
#   * `boptimize!(o::DeviceBOpt)` follows src/BayesianOptimization.jl:176-207 line for line, with the one call
#     `acquire_max(o.opt, lb, ub, restarts)` (:185) replaced by the batched device search: ALL restarts advance in lock
#     step inside libbohip (`bohip_gp_acquire_max`), or are scored in one batch for derivative-free methods;
#   * the model-side generic functions of src/models/gp.jl:2-18,42-47 get methods for the device model.
# Everything else (acquisition types and their setparams!, counters, initialisers, timers, `_evaluate_function`,
# `initialise_model!`, verbosity) is the reference's own code, called as is.
#
# STATUS: there is no Julia toolchain in the build image or on the GPU box, so this file has NOT been executed.
# What is checked mechanically (tests/test_julia_binding.py): every `ccall` below names a symbol of include/bohip.h and its
# (return type, argument types) tuple equals the ctypes signature in bayesianoptimization.jl_amd/_lib.py, which IS
# exercised on the GPU; every symbol of the header is bound here.
module BOHip

import BayesianOptimization
const BO = BayesianOptimization
import BayesianOptimization: BOpt, boptimize!, mean_var, myrand, dims, maxy, update!, optimizemodel!, defaultoptions,
                             acquire_max, acquire_model_max, isdone, maxduration!, maxiterations!, setparams!,
                             AbstractAcquisition, ExpectedImprovement, ProbabilityOfImprovement, UpperConfidenceBound,
                             MutualInformation, MaxMean, ThompsonSamplingSimple, MAPGPOptimizer, Max, Progress, Timings,
                             ScaledSobolIterator, ScaledLHSIterator, IterationCounter, DurationCounter
import NLopt
using LinearAlgebra
using Dates: now
using TimerOutputs: TimerOutput, reset_timer!, @timeit
import Base: show

export BOHipGPE, BOHipMultiGPE, DeviceBOpt

const libbohip = get(ENV, "BOHIP_LIB", "libbohip.so")

struct Best                     # bohip_best: the 16-byte arg-max record
    val::Float64
    idx::Int64                  # 0-based column, -1 = none
end

# =====================================================================================================================
# Raw bindings: ONE `ccall` per symbol of include/bohip.h, argument types in header order.
# =====================================================================================================================
c_last_error() = unsafe_string(ccall((:bohip_last_error, libbohip), Cstring, ()))
c_version() = unsafe_string(ccall((:bohip_version, libbohip), Cstring, ()))
c_device_count() = ccall((:bohip_device_count, libbohip), Cint, ())
c_gp_create(d, cap, kern, dev, out) = ccall((:bohip_gp_create, libbohip), Cint, (Int64, Int64, Cint, Cint, Ptr{Ptr{Cvoid}}), d, cap, kern, dev, out)
c_gp_destroy(h) = ccall((:bohip_gp_destroy, libbohip), Cvoid, (Ptr{Cvoid},), h)
c_gp_set_hyper(h, ll, ls, ln, b) = ccall((:bohip_gp_set_hyper, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64, Float64, Float64), h, ll, ls, ln, b)
c_gp_append(h, X, y, p) = ccall((:bohip_gp_append, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), h, X, y, p)
c_gp_refit(h) = ccall((:bohip_gp_refit, libbohip), Cint, (Ptr{Cvoid},), h)
c_gp_dims(h, d, n) = ccall((:bohip_gp_dims, libbohip), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), h, d, n)
c_gp_maxy(h, out) = ccall((:bohip_gp_maxy, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}), h, out)
c_gp_get_xy(h, X, y) = ccall((:bohip_gp_get_xy, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h, X, y)
c_gp_mll(h, out) = ccall((:bohip_gp_mll, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}), h, out)
c_gp_mll_grad(h, mll, dn, dm, dk) = ccall((:bohip_gp_mll_grad, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, mll, dn, dm, dk)
c_gp_predict(h, Xs, R, mu, var) = ccall((:bohip_gp_predict, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}), h, Xs, R, mu, var)
c_gp_predict_cov(h, Xs, R, mu, cov) = ccall((:bohip_gp_predict_cov, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}), h, Xs, R, mu, cov)
c_gp_score(h, acq, p, Xs, R, sc, best) = ccall((:bohip_gp_score, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Best}), h, acq, p, Xs, R, sc, best)
c_gp_score_grad(h, acq, p, Xs, R, sc, g) = ccall((:bohip_gp_score_grad, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}), h, acq, p, Xs, R, sc, g)
c_gp_acquire_max(h, acq, p, lb, ub, st, R, maxeval, ftol, xtol, xo, fo, best, bx, ev) = ccall((:bohip_gp_acquire_max, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Best}, Ptr{Float64}, Ptr{Int64}), h, acq, p, lb, ub, st, R, maxeval, ftol, xtol, xo, fo, best, bx, ev)
c_direct_create(d, lb, ub, maxeval, stopval, maxtime, out) = ccall((:bohip_direct_create, libbohip), Cint, (Int64, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Float64, Ptr{Ptr{Cvoid}}), d, lb, ub, maxeval, stopval, maxtime, out)
c_direct_destroy(s) = ccall((:bohip_direct_destroy, libbohip), Cvoid, (Ptr{Cvoid},), s)
c_direct_ask(s, X, cap, n) = ccall((:bohip_direct_ask, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Int64}), s, X, cap, n)
c_direct_tell(s, f, n) = ccall((:bohip_direct_tell, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), s, f, n)
c_direct_best(s, bf, bx, ev, it) = ccall((:bohip_direct_best, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}), s, bf, bx, ev, it)
c_gp_direct_max(h, acq, p, lb, ub, maxeval, stopval, maxtime, seed, bf, bx, ev, calls) = ccall((:bohip_gp_direct_max, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Float64, UInt64, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}), h, acq, p, lb, ub, maxeval, stopval, maxtime, seed, bf, bx, ev, calls)
const ACQ_THOMPSON_DRAW = Cint(5)   # BOHIP_ACQ_THOMPSON_DRAW: x -> myrand(model, x), bohip_gp_direct_max only

# :GN_DIRECT_L against ONE device handle in one library call (csrc/direct_l.h does the dividing-rectangles bookkeeping, every
# iteration's points are one scoring call); the multi-device model keeps the Julia loop below.
function _direct_max_device(m, acq::Cint, p::Vector{Float64}, lb::Vector{Float64}, ub::Vector{Float64}, options; seed::UInt64 = UInt64(0))
    bf = Ref(-Inf); bx = similar(lb); ev = Ref(Int64(0)); calls = Ref(Int64(0))
    check(c_gp_direct_max(m.handle, acq, p, lb, ub, max(1, options.maxeval), Float64(get(options, :stopval, Inf)),
                          Float64(get(options, :maxtime, 0.0)), seed, bf, bx, ev, calls))
    bf[], bx, ev[]
end
c_gp_set_maxtime(h, s) = ccall((:bohip_gp_set_maxtime, libbohip), Cint, (Ptr{Cvoid}, Float64), h, s)
c_gp_set_ascent_stop(h, fa, xr, sv) = ccall((:bohip_gp_set_ascent_stop, libbohip), Cint, (Ptr{Cvoid}, Float64, Float64, Float64), h, fa, xr, sv)
c_debug_set_chol_inv_g(blocks) = ccall((:bohip_debug_set_chol_inv_g, libbohip), Cint, (Cint,), blocks)
c_gp_set_jitter(h, rel, tries) = ccall((:bohip_gp_set_jitter, libbohip), Cint, (Ptr{Cvoid}, Float64, Cint), h, rel, tries)
c_gp_thompson(h, Xs, R, S, seed, j0, best) = ccall((:bohip_gp_thompson, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, UInt64, Int64, Ptr{Best}), h, Xs, R, S, seed, j0, best)
c_thompson_normal(seed, s, j) = ccall((:bohip_thompson_normal, libbohip), Float64, (UInt64, Int64, Int64), seed, s, j)
c_gp_score_dev(h, acq, p, dXs, R, dsc, dbest) = ccall((:bohip_gp_score_dev, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}), h, acq, p, dXs, R, dsc, dbest)
c_gp_predict_dev(h, dXs, R, dmu, dvar) = ccall((:bohip_gp_predict_dev, libbohip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}), h, dXs, R, dmu, dvar)
c_gp_set_stream(h, st) = ccall((:bohip_gp_set_stream, libbohip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h, st)
c_gp_synchronize(h) = ccall((:bohip_gp_synchronize, libbohip), Cint, (Ptr{Cvoid},), h)
c_gp_get_factor(h, L) = ccall((:bohip_gp_get_factor, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}), h, L)
c_gp_get_alpha(h, a) = ccall((:bohip_gp_get_alpha, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}), h, a)
c_gp_info(h, what, out) = ccall((:bohip_gp_info, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}), h, what, out)
c_gp_set_batch_hint(h, total) = ccall((:bohip_gp_set_batch_hint, libbohip), Cint, (Ptr{Cvoid}, Int64), h, total)
c_gp_enable_timing(h, on) = ccall((:bohip_gp_enable_timing, libbohip), Cint, (Ptr{Cvoid}, Cint), h, on)
c_gp_get_timing(h, names, ms, cap) = ccall((:bohip_gp_get_timing, libbohip), Cint, (Ptr{Cvoid}, Ptr{Cstring}, Ptr{Float64}, Cint), h, names, ms, cap)
# multi-GPU, one process and a device list (in-library RCCL)
c_mgp_create(d, cap, kern, devs, nd, spd, out) = ccall((:bohip_mgp_create, libbohip), Cint, (Int64, Int64, Cint, Ptr{Cint}, Cint, Cint, Ptr{Ptr{Cvoid}}), d, cap, kern, devs, nd, spd, out)
c_mgp_destroy(h) = ccall((:bohip_mgp_destroy, libbohip), Cvoid, (Ptr{Cvoid},), h)
c_mgp_set_hyper(h, ll, ls, ln, b) = ccall((:bohip_mgp_set_hyper, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64, Float64, Float64), h, ll, ls, ln, b)
c_mgp_append(h, X, y, p) = ccall((:bohip_mgp_append, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), h, X, y, p)
c_mgp_refit(h) = ccall((:bohip_mgp_refit, libbohip), Cint, (Ptr{Cvoid},), h)
c_mgp_score(h, acq, p, Xs, R, sc, best) = ccall((:bohip_mgp_score, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Best}), h, acq, p, Xs, R, sc, best)
c_mgp_set_candidates(h, Xs, R) = ccall((:bohip_mgp_set_candidates, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), h, Xs, R)
c_mgp_score_resident(h, acq, p, best) = ccall((:bohip_mgp_score_resident, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Best}), h, acq, p, best)
c_mgp_thompson(h, Xs, R, S, seed, best) = ccall((:bohip_mgp_thompson, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, UInt64, Ptr{Best}), h, Xs, R, S, seed, best)
c_mgp_acquire_max(h, acq, p, lb, ub, st, R, maxeval, ftol, xtol, xo, fo, best, bx, ev) = ccall((:bohip_mgp_acquire_max, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Best}, Ptr{Float64}, Ptr{Int64}), h, acq, p, lb, ub, st, R, maxeval, ftol, xtol, xo, fo, best, bx, ev)
c_mgp_set_maxtime(h, s) = ccall((:bohip_mgp_set_maxtime, libbohip), Cint, (Ptr{Cvoid}, Float64), h, s)
c_mgp_set_ascent_stop(h, fa, xr, sv) = ccall((:bohip_mgp_set_ascent_stop, libbohip), Cint, (Ptr{Cvoid}, Float64, Float64, Float64), h, fa, xr, sv)
c_mgp_set_jitter(h, rel, tries) = ccall((:bohip_mgp_set_jitter, libbohip), Cint, (Ptr{Cvoid}, Float64, Cint), h, rel, tries)
c_mgp_handle(h, i) = ccall((:bohip_mgp_handle, libbohip), Ptr{Cvoid}, (Ptr{Cvoid}, Cint), h, i)
c_mgp_info(h, what, out) = ccall((:bohip_mgp_info, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}), h, what, out)
# multi-GPU, one process per device (Distributed.jl / MPI.jl carry the 128-byte id)
c_comm_unique_id(id, n) = ccall((:bohip_comm_unique_id, libbohip), Cint, (Ptr{Cvoid}, Int64), id, n)
c_gp_comm_init(h, id, n, rank, nranks) = ccall((:bohip_gp_comm_init, libbohip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint), h, id, n, rank, nranks)
c_gp_comm_destroy(h) = ccall((:bohip_gp_comm_destroy, libbohip), Cint, (Ptr{Cvoid},), h)
c_gp_score_sharded_dev(h, acq, p, dXs, Rl, off, Rt, dsc, best) = ccall((:bohip_gp_score_sharded_dev, libbohip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}), h, acq, p, dXs, Rl, off, Rt, dsc, best)
c_gp_thompson_sharded(h, Xs, Rl, S, seed, off, Rt, best) = ccall((:bohip_gp_thompson_sharded, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, UInt64, Int64, Int64, Ptr{Best}), h, Xs, Rl, S, seed, off, Rt, best)

function check(rc::Integer)
    rc == 0 && return nothing
    msg = c_last_error()
    rc == -2 && throw(PosDefException(0))                   # BOHIP_E_NOTPD (pivot: bohip_gp_info(h, 0))
    error("libbohip error $rc: $msg")
end

# =====================================================================================================================
# Model types.  Field names follow what the reference READS: model.x (d x n), model.y
# (src/BayesianOptimization.jl:117-119, src/acquisitionfunctions.jl:136, test/warmstart.jl:26-27).
# =====================================================================================================================
const KERN = Dict(:SEArd => 0, :SEIso => 1, :Mat52Ard => 2)
abstract type AbstractBOHipModel end

"Device-resident elastic GP on ONE MI355X: drop-in for `ElasticGPE(d; mean, kernel, logNoise, capacity)` (README.md:22-27)."
mutable struct BOHipGPE <: AbstractBOHipModel
    handle::Ptr{Cvoid}
    dim::Int
    x::Matrix{Float64}
    y::Vector{Float64}
    kernel::Symbol
    meanconst::Bool             # MeanConst(beta) (a parameter) or MeanZero()
    mean::Float64
    loglen::Vector{Float64}     # d entries (SEIso: 1)
    logsig::Float64
    logNoise::Float64
end
"The same model replicated over a device list; candidates sharded, winners exchanged over RCCL inside libbohip."
mutable struct BOHipMultiGPE <: AbstractBOHipModel
    handle::Ptr{Cvoid}
    dim::Int
    x::Matrix{Float64}
    y::Vector{Float64}
    kernel::Symbol
    meanconst::Bool
    mean::Float64
    loglen::Vector{Float64}
    logsig::Float64
    logNoise::Float64
    devices::Vector{Cint}
end

function _hyper_args(kernel, d, mean, loglen)
    haskey(KERN, kernel) || throw(ArgumentError("kernel must be one of $(collect(keys(KERN)))"))
    ll = Float64.(collect(loglen))
    length(ll) == (kernel == :SEIso ? 1 : d) || throw(ArgumentError("loglen has the wrong length for $kernel"))
    mean === nothing ? (false, 0.0, ll) : (true, Float64(mean), ll)
end
"`BOHipGPE(d; mean = nothing (MeanZero) | beta (MeanConst), kernel = :SEArd | :SEIso | :Mat52Ard, loglen, logsig, logNoise, capacity, device)`"
function BOHipGPE(d::Integer; mean = nothing, kernel::Symbol = :SEArd, loglen = zeros(kernel == :SEIso ? 1 : d),
                  logsig = 0.0, logNoise = -2.0, capacity = 3000, device = 0)
    mc, beta, ll = _hyper_args(kernel, d, mean, loglen)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(c_gp_create(d, capacity, KERN[kernel], device, h))
    m = BOHipGPE(h[], d, zeros(d, 0), Float64[], kernel, mc, beta, ll, Float64(logsig), Float64(logNoise))
    finalizer(g -> (g.handle == C_NULL || c_gp_destroy(g.handle); g.handle = C_NULL), m)
    push_hyper!(m)
    m
end
function BOHipMultiGPE(d::Integer; devices = collect(0:c_device_count()-1), shards_per_device = 1, mean = nothing,
                       kernel::Symbol = :SEArd, loglen = zeros(kernel == :SEIso ? 1 : d), logsig = 0.0, logNoise = -2.0,
                       capacity = 3000)
    mc, beta, ll = _hyper_args(kernel, d, mean, loglen)
    devs = Cint.(collect(devices))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(c_mgp_create(d, capacity, KERN[kernel], devs, length(devs), shards_per_device, h))
    m = BOHipMultiGPE(h[], d, zeros(d, 0), Float64[], kernel, mc, beta, ll, Float64(logsig), Float64(logNoise), devs)
    finalizer(g -> (g.handle == C_NULL || c_mgp_destroy(g.handle); g.handle = C_NULL), m)
    push_hyper!(m)
    m
end
"GPE(x, y, mean, kernel, logNoise)-style construction from data (test/acquisitionfunctions.jl:4, test/acquisition.jl:2)."
function BOHipGPE(x::AbstractMatrix, y::AbstractVector; kwargs...)
    m = BOHipGPE(size(x, 1); capacity = max(size(x, 2), 1), kwargs...)
    update!(m, x, y)
end

_ll_full(m::AbstractBOHipModel) = m.kernel == :SEIso ? fill(m.loglen[1], m.dim) : m.loglen
push_hyper!(m::BOHipGPE) = check(c_gp_set_hyper(m.handle, _ll_full(m), m.logsig, m.logNoise, m.mean))
push_hyper!(m::BOHipMultiGPE) = check(c_mgp_set_hyper(m.handle, _ll_full(m), m.logsig, m.logNoise, m.mean))
"the replica whose scalar queries (predict, mll, ...) answer for the whole model"
gp_handle(m::BOHipGPE) = m.handle
gp_handle(m::BOHipMultiGPE) = c_mgp_handle(m.handle, 0)

function show(io::IO, ::MIME"text/plain", m::AbstractBOHipModel)
    println(io, "$(typeof(m)) [device-resident, libbohip: $(c_version())]")
    println(io, "  Dim = $(m.dim), observations = $(length(m.y))")
    println(io, "  Kernel: $(m.kernel), loglen = $(m.loglen), logsig = $(m.logsig)")
    println(io, "  Mean: $(m.meanconst ? "MeanConst($(m.mean))" : "MeanZero()"), logNoise = $(m.logNoise)")
    m isa BOHipMultiGPE && println(io, "  devices = $(Int.(m.devices))")
end

# ---- acquisition -> (acq_id, acq_params) of include/bohip.h ------------------------------------------------------------
acqid(::ExpectedImprovement) = Cint(0)
acqid(::ProbabilityOfImprovement) = Cint(1)
acqid(::UpperConfidenceBound) = Cint(2)
acqid(::MutualInformation) = Cint(3)
acqid(::MaxMean) = Cint(4)
acqparams(a::Union{ExpectedImprovement, ProbabilityOfImprovement}) = [a.τ, 0.0]
acqparams(a::UpperConfidenceBound) = [a.βt, 0.0]
acqparams(a::MutualInformation) = [a.sqrtα, a.γ̂]
acqparams(::MaxMean) = [0.0, 0.0]

_cols(m::AbstractBOHipModel, X::AbstractMatrix) =
    (size(X, 1) == m.dim || throw(DimensionMismatch("expected $(m.dim) rows (one point per column)")); Matrix{Float64}(X))

# =====================================================================================================================
# reference src/models/gp.jl:2-18 -- the generic functions through which the loop touches a model
# =====================================================================================================================
function mean_var(m::AbstractBOHipModel, X::AbstractMatrix)                               # :8
    Xc = _cols(m, X); R = size(Xc, 2)
    μ = Vector{Float64}(undef, R); σ² = Vector{Float64}(undef, R)
    check(c_gp_predict(gp_handle(m), Xc, R, μ, σ²))
    μ, σ²
end
function mean_var(m::AbstractBOHipModel, x::AbstractVector)                               # :2-5
    μ, σ² = mean_var(m, reshape(x, :, 1))
    μ[1], σ²[1]
end
function myrand(m::AbstractBOHipModel, x::AbstractVector)                                 # :6  one draw from N(mu, sigma^2)
    μ, σ² = mean_var(m, x)
    μ + sqrt(σ²) * randn()
end
function myrand(m::AbstractBOHipModel, X::AbstractMatrix)                                 # :7  ONE JOINT draw over the columns
    Xc = _cols(m, X); R = size(Xc, 2)
    μ = Vector{Float64}(undef, R); Σ = Matrix{Float64}(undef, R, R)
    check(c_gp_predict_cov(gp_handle(m), Xc, R, μ, Σ))
    jitter = 0.0; scale = max(maximum(diag(Σ)), floatmin(Float64))
    for _ in 1:40                                                                         # GaussianProcesses.jl make_posdef!-style escalation
        C = cholesky(Symmetric(Σ + jitter * I, :L); check = false)
        issuccess(C) && return μ + C.L * randn(R)
        jitter = max(10 * jitter, 1e-12 * scale)
    end
    throw(PosDefException(0))
end
dims(m::AbstractBOHipModel) = size(m.x)                                                   # :9
maxy(m::AbstractBOHipModel) = isempty(m.y) ? -Inf : maximum(m.y)                          # :10
_append(m::BOHipGPE, X, Y) = c_gp_append(m.handle, X, Y, length(Y))
_append(m::BOHipMultiGPE, X, Y) = c_mgp_append(m.handle, X, Y, length(Y))
function update!(m::AbstractBOHipModel, x, y)                                             # :11  append! (incremental factor extension)
    X = Matrix{Float64}(reshape(x, m.dim, :)); Y = Vector{Float64}(vec(collect(y)))
    size(X, 2) == length(Y) || throw(DimensionMismatch("x and y disagree on the number of observations"))
    rc = _append(m, X, Y)
    if rc == 0 || rc == -2                                                                # observations are stored even if the factorisation failed
        m.x = hcat(m.x, X); append!(m.y, Y)
    end
    check(rc)
    m
end
refit!(m::BOHipGPE) = (check(c_gp_refit(m.handle)); m)
refit!(m::BOHipMultiGPE) = (check(c_mgp_refit(m.handle)); m)
function mll(m::AbstractBOHipModel)
    out = Ref(0.0)
    check(c_gp_mll(gp_handle(m), out))
    out[]
end

function defaultoptions(::Type{<:AbstractBOHipModel}, ::Type{<:AbstractAcquisition})     # src/acquisition.jl:4-6
    (method = :LD_LBFGS, restarts = 10, maxeval = 2000)
end
function defaultoptions(::Type{<:AbstractBOHipModel}, ::Type{ThompsonSamplingSimple})    # :7-9
    (method = :GN_DIRECT_L, restarts = 1, maxeval = 2000)
end

# =====================================================================================================================
# acquisitionfunction(a, model)(X) + the arg-max (src/acquisitionfunctions.jl:4-9, src/acquisition.jl:54-68) on the device
# =====================================================================================================================
"scores of all columns, best value, best 1-based column (0 = nothing beat -Inf)"
function score(m::AbstractBOHipModel, a::AbstractAcquisition, X::AbstractMatrix; scores::Bool = true)
    Xc = _cols(m, X); R = size(Xc, 2)
    sc = scores ? Vector{Float64}(undef, R) : Float64[]
    best = Ref(Best(-Inf, -1))
    p = acqparams(a)
    rc = m isa BOHipMultiGPE ? c_mgp_score(m.handle, acqid(a), p, Xc, R, scores ? sc : C_NULL, best) :
                               c_gp_score(m.handle, acqid(a), p, Xc, R, scores ? sc : C_NULL, best)
    check(rc)
    sc, best[].val, Int(best[].idx) + 1
end
"value and gradient (d x R) of the acquisition at the columns of X: the role of wrap_gradient (src/acquisition.jl:11-17)"
function score_grad(m::AbstractBOHipModel, a::AbstractAcquisition, X::AbstractMatrix)
    Xc = _cols(m, X); R = size(Xc, 2)
    sc = Vector{Float64}(undef, R); g = Matrix{Float64}(undef, m.dim, R)
    check(c_gp_score_grad(gp_handle(m), acqid(a), acqparams(a), Xc, R, sc, g))
    sc, g
end

const _NLOPT_ONLY = (:initial_step, :population, :vector_storage, :local_optimizer)
function _check_options(options)
    for k in keys(options)
        k in (:method, :restarts, :maxeval, :maxtime, :ftol_rel, :xtol_abs, :ftol_abs, :xtol_rel, :stopval) && continue
        k in _NLOPT_ONLY ? @warn("acquisition option $k is an NLopt setting the device search does not implement; ignored") :
                           throw(ArgumentError("unknown acquisition option $k"))          # NLopt.Opt rejects unknown properties too
    end
end

"""
The search of `acquire_max(opt, lowerbounds, upperbounds, restarts)` (src/acquisition.jl:54-68) for a device model:
`restarts` Latin-hypercube starts (src/utils.jl:96-120), a local search from each, the best under strict `>`
(first maximum wins).  Does NOT call `setparams!` (the 4-argument method does not either).
"""
function acquire_max_device(a::AbstractAcquisition, m::AbstractBOHipModel, lowerbounds, upperbounds, options)
    _check_options(options)
    lb = Float64.(lowerbounds); ub = Float64.(upperbounds)
    maxf = -Inf; maxx = lb                                                                # :55-56
    (isempty(m.y) || options.restarts <= 0) && return maxf, maxx
    starts = BO.latin_hypercube_sampling(lb, ub, options.restarts)                        # :57 ScaledLHSIterator's matrix, d x restarts
    if string(options.method)[2] == 'D'                                                   # :31  gradient-based local search
        best = Ref(Best(-Inf, -1)); bx = Vector{Float64}(undef, m.dim); ev = Ref{Int64}(0)
        ftol = Float64(get(options, :ftol_rel, 1e-10)); xtol = Float64(get(options, :xtol_abs, 1e-10))
        m isa BOHipGPE && check(c_gp_set_maxtime(m.handle, Float64(get(options, :maxtime, 0.0))))
        m isa BOHipMultiGPE && check(c_mgp_set_maxtime(m.handle, Float64(get(options, :maxtime, 0.0))))
        fabs_ = Float64(get(options, :ftol_abs, 0.0)); xrel = Float64(get(options, :xtol_rel, 0.0)); sval = Float64(get(options, :stopval, Inf))
        m isa BOHipGPE && check(c_gp_set_ascent_stop(m.handle, fabs_, xrel, sval))        # NLopt's defaults = off (test/acquisition.jl:6,9 sets ftol_abs = eps())
        m isa BOHipMultiGPE && check(c_mgp_set_ascent_stop(m.handle, fabs_, xrel, sval))
        rc = m isa BOHipMultiGPE ?
             c_mgp_acquire_max(m.handle, acqid(a), acqparams(a), lb, ub, starts, size(starts, 2), options.maxeval, ftol, xtol,
                               C_NULL, C_NULL, best, bx, ev) :
             c_gp_acquire_max(m.handle, acqid(a), acqparams(a), lb, ub, starts, size(starts, 2), options.maxeval, ftol, xtol,
                              C_NULL, C_NULL, best, bx, ev)
        check(rc)
        return best[].idx < 0 ? (maxf, maxx) : (best[].val, bx)
    end
    if occursin("DIRECT", uppercase(string(options.method)))
        # :GN_DIRECT* -- dividing rectangles, every iteration's new points in ONE device call (same search as the Python mirror's
        # acquisition._batched_direct_l; tests/test_julia_binding.py keeps the two from drifting).  DIRECT ignores the start point and the
        # acquisition is deterministic: every restart would be the same run, so one run.
        if !(m isa BOHipMultiGPE)
            f, x, _ = _direct_max_device(m, Cint(acqid(a)), Float64.(acqparams(a)), lb, ub, options)
            return isfinite(f) ? (f, x) : (maxf, maxx)
        end
        f_batch = X -> score(m, a, X)[1]
        f, x, _ = _batched_direct_l(f_batch, lb, ub, max(1, options.maxeval); stopval = Float64(get(options, :stopval, Inf)),
                                    maxtime = Float64(get(options, :maxtime, 0.0)))
        return isfinite(f) ? (f, x) : (maxf, maxx)
    end
    # other derivative-free methods (:LN_*, non-DIRECT :GN_*): `maxeval` Latin-hypercube candidates per restart, ONE batch on the device
    _warn_not_a_local_search(options.method)
    n = clamp(options.maxeval * options.restarts, options.restarts, 1 << 20)
    cand = BO.latin_hypercube_sampling(lb, ub, n)
    _, f, j = score(m, a, cand; scores = false)
    j == 0 ? (maxf, maxx) : (f, cand[:, j])
end

const _WARNED_METHODS = Set{Symbol}()
# :LN_* and the non-DIRECT :GN_* methods have no device counterpart: said once per process and method, since what comes back is the
# best of a Latin-hypercube candidate set, not that algorithm's result.
function _warn_not_a_local_search(method)
    m = Symbol(method)
    m in _WARNED_METHODS && return
    push!(_WARNED_METHODS, m)
    @warn "acquire_max: method :$m is not implemented as such; maxeval Latin-hypercube candidates per restart are scored in one device batch instead (use :LD_LBFGS or :GN_DIRECT_L for a search)"
end

"""
DIviding RECTangles, locally biased (Gablonsky & Kelley 2001): the role of NLopt's `:GN_DIRECT_L`, the reference's default for
`ThompsonSamplingSimple` (src/acquisition.jl:7-9: restarts = 1, maxeval = 2000), for MAXIMISATION, with ALL the new points of one
iteration evaluated in ONE call `f_batch(X::Matrix) -> Vector` (columns are points).  NLopt's rules for this algorithm (cdirect.c,
`which_alg = 13`): a rectangle's size is its longest side; the potentially optimal set is the upper convex hull of (size, best value
of that size) from the incumbent's size up, one rectangle per size (Jones' epsilon = 0); a cube is trisected along every side, best
sampled value first; any other rectangle along its first longest side only.  Not reproduced: NLopt's evaluation order inside an
iteration.  Statement for statement the Python mirror's `_batched_direct_l`.  Returns (best value, best point, evaluations).
"""
function _batched_direct_l(f_batch, lb::Vector{Float64}, ub::Vector{Float64}, maxeval::Integer; stopval = Inf, maxtime = 0.0)
    d = length(lb); span = ub .- lb
    to_x(U) = lb .+ span .* U                                    # unit cube -> box, columns are points
    deadline = maxtime > 0 ? time() + maxtime : Inf              # NLopt maxtime (the reference's test passes it)
    C = fill(0.5, d, 1)                                          # centres (unit cube)
    Lv = zeros(Int, d, 1)                                        # level per side: side length 3^-level
    clean(v) = [isnan(x) ? -Inf : x for x in v]
    F = clean(vec(f_batch(to_x(C))))
    evals = 1
    while evals < maxeval && !(maximum(F) >= stopval) && time() < deadline
        size_ = vec(minimum(Lv, dims = 1))                       # key of the longest side (smaller = larger rectangle)
        best_of = Dict{Int, Int}()
        for k in sort(unique(size_))                             # one rectangle per size: the best, first on ties
            idx = findall(==(k), size_)
            best_of[k] = idx[argmax(F[idx])]
        end
        jmax = argmax(F)
        # upper hull over (diameter, f) from the incumbent's size towards the larger rectangles
        smax = size_[jmax]
        cand = sort(filter(k -> k <= smax, collect(keys(best_of))), rev = true)           # increasing diameter
        hull = Tuple{Float64, Float64, Int}[]
        for k in cand
            pt = (3.0^(-k), F[best_of[k]], best_of[k])
            while length(hull) >= 2
                (x1, y1, _) = hull[length(hull) - 1]; (x2, y2, _) = last(hull)
                if (y2 - y1) * (pt[1] - x1) <= (pt[2] - y1) * (x2 - x1)                   # the middle point is not above the chord
                    pop!(hull)
                else
                    break
                end
            end
            push!(hull, pt)
        end
        chosen = [j for (_, _, j) in hull]
        # new points of this iteration (capped by the evaluation budget)
        plan = Tuple{Int, Vector{Int}, Int}[]; cols = Vector{Float64}[]
        for j in chosen
            lv = Lv[:, j]; kmin = minimum(lv)
            longest = findall(==(kmin), lv)
            dims = length(longest) == d ? longest : longest[1:1]                          # a cube: every side; otherwise the first longest side
            if evals + length(cols) + 2 * length(dims) > maxeval
                dims = dims[1:max(0, (maxeval - evals - length(cols)) ÷ 2)]
            end
            isempty(dims) && continue
            delta = 3.0^(-(kmin + 1))
            start = length(cols)
            for i in dims, sgn in (+1.0, -1.0)
                c = C[:, j]; c[i] += sgn * delta
                push!(cols, c)
            end
            push!(plan, (j, dims, start))
        end
        isempty(cols) && break
        Unew = reduce(hcat, cols)
        Fnew = clean(vec(f_batch(to_x(Unew))))
        evals += length(Fnew)
        newL = Matrix{Int}(undef, d, length(Fnew))
        for (j, dims, start) in plan
            w = map(t -> max(Fnew[start + 2t - 1], Fnew[start + 2t]), 1:length(dims))
            order = sortperm(-w, alg = MergeSort)                                         # best sampled value first (stable)
            lv = Lv[:, j]
            for t in order
                lv[dims[t]] += 1                                 # the parent shrinks along i; the two children inherit the levels so far
                newL[:, start + 2t - 1] = lv
                newL[:, start + 2t] = lv
            end
            Lv[:, j] = lv
        end
        C = hcat(C, Unew); Lv = hcat(Lv, newL); F = vcat(F, Fnew)
    end
    jb = argmax(F)
    F[jb], to_x(C[:, jb:jb])[:, 1], evals
end

"""
    direct_l_search(f_batch, lb, ub, maxeval; stopval = Inf, maxtime = 0.0)

The same search as `_batched_direct_l` with the bookkeeping in libbohip (`bohip_direct_ask` / `_tell`, csrc/direct_l.h): `f_batch(X)`
scores the columns of X, one call per DIRECT iteration.  Returns (best value, best point, evaluations).
"""
function direct_l_search(f_batch, lb::Vector{Float64}, ub::Vector{Float64}, maxeval::Integer; stopval = Inf, maxtime = 0.0)
    d = length(lb)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(c_direct_create(d, lb, ub, max(1, maxeval), Float64(stopval), Float64(maxtime), h))
    X = Matrix{Float64}(undef, d, max(2d, 64))
    n = Ref(Int64(0))
    try
        while true
            check(c_direct_ask(h[], C_NULL, 0, n))               # size of this iteration's batch
            n[] == 0 && break
            n[] > size(X, 2) && (X = Matrix{Float64}(undef, d, 2 * n[]))
            check(c_direct_ask(h[], X, size(X, 2), n))
            F = Float64.(vec(f_batch(X[:, 1:n[]])))
            check(c_direct_tell(h[], F, n[]))
        end
        bf = Ref(-Inf); bx = Vector{Float64}(undef, d); ev = Ref(Int64(0)); it = Ref(Int64(0))
        check(c_direct_best(h[], bf, bx, ev, it))
        return bf[], bx, Int(ev[])
    finally
        c_direct_destroy(h[])
    end
end

function acquire_max_device(::ThompsonSamplingSimple, m::AbstractBOHipModel, lowerbounds, upperbounds, options)
    # acquisitionfunction(::ThompsonSamplingSimple, model) = x -> myrand(model, x) under a global derivative-free search
    # (src/acquisitionfunctions.jl:107-108, defaults :GN_DIRECT_L)
    _check_options(options)
    lb = Float64.(lowerbounds); ub = Float64.(upperbounds)
    maxf = -Inf; maxx = lb
    isempty(m.y) && return maxf, maxx
    if occursin("DIRECT", uppercase(string(options.method)))
        # the reference's default: dividing rectangles over x -> myrand(model, x), ONE posterior draw per new point
        # (mu + sigma z, src/models/gp.jl:6), a whole iteration's points per device call; `restarts` fresh runs
        f_batch = function (X)
            mu, var = mean_var(m, X)
            mu .+ sqrt.(max.(var, 0.0)) .* randn(length(mu))
        end
        for _ in 1:options.restarts
            f, x, _ = m isa BOHipMultiGPE ?
                _batched_direct_l(f_batch, lb, ub, max(1, options.maxeval); stopval = Float64(get(options, :stopval, Inf)),
                                  maxtime = Float64(get(options, :maxtime, 0.0))) :
                _direct_max_device(m, ACQ_THOMPSON_DRAW, [0.0, 0.0], lb, ub, options; seed = rand(UInt64))   # draws: the library's counter-based generator
            if f > maxf                                          # src/acquisition.jl:62 strict '>'
                maxf = f; maxx = x
            end
        end
        return maxf, maxx
    end
    # other derivative-free methods: one posterior draw per Latin-hypercube candidate, arg-max on the device
    _warn_not_a_local_search(options.method)
    for _ in 1:options.restarts
        cand = BO.latin_hypercube_sampling(lb, ub, max(options.maxeval, 1))
        best = Ref(Best(-Inf, -1))
        seed = rand(UInt64)
        rc = m isa BOHipMultiGPE ? c_mgp_thompson(m.handle, cand, size(cand, 2), 1, seed, best) :
                                   c_gp_thompson(m.handle, cand, size(cand, 2), 1, seed, 0, best)
        check(rc)
        if best[].idx >= 0 && best[].val > maxf
            maxf = best[].val; maxx = cand[:, best[].idx + 1]
        end
    end
    maxf, maxx
end
# the 5-argument method (src/acquisition.jl:48-51): nlopt_setup calls setparams! (:30), then the search
function acquire_max(a::AbstractAcquisition, m::AbstractBOHipModel, lowerbounds, upperbounds, options)
    setparams!(a, m)
    acquire_max_device(a, m, lowerbounds, upperbounds, options)
end

# =====================================================================================================================
# reference src/models/gp.jl:42-77 -- MAP hyper-parameter fit; value + analytic gradient of the marginal likelihood
# come from the device (bohip_gp_mll_grad), parameter order = GaussianProcesses.get_params: [logNoise; mean; kernel]
# =====================================================================================================================
function optimizemodel!(o::MAPGPOptimizer, model::AbstractBOHipModel)                     # :42-47
    if o.i % o.every == 0
        optimizemodel!(model, o.options)
    end
    o.i += 1
end
function _unpack!(m::AbstractBOHipModel, x, options)
    i = 0
    options.noise && (m.logNoise = x[i += 1])
    options.domean && m.meanconst && (m.mean = x[i += 1])
    if options.kern
        nl = length(m.loglen)
        m.loglen = collect(x[i+1:i+nl]); m.logsig = x[i+nl+1]
    end
    push_hyper!(m)
end
function _pack(m::AbstractBOHipModel, options)
    x = Float64[]
    options.noise && push!(x, m.logNoise)
    options.domean && m.meanconst && push!(x, m.mean)
    options.kern && (append!(x, m.loglen); push!(x, m.logsig))
    x
end
function _bounds(m::AbstractBOHipModel, options)                                          # GP.bounds(gp, noisebounds, meanbounds, kernbounds, likbounds)
    lb = Float64[]; ub = Float64[]
    if options.noise
        b = options.noisebounds === nothing ? [-Inf, Inf] : options.noisebounds
        push!(lb, b[1]); push!(ub, b[2])
    end
    if options.domean && m.meanconst
        b = options.meanbounds === nothing ? [[-Inf], [Inf]] : options.meanbounds
        append!(lb, b[1]); append!(ub, b[2])
    end
    if options.kern
        nk = length(m.loglen) + 1
        b = options.kernbounds === nothing ? [fill(-Inf, nk), fill(Inf, nk)] : options.kernbounds
        append!(lb, b[1]); append!(ub, b[2])
    end
    lb, ub
end
function optimizemodel!(m::AbstractBOHipModel, options)                                   # :54-77
    nk = length(m.loglen) + 1
    f = (x, g) -> begin                                                                   # :59-64
        _unpack!(m, x, options)
        mllv = Ref(0.0); dn = Ref(0.0); dm = Ref(0.0); dk = Vector{Float64}(undef, nk)
        rc = c_gp_mll_grad(gp_handle(m), mllv, dn, dm, dk)
        if rc == -2                                                                       # not positive definite for these parameters
            fill!(g, 0.0)
            return -1e300
        end
        check(rc)
        gi = Float64[]
        options.noise && push!(gi, dn[])
        options.domean && m.meanconst && push!(gi, dm[])
        options.kern && append!(gi, dk)
        length(g) > 0 && (g .= gi)
        mllv[]
    end
    lb, ub = _bounds(m, options)
    opt = NLopt.Opt(options.method, length(lb))
    NLopt.lower_bounds!(opt, lb)
    NLopt.upper_bounds!(opt, ub)
    NLopt.maxeval!(opt, options.maxeval)
    NLopt.max_objective!(opt, f)
    fx, x, ret = NLopt.optimize(opt, clamp.(_pack(m, options), lb, ub))
    ret == NLopt.FORCED_STOP && @warn("NLopt returned FORCED_STOP while optimizing the GP.")
    _unpack!(m, x, options)                                                               # leave the model at the optimum, factor fresh
    refit!(m)
    fx, x, ret
end

# =====================================================================================================================
# The optimisation object and the loop (src/BayesianOptimization.jl:59-207 without `opt::NLopt.Opt`)
# =====================================================================================================================
mutable struct DeviceBOpt{F, M, A, AO, MO, Ti}
    func::F
    sense::BO.Sense
    model::M
    acquisition::A
    acquisitionoptions::AO
    modeloptimizer::MO
    lowerbounds::Array{Float64, 1}
    upperbounds::Array{Float64, 1}
    observed_optimum::Float64
    observed_optimizer::Array{Float64, 1}
    model_optimum::Float64
    model_optimizer::Array{Float64, 1}
    iterations::IterationCounter
    duration::DurationCounter
    verbosity::BO.Verbosity
    initializer::Ti
    repetitions::Int
    timeroutput::TimerOutput
end

# A more specific method of the reference's constructor (:89-136): same arguments, same keywords, same checks.
function BOpt(func, model::AbstractBOHipModel, acquisition, modeloptimizer, lowerbounds, upperbounds;
              sense = Max, maxiterations = 10^4, maxduration = Inf, acquisitionoptions = NamedTuple(), repetitions = 1,
              verbosity = Progress, initializer_iterations = 5 * length(lowerbounds),
              initializer = ScaledSobolIterator(lowerbounds, upperbounds, initializer_iterations))
    tnow = time()
    acquisitionoptions = merge(defaultoptions(typeof(model), typeof(acquisition)), acquisitionoptions)
    maxiterations < length(initializer) &&
        throw(ArgumentError("maxiterations = $maxiterations < length(initializer) = $(length(initializer))"))
    maxiterations >= 0 || throw(ArgumentError("maxiterations < 0"))
    maxduration >= 0 || throw(ArgumentError("maxduration < 0"))
    length(lowerbounds) == length(upperbounds) ||
        throw(ArgumentError("length of lowerbounds does not match length of upperbounds"))
    all(lowerbounds .<= upperbounds) ||
        throw(ArgumentError("lowerbounds are not pointwise less than or eqal to upperbounds, they were possibly passed in the wrong order"))
    current_optimum = isempty(model.y) ? -Inf * Int(sense) : Int(sense) * maximum(model.y)
    current_optimizer = isempty(model.y) ? zero(float.(lowerbounds)) : Array(model.x[:, argmax(model.y)])
    _check_options(acquisitionoptions)
    setparams!(acquisition, model)                                                        # nlopt_setup :30 (called from the ctor, :134)
    DeviceBOpt(func, sense, model, acquisition, acquisitionoptions, modeloptimizer, float.(lowerbounds), float.(upperbounds),
               current_optimum, current_optimizer, current_optimum, copy(current_optimizer),
               IterationCounter(0, 0, maxiterations), DurationCounter(tnow, maxduration, tnow, tnow + maxduration),
               verbosity, initializer, repetitions, TimerOutput())
end
isdone(o::DeviceBOpt) = isdone(o.iterations) || isdone(o.duration)                        # :137
maxduration!(o::DeviceBOpt, d) = maxduration!(o.duration, d)
maxiterations!(o::DeviceBOpt, N) = maxiterations!(o.iterations, N)
acquire_max(o::DeviceBOpt) = acquire_max(o.acquisition, o.model, o.lowerbounds, o.upperbounds, o.acquisitionoptions)
function acquire_model_max(o::DeviceBOpt; options = o.acquisitionoptions)                 # src/acquisition.jl:45-47
    acquire_max(MaxMean(), o.model, o.lowerbounds, o.upperbounds, options)
end

function show(io::IO, mime::MIME"text/plain", o::DeviceBOpt)                              # :141-157
    println(io, "Bayesian Optimization object (device search, libbohip)\n\nmodel:")
    show(io, mime, o.model)
    println(io, "\nacquisition:")
    show(io, mime, o.acquisition)
    if o.iterations.i == 0
        println(io, "\nNo observation data.")
    else
        println(io, "\n\nobserved optimum: $(o.observed_optimum)")
        println(io, "observed optimizer: $(o.observed_optimizer)")
        println(io, "model optimum: $(o.model_optimum)")
        println(io, "model optimizer: $(o.model_optimizer)")
        println(io, "iterations: $(o.iterations.i)/$(o.iterations.N)")
        println(io, "duration: $(o.duration.now - o.duration.starttime)/$(o.duration.duration) s")
    end
end

"""
    boptimize!(o::DeviceBOpt)

src/BayesianOptimization.jl:176-207, statement for statement; the acquisition search (:185) runs on the device.
"""
function boptimize!(o::DeviceBOpt)
    BO.init!(o.duration)
    BO.init!(o.iterations)
    reset_timer!(o.timeroutput)
    o.iterations.i == 0 && length(o.initializer) > 0 && BO.initialise_model!(o)
    while !isdone(o)
        o.verbosity >= Progress &&
            @info("$(now())\titeration: $(o.iterations.i)\tcurrent optimum: $(o.observed_optimum)")
        setparams!(o.acquisition, o.model)                                                # :184
        @timeit o.timeroutput "acquisition" begin                                         # :185
            f, x = acquire_max_device(o.acquisition, o.model, o.lowerbounds, o.upperbounds, o.acquisitionoptions)
        end
        ys = Float64[]
        BO.step!(o.iterations)
        for _ in 1:(o.repetitions)
            y = BO._evaluate_function(o, x)
            push!(ys, y)
        end
        @timeit o.timeroutput "model update" update!(o.model, hcat(fill(x, o.repetitions)...), ys)                 # :194-196
        @timeit o.timeroutput "model hyperparameter optimization" optimizemodel!(o.modeloptimizer, o.model)         # :197
    end
    @timeit o.timeroutput "acquisition" begin
        o.model_optimum, o.model_optimizer = acquire_model_max(o)                         # :200
    end
    o.duration.now = time()
    o.verbosity >= Timings && @info(o.timeroutput)
    (observed_optimum = o.observed_optimum,
     observed_optimizer = o.observed_optimizer,
     model_optimum = Int(o.sense) * o.model_optimum,
     model_optimizer = o.model_optimizer)
end

# ---- sharded scoring from a process-per-GPU host (Distributed.jl / MPI.jl) ---------------------------------------------
"128 bytes rank 0 ships to the other ranks before `comm_init!`"
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(c_comm_unique_id(id, length(id)))
    id
end
comm_init!(m::BOHipGPE, id::Vector{UInt8}, rank::Integer, nranks::Integer) = check(c_gp_comm_init(m.handle, id, length(id), rank, nranks))
comm_destroy!(m::BOHipGPE) = check(c_gp_comm_destroy(m.handle))
"S Thompson draws over this rank's candidate columns [offset, offset + R_local) of R_total; S global (value, 1-based column) winners, identical on every rank"
function thompson_sharded(m::BOHipGPE, X::AbstractMatrix, S::Integer, seed::Integer, offset::Integer, R_total::Integer)
    Xc = _cols(m, X); best = Vector{Best}(undef, S)
    check(c_gp_thompson_sharded(m.handle, Xc, size(Xc, 2), S, UInt64(seed), offset, R_total, best))
    [b.val for b in best], [Int(b.idx) + 1 for b in best]
end
"score shards of a larger set with the summation schedule of the whole set (bit-identical to the unsharded call)"
set_batch_hint!(m::BOHipGPE, total::Integer) = check(c_gp_set_batch_hint(m.handle, total))

end # module
