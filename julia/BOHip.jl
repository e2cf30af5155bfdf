# BOHip.jl -- the reference-side binding a maintainer of jbrea/BayesianOptimization.jl would add so that
# `BOpt(f, BOHipGPE(...), ExpectedImprovement(), ...)` runs its GP posterior + acquisition scoring on an
# MI355X through libbohip.so.  NOT EXECUTED in this repository: there is no Julia toolchain in the build
# image or on the GPU box (SURVEY.md section 0 item 4).  Every `ccall` below is mirrored 1:1 by the ctypes
# binding in bayesianoptimization.jl_amd/_lib.py, which IS exercised by tests/ on the GPU.
#
# The model type plugs into the six generic functions through which the reference's loop touches a model
# (reference src/models/gp.jl:2-18) plus `defaultoptions` (src/acquisition.jl:4-9); nothing else in the
# reference has to change except `BOpt.opt::NLopt.Opt` (src/BayesianOptimization.jl:74), which the batched
# `acquire_max` below replaces.
module BOHip

import BayesianOptimization
const BO = BayesianOptimization

const libbohip = get(ENV, "BOHIP_LIB", "libbohip.so")

struct Best
    val::Float64
    idx::Int64
end

const KERN = Dict(:SEArd => 0, :SEIso => 1, :Mat52Ard => 2)
acqid(::BO.ExpectedImprovement) = 0
acqid(::BO.ProbabilityOfImprovement) = 1
acqid(::BO.UpperConfidenceBound) = 2
acqid(::BO.MutualInformation) = 3
acqid(::BO.MaxMean) = 4
acqparams(a::Union{BO.ExpectedImprovement, BO.ProbabilityOfImprovement}) = [a.τ, 0.0]
acqparams(a::BO.UpperConfidenceBound) = [a.βt, 0.0]
acqparams(a::BO.MutualInformation) = [a.sqrtα, a.γ̂]
acqparams(::BO.MaxMean) = [0.0, 0.0]

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:bohip_last_error, libbohip), Cstring, ()))
    rc == -2 && throw(LinearAlgebra.PosDefException(0))     # BOHIP_E_NOTPD
    error("libbohip error $rc: $msg")
end

"Device-resident elastic GP: drop-in for `ElasticGPE(d; mean, kernel, logNoise, capacity)` (README.md:22-27)."
mutable struct BOHipGPE
    handle::Ptr{Cvoid}
    dim::Int
    x::Matrix{Float64}          # d x n host mirror: the reference reads model.x / model.y directly
    y::Vector{Float64}          # (src/BayesianOptimization.jl:117-119, src/acquisitionfunctions.jl:136)
    hyper::Vector{Float64}      # [logNoise; mean; loglen...; logsig] = GP.get_params order
    function BOHipGPE(d::Integer; loglen = zeros(d), logsig = 0.0, logNoise = -2.0, mean = 0.0,
                      kernel::Symbol = :SEArd, capacity = 3000, device = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:bohip_gp_create, libbohip), Cint, (Int64, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
                    d, capacity, KERN[kernel], device, h))
        m = new(h[], d, zeros(d, 0), Float64[], vcat(logNoise, mean, Float64.(loglen), logsig))
        check(ccall((:bohip_gp_set_hyper, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64, Float64, Float64),
                    m.handle, Float64.(loglen), logsig, logNoise, mean))
        finalizer(g -> ccall((:bohip_gp_destroy, libbohip), Cvoid, (Ptr{Cvoid},), g.handle), m)
    end
end

# ---- reference src/models/gp.jl:2-18 ---------------------------------------------------------------------
function BO.mean_var(m::BOHipGPE, X::AbstractMatrix)                                     # :8
    R = size(X, 2); μ = Vector{Float64}(undef, R); σ² = similar(μ)
    Xc = Matrix{Float64}(X)                                                              # d x R column-major, as the ABI wants
    check(ccall((:bohip_gp_predict, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}),
                m.handle, Xc, R, μ, σ²))
    μ, σ²
end
BO.mean_var(m::BOHipGPE, x::AbstractVector) = ((μ, σ²) = BO.mean_var(m, reshape(x, :, 1)); (μ[1], σ²[1]))   # :2-5
BO.myrand(m::BOHipGPE, x::AbstractVector) = ((μ, σ²) = BO.mean_var(m, x); μ + sqrt(σ²) * randn())           # :6
BO.dims(m::BOHipGPE) = size(m.x)                                                          # :9
BO.maxy(m::BOHipGPE) = isempty(m.y) ? -Inf : maximum(m.y)                                 # :10
function BO.update!(m::BOHipGPE, x, y)                                                    # :11
    X = Matrix{Float64}(reshape(x, m.dim, :)); Y = Vector{Float64}(y)
    check(ccall((:bohip_gp_append, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
                m.handle, X, Y, length(Y)))
    m.x = hcat(m.x, X); append!(m.y, Y)
    m
end
BO.defaultoptions(::Type{BOHipGPE}, ::Type{<:BO.AbstractAcquisition}) = (method = :LD_LBFGS, restarts = 4096, maxeval = 200)

# ---- fused acquisitionfunction(a, model)(X) + arg-max of acquire_max (src/acquisitionfunctions.jl:4-9,
#      src/acquisition.jl:54-68): all R Latin-hypercube starts scored in ONE device call -----------------
function score(m::BOHipGPE, a::BO.AbstractAcquisition, X::AbstractMatrix)
    R = size(X, 2); sc = Vector{Float64}(undef, R); best = Ref(Best(-Inf, -1))
    check(ccall((:bohip_gp_score, libbohip), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ref{Best}),
                m.handle, acqid(a), acqparams(a), Matrix{Float64}(X), R, sc, best))
    sc, best[].val, best[].idx + 1                                                        # 1-based for Julia
end
function BO.acquire_max(a::BO.AbstractAcquisition, m::BOHipGPE, lowerbounds, upperbounds, options)
    BO.setparams!(a, m)
    starts = BO.latin_hypercube_sampling(lowerbounds, upperbounds, options.restarts)      # src/utils.jl:101-120
    if string(options.method)[2] == 'D'                                                   # :31  gradient-based: local search
        best = Ref(Best(-Inf, -1)); bx = Vector{Float64}(undef, m.dim); ev = Ref{Int64}(0)
        check(ccall((:bohip_gp_acquire_max, libbohip), Cint,
                    (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Float64, Float64,
                     Ptr{Float64}, Ptr{Float64}, Ref{Best}, Ptr{Float64}, Ref{Int64}),
                    m.handle, acqid(a), acqparams(a), Float64.(lowerbounds), Float64.(upperbounds), Matrix{Float64}(starts),
                    size(starts, 2), options.maxeval, get(options, :ftol_rel, 1e-10), get(options, :xtol_abs, 1e-10),
                    C_NULL, C_NULL, best, bx, ev))
        return best[].idx < 0 ? (-Inf, lowerbounds) : (best[].val, bx)
    end
    _, maxf, j = score(m, a, starts)
    j == 0 ? (-Inf, lowerbounds) : (maxf, starts[:, j])
end

"Sharded scoring: announce the size of the whole candidate set so a shard is summed exactly like the unsharded batch."
set_batch_hint!(m::BOHipGPE, total::Integer) =
    check(ccall((:bohip_gp_set_batch_hint, libbohip), Cint, (Ptr{Cvoid}, Int64), m.handle, total))

# ---- reference src/models/gp.jl:42-77: MAP hyper-parameter fit -------------------------------------------
# f = (x, g) -> (set_params!; update_target_and_dtarget!; g .= gp.dtarget; gp.target) with the device doing the
# rebuild, the marginal likelihood and its analytic gradient; parameter order [logNoise; mean; loglen...; logsig].
function target_and_dtarget!(m::BOHipGPE, x::Vector{Float64}, g::Vector{Float64})
    d = m.dim
    check(ccall((:bohip_gp_set_hyper, libbohip), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64, Float64, Float64),
                m.handle, x[3:2+d], x[3+d], x[1], x[2]))
    mll = Ref(0.0); dn = Ref(0.0); dm = Ref(0.0); dk = Vector{Float64}(undef, d + 1)
    check(ccall((:bohip_gp_mll_grad, libbohip), Cint,
                (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}, Ref{Float64}, Ptr{Float64}), m.handle, mll, dn, dm, dk))
    g[1] = dn[]; g[2] = dm[]; g[3:end] .= dk
    m.hyper .= x
    mll[]
end
function BO.optimizemodel!(m::BOHipGPE, options)                                          # :54-77
    d = m.dim
    lb = vcat(something(options.noisebounds, [-Inf, Inf])[1], -Inf, something(options.kernbounds, [fill(-Inf, d + 1), fill(Inf, d + 1)])[1])
    ub = vcat(something(options.noisebounds, [-Inf, Inf])[2], Inf, something(options.kernbounds, [fill(-Inf, d + 1), fill(Inf, d + 1)])[2])
    opt = BO.NLopt.Opt(options.method, d + 3)
    BO.NLopt.lower_bounds!(opt, lb); BO.NLopt.upper_bounds!(opt, ub); BO.NLopt.maxeval!(opt, options.maxeval)
    BO.NLopt.max_objective!(opt, (x, g) -> target_and_dtarget!(m, x, g))
    BO.NLopt.optimize(opt, copy(m.hyper))
end

end # module
