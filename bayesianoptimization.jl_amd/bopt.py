"""BOpt / boptimize_ / optimize: the reference's public loop (src/BayesianOptimization.jl) driven from the host;
"acquisition" and "model update" -- the two timed regions the reference brackets (:185,:194) -- run on the GPU."""
from __future__ import annotations

import enum
import math
import time
import warnings

import numpy as np

from .acquisition import (AbstractAcquisition, ExpectedImprovement, MaxMean, acquire_max, defaultoptions, setparams_)
from ._lib import NotPositiveDefinite
from .model import ElasticGPE, Mat52Ard, MeanConst, update_
from .utils import (DurationCounter, IterationCounter, ScaledSobolIterator, init_, isdone as _isdone, step_)


class Sense(enum.IntEnum):                                    # :56
    Min = -1
    Max = 1


class Verbosity(enum.IntEnum):                                # :57
    Silent = 0
    Timings = 1
    Progress = 2


Min, Max = Sense.Min, Sense.Max
Silent, Timings, Progress = Verbosity.Silent, Verbosity.Timings, Verbosity.Progress


class ModelOptimizer:                                         # :44
    pass


class NoModelOptimizer(ModelOptimizer):                       # :48-49
    """Don't optimize the model ever."""


class MAPGPOptimizer(ModelOptimizer):
    """src/models/gp.jl:20-52: MAP hyper-parameter fit every ``every`` calls.  Each objective evaluation is a
    full device rebuild (kernel matrix + Cholesky + alpha) plus the analytic gradient 1/2 tr((aa' - cK^-1) dcK)
    formed on the device (bohip_gp_mll_grad); the bounded L-BFGS search itself stays on the host, like the
    reference's NLopt :LD_LBFGS driving GP.update_target_and_dtarget!."""

    def __init__(self, every=10, **kwargs):
        self.i = 0
        self.every = every
        self.options = {**self.defaultoptions(), **kwargs}

    @staticmethod
    def defaultoptions():                                     # :48-52
        return dict(domean=True, kern=True, noise=True, lik=True, meanbounds=None, kernbounds=None, noisebounds=None,
                    likbounds=None, method="LD_LBFGS", maxeval=500)


def optimizemodel_(o, model):                                 # :42-47 and NoModelOptimizer :49
    if isinstance(o, NoModelOptimizer) or o is None:
        return None
    if o.i % o.every == 0:
        _map_fit(model, o.options)
    o.i += 1


def _map_fit(model, opt):                                     # :54-77
    from scipy.optimize import minimize

    if model.nobs == 0:
        return
    names, x0, lo, hi = [], [], [], []
    if opt["noise"]:
        names.append("noise"); x0.append(model.logNoise)
        b = opt["noisebounds"] if opt["noisebounds"] is not None else [-math.inf, math.inf]
        lo.append(b[0]); hi.append(b[1])
    if opt["domean"] and isinstance(model.mean, MeanConst):
        names.append("mean"); x0.append(model.mean.beta)
        b = opt["meanbounds"] if opt["meanbounds"] is not None else [[-math.inf], [math.inf]]
        lo.append(np.ravel(b[0])[0]); hi.append(np.ravel(b[1])[0])
    nk = 0
    if opt["kern"]:
        kp = np.concatenate([model.kernel.ll, [model.kernel.lsigma]])
        nk = kp.size
        names += ["kern"] * nk; x0 += kp.tolist()
        b = opt["kernbounds"] if opt["kernbounds"] is not None else [[-math.inf] * nk, [math.inf] * nk]
        lo += list(np.ravel(b[0]).astype(float)); hi += list(np.ravel(b[1]).astype(float))
    x0 = np.clip(np.array(x0, float), lo, hi)

    def apply(x):
        i = 0
        kw = {}
        if opt["noise"]:
            kw["logNoise"] = x[i]; i += 1
        if opt["domean"] and isinstance(model.mean, MeanConst):
            kw["beta"] = x[i]; i += 1
        if opt["kern"]:
            kw["ll"] = x[i:i + nk - 1]; kw["lsigma"] = x[i + nk - 1]
        model.set_params_(**kw)

    def negmll(x):                                            # f = (x, g) -> ... gp.target, gp.dtarget  (:59-64)
        apply(x)
        try:
            m, dn, dm, dk = model.mll_grad()
        except NotPositiveDefinite:                           # not positive definite for these parameters; device
            return 1e300, np.zeros_like(x)                    # failures (E_HIP, E_NODEVICE, ...) propagate
        g = []
        if opt["noise"]:
            g.append(dn)
        if opt["domean"] and isinstance(model.mean, MeanConst):
            g.append(dm)
        if opt["kern"]:
            g += dk.tolist()
        return -m, -np.asarray(g, dtype=float)

    res = minimize(negmll, x0, jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)),
                   options=dict(maxfun=int(opt["maxeval"])))
    best = res.x if np.isfinite(res.fun) and res.fun < 1e299 else x0
    apply(best)
    model.fit_()
    return -res.fun, best


class BOpt:
    """src/BayesianOptimization.jl:59-136 (same positional arguments, keyword names and validation)."""

    def __init__(self, func, model, acquisition, modeloptimizer, lowerbounds, upperbounds, *, sense=Max,
                 maxiterations=10 ** 4, maxduration=math.inf, acquisitionoptions=None, repetitions=1,
                 verbosity=Progress, initializer_iterations=None, initializer=None, rng=None):
        now = time.time()
        lowerbounds = np.asarray(lowerbounds, dtype=np.float64)
        upperbounds = np.asarray(upperbounds, dtype=np.float64)
        if initializer_iterations is None:
            initializer_iterations = 5 * len(lowerbounds)                        # :101
        if initializer is None:
            initializer = ScaledSobolIterator(lowerbounds, upperbounds, initializer_iterations)
        acquisitionoptions = {**defaultoptions(type(model), type(acquisition)), **(acquisitionoptions or {})}   # :105-106
        if maxiterations < len(initializer):
            raise ValueError(f"maxiterations = {maxiterations} < length(initializer) = {len(initializer)}")      # :107-108
        if not maxiterations >= 0:
            raise ValueError("maxiterations < 0")
        if not maxduration >= 0:
            raise ValueError("maxduration < 0")
        if len(lowerbounds) != len(upperbounds):
            raise ValueError("length of lowerbounds does not match length of upperbounds")
        if not np.all(lowerbounds <= upperbounds):
            raise ValueError("lowerbounds are not pointwise less than or eqal to upperbounds, they were possibly "
                             "passed in the wrong order")
        empty = model.y.size == 0
        current_optimum = -math.inf * int(sense) if empty else int(sense) * float(np.max(model.y))   # :117
        current_optimizer = np.zeros_like(lowerbounds) if empty else np.array(model.x[:, int(np.argmax(model.y))])
        self.func, self.sense, self.model = func, Sense(sense), model
        self.acquisition, self.acquisitionoptions, self.modeloptimizer = acquisition, acquisitionoptions, modeloptimizer
        self.lowerbounds, self.upperbounds = lowerbounds, upperbounds
        self.observed_optimum, self.observed_optimizer = current_optimum, current_optimizer
        self.model_optimum, self.model_optimizer = current_optimum, current_optimizer.copy()
        self.iterations = IterationCounter(0, 0, maxiterations)
        self.duration = DurationCounter(now, maxduration, now, now + maxduration)
        self.verbosity, self.initializer, self.repetitions = Verbosity(verbosity), initializer, repetitions
        self.rng = rng if rng is not None else np.random.default_rng()
        self.timeroutput = {}
        setparams_(acquisition, model)                                          # nlopt_setup :30 (ctor :134)

    def __repr__(self):                                                          # show :141-157
        s = f"Bayesian Optimization object\n\nmodel:\n{self.model!r}\n\nacquisition:\n{type(self.acquisition).__name__}"
        if self.iterations.i == 0:
            return s + "\n\nNo observation data."
        return (s + f"\n\nobserved optimum: {self.observed_optimum}\nobserved optimizer: {self.observed_optimizer}"
                f"\nmodel optimum: {self.model_optimum}\nmodel optimizer: {self.model_optimizer}"
                f"\niterations: {self.iterations.i}/{self.iterations.N}"
                f"\nduration: {self.duration.now - self.duration.starttime}/{self.duration.duration} s")


def isdone(o):                                                                  # :137
    return _isdone(o.iterations) or _isdone(o.duration)


class _timeit:
    """@mytimeit (src/utils.jl:1-7) with the reference's section names."""

    def __init__(self, o, name):
        self.o, self.name = o, name

    def __enter__(self):
        self.t = time.perf_counter()

    def __exit__(self, *a):
        rec = self.o.timeroutput.setdefault(self.name, [0, 0.0])
        rec[0] += 1
        rec[1] += time.perf_counter() - self.t


def _evaluate_function(o, x):                                                   # :209-216
    with _timeit(o, "function evaluation"):
        y = int(o.sense) * o.func(x)
    if y > int(o.sense) * o.observed_optimum:
        o.observed_optimum = int(o.sense) * y
        o.observed_optimizer = x
    return y


def initialise_model_(o):                                                       # :159-172
    ys, xs = [], []
    for x in o.initializer:
        for _ in range(o.repetitions):
            ys.append(_evaluate_function(o, x))
            xs.append(x)
    o.iterations.i = o.iterations.c = len(ys) // o.repetitions
    with _timeit(o, "model update"):
        update_(o.model, np.stack(xs, axis=1), np.array(ys))
    with _timeit(o, "model hyperparameter optimization"):
        optimizemodel_(o.modeloptimizer, o.model)


def boptimize_(o):
    """boptimize!(o) :176-207.  Re-calling resumes: init! zeroes the per-call counter but keeps the cumulative one."""
    init_(o.duration)
    init_(o.iterations)
    o.timeroutput.clear()
    if o.iterations.i == 0 and len(o.initializer) > 0:
        initialise_model_(o)
    while not isdone(o):
        if o.verbosity >= Progress:
            print(f"{time.strftime('%Y-%m-%dT%H:%M:%S')}\titeration: {o.iterations.i}\tcurrent optimum: {o.observed_optimum}")
        setparams_(o.acquisition, o.model)                                       # :184
        with _timeit(o, "acquisition"):
            f, x = acquire_max(o.acquisition, o.model, o.lowerbounds, o.upperbounds, o.acquisitionoptions, o.rng,
                               setparams=False)                                  # :185 (4-argument method: no second setparams!)
        ys = []
        step_(o.iterations)
        for _ in range(o.repetitions):
            ys.append(_evaluate_function(o, x))
        with _timeit(o, "model update"):
            update_(o.model, np.stack([x] * o.repetitions, axis=1), np.array(ys))   # :194-196
        with _timeit(o, "model hyperparameter optimization"):
            optimizemodel_(o.modeloptimizer, o.model)
    with _timeit(o, "acquisition"):
        if o.model.nobs > 0:
            o.model_optimum, o.model_optimizer = acquire_max(MaxMean(), o.model, o.lowerbounds, o.upperbounds,
                                                             o.acquisitionoptions, o.rng)   # acquire_model_max :200
    o.duration.now = time.time()
    if o.verbosity >= Timings:
        for k, (n, t) in o.timeroutput.items():
            print(f"  {k:40s} calls {n:6d}  {t:10.4f} s")
    return dict(observed_optimum=o.observed_optimum, observed_optimizer=o.observed_optimizer,
                model_optimum=int(o.sense) * o.model_optimum, model_optimizer=o.model_optimizer)   # :203-206


def merge_with_defaults(f, lowerbounds, upperbounds, optkwargs):                # :238-289
    args_keys = ("model", "acquisition", "modeloptimizer")
    kwargs_keys = ("sense", "maxiterations", "maxduration", "acquisitionoptions", "repetitions", "verbosity",
                   "initializer_iterations", "initializer")
    if not set(optkwargs) <= set(args_keys) | set(kwargs_keys):
        raise ValueError("use of unsupported keyword arguments")                 # ArgumentError :250-251
    if len(lowerbounds) != len(upperbounds):
        raise ValueError("length of lowerbounds does not match length of upperbounds")
    inputdimension = len(lowerbounds)
    params = dict(optkwargs)
    if "model" not in params:                                                    # :259-264
        params["model"] = ElasticGPE(inputdimension, mean=MeanConst(0.0),
                                     kernel=Mat52Ard(np.zeros(inputdimension), 0.0), logNoise=-2.0, capacity=3000)
    if "acquisition" not in params:
        params["acquisition"] = ExpectedImprovement()
    if "modeloptimizer" not in params:                                           # :266-272
        params["modeloptimizer"] = MAPGPOptimizer(every=20, noisebounds=[-4, 3],
                                                  kernbounds=[[-3.0] * inputdimension + [-3.0], [4.0] * inputdimension + [3.0]],
                                                  maxeval=100)
    params.setdefault("maxiterations", 10 ** 3)
    args = (f, *[params[k] for k in args_keys], lowerbounds, upperbounds)
    kwargs = {k: v for k, v in params.items() if k in kwargs_keys}
    return args, kwargs


def optimize(f, lowerbounds, upperbounds, **optkwargs):                         # :230-234
    args, kwargs = merge_with_defaults(f, lowerbounds, upperbounds, optkwargs)
    return boptimize_(BOpt(*args, **kwargs))
