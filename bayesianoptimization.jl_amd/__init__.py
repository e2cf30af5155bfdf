"""bohip -- MI355X-native GP-posterior + acquisition scoring behind the BayesianOptimization.jl API surface.

Host mirror (Python, since no Julia toolchain exists in the build image) of the reference's exported names
(src/BayesianOptimization.jl:20-40; Julia's ``f!`` is spelled ``f_``); all arithmetic on the hot path runs in
libbohip.so (HIP, gfx950).  See DESIGN.md / INTEGRATION.md.
"""
from . import _lib
from ._lib import BohipError, NotPositiveDefinite
from .model import (ElasticGPE, MeanConst, MeanZero, SEArd, SEIso, Mat52Ard, mean_var, myrand, dims, maxy, update_)
from .multigpu import MultiGPE, comm_unique_id
from .acquisition import (ExpectedImprovement, ProbabilityOfImprovement, UpperConfidenceBound, ThompsonSamplingSimple,
                          MutualInformation, MaxMean, BrochuBetaScaling, NoBetaScaling, acquisitionfunction, setparams_,
                          acquire_max, acquire_model_max, defaultoptions)
from .bopt import (BOpt, boptimize_, optimize, merge_with_defaults, MAPGPOptimizer, NoModelOptimizer, optimizemodel_,
                   Min, Max, Silent, Timings, Progress, isdone)
from .utils import (ScaledSobolIterator, ScaledLHSIterator, latin_hypercube_sampling, maxduration_, maxiterations_,
                    IterationCounter, DurationCounter)

GPE = ElasticGPE.from_data

__all__ = ["BOpt", "ExpectedImprovement", "ProbabilityOfImprovement", "UpperConfidenceBound", "ThompsonSamplingSimple",
           "MutualInformation", "boptimize_", "MAPGPOptimizer", "NoModelOptimizer", "Min", "Max", "BrochuBetaScaling",
           "NoBetaScaling", "Silent", "Timings", "Progress", "ScaledSobolIterator", "ScaledLHSIterator",
           "maxduration_", "maxiterations_", "optimize",
           "ElasticGPE", "MultiGPE", "GPE", "MeanConst", "MeanZero", "SEArd", "SEIso", "Mat52Ard"]
