"""bohip -- MI355X-native GP-posterior + acquisition scoring behind the BayesianOptimization.jl API surface.

Host mirror (Python, since no Julia toolchain exists in the build image) of the reference's
exported names; all arithmetic on the hot path runs in libbohip.so (HIP, gfx950).  See DESIGN.md.
"""
from . import _lib
from ._lib import BohipError, NotPositiveDefinite
from .model import (ElasticGPE, MeanConst, MeanZero, SEArd, SEIso, Mat52Ard, mean_var, myrand, dims, maxy,
                    update_)

__all__ = ["ElasticGPE", "MeanConst", "MeanZero", "SEArd", "SEIso", "Mat52Ard", "mean_var", "myrand", "dims",
           "maxy", "update_", "BohipError", "NotPositiveDefinite"]
