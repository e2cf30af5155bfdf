"""Host-side helpers mirroring reference src/utils.jl (counters, normal_pdf/cdf, Sobol / LHS initialisers)."""
from __future__ import annotations

import math
import time

import numpy as np


# ---- src/utils.jl:9-23 ---------------------------------------------------------------------------------
class IterationCounter:
    def __init__(self, c=0, i=0, N=0):
        self.c, self.i, self.N = c, i, N


class DurationCounter:
    def __init__(self, starttime, duration, now, endtime):
        self.starttime, self.duration, self.now, self.endtime = starttime, duration, now, endtime


def isdone(s):
    if isinstance(s, IterationCounter):
        return s.c == s.N                                    # :14
    s.now = time.time()                                      # :35
    return s.now >= s.endtime


def step_(s: IterationCounter):                              # step! :15
    s.c += 1
    s.i += 1


def init_(s):                                                # init! :16 / :31-34
    if isinstance(s, IterationCounter):
        s.c = 0
    else:
        s.starttime = time.time()
        s.endtime = s.starttime + s.duration


def maxiterations_(s, N):                                    # maxiterations! :23
    (s.iterations if hasattr(s, "iterations") else s).N = N


def maxduration_(s, d):                                      # maxduration! :42
    (s.duration if hasattr(s, "iterations") else s).duration = d


def sample(lowerbounds, upperbounds, rng=None):              # :44-46
    rng = rng if rng is not None else np.random.default_rng()
    lb, ub = np.asarray(lowerbounds, float), np.asarray(upperbounds, float)
    return rng.random(lb.size) * (ub - lb) + lb


def normal_pdf(mu, s2):                                      # :48
    return 1 / math.sqrt(2 * math.pi * s2) * math.exp(-mu ** 2 / (2 * s2))


def normal_cdf(mu, s2):                                      # :49
    return 1 / 2 * (1 + math.erf(mu / math.sqrt(2 * s2)))


# ---- initialisers: src/utils.jl:64-120 -----------------------------------------------------------------
class ScaledSobolIterator:
    """N points of a Sobol sequence scaled to [lb, ub]; the first N points are skipped (src/utils.jl:79-83).
    (scipy's generator; the direction numbers differ from Sobol.jl, so values are not bit-comparable.)"""

    def __init__(self, lowerbounds, upperbounds, N, seed=None):
        from scipy.stats import qmc

        self.lowerbounds = np.asarray(lowerbounds, float)
        self.upperbounds = np.asarray(upperbounds, float)
        self.N = int(N)
        self._seq = qmc.Sobol(d=self.lowerbounds.size, scramble=False, seed=seed)
        if self.N > 0:
            self._seq.fast_forward(self.N)

    def __len__(self):
        return self.N

    def __iter__(self):
        for _ in range(self.N):
            u = self._seq.random(1)[0]
            yield self.lowerbounds + u * (self.upperbounds - self.lowerbounds)


def latin_hypercube_sampling(mins, maxs, n, rng=None):
    """src/utils.jl:101-120: one point per stratum and dimension, independent shuffles.  Returns d x n."""
    rng = rng if rng is not None else np.random.default_rng()
    mins, maxs = np.asarray(mins, float), np.asarray(maxs, float)
    if mins.shape != maxs.shape:
        raise ValueError("mins and maxs should have the same length")            # DimensionMismatch :104-105
    if not np.all(mins <= maxs):
        raise ValueError("mins[i] should not exceed maxs[i]")                    # ArgumentError :106-107
    dims = mins.size
    result = np.zeros((dims, n), order="F")
    for i in range(dims):
        dimstep = (maxs[i] - mins[i]) / n
        cubedim = mins[i] + dimstep * (np.arange(n) + rng.random(n))
        rng.shuffle(cubedim)
        result[i, :] = cubedim
    return result


class ScaledLHSIterator:
    """src/utils.jl:96-98: column iterator over a Latin-hypercube sample."""

    def __init__(self, lowerbounds, upperbounds, N, rng=None):
        self.data = latin_hypercube_sampling(lowerbounds, upperbounds, N, rng)

    def __len__(self):
        return self.data.shape[1]

    def __iter__(self):
        for j in range(self.data.shape[1]):
            yield self.data[:, j]
