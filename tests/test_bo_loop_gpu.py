"""GPU tests of the drop-in API surface: the reference's integration tests restated against the host mirror
(test/acquisition.jl, test/acquisitionfunctions.jl, test/warmstart.jl, test/branin.jl, README example)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bo():
    import bohip

    assert bohip._lib.load().bohip_device_count() > 0
    return bohip


def branin(x, noiselevel=0.0, rng=None):                               # test/branin.jl:1-5
    x1, x2 = x
    a, b, c, r, s, t = 1, 5.1 / (4 * math.pi ** 2), 5 / math.pi, 6, 10, 1 / (8 * math.pi)
    n = noiselevel * (rng.standard_normal() if rng is not None else 0.0)
    return a * (x2 - b * x1 ** 2 + c * x1 - r) ** 2 + s * (1 - t) * math.cos(x1) + s + n


BRANIN_MIN = 0.397887


def test_acquire_max_maxmean_single_observation(bo):                    # test/acquisition.jl:2,11-12
    model = bo.GPE(np.array([1.0]), np.array([2.0]), bo.MeanZero(), bo.SEIso(1.0, 0.0))
    ac = bo.MaxMean()
    opts = {**bo.defaultoptions(type(model), type(ac)), **dict(maxtime=3.0, ftol_abs=np.finfo(float).eps)}
    import warnings
    with warnings.catch_warnings():                                      # maxtime AND ftol_abs are honoured by the device ascent: no "ignored" warning
        warnings.simplefilter("error")
        maxf, maxx = bo.acquire_max(ac, model, [-5.0], [5.0], {**opts, "restarts": 10}, np.random.default_rng(0))
    assert maxx == pytest.approx([1.0], abs=1e-5)                       # @test maxx ≈ [1.0]
    assert maxf == pytest.approx(2 / (1 + math.exp(-4.0)), rel=1e-9)


def test_acquisition_functions_batch_and_single(bo):                    # test/acquisitionfunctions.jl:1-12
    rng = np.random.default_rng(0)
    for ac in [bo.ProbabilityOfImprovement(), bo.ExpectedImprovement(), bo.UpperConfidenceBound(),
               bo.ThompsonSamplingSimple(), bo.MutualInformation()]:
        model = bo.GPE(rng.random((3, 4)), rng.random(4), bo.MeanZero(), bo.SEIso(0.0, 0.0))
        bo.setparams_(ac, model)
        acfunc = bo.acquisitionfunction(ac, model, rng)
        x = rng.random((3, 2))
        acvector = acfunc(x)
        assert len(acvector) == 2
        if not isinstance(ac, bo.ThompsonSamplingSimple):
            assert acvector[0] == acfunc(x[:, 0])                       # bit-exact, as the reference asserts


def make_opt(bo, model, ac, **kw):
    return bo.BOpt(lambda x: branin(x), model, ac,
                   bo.MAPGPOptimizer(every=50, noisebounds=[-4, 3], kernbounds=[[-1, -1, 0], [4, 4, 10]], maxeval=40),
                   [-5.0, 0.0], [10.0, 15.0], sense=bo.Min, verbosity=bo.Silent, rng=np.random.default_rng(5), **kw)


def test_warmstart_bookkeeping(bo):                                     # test/warmstart.jl
    rng = np.random.default_rng(123)
    x_premade = rng.random((2, 10)) * 15.0 - np.array([[5.0], [0.0]])
    y_premade = -np.array([branin(x_premade[:, i]) for i in range(10)])
    new_model = lambda: bo.ElasticGPE(2, mean=bo.MeanConst(-10.0), kernel=bo.SEArd([0.0, 0.0], 5.0), logNoise=-2.0, capacity=3000)
    # "Initial Sampling Tracking" :18-33
    opt = make_opt(bo, new_model(), bo.ExpectedImprovement(), maxiterations=10, initializer_iterations=10)
    bo.boptimize_(opt)
    assert opt.observed_optimum == int(opt.sense) * np.max(opt.model.y)
    assert len(opt.model.y) == 10
    # "Pre-made model" :36-52
    model_premade = new_model()
    model_premade.append_(x_premade, y_premade)
    opt = make_opt(bo, model_premade, bo.ExpectedImprovement(), maxiterations=10, initializer_iterations=5)
    assert opt.observed_optimum == int(opt.sense) * np.max(y_premade)
    assert np.array_equal(opt.observed_optimizer, x_premade[:, int(np.argmax(y_premade))])
    # "Initial iterations to 0 on pre-made model" :55-76
    ac = bo.ExpectedImprovement()
    opt = make_opt(bo, model_premade, ac, maxiterations=0, initializer_iterations=0)
    bo.boptimize_(opt)
    assert opt.acquisition.tau == np.max(y_premade)                     # :64
    assert opt.model.x.size == x_premade.size and len(opt.model.y) == len(y_premade)
    opt.iterations.N = 5
    bo.boptimize_(opt)
    assert len(opt.model.y) == len(y_premade) + 5                       # resuming appends exactly 5
    assert opt.model.info(3) >= 5                                       # ... through the incremental device path


def test_constructor_validation(bo):                                    # src/BayesianOptimization.jl:107-115
    m = bo.ElasticGPE(2)
    common = (lambda x: 0.0, m, bo.ExpectedImprovement(), bo.NoModelOptimizer())
    with pytest.raises(ValueError, match="maxiterations"):
        bo.BOpt(*common, [0, 0], [1, 1], maxiterations=3, initializer_iterations=10)
    with pytest.raises(ValueError, match="lowerbounds"):
        bo.BOpt(*common, [0, 0], [1, -1])
    with pytest.raises(ValueError, match="length"):
        bo.BOpt(*common, [0, 0], [1])
    with pytest.raises(ValueError):
        bo.BOpt(*common, [0, 0], [1, 1], maxduration=-1)


@pytest.mark.parametrize("acname", ["EI", "UCB", "PI", "MI", "Thompson"])
def test_branin_regret(bo, acname):                                     # test/branin.jl:17-38 (regret < 0.05)
    ac = {"EI": bo.ExpectedImprovement, "UCB": bo.UpperConfidenceBound, "PI": bo.ProbabilityOfImprovement,
          "MI": bo.MutualInformation, "Thompson": bo.ThompsonSamplingSimple}[acname]()
    model = bo.ElasticGPE(2, mean=bo.MeanConst(-10.0), kernel=bo.SEArd([0.0, 0.0], 5.0), logNoise=-2.0, capacity=3000)
    opt = make_opt(bo, model, ac, maxiterations=120)
    res = bo.boptimize_(opt)
    assert abs(res["observed_optimum"] - BRANIN_MIN) < 0.05
    assert len(model.y) == 120 and model.info(3) >= 100                 # appends were incremental
    mins = [(-math.pi, 12.275), (math.pi, 2.275), (9.42478, 2.475)]
    assert min(math.dist(m, res["observed_optimizer"]) for m in mins) < 0.5


def test_readme_example_shape(bo):                                      # README.md:15-48: repetitions=5, UCB, Min, restarts=5
    rng = np.random.default_rng(0)
    f = lambda x: float(np.sum((x - 1) ** 2) + rng.standard_normal())
    model = bo.ElasticGPE(2, mean=bo.MeanConst(0.0), kernel=bo.SEArd([0.0, 0.0], 5.0), logNoise=0.0, capacity=3000)
    opt = bo.BOpt(f, model, bo.UpperConfidenceBound(),
                  bo.MAPGPOptimizer(every=50, noisebounds=[-4, 3], kernbounds=[[-1, -1, 0], [4, 4, 10]], maxeval=40),
                  [-5.0, -5.0], [5.0, 5.0], repetitions=5, maxiterations=30, sense=bo.Min,
                  acquisitionoptions=dict(method="LD_LBFGS", restarts=5, maxtime=0.1, maxeval=1000), verbosity=bo.Silent,
                  rng=np.random.default_rng(1))
    res = bo.boptimize_(opt)
    assert len(model.y) == 30 * 5 and model.x.shape == (2, 150)
    assert np.linalg.norm(res["model_optimizer"] - 1.0) < 1.0           # noisy quadratic with minimum at (1, 1)
    assert set(res) == {"observed_optimum", "observed_optimizer", "model_optimum", "model_optimizer"}   # :203-206
    assert {"acquisition", "model update", "function evaluation"} <= set(opt.timeroutput)


def test_optimize_convenience_defaults(bo):                             # src/BayesianOptimization.jl:230-234, 259-272
    res = bo.optimize(lambda x: -float(np.sum((x - 0.3) ** 2)), [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], maxiterations=25,
                      verbosity=bo.Silent)
    assert res["observed_optimum"] > -0.05
