"""The reference's README example (README.md:15-48 of jbrea/BayesianOptimization.jl) with the same names and keywords,
running on libbohip (needs an MI355X).  Julia's `f!` is spelled `f_`; symbols (:LD_LBFGS) are strings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bohip as bo

f = lambda x: float(np.sum((x - 1) ** 2) + np.random.randn())            # noisy objective, minimum at (1, 1)

model = bo.ElasticGPE(2,                                                   # 2 input dimensions
                      mean=bo.MeanConst(0.0), kernel=bo.SEArd([0.0, 0.0], 5.0), logNoise=0.0,
                      capacity=3000)                                       # the initial capacity of the GP is 3000 samples
# NOT PORTED: README.md:27 `set_priors!(model.mean, [Normal(1, 2)])` -- a prior on the mean-function parameter for the MAP fit of
# optimizemodel!.  Priors on hyper-parameters live in GaussianProcesses.jl / Distributions.jl on the host and are out of this build's
# scope: the MAP fit below maximises the marginal likelihood alone, within the bounds.
modeloptimizer = bo.MAPGPOptimizer(every=50, noisebounds=[-4, 3],          # bounds of the logNoise
                                   kernbounds=[[-1, -1, 0], [4, 4, 10]],   # bounds of the 3 parameters GaussianProcesses.get_param_names(model.kernel)
                                   maxeval=40)
opt = bo.BOpt(f, model,
              bo.UpperConfidenceBound(),                                   # type of acquisition
              modeloptimizer,
              [-5.0, -5.0], [5.0, 5.0],                                    # lowerbounds, upperbounds
              repetitions=5,                                               # evaluate the function for each input 5 times
              maxiterations=100,                                           # evaluate at 100 input positions
              sense=bo.Min,                                                # minimize the function
              acquisitionoptions=dict(method="LD_LBFGS",                   # run optimization of acquisition function with NLopts :LD_LBFGS method
                                      restarts=5,                          # run the NLopt method from 5 random initial conditions each time
                                      maxtime=0.1,                         # run the NLopt method for at most 0.1 second each time
                                      maxeval=1000),                       # run the NLopt methods for at most 1000 iterations
              verbosity=bo.Progress)

result = bo.boptimize_(opt)
print("observed optimum", result["observed_optimum"], "at", result["observed_optimizer"])
print("model optimum   ", result["model_optimum"], "at", result["model_optimizer"])
