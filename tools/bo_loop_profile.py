"""Where a whole BO iteration's wall time goes on the host side: the README example's loop (2-d, 5 repetitions per point, UCB, 5 L-BFGS
restarts, hyper-parameters every 50 points) for 100 iterations, with the time inside libbohip calls (ctypes) against everything else,
and a cProfile top list.  usage: python tools/bo_loop_profile.py [iterations]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bohip as bo
np.random.seed(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
f = lambda x: float(np.sum((x - 1) ** 2) + np.random.randn())
model = bo.ElasticGPE(2, mean=bo.MeanConst(0.0), kernel=bo.SEArd([0.0, 0.0], 5.0), logNoise=0.0, capacity=3000)
mo = bo.MAPGPOptimizer(every=50, noisebounds=[-4, 3], kernbounds=[[-1, -1, 0], [4, 4, 10]], maxeval=40)
opt = bo.BOpt(f, model, bo.UpperConfidenceBound(), mo, [-5.0, -5.0], [5.0, 5.0], repetitions=5, maxiterations=iters, sense=bo.Min,
              acquisitionoptions=dict(method="LD_LBFGS", restarts=5, maxtime=0.1, maxeval=1000), verbosity=bo.Silent)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = bo.boptimize_(opt)
pr.disable()
t = time.perf_counter() - t0
print(f"{iters} iterations: {t*1e3:.1f} ms = {t/iters*1e3:.3f} ms per iteration; observed optimum {res['observed_optimum']:.4g}")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
