"""Secondary measurements (not the driver's bench line): BASELINE.json configs[2..4] on ONE GPU plus the
model-update path.  Prints one JSON object per config.   python tools/bench_configs.py [c2 c3 c4 c5 append]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bohip

def synth(N, d, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    return X, y

def model_for(N, d):
    X, y = synth(N, d)
    m = bohip.ElasticGPE(d, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.enable_timing(True)
    m.append_(X.T, y); m.fit_()
    return m, X, y, dict(m.timing())

def timed(fn, reps):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps

def score_cfg(name, N, d, R, reps=10):
    m, X, y, fit = model_for(N, d)
    Xs = np.asfortranarray(np.random.default_rng(1).random((d, R)))
    tau = float(y.max())
    t = timed(lambda: m.score("EI", [tau], Xs, want_scores=False), reps)
    st = dict(m.timing())
    flops = R * (N * N + 2.0 * N)
    tg = sum(v for k, v in m.timing() if k == "trigemm_sq")
    out = dict(config=name, N=N, d=d, R=R, host_call_ms=t * 1e3, candidates_per_s_host_buffers=R / t, stage_ms_last_call=st,
               trigemm_ms=tg, trigemm_tflops=flops / (tg * 1e-3) / 1e12, model_update_ms=fit,
               **({"cholesky_gflops": (N ** 3 / 3) / (fit["cholesky"] * 1e-3) / 1e9} if "cholesky" in fit else
                  {"factor_and_inverse_gflops": (2 * N ** 3 / 3) / (fit["cholesky+inverse"] * 1e-3) / 1e9}),
               build_cov_GBps=(8.0 * N * (N + 1) / 2) / (fit["build_cov"] * 1e-3) / 1e9)
    print(json.dumps(out)); sys.stdout.flush()

def thompson_cfg(N=3000, d=8, R=65536, S=1024):
    m, X, y, fit = model_for(N, d)
    Xs = np.asfortranarray(np.random.default_rng(2).random((d, R)))
    t = timed(lambda: m.thompson(Xs, S, seed=7), 3)
    st = dict(m.timing())
    print(json.dumps(dict(config="c5-thompson-1gpu", N=N, d=d, R=R, S=S, host_call_ms=t * 1e3, draws_per_s=S * R / t,
                          stage_ms_last_call=st, thompson_kernel_draws_per_s=S * R / (st["thompson"] * 1e-3)))); sys.stdout.flush()

def append_cfg(N=3000, d=8):
    m, X, y, fit = model_for(N - 64, d)
    rng = np.random.default_rng(5)
    ts = []
    for p in (1, 1, 1, 5, 5, 1):
        xn = rng.random((d, p)); yn = rng.standard_normal(p)
        t0 = time.perf_counter(); m.append_(xn, yn); ts.append(((time.perf_counter() - t0) * 1e3, p, dict(m.timing())))
    print(json.dumps(dict(config="append", N=N, appends=[dict(ms=a, p=b, stages=c) for a, b, c in ts], full_refit_ms=fit)))

which = sys.argv[1:] or ["c2", "c3", "c4", "c5", "append"]
if "c2" in which: score_cfg("c2", 3000, 8, 4096)
if "c3" in which: score_cfg("c3-1gpu", 3000, 8, 32768, reps=4)
if "c4" in which: score_cfg("c4", 10000, 16, 4096, reps=4)
if "c5" in which: thompson_cfg()
if "append" in which: append_cfg()
