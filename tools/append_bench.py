"""What a BO iteration's model update costs: bohip_gp_append of ONE observation to a model of N (incremental factor / inverse / alpha
update), microseconds per call (median of 60 appends, starting from a fresh fit).  BOHIP_LIB selects a build for an A/B."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
for N, d in ((200, 2), (1000, 4), (3000, 8), (10000, 16)):
    rng = np.random.default_rng(N)
    X = rng.random((N + 80, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N + 80)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N + 100)
    m.append_(X[:N].T, y[:N]); m.fit_()
    for i in range(N, N + 10): m.append_(X[i:i + 1].T, y[i:i + 1])
    ts = []
    for i in range(N + 10, N + 70):
        t0 = time.perf_counter(); m.append_(X[i:i + 1].T, y[i:i + 1]); ts.append(time.perf_counter() - t0)
    mu, var = m.predict_f(X[N + 70:N + 80].T)
    print(f"N={N} d={d}: append of one observation {np.median(ts)*1e6:7.1f} us (p10 {np.percentile(ts,10)*1e6:.1f})   check: mu[0] {mu[0]:.12g} var[0] {var[0]:.6e}", flush=True)
    m.close()
