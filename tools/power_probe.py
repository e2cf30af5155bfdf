"""Is k_trigemm_sq clock-(power-)limited?  Same instruction stream, different operand data:
  (a) the headline workload (random LHS candidates: K*' dense),
  (b) candidates far outside the observations' box (every K*' entry underflows to exactly 0: the B operand of
      every MFMA is zero, same instruction count, same memory traffic),
  (c) (a) again (drift check).
Prints the event time of the dominant kernel for each.  A large (b) < (a) gap with identical instruction streams is the
DVFS give-back of MI355X_MICROARCH.md: the chip clocks to its power budget, and the matrix pipe's energy depends on its data.
Usage: python tools/power_probe.py [R]   (LIBV=abl/libbohip_xxx.so selects a variant build)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bohip import _lib
if os.environ.get("LIBV"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["LIBV"])
import bohip
from bench import synth, lhs, N_OBS, DIM

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X, y = synth(0)
m = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(np.full(DIM, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N_OBS)
m.append_(X.T, y)
m.fit_()
tau = float(y.max())
Xs = lhs(R, 1)
far = Xs + 1.0e3          # r^2 ~ 4e6 * d: exp(-r^2/2) == 0.0 exactly
flops = R * (N_OBS * N_OBS + 2.0 * N_OBS)


def run(name, xs, reps=60):
    m.enable_timing(True)
    for _ in range(10):
        m.score("EI", [tau], xs.T, want_scores=False)
    ts = []
    for _ in range(reps):
        m.score("EI", [tau], xs.T, want_scores=False)
        ts.append(dict(m.timing()).get("trigemm_sq", float("nan")))
    ts = np.array(ts)
    med = float(np.median(ts))
    mhz = m.info(_lib.INFO_KERNEL_CLOCK_MHZ)     # core clock under the kernel (sampled workgroups: clock64 / wall_clock64)
    print(f"{name:34s} trigemm_sq median {med * 1e3:7.1f} us  (min {ts.min() * 1e3:7.1f}, max {ts.max() * 1e3:7.1f})  "
          f"{flops / (med * 1e-3) / 1e12:5.1f} TF/s = {flops / (med * 1e-3) / 1e12 / 78.6:.3f} of 78.6;  core clock {mhz} MHz -> "
          f"{med * mhz:7.1f} kcycles per launch, {flops / (med * 1e-3) / 1e12 / (78.6 * mhz / 2400.0):.3f} of the peak at that clock", flush=True)
    return med


a = run("(a) LHS candidates (dense K*')", Xs)
b = run("(b) far candidates (K*' == 0)", far)
c = run("(c) LHS candidates again", Xs)
print(f"zero-operand speed-up: {a / b:.3f}x   (drift a->c {c / a:.3f})")
