#!/usr/bin/env python
"""bench.py -- acquisition-candidates/sec of the fused GP-posterior + ExpectedImprovement + arg-max
hot path on MI355X (BASELINE.json configs[1]: N=3000 observations, d=8, SEArd, EI, R=4096 restarts
per GPU; at --gpus G the candidate set is R*G sharded over G ranks = configs[2] at G=8).

One "step" = one pass of the hot path over one batch of candidates already resident in HBM:
k_kstar (cross-covariances) -> k_trigemm_sq (V = L^-1 K*, sum v^2, mu) -> k_score (sigma^2, EI,
block arg-max) -> k_argmax_final, then (G > 1) ONE RCCL all-gather of the 16-byte (value, GLOBAL index)
record per GPU and the same reduction kernel on every GPU -- both inside libbohip, on the handle's stream;
the 16-byte result lands in pinned host memory and is read after the stream synchronisation.

Launch:  python bench.py [--gpus N --steps K --warmup W]
             N = 1: one handle.  N > 1 without a launcher: ONE process drives N devices through the
             in-library multi-GPU entry points (bohip_mgp_*: ncclCommInitAll, worker thread per device).
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
             one process per GPU; rank 0 makes an ncclUniqueId, torch.distributed only carries it (and the
             barrier / max-over-ranks of the clock); the exchange itself is bohip_gp_score_sharded_dev.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_OBS, DIM, R_PER_GPU = 3000, 8, 4096
SPIN_S = float(os.environ.get("BOHIP_BENCH_SPIN_S", "4"))   # untimed repetition of the step for an outside activity sampler (reported as untimed_spin_s)
STRONG_R_TOTAL = 32768   # --strong: BASELINE configs[2] (R = 32768 restarts in total) at every --gpus
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix = vector peak (256 CU x 4 SIMD x 2.4 GHz x 32 FLOP/clk); the
#                          MICROARCH guide lists no FP64 row; tools/ubench_fp64b.hip measures 73 TF/s sustained.


def synth(seed=0):
    """BASELINE.md synthetic inputs: X~U[0,1]^{N x d}, y = sum sin(3x) + 0.1 N(0,1), ll=log .5, ls=0, logNoise=-2."""
    rng = np.random.default_rng(seed)
    X = rng.random((N_OBS, DIM))
    y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N_OBS)
    return X, y


def lhs(R, seed):
    """Latin-hypercube candidates on [0,1]^d (reference src/utils.jl:101-120 semantics)."""
    rng = np.random.default_rng(seed)
    out = np.empty((R, DIM))
    for k in range(DIM):
        col = (np.arange(R) + rng.random(R)) / R
        rng.shuffle(col)
        out[:, k] = col
    return out


def cpu_baseline(X, y, Xs, tau, budget_candidates=1536):
    """The CPU port (oracle/gp_oracle.c) of the reference's path: one candidate at a time, k* column,
    mu dot, forward substitution on L, clamp, EI, strict-'>' arg-max.  Single thread (the reference is
    single-threaded) on a bounded sample; an all-cores figure is reported beside it."""
    from oracle.oracle import COracle

    orc = COracle()
    ll = np.full(DIM, np.log(0.5))
    cK = orc.build_cK(X, ll, 0.0, -2.0)
    t0 = time.perf_counter()
    L = orc.cholesky(cK)                                   # cpu-chol: row-by-row restatement, one thread
    t_chol = time.perf_counter() - t0
    alpha = orc.alpha(L, y, 0.0)
    # what the reference itself runs for this step is LAPACK potrf (ElasticPDMats -> LinearAlgebra.cholesky!):
    # the same factorisation through SciPy's LAPACK on one thread and on all of them, beside the scalar restatement
    lapack = {}
    try:
        import scipy.linalg as sl
        from threadpoolctl import threadpool_limits

        def potrf_time(nthreads, reps):
            best = float("inf")
            with threadpool_limits(limits=nthreads):
                for _ in range(reps):
                    t0_ = time.perf_counter()
                    Ll = sl.cholesky(cK, lower=True, check_finite=False)
                    best = min(best, time.perf_counter() - t0_)
            return best, Ll

        t1t, Ll = potrf_time(1, 2)
        tall, _ = potrf_time(None, 3)
        lapack = {"lapack_potrf_1_thread_gflops": (N_OBS ** 3 / 3.0) / t1t / 1e9,
                  "lapack_potrf_all_threads_gflops": (N_OBS ** 3 / 3.0) / tall / 1e9,
                  "lapack_sample": f"scipy.linalg.cholesky (LAPACK dpotrf), N={N_OBS}: {t1t * 1e3:.0f} ms on 1 thread, {tall * 1e3:.0f} ms on all "
                                   f"threads ({os.cpu_count()} logical cores); max |L - L_port| = {float(np.abs(Ll - L).max()):.1e}"}
    except Exception as e:   # noqa: BLE001  (reported, never fatal: the port's figure stays)
        lapack = {"lapack_sample": f"unavailable: {e}"}
    sample = Xs[:budget_candidates]
    t0 = time.perf_counter()
    orc.score(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], sample, nthreads=1)
    t1 = time.perf_counter() - t0
    ncores = min(orc.max_threads(), os.cpu_count() or 1)
    big = Xs[: min(len(Xs), max(budget_candidates, 16 * ncores))]
    t0 = time.perf_counter()
    orc.score(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], big, nthreads=ncores)
    tn = time.perf_counter() - t0
    gsample = Xs[:192]
    t0 = time.perf_counter()
    orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], gsample)   # cpu-ref-grad-1t: value + analytic gradient
    tg = time.perf_counter() - t0
    search = default_search_on_oracle(orc, X, y, ll, L, alpha, tau)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": len(sample) / t1, "unit": "candidates/s", "cores": 1, "kind": "port",
        "sample": f"first {len(sample)} of the {len(Xs)} candidates of this workload, oracle/gp_oracle.c "
                  f"(restatement of the reference path; the Julia package cannot run here), {t1:.1f} s",
        "allcores": {"value": len(big) / tn, "cores": ncores, "sample": f"{len(big)} candidates, {tn:.1f} s"},
        "with_gradient": {"value": len(gsample) / tg, "cores": 1, "sample": f"{len(gsample)} candidates, {tg:.1f} s "
                          "(the reference's default :LD_LBFGS path evaluates value and gradient)"},
        "default_search": search,
        "cholesky": {"gflops": (N_OBS ** 3 / 3.0) / t_chol / 1e9, "cores": 1, "sample": f"N={N_OBS}, {t_chol:.1f} s (row-by-row scalar port)", **lapack},
        "cpu_model": cpu_model, "host_cores": os.cpu_count(),
    }


BETA_T = 10.152008469453344   # BrochuBetaScaling(0.1) at N=3000, d=8 (src/acquisitionfunctions.jl:91-95)


def default_search_on_oracle(orc, X, y, ll, L, alpha, tau):
    """cpu_baseline leg: what the reference's DEFAULT search (src/acquisition.jl:4-6: :LD_LBFGS, 10 restarts, maxeval 2000) costs on the CPU
    restatement -- SciPy's L-BFGS-B (bounds, ftol = gtol = 1e-10) maximising the oracle's value + analytic gradient from the same ten
    Latin-hypercube starts `default_usage` / `default_usage_ei` use on the device: evaluations per start, the best end value, seconds."""
    try:
        from scipy.optimize import minimize
    except Exception as e:      # noqa: BLE001
        return {"note": f"SciPy unavailable: {e}"}
    starts = lhs(10, seed=7)
    out = {}
    for acq, prm in (("UCB", [BETA_T]), ("EI", [tau])):
        nf, fs = [], []
        t0 = time.perf_counter()
        for r in range(len(starts)):
            cnt = [0]

            def negfg(x):
                cnt[0] += 1
                sc, g = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, acq, prm, x[None, :].copy())
                return -float(sc[0]), -g[0]

            res = minimize(negfg, starts[r], jac=True, method="L-BFGS-B", bounds=[(0.0, 1.0)] * DIM,
                           options=dict(maxiter=2000, maxfun=2000, ftol=1e-10, gtol=1e-10))
            nf.append(cnt[0])
            fs.append(-float(res.fun))
        out[acq] = {"evaluations_per_start": nf, "max_evaluations_per_start": int(max(nf)), "evaluations": int(sum(nf)),
                    "best_value": float(max(fs)), "seconds": time.perf_counter() - t0, "cores": 1}
    out["note"] = ("SciPy L-BFGS-B on oracle/gp_oracle.c, one start after the other (as the reference's loop does); the device runs all ten "
                   "starts in lock-free passes: compare max_evaluations_per_start with default_usage*.evaluations")
    return out


def traffic_child():
    """`bench.py --traffic-child` (run by measure_traffic under `rocprofv3 --pmc ...`): the headline model and a few steps of the
    headline call, nothing else -- no torch, no CPU baseline, no output."""
    import bohip

    X, y = synth(0)
    m = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(np.full(DIM, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N_OBS)
    m.append_(X.T, y)
    Xs = lhs(R_PER_GPU, seed=1)
    for _ in range(6):
        m.score("EI", [float(y.max())], Xs.T, want_scores=False)


def measure_traffic():
    """roofline.traffic measured IN THIS RUN: two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass,
    MI355X_MICROARCH.md) over a child process that repeats the headline call; per-launch average of k_trigemm_sq, FETCH_SIZE x 2
    (the guide's gfx950 correction), both counters in KiB.  Returns (bytes per launch or None, source text)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None or os.environ.get("BOHIP_BENCH_TRAFFIC") == "0":
        return None, "rocprofv3 not on PATH" if exe is None else "disabled (BOHIP_BENCH_TRAFFIC=0)"
    raw = {}
    tmp = tempfile.mkdtemp(prefix="bohip_traffic_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            env = dict(os.environ, TMPDIR="/tmp", BOHIP_BENCH_TRAFFIC="0")
            r = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
                                os.path.join(ROOT, "bench.py"), "--traffic-child"], cwd="/tmp", env=env, capture_output=True,
                               text=True, timeout=300)
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_trigemm_sq" in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"rocprofv3 --pmc {ctr} produced no k_trigemm_sq rows (rc {r.returncode}): {r.stderr[-200:]!r}"
            raw[ctr] = sum(vals[1:]) / max(1, len(vals) - 1) if len(vals) > 1 else vals[0]   # (first launch: cold caches)
    except Exception as e:      # noqa: BLE001
        return None, f"rocprofv3 pass failed: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = raw["FETCH_SIZE"] * 1024.0 * 2.0 + raw["WRITE_SIZE"] * 1024.0
    return total, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --traffic-child` (5 launches each); "
                   f"FETCH_SIZE {raw['FETCH_SIZE']:.0f} KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE {raw['WRITE_SIZE']:.0f} KiB")


_OUT_FD = None


def quiet_stdout():
    """RCCL (and anything else below us) may print banners on the C-level stdout; the contract is ONE JSON line there.
    So fd 1 is pointed at stderr for the whole run and the JSON line is written to the saved descriptor at the end."""
    global _OUT_FD
    if _OUT_FD is None:
        sys.stdout.flush()
        _OUT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if _OUT_FD is None:
        print(line, flush=True)
    else:
        os.write(_OUT_FD, (line + "\n").encode())


def r_per_gpu(args, world):
    """weak scaling (default): R = 4096 per GPU; --strong: R_total = 32768 (BASELINE configs[2]) cut into `world` shards"""
    return (STRONG_R_TOTAL // world) if args.strong else R_PER_GPU


def set_inverse_queues(on):
    """The executor form grows W = L^-1 behind the factorisation's pivot chain (one stage `cholesky+inverse`); switched off, the
    factorisation runs alone and the inverse follows as its own stage -- the way to time the Cholesky against its own flop count."""
    from bohip import _lib
    return _lib.load().bohip_debug_set_chol_inv_g(8 if on is True else int(on))   # declared in include/bohip.h (benchmarks only)


def refit_figures(model, N, reps):
    """model update, both ways: as shipped (factorisation and inverse as one stage when the executor form runs) and with the
    inverse queues off (factorisation alone -> its own TFLOP/s; the level-by-level inverse behind it)"""
    fused = median_refit_ms(model, reps)
    old = set_inverse_queues(0)
    try:
        split = median_refit_ms(model, reps)
    finally:
        set_inverse_queues(old)
    model.set_params_(logNoise=-2.0)
    model.fit_()   # the resident model is the one the shipped configuration produces (W differs in the last bits between the two)
    fl = N ** 3 / 3.0
    out = {"model_update_ms": fused, "model_update_ms_inverse_after": split,
           "cholesky_alone_ms": split.get("cholesky"), "cholesky_alone_tflops": fl / (split.get("cholesky", float("nan")) * 1e-3) / 1e12}
    if "cholesky+inverse" in fused:
        t = fused["cholesky+inverse"]
        out["factor_and_inverse_ms"] = t
        out["factor_and_inverse_tflops"] = 2.0 * fl / (t * 1e-3) / 1e12   # N^3/3 each
        out["factor_and_inverse_frac_of_fp64_peak"] = out["factor_and_inverse_tflops"] / FP64_PEAK_TFLOPS
        out["inverse_after_ms"] = split.get("cholesky", float("nan")) + split.get("tri_inverse", float("nan"))
    return out


def median_refit_ms(model, reps=7):
    """model_update_ms: median over `reps` full refits (a single refit right after the first allocation is a poor sample)"""
    runs = []
    for _ in range(reps):
        model.set_params_(logNoise=-2.0)
        model.fit_()
        runs.append(dict(model.timing()))
    keys = runs[0].keys()
    return {k: float(np.median([r[k] for r in runs if k in r])) for k in keys}


def report(args, world, elapsed, stage_sum, info_ms, fit_ms, val, idx, X, y, Xs_all, tau, mode, extra=None, n_devices=None):
    ms_per_step = elapsed / args.steps * 1e3
    R_GPU = r_per_gpu(args, world)
    R_total = R_GPU * world
    value = R_total * args.steps / elapsed
    stage_ms = {k: v / args.steps for k, v in stage_sum.items()}
    tg_ms = stage_ms.get("trigemm_sq", float("nan"))
    # one launch covers one K*' chunk; the library says how many chunks the shard's call is cut into (equal-sized multiples of 512
    # candidates, BOHIP_INFO_SCORE_LAUNCHES) -- `launches` arrives in `extra` from the caller that holds the handle
    launches = int((extra or {}).pop("_launches", 0)) or max(1, -(-R_GPU // 4096))
    clock_mhz = int((extra or {}).pop("_clock_mhz", 0))
    spd = int((extra or {}).pop("_shards_per_device", 1))   # logical shards that share one device run one after the other
    flops_per_launch = (R_GPU / launches) * (N_OBS * N_OBS + 2.0 * N_OBS)  # triangular contraction + mu row
    tg_ms = tg_ms / launches
    achieved = flops_per_launch / (tg_ms * 1e-3) / 1e12
    traffic, traffic_source = (None, "not measured on this rank")
    if world == 1 and not args.strong:
        traffic, traffic_source = measure_traffic()
    if traffic is None:
        tpath = os.path.join(ROOT, "profiles", "traffic_trigemm_sq.json")
        why = traffic_source
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                traffic_source = f"profiles/traffic_trigemm_sq.json (earlier rocprofv3 --pmc passes of this command; NOT measured in this run: {why})"
            except Exception:
                traffic = None
    out = {
        "metric": "acquisition-candidates/sec (N=3000,d=8)", "value": value, "unit": "candidates/s",
        "n_gpus": world if n_devices is None else n_devices, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[2]: N=3000 obs, d=8, SEArd, ExpectedImprovement, R=32768 restarts in total, "
                                "sharded over the GPUs (LHS candidates resident in HBM)") if args.strong else
                               ("BASELINE configs[1]: N=3000 obs, d=8, SEArd, ExpectedImprovement, R=4096 "
                                "restarts per GPU (LHS candidates resident in HBM)"),
                   "N": N_OBS, "d": DIM, "R_per_gpu": R_GPU, "R_total": R_total, "acquisition": "EI",
                   "parallelism": (f"one handle on one GPU: no collective is issued ({mode})" if world == 1 else
                                   f"candidates sharded x{world}, one 16-byte RCCL all-gather + device-side reduce ({mode})")},
        "roofline": {"bound": "mfma", "kernel": "k_trigemm_sq", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_source": traffic_source,
                     "avg_launch_ms": tg_ms, "flops_per_launch": flops_per_launch, "launches_per_step": launches,
                     "step_over_kernel": ms_per_step / (tg_ms * launches * spd),
                     # MI355X clocks to its power budget (MI355X_MICROARCH.md, DVFS): `peak` is the 2.4 GHz figure; the clock the chip
                     # actually sustained under this kernel during the timed region is measured inside the kernel (every 33rd workgroup
                     # counts core-clock cycles against the 100 MHz wall clock)
                     **({"sustained_clock_mhz": clock_mhz, "peak_at_sustained_clock": FP64_PEAK_TFLOPS * clock_mhz / 2400.0,
                         "frac_of_peak_at_sustained_clock": achieved / (FP64_PEAK_TFLOPS * clock_mhz / 2400.0)} if clock_mhz > 0 else {})},
        "stage_ms": {**{k: v for k, v in info_ms.items() if k != "trigemm_sq"}, **stage_ms},
        "stage_ms_note": "trigemm_sq: HIP events around the kernel over the TIMED steps (what roofline.achieved uses; the kernel trace in "
                         "profiles/ agrees).  Every other stage: event brackets of the warm-up steps with ALL stages bracketed -- a bracket adds "
                         "the launch gap and two event records (~10-12 us) to a short kernel: `kstar` reads ~40 us here where the kernel trace "
                         "says 28 us for k_kstar itself (2.5 vs 3.6 TB/s of K*' written); the trace is the kernel, the bracket is the stage",
        "model_update_ms": fit_ms.get("model_update_ms", fit_ms),
        "cholesky": ({"N": N_OBS, "gflops": fit_ms["cholesky_alone_tflops"] * 1e3,
                      "sample": "median of 7 full refits with the executor's inverse queues off (factorisation alone)",
                      **{k: v for k, v in fit_ms.items() if k != "model_update_ms"}} if "model_update_ms" in fit_ms else
                     {"N": N_OBS, "note": "sharded run: stage times of one replica as shipped (factorisation + inverse are one stage "
                                          "when the executor form runs); the one-GPU line carries the factorisation-alone figure"}),
        "best": {"value": val, "index": idx},
    }
    if n_devices is not None and n_devices != world:
        # TEST mode (BOHIP_LOGICAL_SHARDS=1): `world` logical shards on n_devices GPU(s) -- NOT a multi-GPU measurement
        out["logical_shards"] = world
        out["test_mode"] = f"{world} logical shards on {n_devices} GPU(s): exercises the sharded path, says nothing about {world} GPUs"
    if "exchange" in info_ms:   # the all-gather of the 16-byte records + the device-side reduce, by HIP events (warm-up steps)
        out["exchange_us"] = info_ms["exchange"] * 1e3
    if extra:
        out.update(extra)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(X, y, Xs_all, tau)
    emit(json.dumps(out))


def timed(args, step, sync, barrier=None, barrier_after=True):
    """W untimed steps were done by the caller; time exactly K steps between barrier + synchronise on both sides.
    barrier_after=False: the steps themselves end in a collective, so after the last one + synchronise every rank has seen
    every other rank finish; the closing barrier would only add its own cost to the bracket."""
    if barrier:
        barrier()
    sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    sync()
    if barrier and barrier_after:
        barrier()
    return time.perf_counter() - t0, last


def main_single_process(args):
    """--gpus N inside ONE process (N = 1: one handle; N > 1: the in-library multi-GPU path, bohip_mgp_*)."""
    import torch

    import bohip
    from bohip import _lib

    lib = _lib.load()
    G = args.gpus
    ndev = lib.bohip_device_count()
    spd = 1
    devices = list(range(G))
    if G > ndev:
        if os.environ.get("BOHIP_LOGICAL_SHARDS") == "1":    # TEST mode: G logical shards on the devices that exist
            devices, spd = [0], G
        else:
            raise SystemExit(f"--gpus {G} but only {ndev} device(s) visible")
    X, y = synth(0)
    tau = float(y.max())
    R_GPU = r_per_gpu(args, G)
    R_total = R_GPU * G
    Xs_all = lhs(R_total, seed=1)
    ll = np.full(DIM, np.log(0.5))
    params = (C.c_double * 2)(tau, 0.0)
    if G == 1:
        model = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=N_OBS)
        model.enable_timing(True)
        model.append_(X.T, y)
        model.fit_()
        fit_ms = refit_figures(model, N_OBS, 7)
        dev = torch.device("cuda", 0)
        dXs = torch.from_numpy(np.ascontiguousarray(Xs_all)).to(dev)  # [R][d] = d x R column-major, resident in HBM
        # the 16-byte result record is written by the arg-max kernel straight into pinned host memory and read after
        # the stream synchronisation (no copy command)
        h_best = torch.tensor([0, -1], dtype=torch.int64).pin_memory()
        h_best_np = h_best.numpy()
        _lib.check(lib.bohip_gp_set_stream(model._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

        def step():
            _lib.check(lib.bohip_gp_score_dev(model._h, _lib.ACQ["EI"], params, C.c_void_p(dXs.data_ptr()), R_GPU, None,
                                              C.c_void_p(h_best.data_ptr())))
            _lib.check(lib.bohip_gp_synchronize(model._h))
            i = int(h_best_np[1])
            return (float(h_best_np[:1].view(np.float64)[0]), i) if i >= 0 else (-np.inf, -1)

        info_ms = {}
        for _ in range(args.warmup):
            step()
            info_ms = dict(model.timing())      # every stage bracketed by events: for the report only
        # first the same call through the host-pointer entry point (H2D of the 256 KB of candidates inside the call): a
        # reported side figure, never `value`
        model.enable_timing(0)
        Xs_host = np.ascontiguousarray(Xs_all)
        bestrec = _lib.Best()
        xp = Xs_host.ctypes.data_as(C.POINTER(C.c_double))

        def hstep():
            _lib.check(lib.bohip_gp_score(model._h, _lib.ACQ["EI"], params, xp, R_GPU, None, C.byref(bestrec)))
            return bestrec.val, bestrec.idx

        for _ in range(3):
            hstep()
        el_h, (hv, hi) = timed(args, hstep, torch.cuda.synchronize)
        # timed region: only the dominant kernel carries events (2 records per step), and they are READ after the loop
        # (mode 3): bracketing all stages costs ~25 us per step, reading the pair inside the loop another ~2 us
        model.enable_timing(3)
        step()
        model.timing(4096)
        elapsed, (val, idx) = timed(args, step, torch.cuda.synchronize)
        stage_sum = {}
        for name, ms in model.timing(4096):
            stage_sum[name] = stage_sum.get(name, 0.0) + ms
        clock_mhz = model.info(_lib.INFO_KERNEL_CLOCK_MHZ)   # core clock under k_trigemm_sq over the timed region (sampled workgroups)
        # UNTIMED: a few more seconds of the same step, so that a coarse GPU-activity sampler around this process (one sample every 5 s)
        # sees the device busy (the timed region is ~20 ms of a run whose remainder is side figures and the CPU baseline); the JSON says
        # so (untimed_spin_s).  BOHIP_BENCH_SPIN_S overrides.
        model.enable_timing(0)
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < SPIN_S:
            step()
        n_launch = C.c_int64(0)
        _lib.check(lib.bohip_gp_info(model._h, _lib.INFO_SCORE_LAUNCHES, C.byref(n_launch)))
        extra = {"_launches": int(n_launch.value), "_clock_mhz": clock_mhz, "value_host_buffers": R_total * args.steps / el_h, "ms_per_step_host_buffers": el_h / args.steps * 1e3,
                 "host_buffers_note": "same workload through bohip_gp_score: host X* in (pageable, 256 KB H2D inside the "
                                      "call), 16-byte record out; `value` is the HBM-resident rate",
                 "host_buffers_same_winner": bool(hv == val and hi == idx)}
        extra["untimed_spin_s"] = SPIN_S   # (gpu_busy of an outside sampler reflects this untimed repetition of the step, not the timed region)
        extra["default_usage"] = default_usage(model, tau)
        extra["default_usage_ei"] = default_usage(model, tau, "EI")
        extra["thompson_default"] = thompson_default(model)
        extra["thompson_c5"] = thompson_c5(model)
        # (last user of `model`: the call changes the hyper-parameters' staleness, nothing after it looks at the model)
        extra["mll_grad_ms"] = {"N=3000": mll_grad_ms(model)}      # value + gradient of the marginal likelihood (optimizemodel!)
        if not args.no_c4:
            extra["cholesky_c4"] = cholesky_c4(bohip)
            extra["mll_grad_ms"]["N=10000"] = extra["cholesky_c4"].pop("_mll_grad_ms")
        report(args, 1, elapsed, stage_sum, info_ms, fit_ms, val, idx, X, y, Xs_all, tau, "one handle", extra)
        return
    # ---- N > 1 in one process: bohip_mgp_* -------------------------------------------------------------------
    model = bohip.MultiGPE(DIM, devices=devices, shards_per_device=spd, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0),
                           logNoise=-2.0, capacity=N_OBS)
    g0 = lib.bohip_mgp_handle(model._h, 0)
    lib.bohip_gp_enable_timing(g0, 1)
    model.append_(X.T, y)
    model.fit_()
    names = (C.c_char_p * 4096)()
    msb = (C.c_double * 4096)()

    def timing0(sync=False):
        if sync:
            lib.bohip_gp_synchronize(g0)        # collects the event times of device 0
        n = lib.bohip_gp_get_timing(g0, names, msb, 4096)
        return [(names[i].decode(), msb[i]) for i in range(min(n, 4096))]

    fit_ms = dict(timing0())
    for _ in range(4):          # (best of five refits: the first one carries the allocations)
        model._push_hyper()     # marks the factor stale on every replica
        model.fit_()
        for k_, v_ in timing0():
            fit_ms[k_] = min(fit_ms.get(k_, v_), v_)
    model.set_candidates(Xs_all.T)
    best = _lib.Best()

    def step():
        _lib.check(lib.bohip_mgp_score_resident(model._h, _lib.ACQ["EI"], params, C.byref(best)))
        return best.val, best.idx

    info_ms = {}
    for _ in range(args.warmup):
        step()
        info_ms = {}
        for k, v in timing0(True):              # spd > 1: several shards on device 0 -> sum per stage
            info_ms[k] = info_ms.get(k, 0.0) + v
    lib.bohip_gp_enable_timing(g0, 3)
    step()
    timing0()
    ex0 = model.info(_lib.MGP_INFO_EXCHANGES)
    elapsed, (val, idx) = timed(args, step, torch.cuda.synchronize)
    stage_sum = {}
    for name, ms in timing0():
        stage_sum[name] = stage_sum.get(name, 0.0) + ms / spd   # per shard launch
    ex1 = model.info(_lib.MGP_INFO_EXCHANGES)
    mode = f"one process, {len(devices)} device(s) x {spd} shard(s), in-library RCCL {model.info(_lib.MGP_INFO_RCCL_VERSION)}"
    n_launch = C.c_int64(0)
    lib.bohip_gp_info(g0, _lib.INFO_SCORE_LAUNCHES, C.byref(n_launch))
    # the collective, read back from the communicator itself: a line that claims N GPUs must have exchanged over N ranks
    nranks = model.info(_lib.MGP_INFO_COMM_NRANKS)
    if len(devices) > 1 and nranks != len(devices):
        raise SystemExit(f"bench.py --gpus {G}: the in-library communicator has {nranks} ranks, expected {len(devices)}")
    extra = {"_launches": int(n_launch.value), "_shards_per_device": spd,
             "rccl": {"nranks": int(nranks), "version": int(model.info(_lib.MGP_INFO_RCCL_VERSION)),
                      "allgathers_per_step": (ex1 - ex0) / float(args.steps), "read_from": "ncclCommCount / ncclGetVersion / the library's exchange counter"}}
    if not args.strong and STRONG_R_TOTAL % G == 0:
        # BASELINE configs[2] beside the weak-scaling line: R = 32768 in total over the same devices
        Xs_s = lhs(STRONG_R_TOTAL, seed=1)
        model.set_candidates(Xs_s.T)
        for _ in range(3):
            step()
        el_s, (v_s, i_s) = timed(args, step, torch.cuda.synchronize)
        extra["configs2_strong"] = {"workload": f"BASELINE configs[2]: R = {STRONG_R_TOTAL} restarts in total over {len(devices)} device(s)",
                                    "R_total": STRONG_R_TOTAL, "ms_per_step": el_s / args.steps * 1e3, "value": STRONG_R_TOTAL * args.steps / el_s,
                                    "unit": "candidates/s", "best": {"value": v_s, "index": i_s}}
    report(args, G, elapsed, stage_sum, info_ms, fit_ms, val, idx, X, y, Xs_all, tau, mode, extra, n_devices=len(devices))


def default_usage(model, tau, acq="UCB"):
    """What the reference does BY DEFAULT (src/acquisition.jl:4-6: method :LD_LBFGS, restarts 10, maxeval 2000) on the headline
    model: acquire_max = 10 Latin-hypercube starts, each refined by a gradient-based local search.  On the device all starts
    advance on their own schedule (bohip_gp_acquire_max, free-running driver): one value + gradient pass of the model per evaluation.  Reported beside the
    headline metric; the CPU figures to hold against it are cpu_baseline.with_gradient (one candidate's value + gradient at a time) and
    cpu_baseline.default_search (SciPy's L-BFGS-B on the oracle from the same starts).
    acq = "UCB": the README's acquisition at BrochuBetaScaling's beta_t;  acq = "EI": the DEFAULT acquisition of `BOpt`
    (src/BayesianOptimization.jl:265) at tau = max y -- BASELINE configs[1]'s acquisition through the default search."""
    R = 10
    starts = np.asfortranarray(lhs(R, seed=7).T)
    lb, ub = np.zeros(DIM), np.ones(DIM)
    prm = [BETA_T] if acq == "UCB" else [tau]
    model.ascend(acq, prm, lb, ub, starts, 2000)
    runs = []
    for _ in range(5):
        t0 = time.perf_counter()
        f, Xb, bf, bi, bx, ev = model.ascend(acq, prm, lb, ub, starts, 2000)
        runs.append((time.perf_counter() - t0, ev))
    t, ev = sorted(runs)[len(runs) // 2]
    sg = []
    for _ in range(20):
        t0 = time.perf_counter()
        model.score_grad(acq, prm, starts)
        sg.append(time.perf_counter() - t0)
    name = ("UpperConfidenceBound (BrochuBetaScaling, the README's acquisition)" if acq == "UCB" else
            "ExpectedImprovement at tau = max y (the default acquisition of BOpt, src/BayesianOptimization.jl:265)")
    out = {"workload": f"acquire_max, N={N_OBS}, d={DIM}, {name}, 10 restarts, :LD_LBFGS, maxeval 2000 (the reference's defaultoptions)",
           "acquire_max_ms": t * 1e3, "evaluations": int(ev), "us_per_evaluation": t / max(ev, 1) * 1e6,
           "score_grad_call_us": float(np.median(sg)) * 1e6, "best": {"value": float(bf), "index": int(bi)},
           "end_values": [float(v) for v in f],
           "note": "an evaluation = value + gradient of all 10 starts in one pass (two kernels, K*' + V' + posterior and U' + gradient, kernels_small.hip); "
                   "compare with 10 / cpu_baseline.with_gradient.value seconds per such pass on one CPU core"}
    if acq == "UCB":
        out["small_model"] = default_usage_small()
    else:
        out["note_ei"] = ("EI at a Latin-hypercube start of this model is 1e-20 .. 1e-70 with a gradient to match: the first step of the search moves "
                          "such a start, finds no Armijo improvement above ftol_rel / xtol_abs and retires it within 2-3 evaluations -- "
                          "cpu_baseline.default_search.EI shows SciPy on the oracle leaving every start at its first")
    return out


def thompson_default(model):
    """The reference's default for ThompsonSamplingSimple (src/acquisition.jl:7-9: :GN_DIRECT_L, restarts 1, maxeval 2000, one posterior draw
    per point, src/acquisitionfunctions.jl:107-108) on the headline model.  A device model takes the search as ONE library call
    (bohip_gp_direct_max: the dividing-rectangles bookkeeping of csrc/direct_l.h, every iteration's new points in one bohip_gp_predict);
    beside it the same search driven from Python through bohip_direct_ask / _tell with predict_f as the objective (what a host-side
    objective costs), and the NumPy twin's bookkeeping the round-6 figure of 27.7 ms was made of."""
    from bohip.acquisition import ThompsonSamplingSimple, acquire_max, defaultoptions, direct_l_search, _batched_direct_l
    opts = defaultoptions(type(model), ThompsonSamplingSimple)
    lb, ub = np.zeros(DIM), np.ones(DIM)
    acquire_max(ThompsonSamplingSimple(), model, lb, ub, opts, rng=np.random.default_rng(5), setparams=False)
    runs = []
    for i in range(5):
        t0 = time.perf_counter()
        acquire_max(ThompsonSamplingSimple(), model, lb, ub, opts, rng=np.random.default_rng(6 + i), setparams=False)
        runs.append(time.perf_counter() - t0)
    t = sorted(runs)[2]
    _, _, ev, calls = model.direct_max("ThompsonDraw", None, lb, ub, opts["maxeval"], seed=11)
    gen = np.random.default_rng(3)

    def f_batch(X):
        mu, var = model.predict_f(X)
        return mu + np.sqrt(np.maximum(var, 0.0)) * gen.standard_normal(mu.size)

    def timed(fn):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn(f_batch, lb, ub, opts["maxeval"])
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[1] * 1e3
    return {"workload": f"acquire_max, N={N_OBS}, d={DIM}, ThompsonSamplingSimple, :GN_DIRECT_L, restarts 1, maxeval 2000 (the reference's defaultoptions)",
            "acquire_max_ms": t * 1e3, "device_calls": int(calls), "direct_iterations": int(calls) - 1, "evaluations": int(ev),
            "us_per_device_call": t / max(calls, 1) * 1e6,
            "ask_tell_from_python_ms": timed(direct_l_search), "numpy_twin_ms": timed(_batched_direct_l),
            "note": "acquire_max_ms: bohip_gp_direct_max, one library call (bookkeeping + one bohip_gp_predict per DIRECT iteration + the draws); "
                    "ask_tell_from_python_ms: the library's bookkeeping with predict_f called from Python per iteration; numpy_twin_ms: round 6's "
                    "route (acquisition._batched_direct_l, now the test twin)"}


def thompson_c5(model):
    """BASELINE configs[4] on ONE GPU: 1024 posterior draws x 65536 candidates (d = 8, N = 3000): bohip_gp_thompson = mu / sigma^2 of all
    candidates (k_kstar + k_trigemm_sq, 16 chunks) + k_thompson (counter-based normals, arg-max per draw, S x R never materialised)."""
    R, S = 65536, 1024
    Xs = np.asfortranarray(np.random.default_rng(2).random((DIM, R)))
    model.enable_timing(True)
    model.thompson(Xs, S, seed=7)
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        model.thompson(Xs, S, seed=7)
        runs.append((time.perf_counter() - t0, dict(model.timing())))
    model.enable_timing(0)
    t, st = sorted(runs, key=lambda r: r[0])[1]
    return {"workload": f"N={N_OBS}, d={DIM}, {S} draws x {R} candidates, one GPU (host X* in)", "host_call_ms": t * 1e3, "draws_per_s": S * R / t,
            "kernel_ms": st.get("thompson"), "kernel_draws_per_s": S * R / (st["thompson"] * 1e-3) if st.get("thompson") else None,
            "stage_ms": st}


def default_usage_small():
    """The same call on a model of the size the reference's own examples and tests build (Branin, 2-d, 200 observations): the whole
    acquire_max is ONE launch there (one workgroup per start point, kernels_ascent.hip k_ascent_wg)."""
    import bohip
    N, d, R = 200, 2, 10
    rng = np.random.default_rng(11)
    X = rng.random((N, d))
    y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    starts = np.asfortranarray(rng.random((d, R)))
    lb, ub = np.zeros(d), np.ones(d)
    m.ascend("UCB", [2.0], lb, ub, starts, 2000)
    runs = []
    for _ in range(9):
        t0 = time.perf_counter()
        f, Xb, bf, bi, bx, ev = m.ascend("UCB", [2.0], lb, ub, starts, 2000)
        runs.append((time.perf_counter() - t0, ev))
    t, ev = sorted(runs)[len(runs) // 2]
    m.close()
    return {"workload": f"acquire_max, N={N}, d={d}, UCB, 10 restarts, :LD_LBFGS", "acquire_max_ms": t * 1e3, "evaluations": int(ev),
            "us_per_evaluation": t / max(ev, 1) * 1e6}


def mll_grad_ms(model, reps=5):
    """bohip_gp_mll_grad (row N2: what every evaluation of the reference's optimizemodel! costs, src/models/gp.jl:42-77): refit +
    cK^-1 = W'W + one pass over K and dK; median of `reps` host calls, each behind a change of the hyper-parameters (the refit is inside)."""
    def once():
        model.set_params_(logNoise=-2.0)      # what an optimizer step does: new hyper-parameters -> the factor is stale -> the call refits
        t0 = time.perf_counter()
        model.mll_grad()
        return time.perf_counter() - t0
    once()
    return float(np.median([once() for _ in range(reps)])) * 1e3


def cholesky_c4(bohip):
    """BASELINE configs[3]: N=10000 obs, d=16, SEArd -- the blocked-Cholesky path at the size BASELINE.json names, beside the
    headline workload: kernel-matrix assembly (HBM-write bound), factorisation (MFMA bound), triangular inverse."""
    N, d = 10000, 16
    rng = np.random.default_rng(3)
    X = rng.random((N, d))
    y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.enable_timing(True)
    m.append_(X.T, y)
    fig = refit_figures(m, N, reps=5)
    m.enable_timing(False)
    mg = mll_grad_ms(m, reps=3)
    m.close()
    ms = fig["model_update_ms"]
    ch = fig["cholesky_alone_ms"]
    bc = ms.get("build_cov", float("nan"))
    return {"N": N, "d": d, **fig, "cholesky_tflops": (N ** 3 / 3.0) / (ch * 1e-3) / 1e12,
            "cholesky_frac_of_fp64_peak": (N ** 3 / 3.0) / (ch * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            "build_cov_gb_per_s": 8.0 * N * (N + 1) / 2 / (bc * 1e-3) / 1e9, "sample": "median of 5 full refits, each way", "_mll_grad_ms": mg}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong", action="store_true", help="strong scaling: R = 32768 candidates in total (BASELINE configs[2]) at every --gpus")
    ap.add_argument("--no-c4", action="store_true", help="skip the N=10000 model-update figure (cholesky_c4)")
    ap.add_argument("--traffic-child", action="store_true", help="(internal) the child process of measure_traffic()")
    args = ap.parse_args()
    if args.traffic_child:
        return traffic_child()
    if args.strong and STRONG_R_TOTAL % args.gpus:
        raise SystemExit("--strong needs --gpus to divide 32768")
    quiet_stdout()

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libbohip has no CPU path")
    if "WORLD_SIZE" not in os.environ and os.environ.get("BOHIP_FORCE_DIST") != "1":
        return main_single_process(args)

    # ---- one process per GPU (torch.distributed.run) ----------------------------------------------------------
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # BOHIP_SHARE_GPU=1 + BOHIP_DIST_BACKEND=gloo: a TEST mode that runs all ranks on GPU 0 (RCCL cannot place two ranks
    # on one device, so the records travel through gloo and dist.py's host reduce instead of the in-library exchange)
    share_gpu = os.environ.get("BOHIP_SHARE_GPU") == "1"
    backend = os.environ.get("BOHIP_DIST_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)

    import bohip
    from bohip import _lib
    from bohip.dist import allgather_best

    lib = _lib.load()
    X, y = synth(0)
    tau = float(y.max())
    R_GPU = r_per_gpu(args, world)
    R_total = R_GPU * world
    Xs_all = lhs(R_total, seed=1)
    lo = rank * R_GPU
    Xs_local = Xs_all[lo:lo + R_GPU]
    ll = np.full(DIM, np.log(0.5))
    model = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0,
                             capacity=N_OBS, device=local_rank)
    model.enable_timing(True)
    model.append_(X.T, y)  # every rank factors the same model redundantly (192 KB broadcast beats 36 MB of L)
    model.fit_()
    fit_ms = dict(model.timing())
    in_library = not share_gpu and os.environ.get("BOHIP_BENCH_NO_INLIB") != "1"
    if in_library:
        # every rank must take the same path: agree on whether the in-library communicator came up everywhere, else fall back
        # to the torch.distributed exchange of dist.py (same records, same reduction, one all_gather + a host reduce)
        ok = 1
        try:
            ids = [bohip.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)      # torch.distributed is only the courier of the 128 bytes
            model.comm_init(ids[0], rank, world)
        except Exception as e:                          # noqa: BLE001
            print(f"[rank {rank}] in-library RCCL unavailable ({e}); using the torch.distributed exchange", file=sys.stderr)
            ok = 0
        t_ok = torch.tensor([ok], device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        if int(t_ok.item()) == 0:
            if ok:
                model.comm_destroy()
            in_library = False
    if not in_library:
        model.set_batch_hint(R_total)

    dev = torch.device("cuda", local_rank)
    dXs = torch.from_numpy(np.ascontiguousarray(Xs_local)).to(dev)  # [R][d] = d x R column-major
    d_best = torch.tensor([0, -1, lo], dtype=torch.int64, device=dev)
    h_best = torch.tensor([0, -1], dtype=torch.int64).pin_memory()
    h_best_np = h_best.numpy()
    if os.environ.get("BOHIP_BENCH_TORCH_STREAM") == "1":   # (test mode: run the library on torch's current stream)
        stream = torch.cuda.current_stream()
        _lib.check(lib.bohip_gp_set_stream(model._h, C.c_void_p(stream.cuda_stream)))
    else:
        # the handle keeps its own high-priority stream, as in the single-process run: on torch's current (= the legacy
        # default) stream the same kernels measured 12 % slower (0.675 vs 0.600 ms for k_trigemm_sq).  The candidates were
        # uploaded on torch's stream: make them visible first.
        torch.cuda.synchronize()
    params = (C.c_double * 2)(tau, 0.0)

    def step():
        if in_library:
            model.score_sharded_dev("EI", [tau], dXs.data_ptr(), R_GPU, lo, R_total, h_best.data_ptr())
            _lib.check(lib.bohip_gp_synchronize(model._h))
            i = int(h_best_np[1])
            return (float(h_best_np[:1].view(np.float64)[0]), i) if i >= 0 else (-np.inf, -1)
        _lib.check(lib.bohip_gp_score_dev(model._h, _lib.ACQ["EI"], params, C.c_void_p(dXs.data_ptr()),
                                          R_GPU, None, C.c_void_p(d_best.data_ptr())))
        _lib.check(lib.bohip_gp_synchronize(model._h))   # the record is written on the handle's stream, the collective runs on torch's
        val, idx = allgather_best(d_best, lo, world, force_collective=True)
        return val, idx

    info_ms = {}
    dist.barrier()                  # (torch-level rendezvous; the device may idle for milliseconds here)
    for _ in range(args.warmup):
        step()
        info_ms = dict(model.timing())
    model.enable_timing(3)
    # The bracket around the K timed steps.  Every step ends in a collective (the all-gather of the records), so an untimed
    # step IS a barrier -- no rank leaves it before all have entered -- and it keeps the GPU busy; torch's dist.barrier()
    # costs ~3 ms of idle device, after which the kernels run ~12 % slower for some 20 ms (measured: 0.748 ms per step over
    # 30 steps right behind it against 0.663 over 200).  So: rendezvous first, then the W warm-up steps plus enough further
    # untimed steps to have the device at its working clocks (the single-process run does 33 host-buffer steps at this
    # point), then the bracket with a collective step as its barrier.
    for _ in range(max(0, 40 - args.warmup)):
        step()
    model.timing(4096)

    def step_barrier():
        step()
        model.timing(4096)          # (drop the barrier step's event records)

    ex0 = model.info(_lib.INFO_COMM_EXCHANGES)
    elapsed, (val, idx) = timed(args, step, torch.cuda.synchronize, step_barrier, barrier_after=False)
    ex1 = model.info(_lib.INFO_COMM_EXCHANGES)
    stage_sum = {}
    for name, ms in model.timing(4096):
        stage_sum[name] = stage_sum.get(name, 0.0) + ms
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    extra = {"_launches": model.info(_lib.INFO_SCORE_LAUNCHES)}
    if in_library:
        # the collective, read back from the communicator itself: a line that claims N GPUs must have exchanged over N ranks
        nranks = model.info(_lib.INFO_COMM_NRANKS)
        if nranks != world:
            raise SystemExit(f"[rank {rank}] bench.py --gpus {world}: the communicator has {nranks} ranks")
        extra["rccl"] = {"nranks": int(nranks), "version": int(model.info(_lib.INFO_COMM_RCCL_VERSION)),
                         "allgathers_per_step": (ex1 - ex0 - 1) / float(args.steps),   # (- 1: the collective step that serves as the opening barrier)
                         "read_from": "ncclCommCount / the handle's exchange counter"}
    if not args.strong and STRONG_R_TOTAL % world == 0:
        # BASELINE configs[2] beside the weak-scaling line: R = 32768 in total, this rank's contiguous shard
        R_s = STRONG_R_TOTAL // world
        Xs_s = lhs(STRONG_R_TOTAL, seed=1)
        lo_s = rank * R_s
        dXs_s = torch.from_numpy(np.ascontiguousarray(Xs_s[lo_s:lo_s + R_s])).to(dev)
        d_best_s = torch.tensor([0, -1, lo_s], dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        if not in_library:
            model.set_batch_hint(STRONG_R_TOTAL)

        def step_s():
            if in_library:
                model.score_sharded_dev("EI", [tau], dXs_s.data_ptr(), R_s, lo_s, STRONG_R_TOTAL, h_best.data_ptr())
                _lib.check(lib.bohip_gp_synchronize(model._h))
                i = int(h_best_np[1])
                return (float(h_best_np[:1].view(np.float64)[0]), i) if i >= 0 else (-np.inf, -1)
            _lib.check(lib.bohip_gp_score_dev(model._h, _lib.ACQ["EI"], params, C.c_void_p(dXs_s.data_ptr()), R_s, None, C.c_void_p(d_best_s.data_ptr())))
            _lib.check(lib.bohip_gp_synchronize(model._h))
            return allgather_best(d_best_s, lo_s, world, force_collective=True)

        model.enable_timing(0)
        for _ in range(5):
            step_s()
        el_s, (v_s, i_s) = timed(args, step_s, torch.cuda.synchronize, step_s, barrier_after=False)
        t = torch.tensor([el_s], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_s = float(t.item())
        extra["configs2_strong"] = {"workload": f"BASELINE configs[2]: R = {STRONG_R_TOTAL} restarts in total, sharded over {world} rank(s)",
                                    "R_total": STRONG_R_TOTAL, "ms_per_step": el_s / args.steps * 1e3, "value": STRONG_R_TOTAL * args.steps / el_s,
                                    "unit": "candidates/s", "best": {"value": v_s, "index": i_s}}
    if rank == 0:
        mode = ("one process per GPU, in-library RCCL (bohip_gp_score_sharded_dev)" if in_library
                else f"one process per GPU, TEST exchange through torch.distributed/{backend}")
        report(args, world, elapsed, stage_sum, info_ms, fit_ms, val, idx, X, y, Xs_all, tau, mode, extra)
    if in_library:
        model.comm_destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
