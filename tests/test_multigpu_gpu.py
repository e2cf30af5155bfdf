"""GPU tests of the multi-GPU entry points of the C ABI (include/bohip.h: bohip_mgp_*, bohip_gp_comm_*,
bohip_gp_*_sharded).  A gpurun box has ONE MI355X, so the G-shard partition and the RCCL exchange are exercised with a
G = 1 communicator and G logical shards on that device (SURVEY.md 8e caveat): the records still travel through
ncclAllGather and k_reduce_records, and every result must equal the one-handle result bit for bit."""
import ctypes as C
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synth
from test_parity_gpu import bohip, make_model  # noqa: F401  (fixture + helper)

pytestmark = pytest.mark.gpu


def make_multi(bohip, X, y, ll, spd, lsig=0.0, lnoise=-2.0, beta=0.0, devices=(0,)):
    m = bohip.MultiGPE(X.shape[1], devices=devices, shards_per_device=spd, mean=bohip.MeanConst(beta),
                       kernel=bohip.SEArd(ll, lsig), logNoise=lnoise, capacity=len(y))
    m.append_(X.T, y)
    return m


@pytest.mark.parametrize("N,d,R,spd", [(300, 3, 1000, 8), (900, 5, 4099, 8), (257, 2, 5, 8), (500, 4, 64, 3), (1500, 6, 700, 4)])
def test_sharded_score_is_bit_identical_to_one_handle(bohip, N, d, R, spd):
    from bohip import _lib

    X, y, Xs = synth(N, d, R, seed=N + R)
    ll = np.linspace(-0.8, -0.3, d)
    one = make_model(bohip, X, y, ll, 0.1, -2.0, 0.05)
    mg = make_multi(bohip, X, y, ll, spd, 0.1, -2.0, 0.05)
    assert mg.n_shards == spd and mg.info(_lib.MGP_INFO_RCCL_VERSION) > 20000
    np.testing.assert_array_equal(mg.replica_factor(0), one.factor())
    tau = float(y.max())
    for acq, p in [("EI", [tau]), ("UCB", [2.0]), ("MaxMean", [])]:
        sc1, bv1, bi1 = one.score(acq, p, Xs.T)
        n0 = mg.info(_lib.MGP_INFO_EXCHANGES)
        scg, bvg, big = mg.score(acq, p, Xs.T)
        assert mg.info(_lib.MGP_INFO_EXCHANGES) == n0 + 1               # exactly ONE all-gather per call
        np.testing.assert_array_equal(scg, sc1)                         # R < G, R % G != 0 included
        assert (bvg, big) == (bv1, bi1)
    # resident candidates: same winner, no host buffers in the call
    mg.set_candidates(Xs.T)
    assert mg.score_resident("EI", [tau]) == one.score("EI", [tau], Xs.T)[1:]


def test_direct_l_on_the_replicated_model_equals_one_handle(bohip):
    """:GN_DIRECT_L on the device-list model runs on its first replica in one call: same search as on a single handle, through acquire_max too"""
    from bohip.acquisition import acquire_max, UpperConfidenceBound

    X, y, _ = synth(400, 3, 1, seed=77)
    ll = np.linspace(-0.8, -0.3, 3)
    one = make_model(bohip, X, y, ll, 0.1, -2.0, 0.05)
    mg = make_multi(bohip, X, y, ll, 4, 0.1, -2.0, 0.05)
    lb, ub = np.zeros(3), np.ones(3)
    for acq, p in [("UCB", [2.0]), ("EI", [float(np.median(y))])]:
        f1, x1, e1, c1 = one.direct_max(acq, p, lb, ub, 500)
        f2, x2, e2, c2 = mg.direct_max(acq, p, lb, ub, 500)
        assert (f1, e1, c1) == (f2, e2, c2) and np.array_equal(x1, x2)
    d1 = one.direct_max("ThompsonDraw", None, lb, ub, 300, seed=9)
    d2 = mg.direct_max("ThompsonDraw", None, lb, ub, 300, seed=9)
    assert d1[0] == d2[0] and np.array_equal(d1[1], d2[1])
    a = UpperConfidenceBound(beta_t=2.0)
    fa, xa = acquire_max(a, mg, lb, ub, dict(method="GN_DIRECT_L", restarts=1, maxeval=500), setparams=False)
    fb, xb = acquire_max(a, one, lb, ub, dict(method="GN_DIRECT_L", restarts=1, maxeval=500), setparams=False)
    assert fa == fb and np.array_equal(xa, xb)


def test_sharded_ties_and_nan_follow_the_reference_rule(bohip):
    """first maximum wins (src/acquisition.jl:62) across shard boundaries; NaN / -Inf records never win"""
    X, y, _ = synth(64, 2, 1, seed=5)
    ll = np.array([-0.5, -0.5])
    one = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    mg = make_multi(bohip, X, y, ll, 8)
    base = np.random.default_rng(0).random((16, 2))
    Xs = np.concatenate([base] * 8)                                     # every shard holds the same 16 candidates
    sc, bv, bi = mg.score("UCB", [1.5], Xs.T)
    assert bi == int(np.argmax(sc)) < 16 and (bv, bi) == one.score("UCB", [1.5], Xs.T)[1:]
    sc, bv, bi = mg.score("EI", [float("nan")], Xs.T)                  # every score NaN -> nothing beats -Inf
    assert np.all(np.isnan(sc)) and bv == -math.inf and bi == -1


def test_sharded_thompson_matches_one_handle(bohip):
    X, y, Xs = synth(400, 4, 3001, seed=12)
    ll = np.full(4, -0.5)
    one = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    mg = make_multi(bohip, X, y, ll, 8)
    bv1, bi1 = one.thompson(Xs.T, 64, seed=9)
    bvg, big = mg.thompson(Xs.T, 64, seed=9)
    np.testing.assert_array_equal(big, bi1)
    np.testing.assert_array_equal(bvg, bv1)


def test_sharded_acquire_max_matches_one_handle(bohip):
    X, y, _ = synth(300, 3, 1, seed=21)
    ll = np.full(3, -0.6)
    one = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    mg = make_multi(bohip, X, y, ll, 4)
    starts = np.random.default_rng(3).random((3, 24))
    lb, ub = np.zeros(3), np.ones(3)
    f1, X1, bf1, bi1, bx1, _ = one.ascend("EI", [float(y.max())], lb, ub, starts, maxeval=400)
    fg, Xg, bfg, big, bxg, _ = mg.ascend("EI", [float(y.max())], lb, ub, starts, maxeval=400)
    np.testing.assert_allclose(fg, f1, rtol=1e-9, atol=1e-14)           # every start converges on its own: same optimum
    assert big == bi1 and bfg == pytest.approx(bf1, rel=1e-9)
    np.testing.assert_allclose(bxg, bx1, atol=1e-6)
    np.testing.assert_array_equal(bxg, Xg[:, big])


def test_communicator_on_a_handle_world_size_one(bohip):
    """one-process-per-GPU form with a 1-rank communicator: ncclCommInitRank, score_sharded_dev into pinned host memory"""
    import torch

    X, y, Xs = synth(700, 5, 2048, seed=31)
    ll = np.full(5, -0.5)
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    tau = float(y.max())
    sc, bv, bi = m.score("EI", [tau], Xs.T)
    m.comm_init(bohip.comm_unique_id(), 0, 1)
    with pytest.raises(bohip.BohipError):
        m.comm_init(bohip.comm_unique_id(), 0, 1)                       # one communicator per handle
    dXs = torch.from_numpy(np.ascontiguousarray(Xs)).to("cuda:0")
    h_best = torch.zeros(2, dtype=torch.int64).pin_memory()
    torch.cuda.synchronize()
    off = 5000                                                          # this "rank" holds columns [5000, 7048) of 10000
    m.score_sharded_dev("EI", [tau], dXs.data_ptr(), 2048, off, 10000, h_best.data_ptr())
    m.synchronize()
    hb = h_best.numpy()
    assert float(hb[:1].view(np.float64)[0]) == bv and int(hb[1]) == bi + off
    bvt, bit = m.thompson_sharded(Xs.T, 32, 4, off, 10000)
    bv1, bi1 = m.thompson(Xs.T, 32, seed=4, j0=off)
    np.testing.assert_array_equal(bit, bi1 + off)
    np.testing.assert_array_equal(bvt, bv1)
    m.comm_destroy()
    with pytest.raises(bohip.BohipError):
        m.score_sharded_dev("EI", [tau], dXs.data_ptr(), 2048, 0, 2048, h_best.data_ptr())


def test_mgp_error_codes(bohip):
    from bohip import _lib

    lib = _lib.load()
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.bohip_mgp_create(2, 10, 0, devs, 2, 1, C.byref(h)) == _lib.E_ARG        # duplicate ordinal
    assert b"twice" in lib.bohip_last_error()
    assert lib.bohip_mgp_create(2, 10, 0, devs, 0, 1, C.byref(h)) == _lib.E_ARG
    assert lib.bohip_mgp_create(2, 10, 0, devs, 1, 0, C.byref(h)) == _lib.E_ARG
    bad = (C.c_int * 1)(63)
    assert lib.bohip_mgp_create(2, 10, 0, bad, 1, 1, C.byref(h)) == _lib.E_ARG         # ordinal out of range
    mg = bohip.MultiGPE(2, devices=[0], shards_per_device=2)
    with pytest.raises(bohip.BohipError) as e:
        mg.score("EI", [0.0], np.zeros((2, 4)))                                        # no observations yet
    assert e.value.code == _lib.E_STATE


def _run_bench(cmd, env=None):
    out = subprocess.run(cmd, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_launches_the_way_the_driver_does(bohip):
    """`python bench.py --gpus N` must run by itself (N = 1: one handle; N > 1: in-library multi-GPU path -- here with
    logical shards, the box has one GPU), and under torch.distributed.run the exchange is the in-library RCCL one."""
    sys.path.insert(0, ROOT)
    import bench

    one = _run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                      "--no-cpu-baseline"])
    assert one["n_gpus"] == 1 and one["value_host_buffers"] > 0 and one["host_buffers_same_winner"] is True
    two = _run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                      "--no-cpu-baseline"], env={"BOHIP_LOGICAL_SHARDS": "1"})
    # logical shards are a TEST mode and the line says so: n_gpus counts devices, the shard count has its own field
    assert two["n_gpus"] == 1 and two["logical_shards"] == 2 and "test_mode" in two
    assert two["config"]["R_total"] == 8192 and "in-library RCCL" in two["config"]["parallelism"]
    assert "no collective" in one["config"]["parallelism"] and one["roofline"]["step_over_kernel"] >= 1.0
    assert 0 < two["roofline"]["frac"] < 1
    tr = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                     "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
                     "--warmup", "1", "--no-cpu-baseline"])
    assert "bohip_gp_score_sharded_dev" in tr["config"]["parallelism"] and tr["best"] == one["best"]
    strong = _run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--strong",
                         "--no-cpu-baseline", "--no-c4"])
    assert strong["scaling"] == "strong" and strong["config"]["R_total"] == 32768 and strong["roofline"]["launches_per_step"] == 4   # 4 x 8192: equal chunks of about 256 MB, counted by the library
    assert "cholesky_c4" in one and one["cholesky_c4"]["N"] == 10000
    X, y = bench.synth(0)
    ll = np.full(bench.DIM, np.log(0.5))
    m = bohip.ElasticGPE(bench.DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=bench.N_OBS)
    m.append_(X.T, y)
    _, bv, bi = m.score("EI", [float(y.max())], bench.lhs(8192, seed=1).T)
    assert two["best"] == {"value": bv, "index": bi}
    _, bv, bi = m.score("EI", [float(y.max())], bench.lhs(4096, seed=1).T)
    assert one["best"] == {"value": bv, "index": bi}


def test_worker_threads_and_verified_exchange_on_one_device(bohip):
    """The per-device worker threads are on by default only with more than one device, so the one-GPU tests above run the inline
    branch of mgp_for_each.  BOHIP_MGP_THREADS=1 forces the threaded branch (one worker for the one device, 4 logical shards);
    BOHIP_MGP_VERIFY=1 keeps the cross-copy check of the exchange on.  Same bits as one handle."""
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, bohip
rng = np.random.default_rng(3)
N, d, R = 700, 4, 3001
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N); Xs = rng.random((R, d))
ll = np.full(d, -0.5)
one = bohip.ElasticGPE(d, kernel=bohip.SEArd(ll, 0.1), logNoise=-2.0, capacity=N); one.append_(X.T, y)
mg = bohip.MultiGPE(d, devices=(0,), shards_per_device=4, kernel=bohip.SEArd(ll, 0.1), logNoise=-2.0, capacity=N); mg.append_(X.T, y)
tau = float(y.max())
for acq, p in (("EI", [tau]), ("UCB", [2.0])):
    s1, v1, i1 = one.score(acq, p, Xs.T); sg, vg, ig = mg.score(acq, p, Xs.T)
    assert np.array_equal(s1, sg) and (v1, i1) == (vg, ig), acq
v, i = one.thompson(Xs.T, 64, seed=9); vg, ig = mg.thompson(Xs.T, 64, seed=9)
assert np.array_equal(v, vg) and np.array_equal(i, ig)
mg.set_maxtime(5.0)
f, Xa, bv, bi, bx, ev = mg.ascend("EI", [tau], np.zeros(d), np.ones(d), Xs[:24].T, maxeval=200)
f1, X1, bv1, bi1, bx1, ev1 = one.ascend("EI", [tau], np.zeros(d), np.ones(d), Xs[:24].T, maxeval=200)
assert bi == bi1 and abs(bv - bv1) <= 1e-9 * max(1.0, abs(bv1))
print("THREADS-OK")
""" % ROOT
    o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BOHIP_MGP_THREADS="1", BOHIP_MGP_VERIFY="1"), capture_output=True,
                       text=True, timeout=600)
    assert o.returncode == 0 and "THREADS-OK" in o.stdout, o.stderr[-3000:]


def test_two_real_devices_if_present(bohip):
    """devices = [0, 1]: worker thread per device, ncclCommInitAll over two ranks, the pinned result block both devices write.
    SKIPPED on a one-GPU box (every gpurun box): until a node with more than one GPU has run this, the n_devices > 1 path of
    csrc/multigpu.hip is UNVERIFIED on hardware (DESIGN.md section 9 says so)."""
    from bohip import _lib

    if _lib.load().bohip_device_count() < 2:
        pytest.skip("one GPU visible: the n_devices > 1 path stays unverified on hardware")
    X, y, Xs = synth(1200, 5, 4099, seed=4)
    ll = np.linspace(-0.8, -0.3, 5)
    one = make_model(bohip, X, y, ll, 0.1, -2.0, 0.05)
    mg = make_multi(bohip, X, y, ll, 1, 0.1, -2.0, 0.05, devices=(0, 1))
    tau = float(y.max())
    for acq, p in [("EI", [tau]), ("UCB", [2.0])]:
        sc1, bv1, bi1 = one.score(acq, p, Xs.T)
        scg, bvg, big = mg.score(acq, p, Xs.T)
        np.testing.assert_array_equal(scg, sc1)
        assert (bvg, big) == (bv1, bi1)
    mg.set_candidates(Xs.T)
    assert mg.score_resident("EI", [tau]) == one.score("EI", [tau], Xs.T)[1:]


_TWO_RANK_CODE = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np
import bohip
from bohip import _lib
from conftest_free import synth
rank, idfile, outfile = int(sys.argv[1]), sys.argv[2], sys.argv[3]
res = {"rank": rank}
try:
    if rank == 0:
        uid = bytes(bohip.comm_unique_id())
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120: raise SystemExit("no unique id from rank 0")
            time.sleep(0.02)
        uid = open(idfile, "rb").read()
    X, y, Xs = synth(500, 4, 2048, seed=9)
    ll = np.full(4, -0.6)
    m = bohip.ElasticGPE(4, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=len(y))
    m.append_(X.T, y)
    tau = float(y.max())
    t0 = time.time()
    try:
        m.comm_init(uid, rank, 2)            # ncclCommInitRank with nranks = 2: collective, both processes are in it now
        res["init"] = "ok"
    except bohip.BohipError as e:
        res["init"] = "error"; res["code"] = e.code; res["text"] = str(e)
    res["init_s"] = time.time() - t0
    if res["init"] == "ok":
        import torch
        lo, hi = (0, 1024) if rank == 0 else (1024, 2048)
        dXs = torch.from_numpy(np.ascontiguousarray(Xs[lo:hi])).cuda()
        best = torch.zeros(2, dtype=torch.int64).pin_memory()
        t_ex = []
        for it in range(6):
            t1 = time.perf_counter()
            m.score_sharded_dev("EI", [tau], dXs.data_ptr(), hi - lo, lo, 2048, best.data_ptr())
            m.synchronize()
            t_ex.append(time.perf_counter() - t1)
        res["best"] = [float(best.numpy()[:1].view(np.float64)[0]), int(best.numpy()[1])]
        res["call_ms"] = 1e3 * min(t_ex)
        _, bv, bi = m.score("EI", [tau], Xs.T)
        res["single"] = [float(bv), int(bi)]
        m.comm_destroy()
except BaseException as e:      # noqa: BLE001
    res["exception"] = repr(e)
json.dump(res, open(outfile, "w"))
'''


def test_two_processes_one_device_through_rccl_itself(bohip, tmp_path):
    """bohip_comm_unique_id + bohip_gp_comm_init with TWO ranks (two processes, both on GPU 0): the one-process-per-GPU exchange
    of libbohip driven through RCCL's own bootstrap with nranks = 2 -- the closest a one-GPU box gets to the 8-GPU launch.
    RCCL may refuse two ranks on one device ("Duplicate GPU detected"): then BOTH ranks must come back with BOHIP_E_COMM and
    RCCL's own error text within the time limit (no hang, no crash), which still proves that the id exchange, ncclCommInitRank and
    the error path work across processes.  If this RCCL build accepts it, the in-library all-gather + reduce run for real and
    every rank must hold the winner of the unsharded call."""
    helper = tmp_path / "conftest_free.py"
    helper.write_text("import sys\nsys.path.insert(0, %r)\nfrom conftest import synth\n" % os.path.join(ROOT, "tests"))
    code = tmp_path / "rank.py"
    code.write_text(_TWO_RANK_CODE % ROOT)
    idfile = str(tmp_path / "uid.bin")
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""), NCCL_DEBUG="WARN",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(code), str(r), idfile, str(tmp_path / f"out{r}.json")], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank RCCL initialisation did not return within 300 s")
    res = [json.load(open(tmp_path / f"out{r}.json")) for r in (0, 1)]
    for r in res:
        assert "exception" not in r, (r, outs)
    if all(r["init"] == "ok" for r in res):
        for r in res:
            assert r["best"] == r["single"], r                                   # both ranks hold the unsharded winner, bit for bit
        print(f"RCCL accepted two ranks on one device: sharded call {res[0]['call_ms']:.3f} / {res[1]['call_ms']:.3f} ms")
    else:
        from bohip import _lib
        for r in res:
            assert r["init"] == "error" and r["code"] == _lib.E_COMM, (r, outs)
            assert "CommInitRank" in r["text"], r                                # the failing call and RCCL's own error string
        print("RCCL refuses two ranks on one device:", res[0]["text"][:200])
