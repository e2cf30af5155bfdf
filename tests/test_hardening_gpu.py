"""GPU tests of the failure paths of the model update (csrc/bohip.hip refit, csrc/kernels_chol.hip, csrc/kernels_exec.hip):
a dataflow factorisation that times out on a dependency must degrade into the launch-chained form -- never hang, never
return a wrong factor -- and say so through bohip_gp_info; two processes that refit on ONE GPU at the same time (neither can
count on its persistent workgroups being resident) must both get the right factor; the jitter escalation switch
(bohip_gp_set_jitter: the role of GaussianProcesses.jl's make_posdef! behind src/models/gp.jl:11,16, UPSTREAM-UNVERIFIED
and therefore off by default) against its oracle twin."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synth
from test_parity_gpu import bohip  # noqa: F401

pytestmark = pytest.mark.gpu

CODE = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np, bohip
from bohip import _lib
out = {}
for N, d in %s:
    rng = np.random.default_rng(N)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, -0.6), 0.1), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    forms = [m.info(_lib.INFO_CHOL_FORM)]
    for _ in range(%d):
        m.set_params_(logNoise=-2.0); m.fit_()
        forms.append(m.info(_lib.INFO_CHOL_FORM))
    L = m.factor()
    mu, var = m.predict_f(X[:40].T + 0.01)
    out[str(N)] = dict(Lsum=float(np.abs(L).sum()), Ldiag=np.diag(L)[::97].tolist(), Llast=L[-1, ::211].tolist(), mu=mu.tolist(),
                       var=var.tolist(), forms=forms, fallbacks=m.info(_lib.INFO_CHOL_FALLBACKS), abort_tiles=m.info(_lib.INFO_CHOL_ABORT_TILES))
    m.close()
print("RESULT" + json.dumps(out))
'''


def run(sizes, refits, **env):
    o = subprocess.run([sys.executable, "-c", CODE % (ROOT, repr(sizes), refits)], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900)
    assert o.returncode == 0, o.stderr[-3000:]
    return json.loads([l for l in o.stdout.splitlines() if l.startswith("RESULT")][-1][6:]), o.stderr


def same_factor(a, b):
    assert a["Lsum"] == pytest.approx(b["Lsum"], rel=1e-12)
    for k in ("Ldiag", "Llast", "mu"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a["var"], b["var"], rtol=1e-7, atol=1e-12)


def test_a_timed_out_dataflow_factorisation_falls_back_and_reports_it():
    """BOHIP_CHOL_SPIN_US=1 makes every in-kernel wait give up after a microsecond: the first refit of each size runs a dataflow
    form (the executor), times out, and is redone launch-chained in the same call."""
    sizes = ((1500, 3), (4300, 4))
    ref, _ = run(sizes, 0, BOHIP_CHOL_DATAFLOW="0")
    got, err = run(sizes, 1, BOHIP_CHOL_SPIN_US="1")
    assert "timed out on a dependency" in err
    first = got[str(sizes[0][0])]
    assert first["fallbacks"] == 1 and first["abort_tiles"] == (sizes[0][0] + 1 + 127) // 128
    assert first["forms"][0] == 0 and first["forms"][-1] == 0           # the form of the refit that delivered the factor: the launch chain
    assert got[str(sizes[1][0])]["fallbacks"] == 0                      # the switch is process-wide: later handles do not try again
    for N, _ in sizes:
        same_factor(got[str(N)], ref[str(N)])


def test_forms_are_reported_and_strict_mode_errors_instead_of_falling_back():
    got, _ = run(((1500, 3), (4300, 4)), 0)
    assert got["1500"]["forms"] == [4] and got["4300"]["forms"] == [4]     # the executor with its inverse queues (the default from 4 row tiles on)
    assert got["1500"]["fallbacks"] == 0 and got["4300"]["abort_tiles"] == 0
    ref = got
    got, _ = run(((1500, 3), (4300, 4)), 0, BOHIP_CHOL_INV_G="0")           # without them: the first dataflow form below 32 row tiles
    assert got["1500"]["forms"] == [1] and got["4300"]["forms"] == [4]
    for N in ("1500", "4300"):
        same_factor(got[N], ref[N])
    o = subprocess.run([sys.executable, "-c", CODE % (ROOT, "((1500, 3),)", 0)], env=dict(os.environ, BOHIP_CHOL_SPIN_US="1", BOHIP_CHOL_DF_STRICT="1"),
                       capture_output=True, text=True, timeout=600)
    assert o.returncode != 0 and "timed out on a dependency" in o.stderr


def test_two_processes_refit_on_one_gpu_at_the_same_time():
    """Both factorisations rely on persistent workgroups; with two processes on the device neither is alone on the chip.  Whatever the
    interleaving: no hang (the waits are bounded by wall clock), the right factor in both, and the counters tell what happened."""
    sizes = "((3000, 8), (4500, 8))"
    ref, _ = run(((3000, 8), (4500, 8)), 0, BOHIP_CHOL_DATAFLOW="0")
    procs = [subprocess.Popen([sys.executable, "-c", CODE % (ROOT, sizes, 6)], env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for _ in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
        got = json.loads([l for l in so.splitlines() if l.startswith("RESULT")][-1][6:])
        for N in ("3000", "4500"):
            same_factor(got[N], ref[N])
            assert got[N]["fallbacks"] in (0, 1)          # a time-out under contention is allowed; a wrong factor is not
            print(f"N={N}: forms {got[N]['forms']}, fall-backs {got[N]['fallbacks']}")


def test_jitter_escalation_switch_against_its_oracle_twin(bohip, orc):
    from bohip import _lib
    from oracle.oracle import fit_with_jitter

    # README-shaped and singular in float64: every position observed five times, sigma_f^2 = e^10, noise e^-50
    rng = np.random.default_rng(11)
    pos = rng.random((60, 2)) * 10 - 5
    X = np.repeat(pos, 5, axis=0)
    y = -(((X - 1) ** 2).sum(1) + rng.standard_normal(len(X)))
    ll, lsig, lnoise = np.zeros(2), 5.0, -25.0
    m = bohip.ElasticGPE(2, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, lsig), logNoise=lnoise, capacity=len(y) + 8)
    with pytest.raises(bohip.BohipError) as e:                       # default: the failing pivot is reported, nothing is added
        m.append_(X.T, y)
    assert e.value.code == _lib.E_NOTPD and m.info(_lib.INFO_PIVOT) > 0 and m.info(_lib.INFO_JITTER_STEPS) == 0
    with pytest.raises(np.linalg.LinAlgError):
        orc.fit(X, y, ll, lsig, lnoise, 0.0)
    m.set_jitter(1e-8, 10)
    m.fit_()
    L_o, alpha_o, tries_o, added_o = fit_with_jitter(orc, X, y, ll, lsig, lnoise, 0.0, 1e-8, 10)
    assert m.info(_lib.INFO_JITTER_STEPS) == tries_o >= 1
    cK = orc.build_cK(X, ll, lsig, lnoise) + added_o * np.eye(len(y))
    L = m.factor()
    np.testing.assert_allclose(L @ L.T, cK, rtol=0, atol=1e-9 * np.abs(cK).max())
    Xs = rng.random((32, 2)) * 10 - 5
    mu, var = m.predict_f(Xs.T)
    mu_o, var_o = orc.predict(X, ll, lsig, 0.0, L_o, alpha_o, Xs)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-5, atol=1e-5 * np.abs(mu_o).max())
    np.testing.assert_allclose(var, var_o, rtol=1e-5, atol=1e-6 * np.exp(2 * lsig))
    # an incremental append after a jittered refit carries the same jitter on its new diagonal entries: the factor stays the
    # factor of ONE matrix, cK + J I
    Xa = rng.random((3, 2)) * 10 - 5
    ya = -(((Xa - 1) ** 2).sum(1))
    n_app = m.info(_lib.INFO_APPENDS)
    m.append_(Xa.T, ya)
    assert m.info(_lib.INFO_APPENDS) == n_app + 1                    # the incremental path, not a refit
    X2 = np.vstack([X, Xa])
    cK2 = orc.build_cK(X2, ll, lsig, lnoise) + added_o * np.eye(len(X2))
    L2 = m.factor()
    np.testing.assert_allclose(L2 @ L2.T, cK2, rtol=0, atol=1e-9 * np.abs(cK2).max())
    m.set_jitter(0.0, 0)                                             # and off again
    m.set_params_(logNoise=lnoise)
    with pytest.raises(bohip.BohipError):
        m.fit_()
