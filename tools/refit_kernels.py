"""Kernel-trace timeline of ONE refit (rocprofv3 --kernel-trace): start/end of every kernel of the last factorisation.
Run:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rk -o rk -- python $REPO/tools/refit_kernels.py run 3000 ;
      python $REPO/tools/refit_kernels.py show /tmp/rk 60"""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np, bohip
    N, d = int(sys.argv[2]), 8
    rng = np.random.default_rng(0)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    for _ in range(3):
        m.set_params_(logNoise=-2.0); m.fit_()
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bohip::", "").replace("void ", ""), r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?"))) for r in rows)
    i0 = max(i for i, k in enumerate(ks) if "k_build_cov" in k[2])
    ks = ks[i0:]
    t0 = ks[0][0]
    for s, e, n, q, g in ks[:int(sys.argv[3])]:
        print(f"  q{q:>3} {n[:22]:22s} grid {g:>8} start {(s - t0) / 1e3:8.1f} us  end {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}")
