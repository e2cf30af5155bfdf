"""rocprofv3 kernel-trace timeline of ONE refit at N=10000: how much of the wall time is the diagonal chain, what overlaps.
Run:  rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -o rt -- python tools/refit_trace.py run ; python tools/refit_trace.py analyse /tmp/rt"""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np, bohip
    N, d = int(os.environ.get("N", 10000)), 16
    rng = np.random.default_rng(0)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    m.set_params_(logNoise=-2.0); m.fit_()
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bohip::", ""), r.get("Queue_Id", "?")) for r in rows]
    ks.sort()
    # last refit = from the last k_build_cov on
    i0 = max(i for i, k in enumerate(ks) if "k_build_cov" in k[2])
    ks = ks[i0:]
    t0 = ks[0][0]; t1 = max(k[1] for k in ks)
    print(f"refit span {(t1 - t0) / 1e6:.3f} ms, {len(ks)} kernels")
    by = {}
    for s, e, n, q in ks:
        key = (n[:28], q)
        by.setdefault(key, [0, 0.0]); by[key][0] += 1; by[key][1] += (e - s) / 1e3
    for (n, q), (c, us) in sorted(by.items(), key=lambda x: -x[1][1]):
        print(f"  {n:28s} queue {q:>3}  n={c:4d}  total {us / 1e3:8.3f} ms  avg {us / c:7.1f} us")
    # chain = potf2 kernels: gaps between the end of one potf2 and the start of the next
    pf = [(s, e) for s, e, n, q in ks if "potf2" in n]
    gaps = [(pf[i + 1][0] - pf[i][1]) / 1e3 for i in range(len(pf) - 1)]
    import statistics
    print(f"potf2: n={len(pf)}, mean duration {statistics.mean((e - s) / 1e3 for s, e in pf):.1f} us, gap to next potf2: mean {statistics.mean(gaps):.1f} us, median {statistics.median(gaps):.1f}, max {max(gaps):.1f}")
    print("gaps by position (us):", [round(g) for g in gaps[:12]], "...", [round(g) for g in gaps[-8:]])
    # what the main queue did between potf2 #3 and #5 (one block boundary inside)
    qmain = [k for k in ks if "potf2" in k[2]][0][3]
    a, b = pf[3][0], pf[5][1]
    for s_, e_, n, q in ks:
        if s_ >= a and e_ <= b + 1:
            print(f"   {'main' if q == qmain else 'side'}  {n[:14]:14s} start {(s_ - a) / 1e3:8.1f} us  dur {(e_ - s_) / 1e3:7.1f} us")
