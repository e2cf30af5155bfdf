"""What is the executor kernel (k_chol_exec) bound by?  rocprofv3 --pmc passes over a child that refits a model of N observations:
fabric-side bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE), L2 hits / misses, matrix-pipe busy cycles.
  python tools/chol_pmc.py [N=10000] [d=16]          (parent: runs the passes, prints per-launch averages of every kernel seen)
  python tools/chol_pmc.py --child N d                (the profiled process)
Counters go in separate passes (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE takes 3); no trace domains beside --pmc."""
import csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
          ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"], ["TCP_TCC_READ_REQ_sum"]]


def child(n, d):
    import numpy as np
    import bohip
    rng = np.random.default_rng(0)
    X = rng.random((n, d))
    y = np.sin(X.sum(1))
    m = bohip.ElasticGPE(d, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=n)
    m.append_(X.T, y)
    for _ in range(4):
        m.fit_()
    if os.environ.get("CHOL_PMC_SCORE"):
        Xs = rng.random((4096, d))
        for _ in range(3):
            m.score("EI", [float(y.max())], Xs.T, want_scores=False)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    exe = shutil.which("rocprofv3")
    tmp = tempfile.mkdtemp(prefix="bohip_cholpmc_", dir="/tmp")
    res = {}
    try:
        for ctrs in PASSES:
            out = os.path.join(tmp, "_".join(ctrs))
            r = subprocess.run([exe, "--pmc", *ctrs, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
                                os.path.abspath(__file__), "--child", str(n), str(d)], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                               capture_output=True, text=True, timeout=600)
            rows = 0
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    k = row.get("Kernel_Name", "").split("(")[0][:60]
                    res.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    rows += 1
            if rows == 0:
                print(f"pass {ctrs}: no rows (rc {r.returncode}) {r.stderr[-300:]!r}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(f"N = {n}, d = {d}: per-launch averages (first launch of each kernel dropped when there are several)")
    for k, cs in sorted(res.items()):
        parts = []
        for c, v in sorted(cs.items()):
            v = v[1:] if len(v) > 1 else v
            parts.append(f"{c} {sum(v) / len(v):.4g} (x{len(v)})")
        print(f"  {k:60s} " + "  ".join(parts))
    for k, cs in res.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f = cs["FETCH_SIZE"][1:] or cs["FETCH_SIZE"]
            w = cs["WRITE_SIZE"][1:] or cs["WRITE_SIZE"]
            fb, wb = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
            if fb + wb > 1e8:
                print(f"  {k[:40]:40s} fabric-side read {fb / 1e9:.2f} GB (FETCH_SIZE x 2), write {wb / 1e9:.2f} GB per launch")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
