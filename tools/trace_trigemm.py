"""Per-workgroup timeline of k_trigemm_sq (needs csrc/abl/libbohip_trace.so = build with -DBOHIP_TRACE=1).
Prints: span of the launch, per-CU busy statistics, how much of the span the average CU spends idle at the tail."""
import ctypes as C, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "bayesianoptimization.jl_amd", "csrc")
shutil.copy(os.path.join(csrc, "libbohip.so"), "/tmp/libbohip_real.so")
shutil.copy(os.path.join(csrc, "abl", "libbohip_trace.so"), os.path.join(csrc, "libbohip.so.tmp"))
os.replace(os.path.join(csrc, "libbohip.so.tmp"), os.path.join(csrc, "libbohip.so"))
try:
    import bohip
    from bohip import _lib
    from bench import synth, lhs, N_OBS, DIM
    lib = _lib.load()
    X, y = synth(0)
    R = int(os.environ.get("R", 4096))
    Xs = lhs(R, 1)
    m = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(np.full(DIM, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N_OBS)
    m.append_(X.T, y)
    for _ in range(5):
        m.score("EI", [float(y.max())], Xs.T)
    T = (N_OBS + 1 + 127) // 128
    CT = (R + 63) // 64
    pcs = (C.c_int * 4096)()
    NP = lib.bohip_debug_trigemm_pieces(T, C.c_int64(N_OBS), pcs, 4096)
    pieces = np.array(pcs[:NP])
    nb = 8 * ((CT + 7) // 8) * NP
    buf = (C.c_ulonglong * (6 * nb))()
    lib.bohip_debug_trace_read.restype = C.c_int
    assert lib.bohip_debug_trace_read(buf, C.c_int64(6 * nb)) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 6).astype(np.int64)
    t0, t1, hw, xcc_reg = t[:, 0].copy(), t[:, 1].copy(), t[:, 2], t[:, 3]
    # core clock while the kernel runs: s_memtime (clock64) against s_memrealtime (wall_clock64, 100 MHz), per workgroup
    dw, dc = (t[:, 1] - t[:, 0]).astype(np.float64), (t[:, 5] - t[:, 4]).astype(np.float64)
    okc = (t[:, 1] > 0) & (dw > 500)
    mhz = dc[okc] / dw[okc] * 100.0
    print(f"core clock from clock64 / wall_clock64 over {okc.sum()} workgroups: mean {mhz.mean():.0f} MHz, min {mhz.min():.0f}, max {mhz.max():.0f}")
    xcc = xcc_reg & 0xF
    print('XCC_ID register values seen:', sorted(set(xcc_reg.tolist()))[:16], ' agreement with blockIdx%8:', np.mean((xcc_reg & 7) == xcc))
    ok = t1 > 0
    base = t0[ok].min()         # wall_clock64: one 100 MHz counter for the whole device
    t0 -= base; t1 -= base
    start = 0
    span = t1[ok].max()
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"blocks {nb} (with work {ok.sum()}), span {span} ticks, distinct CUs {len(set(key[ok]))}")
    slot = np.arange(nb) >> 3
    n_local = (CT + 7) // 8
    pc = pieces[slot // n_local]
    rt, mode = pc & 0xffff, pc >> 16
    units = np.where(mode == 0, rt + 1.0, np.where(mode == 2, (rt + 1.0) / 2, (rt + 0.5) / 2))   # job length in 128 x 64 x 128 units
    for key_ in sorted(set(zip(rt[ok].tolist(), mode[ok].tolist()))):
        sel = ok & (rt == key_[0]) & (mode == key_[1])
        name = f"{key_[0]:2d}{['  ', 'a ', 'b ', 'a*'][key_[1]]}"
        print(f"  rt={name} jobs {sel.sum():4d} dur mean {np.mean((t1 - t0)[sel]):8.1f} ticks  per-unit {np.mean((t1 - t0)[sel]) / units[sel][0]:6.1f}  start mean {np.mean(t0[sel] - start):9.1f}  end mean {np.mean(t1[sel]):9.1f}")
    ends = {}
    busy = {}
    for k_, a, b in zip(key[ok], t0[ok], t1[ok]):
        ends[k_] = max(ends.get(k_, 0), b - start)
        busy[k_] = busy.get(k_, 0) + (b - a)
    for x in range(8):   # does a whole XCD (a fixed eighth of the jobs: blockIdx % 8) finish early or late?
        sel = ok & (xcc == x)
        ex = [v for k_, v in ends.items() if (k_ >> 8) == x]
        print(f"  XCD {x}: CUs {len(ex):3d}  jobs {sel.sum():4d}  last end {t1[sel].max():7d}  mean CU end {np.mean(ex):9.0f}  sum of WG durations {np.sum((t1 - t0)[sel]):10d}")
    e = np.array(list(ends.values()))
    print(f"per-CU last end: min {e.min()} mean {e.mean():.0f} max {e.max()} (span {span}); mean idle tail {(span - e.mean()) / span:.3%}")
    bz = np.array(list(busy.values()))
    print(f"per-CU sum of WG durations / span: mean {np.mean(bz) / span:.3f} (2.0 = two resident WGs all the time), min {bz.min() / span:.3f}, max {bz.max() / span:.3f}")
    ev = {}
    for k_, a, b in zip(key[ok], t0[ok], t1[ok]):
        ev.setdefault(k_, []).append((a - start, 1)); ev[k_].append((b - start, -1))
    one = zero = 0
    for k_, lst in ev.items():
        lst.sort(); lvl = 0; prev = 0
        for tt, d in lst:
            if lvl == 1: one += tt - prev
            if lvl == 0: zero += tt - prev
            prev = tt; lvl += d
        zero += span - prev
    n = len(ev)
    print(f"fraction of CU-time with ONE resident WG {one / (n * span):.3%}, with NONE {zero / (n * span):.3%}")
    # phases of the main loop per wave (core-clock cycles): issue+LDS+MFMA | vmcnt wait | barrier wait
    pb = (C.c_ulonglong * (nb * 8 * 4))()
    lib.bohip_debug_phase_read.restype = C.c_int
    assert lib.bohip_debug_phase_read(pb, C.c_int64(nb * 8 * 4)) == 0
    ph = np.frombuffer(pb, dtype=np.uint64).reshape(nb, 8, 4).astype(np.float64)
    it = ph[:, :, 3]
    for name, sel in [("first-wave older (rt 20-23)", (rt >= 20)), ("first-wave younger (rt 16-19)", (rt >= 16) & (rt < 20)), ("half jobs", mode > 0), ("all", rt >= 0)]:
        p = ph[sel]
        n = p[:, :, 3].sum()
        a, b, c = p[:, :, 0].sum() / n, p[:, :, 1].sum() / n, p[:, :, 2].sum() / n
        print(f"{name:30s} per iteration: issue+lds+mfma {a:7.0f}  vmcnt wait {b:6.0f}  barrier wait {c:6.0f}  total {a + b + c:7.0f} cycles")
    w = ph[rt >= 20]
    print("older jobs, per wave id (0-3 k-half 0, 4-7 k-half 1): barrier wait", np.round(w[:, :, 2].sum(0) / w[:, :, 3].sum(0)), " issue+mfma", np.round(w[:, :, 0].sum(0) / w[:, :, 3].sum(0)))
finally:
    shutil.copy("/tmp/libbohip_real.so", os.path.join(csrc, "libbohip.so.tmp"))
    os.replace(os.path.join(csrc, "libbohip.so.tmp"), os.path.join(csrc, "libbohip.so"))
