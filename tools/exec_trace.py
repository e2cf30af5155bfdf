"""Timeline of the executor form of the factorisation (csrc/kernels_exec.hip): per-task claim / start / end times and the chain's
marks.  Build: make -C bayesianoptimization.jl_amd/csrc abl/libbohip_choltrace.so.  wall_clock64 ticks are 10 ns.
usage: python tools/exec_trace.py [N]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bohip import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "abl", "libbohip_choltrace.so")
import bohip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = 8
rng = np.random.default_rng(0)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
for _ in range(3):
    m.set_params_(logNoise=-2.0); m.fit_()
lib = C.CDLL(_lib.LIB_PATH)
T = (N + 1 + 127) // 128
ld = ((N + 1 + 127) // 128) * 128 + 16
# the records (same builder, fake addresses) tell which task is what
f = lib.bohip_debug_exec_tasks
f.restype = C.c_int64
f.argtypes = [C.c_int, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
NQ = 6
INV_G = int(os.environ.get("BOHIP_CHOL_INV_G", "8"))
qbeg = (C.c_int * (NQ + 1))(); lay = (C.c_int64 * 16)()
n = f(T, ld, 1 << 44, 2 << 44, 3 << 44, 4 << 44, INV_G, None, 0, qbeg, lay)
recs = np.zeros((n, 16), dtype=np.uint64)
f(T, ld, 1 << 44, 2 << 44, 3 << 44, 4 << 44, INV_G, recs.ctypes.data_as(C.c_void_p), n, qbeg, lay)
kc = recs.view(np.int32)[:, 22]
ct = (C.c_ulonglong * 8192)()
lib.bohip_debug_chol_trace_read.argtypes = [C.c_void_p, C.c_int64]
assert lib.bohip_debug_chol_trace_read(ct, 8192) == 0
ct = np.array(ct, dtype=np.int64)
et = (C.c_ulonglong * (8 * min(n, (1 << 18))))()
lib.bohip_debug_exec_trace_read.argtypes = [C.c_void_p, C.c_int64]
assert lib.bohip_debug_exec_trace_read(et, 8 * min(n, (1 << 18))) == 0
et = np.array(et, dtype=np.int64).reshape(-1, 8)
t0 = ct[3072]
us = lambda v: (np.asarray(v, dtype=np.float64) - t0) / 100.0
qb = list(qbeg)
print(f"N={N} T={T}: {n} tasks, queues {qb}; chain total {us(ct[3072 + 2 * (T - 1) + 1]):.0f} us; executor last task end {us(et[:, 2].max()):.0f} us")
# utilisation: busy worker-time per window, by queue
end = us(et[:, 2].max())
W = 500.0
names = ["urgent", "solve/late", "early", "inv-rows", "bulk", "inv-waves"]
print(f"window(us)   busy workers (of 512) by queue [{', '.join(names)}]  | idle-looking share | chain blocks finished")
blk_end = us(ct[3072 + 1:3072 + 2 * T:2])
for w0 in np.arange(0, end, W):
    row = []
    for q in range(NQ):
        s, e = us(et[qb[q]:qb[q + 1], 1]), us(et[qb[q]:qb[q + 1], 2])
        row.append(np.clip(np.minimum(e, w0 + W) - np.maximum(s, w0), 0, None).sum() / W)
    print(f"{w0:8.0f}   " + " ".join(f"{r:6.1f}" for r in row) + f"   total {sum(row):6.1f}   | blocks done {int((blk_end < w0 + W).sum())}")
print("window(us)   workers LOOKING or WAITING for a claimed task's counters, by queue of the task they then ran")
for w0 in np.arange(0, end, W):
    row = []
    for q in range(NQ):
        s_, e_ = us(et[qb[q]:qb[q + 1], 0]), us(et[qb[q]:qb[q + 1], 1])
        row.append(np.clip(np.minimum(e_, w0 + W) - np.maximum(s_, w0), 0, None).sum() / W)
    print(f"{w0:8.0f}   " + " ".join(f"{r:6.1f}" for r in row) + f"   total {sum(row):6.1f}")
dur = us(et[:, 2]) - us(et[:, 1])
look = us(et[:, 1]) - us(et[:, 0])
for q in range(NQ):
    sl = slice(qb[q], qb[q + 1])
    k_ = kc[sl]
    print(f"queue {q} ({names[q]}): {qb[q+1]-qb[q]} tasks, run time mean {dur[sl].mean():.1f} us (per 128 of K: {(dur[sl] / (k_ / 8)).mean():.2f} us), look+wait before start mean {look[sl].mean():.1f} us")
# critical latency per block: pivot end -> inverse published -> first Solve row -> first-row Late tiles -> the chain sees rest[k]
print("block:  pivot end | inverse +us | S(k+3,k) by the chain's solve_follower | first-row Late start,end | Solve far rows start,end | follower saw rest[k] (relative to pivot end)")
NSF = int(lay[10])
pos0, pos1 = qb[0], qb[1]
rows = {}
for k in range(T - 3):
    rb = k + 3 + NSF; nbu = min(2, max(0, T - rb))
    nurg = 2 * (3 + max(0, min(T, rb) - (k + 4))) + 4 * nbu
    nsolve = 2 * max(0, T - rb - 2)
    late = et[pos0:pos0 + 6]; solve = et[pos1:pos1 + nsolve]
    rows[k] = late
    pos0 += nurg
    pos1 += 2 * nsolve
    pe = us(ct[3072 + 2 * k + 1])
    if k < 6 or k % 8 == 0:
        sv = f"{us(solve[:, 1]).min() - pe:6.1f} {us(solve[:, 2]).max() - pe:6.1f}" if nsolve else "   -      -  "
        print(f"{k:4d}: {pe:9.1f} | {us(ct[4096 + k]) - pe:6.1f} | {us(ct[4608 + k]) - pe:6.1f} | "
              f"{us(late[:, 1]).min() - pe:6.1f} {us(late[:, 2]).max() - pe:6.1f} | {sv} | {us(ct[4352 + k + 1]) - pe:6.1f} | next pivot start {us(ct[3072 + 2 * (k + 1)]) - pe:6.1f}")
# the chain's own waits, per block k (us relative to the pivot end of block k-1): what held pivot k back?
pdur = [(us(ct[3072 + 2 * k + 1]) - us(ct[3072 + 2 * k])) for k in range(T)]
gaps = [(us(ct[3072 + 2 * (k + 1)]) - us(ct[3072 + 2 * k + 1])) for k in range(T - 1)]
print(f"pivot duration mean {np.mean(pdur):.1f} us (min {np.min(pdur):.1f} max {np.max(pdur):.1f}); gap to the next pivot mean {np.mean(gaps):.1f} us (median {np.median(gaps):.1f}, max {np.max(gaps):.1f})")
print("block k: [rel. to pivot k-1 end] owner waits for crit[k-2] from .. to | follower of row k saw crit[k-2] | owner saw last panel of L(k,k-1) | pivot k start || first-row Late(k-1) tasks: start / stage-2 in / loop done / end")
for k in (list(range(2, T - 3)) if T <= 32 else list(range(2, 8)) + list(range(8, T - 3, 6))):
    pe = us(ct[3072 + 2 * (k - 1) + 1])
    lt = rows.get(k - 1)
    ls = " ".join(f"[{us(r[1]) - pe:.0f}/{us(r[4]) - pe:.0f}/{us(r[5]) - pe:.0f}/{us(r[2]) - pe:.0f}]" for r in lt) if lt is not None else ""
    print(f"{k:4d}: {us(ct[4864 + k]) - pe:6.1f} .. {us(ct[5120 + k]) - pe:6.1f} | {us(ct[5632 + k - 1]) - pe:6.1f} | {us(ct[5376 + k]) - pe:6.1f} | {us(ct[3072 + 2 * k]) - pe:6.1f} || {ls}")

# the solve followers' chain: follower f of block k (row k+3+f) -- input tile final / S complete -- and the urgent Late task that
# produced its input (Late(k-1) of tile (k+3+f, k): record pair 3 + f of block k-1's urgent group), all relative to pivot k-1's end
print("solve followers, block k: per follower f [input seen / S done] and the Late(k-1) task of its input tile [start/stage-2 in/end] (us rel. to pivot k-1 end; pivot k start in the last column)")
posu = {}
p_ = qb[0]
for k in range(T - 3):
    posu[k] = p_
    p_ += 2 * (3 + max(0, min(T, k + 3 + NSF) - (k + 4))) + 4 * min(2, max(0, T - (k + 3 + NSF)))
for k in list(range(2, 6)) + list(range(6, T - 4, 5)):
    pe = us(ct[3072 + 2 * (k - 1) + 1])
    parts = []
    for f in range(min(NSF, 4)):
        if k + 3 + f >= T:
            continue
        txt = f"f{f} [{us(ct[7168 + 256 * f + k]) - pe:.0f}/{us(ct[6144 + 256 * f + k]) - pe:.0f}]"
        # input tile (k+3+f, k) = column c = k of Late(k-1): row i = (k-1) + 4 + f -> urgent iff f + 1 < NSF: pair index 3 + f
        if f + 1 < NSF and k - 1 in posu:
            r0 = et[posu[k - 1] + 2 * (3 + f)]
            r1 = et[posu[k - 1] + 2 * (3 + f) + 1]
            txt += f" <- Late [{min(us(r0[1]), us(r1[1])) - pe:.0f}/{max(us(r0[4]), us(r1[4])) - pe:.0f}/{max(us(r0[2]), us(r1[2])) - pe:.0f}]"
        parts.append(txt)
    print(f"{k:4d}: " + "  ".join(parts) + f"   | pivot k start {us(ct[3072 + 2 * k]) - pe:.0f}, end {us(ct[3072 + 2 * k + 1]) - pe:.0f}")

GRP = INV_G >= 2 and T >= int(os.environ.get("BOHIP_CHOL_INV_GRP_MIN", "40"))   # the inverse queues in their group form (exec_task_list)
if GRP and qb[4] > qb[3]:
    A_addr = recs[:, 0].astype(np.int64); C_addr = recs[:, 2].astype(np.int64)
    bS, bW = 2 << 44, 3 << 44
    rmw = recs.view(np.int32)[:, 24]
    q3 = np.arange(qb[3], qb[4]); q5 = np.arange(qb[5], qb[6])
    dprod = q3[(C_addr[q3] >= bW)]                                   # D_g: the products of the short row chain (C in W)
    drow = ((C_addr[dprod] - bW) // 8 // ld) // 128
    is_prod = (A_addr[q5] >= bW)                                      # queue 5: group products have their A operand in W, rounds in S
    prow = ((C_addr[q5] - bW) // 8 // ld) // 128
    fin = (~is_prod) & ((rmw[q5] & 4) != 0)                           # the round that completes P (stores P' too)
    print("inverse, group form: group g (rows) | pivot of its last row ends (us) | D_g done | last round into its rows: first start, last end | "
          "products: first start, last end, records, mean run (all relative to that pivot's end)")
    for g in range(0, (T + INV_G - 1) // INV_G):
        r0, r1 = INV_G * g, min(T, INV_G * (g + 1))
        pe = us(ct[3072 + 2 * (r1 - 1) + 1])
        sd = dprod[(drow >= r0) & (drow < r1)]
        txt = f"{g:3d} ({r0:2d}..{r1 - 1:2d}) | {pe:8.1f} | " + (f"{us(et[sd, 2]).max() - pe:8.1f}" if len(sd) else "       -")
        sf = q5[fin & (prow >= r0) & (prow < r1)]
        sp = q5[is_prod & (prow >= r0) & (prow < r1)]
        txt += " | " + (f"{us(et[sf, 1]).min() - pe:8.1f} {us(et[sf, 2]).max() - pe:8.1f}" if len(sf) else "       -        -")
        txt += " | " + (f"{us(et[sp, 1]).min() - pe:8.1f} {us(et[sp, 2]).max() - pe:8.1f} {len(sp):5d} {dur[sp].mean():6.1f}" if len(sp) else "       -")
        print(txt)
    rd = q5[~is_prod]
    print(f"rounds: {len(rd)} records, mean run {dur[rd].mean():.1f} us, look+wait {look[rd].mean():.1f}; products: {int(is_prod.sum())} records, "
          f"mean run {dur[q5[is_prod]].mean():.1f} us, look+wait {look[q5[is_prod]].mean():.1f}")

# the inverse's row chain: per row i, its partial / last / product records (first start .. last end), relative to the end of pivot i
if qb[4] > qb[3] and not GRP:
    print("inverse row chain: row i | pivot i end (us) | partial: first start, last end | last piece: first start, last end | product: first start, last end   (rel. to pivot i end)")
    Cc = recs[:, 2]
    rws = recs.view(np.int32)[:, 24]
    base_W = 3 << 44
    sl = slice(qb[3], qb[4])
    off = (Cc[sl].astype(np.int64) - base_W) // 8
    row_i = (off // ld) // 128
    kind = np.where((rws[sl] & 3) == 0, 2, np.where((rws[sl] & 4) != 0, 1, 0))   # 0 partial (both pieces), 1 last (stores Z'), 2 product
    for i in list(range(1, 8)) + list(range(8, T, max(1, T // 12))) + [T - 1]:
        pe = us(ct[3072 + 2 * i + 1])
        txt = []
        for kd in range(3):
            sel = np.where((row_i == i) & (kind == kd))[0]
            if len(sel) == 0:
                txt.append("      -       -")
                continue
            e = et[qb[3] + sel]
            txt.append(f"{us(e[:, 1]).min() - pe:7.1f} {us(e[:, 2]).max() - pe:7.1f}")
        print(f"{i:4d}: {pe:9.1f} | " + " | ".join(txt))

    if os.environ.get("ROWDUMP"):
        for i in [int(v) for v in os.environ["ROWDUMP"].split(",")]:
            pe = us(ct[3072 + 2 * i + 1])
            print(f"row {i} (pivot end {pe:.1f}; inverse published {us(ct[4096 + i]) - pe:+.1f}): kind j half | looking since / start / end (rel. to pivot end) | workgroup | queue position")
            for kd, nm in ((0, "partial"), (1, "last"), (2, "product")):
                for ix in np.where((row_i == i) & (kind == kd))[0]:
                    e = et[qb[3] + ix]
                    jj = ((off[ix] % ld) // 128, ((off[ix] % ld) % 128) // 64)
                    print(f"   {nm:8s} j={jj[0]:2d} h={jj[1]} | {us(e[0]) - pe:8.1f} {us(e[1]) - pe:8.1f} {us(e[2]) - pe:8.1f} | wg {int(e[3]):3d} | {ix}")

# the inverse's row chain against the pivots: row i's products W(i, j) = W_ii Z(i, j) (queue 3 records whose A operand is a diagonal tile of W)
if qb[4] > qb[3] and not GRP:
    A_addr = recs.view(np.uint64)[:, 0].astype(np.int64)
    baseW = 3 << 44
    in_q3 = np.arange(qb[3], qb[4])
    offA = (A_addr[in_q3] - baseW) // 8
    isW = (A_addr[in_q3] >= baseW) & (A_addr[in_q3] < (4 << 44))
    rt, ctile = offA // (128 * ld), (offA % ld) // 128
    prod = isW & (rt == ctile)
    print("inverse rows: row i -- last product of the row ends (us after pivot i's end) | since the previous row's")
    prev = None
    for i in range(1, T):
        sel = in_q3[prod & (rt == i)]
        if len(sel) == 0: continue
        e = us(et[sel, 2]).max()
        pe = us(ct[3072 + 2 * i + 1])
        if i < 8 or i % 4 == 0 or i >= T - 3:
            print(f"  row {i:3d}: {e - pe:8.1f} | {'' if prev is None else f'{e - prev:7.1f}'}")
        prev = e

# round 6: solve follower 1 of block 5, phase marks inside follow_block (thread 0 = wave 0): panel start | staged (before the barrier) | barrier passed | solved + barrier | updated (before the barrier)
if ct[5920] > 0:
    print("solve follower 1, block 5, per panel (us since its first panel started): start | staged | +barrier | solved+barrier | updated")
    for p_ in range(8):
        print("  panel", p_, [round((ct[5920 + 5 * p_ + j] - ct[5920]) / 100.0, 2) for j in range(5)], "| inside the solve (wave 0): products done, stores issued", [round((ct[5960 + 3 * p_ + j] - ct[5920]) / 100.0, 2) for j in range(2)])
# round 6: second-piece durations of the two-piece urgent tasks (stage-2 in -> loop done; K = 128: ~11 us on a quiet chip) -- which are slow, on which
# workgroup, and how many bulk / wave tasks ran at that moment
p2 = us(et[qb[0]:qb[1], 5]) - us(et[qb[0]:qb[1], 4])
two = (et[qb[0]:qb[1], 4] != 0) & (p2 > 0) & (p2 < 200)
st2 = us(et[qb[0]:qb[1], 4])
wk = et[qb[0]:qb[1], 3].astype(np.int64)
bs, be = us(et[qb[4]:qb[5], 1]), us(et[qb[4]:qb[5], 2])
ws, we = (us(et[qb[5]:qb[6], 1]), us(et[qb[5]:qb[6], 2])) if qb[6] > qb[5] else (np.zeros(0), np.zeros(0))
rs, re_ = us(et[qb[3]:qb[4], 1]), us(et[qb[3]:qb[4], 2])
print("two-piece urgent tasks: second piece (us) | at (us) | workgroup | bulk / inverse-wave / inverse-row tasks running at its start")
slow_w, fast_w = [], []
for j in np.flatnonzero(two):
    t_ = st2[j]
    nb_, nw_, nr_ = int(((bs <= t_) & (be > t_)).sum()), int(((ws <= t_) & (we > t_)).sum()), int(((rs <= t_) & (re_ > t_)).sum())
    (slow_w if p2[j] > 16 else fast_w).append((nb_, nw_, nr_))
    if p2[j] > 16: print(f"   {p2[j]:6.1f} | {t_:7.0f} | wg {wk[j]:3d} | {nb_:3d} {nw_:3d} {nr_:3d}")
for name, arr in (("slow (> 16 us)", slow_w), ("the others", fast_w)):
    a = np.array(arr) if arr else np.zeros((0, 3))
    print(f"{name}: {len(arr)} tasks; bulk / wave / row tasks running beside them, mean: " + (" ".join(f"{v:.1f}" for v in a.mean(0)) if len(arr) else "-"))
