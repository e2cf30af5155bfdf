"""CPU simulation of k_trigemm_sq's job schedule on one XCD (the XCDs run their eighth of the jobs independently): which row
tiles should be issued as two 64-row half jobs so that the launch ends evenly?  Model (calibrated on tools/trace_trigemm.py,
profiles/r02_trigemm_workgroup_timeline.txt): a CU holds at most two workgroups; two residents share its matrix pipe 55 : 45
(the older one first), a lone resident runs at 0.85 of it; every job has a fixed part (prologue latency, fold, epilogue) that
does not use the pipe; workgroups are dispatched in order to the first free slot.
Usage: python tools/sim_trigemm_tail.py [T] [candidate tiles per XCD]"""
import heapq, sys
import numpy as np

T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
NCT = int(sys.argv[2]) if len(sys.argv) > 2 else 8
CUS, FIXED, LAST_HALF = 32, 0.45, True     # fixed cost per job in units (1 unit = 128 x 64 x 128 contraction ~ 9.3 us of one CU)


def pieces(halved):
    """(cost in units, label) per row piece, heaviest first"""
    out = []
    for rt in range(T):
        if rt == T - 1 and LAST_HALF:
            out.append(((rt + 1) / 2.0, f"{rt}L")); continue
        if rt in halved:
            out.append(((rt + 0.5) / 2.0, f"{rt}a")); out.append(((rt + 1.0) / 2.0, f"{rt}b"))
        else:
            out.append((rt + 1.0, f"{rt}"))
    out.sort(key=lambda p: -p[0])
    return out


def simulate(halved, speed=1.0, seed=0):
    rng = np.random.default_rng(seed)
    jobs = [c for c, _ in pieces(halved) for _ in range(NCT)]
    # event simulation with processor sharing inside a CU
    res = [[] for _ in range(CUS)]      # per CU: list of [remaining pipe work, remaining fixed time, age]
    t, nxt, age = 0.0, 0, 0
    ends = [0.0] * CUS
    busy_area = 0.0

    def rates(r):
        act = [j for j in r if j[1] <= 0]          # jobs past their fixed part use the pipe
        if len(act) == 2:
            o, y = (act[0], act[1]) if act[0][2] < act[1][2] else (act[1], act[0])
            return {id(o): 0.55 * speed, id(y): 0.45 * speed}
        if len(act) == 1:
            return {id(act[0]): (0.85 if len(r) == 1 else 0.97) * speed}
        return {}
    while True:
        for c in range(CUS):                        # dispatch in order to free slots (fewest residents first)
            pass
        order = sorted(range(CUS), key=lambda c: len(res[c]))
        for c in order:
            while len(res[c]) < 2 and nxt < len(jobs):
                res[c].append([jobs[nxt] * (1 + 0.02 * rng.standard_normal()), FIXED * 0.5, age]); age += 1; nxt += 1
                break
        if all(len(r) == 0 for r in res):
            break
        if any(len(r) < 2 for r in res) and nxt < len(jobs):
            continue
        # next event
        dt = 1e9
        for r in res:
            rt_ = rates(r)
            for j in r:
                if j[1] > 0: dt = min(dt, j[1])
                elif id(j) in rt_: dt = min(dt, j[0] / rt_[id(j)])
        for c, r in enumerate(res):
            rt_ = rates(r)
            for j in r:
                if j[1] > 0: j[1] -= dt
                elif id(j) in rt_: j[0] -= dt * rt_[id(j)]
            if r: busy_area += dt
            keep = [j for j in r if j[1] > 1e-12 or j[0] > 1e-9]
            if len(keep) < len(r): ends[c] = t + dt
            res[c] = keep
        t += dt
    work = sum(jobs)
    return t, np.mean(ends), work


base = None
cands = [set()] + [set(range(a, b + 1)) for a in range(0, 12) for b in range(a, 16) if b - a <= 9]
rows = []
for h in cands:
    ts = [simulate(h, seed=s) for s in range(6)]
    span = np.mean([x[0] for x in ts]); mean_end = np.mean([x[1] for x in ts]); work = ts[0][2]
    rows.append((span, mean_end, work, sorted(h)))
rows.sort(key=lambda r: r[0])
for span, mean_end, work, h in rows[:12] + [r for r in rows if not r[3]]:
    print(f"halved {str(h):44s} span {span:7.2f} units  mean CU end {mean_end:7.2f}  idle tail {(span - mean_end) / span:6.2%}  "
          f"pipe work per CU {work / CUS:6.2f}  jobs {len(pieces(set(h))) * NCT}")
