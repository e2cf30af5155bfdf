"""CPU replay of the executor form of the blocked Cholesky (csrc/kernels_exec.hip, `cholesky_exec` in csrc/bohip.hip).

The task records are built on the HOST (a pure function of addresses, ld and T) and executed on the device by k_chol_exec.
Here the real builder is called through its test hook with fake base addresses, and the records are replayed with NumPy
against a model of the persistent chain kernel (k_chol_chain in its mode2 = 100 view, with its solve_follower workgroup that
delivers S(k+3, k)):

  * every queue is consumed strictly in order, a task only when all of its dependency counters have reached their value
    -- exactly the device's claim rule -- under several adversarial interleavings (tasks as early as possible / chain as
    early as possible / random);
  * the replay must never dead-lock, every record must be consumed, and the factor must equal numpy's Cholesky.

A dependency the builder forgot shows up as a wrong factor in the "tasks first" order; a circular or mis-indexed one as a
dead-lock.  Three more rules of the device that a sequential replay would not notice by itself are enforced on the way: a location
read through LDS-DMA is never written afterwards; W / W' tiles that are read through LDS-DMA are written exactly once (the XCDs' L2s
do not see each other's stores); and the copy S -> L happens when the chain kernel's roles are done, after which nothing touches L.
Both forms of the inverse queues are replayed (row by row, and the group form the library uses from 28 row tiles on).
(No GPU needed: this is host logic.)"""
import ctypes as C

import numpy as np
import pytest

TILE, CT, KC, NONE = 128, 64, 16, 0xFFFFFFFF
NQ = 6   # queues: urgent | solve + late | early | inverse: row chain | bulk | inverse: waves
BASE_L, BASE_S, BASE_W, BASE_WT = 1 << 44, 2 << 44, 3 << 44, 4 << 44
INV_G = 2   # blocks per piece of the inverse queue's long contractions (small, so that multi-piece tiles occur at test sizes)


def get_tasks(T, ld, inv_g=INV_G, groups=False):
    """groups: the inverse queues in their group form (the library's choice from BOHIP_CHOL_INV_GRP_MIN row tiles on), else row by row"""
    import os

    from bohip import _lib

    os.environ["BOHIP_CHOL_INV_GRP_MIN"] = "0" if groups else "1000000"

    lib = C.CDLL(_lib.LIB_PATH)
    f = lib.bohip_debug_exec_tasks
    f.restype = C.c_int64
    f.argtypes = [C.c_int, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int),
                  C.POINTER(C.c_int64)]
    qbeg = (C.c_int * (NQ + 1))()
    layout = (C.c_int64 * 16)()
    n = f(T, ld, BASE_L, BASE_S, BASE_W, BASE_WT, inv_g, None, 0, qbeg, layout)
    buf = np.zeros((n, 16), dtype=np.uint64)
    assert f(T, ld, BASE_L, BASE_S, BASE_W, BASE_WT, inv_g, buf.ctypes.data_as(C.c_void_p), n, qbeg, layout) == n
    names = ["panel", "solved", "crit", "rest", "col", "farall", "fol", "colall", "colr", "xp", "nsf", "inv", "xp3", "bulk_tiles_per_claim"]
    del os.environ["BOHIP_CHOL_INV_GRP_MIN"]
    return buf, list(qbeg), dict(zip(names, layout))


def decode(rec):
    w = rec.view(np.uint32)
    t = dict(A=int(rec[0]), B=int(rec[1]), C=int(rec[2]), P=int(rec[3]))
    t["dep"] = [(int(w[8 + d]), int(w[14 + d])) for d in range(6) if int(w[8 + d]) != NONE]
    t["sig"] = [int(w[20 + s]) for s in range(2) if int(w[20 + s]) != NONE]
    i32 = rec.view(np.int32)
    t["kc"], t["diag_h"], t["rmw"], t["ct"] = int(i32[22]), int(i32[23]), int(i32[24]) & 3, (int(i32[24]) & 4) != 0
    t["kc_split"] = int(i32[26])
    t["dep2"] = [(int(w[27 + s]), int(w[29 + s])) for s in range(2) if int(w[27 + s]) != NONE]
    if t["kc_split"] == 0:
        assert not t["dep2"]
    return t


def replay(T, order, seed=0, inv_g=INV_G, groups=False):
    ld = TILE * T + 16
    N = TILE * T
    rng = np.random.default_rng(seed)
    recs, qbeg, lay = get_tasks(T, ld, inv_g, groups)
    tasks = [decode(r) for r in recs]
    # a well-conditioned SPD matrix (kernel matrix of random points + noise), lower triangle only -- as k_build_cov leaves it
    X = rng.random((N, 3))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    K = np.exp(-0.5 * d2 / 0.09) + 0.05 * np.eye(N)
    mats = {BASE_L: np.zeros((ld, ld)), BASE_S: np.zeros((ld, ld)), BASE_W: np.zeros((ld, ld)), BASE_WT: np.zeros((ld, ld))}
    mats[BASE_L][:N, :N] = np.tril(K)
    # poison what must be written before it is read
    mats[BASE_S][:] = np.nan
    for i in range(T):   # W, W': everything but the diagonal tiles (their zero halves are relied on)
        for j in range(T):
            if i != j:
                mats[BASE_W][i * TILE:(i + 1) * TILE, j * TILE:(j + 1) * TILE] = np.nan
                mats[BASE_WT][i * TILE:(i + 1) * TILE, j * TILE:(j + 1) * TILE] = np.nan
    flags = np.zeros(lay["xp3"] + 12 * T, dtype=np.int64)
    pre3 = lay["xp3"] + 8 * T   # pre3[4 k + j]: tile (k+3, k+1+j) carries every block before k (the executor's share)
    # operands of the contraction engine arrive through LDS-DMA, i.e. through caches nothing invalidates while the kernel runs:
    # a location that was read that way must never be written afterwards
    dma_read = {b: np.zeros((ld, ld), dtype=bool) for b in mats}
    # ... and the other way round: the executor's workgroups sit on eight XCDs whose L2s do not see each other's stores, so a location
    # that tasks write MORE THAN ONCE (a sum that grows in place) may be served stale to a later LDS-DMA read -- such locations may
    # only ever be read through the read-modify-write path (agent-scope loads).  (The first group form of the inverse summed Z in place
    # in W and read the finished tile as an operand: right in every replay, 1 % off on the device now and then.)  Enforced for W and W';
    # the trailing-matrix tiles of L are the known exception -- the row solves read them after many rounds, the last of them a
    # read-modify-write long after the others, and thousands of refits have compared bit for bit (tools/chol_stress.py).
    writes = {b: np.zeros((ld, ld), dtype=np.uint8) for b in mats}

    def view(addr, rows, cols):
        base = addr & ~((1 << 44) - 1)
        off = (addr - base) // 8
        r, c = divmod(off, ld)
        assert (addr - base) % 8 == 0 and base in mats and r + rows <= ld and c + cols <= ld, hex(addr)
        return mats[base][r:r + rows, c:c + cols]

    def mark(addr, rows, cols, reading):
        base = addr & ~((1 << 44) - 1)
        r, c = divmod((addr - base) // 8, ld)
        region = dma_read[base][r:r + rows, c:c + cols]
        wr = writes[base][r:r + rows, c:c + cols]
        if reading:
            region[:] = True
            assert base == BASE_L or wr.max(initial=0) <= 1, f"LDS-DMA read of a location that tasks wrote more than once: {hex(addr)}"
        else:
            assert not region.any(), f"write to a location an earlier task read through LDS-DMA: {hex(addr)}"
            np.minimum(wr + 1, 200, out=wr, casting="unsafe")

    def tile(base, i, j):
        return mats[base][i * TILE:(i + 1) * TILE, j * TILE:(j + 1) * TILE]

    def run_task(t):
        K_ = KC * t["kc"]
        A = view(t["A"], TILE, K_)
        B = view(t["B"], CT, K_)
        Cv = view(t["C"], TILE, CT)
        assert not np.isnan(A).any() and not np.isnan(B).any(), "operand read before it was written"
        if t["kc_split"]:   # the first piece ran on what was there when the task became claimable: must be the final operands
            K1 = KC * t["kc_split"]
            assert id(t) in snap, "two-piece task ran without ever having been claimable on its first-stage counters alone"
            a1, b1 = snap[id(t)]
            assert np.array_equal(a1, A[:, :K1]) and np.array_equal(b1, B[:, :K1]), "first-piece operand changed after the claim"
        mark(t["A"], TILE, K_, True)
        mark(t["B"], CT, K_, True)
        prod = A @ B.T
        keep = np.ones((TILE, CT), dtype=bool)
        if t["diag_h"] >= 0:   # half of a diagonal tile: the strict upper triangle is never written (nor meaningful when read)
            keep = (CT * t["diag_h"] + np.arange(CT)[None, :]) <= np.arange(TILE)[:, None]
        if t["rmw"] == 1:
            assert not np.isnan(Cv[keep]).any(), "read-modify-write of a tile nobody has written"
            new = Cv - prod
            if t["P"] and not t["ct"]:
                Pv = view(t["P"], TILE, CT)
                assert not np.isnan(Pv[keep]).any(), "P read before it was written"
                new = new - Pv
        else:
            assert t["P"] == 0 or t["ct"]
            new = prod if t["rmw"] == 0 else -prod
        mark(t["C"], TILE, CT, False)
        Cv[keep] = new[keep]
        if t["ct"]:   # the same values once more, transposed, to the [64][128] block at P
            assert t["diag_h"] < 0
            mark(t["P"], CT, TILE, False)
            view(t["P"], CT, TILE)[:, :] = new.T
        for s in t["sig"]:
            flags[s] += 8   # eight waves add one each

    def ready(t):
        # the second-stage counters are waited for inside the task after its first `kc_split` chunks; the replay runs a task
        # in one piece, so it takes them as ordinary dependencies (and checks below that the first piece would not have needed them)
        return all(flags[i] >= w for i, w in t["dep"] + t["dep2"])

    snap = {}
    heads = [qbeg[q] for q in range(NQ)]
    chain_k = 0   # next block of the chain
    copied = False

    def chain_can_run():
        return chain_k < T and (chain_k == 0 or flags[lay["rest"] + chain_k - 1] >= 24 or chain_k + 2 >= T)

    def chain_step():
        nonlocal chain_k
        k = chain_k
        L = mats[BASE_L]
        S = mats[BASE_S]
        if k >= 1 and k + 2 < T:
            assert flags[lay["rest"] + k - 1] >= 24
        Lkk = np.linalg.cholesky(np.tril(tile(BASE_L, k, k)) + np.tril(tile(BASE_L, k, k), -1).T)
        tile(BASE_L, k, k)[:, :] = Lkk
        tile(BASE_W, k, k)[:, :] = np.linalg.inv(Lkk)
        tile(BASE_WT, k, k)[:, :] = np.linalg.inv(Lkk).T
        flags[lay["solved"] + k] = 1
        for r in (k + 1, k + 2):
            if r < T:
                tile(BASE_S, r, k)[:, :] = np.linalg.solve(Lkk, tile(BASE_L, r, k).T).T
                flags[lay["xp"] + (k * T + r) * 8 + 7] = 1
        if k + 1 < T:
            s1 = tile(BASE_S, k + 1, k)
            tile(BASE_L, k + 1, k + 1)[:, :] -= np.tril(s1 @ s1.T)
        if k + 2 < T:
            s1, s2 = tile(BASE_S, k + 1, k), tile(BASE_S, k + 2, k)
            tile(BASE_L, k + 2, k + 1)[:, :] -= s2 @ s1.T
            tile(BASE_L, k + 2, k + 2)[:, :] -= np.tril(s2 @ s2.T)
        chain_k += 1

    NSF = lay["nsf"]
    tf_k = [0] * NSF   # next block of each of the chain kernel's solve_follower workgroups (follower f: row k+3+f of block k)

    def tf_can_run(f):
        k = tf_k[f]
        r = k + 3 + f
        if r >= T or k >= chain_k:
            return False
        if k == 0:
            return True
        nb = max(k // 4 - 1, 0)
        return flags[lay["xp"] + ((k - 1) * T + r) * 8] >= 16 * (nb + 1)

    def tf_step(f):
        k = tf_k[f]
        r = k + 3 + f
        Lkk = tile(BASE_L, k, k)
        tile(BASE_S, r, k)[:, :] = np.linalg.solve(Lkk, tile(BASE_L, r, k).T).T
        flags[lay["colr"] + k * T + r] = 16
        tf_k[f] += 1

    g3_k = 0   # next block of the chain kernel's gated updates of row k+3 (gated_worker3): tiles (k+3, k+1 .. k+3) get block k

    def g3_can_run():
        k = g3_k
        if k + 3 >= T or chain_k <= k or tf_k[0] <= k:
            return False
        return k == 0 or all(flags[pre3 + 4 * k + j] >= 16 for j in range(3))

    def g3_step():
        nonlocal g3_k
        k = g3_k
        s3 = tile(BASE_S, k + 3, k)
        for c in (k + 1, k + 2, k + 3):
            upd = s3 @ tile(BASE_S, c, k).T
            tile(BASE_L, k + 3, c)[:, :] -= np.tril(upd) if c == k + 3 else upd
        flags[lay["rest"] + k] = 24
        g3_k += 1

    steps = 0
    while True:
        for q in range(NQ):   # a two-piece task at a queue head whose first-stage counters are in: the device would start its first piece now
            if heads[q] < qbeg[q + 1]:
                t = tasks[heads[q]]
                if t["kc_split"] and id(t) not in snap and all(flags[i] >= w for i, w in t["dep"]):
                    K1 = KC * t["kc_split"]
                    assert 0 < t["kc_split"] < t["kc"]
                    a1, b1 = view(t["A"], TILE, K1).copy(), view(t["B"], CT, K1).copy()
                    assert not np.isnan(a1).any() and not np.isnan(b1).any(), "first piece would read an operand before it was written"
                    snap[id(t)] = (a1, b1)
        runnable = [q for q in range(NQ) if heads[q] < qbeg[q + 1] and ready(tasks[heads[q]])]
        can_chain = chain_can_run()
        tfs = [("tf", f) for f in range(NSF) if tf_can_run(f)] + ([("g3", 0)] if g3_can_run() else [])
        if not runnable and not can_chain and not tfs:
            break
        if order == "tasks_first":
            pick = runnable[0] if runnable else ("chain" if can_chain else tfs[-1])
        elif order == "low_priority_first":
            pick = runnable[-1] if runnable else (tfs[0] if tfs else "chain")
        elif order == "chain_first":
            pick = "chain" if can_chain else (tfs[0] if tfs else runnable[0])
        else:
            opts = runnable + (["chain"] if can_chain else []) + tfs
            pick = opts[rng.integers(len(opts))]
        if pick == "chain":
            chain_step()
        elif isinstance(pick, tuple):
            if pick[0] == "g3":
                g3_step()
            else:
                tf_step(pick[1])
        else:
            t_ = tasks[heads[pick]]
            if copied:   # the copy S -> L has run: whatever is left (the inverse) must not touch L any more
                assert all((t_[o] & ~((1 << 44) - 1)) != BASE_L for o in ("A", "B", "C") if t_[o]) and (not t_["P"] or (t_["P"] & ~((1 << 44) - 1)) != BASE_L), \
                    "a task reads or writes L after the chain kernel has ended"
            run_task(t_)
            heads[pick] += 1
        steps += 1
        if not copied and chain_k == T and tf_k == [max(T - 3 - f, 0) for f in range(NSF)] and g3_k == max(T - 3, 0):
            # k_copy_offdiag_tiles runs right behind the chain KERNEL (cholesky_exec), beside what the executor still has to do: the solved
            # panels go home now
            for i in range(T):
                for j in range(i):
                    assert not np.isnan(tile(BASE_S, i, j)).any(), f"S({i}, {j}) not solved when the chain kernel ends"
                    tile(BASE_L, i, j)[:, :] = tile(BASE_S, i, j)
            copied = True
    assert chain_k == T, f"dead-lock: chain stopped at block {chain_k} of {T}, queue heads {heads} of {qbeg}"
    assert tf_k == [max(T - 3 - f, 0) for f in range(NSF)] and g3_k == max(T - 3, 0)
    assert heads == qbeg[1:], f"records left over: heads {heads}, queues {qbeg}"
    assert copied
    L = mats[BASE_L]
    ref = np.linalg.cholesky(K)
    got = np.tril(L[:N, :N])
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-12, err
    if inv_g > 0:   # the inverse queue: W = L^-1 in the lower triangle of W, the same entries in the upper triangle of W'
        if groups and inv_g >= 2:   # (the group form needs groups of at least two rows)
            assert qbeg[4] > qbeg[3] and (qbeg[NQ] > qbeg[NQ - 1]) == (T > inv_g)   # (rounds and products from the second group on)
        else:
            assert qbeg[4] > qbeg[3] and (qbeg[NQ] > qbeg[NQ - 1]) == (T > inv_g + 1)   # (a wave needs a whole chunk of rows above it)
        Wref = np.linalg.inv(ref)
        W = np.tril(mats[BASE_W][:N, :N])
        assert not np.isnan(W).any()
        assert np.abs(W - Wref).max() / np.abs(Wref).max() < 1e-11
        assert np.array_equal(np.triu(mats[BASE_WT][:N, :N]), W.T)
    else:
        assert qbeg[NQ] == qbeg[NQ - 1] and qbeg[4] == qbeg[3]
    return len(tasks)


@pytest.mark.parametrize("order", ["tasks_first", "low_priority_first", "chain_first", "random"])
@pytest.mark.parametrize("T", [2, 3, 4, 5, 9, 14])
def test_executor_records_replay_to_the_cholesky_factor(T, order):
    replay(T, order)


def test_executor_records_random_orders_mid_size():
    for seed in range(3):
        replay(18, "random", seed=seed, inv_g=(2, 3, 8)[seed])


@pytest.mark.parametrize("order", ["tasks_first", "low_priority_first", "chain_first", "random"])
@pytest.mark.parametrize("T,inv_g", [(2, 2), (3, 2), (5, 2), (9, 2), (14, 3), (18, 8), (19, 4)])
def test_executor_records_replay_with_the_inverse_in_group_form(T, inv_g, order):
    replay(T, order, inv_g=inv_g, groups=True)


@pytest.mark.parametrize("T,order", [(9, "random"), (14, "tasks_first"), (14, "low_priority_first"), (17, "chain_first"), (17, "random")])
def test_executor_records_without_the_inverse_queue(T, order):
    """inv_g = 0 (the factorisation alone): up to 56 row tiles the Early sums are TWO-piece tasks since round 6 (their last block is waited
    for inside the task, exec_task_list) -- the replay checks that the first piece reads only final operands and that nothing dead-locks."""
    recs, qbeg, lay = get_tasks(T, TILE * T + 16, 0)
    early = [decode(r) for r in recs[qbeg[2]:qbeg[3]]]
    assert any(t["kc_split"] > 0 for t in early) and any(t["kc_split"] == 0 for t in early)   # (one-block windows stay one piece)
    replay(T, order, inv_g=0)


def test_every_tile_gets_every_block_once():
    """Pure bookkeeping at the largest supported size: per tile, the contraction ranges of bulk + Early + Late are disjoint,
    contiguous from block 0 and end where the chain takes over; per-tile read-modify-write rounds are ordered by `ver`."""
    T = 96
    ld = TILE * T + 16
    recs, qbeg, lay = get_tasks(T, ld)
    cover = {}
    for q in range(NQ):
        for r in recs[qbeg[q]:qbeg[q + 1]]:
            t = decode(r)
            if t["C"] >> 44 == BASE_L >> 44 and t["rmw"]:        # a trailing-matrix tile: which blocks does this record subtract?
                off = (t["C"] - BASE_L) // 8
                i, c = (off // ld) // TILE, (off % ld) // TILE
                kb = (((t["A"] - BASE_S) // 8) % ld) // TILE
                half = ((off % ld) % TILE) // CT
                blocks = list(range(kb, kb + t["kc"] * KC // TILE))
                cover.setdefault((i, c, half), []).extend(blocks)
            elif t["C"] >> 44 == BASE_S >> 44 and (((t["C"] - BASE_S) // 8) // ld) // TILE <= (((t["C"] - BASE_S) // 8) % ld) // TILE and t["rmw"] == 0 \
                    and t["A"] >> 44 == BASE_S >> 44:            # an Early record: P(i, c) lives at the mirror tile (c, i)
                off = (t["C"] - BASE_S) // 8
                c, i = (off // ld) // TILE, (off % ld) // TILE
                kb = (((t["A"] - BASE_S) // 8) % ld) // TILE
                half = ((off % ld) % TILE) // CT
                cover.setdefault((i, c, half), []).extend(range(kb, kb + t["kc"] * KC // TILE))
    assert cover
    for (i, c, half), blocks in cover.items():
        want_last = c - 1 if i >= c + 3 else i - 4   # the chain applies the remaining blocks itself (its window spans three rows)
        assert sorted(blocks) == list(range(0, want_last + 1)), ((i, c, half), sorted(blocks), want_last)


def test_bulk_claims_of_two_tiles_never_wait_for_themselves():
    """From 72 row tiles on the bulk queue is claimed two tiles (four records) at a time.  A claim starts when the counters of ALL its
    records are in, and claims are the consecutive blocks of four records from the queue's start -- so two rounds on ONE tile must never
    share a block (the claim would wait for itself: 200 ms, then the fall-back).  That happens where a group has a single tile left
    (T = 4 m + 9); the library checks the record list and goes back to one tile per claim there (`exec_bulk_stride_ok`)."""
    seen = set()
    for T in range(60, 97):
        ld = TILE * T + 16
        recs, qbeg, lay = get_tasks(T, ld, 8, groups=True)
        C_ = [int(r[2]) for r in recs[qbeg[4]:qbeg[5]]]
        ok = all(len(set(C_[b:b + 4])) == len(C_[b:b + 4]) for b in range(0, len(C_), 4))
        assert lay["bulk_tiles_per_claim"] == (2 if ok else 1), T
        seen.add(ok)
    assert seen == {True, False}   # (both cases occur in the range, so the check is alive)
