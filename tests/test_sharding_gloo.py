"""N > 1 path on CPU: world_size-2 gloo.  Window bounds, the fixed-stride all-gather and the merge are the
product code (minorseq_amd/sharding.py); the per-window compute is done by the oracle here because there is no
GPU in this container — on the GPU box the same exchange is jl_allgather_variants over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
from minorseq_amd import capi, msa, sharding, synth


def test_window_bounds_cover_every_codon_once():
    for total in (10, 299, 3000, 9719):
        for world in (1, 2, 3, 8):
            wb = sharding.window_bounds(total, world)
            assert wb[0][0] == 0 and wb[-1][1] == total
            for frame in range(3):
                owners = {}
                for r, (b, e) in enumerate(wb):
                    for c in range(frame, total - 2, 3):
                        if b <= c and c + 2 < e:
                            owners.setdefault(c, []).append(r)
                assert all(len(v) == 1 for v in owners.values())
                assert sorted(owners) == list(range(frame, total - 2, 3))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N, L = 2500, 600
GENES = np.array([(1, 601), (2, 300), (301, 598)], dtype=capi.GENE)


def _data():
    sp = synth.SynthParams(seed=17, sub_rate=0.012, minor_permille=(70, 60, 50, 40), partial_rate=0.1)
    ref = synth.reference(sp.seed, L)
    return ref, synth.rows(sp, L, 0, N, ref)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.load()
        ref, rows = _data()
        wb = sharding.window_bounds(L, world)
        b, e = wb[rank]
        prm = oracle_lib.default_params(n_tests=sharding.default_n_tests(GENES))   # GLOBAL Bonferroni factor
        local = orc.call(rows[:, b:e], GENES, win_begin=b, refseq=ref, params=prm)
        tables = sharding.allgather_tables(local)
        merged = sharding.merge_tables(tables, [w[0] for w in wb])
        if rank == 0:
            q.put((merged.tobytes(), [len(t) for t in tables]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_call_equals_unsharded(world, oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    raw, counts = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    merged = np.frombuffer(raw, dtype=capi.VARIANT)
    ref, rows = _data()
    full = oracle.call(rows, GENES, refseq=ref)
    assert len(full) >= 8 and sum(counts) == len(full) and min(counts) > 0
    assert (merged == full).all()


def test_allgather_overflow_is_loud():
    with pytest.raises(OverflowError):
        sharding.allgather_tables(np.zeros(5, dtype=capi.VARIANT), cap_rows=4)


def _xwin_worker(rank, world, port, q):
    """configs[3]/[4] on CPU ranks: reads span all windows.  Call per window (oracle), all-gather of the table (gloo),
    the product's host-side plan of the column exchange (jl_xwin_plan: positions, remapped table, owner rank of each
    position), the owner's three columns broadcast over gloo where the GPU build uses ncclBroadcast, phasing on the
    compact matrix."""
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.load()
        ref, rows = _data()
        wb = sharding.window_bounds(L, world)
        b, e = wb[rank]
        mine = np.ascontiguousarray(rows[:, b:e])                   # this rank only ever touches its own window
        prm = oracle_lib.default_params(n_tests=sharding.default_n_tests(GENES))
        local = orc.call(mine, GENES, win_begin=b, refseq=ref, params=prm)
        merged = sharding.merge_tables(sharding.allgather_tables(local), [w[0] for w in wb])
        remapped, pos, owner = capi.xwin_plan([w[0] for w in wb], [w[1] - w[0] for w in wb], merged)
        assert (owner >= 0).all() and len(pos) == len(np.unique(merged["col"]))
        compact = np.empty((N, 3 * len(pos)), dtype=np.uint8)
        for k, (c, w) in enumerate(zip(pos, owner)):
            buf = torch.from_numpy(np.ascontiguousarray(mine[:, c - b: c - b + 3]) if w == rank else np.zeros((N, 3), dtype=np.uint8))
            dist.broadcast(buf, int(w))
            compact[:, 3 * k: 3 * k + 3] = buf.numpy()
        ph = orc.phase(compact, remapped)
        if rank == world - 1:     # any rank: phasing is replicated
            q.put((merged.tobytes(), pos.tobytes(), owner.tobytes(), ph["summary"], ph["hap_count"].tobytes(), ph["read_hap"].tobytes(),
                   ph["hit"].tobytes()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_cross_window_phasing_plan_and_exchange(world, oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xwin_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged_b, pos_b, owner_b, summary, hc, rh, hit = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, rows = _data()
    full = oracle.call(rows, GENES, refseq=ref)
    exp = oracle.phase(rows, full)
    assert (np.frombuffer(merged_b, dtype=capi.VARIANT) == full).all()
    pos, owner = np.frombuffer(pos_b, dtype=np.uint32), np.frombuffer(owner_b, dtype=np.int32)
    assert (pos == exp["pos_cols"]).all()
    wb = sharding.window_bounds(L, world)
    assert all(wb[w][0] <= c and c + 3 <= wb[w][1] for c, w in zip(pos, owner))
    assert len(set(owner.tolist())) >= 2                         # the positions really live on different ranks
    assert summary == exp["summary"]
    assert (np.frombuffer(hc, dtype=np.uint32) == exp["hap_count"]).all()
    assert (np.frombuffer(rh, dtype=np.uint16) == exp["read_hap"]).all()
    assert (np.frombuffer(hit, dtype=np.uint8).reshape(exp["hit"].shape) == exp["hit"]).all()


def _slice_groups(compact):
    """CPU stand-in for jl_phase_groups_async / _fetch on one slice of the reads: flags, patterns of the clean reads, the
    groups (pattern, count) in first-occurrence order like a hash table would hold them, and the partial summary."""
    vp = compact.shape[1] // 3
    gap = (compact == 4).any(axis=1)
    het = (compact == 5).any(axis=1)
    par = (compact == 6).any(axis=1)
    dirty = gap | het | par
    clean = compact[~dirty].astype(np.uint32)
    codes = (clean[:, 0::3] * 16 + clean[:, 1::3] * 4 + clean[:, 2::3]).astype(np.uint8) if vp else np.zeros((len(clean), 0), dtype=np.uint8)
    uniq, first, inverse, counts = np.unique(codes, axis=0, return_index=True, return_inverse=True, return_counts=True)
    order = np.argsort(first)                       # not sorted by pattern: the merge must not rely on any order
    rank_of = np.empty(len(order), dtype=np.int64)
    rank_of[order] = np.arange(len(order))
    summary = dict(reported_reads=0, insufficient_reads=int((~dirty).sum()), damaged_reads=int(dirty.sum()), marginal_gap=int(gap.sum()),
                   marginal_heteroduplex=int(het.sum()), marginal_partial=int(par.sum()), n_positions=vp, n_haplotypes=0)
    group_of_clean = rank_of[np.asarray(inverse).reshape(-1)]
    return dict(patterns=uniq[order], counts=counts[order].astype(np.uint32), summary=summary), dirty, group_of_clean


def _run_ops(ops, rank, planes_window, win_begin, pos, n_mine):
    """Executes one rank's op list (jl_xwin_slice_plan) with gloo point-to-point calls IN LIST ORDER: what
    jl_xwin_phase_sharded does with the pack kernel, ncclSend and ncclRecv.  `planes_window`: this rank's window in the
    device layout (msa.pack_planes: [column][plane][bytes]).  Returns the compact matrix of the rank's slice, by rows."""
    import torch
    stride_me = msa.plane_stride(n_mine) if n_mine else 0
    pad_rows = np.array([0x00, 0xFF, 0xFF] * 3, dtype=np.uint8)   # 'not covered' = code 6: plane 0 clear, planes 1 and 2 set
    compact = np.tile(np.repeat(pad_rows, stride_me), len(pos)) if stride_me else np.zeros(0, dtype=np.uint8)
    pending, keep = [], []

    def message(op):   # slice [read_begin, +n_reads) of the 9 * k_count plane rows of the owned columns, each padded to dst_stride
        msg = np.tile(pad_rows[:, None], (op["k_count"], op["dst_stride"]))
        b0, nb = op["read_begin"] // 8, (op["n_reads"] + 7) // 8
        for i in range(op["k_count"]):
            c = int(pos[op["k_begin"] + i]) - win_begin
            msg[9 * i: 9 * i + 9, :nb] = planes_window[c: c + 3, :, b0: b0 + nb].reshape(9, nb)
        assert msg.size == op["bytes"] and op["dst_stride"] == msa.plane_stride(op["n_reads"])
        return msg.reshape(-1)

    for op in ops:
        if op["op"] == capi.XWIN_OP_LOCAL:
            compact[op["dst_offset"]: op["dst_offset"] + op["bytes"]] = message(op)
        elif op["op"] == capi.XWIN_OP_SEND:
            t = torch.from_numpy(message(op).copy())
            keep.append(t)
            pending.append((dist.isend(t, op["peer"]), None, None))
        else:
            t = torch.empty(op["bytes"], dtype=torch.uint8)
            pending.append((dist.irecv(t, op["peer"]), t, op))
    for req, t, op in pending:
        req.wait()
        if t is not None:
            compact[op["dst_offset"]: op["dst_offset"] + op["bytes"]] = t.numpy()
    if not n_mine:
        return np.zeros((0, 3 * len(pos)), dtype=np.uint8)
    return msa.unpack_planes(compact.reshape(3 * len(pos), 3, stride_me), n_mine)


def _sharded_phase_worker(rank, world, port, q):
    """SURVEY §8e option A on CPU ranks: call per window (oracle), all-gather of the table (gloo), the product's schedule
    of the column-slice exchange (jl_xwin_slice_plan) EXECUTED op by op over gloo send / recv, every rank groups its own
    slice of the reads, the group tables are all-gathered and merged and the haplotypes selected by the product's C
    functions (jl_merge_tables / jl_merge_groups / jl_select_haplotypes through sharding.py), and each rank maps its reads."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.load()
        ref, rows = _data()
        wb = sharding.window_bounds(L, world)
        b, e = wb[rank]
        mine = np.ascontiguousarray(rows[:, b:e])
        prm = oracle_lib.default_params(n_tests=sharding.default_n_tests(GENES))
        local = orc.call(mine, GENES, win_begin=b, refseq=ref, params=prm)
        merged = sharding.merge_tables(sharding.allgather_tables(local), [w[0] for w in wb])
        wbeg, wnc = [w[0] for w in wb], [w[1] - w[0] for w in wb]
        remapped, pos, owner = capi.xwin_plan(wbeg, wnc, merged)
        sb = sharding.read_slices(N, world)
        assert sb[0] == 0 and sb[-1] == N and all(sb[k] % 256 == 0 for k in range(world) if sb[k + 1] > sb[k])
        n_mine = sb[rank + 1] - sb[rank]
        ops = capi.xwin_slice_plan(wbeg, wnc, list(range(world)), merged, sb, world, rank)
        compact = _run_ops(ops, rank, msa.pack_planes(mine), b, pos, n_mine)
        table, dirty, group_of_clean = _slice_groups(compact)
        tables = sharding.allgather_groups(table)
        patterns, counts, index = sharding.merge_groups(tables)
        ph = sharding.select_haplotypes(patterns, counts, remapped, 3 * np.arange(len(pos)), 10, [t["summary"] for t in tables])
        ids = np.full(n_mine, 0xFFFF, dtype=np.uint16)
        ids[~dirty] = ph["hap_of_merged"][index[rank]][group_of_clean].astype(np.uint16)
        gathered, all_ops = [None] * world, [None] * world
        dist.all_gather_object(gathered, ids)
        dist.all_gather_object(all_ops, ops)
        if rank == 0:
            q.put((ph["summary"], ph["hap_count"].tobytes(), ph["hap_pattern"].tobytes(), ph["hit"].tobytes(), ph["cooc"].tobytes(),
                   np.concatenate(gathered).tobytes(), merged.tobytes(), all_ops, owner.tolist()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _check_pairing(all_ops, world):
    """Every send has exactly one matching receive (same byte count) in the peer's list, at most one message per ordered
    pair of ranks, and both sides meet their peers in ascending order — the order RCCL pairs point-to-point calls by."""
    sends, recvs = {}, {}
    for r, ops in enumerate(all_ops):
        peers = [o["peer"] for o in ops if o["op"] != capi.XWIN_OP_LOCAL]
        assert peers == sorted(peers)
        assert sum(1 for o in ops if o["op"] == capi.XWIN_OP_LOCAL) <= 1
        for o in ops:
            if o["op"] == capi.XWIN_OP_SEND:
                assert (r, o["peer"]) not in sends
                sends[(r, o["peer"])] = o
            elif o["op"] == capi.XWIN_OP_RECV:
                assert (o["peer"], r) not in recvs
                recvs[(o["peer"], r)] = o
    assert sorted(sends) == sorted(recvs)
    for key, snd in sends.items():
        rcv = recvs[key]
        assert all(snd[f] == rcv[f] for f in ("bytes", "k_begin", "k_count", "read_begin", "n_reads", "dst_stride", "dst_offset"))
    return len(sends)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_phasing_sharded_by_reads_over_gloo(world, oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_phase_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    summary, hc, hp, hit, cooc, rh, merged_b, all_ops, owner = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref, rows = _data()
    full = oracle.call(rows, GENES, refseq=ref)
    assert np.frombuffer(merged_b, dtype=capi.VARIANT).tobytes() == full.tobytes()
    exp = oracle.phase(rows, full)
    assert summary == exp["summary"] and summary["n_haplotypes"] >= 3
    h, vp, nv = summary["n_haplotypes"], summary["n_positions"], len(full)
    assert (np.frombuffer(hc, dtype=np.uint32) == exp["hap_count"]).all()
    assert (np.frombuffer(hp, dtype=np.uint8).reshape(h, vp) == exp["hap_pattern"]).all()
    assert (np.frombuffer(hit, dtype=np.uint8).reshape(nv, h) == exp["hit"]).all()
    assert (np.frombuffer(cooc, dtype=np.uint32).reshape(nv, nv) == exp["cooc"]).all()
    assert (np.frombuffer(rh, dtype=np.uint16) == exp["read_hap"]).all()
    # the schedule: owners with positions send to every peer that has reads, nothing else travels
    n_msgs = _check_pairing(all_ops, world)
    owners = set(owner)
    sb = sharding.read_slices(N, world)
    readers = {s for s in range(world) if sb[s + 1] > sb[s]}
    assert n_msgs == sum(1 for o in owners for s in readers if s != o)
    assert len(owners) >= 2


def test_read_slices_and_group_merge_edge_cases():
    assert sharding.read_slices(1000, 8) == [0, 256, 512, 768, 1000, 1000, 1000, 1000, 1000]
    assert sharding.read_slices(4096, 4) == [0, 1024, 2048, 3072, 4096]
    assert sharding.read_slices(1, 3) == [0, 1, 1, 1]
    a = dict(patterns=np.array([[3, 1], [0, 2]], dtype=np.uint8), counts=np.array([4, 6], dtype=np.uint32))
    b = dict(patterns=np.zeros((0, 2), dtype=np.uint8), counts=np.zeros(0, dtype=np.uint32))
    c = dict(patterns=np.array([[0, 2], [3, 1], [3, 0]], dtype=np.uint8), counts=np.array([5, 6, 9], dtype=np.uint32))
    p, n, idx = sharding.merge_groups([a, b, c])
    assert p.tolist() == [[0, 2], [3, 0], [3, 1]] and n.tolist() == [11, 9, 10]
    assert idx[0].tolist() == [2, 0] and len(idx[1]) == 0 and idx[2].tolist() == [0, 2, 1]
    v = np.zeros(2, dtype=capi.VARIANT)
    v["col"] = [0, 3]
    v["codon"] = [3, 2]
    ph = sharding.select_haplotypes(p, n, v, [0, 3], 10)
    # 11 reads of (0,2) first, then the 10 of (3,1); the 9 of (3,0) are insufficient
    assert ph["hap_count"].tolist() == [11, 10] and ph["hap_pattern"].tolist() == [[0, 2], [3, 1]]
    assert ph["hap_of_merged"].tolist() == [0, sharding.HAP_INSUFFICIENT, 1]
    assert ph["hit"].tolist() == [[0, 1], [1, 0]] and ph["cooc"].tolist() == [[10, 0], [0, 11]]
    assert ph["summary"]["reported_reads"] == 21 and ph["summary"]["insufficient_reads"] == 9
    # equal counts: pattern ascending
    ph = sharding.select_haplotypes(np.array([[1, 9], [1, 2], [0, 7]], dtype=np.uint8), [12, 12, 12], v[:1], [0, 3], 10)
    assert ph["hap_pattern"].tolist() == [[0, 7], [1, 2], [1, 9]] and ph["hap_of_merged"].tolist() == [2, 1, 0]
    # positions but not one clean read (every read flagged): no groups, the positions still count
    p0, n0, _ = sharding.merge_groups([b])
    ph = sharding.select_haplotypes(p0, n0, v, [0, 3], 10, [dict(damaged_reads=7, marginal_gap=7, marginal_heteroduplex=0, marginal_partial=1)])
    assert ph["summary"]["n_positions"] == 2 and ph["summary"]["damaged_reads"] == 7 and ph["hit"].shape == (2, 0)
    # nothing to phase
    p0, n0, _ = sharding.merge_groups([b])
    ph = sharding.select_haplotypes(p0, n0, v[:0], [], 10)
    assert ph["summary"]["n_haplotypes"] == 0 and ph["hap_count"].tolist() == []
