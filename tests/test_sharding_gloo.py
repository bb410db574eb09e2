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
from minorseq_amd import capi, sharding, synth


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
