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
