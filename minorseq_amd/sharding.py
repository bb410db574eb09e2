"""Window sharding of the reference across ranks and the one exchange of the path.

Columns/codons are independent (doc/JULIET.md:261-264: "Each gene is treated separately"), so rank r owns a
contiguous window of reference columns and calls it alone with the GLOBAL Bonferroni factor; the only exchange
is an all-gather of the fixed-stride variant table.  On GPUs that exchange is `jl_allgather_variants`
(RCCL over xGMI, in the C ABI); this module holds the rank-independent logic — window bounds, table merge — and a
`torch.distributed` all-gather of the same fixed-stride payload that runs on any backend (gloo in the CPU tests).
"""
import numpy as np

from .capi import GENE, VARIANT, VARIANT_CAP


def window_bounds(n_cols_total: int, world: int):
    """[(begin, end)] per rank, 0-based reference columns.  Windows overlap by two columns so that a codon is
    evaluated by exactly one rank — the one whose own range [begin, next begin) holds its first base —
    whatever the reading frame of the gene it belongs to."""
    if world < 1 or n_cols_total < 1:
        raise ValueError("world and n_cols_total must be positive")
    cuts = [(n_cols_total * r) // world for r in range(world + 1)]
    return [(cuts[r], min(n_cols_total, cuts[r + 1] + (2 if r + 1 < world else 0))) for r in range(world)]


def default_n_tests(genes) -> float:
    """Bonferroni factor shared by all ranks: codons over all genes (docs/SPEC.md §5)."""
    genes = np.asarray(genes, dtype=GENE)
    ok = (genes["begin"] > 0) & (genes["end"] > genes["begin"])
    return float(((genes["end"][ok] - genes["begin"][ok]) // 3).sum())


def merge_tables(tables, win_begins):
    """Per-rank tables (window-relative `col`) -> one table in (gene, codon_pos, codon) order, global columns."""
    parts = []
    for t, b in zip(tables, win_begins):
        t = np.array(t, dtype=VARIANT, copy=True)
        t["col"] += np.uint32(b)
        parts.append(t)
    allv = np.concatenate(parts) if parts else np.zeros(0, dtype=VARIANT)
    order = np.lexsort((allv["codon"], allv["codon_pos"], allv["gene"]))
    return allv[order]


def allgather_tables(local_rows, group=None, cap_rows=VARIANT_CAP):
    """All-gather of the fixed-stride variant table with torch.distributed (any backend).
    Payload per rank: cap_rows * 48 bytes + an 8-byte count, exactly what jl_allgather_variants sends."""
    import torch
    import torch.distributed as dist

    local_rows = np.ascontiguousarray(local_rows, dtype=VARIANT)
    if len(local_rows) > cap_rows:
        raise OverflowError(f"{len(local_rows)} rows exceed the all-gather stride {cap_rows}")
    world = dist.get_world_size(group)
    buf = np.zeros(cap_rows * VARIANT.itemsize + 8, dtype=np.uint8)
    buf[: local_rows.nbytes] = local_rows.view(np.uint8)
    buf[-8:] = np.array([len(local_rows)], dtype=np.uint64).view(np.uint8)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    send = torch.from_numpy(buf).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    out = []
    for r in recv:
        a = r.cpu().numpy()
        n = int(a[-8:].view(np.uint64)[0])
        out.append(a[: n * VARIANT.itemsize].view(VARIANT).copy())
    return out
