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
    """Per-rank tables (window-relative `col`) -> one table in (gene, codon_pos, codon) order, global columns.
    The product code is jl_merge_tables (C ABI, host only); this is its caller."""
    from . import capi
    return capi.merge_tables(tables, win_begins)


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


# ---------------------------------------------------------------------------------------------------------------------
# Phasing sharded by reads (SURVEY.md §8e option A).  Rank s groups the reads of its slice (jl_phase_groups_async);
# what follows is the rank-independent part: slice bounds, the merge of the exported group tables and the selection
# of docs/SPEC.md §8 on the merged groups.  A few thousand rows at most: host work.
HAP_INSUFFICIENT = 0xFFFE
MAX_HAPLOTYPES = 702


def read_slices(n_reads: int, world: int):
    """world + 1 slice starts: rank s phases reads [b[s], b[s+1]).  Starts are multiples of 256 reads (a 128-byte line
    of every column), the last entry is n_reads."""
    per = -(-n_reads // world)          # ceil
    per = -(-per // 256) * 256          # up to a multiple of 256
    return [min(n_reads, s * per) for s in range(world)] + [n_reads]


def merge_groups(tables):
    """tables: per rank dict(patterns uint8[G][Vp], counts uint32[G]) as jl_phase_groups_fetch returns them.
    -> (patterns uint8[M][Vp], counts int64[M], index): the distinct patterns in ascending order (codon codes compared
    position by position), their summed counts, and per rank the merged row of each of its groups.
    The product code is jl_merge_groups (C ABI, host only); this is its caller."""
    from . import capi
    return capi.merge_groups(tables)


def select_haplotypes(patterns, counts, variants, pos_cols, min_reads=10, partials=()):
    """docs/SPEC.md §8 on merged groups: reported = count >= min_reads, ordered by (count desc, pattern asc), at most 702;
    hit, co-occurrence and the read categories (the partial summaries carry each slice's damaged reads and marginals).
    `variants`: the remapped table (col = 3 * position index); pos_cols: the compact columns of the positions.
    -> dict like Juliet.phase_fetch (no read_hap) + hap_of_merged int64[M] (HAP_INSUFFICIENT where not reported).
    The product code is jl_select_haplotypes (C ABI, host only); this is its caller."""
    from . import capi
    return capi.select_haplotypes(patterns, counts, variants, pos_cols, min_reads, partials)


def allgather_groups(local, group=None):
    """All-gather of one rank's exported groups (patterns, counts, partial summary) with torch.distributed — the control
    plane: a few KB per rank.  Returns the list of every rank's table."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    out = [None] * world
    dist.all_gather_object(out, dict(patterns=np.asarray(local["patterns"]), counts=np.asarray(local["counts"]),
                                     summary=dict(local["summary"])), group=group)
    return out
