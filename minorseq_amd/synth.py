"""numpy mirror of csrc/jl_synth.h — deterministic synthetic aligned CCS reads.

The device fills the resident matrix itself (jl_synth_fill); this mirror exists so tests and the
CPU-baseline leg of bench.py can materialise the same reads by-row for the oracle.
"""
from dataclasses import dataclass, field

import numpy as np

_U = np.uint64
N_EDITS = 5
_AA = (41, 65, 181, 190, 215)  # M41L K65R Y181C G190A T215Y
_HAP_OF = (1, 2, 3, 3, 4)      # A.3: Y181C + G190A co-occur


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + _U(0x9E3779B97F4A7C15)
        z = (x ^ (x >> _U(30))) * _U(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U(27))) * _U(0x94D049BB133111EB)
        return z ^ (z >> _U(31))


@dataclass
class SynthParams:
    seed: int = 2
    sub_rate: float = 1.75e-4
    del_rate: float = 1.3e-3
    mask_rate: float = 2.0e-2
    partial_rate: float = 0.0
    minor_permille: tuple = (10, 10, 10, 10)


@dataclass
class Plan:
    seed: int
    n_cols: int
    t_mask: int
    t_del: int
    t_sub: int
    t_partial: int
    cum_permille: tuple
    edit_col: list = field(default_factory=list)
    edit_base: list = field(default_factory=list)
    edit_hap: list = field(default_factory=list)


def reference(seed: int, n_cols: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _U(seed) ^ _U(0x6A756C696574)
        ref = (splitmix64(base + np.arange(n_cols, dtype=np.uint64)) & _U(3)).astype(np.uint8)
    for c in range(0, n_cols - 2, 3):
        cod = 16 * int(ref[c]) + 4 * int(ref[c + 1]) + int(ref[c + 2])
        if cod in (48, 50, 56):  # TAA TAG TGA
            ref[c] = 1
    return ref


def make_plan(sp: SynthParams, n_cols: int, ref: np.ndarray) -> Plan:
    two24 = 16777216.0
    t_mask = int(np.floor(sp.mask_rate * two24))
    t_del = t_mask + int(np.floor(sp.del_rate * two24))
    t_sub = t_del + int(np.floor(sp.sub_rate * two24))
    cum = tuple(int(x) for x in np.cumsum(sp.minor_permille))
    pl = Plan(sp.seed, n_cols, t_mask, t_del, t_sub, int(np.floor(sp.partial_rate * two24)), cum)
    p = n_cols // 3
    for k in range(N_EDITS):
        kk = (_AA[k] * p) // 1000
        col = min(3 * kk + k % 3, n_cols - 1)
        pl.edit_col.append(col)
        pl.edit_base.append((int(ref[col]) + 1 + (k & 1)) & 3)
        pl.edit_hap.append(_HAP_OF[k])
    return pl


def read_draws(pl: Plan, reads: np.ndarray):
    """haplotype, start, end for each read index."""
    reads = np.asarray(reads, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _U(pl.seed) * _U(0x100000001B3)
        uh = splitmix64(base + _U(2) * reads)
        up = splitmix64(base + _U(2) * reads + _U(1))
    v = (uh % _U(1000)).astype(np.int64)
    hap = np.zeros(len(reads), dtype=np.int64)
    c = pl.cum_permille
    hap[v < c[3]] = 4
    hap[v < c[2]] = 3
    hap[v < c[1]] = 2
    hap[v < c[0]] = 1
    q = pl.n_cols // 4 + 1
    partial = (up >> _U(40)).astype(np.int64) < pl.t_partial
    s = np.where(partial, (up & _U(0xFFFF)).astype(np.int64) % q, 0)
    e = np.where(partial, pl.n_cols - ((up >> _U(16)) & _U(0xFFFF)).astype(np.int64) % q, pl.n_cols)
    return hap, s, e


def rows(sp: SynthParams, n_cols: int, read_begin: int, read_end: int, ref: np.ndarray = None) -> np.ndarray:
    """By-row uint8[read_end-read_begin][n_cols] codes of reads [read_begin, read_end)."""
    if ref is None:
        ref = reference(sp.seed, n_cols)
    pl = make_plan(sp, n_cols, ref)
    out = np.empty((read_end - read_begin, n_cols), dtype=np.uint8)
    cols = np.arange(n_cols, dtype=np.uint64)
    chunk = max(1, (1 << 22) // max(n_cols, 1))
    for r0 in range(read_begin, read_end, chunk):
        r1 = min(read_end, r0 + chunk)
        reads = np.arange(r0, r1, dtype=np.uint64)
        hap, s, e = read_draws(pl, reads)
        b = np.broadcast_to(ref.astype(np.uint8), (r1 - r0, n_cols)).copy()
        for k in range(N_EDITS):
            b[hap == pl.edit_hap[k], pl.edit_col[k]] = pl.edit_base[k]
        with np.errstate(over="ignore"):
            u = splitmix64(_U(pl.seed) + _U(0x632BE59BD9B4E019) * (reads[:, None] * _U(n_cols) + cols[None, :] + _U(1)))
        v = (u >> _U(40)).astype(np.int32)
        cell = b
        is_sub = (v >= pl.t_del) & (v < pl.t_sub)
        if is_sub.any():
            us = u[is_sub]
            cell[is_sub] = ((b[is_sub].astype(np.int64) + 1 + ((us >> _U(8)) % _U(3)).astype(np.int64)) & 3).astype(np.uint8)
        cell[v < pl.t_del] = 4
        cell[v < pl.t_mask] = 5
        ci = np.arange(n_cols)[None, :]
        cell[(ci < s[:, None]) | (ci >= e[:, None])] = 6
        out[r0 - read_begin:r1 - read_begin] = cell
    return out


def raw_records(seed, n_reads, n_cols, ref_seed=None, path=None, extra=()):
    """The synthetic reads as BAM-style record arrays (what a decoder hands to jl_records_append), written by the C++
    generator `juliet-synth --raw-out` (minorseq_amd/host/synth_bam.cpp: the same jl_synth.h cell function as everything
    else) and read back: dict(pos, cigar, cig_off, seq4, seq_off, qual, qual_off)."""
    import os
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "juliet-synth")
    own = path is None
    if own:
        fd, path = tempfile.mkstemp(suffix=".jlraw")
        os.close(fd)
    try:
        cmd = [exe, "--reads", str(n_reads), "--cols", str(n_cols), "--seed", str(seed), "--raw-out", path, *extra]
        if ref_seed is not None:
            cmd += ["--ref-seed", str(ref_seed)]
        subprocess.check_call(cmd)
        hdr = np.fromfile(path, dtype=np.uint64, count=8)
        assert hdr[0] == 0x4A4C524157303031, "not a juliet-synth --raw-out file"
        n, n_cig, n_seq, n_qual = (int(x) for x in hdr[1:5])
        off = 64
        out = {}
        for name, dt, cnt in (("pos", np.int32, n), ("cigar", np.uint32, n_cig), ("cig_off", np.uint64, n + 1), ("seq4", np.uint8, n_seq),
                              ("seq_off", np.uint64, n + 1), ("qual", np.uint8, n_qual), ("qual_off", np.uint64, n + 1)):
            out[name] = np.fromfile(path, dtype=dt, count=cnt, offset=off)
            off += (cnt * np.dtype(dt).itemsize + 7) // 8 * 8
        return out
    finally:
        if own and os.path.exists(path):
            os.remove(path)
