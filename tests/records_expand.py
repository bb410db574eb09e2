"""BAM-style record arrays -> the by-row symbol matrix of a column window, in numpy: the CPU statement of what the device
ingest must produce (test infrastructure, like oracle/; nothing of the product imports it).

Behaviour: doc/JULIET.md:26-27 (insertions are dropped, deletions show as '-'), :53 (PacBio cigars are = X I D S H N P; M is an
error), :256-259 (a base below the QV threshold shows as N).  Symbols: A C G T = 0..3, '-' = 4, N = 5, not covered = 6.
Every base that is not exactly A, C, G or T in BAM's 4-bit alphabet is a filtered base (N).

Vectorised over the cigar ops of a slab of reads (a slab = `slab_cells` matrix cells at most), so that 3 x 10^8 cells take
seconds; `expand_reference` below is the same thing as three nested Python loops, for pinning this file on small cases."""
import numpy as np

_REF_OPS = np.zeros(16, dtype=bool)
_REF_OPS[[2, 3, 7, 8]] = True          # D N = X consume the reference
_QRY_OPS = np.zeros(16, dtype=bool)
_QRY_OPS[[1, 4, 7, 8]] = True          # I S = X consume the query
_BASE_LUT = np.full(16, 5, dtype=np.uint8)
_BASE_LUT[[1, 2, 4, 8]] = [0, 1, 2, 3]


def expand(rec, n_cols, win_begin=0, min_qv=0, read_begin=0, read_end=None, slab_cells=1 << 25):
    """uint8[read_end - read_begin][n_cols].  rec: dict(pos, cigar, cig_off, seq4, seq_off[, qual, qual_off])."""
    pos, cigar, cig_off, seq4, seq_off = (np.asarray(rec[k]) for k in ("pos", "cigar", "cig_off", "seq4", "seq_off"))
    qual = rec.get("qual") if min_qv else None
    qual_off = np.asarray(rec["qual_off"]).astype(np.int64) if qual is not None else None
    n = len(pos)
    read_end = n if read_end is None else read_end
    out = np.full((read_end - read_begin, n_cols), 6, dtype=np.uint8)
    cig_off = cig_off.astype(np.int64)
    seq_off = seq_off.astype(np.int64)
    slab = max(1, slab_cells // max(n_cols, 1))
    for r0 in range(read_begin, read_end, slab):
        r1 = min(read_end, r0 + slab)
        k0, k1 = int(cig_off[r0]), int(cig_off[r1])
        if k1 == k0:
            continue
        words = cigar[k0:k1].astype(np.int64)
        op, ln = words & 15, words >> 4
        if (op == 0).any():
            raise ValueError("cigar M")
        n_ops = np.diff(cig_off[r0:r1 + 1])
        rd = np.repeat(np.arange(r0, r1), n_ops)                      # read of every op
        rl = np.where(_REF_OPS[op], ln, 0)
        ql = np.where(_QRY_OPS[op], ln, 0)
        rc, qc = np.cumsum(rl), np.cumsum(ql)
        first = cig_off[r0:r1] - k0                                    # index of every read's first op in the slab
        has = n_ops > 0
        r_base = np.zeros(r1 - r0, dtype=np.int64)
        q_base = np.zeros(r1 - r0, dtype=np.int64)
        r_base[has] = (rc - rl)[first[has]]
        q_base[has] = (qc - ql)[first[has]]
        r_start = rc - rl - r_base[rd - r0] + pos[rd].astype(np.int64) - win_begin   # window column of the op's first base
        q_start = qc - ql - q_base[rd - r0]
        # deletions: '-' over their columns
        for code, sel in ((4, op == 2), (None, (op == 7) | (op == 8))):
            idx = np.nonzero(sel & (ln > 0))[0]
            if len(idx) == 0:
                continue
            l = ln[idx]
            tot = int(l.sum())
            ramp = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(l) - l, l)
            col = np.repeat(r_start[idx], l) + ramp
            row = np.repeat(rd[idx], l)
            keep = (col >= 0) & (col < n_cols)
            if code is not None:
                out[row[keep] - read_begin, col[keep]] = code
                continue
            q = (np.repeat(q_start[idx], l) + ramp)[keep]
            row, col = row[keep], col[keep]
            byte = seq4[seq_off[row] + (q >> 1)]
            nib = np.where(q & 1, byte & 15, byte >> 4)
            sym = _BASE_LUT[nib]
            if qual is not None:
                qv = qual[qual_off[row] + q]
                sym = np.where((qv != 0xFF) & (qv < min_qv), 5, sym).astype(np.uint8)
            out[row - read_begin, col] = sym
    return out


def expand_reference(rec, n_cols, win_begin=0, min_qv=0):
    """The same matrix op by op, base by base (small cases only)."""
    pos, cigar, cig_off, seq4, seq_off = (rec[k] for k in ("pos", "cigar", "cig_off", "seq4", "seq_off"))
    n = len(pos)
    out = np.full((n, n_cols), 6, dtype=np.uint8)
    for r in range(n):
        c, q = int(pos[r]) - win_begin, 0
        for w in cigar[int(cig_off[r]):int(cig_off[r + 1])]:
            op, ln = int(w) & 15, int(w) >> 4
            if op == 0:
                raise ValueError("cigar M")
            for _ in range(ln):
                if op in (7, 8):
                    if 0 <= c < n_cols:
                        byte = int(seq4[int(seq_off[r]) + (q >> 1)])
                        sym = int(_BASE_LUT[byte & 15 if q & 1 else byte >> 4])
                        if min_qv and rec.get("qual") is not None:
                            qv = int(rec["qual"][int(rec["qual_off"][r]) + q])
                            if qv != 0xFF and qv < min_qv:
                                sym = 5
                        out[r, c] = sym
                    c += 1
                    q += 1
                elif op == 2:
                    if 0 <= c < n_cols:
                        out[r, c] = 4
                    c += 1
                elif op == 3:
                    c += 1
                elif op in (1, 4):
                    q += 1
    return out
