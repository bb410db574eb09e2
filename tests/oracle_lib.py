"""ctypes view of oracle/libjuliet_oracle.so — the CPU restatement, test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libjuliet_oracle.so")

MAX_HAP = 702
HAP_INSUFFICIENT = 0xFFFE
HAP_DAMAGED = 0xFFFF

GENE = np.dtype([("begin", "<u4"), ("end", "<u4")])
VARIANT = np.dtype([("gene", "<u4"), ("codon_pos", "<u4"), ("col", "<u4"), ("ref_codon", "u1"), ("codon", "u1"),
                    ("flags", "<u2"), ("count", "<u4"), ("coverage", "<u4"), ("expected", "<u4"), ("pad_", "<u4"),
                    ("p_value", "<f8"), ("log_p", "<f8")])
SUMMARY = np.dtype([(n, "<u4") for n in ("reported_reads", "insufficient_reads", "damaged_reads", "marginal_gap",
                                         "marginal_heteroduplex", "marginal_partial", "n_positions",
                                         "n_haplotypes")])


class ErrorModel(C.Structure):
    _fields_ = [("match", C.c_double), ("substitution", C.c_double), ("deletion", C.c_double)]


class Params(C.Structure):
    _fields_ = [("alpha", C.c_double), ("n_tests", C.c_double), ("err", ErrorModel), ("expected_round", C.c_int32),
                ("tail", C.c_int32)]


def default_params(n_tests=0.0, alpha=0.01, match=0.998826, substitution=5.8e-5, deletion=1.0e-3, tail=0):
    return Params(alpha, n_tests, ErrorModel(match, substitution, deletion), 0, tail)


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_fisher.restype = C.c_double
        lib.orc_fisher.argtypes = [C.c_uint32] * 4 + [C.c_int32, C.POINTER(C.c_double)]
        lib.orc_default_n_tests.restype = C.c_double
        lib.orc_expected.restype = C.c_uint32
        lib.orc_expected.argtypes = [C.POINTER(Params), C.c_uint32, C.c_int, C.c_int]
        lib.orc_set_threads.argtypes = [C.c_int]
        lib.orc_set_threads.restype = None
        assert lib.orc_sizeof_variant() == VARIANT.itemsize
        assert lib.orc_sizeof_params() == C.sizeof(Params)

    def set_columns(self, rows):
        """Registers a column-major copy of `rows` (or forgets it: None) for the multi-threaded counting sweeps."""
        if rows is None:
            self._cols = None
            self.lib.orc_set_columns(None, C.c_uint64(0), C.c_uint32(0))
            return
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        self._cols = np.ascontiguousarray(rows.T)
        self.lib.orc_set_columns(self._cols.ctypes.data_as(C.c_void_p), C.c_uint64(rows.shape[0]), C.c_uint32(rows.shape[1]))

    def set_threads(self, n):
        """1 = the plain restatement; more = OpenMP over reads in the two counting sweeps (same results)."""
        self.lib.orc_set_threads(int(n))

    def max_threads(self):
        return int(self.lib.orc_max_threads())

    def pileup(self, msa):
        msa = np.ascontiguousarray(msa, dtype=np.uint8)
        n, l = msa.shape
        out = np.zeros((l, 6), dtype=np.uint32)
        self.lib.orc_pileup(msa.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(l),
                            out.ctypes.data_as(C.c_void_p))
        return out

    def codon_hist(self, msa, start_cols):
        msa = np.ascontiguousarray(msa, dtype=np.uint8)
        n, l = msa.shape
        sc = np.ascontiguousarray(start_cols, dtype=np.uint32)
        hist = np.zeros((len(sc), 64), dtype=np.uint32)
        cov = np.zeros(len(sc), dtype=np.uint32)
        self.lib.orc_codon_hist(msa.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(l),
                                sc.ctypes.data_as(C.c_void_p), C.c_uint32(len(sc)),
                                hist.ctypes.data_as(C.c_void_p), cov.ctypes.data_as(C.c_void_p))
        return hist, cov

    def fisher(self, a, b, c, d, tail=0):
        lp = C.c_double()
        p = self.lib.orc_fisher(a, b, c, d, tail, C.byref(lp))
        return p, lp.value

    def expected(self, prm, cov, ref, j):
        return self.lib.orc_expected(C.byref(prm), cov, ref, j)

    def call(self, msa, genes, win_begin=0, refseq=None, params=None, cap=65536):
        msa = np.ascontiguousarray(msa, dtype=np.uint8)
        n, l = msa.shape
        genes = np.ascontiguousarray(genes, dtype=GENE)
        prm = params or default_params()
        out = np.zeros(cap, dtype=VARIANT)
        n_out = C.c_uint32()
        if refseq is not None:
            refseq = np.ascontiguousarray(refseq, dtype=np.uint8)
            rp, rl = refseq.ctypes.data_as(C.c_void_p), len(refseq)
        else:
            rp, rl = None, 0
        rc = self.lib.orc_call(msa.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(l), C.c_uint32(win_begin),
                               genes.ctypes.data_as(C.c_void_p), C.c_uint32(len(genes)), rp, C.c_uint32(rl),
                               C.byref(prm), out.ctypes.data_as(C.c_void_p), C.c_uint32(cap), C.byref(n_out))
        assert rc == 0, "oracle variant table overflow"
        return out[: n_out.value].copy()

    def phase(self, msa, variants, min_reads=10):
        msa = np.ascontiguousarray(msa, dtype=np.uint8)
        n, l = msa.shape
        variants = np.ascontiguousarray(variants, dtype=VARIANT)
        nv = len(variants)
        summ = np.zeros(1, dtype=SUMMARY)
        pos_cols = np.zeros(max(nv, 1), dtype=np.uint32)
        hap_count = np.zeros(MAX_HAP, dtype=np.uint32)
        hap_first = np.zeros(MAX_HAP, dtype=np.uint32)
        hap_pattern = np.zeros((MAX_HAP, max(nv, 1)), dtype=np.uint8)
        hit = np.zeros((max(nv, 1), MAX_HAP), dtype=np.uint8)
        read_hap = np.zeros(n, dtype=np.uint16)
        cooc = np.zeros((max(nv, 1), max(nv, 1)), dtype=np.uint32)
        vp = C.c_void_p
        self.lib.orc_phase(msa.ctypes.data_as(vp), C.c_uint64(n), C.c_uint32(l), variants.ctypes.data_as(vp),
                           C.c_uint32(nv), C.c_uint32(min_reads), summ.ctypes.data_as(vp),
                           pos_cols.ctypes.data_as(vp), hap_count.ctypes.data_as(vp), hap_first.ctypes.data_as(vp),
                           hap_pattern.ctypes.data_as(vp), hit.ctypes.data_as(vp), read_hap.ctypes.data_as(vp),
                           cooc.ctypes.data_as(vp))
        s = summ[0]
        h, p = int(s["n_haplotypes"]), int(s["n_positions"])
        return dict(summary={k: int(s[k]) for k in SUMMARY.names}, pos_cols=pos_cols[:p].copy(),
                    hap_count=hap_count[:h].copy(), hap_first=hap_first[:h].copy(),
                    hap_pattern=hap_pattern[:h, :p].copy(), hit=hit[:nv, :h].copy(), read_hap=read_hap,
                    cooc=cooc[:nv, :nv].copy())


def _insertions(self, n_cols, win_begin, pos, cigar, cig_off, seq4, seq_off):
    """orc_insertions: (len_hist[n_cols][32], base_counts[n_cols][30][4]) of aligned records (BAM-decoded arrays)."""
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    cig_off = np.ascontiguousarray(cig_off, dtype=np.uint64)
    seq4 = np.ascontiguousarray(seq4, dtype=np.uint8)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    lh = np.zeros((n_cols, 32), dtype=np.uint32)
    bc = np.zeros((n_cols, 30, 4), dtype=np.uint32)
    vp = C.c_void_p
    self.lib.orc_insertions(C.c_uint64(len(pos)), C.c_uint32(n_cols), C.c_uint32(win_begin), pos.ctypes.data_as(vp),
                            cigar.ctypes.data_as(vp), cig_off.ctypes.data_as(vp), seq4.ctypes.data_as(vp), seq_off.ctypes.data_as(vp),
                            lh.ctypes.data_as(vp), bc.ctypes.data_as(vp))
    return lh, bc


def _fuse(self, col_counts, len_hist, base_counts, min_frac=0.5, min_distance=10):
    """orc_fuse: the consensus string of a window (doc/FUSE.md:17-20)."""
    col_counts = np.ascontiguousarray(col_counts, dtype=np.uint32)
    n_cols = len(col_counts)
    out = C.create_string_buffer(n_cols * 31 + 1)
    vp = C.c_void_p
    self.lib.orc_fuse.restype = C.c_uint32
    lh = None if len_hist is None else np.ascontiguousarray(len_hist, dtype=np.uint32)
    bc = None if base_counts is None else np.ascontiguousarray(base_counts, dtype=np.uint32)
    n = self.lib.orc_fuse(C.c_uint32(n_cols), col_counts.ctypes.data_as(vp), None if lh is None else lh.ctypes.data_as(vp),
                          None if bc is None else bc.ctypes.data_as(vp), C.c_double(min_frac), C.c_uint32(min_distance), out)
    return out.raw[:n].decode()


def _fuse_records(self, rows, win_begin, pos, cigar, cig_off, seq4, seq_off, min_frac=0.5, min_distance=10):
    """orc_fuse_records: the consensus straight from the records and the by-row matrix (no counters): the independent
    checker of the front end's consensus."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, l = rows.shape
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    cig_off = np.ascontiguousarray(cig_off, dtype=np.uint64)
    seq4 = np.ascontiguousarray(seq4, dtype=np.uint8)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    out = C.create_string_buffer(l * 31 + 1)
    vp = C.c_void_p
    self.lib.orc_fuse_records.restype = C.c_uint32
    k = self.lib.orc_fuse_records(rows.ctypes.data_as(vp), C.c_uint64(n), C.c_uint32(l), C.c_uint32(win_begin), pos.ctypes.data_as(vp),
                                  cigar.ctypes.data_as(vp), cig_off.ctypes.data_as(vp), seq4.ctypes.data_as(vp), seq_off.ctypes.data_as(vp),
                                  C.c_double(min_frac), C.c_uint32(min_distance), out)
    return out.raw[:k].decode()


Oracle.insertions = _insertions
Oracle.fuse = _fuse
Oracle.fuse_records = _fuse_records


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    src = os.path.join(ORACLE_DIR, "juliet_oracle.c")
    if not os.path.exists(LIB) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB)):
        build()
    return Oracle(C.CDLL(LIB))
