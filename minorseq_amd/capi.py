"""ctypes binding of libjuliet_hip.so (include/juliet_hip.h) — the reference-side FFI stub, in Python.

The reference has no API for this path beyond the `juliet` command line (doc/JULIET.md:62-66), so the
class below mirrors the documented stages instead: pileup -> call (-> all-gather) -> phase, with the
documented knobs (`--min-perc`, `--max-perc`, `--drm-only`, phasing on/off) as arguments.
Fails loudly when the library or a gfx950 device is missing: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libjuliet_hip.so")

MAX_HAPLOTYPES = 702
HAP_INSUFFICIENT = 0xFFFE
HAP_DAMAGED = 0xFFFF
VARIANT_CAP = 4096
PACK_PATTERN_BYTES, PACK_HIT_BYTES = 8192, 16384   # jl_internal.h: what the pinned result block holds of hap_pattern / hit

GENE = np.dtype([("begin", "<u4"), ("end", "<u4")])
VARIANT = np.dtype([("gene", "<u4"), ("codon_pos", "<u4"), ("col", "<u4"), ("ref_codon", "u1"), ("codon", "u1"),
                    ("flags", "<u2"), ("count", "<u4"), ("coverage", "<u4"), ("expected", "<u4"), ("pad_", "<u4"),
                    ("p_value", "<f8"), ("log_p", "<f8")])
SUMMARY_FIELDS = ("reported_reads", "insufficient_reads", "damaged_reads", "marginal_gap", "marginal_heteroduplex",
                  "marginal_partial", "n_positions", "n_haplotypes")
SUMMARY = np.dtype([(n, "<u4") for n in SUMMARY_FIELDS])

# every symbol include/juliet_hip.h declares (checked by tests/test_capi_exports.py)
EXPORTS = ("jl_abi_version", "jl_strerror", "jl_device_count", "jl_ctx_create", "jl_ctx_destroy", "jl_last_error",
           "jl_sync", "jl_col_stride", "jl_plane_stride", "jl_msa_upload", "jl_msa_alloc", "jl_msa_adopt", "jl_msa_pack_rows",
           "jl_msa_ingest_records", "jl_records_begin", "jl_records_append", "jl_records_finish", "jl_records_window", "jl_records_window_async", "jl_records_drop", "jl_msa_track_insertions", "jl_insertions_fetch", "jl_msa_download", "jl_synth_fill", "jl_synth_fill_window", "jl_pileup_async", "jl_n_positions", "jl_pileup_fetch",
           "jl_consensus_fetch", "jl_call_async", "jl_call_fetch", "jl_variant_table_device", "jl_phase_async", "jl_phase_fetch",
           "jl_ctx_stream", "jl_run_async", "jl_run_wait", "jl_run_done", "jl_run_view_get", "jl_group_create", "jl_group_destroy",
           "jl_group_last_error", "jl_group_run_async", "jl_group_run_masked_async", "jl_group_views", "jl_group_time_pileup", "jl_fisher_eval", "jl_fisher_eval_tail", "jl_expand_read_hap", "jl_time_run", "jl_time_pileup", "jl_time_pileup_set", "jl_run_pileup_clock", "jl_run_pileup_ms", "jl_pileup_kernel_name", "jl_comm_unique_id", "jl_comm_create", "jl_comm_create_inproc", "jl_comm_destroy", "jl_comm_info",
           "jl_allgather_variants", "jl_allgather_variants_async", "jl_allgather_variants_async_many", "jl_allgather_variants_many", "jl_group_exchange_bind", "jl_group_exchange_collect", "jl_xwin_plan", "jl_xwin_assemble_local",
           "jl_xwin_assemble_rccl", "jl_xwin_assemble_slice_local", "jl_xwin_assemble_slice_rccl", "jl_phase_groups_async",
           "jl_phase_groups_fetch", "jl_phase_regroup", "jl_merge_tables", "jl_merge_groups", "jl_select_haplotypes",
           "jl_xwin_slice_plan", "jl_xwin_create", "jl_xwin_destroy", "jl_xwin_last_error", "jl_xwin_phase_sharded",
           "jl_xwin_read_hap_fetch", "jl_xwin_stage_us", "jl_allgather_groups")


class ErrorModel(C.Structure):
    _fields_ = [("match", C.c_double), ("substitution", C.c_double), ("deletion", C.c_double)]


class Params(C.Structure):
    _fields_ = [("alpha", C.c_double), ("n_tests", C.c_double), ("err", ErrorModel), ("expected_round", C.c_int32),
                ("tail", C.c_int32), ("min_perc", C.c_double), ("max_perc", C.c_double)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("sub_rate", C.c_double), ("del_rate", C.c_double), ("mask_rate", C.c_double),
                ("partial_rate", C.c_double), ("minor_permille", C.c_uint32 * 4), ("reserved", C.c_uint32)]


class PhaseSummary(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in SUMMARY_FIELDS]


class XwinOp(C.Structure):
    """jl_xwin_op: one step of a rank's column-slice exchange (include/juliet_hip.h)."""
    _fields_ = [("op", C.c_int32), ("peer", C.c_int32), ("k_begin", C.c_uint32), ("k_count", C.c_uint32),
                ("read_begin", C.c_uint64), ("n_reads", C.c_uint64), ("dst_stride", C.c_uint64), ("bytes", C.c_uint64),
                ("dst_offset", C.c_uint64)]


XWIN_OP_LOCAL, XWIN_OP_SEND, XWIN_OP_RECV = 0, 1, 2


class XwinResult(C.Structure):
    """jl_xwin_result: what jl_xwin_phase_sharded returns (pointers into the session)."""
    _fields_ = [("n_variants", C.c_uint32), ("n_positions", C.c_uint32), ("n_haplotypes", C.c_uint32), ("n_groups", C.c_uint32),
                ("summary", PhaseSummary), ("merged", C.c_void_p), ("pos_global", C.c_void_p), ("hap_count", C.c_void_p),
                ("hap_pattern", C.c_void_p), ("hit", C.c_void_p), ("cooc", C.c_void_p), ("slice_begin", C.c_uint64),
                ("slice_reads", C.c_uint64), ("read_hap_bits", C.c_uint32), ("reserved", C.c_uint32)]


class RunView(C.Structure):
    """jl_run_view: pointers into the context's pinned result block (include/juliet_hip.h)."""
    _fields_ = [("complete", C.c_uint32), ("n_variants", C.c_uint32), ("phased", C.c_uint32),
                ("n_positions", C.c_uint32), ("n_haplotypes", C.c_uint32), ("n_var_phase", C.c_uint32),
                ("n_reads", C.c_uint64), ("summary", PhaseSummary), ("variants", C.c_void_p),
                ("pos_cols", C.c_void_p), ("hap_count", C.c_void_p), ("hap_pattern", C.c_void_p),
                ("hit", C.c_void_p), ("cooc", C.c_void_p), ("read_hap", C.c_void_p), ("read_hap_packed", C.c_void_p),
                ("read_hap_bits", C.c_uint32), ("reserved", C.c_uint32)]


def _view(addr, dtype, count):
    """numpy array over `count` items of host memory at `addr` (no copy; the owner must outlive the array)."""
    buf = (C.c_char * (int(count) * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=int(count))


ERROR_MODELS = {  # docs/SPEC.md §5 (values UNPINNED)
    "sequel": (0.998826, 5.8e-5, 1.0e-3),
    "permissive": (0.99764, 1.2e-4, 2.0e-3),
}


def default_params(n_tests=0.0, alpha=0.01, chemistry="sequel", min_perc=-1.0, max_perc=-1.0, expected_round=0, tail=0):
    m, s, d = ERROR_MODELS[chemistry]
    return Params(alpha, n_tests, ErrorModel(m, s, d), expected_round, tail, min_perc, max_perc)


def expand_ids(packed, bits, n_reads):
    """Per-read haplotype ids as the kernels store them (4 / 8 / 16 bits per read, include/juliet_hip.h jl_run_view) ->
    uint16 ids (HAP_INSUFFICIENT / HAP_DAMAGED for the two unreported categories)."""
    if bits == 16:
        return packed.view(np.uint16)[:n_reads]
    p = packed.view(np.uint8)
    if bits == 4:
        b = p[: (n_reads + 1) // 2]
        c = np.empty(2 * len(b), dtype=np.uint16)
        c[0::2] = b & 15
        c[1::2] = b >> 4
        c = c[:n_reads]
        out = c.copy()
        out[c == 14] = HAP_INSUFFICIENT
        out[c == 15] = HAP_DAMAGED
        return out
    if bits == 8:
        c = p[:n_reads].astype(np.uint16)
        out = c.copy()
        out[c == 254] = HAP_INSUFFICIENT
        out[c == 255] = HAP_DAMAGED
        return out
    raise ValueError(f"per-read ids of width {bits}")


class JulietError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libjuliet_hip: status {status}: {msg}")
        self.status = status


_lib = None


def load_library(path=LIB_PATH):
    """Load the C-ABI library; raises if it was not built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with minorseq_amd/csrc/Makefile (hipcc, gfx950). "
                          "There is no CPU fallback for the juliet hot path.")
    lib = C.CDLL(path)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    lib.jl_strerror.restype = C.c_char_p
    lib.jl_last_error.restype = C.c_char_p
    lib.jl_last_error.argtypes = [vp]
    lib.jl_pileup_kernel_name.restype = C.c_char_p
    lib.jl_col_stride.restype = u64
    lib.jl_col_stride.argtypes = [u64]
    lib.jl_plane_stride.restype = u64
    lib.jl_plane_stride.argtypes = [u64]
    lib.jl_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.jl_ctx_stream.argtypes = [vp]
    lib.jl_ctx_stream.restype = vp
    lib.jl_ctx_destroy.argtypes = [vp]
    lib.jl_ctx_destroy.restype = None
    lib.jl_sync.argtypes = [vp]
    lib.jl_msa_upload.argtypes = [vp, vp, u64, u32, u64, u32]
    lib.jl_msa_alloc.argtypes = [vp, u64, u32, u32]
    lib.jl_msa_adopt.argtypes = [vp, vp, u64, u32, u64, u32]
    lib.jl_msa_pack_rows.argtypes = [vp, vp, u64, u32, u32]
    lib.jl_msa_ingest_records.argtypes = [vp, u64, u32, u32] + [vp] * 7 + [u32]
    lib.jl_records_begin.argtypes = [vp, u64, u64, u64, u64]
    lib.jl_records_append.argtypes = [vp, u64] + [vp] * 7
    lib.jl_records_finish.argtypes = [vp, u32, u32, u32]
    lib.jl_records_window.argtypes = [vp, vp, u32, u32, u32]
    lib.jl_records_window_async.argtypes = [vp, vp, u32, u32, u32]
    lib.jl_records_drop.argtypes = [vp]
    lib.jl_msa_track_insertions.argtypes = [vp, C.c_int]
    lib.jl_insertions_fetch.argtypes = [vp, vp, vp]
    lib.jl_msa_download.argtypes = [vp, vp, u64]
    lib.jl_synth_fill.argtypes = [vp, C.POINTER(SynthParams), vp]
    lib.jl_synth_fill_window.argtypes = [vp, C.POINTER(SynthParams), vp, u32]
    lib.jl_pileup_async.argtypes = [vp, vp, u32, vp, u32]
    lib.jl_n_positions.argtypes = [vp]
    lib.jl_n_positions.restype = u32
    lib.jl_pileup_fetch.argtypes = [vp] * 7
    lib.jl_consensus_fetch.argtypes = [vp, vp]
    lib.jl_call_async.argtypes = [vp, C.POINTER(Params), vp]
    lib.jl_call_fetch.argtypes = [vp, vp, u32, C.POINTER(u32)]
    lib.jl_variant_table_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u32)]
    lib.jl_phase_async.argtypes = [vp, vp, u32, u32]
    lib.jl_phase_fetch.argtypes = [vp] * 8 + [u32]
    lib.jl_run_async.argtypes = [vp, vp, u32, vp, u32, C.POINTER(Params), vp, C.c_int, u32, C.c_int]
    lib.jl_time_run.argtypes = [vp, vp, u32, vp, u32, C.POINTER(Params), vp, C.c_int, u32, C.c_int, u32, C.POINTER(C.c_float)]
    lib.jl_group_create.argtypes = [vp, u32, C.POINTER(vp)]
    lib.jl_group_destroy.argtypes = [vp]
    lib.jl_group_destroy.restype = None
    lib.jl_group_last_error.argtypes = [vp]
    lib.jl_group_last_error.restype = C.c_char_p
    lib.jl_group_run_async.argtypes = [vp, vp, u32, vp, u32, C.POINTER(Params), C.c_int, u32, C.c_int]
    lib.jl_group_run_masked_async.argtypes = [vp, vp, u32, vp, u32, C.POINTER(Params), vp, C.c_int, u32, C.c_int]
    lib.jl_group_views.argtypes = [vp, C.c_void_p, u32, C.POINTER(u32)]
    lib.jl_group_time_pileup.argtypes = [vp, u32, u32, C.POINTER(C.c_float), C.POINTER(u64)]
    lib.jl_run_wait.argtypes = [vp]
    lib.jl_run_done.argtypes = [vp]
    lib.jl_run_view_get.argtypes = [vp, C.POINTER(RunView)]
    lib.jl_fisher_eval.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    lib.jl_fisher_eval_tail.argtypes = [vp, vp, vp, vp, u32, C.c_int, vp, vp]
    lib.jl_expand_read_hap.argtypes = [vp, u32, u64, vp]
    lib.jl_time_pileup.argtypes = [vp, u32, C.POINTER(C.c_float)]
    lib.jl_time_pileup_set.argtypes = [vp, u32, u32, C.POINTER(C.c_float)]
    lib.jl_run_pileup_clock.argtypes = [vp, C.c_int]
    lib.jl_run_pileup_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(u64)]
    lib.jl_comm_unique_id.argtypes = [vp]
    lib.jl_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.jl_comm_create_inproc.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.jl_comm_destroy.argtypes = [vp]
    lib.jl_comm_destroy.restype = None
    lib.jl_comm_info.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
    lib.jl_allgather_variants.argtypes = [vp, vp, vp, vp, u32]
    lib.jl_allgather_variants_async.argtypes = [vp, vp]
    lib.jl_allgather_variants_async_many.argtypes = [vp, u32, vp]
    lib.jl_allgather_variants_many.argtypes = [vp, u32, vp, vp, vp, u32]
    lib.jl_group_exchange_bind.argtypes = [vp, vp]
    lib.jl_group_exchange_collect.argtypes = [vp, vp, vp, u32]
    lib.jl_xwin_plan.argtypes = [vp, vp, u32, vp, u32, vp, vp, vp, C.POINTER(u32)]
    lib.jl_xwin_assemble_local.argtypes = [vp, vp, u32, vp, u32, vp, vp, C.POINTER(u32)]
    lib.jl_xwin_assemble_rccl.argtypes = [vp, vp, vp, vp, vp, vp, u32, vp, vp, C.POINTER(u32)]
    lib.jl_xwin_assemble_slice_local.argtypes = [vp, vp, u32, vp, u32, u64, u64, vp, vp, C.POINTER(u32)]
    lib.jl_xwin_assemble_slice_rccl.argtypes = [vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, C.POINTER(u32)]
    lib.jl_phase_groups_async.argtypes = [vp, vp, u32]
    lib.jl_phase_groups_fetch.argtypes = [vp, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(u32), vp, u32, vp]
    lib.jl_phase_regroup.argtypes = [vp, vp, u32, u32, vp]
    lib.jl_merge_tables.argtypes = [vp, vp, vp, u32, vp, u32, C.POINTER(u32)]
    lib.jl_merge_groups.argtypes = [vp, vp, vp, vp, u32, u32, vp, vp, u32, C.POINTER(u32), vp]
    lib.jl_select_haplotypes.argtypes = [vp, vp, u32, u32, vp, u32, vp, u32, vp, u32, vp, vp, vp, vp, u32, vp, vp]
    lib.jl_xwin_slice_plan.argtypes = [vp, vp, vp, u32, vp, u32, vp, C.c_int32, C.c_int32, vp, u32, C.POINTER(u32)]
    lib.jl_xwin_create.argtypes = [vp, u32, vp, vp, vp, vp, u32, vp, C.POINTER(vp)]
    lib.jl_xwin_destroy.argtypes = [vp]
    lib.jl_xwin_destroy.restype = None
    lib.jl_xwin_last_error.argtypes = [vp]
    lib.jl_xwin_last_error.restype = C.c_char_p
    lib.jl_xwin_phase_sharded.argtypes = [vp, u32, C.POINTER(XwinResult)]
    lib.jl_xwin_read_hap_fetch.argtypes = [vp, vp]
    lib.jl_xwin_stage_us.argtypes = [vp, vp]
    lib.jl_allgather_groups.argtypes = [vp, vp, u32, u32, vp, vp, vp, vp, C.POINTER(u32)]
    if lib.jl_abi_version() != 5:
        raise ImportError("libjuliet_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _LazyIds:
    """The per-read ids of a run view: packed bytes in pinned memory + their width; expands to uint16 on first use
    (a step loop that only reads the small results never touches them)."""

    def __init__(self, packed, bits, n_reads):
        self.packed, self.bits, self.n_reads = packed, int(bits), int(n_reads)
        self._ids = None

    def ids(self):
        if self._ids is None:
            self._ids = expand_ids(self.packed, self.bits, self.n_reads)
        return self._ids

    def __array__(self, dtype=None, copy=None):
        a = self.ids()
        return a if dtype is None else a.astype(dtype)

    def __eq__(self, other):
        return self.ids() == other

    def __getitem__(self, k):
        return self.ids()[k]

    def __len__(self):
        return self.n_reads

    def copy(self):
        return self.ids().copy()


class Juliet:
    """One context = one GPU = one reference window of aligned CCS reads resident in HBM."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.jl_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise JulietError(rc, self.lib.jl_last_error(None).decode())
        self.h = h
        self.n_reads = 0
        self.n_cols = 0
        self.col_stride = 0
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.jl_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise JulietError(rc, self.lib.jl_last_error(self.h).decode())
        return rc

    # ------------------------------------------------------------------ residency
    def _shape(self, n_reads, n_cols, stride=None):
        """col_stride: the column stride of the INTERCHANGE format (what download_columns returns); the resident planes have
        plane_stride bytes per plane (the library's, or the caller's for an adopted matrix)."""
        self.n_reads, self.n_cols = int(n_reads), int(n_cols)
        self.col_stride = int(self.lib.jl_col_stride(int(n_reads)))
        self.plane_stride = int(stride) if stride is not None else int(self.lib.jl_plane_stride(int(n_reads)))

    def upload_columns(self, packed, n_reads, win_begin=0):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        n_cols, stride = packed.shape
        self._chk(self.lib.jl_msa_upload(self.h, _p(packed), n_reads, n_cols, stride, win_begin))
        self._shape(n_reads, n_cols)

    def upload_rows(self, rows, win_begin=0):
        """By-row uint8 codes; transposed into the resident planes on the device."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n, l = rows.shape
        self._chk(self.lib.jl_msa_pack_rows(self.h, _p(rows), n, l, win_begin))
        self._shape(n, l)

    def ingest_records(self, n_cols, win_begin, pos, cigar, cig_off, seq4, seq_off, qual=None, qual_off=None, min_qv=0):
        """Aligned records (BAM-decoded arrays) -> resident matrix, cigar expansion on the device."""
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        cig_off = np.ascontiguousarray(cig_off, dtype=np.uint64)
        seq4 = np.ascontiguousarray(seq4, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        if qual is not None:
            qual = np.ascontiguousarray(qual, dtype=np.uint8)
            qual_off = np.ascontiguousarray(qual_off, dtype=np.uint64)
        n = len(pos)
        self._chk(self.lib.jl_msa_ingest_records(self.h, n, n_cols, win_begin, _p(pos), _p(cigar), _p(cig_off), _p(seq4),
                                                 _p(seq_off), _p(qual), _p(qual_off), min_qv))
        self._shape(n, n_cols)

    def ingest_records_chunked(self, n_cols, win_begin, pos, cigar, cig_off, seq4, seq_off, qual=None, qual_off=None,
                               min_qv=0, chunk_reads=1000, hints=(0, 0, 0, 0)):
        """The same through jl_records_begin / _append / _finish, `chunk_reads` records per append; every chunk passes
        slices of the caller's arrays with offsets that do not start at 0 (the library rebases them)."""
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        cig_off = np.ascontiguousarray(cig_off, dtype=np.uint64)
        seq4 = np.ascontiguousarray(seq4, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        if qual is not None:
            qual = np.ascontiguousarray(qual, dtype=np.uint8)
            qual_off = np.ascontiguousarray(qual_off, dtype=np.uint64)
        n = len(pos)
        hq = hints[3] if qual is not None else 0
        self._chk(self.lib.jl_records_begin(self.h, hints[0], hints[1], hints[2], hq))
        for a in range(0, n, chunk_reads):
            b = min(n, a + chunk_reads)
            co, so = np.ascontiguousarray(cig_off[a:b + 1]), np.ascontiguousarray(seq_off[a:b + 1])
            qo = np.ascontiguousarray(qual_off[a:b + 1]) if qual is not None else None
            self._chk(self.lib.jl_records_append(self.h, b - a, _p(np.ascontiguousarray(pos[a:b])), _p(cigar), _p(co), _p(seq4),
                                                 _p(so), _p(qual), _p(qo)))
        self._chk(self.lib.jl_records_finish(self.h, n_cols, win_begin, min_qv))
        self._shape(n, n_cols)

    def records_upload(self, pos, cigar, cig_off, seq4, seq_off, qual=None, qual_off=None):
        """jl_records_begin + one jl_records_append: the records stay resident on this context (for records_window /
        records_window_async on other contexts of the device) until records_drop."""
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        cig_off = np.ascontiguousarray(cig_off, dtype=np.uint64)
        seq4 = np.ascontiguousarray(seq4, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        if qual is not None:
            qual = np.ascontiguousarray(qual, dtype=np.uint8)
            qual_off = np.ascontiguousarray(qual_off, dtype=np.uint64)
        n = len(pos)
        self._chk(self.lib.jl_records_begin(self.h, n, len(cigar), len(seq4), len(qual) if qual is not None else 0))
        self._chk(self.lib.jl_records_append(self.h, n, _p(pos), _p(cigar), _p(cig_off), _p(seq4), _p(seq_off), _p(qual), _p(qual_off)))
        self._n_records = n

    def records_window(self, records, n_cols, win_begin=0, min_qv=0, wait=True):
        """This context's resident matrix from the records uploaded to `records` (another context of the device);
        wait=False only enqueues the build on this context's stream (jl_records_window_async)."""
        fn = self.lib.jl_records_window if wait else self.lib.jl_records_window_async
        self._chk(fn(records.h, self.h, n_cols, win_begin, min_qv))
        self._shape(records._n_records, n_cols)

    def records_drop(self):
        self._chk(self.lib.jl_records_drop(self.h))

    def track_insertions(self, on=True):
        self._chk(self.lib.jl_msa_track_insertions(self.h, 1 if on else 0))

    def insertions_fetch(self):
        """(len_hist[n_cols][32], base_counts[n_cols][30][4]) of the last ingest_records with tracking on."""
        lh = np.zeros((self.n_cols, 32), dtype=np.uint32)
        bc = np.zeros((self.n_cols, 30, 4), dtype=np.uint32)
        self._chk(self.lib.jl_insertions_fetch(self.h, _p(lh), _p(bc)))
        return lh, bc

    def alloc(self, n_reads, n_cols, win_begin=0):
        self._chk(self.lib.jl_msa_alloc(self.h, n_reads, n_cols, win_begin))
        self._shape(n_reads, n_cols)

    def adopt(self, device_ptr, n_reads, n_cols, plane_stride, win_begin=0, keep_alive=None):
        """Caller-owned device memory in the resident format: [n_cols][3][plane_stride] bytes (msa.pack_planes)."""
        self._chk(self.lib.jl_msa_adopt(self.h, C.c_void_p(device_ptr), n_reads, n_cols, plane_stride, win_begin))
        self._shape(n_reads, n_cols, plane_stride)
        self._keep = keep_alive

    def synth_fill(self, sp, ref):
        """sp: minorseq_amd.synth.SynthParams; fills the resident matrix on the device."""
        csp = SynthParams(sp.seed, sp.sub_rate, sp.del_rate, sp.mask_rate, sp.partial_rate,
                          (C.c_uint32 * 4)(*sp.minor_permille), 0)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        assert len(ref) == self.n_cols
        self._chk(self.lib.jl_synth_fill(self.h, C.byref(csp), _p(ref)))

    def synth_fill_window(self, sp, ref_full):
        """The window [win_begin, win_begin + n_cols) of the reads synth.rows(sp, len(ref_full), ...) describes."""
        csp = SynthParams(sp.seed, sp.sub_rate, sp.del_rate, sp.mask_rate, sp.partial_rate,
                          (C.c_uint32 * 4)(*sp.minor_permille), 0)
        ref_full = np.ascontiguousarray(ref_full, dtype=np.uint8)
        self._chk(self.lib.jl_synth_fill_window(self.h, C.byref(csp), _p(ref_full), len(ref_full)))

    def download_columns(self):
        out = np.empty((self.n_cols, self.col_stride), dtype=np.uint8)
        self._chk(self.lib.jl_msa_download(self.h, _p(out), out.nbytes))
        return out

    # ------------------------------------------------------------------ stages
    def pileup_async(self, genes, refseq=None):
        genes = np.ascontiguousarray(genes, dtype=GENE)
        if refseq is not None:
            refseq = np.ascontiguousarray(refseq, dtype=np.uint8)
        self._chk(self.lib.jl_pileup_async(self.h, _p(genes), len(genes), _p(refseq),
                                           0 if refseq is None else len(refseq)))

    def pileup_fetch(self):
        p = self.lib.jl_n_positions(self.h)
        col = np.zeros((self.n_cols, 6), dtype=np.uint32)
        pg, pk, pc = (np.zeros(p, dtype=np.uint32) for _ in range(3))
        hist = np.zeros((p, 64), dtype=np.uint32)
        cov = np.zeros(p, dtype=np.uint32)
        self._chk(self.lib.jl_pileup_fetch(self.h, _p(col), _p(pg), _p(pk), _p(pc), _p(hist), _p(cov)))
        return dict(col_counts=col, pos_gene=pg, pos_codon=pk, pos_col=pc, hist=hist, coverage=cov)

    def consensus(self):
        """Majority symbol per column of the last pileup: 0..3 base, 4 majority deletion, 5 uncovered."""
        out = np.zeros(self.n_cols, dtype=np.uint8)
        self._chk(self.lib.jl_consensus_fetch(self.h, _p(out)))
        return out

    def call_async(self, params=None, drm_masks=None):
        prm = params or default_params()
        if drm_masks is not None:
            drm_masks = np.ascontiguousarray(drm_masks, dtype=np.uint64)
            assert len(drm_masks) == self.lib.jl_n_positions(self.h)
        self._chk(self.lib.jl_call_async(self.h, C.byref(prm), _p(drm_masks)))

    def call_fetch(self, cap=VARIANT_CAP):
        out = np.zeros(cap, dtype=VARIANT)
        n = C.c_uint32()
        self._chk(self.lib.jl_call_fetch(self.h, _p(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def variant_table_device(self):
        rows, cnt, cap = C.c_void_p(), C.c_void_p(), C.c_uint32()
        self._chk(self.lib.jl_variant_table_device(self.h, C.byref(rows), C.byref(cnt), C.byref(cap)))
        return rows.value, cnt.value, cap.value

    def phase_async(self, variants=None, min_reads=10):
        if variants is None:
            self._chk(self.lib.jl_phase_async(self.h, None, 0, min_reads))
        else:
            variants = np.ascontiguousarray(variants, dtype=VARIANT)
            # a zero-length table still needs a non-NULL pointer to mean "this table", not "the resident one"
            buf = variants if len(variants) else np.zeros(1, dtype=VARIANT)
            self._chk(self.lib.jl_phase_async(self.h, _p(buf), len(variants), min_reads))

    def phase_fetch(self, want_reads=True, cap_var=VARIANT_CAP):
        summ = np.zeros(1, dtype=SUMMARY)
        pos_cols = np.zeros(cap_var, dtype=np.uint32)
        hap_count = np.zeros(MAX_HAPLOTYPES, dtype=np.uint32)
        hap_pattern = np.zeros((MAX_HAPLOTYPES, cap_var), dtype=np.uint8)
        hit = np.zeros((cap_var, MAX_HAPLOTYPES), dtype=np.uint8)
        read_hap = np.zeros(self.n_reads, dtype=np.uint16) if want_reads else None
        cooc = np.zeros((cap_var, cap_var), dtype=np.uint32) if cap_var <= 1024 else None
        self._chk(self.lib.jl_phase_fetch(self.h, _p(summ), _p(pos_cols), _p(hap_count), _p(hap_pattern), _p(hit),
                                          _p(read_hap), _p(cooc), cap_var))
        s = {k: int(summ[0][k]) for k in SUMMARY_FIELDS}
        h, vp = s["n_haplotypes"], s["n_positions"]
        return dict(summary=s, pos_cols=pos_cols[:vp].copy(), hap_count=hap_count[:h].copy(),
                    hap_pattern=hap_pattern[:h, :vp].copy(), hit=hit, read_hap=read_hap, cooc=cooc)

    # ------------------------------------------------------------------ phasing sharded by reads (SURVEY §8e option A)
    def phase_groups_async(self, variants):
        """Keys + grouping of the resident matrix (a slice of the reads); the groups are exported, not ranked."""
        variants = np.ascontiguousarray(variants, dtype=VARIANT)
        buf = variants if len(variants) else np.zeros(1, dtype=VARIANT)
        self._chk(self.lib.jl_phase_groups_async(self.h, _p(buf), len(variants)))

    def phase_groups_fetch(self, cap_var=64):
        """dict(patterns uint8[G][Vp], counts uint32[G], pos_cols, summary): the groups of the last phase_groups_async."""
        ng, vp = C.c_uint32(), C.c_uint32()
        summ = np.zeros(1, dtype=SUMMARY)
        self._chk(self.lib.jl_phase_groups_fetch(self.h, None, 0, None, 0, C.byref(ng), C.byref(vp), None, 0, _p(summ)))
        g, v = ng.value, vp.value
        patterns = np.zeros((max(g, 1), max(v, 1)), dtype=np.uint8)
        counts = np.zeros(max(g, 1), dtype=np.uint32)
        pos_cols = np.zeros(max(v, 1), dtype=np.uint32)
        self._chk(self.lib.jl_phase_groups_fetch(self.h, _p(patterns), patterns.shape[1], _p(counts), len(counts), C.byref(ng),
                                                 C.byref(vp), _p(pos_cols), len(pos_cols), _p(summ)))
        return dict(patterns=patterns[:g, :v].copy(), counts=counts[:g].copy(), pos_cols=pos_cols[:v].copy(),
                    summary={k: int(summ[0][k]) for k in SUMMARY_FIELDS})

    def phase_regroup(self, hap_of_group, n_haplotypes, want_reads=True):
        """hap_of_group[q] = haplotype of exported group q after the merge (HAP_INSUFFICIENT: not reported) -> per-read ids."""
        hap = np.ascontiguousarray(hap_of_group, dtype=np.uint16)
        read_hap = np.zeros(self.n_reads, dtype=np.uint16) if want_reads else None
        self._chk(self.lib.jl_phase_regroup(self.h, _p(hap if len(hap) else np.zeros(1, dtype=np.uint16)), len(hap), n_haplotypes,
                                            _p(read_hap)))
        return read_hap

    def xwin_assemble_slice_local(self, windows, merged, read_begin, n_slice):
        """This context becomes the compact matrix of reads [read_begin, read_begin + n_slice) of the windows' variant
        columns.  Returns (remapped table, pos_global)."""
        merged = np.ascontiguousarray(merged, dtype=VARIANT)
        arr = (C.c_void_p * len(windows))(*[w.h for w in windows])
        remapped = np.zeros(max(len(merged), 1), dtype=VARIANT)
        pos_global = np.zeros(max(len(merged), 1), dtype=np.uint32)
        vp = C.c_uint32()
        self._chk(self.lib.jl_xwin_assemble_slice_local(self.h, arr, len(windows), _p(merged if len(merged) else remapped), len(merged),
                                                        read_begin, n_slice, _p(remapped), _p(pos_global), C.byref(vp)))
        if vp.value and n_slice:
            self._shape(n_slice, 3 * vp.value)
        return remapped[: len(merged)], pos_global[: vp.value]

    def xwin_assemble_slice_rccl(self, window, comm, win_begins, win_ncols, merged, slice_begin):
        """One window per rank: this context becomes the compact matrix of THIS rank's slice of the reads
        (slice_begin: world + 1 starts, sharding.read_slices).  Returns (remapped table, pos_global, reads of the slice)."""
        merged = np.ascontiguousarray(merged, dtype=VARIANT)
        wb = np.ascontiguousarray(win_begins, dtype=np.uint32)
        wn = np.ascontiguousarray(win_ncols, dtype=np.uint32)
        sb = np.ascontiguousarray(slice_begin, dtype=np.uint64)
        remapped = np.zeros(max(len(merged), 1), dtype=VARIANT)
        pos_global = np.zeros(max(len(merged), 1), dtype=np.uint32)
        vp = C.c_uint32()
        self._chk(self.lib.jl_xwin_assemble_slice_rccl(self.h, window.h, comm, _p(wb), _p(wn), _p(merged if len(merged) else remapped),
                                                       len(merged), _p(sb), _p(remapped), _p(pos_global), C.byref(vp)))
        return remapped[: len(merged)], pos_global[: vp.value], vp.value

    def fisher_eval(self, a, c, cov, tail=0):
        a, c, cov = (np.ascontiguousarray(x, dtype=np.uint32) for x in (a, c, cov))
        p = np.zeros(len(a), dtype=np.float64)
        lp = np.zeros(len(a), dtype=np.float64)
        self._chk(self.lib.jl_fisher_eval_tail(self.h, _p(a), _p(c), _p(cov), len(a), tail, _p(p), _p(lp)))
        return p, lp

    def sync(self):
        self._chk(self.lib.jl_sync(self.h))

    def run_pileup_clock(self, on=True):
        """Clock nodes around the pileup launch of this context's runs (jl_run_pileup_clock)."""
        self._chk(self.lib.jl_run_pileup_clock(self.h, 1 if on else 0))

    def run_pileup_ms(self):
        """ms the pileup of the last run took where it ran (jl_run_pileup_ms; waits for the run)."""
        ms = C.c_float()
        self._chk(self.lib.jl_run_pileup_ms(self.h, C.byref(ms), None))
        return float(ms.value)

    def run_pileup_interval(self):
        """(begin, end) of the last run's pileup in ms of the device's 100 MHz clock (one clock for all contexts of a device)."""
        ms, t0 = C.c_float(), C.c_uint64()
        self._chk(self.lib.jl_run_pileup_ms(self.h, C.byref(ms), C.byref(t0)))
        return 1e-5 * t0.value, 1e-5 * t0.value + float(ms.value)

    def time_run(self, genes, refseq=None, params=None, drm_masks=None, phasing=True, min_reads=10, want_read_hap=True, reps=20):
        """jl_time_run: average ms per run of `reps` runs one after the other, launched and waited for inside the library
        (the latency of the path at the C ABI, without this interpreter between the runs)."""
        g = np.ascontiguousarray(genes, dtype=GENE)
        r = None if refseq is None else np.ascontiguousarray(refseq, dtype=np.uint8)
        d = None if drm_masks is None else np.ascontiguousarray(drm_masks, dtype=np.uint64)
        prm = params or default_params()
        ms = C.c_float()
        self._chk(self.lib.jl_time_run(self.h, _p(g), len(g), _p(r), 0 if r is None else len(r), C.byref(prm), _p(d),
                                       1 if phasing else 0, min_reads, 1 if want_read_hap else 0, reps, C.byref(ms)))
        return float(ms.value)

    def time_pileup(self, reps=20):
        ms = C.c_float()
        self._chk(self.lib.jl_time_pileup(self.h, reps, C.byref(ms)))
        return ms.value

    # ------------------------------------------------------------------ the whole path in one enqueue
    def run_async(self, genes, refseq=None, params=None, drm_masks=None, phasing=True, min_reads=10,
                  want_read_hap=True):
        """pileup -> call (-> phase) as one captured HIP graph + one pinned result copy (jl_run_async)."""
        key = (id(genes), id(refseq), id(params), id(drm_masks))
        c = getattr(self, "_run_args", None)
        if c is None or c[0] != key:   # argument marshalling is cached: repeated steps pass the same objects
            g = np.ascontiguousarray(genes, dtype=GENE)
            r = None if refseq is None else np.ascontiguousarray(refseq, dtype=np.uint8)
            d = None if drm_masks is None else np.ascontiguousarray(drm_masks, dtype=np.uint64)
            prm = params or default_params()
            c = (key, (genes, refseq, params, drm_masks), (g, r, d, prm),
                 (_p(g), len(g), _p(r), 0 if r is None else len(r), C.byref(prm), _p(d)))
            self._run_args = c
        a = c[3]
        rc = self.lib.jl_run_async(self.h, a[0], a[1], a[2], a[3], a[4], a[5], 1 if phasing else 0, min_reads,
                                   1 if want_read_hap else 0)
        if rc:
            self._chk(rc)

    def _bufs(self, cap_var):
        b = getattr(self, "_fetch_bufs", None)
        if b is None or b["cap_var"] != cap_var or len(b["read_hap"]) != self.n_reads:
            b = dict(cap_var=cap_var, variants=np.zeros(VARIANT_CAP, dtype=VARIANT), n=C.c_uint32(),
                     summ=np.zeros(1, dtype=SUMMARY), pos_cols=np.zeros(cap_var, dtype=np.uint32),
                     hap_count=np.zeros(MAX_HAPLOTYPES, dtype=np.uint32),
                     hap_pattern=np.zeros((MAX_HAPLOTYPES, cap_var), dtype=np.uint8),
                     hit=np.zeros((cap_var, MAX_HAPLOTYPES), dtype=np.uint8),
                     read_hap=np.zeros(self.n_reads, dtype=np.uint16),
                     cooc=np.zeros((cap_var, cap_var), dtype=np.uint32))
            b["ptr"] = {k: _p(v) for k, v in b.items() if isinstance(v, np.ndarray)}
            b["n_ref"] = C.byref(b["n"])
            self._fetch_bufs = b
        return b

    def run_fetch(self, phasing=True, want_read_hap=True, cap_var=64):
        """Results of the last run_async.  Returned arrays are views of buffers reused by the next call."""
        b = self._bufs(cap_var)
        q = b["ptr"]
        rc = self.lib.jl_call_fetch(self.h, q["variants"], VARIANT_CAP, b["n_ref"])
        if rc:
            self._chk(rc)
        nv = b["n"].value
        out = dict(variants=b["variants"][:nv])
        if phasing:
            rc = self.lib.jl_phase_fetch(self.h, q["summ"], q["pos_cols"], q["hap_count"], q["hap_pattern"], q["hit"],
                                         q["read_hap"] if want_read_hap else None, q["cooc"], cap_var)
            if rc:
                self._chk(rc)
            s = dict(zip(SUMMARY_FIELDS, b["summ"][0].tolist()))
            h, vp = s["n_haplotypes"], s["n_positions"]
            out["phase"] = dict(summary=s, pos_cols=b["pos_cols"][:vp], hap_count=b["hap_count"][:h],
                                hap_pattern=b["hap_pattern"][:h, :vp], hit=b["hit"][:nv, :h],
                                read_hap=b["read_hap"] if want_read_hap else None, cooc=b["cooc"][:nv, :nv])
        return out

    def run_wait(self):
        """Block until the last run_async's results are on the host (pinned sequence word, no HIP sync)."""
        rc = self.lib.jl_run_wait(self.h)
        if rc:
            self._chk(rc)

    def run_done(self):
        return bool(self.lib.jl_run_done(self.h))

    def run_view_raw(self):
        """The jl_run_view struct itself (counts, read categories, pointers into the pinned result block): waits for the
        run like run_view, without building numpy views.  `complete` = 0: use run_fetch."""
        v = getattr(self, "_rv", None)
        if v is None:
            v = self._rv = RunView()
            self._rv_ref = C.byref(v)
            self._rv_cache = {}
        rc = self.lib.jl_run_view_get(self.h, self._rv_ref)
        if rc:
            self._chk(rc)
        return v

    def run_view(self):
        """Zero-copy results of the last run_async: numpy views of the pinned block the kernels stored into
        (valid until the next run on this context).  Returns None when the results do not fit that block
        (more than 128 variants / positions / haplotypes) — use run_fetch then."""
        v = getattr(self, "_rv", None)
        if v is None:
            v = self._rv = RunView()
            self._rv_ref = C.byref(v)
            self._rv_cache = {}
        rc = self.lib.jl_run_view_get(self.h, self._rv_ref)
        if rc:
            self._chk(rc)
        if not v.complete:
            return None
        key = (v.variants, v.read_hap_packed, v.n_reads, v.phased)
        c = self._rv_cache.get(key)
        if c is None:   # base arrays over the whole block, built once per allocation
            c = dict(variants=_view(v.variants, VARIANT, 128))
            if v.phased:
                c.update(pos_cols=_view(v.pos_cols, np.uint32, 128), hap_count=_view(v.hap_count, np.uint32, 128),
                         hap_pattern=_view(v.hap_pattern, np.uint8, PACK_PATTERN_BYTES), hit=_view(v.hit, np.uint8, PACK_HIT_BYTES),
                         cooc=_view(v.cooc, np.uint32, 4096) if v.cooc else None,
                         ids=_view(v.read_hap_packed, np.uint8, 2 * v.n_reads) if v.read_hap_packed else None)
            self._rv_cache = {key: c}
        nv = v.n_variants
        if not v.phased:
            return dict(variants=c["variants"][:nv])
        # the sliced and reshaped views are kept while the results keep their shape (a step loop's runs do): they look at
        # the same pinned block, whose contents the run has replaced; the summary and the ids' holder are made per run
        h, vp, nvp = v.n_haplotypes, v.n_positions, v.n_var_phase
        shape = (nv, h, vp, nvp, bool(v.cooc))
        sh = c.get("shaped")
        if sh is None or sh[0] != shape:
            sh = c["shaped"] = (shape, c["variants"][:nv],
                                dict(pos_cols=c["pos_cols"][:vp], hap_count=c["hap_count"][:h],
                                     hap_pattern=c["hap_pattern"][:h * vp].reshape(h, vp), hit=c["hit"][:nvp * h].reshape(nvp, h),
                                     cooc=None if c["cooc"] is None or not v.cooc else c["cooc"][:nvp * nvp].reshape(nvp, nvp)))
        ph = dict(sh[2])
        ph["summary"] = {n: getattr(v.summary, n) for n in SUMMARY_FIELDS}
        ph["read_hap"] = _LazyIds(c["ids"], v.read_hap_bits, v.n_reads) if c["ids"] is not None else None
        return dict(variants=sh[1], phase=ph)

    def run(self, genes, refseq=None, params=None, drm_masks=None, phasing=True, min_reads=10):
        """The whole hot path, blocking; arrays are copies."""
        self.run_async(genes, refseq, params, drm_masks, phasing, min_reads, want_read_hap=True)
        cap = 64
        while True:
            try:
                out = self.run_fetch(phasing, True, cap_var=cap)
                break
            except JulietError as e:
                if e.status != -5 or cap >= VARIANT_CAP:
                    raise
                cap = min(VARIANT_CAP, cap * 8)
        res = dict(variants=out["variants"].copy())
        if phasing:
            res["phase"] = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in out["phase"].items()}
        return res


def xwin_plan(win_begins, win_ncols, merged):
    """Host-only plan of the cross-window column exchange (jl_xwin_plan): (remapped table, global positions, owner window
    of each position).  Needs no GPU."""
    lib = load_library()
    merged = np.ascontiguousarray(merged, dtype=VARIANT)
    wb = np.ascontiguousarray(win_begins, dtype=np.uint32)
    wn = np.ascontiguousarray(win_ncols, dtype=np.uint32)
    remapped = np.zeros(max(1, len(merged)), dtype=VARIANT)
    pos = np.zeros(max(1, len(merged)), dtype=np.uint32)
    owner = np.zeros(max(1, len(merged)), dtype=np.int32)
    vp = C.c_uint32()
    rc = lib.jl_xwin_plan(_p(wb), _p(wn), len(wb), _p(merged), len(merged), _p(remapped), _p(pos), _p(owner), C.byref(vp))
    if rc:
        raise JulietError(rc, "jl_xwin_plan")
    return remapped[: len(merged)].copy(), pos[: vp.value].copy(), owner[: vp.value].copy()


def merge_tables(tables, win_begins):
    """jl_merge_tables: per-window tables (window-relative `col`) -> one table, global columns, (gene, codon_pos, codon) order."""
    lib = load_library()
    tabs = [np.ascontiguousarray(t, dtype=VARIANT) for t in tables]
    n = len(tabs)
    ptrs = (C.c_void_p * max(1, n))(*[t.ctypes.data if len(t) else None for t in tabs])
    counts = np.array([len(t) for t in tabs], dtype=np.uint32)
    wb = np.ascontiguousarray(win_begins, dtype=np.uint32)
    total = int(counts.sum())
    out = np.zeros(max(1, total), dtype=VARIANT)
    got = C.c_uint32()
    rc = lib.jl_merge_tables(ptrs, _p(counts), _p(wb), n, _p(out), len(out), C.byref(got))
    if rc:
        raise JulietError(rc, "jl_merge_tables")
    return out[: got.value].copy()


def merge_groups(tables):
    """jl_merge_groups: per-slice group tables dict(patterns uint8[G][Vp], counts uint32[G]) -> (distinct patterns ascending,
    summed counts int64, per table the merged row of each of its groups)."""
    lib = load_library()
    vp = max((np.asarray(t["patterns"]).reshape(len(t["counts"]), -1).shape[1] for t in tables if len(t["counts"])), default=0)
    pats = [np.ascontiguousarray(np.asarray(t["patterns"], dtype=np.uint8).reshape(len(t["counts"]), -1)[:, :vp]) if len(t["counts"])
            else np.zeros((0, vp), dtype=np.uint8) for t in tables]
    cnts = [np.ascontiguousarray(t["counts"], dtype=np.uint32) for t in tables]
    n = len(tables)
    total = sum(len(c) for c in cnts)
    index = [np.zeros(max(1, len(c)), dtype=np.uint32) for c in cnts]
    pp = (C.c_void_p * max(1, n))(*[p.ctypes.data if p.size else None for p in pats])
    cp = (C.c_void_p * max(1, n))(*[c.ctypes.data if len(c) else None for c in cnts])
    ip = (C.c_void_p * max(1, n))(*[i.ctypes.data for i in index])
    strides = np.full(max(1, n), max(vp, 1), dtype=np.uint32)
    strides[:n] = [p.shape[1] if p.size else max(vp, 1) for p in pats]
    ng = np.array([len(c) for c in cnts] or [0], dtype=np.uint32)
    mp = np.zeros((max(1, total), max(1, vp)), dtype=np.uint8)
    mc = np.zeros(max(1, total), dtype=np.uint64)
    got = C.c_uint32()
    # merged_patterns is [cap][vp]: contiguous rows of exactly vp bytes
    flat = np.zeros(max(1, total) * max(1, vp), dtype=np.uint8)
    rc = lib.jl_merge_groups(pp, _p(strides), cp, _p(ng), n, vp, _p(flat), _p(mc), max(1, total), C.byref(got), ip)
    if rc:
        raise JulietError(rc, "jl_merge_groups")
    m = got.value
    mp = flat[: m * vp].reshape(m, vp).copy() if vp else np.zeros((m, 0), dtype=np.uint8)
    return mp, mc[:m].astype(np.int64), [i[: len(c)].astype(np.int64) for i, c in zip(index, cnts)]


def select_haplotypes(patterns, counts, variants, pos_cols, min_reads=10, partials=()):
    """jl_select_haplotypes (docs/SPEC.md §8 on merged groups) -> dict like Juliet.phase_fetch (no read_hap) +
    hap_of_merged int64[M] (HAP_INSUFFICIENT where not reported)."""
    lib = load_library()
    patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    m = len(counts)
    vp = len(pos_cols)
    patterns = patterns.reshape(m, vp) if m else np.zeros((0, vp), dtype=np.uint8)
    variants = np.ascontiguousarray(variants, dtype=VARIANT)
    nv = len(variants)
    pc = np.ascontiguousarray(pos_cols, dtype=np.uint32)
    parts = np.zeros(max(1, len(partials)), dtype=SUMMARY)
    for k, p in enumerate(partials):
        for f in ("damaged_reads", "marginal_gap", "marginal_heteroduplex", "marginal_partial"):
            parts[k][f] = int(p[f])
    summ = np.zeros(1, dtype=SUMMARY)
    hap_count = np.zeros(MAX_HAPLOTYPES, dtype=np.uint32)
    hap_pattern = np.zeros((MAX_HAPLOTYPES, max(1, vp)), dtype=np.uint8)
    hcap = max(1, min(MAX_HAPLOTYPES, m))
    hit = np.zeros((max(1, nv), hcap), dtype=np.uint8)
    cooc = np.zeros((max(1, nv), max(1, nv)), dtype=np.uint32)
    hom = np.zeros(max(1, m), dtype=np.uint16)
    flatp = np.ascontiguousarray(patterns).reshape(-1) if patterns.size else np.zeros(1, dtype=np.uint8)
    hp_flat = np.zeros(MAX_HAPLOTYPES * max(1, vp), dtype=np.uint8)
    rc = lib.jl_select_haplotypes(_p(flatp), _p(counts if m else np.zeros(1, dtype=np.uint64)), m, vp,
                                  _p(variants if nv else np.zeros(1, dtype=VARIANT)), nv, _p(pc if vp else np.zeros(1, dtype=np.uint32)),
                                  min_reads, _p(parts), len(partials), _p(summ), _p(hap_count), _p(hp_flat), _p(hit), hcap, _p(cooc), _p(hom))
    if rc:
        raise JulietError(rc, "jl_select_haplotypes")
    s = {k: int(summ[0][k]) for k in SUMMARY_FIELDS}
    h = s["n_haplotypes"]
    hap_of_merged = hom[:m].astype(np.int64)
    return dict(summary=s, hap_count=hap_count[:h].copy(), hap_pattern=hp_flat[: h * vp].reshape(h, vp).copy() if vp else np.zeros((h, 0), dtype=np.uint8),
                hit=hit[:nv, :h].copy(), cooc=cooc[:nv, :nv].copy(), hap_of_merged=hap_of_merged)


def xwin_slice_plan(win_begins, win_ncols, win_rank, merged, slice_begin, world, rank):
    """jl_xwin_slice_plan: the ops of `rank` in issue order, as dicts (op, peer, k_begin, k_count, read_begin, n_reads,
    dst_stride, bytes, dst_offset).  Needs no GPU."""
    lib = load_library()
    merged = np.ascontiguousarray(merged, dtype=VARIANT)
    wb = np.ascontiguousarray(win_begins, dtype=np.uint32)
    wn = np.ascontiguousarray(win_ncols, dtype=np.uint32)
    wr = np.ascontiguousarray(win_rank, dtype=np.int32)
    sb = np.ascontiguousarray(slice_begin, dtype=np.uint64)
    cap = 2 * world + 1
    ops = (XwinOp * cap)()
    n = C.c_uint32()
    rc = lib.jl_xwin_slice_plan(_p(wb), _p(wn), _p(wr), len(wb), _p(merged if len(merged) else np.zeros(1, dtype=VARIANT)), len(merged),
                                _p(sb), world, rank, ops, cap, C.byref(n))
    if rc:
        raise JulietError(rc, "jl_xwin_slice_plan")
    return [{f: getattr(ops[k], f) for f, _ in XwinOp._fields_} for k in range(n.value)]


class Xwin:
    """A cross-window phasing session (jl_xwin_*): this rank's windows of ONE reference whose reads span every window.
    `windows`: this rank's Juliet contexts in ascending column order; `comm`: the RCCL communicator handle (None when every
    window is on this device); `win_begins/win_ncols/win_rank`: the layout of ALL windows; `slice_begin`: world + 1 starts."""

    def __init__(self, windows, win_begins, win_ncols, win_rank, slice_begin, comm=None):
        self.lib = load_library()
        self.windows = list(windows)
        arr = (C.c_void_p * len(self.windows))(*[w.h for w in self.windows])
        wb = np.ascontiguousarray(win_begins, dtype=np.uint32)
        wn = np.ascontiguousarray(win_ncols, dtype=np.uint32)
        wr = np.ascontiguousarray(win_rank, dtype=np.int32)
        sb = np.ascontiguousarray(slice_begin, dtype=np.uint64)
        h = C.c_void_p()
        rc = self.lib.jl_xwin_create(arr, len(self.windows), comm, _p(wb), _p(wn), _p(wr), len(wb), _p(sb), C.byref(h))
        if rc:
            raise JulietError(rc, "jl_xwin_create: bad layout (windows / ranks / slices)")
        self.h = h
        self._res = XwinResult()
        self._res_ref = C.byref(self._res)

    def close(self):
        if getattr(self, "h", None):
            self.lib.jl_xwin_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def phase_raw(self, min_reads=10):
        """One jl_xwin_phase_sharded call; returns the result struct (pointers into the session, no copies)."""
        rc = self.lib.jl_xwin_phase_sharded(self.h, min_reads, self._res_ref)
        if rc:
            raise JulietError(rc, self.lib.jl_xwin_last_error(self.h).decode())
        return self._res

    def phase(self, min_reads=10, want_reads=True):
        """dict like phase_across_windows' (copies): merged table, summary, haplotypes, hit, cooc, pos_cols (global
        columns), read_hap of THIS rank's slice."""
        r = self.phase_raw(min_reads)
        nv, vp, h = r.n_variants, r.n_positions, r.n_haplotypes
        merged = _view(r.merged, VARIANT, nv).copy() if nv else np.zeros(0, dtype=VARIANT)
        out = dict(merged=merged, pos_cols=_view(r.pos_global, np.uint32, vp).copy() if vp else np.zeros(0, dtype=np.uint32),
                   summary={n: getattr(r.summary, n) for n in SUMMARY_FIELDS}, n_groups=r.n_groups,
                   slice=(int(r.slice_begin), int(r.slice_reads)))
        if vp:
            out.update(hap_count=_view(r.hap_count, np.uint32, h).copy() if h else np.zeros(0, dtype=np.uint32),
                       hap_pattern=_view(r.hap_pattern, np.uint8, h * vp).reshape(h, vp).copy() if h else np.zeros((0, vp), dtype=np.uint8),
                       hit=_view(r.hit, np.uint8, nv * h).reshape(nv, h).copy() if nv * h else np.zeros((nv, h), dtype=np.uint8),
                       cooc=_view(r.cooc, np.uint32, nv * nv).reshape(nv, nv).copy() if nv else np.zeros((0, 0), dtype=np.uint32))
        if want_reads:
            out["read_hap"] = self.read_hap()
        return out

    STAGES = ("tables on the host", "gather + merge of the tables", "plan", "pack + exchange + grouping enqueued", "wait for the groups",
              "merge + selection", "per-read ids")

    def stage_us(self):
        """Host microseconds of the last phase call by stage (jl_xwin_stage_us)."""
        out = np.zeros(8, dtype=np.float32)
        self.lib.jl_xwin_stage_us(self.h, _p(out))
        return dict(zip(self.STAGES, out.tolist()))

    def read_hap(self):
        n = int(self._res.slice_reads)
        ids = np.zeros(max(1, n), dtype=np.uint16)
        rc = self.lib.jl_xwin_read_hap_fetch(self.h, _p(ids))
        if rc:
            raise JulietError(rc, self.lib.jl_xwin_last_error(self.h).decode())
        return ids[:n]


def phase_across_windows(windows, merged, min_reads=10, comm=None, win_begins=None, win_ncols=None, device=0):
    """Cross-window phasing: `windows` = contexts holding the SAME reads over different column
    windows (all on this device), or a single context plus an RCCL `comm` when every rank owns one window.
    `merged`: all windows' variant rows with GLOBAL columns.  Returns (phase result, global columns of the positions)."""
    merged = np.ascontiguousarray(merged, dtype=VARIANT)
    pc = Juliet(device)
    remapped = np.zeros(max(1, len(merged)), dtype=VARIANT)
    pos_global = np.zeros(max(1, len(merged)), dtype=np.uint32)
    vp = C.c_uint32()
    if comm is None:
        arr = (C.c_void_p * len(windows))(*[w.h for w in windows])
        pc._chk(pc.lib.jl_xwin_assemble_local(pc.h, arr, len(windows), _p(merged), len(merged), _p(remapped), _p(pos_global),
                                              C.byref(vp)))
    else:
        wb = np.ascontiguousarray(win_begins, dtype=np.uint32)
        wn = np.ascontiguousarray(win_ncols, dtype=np.uint32)
        pc._chk(pc.lib.jl_xwin_assemble_rccl(pc.h, windows[0].h, comm, _p(wb), _p(wn), _p(merged), len(merged), _p(remapped),
                                             _p(pos_global), C.byref(vp)))
    if vp.value == 0:
        pc.close()
        return None, pos_global[:0]
    pc._shape(windows[0].n_reads, 3 * vp.value, windows[0].plane_stride)
    pc.phase_async(remapped[: len(merged)], min_reads)
    ph = pc.phase_fetch(want_reads=True, cap_var=max(1, len(merged)))
    ph["hit"] = ph["hit"][: len(merged), : ph["summary"]["n_haplotypes"]].copy()
    if ph["cooc"] is not None:
        ph["cooc"] = ph["cooc"][: len(merged), : len(merged)].copy()
    pc.close()
    return ph, pos_global[: vp.value].copy()


def phase_sharded_by_reads(windows, merged, n_shards, min_reads=10, device=0, want_reads=True):
    """Cross-window phasing with the reads sharded (SURVEY §8e option A), every shard on THIS device — what the ranks of a
    multi-GPU run do, one after the other: slice assembly, grouping + export per shard, merge and selection on the host
    (sharding.merge_groups / select_haplotypes), the ids of every shard from jl_phase_regroup.
    Returns (phase result as phase_across_windows gives it, global columns of the positions)."""
    from . import sharding

    merged = np.ascontiguousarray(merged, dtype=VARIANT)
    n = windows[0].n_reads
    bounds = sharding.read_slices(n, n_shards)
    shards, tables, remapped, pos_global = [], [], None, None
    for s in range(n_shards):
        pc = Juliet(device)
        remapped, pos_global = pc.xwin_assemble_slice_local(windows, merged, bounds[s], bounds[s + 1] - bounds[s])
        if len(pos_global) == 0:
            pc.close()
            for q in shards:
                q.close()
            return None, pos_global
        if bounds[s + 1] > bounds[s]:
            pc.phase_groups_async(remapped)
            tables.append(pc.phase_groups_fetch(cap_var=max(1, len(pos_global))))
        else:   # a rank without reads contributes nothing
            tables.append(dict(patterns=np.zeros((0, len(pos_global)), dtype=np.uint8), counts=np.zeros(0, dtype=np.uint32),
                               pos_cols=3 * np.arange(len(pos_global), dtype=np.uint32),
                               summary={k: 0 for k in SUMMARY_FIELDS}))
        shards.append(pc)
    patterns, counts, index = merge_groups(tables)
    pos_cols = next(t["pos_cols"] for t in tables if len(t["pos_cols"]))
    ph = select_haplotypes(patterns, counts, remapped, pos_cols, min_reads, [t["summary"] for t in tables])
    ids = []
    for s, pc in enumerate(shards):
        if bounds[s + 1] > bounds[s]:
            hap = ph["hap_of_merged"][index[s]].astype(np.uint16)
            r = pc.phase_regroup(hap, ph["summary"]["n_haplotypes"], want_reads)
            if want_reads:
                ids.append(r)
        pc.close()
    ph["read_hap"] = np.concatenate(ids) if want_reads and ids else None
    ph["pos_cols"] = pos_global.copy()
    return ph, pos_global.copy()


class Group:
    """Several resident windows (Juliet contexts of one device) run through the path in three launches
    (jl_group_run_async); results are read from each context as after run_async (run_view / run_fetch)."""

    def __init__(self, ctxs):
        self.lib = load_library()
        self.ctxs = list(ctxs)
        arr = (C.c_void_p * len(self.ctxs))(*[c.h for c in self.ctxs])
        h = C.c_void_p()
        rc = self.lib.jl_group_create(arr, len(self.ctxs), C.byref(h))
        if rc != 0:
            raise JulietError(rc, "jl_group_create failed")
        self.h = h
        self._args = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.jl_group_destroy(self.h)
            self.h = None

    def bind_exchange(self, comm):
        """jl_group_exchange_bind: every run of the group from now on carries the all-gather of its windows' table heads
        (comm: the jl_comm handle, None unbinds).  A collective the first time a communicator is bound."""
        rc = self.lib.jl_group_exchange_bind(self.h, comm)
        if rc:
            raise JulietError(rc, self.lib.jl_group_last_error(self.h).decode())

    def exchange_collect(self, world, cap_rows=128, rows=None, counts=None):
        """jl_group_exchange_collect: the group's oldest pending exchange -> (rows [windows][world][cap_rows] of VARIANT,
        counts [windows][world])."""
        n = len(self.ctxs)
        if rows is None:
            rows = np.zeros(n * world * cap_rows, dtype=VARIANT)
            counts = np.zeros(n * world, dtype=np.uint32)
        rc = self.lib.jl_group_exchange_collect(self.h, rows.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), cap_rows)
        if rc:
            raise JulietError(rc, self.lib.jl_group_last_error(self.h).decode())
        return rows.reshape(n, world, cap_rows), counts.reshape(n, world)

    def views(self):
        """jl_group_views: the jl_run_view of every window after a group run, ONE call (it waits for each window in turn).
        Returns a numpy record array over the structs (fields complete, n_variants, n_haplotypes, ...), reused between
        calls."""
        if getattr(self, "_views", None) is None:
            self._views = (RunView * len(self.ctxs))()
            self._views_n = C.c_uint32()
            dt = np.dtype(dict(names=["complete", "n_variants", "phased", "n_positions", "n_haplotypes", "n_var_phase"],
                               formats=[np.uint32] * 6, offsets=[0, 4, 8, 12, 16, 20], itemsize=C.sizeof(RunView)))
            self._views_np = np.frombuffer(self._views, dtype=dt)
        rc = self.lib.jl_group_views(self.h, self._views, len(self.ctxs), C.byref(self._views_n))
        if rc:
            raise JulietError(rc, self.lib.jl_group_last_error(self.h).decode())
        return self._views_np

    def run_masked_async(self, genes, refseq, params, drm_masks, phasing=True, min_reads=10, want_read_hap=True):
        """drm_masks: one uint64[P] array (or None) per window (--drm-only, doc/JULIET.md:370)."""
        g = np.ascontiguousarray(genes, dtype=GENE)
        r = None if refseq is None else np.ascontiguousarray(refseq, dtype=np.uint8)
        prm = params or default_params()
        keep = [None if m is None else np.ascontiguousarray(m, dtype=np.uint64) for m in drm_masks]
        arr = (C.c_void_p * len(keep))(*[None if m is None else m.ctypes.data for m in keep])
        rc = self.lib.jl_group_run_masked_async(self.h, _p(g), len(g), _p(r), 0 if r is None else len(r), C.byref(prm), arr,
                                                1 if phasing else 0, min_reads, 1 if want_read_hap else 0)
        if rc:
            raise JulietError(rc, self.lib.jl_group_last_error(self.h).decode())

    def run_async(self, genes, refseq=None, params=None, phasing=True, min_reads=10, want_read_hap=True):
        key = (id(genes), id(refseq), id(params))
        c = self._args
        if c is None or c[0] != key:   # marshalling cached: repeated steps pass the same objects
            g = np.ascontiguousarray(genes, dtype=GENE)
            r = None if refseq is None else np.ascontiguousarray(refseq, dtype=np.uint8)
            prm = params or default_params()
            c = self._args = (key, (genes, refseq, params), (g, r, prm),
                              (_p(g), len(g), _p(r), 0 if r is None else len(r), C.byref(prm)))
        a = c[3]
        rc = self.lib.jl_group_run_async(self.h, a[0], a[1], a[2], a[3], a[4], 1 if phasing else 0, min_reads,
                                         1 if want_read_hap else 0)
        if rc:
            raise JulietError(rc, self.lib.jl_group_last_error(self.h).decode())


def time_pileup_groups(groups, reps=20):
    """(average ms per grouped pileup launch, algorithmic bytes per launch), rotating over the groups
    (jl_group_time_pileup)."""
    arr = (C.c_void_p * len(groups))(*[g.h for g in groups])
    ms, nbytes = C.c_float(), C.c_uint64()
    rc = groups[0].lib.jl_group_time_pileup(arr, len(groups), reps, C.byref(ms), C.byref(nbytes))
    if rc:
        raise JulietError(rc, groups[0].lib.jl_group_last_error(groups[0].h).decode())
    return float(ms.value), int(nbytes.value)


def time_pileup_set(ctxs, reps=20):
    """Average ms per pileup launch over the contexts' windows in rotation (jl_time_pileup_set)."""
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    ms = C.c_float()
    ctxs[0]._chk(ctxs[0].lib.jl_time_pileup_set(arr, len(ctxs), reps, C.byref(ms)))
    return float(ms.value)


def haplotype_name(h: int) -> str:
    """Haplotype ids `[A-Z]{1}[a-z]?` (doc/JULIET.md:198): A..Z, then Aa..Az, Ba.. (docs/SPEC.md §8)."""
    if h < 26:
        return chr(65 + h)
    h -= 26
    return chr(65 + h // 26) + chr(97 + h % 26)
