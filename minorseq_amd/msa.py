"""MSA layouts (docs/SPEC.md §1): by-row uint8 codes <-> the resident bit planes (the device layout: three planes per
column, plane k = bit k of every read's code) <-> column-packed nibbles (the host interchange format of jl_msa_upload /
jl_msa_download).

Symbol codes: A C G T = 0..3, '-' = 4 (deletion), 'N' = 5 (QV-filtered, doc/JULIET.md:256-259),
' ' = 6 (read does not cover the column).
"""
import numpy as np

SYM_A, SYM_C, SYM_G, SYM_T, SYM_GAP, SYM_MASK, SYM_NONE = range(7)
ALPHABET = "ACGT-N "


def col_stride(n_reads: int) -> int:
    """Bytes per column: ceil(n/2) rounded up to 128 (include/juliet_hip.h jl_col_stride)."""
    return ((n_reads + 1) // 2 + 127) // 128 * 128


def pack_columns(rows: np.ndarray) -> np.ndarray:
    """uint8[N][L] codes -> uint8[L][col_stride] nibbles; read i in byte i//2, low nibble for even i; pad = 6."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    if rows.ndim != 2:
        raise ValueError("rows must be [n_reads][n_cols]")
    if rows.size and rows.max() > 6:
        raise ValueError("symbol code > 6")
    n, l = rows.shape
    stride = col_stride(n)
    cols = np.full((l, stride * 2), SYM_NONE, dtype=np.uint8)
    cols[:, :n] = rows.T
    return (cols[:, 0::2] | (cols[:, 1::2] << 4)).astype(np.uint8)


def plane_stride(n_reads: int) -> int:
    """Bytes per plane: ceil(n/1024) * 128 (include/juliet_hip.h jl_plane_stride)."""
    return (n_reads + 1023) // 1024 * 128


def pack_planes(rows: np.ndarray, stride: int = None) -> np.ndarray:
    """uint8[N][L] codes -> uint8[L][3][plane_stride]: plane k of column c holds bit k of every read's code, read i in bit
    i & 7 of byte i >> 3; reads past N are padding = code 6."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    if rows.ndim != 2:
        raise ValueError("rows must be [n_reads][n_cols]")
    if rows.size and rows.max() > 6:
        raise ValueError("symbol code > 6")
    n, l = rows.shape
    stride = plane_stride(n) if stride is None else stride
    cols = np.full((l, stride * 8), SYM_NONE, dtype=np.uint8)
    cols[:, :n] = rows.T
    out = np.empty((l, 3, stride), dtype=np.uint8)
    for k in range(3):
        out[:, k, :] = np.packbits((cols >> k) & 1, axis=1, bitorder="little")
    return out


def unpack_planes(planes: np.ndarray, n_reads: int) -> np.ndarray:
    """Inverse of pack_planes: uint8[L][3][plane_stride] -> uint8[N][L]."""
    planes = np.ascontiguousarray(planes, dtype=np.uint8)
    l = planes.shape[0]
    cols = np.zeros((l, planes.shape[2] * 8), dtype=np.uint8)
    for k in range(3):
        cols |= np.unpackbits(planes[:, k, :], axis=1, bitorder="little") << k
    return np.ascontiguousarray(cols[:, :n_reads].T)


def unpack_columns(packed: np.ndarray, n_reads: int) -> np.ndarray:
    """Inverse of pack_columns: uint8[L][col_stride] -> uint8[N][L]."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    l, stride = packed.shape
    cols = np.empty((l, stride * 2), dtype=np.uint8)
    cols[:, 0::2] = packed & 15
    cols[:, 1::2] = packed >> 4
    return np.ascontiguousarray(cols[:, :n_reads].T)


def encode(text_rows) -> np.ndarray:
    """List of equal-length strings over 'ACGT-N ' -> uint8 codes (tests)."""
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(ALPHABET):
        lut[ord(ch)] = i
    arr = np.array([np.frombuffer(r.encode(), dtype=np.uint8) for r in text_rows])
    out = lut[arr]
    if (out == 255).any():
        raise ValueError("character outside 'ACGT-N '")
    return out


def codon_index(codon: str) -> int:
    return 16 * "ACGT".index(codon[0]) + 4 * "ACGT".index(codon[1]) + "ACGT".index(codon[2])


def codon_string(idx: int) -> str:
    return "ACGT"[idx >> 4] + "ACGT"[(idx >> 2) & 3] + "ACGT"[idx & 3]
