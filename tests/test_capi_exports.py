"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/juliet_hip.h declares,
its PODs have the documented sizes, and without a GPU it refuses loudly (no CPU fallback).  No compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from minorseq_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "juliet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    declared = _declared()
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/juliet_hip.h but not exported"
    assert sorted(capi.EXPORTS) == declared


def test_header_is_the_export_list():
    """-fvisibility=hidden + the header's visibility pragma + csrc/exports.map: the dynamic symbol table of the library
    holds exactly the functions include/juliet_hip.h declares — no internal jl_* helper, no mangled C++ symbol, no kernel
    handle."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    kinds = {line.split()[-2] for line in out.splitlines() if line.strip()}
    assert exported == _declared()
    assert kinds == {"T"}


def test_pod_layouts():
    assert capi.VARIANT.itemsize == 48          # the all-gather row
    assert capi.GENE.itemsize == 8
    assert capi.SUMMARY.itemsize == 32
    assert C.sizeof(capi.Params) == 64
    lib = capi.load_library()
    assert lib.jl_col_stride(1) == 128 and lib.jl_col_stride(256) == 128 and lib.jl_col_stride(257) == 256
    assert lib.jl_col_stride(100000) == 50048
    # the resident planes: whole 128-byte lines, 1024 reads each; 3 bits per cell
    assert lib.jl_plane_stride(1) == 128 and lib.jl_plane_stride(1024) == 128 and lib.jl_plane_stride(1025) == 256
    assert lib.jl_plane_stride(100000) == 12544 and 3 * 3000 * lib.jl_plane_stride(100000) < 113e6
    assert lib.jl_strerror(0) == b"ok" and b"gfx950" in lib.jl_strerror(-2)


def test_no_cpu_fallback():
    lib = capi.load_library()
    if lib.jl_device_count() > 0:
        pytest.skip("a gfx950 device is present")
    with pytest.raises(capi.JulietError) as e:
        capi.Juliet(0)
    assert e.value.status == -2 and "no CPU fallback" in str(e.value)


def test_haplotype_names():
    # doc/JULIET.md:198  [A-Z]{1}[a-z]?
    names = [capi.haplotype_name(h) for h in range(capi.MAX_HAPLOTYPES)]
    assert names[0] == "A" and names[25] == "Z" and names[26] == "Aa" and names[51] == "Az" and names[52] == "Ba"
    assert names[-1] == "Zz" and len(set(names)) == 702
    assert all(re.fullmatch(r"[A-Z][a-z]?", n) for n in names)


def test_header_is_plain_c_and_links(tmp_path):
    """include/juliet_hip.h is a C header (no C++/torch types): a C99 program compiles against it, links the
    library and runs without a GPU (it only asks for the ABI version and the device count)."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('#include <stdio.h>\n#include "juliet_hip.h"\n'
                   'int main(void) {\n'
                   '    jl_params p = {0.01, 0.0, {0.998826, 5.8e-5, 1.0e-3}, 0, 0, -1.0, -1.0};\n'
                   '    jl_ctx *ctx = 0;\n'
                   '    int n = jl_device_count();\n'
                   '    int rc = n ? 0 : jl_ctx_create(0, 0, &ctx);\n'
                   '    printf("%d %d %d %s|%s\\n", jl_abi_version(), n, rc, jl_strerror(rc), jl_last_error(0));\n'
                   '    return (int)(p.alpha > 1.0);\n}\n')
    exe = tmp_path / "use"
    lib_dir = os.path.join(ROOT, "minorseq_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-o", str(exe), str(src), "-L" + lib_dir, "-ljuliet_hip", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ver, n, rc, msg = out.stdout.strip().split(" ", 3)
    assert ver == "5"
    if n == "0":
        assert rc == "-2" and "no CPU fallback" in msg      # loud failure without a device


C_MERGE = r"""
#include <stdio.h>
#include <string.h>
#include "juliet_hip.h"
/* The host side of phasing sharded by reads, from plain C: two windows' tables -> one; two slices' groups -> the merged
 * groups; the haplotypes of the MERGED counts (doc/JULIET.md:253-254); the schedule of the column-slice exchange. */
int main(void) {
    jl_variant a[2], b[1], merged[3];
    const jl_variant *tabs[2];
    uint32_t cnt[2] = {2, 1}, wb[2] = {0, 300}, n = 0, m = 0, k;
    uint8_t p0[4] = {3, 1, 0, 2}, p1[6] = {0, 2, 3, 1, 3, 0}, mp[10];
    const uint8_t *pats[2];
    uint32_t c0[2] = {4, 6}, c1[3] = {5, 6, 9}, ng[2] = {2, 3}, ps[2] = {2, 2}, i0[2], i1[3];
    const uint32_t *cnts[2];
    uint32_t *idx[2];
    uint64_t mc[5];
    uint32_t pos_cols[2] = {0, 3}, hap_count[702], cooc[4];
    uint8_t hap_pattern[702 * 2], hit[2 * 3];
    uint16_t hom[3];
    jl_phase_summary part[2], sum;
    jl_variant v[2];
    jl_xwin_op ops[5];
    uint32_t wn[2] = {302, 300}, n_ops = 0;
    int32_t wr[2] = {0, 1};
    uint64_t sb[3] = {0, 512, 1000};
    memset(a, 0, sizeof a); memset(b, 0, sizeof b); memset(part, 0, sizeof part); memset(v, 0, sizeof v);
    a[0].gene = 0; a[0].codon_pos = 9; a[0].codon = 7; a[0].col = 24;
    a[1].gene = 0; a[1].codon_pos = 3; a[1].codon = 1; a[1].col = 6;
    b[0].gene = 0; b[0].codon_pos = 104; b[0].codon = 2; b[0].col = 9;
    tabs[0] = a; tabs[1] = b;
    if (jl_merge_tables(tabs, cnt, wb, 2, merged, 3, &n) != JL_OK || n != 3) return 1;
    if (merged[0].col != 6 || merged[1].col != 24 || merged[2].col != 309) return 2;
    if (jl_merge_tables(tabs, cnt, wb, 2, merged, 2, &n) != JL_ERR_OVERFLOW || n != 3) return 3;
    pats[0] = p0; pats[1] = p1; cnts[0] = c0; cnts[1] = c1; idx[0] = i0; idx[1] = i1;
    if (jl_merge_groups(pats, ps, cnts, ng, 2, 2, mp, mc, 5, &m, idx) != JL_OK || m != 3) return 4;
    if (mp[0] != 0 || mp[1] != 2 || mc[0] != 11 || mc[1] != 9 || mc[2] != 10 || i0[0] != 2 || i1[2] != 1) return 5;
    v[0].col = 0; v[0].codon = 3; v[1].col = 3; v[1].codon = 2;
    part[0].damaged_reads = 7; part[1].damaged_reads = 5; part[1].marginal_gap = 2;
    if (jl_select_haplotypes(mp, mc, m, 2, v, 2, pos_cols, 10, part, 2, &sum, hap_count, hap_pattern, hit, 3, cooc, hom) != JL_OK) return 6;
    if (sum.n_haplotypes != 2 || sum.reported_reads != 21 || sum.insufficient_reads != 9 || sum.damaged_reads != 12 || sum.marginal_gap != 2) return 7;
    if (hap_count[0] != 11 || hap_count[1] != 10 || hom[0] != 0 || hom[1] != JL_HAP_INSUFFICIENT || hom[2] != 1) return 8;
    if (hit[0] != 0 || hit[1] != 1 || hit[3] != 1 || hit[4] != 0 || cooc[0] != 10 || cooc[3] != 11 || cooc[1] != 0) return 9;
    /* two ranks, one window each; both positions belong to rank 1's window except column 6 */
    if (jl_xwin_slice_plan(wb, wn, wr, 2, merged, 3, sb, 2, 0, ops, 5, &n_ops) != JL_OK || n_ops != 3) return 10;
    if (ops[0].op != JL_XWIN_OP_LOCAL || ops[1].op != JL_XWIN_OP_SEND || ops[2].op != JL_XWIN_OP_RECV) return 11;
    if (ops[0].k_count != 2 || ops[1].peer != 1 || ops[1].n_reads != 488 || ops[1].dst_stride != 128 || ops[1].bytes != 9 * 2 * 128) return 12;
    if (ops[2].k_begin != 2 || ops[2].k_count != 1 || ops[2].dst_stride != 128 || ops[2].dst_offset != 9 * 2 * 128) return 13;
    for (k = 0; k < n_ops; ++k) printf("%d:%d:%llu ", ops[k].op, ops[k].peer, (unsigned long long)ops[k].bytes);
    printf("\n");
    return 0;
}
"""


def test_merge_and_schedule_from_c99(tmp_path):
    """jl_merge_tables / jl_merge_groups / jl_select_haplotypes / jl_xwin_slice_plan are host-only: a C99 program
    drives them without a GPU and checks the answers."""
    import subprocess
    src = tmp_path / "merge.c"
    src.write_text(C_MERGE)
    exe = tmp_path / "merge"
    lib_dir = os.path.join(ROOT, "minorseq_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-o", str(exe), str(src), "-L" + lib_dir, "-ljuliet_hip", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.strip() == "0:0:2304 1:1:2304 2:1:1152"
