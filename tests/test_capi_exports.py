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


def test_pod_layouts():
    assert capi.VARIANT.itemsize == 48          # the all-gather row
    assert capi.GENE.itemsize == 8
    assert capi.SUMMARY.itemsize == 32
    assert C.sizeof(capi.Params) == 64
    lib = capi.load_library()
    assert lib.jl_col_stride(1) == 128 and lib.jl_col_stride(256) == 128 and lib.jl_col_stride(257) == 256
    assert lib.jl_col_stride(100000) == 50048
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
    assert ver == "3"
    if n == "0":
        assert rc == "-2" and "no CPU fallback" in msg      # loud failure without a device
