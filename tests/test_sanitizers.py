"""Host code under AddressSanitizer + UBSan (SURVEY.md §4 test plan): the CPU oracle and the front end's
BAM/BGZF/config/ingest paths.  CPU only (GPU sanitizers are not available on this pool)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "minorseq_amd", "host")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1"]


@pytest.fixture(scope="module")
def san_bins(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    synth = str(d / "juliet-synth-san")
    subprocess.check_call(["g++", "-std=c++17", *SAN, "-I" + os.path.join(ROOT, "include"), "-o", synth,
                           os.path.join(HOST, "synth_bam.cpp"), "-lz", "-lpthread"])
    # the front end's GPU-free diagnostics (--dump-msa / --dump-config) need no device library at run time, but the
    # binary links the C ABI: use the real .so
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "minorseq_amd", "csrc")])
    juliet = str(d / "juliet-san")
    subprocess.check_call(["g++", "-std=c++17", *SAN, "-I" + os.path.join(ROOT, "include"), "-o", juliet,
                           os.path.join(HOST, "juliet_main.cpp"), "-L" + os.path.join(ROOT, "minorseq_amd"), "-ljuliet_hip",
                           "-lz", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "minorseq_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    orc = str(d / "oracle_san_test")
    src = str(d / "drive.c")
    open(src, "w").write(r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "juliet_oracle.c"
int main(void) {
    enum { N = 700, L = 90 };
    uint8_t *m = malloc(N * L);
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < N * L; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; m[i] = (x % 100) < 90 ? (i % L) % 4 : x % 7; }
    for (int i = 0; i < 40; ++i) m[i * L + 30] = 3;   /* a planted variant */
    uint32_t col[L * 6];
    orc_pileup(m, N, L, col);
    orc_gene genes[2] = {{1, L + 1}, {2, L}};
    orc_params prm = {0.3, 1.0, {0.998826, 5.8e-5, 1e-3}, 0, 0};
    orc_variant v[256];
    uint32_t n = 0;
    orc_call(m, N, L, 0, genes, 2, NULL, 0, &prm, v, 256, &n);
    if (n > 256) n = 256;
    orc_phase_summary s;
    uint32_t *pos = malloc((n + 1) * 4), *hc = malloc(ORC_MAX_HAP * 4), *hf = malloc(ORC_MAX_HAP * 4), *co = malloc((size_t)(n + 1) * (n + 1) * 4);
    uint8_t *hp = malloc((size_t)ORC_MAX_HAP * (n + 1)), *hit = malloc((size_t)(n + 1) * ORC_MAX_HAP);
    uint16_t *rh = malloc(N * 2);
    orc_phase(m, N, L, v, n, 3, &s, pos, hc, hf, hp, hit, rh, co);
    double lp, p = orc_fisher(29, 2500, 1, 2528, 0, &lp);
    printf("%u variants, %u haplotypes, p=%g\n", n, s.n_haplotypes, p);
    free(m); free(pos); free(hc); free(hf); free(co); free(hp); free(hit); free(rh);
    return !(n > 0 && s.reported_reads + s.insufficient_reads + s.damaged_reads == N);
}
''')
    subprocess.check_call(["gcc", "-std=c11", *SAN, "-I" + os.path.join(ROOT, "oracle"), "-o", orc, src, "-lm"])
    return d, synth, juliet, orc


def test_oracle_under_asan_ubsan(san_bins):
    _, _, _, orc = san_bins
    out = subprocess.run([orc], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]


def test_front_end_ingest_under_asan_ubsan(san_bins):
    d, synth, juliet, _ = san_bins
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")   # the HIP runtime the .so links keeps allocations alive
    bam, cfg, msa_out, cfg_out = (str(d / x) for x in ("s.bam", "s.json", "s.msa", "c.json"))
    r = subprocess.run([synth, "--reads", "800", "--cols", "600", "--seed", "3", "--partial", "0.3", "--ref-offset", "100",
                        "-o", bam, "--config-out", cfg], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([juliet, "-c", cfg, "--dump-msa", msa_out, bam], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    # the same with the filters that make the one-pass record parser read qualities and tags
    r = subprocess.run([juliet, "-c", cfg, "--min-qv", "10", "--min-rq", "0.5", "--dump-msa", msa_out, bam],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([juliet, "-c", cfg, "-r", "130-400", "--dump-config", cfg_out], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    # malformed inputs must fail cleanly, not crash
    open(str(d / "trunc.bam"), "wb").write(open(bam, "rb").read()[:5000])
    r = subprocess.run([juliet, "--dump-msa", msa_out, str(d / "trunc.bam")], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
    open(str(d / "garbage.bam"), "wb").write(b"\x1f\x8b" + b"not a bam" * 50)
    r = subprocess.run([juliet, "--dump-msa", msa_out, str(d / "garbage.bam")], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "AddressSanitizer" not in r.stderr


def test_front_end_pipeline_under_tsan(tmp_path):
    """The front end is four kinds of threads (inflate workers, the record parser, the uploader that hands chunks to the
    device, the context start-up): ThreadSanitizer over the hand-overs.  No GPU is needed for that — without one the
    context fails, the uploader still takes every chunk off the queue and recycles it, and the run ends with the
    'no usable GPU' message (exit 3); with one it runs through."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "minorseq_amd", "csrc")])
    synth = os.path.join(ROOT, "minorseq_amd", "bin", "juliet-synth")
    if not os.path.exists(synth):
        subprocess.check_call(["make", "-s", "-C", HOST])
    juliet = str(tmp_path / "juliet-tsan")
    subprocess.check_call(["g++", "-std=c++17", "-fsanitize=thread", "-g", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", juliet,
                           os.path.join(HOST, "juliet_main.cpp"), "-L" + os.path.join(ROOT, "minorseq_amd"), "-ljuliet_hip",
                           "-lz", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "minorseq_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    bam, cfg = str(tmp_path / "t.bam"), str(tmp_path / "t.json")
    subprocess.check_call([synth, "--reads", "30000", "--cols", "300", "--seed", "9", "-o", bam, "--config-out", cfg])   # 4 chunks
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
    r = subprocess.run([juliet, "-c", cfg, "--mode-phasing", bam, str(tmp_path / "o.json")], capture_output=True, text=True, env=env,
                       timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode in (0, 3), r.stderr[-2000:]
