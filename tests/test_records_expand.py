"""Pins tests/records_expand.py — the numpy statement of "records -> matrix" that the full-size device-ingest tests compare
against — on the CPU: it must invert rows_to_records (the record builder of the small ingest tests, noise ops included),
agree with its own three-nested-loops form, and turn juliet-synth's raw records back into synth.rows."""
import numpy as np
import pytest

import records_expand
from minorseq_amd import synth
from test_gpu_parity import rows_to_records

KEYS = ("pos", "cigar", "cig_off", "seq4", "seq_off", "qual", "qual_off")


@pytest.mark.parametrize("n,l,win", [(60, 90, (0, 90)), (200, 300, (37, 251)), (17, 700, (100, 700))])
def test_expander_inverts_the_record_builder(n, l, win):
    rng = np.random.default_rng(n + l)
    sp = synth.SynthParams(seed=n + l, partial_rate=0.3, del_rate=0.03, mask_rate=0.03, sub_rate=0.01)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    rows[5, 40:60] = 6
    rows[7] = 6
    rec = dict(zip(KEYS, rows_to_records(rows, ref, rng)))
    b, e = win
    got = records_expand.expand(rec, e - b, b, slab_cells=4096)
    assert (got == rows[:, b:e]).all()
    assert (records_expand.expand_reference(rec, e - b, b) == got).all()
    # the builder gives every aligned base quality 93, every clipped / inserted one less: a threshold of 94 masks all bases
    masked = records_expand.expand(rec, e - b, b, min_qv=94)
    exp = rows[:, b:e].copy()
    exp[exp < 4] = 5
    assert (masked == exp).all()
    assert (records_expand.expand(rec, e - b, b, min_qv=50) == rows[:, b:e]).all()
    # a range of reads
    assert (records_expand.expand(rec, e - b, b, read_begin=3, read_end=11) == rows[3:11, b:e]).all()
    bad = dict(rec)
    bad["cigar"] = rec["cigar"].copy()
    bad["cigar"][0] &= ~np.uint32(15)
    with pytest.raises(ValueError):
        records_expand.expand(bad, l)


def test_expander_gives_back_the_generators_rows():
    n, l = 3000, 450
    sp = synth.SynthParams(seed=11)
    rows = synth.rows(sp, l, 0, n, synth.reference(11, l))
    plain = synth.raw_records(11, n, l)
    assert (records_expand.expand(plain, l) == rows).all()
    # insertions, clips and poor qualities change nothing until a threshold asks for the qualities
    noisy = synth.raw_records(11, n, l, extra=("--ins-ppm", "3000", "--clips", "--low-qv-ppm", "20000"))
    assert len(noisy["cigar"]) > len(plain["cigar"]) and len(noisy["seq4"]) > len(plain["seq4"])
    assert (noisy["qual"] < 20).any()
    assert (records_expand.expand(noisy, l) == rows).all()
    m = records_expand.expand(noisy, l - 50, 20, min_qv=20)
    assert (m == records_expand.expand_reference(noisy, l - 50, 20, min_qv=20)).all()
    diff = m != rows[:, 20:l - 30]
    assert diff.any() and (m[diff] == 5).all() and 0.01 < diff.mean() < 0.03
