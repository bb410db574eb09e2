"""Property tests of the CPU restatement: hypothesis over small random MSAs."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

import oracle_lib
from minorseq_amd import msa

ORC = oracle_lib.load()


@st.composite
def matrices(draw):
    n = draw(st.integers(1, 60))
    l = draw(st.integers(3, 24))
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 4, size=l)
    m = np.tile(base, (n, 1)).astype(np.uint8)
    noise = rng.random((n, l))
    m[noise < 0.10] = rng.integers(0, 4, size=int((noise < 0.10).sum()))
    m[(noise >= 0.10) & (noise < 0.16)] = msa.SYM_GAP
    m[(noise >= 0.16) & (noise < 0.22)] = msa.SYM_MASK
    lo = rng.integers(0, l // 2 + 1, size=n)
    hi = l - rng.integers(0, l // 2 + 1, size=n)
    cols = np.arange(l)[None, :]
    m[(cols < lo[:, None]) | (cols >= hi[:, None])] = msa.SYM_NONE
    return m


@settings(max_examples=120, deadline=None)
@given(matrices())
def test_pileup_and_histogram_invariants(m):
    n, l = m.shape
    col = ORC.pileup(m)
    assert (col.sum(axis=1) == (m != msa.SYM_NONE).sum(axis=0)).all()      # one symbol per covering read
    starts = np.arange(0, l - 2, dtype=np.uint32)
    hist, cov = ORC.codon_hist(m, starts)
    assert (hist.sum(axis=1) == cov).all()
    real = col[:, :4].sum(axis=1)
    assert (cov <= np.minimum.reduce([real[starts + k] for k in range(3)])).all()
    assert (cov >= n - sum((m[:, starts + k] > 3).sum(axis=0) for k in range(3))).all()
    # pack/unpack is lossless
    assert (msa.unpack_columns(msa.pack_columns(m), n) == m).all()


@settings(max_examples=60, deadline=None)
@given(matrices(), st.integers(0, 2**31 - 1))
def test_call_and_phase_invariants(m, seed):
    n, l = m.shape
    genes = np.array([(1, l + 1)], dtype=oracle_lib.GENE)
    prm = oracle_lib.default_params(alpha=0.3, n_tests=1.0)
    v = ORC.call(m, genes, params=prm)
    assert (v["count"] <= v["coverage"]).all() and (v["codon"] != v["ref_codon"]).all()
    assert (v["p_value"] < 0.3).all() and (v["p_value"] >= 0).all()
    order = np.lexsort((v["codon"], v["codon_pos"], v["gene"]))
    assert (order == np.arange(len(v))).all()                               # rows are sorted
    ph = ORC.phase(m, v, min_reads=2)
    s = ph["summary"]
    if s["n_positions"]:
        assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == n
        assert ph["hap_count"].sum() == s["reported_reads"]
        assert (np.diff(ph["hap_count"].astype(np.int64)) <= 0).all()
        assert (np.diagonal(ph["cooc"]) == (ph["hit"] * ph["hap_count"][None, :]).sum(axis=1)).all()
    # permuting the reads changes nothing but the read order
    perm = np.random.default_rng(seed).permutation(n)
    v2 = ORC.call(m[perm], genes, params=prm)
    assert (v2 == v).all()
    ph2 = ORC.phase(m[perm], v2, min_reads=2)
    assert ph2["summary"] == s and (ph2["read_hap"] == ph["read_hap"][perm]).all()
