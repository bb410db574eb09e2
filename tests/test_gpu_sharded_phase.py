"""Cross-window phasing with the READS sharded (SURVEY.md §8e option A): every shard groups its slice of the reads on the
device, the group tables are merged on the host, the merged groups' haplotype ids go back to the shards.  The result must
be the unsharded one — summary, haplotypes, hit, co-occurrence and every read's id — for any number of shards."""
import os

import numpy as np
import pytest

from minorseq_amd import capi, msa, sharding, synth

pytestmark = pytest.mark.gpu


def assert_same(got, exp, n_var):
    assert got["summary"] == exp["summary"]
    h = exp["summary"]["n_haplotypes"]
    assert (got["pos_cols"] == exp["pos_cols"]).all()
    assert (got["hap_count"] == exp["hap_count"]).all()
    assert (got["hap_pattern"] == exp["hap_pattern"]).all()
    assert (got["hit"][:n_var, :h] == exp["hit"]).all()
    assert (got["cooc"][:n_var, :n_var] == exp["cooc"]).all()
    assert (got["read_hap"] == exp["read_hap"]).all()


def windows_of(rows, n, l, world, genes, ref):
    wb = sharding.window_bounds(l, world)
    prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
    ctxs, tables = [], []
    for b, e in wb:
        c = capi.Juliet(0)
        c.upload_columns(msa.pack_columns(rows[:, b:e]), n, win_begin=b)
        c.pileup_async(genes, ref)
        c.call_async(prm)
        tables.append(c.call_fetch())
        ctxs.append(c)
    return ctxs, sharding.merge_tables(tables, [b for b, _ in wb])


@pytest.mark.parametrize("n,shards", [(7000, 1), (7000, 2), (7000, 3), (7000, 8), (1000, 8), (100_000, 4)])
def test_sharded_by_reads_equals_unsharded(oracle, n, shards):
    l = 900
    sp = synth.SynthParams(seed=29 + n, minor_permille=(70, 60, 50, 40), partial_rate=0.15)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    rows[: n // 20, 700:703] = (rows[: n // 20, 700:703] + 1) % 4          # haplotypes that span windows
    rows[n // 40: n // 16, 820:823] = (rows[n // 40: n // 16, 820:823] + 2) % 4
    ctxs, merged = windows_of(rows, n, l, 3, genes, ref)
    exp = oracle.phase(rows, oracle.call(rows, genes, refseq=ref))
    got, pos_global = capi.phase_sharded_by_reads(ctxs, merged, shards)
    assert_same(got, exp, len(merged))
    # the replicated form (option B) on the same windows gives the same
    rep, _ = capi.phase_across_windows(ctxs, merged)
    assert rep["summary"] == got["summary"] and (rep["read_hap"] == got["read_hap"]).all()
    for c in ctxs:
        c.close()


def test_a_group_below_the_threshold_in_every_shard_is_still_reported(oracle):
    """The >= 10 reads rule applies to the MERGED count: a pattern carried by 4 reads in each of three slices is reported
    (12 reads) although no slice holds 10 of them; 9 reads in a single slice are not, nor are the 4 reads of the first
    slice that carry the major edit as well (a pattern of their own)."""
    n, l = 4 * 1024, 60
    ref = synth.reference(5, l)
    rows = np.tile(ref, (n, 1)).astype(np.uint8)
    rows[: n // 3, 30:33] = (rows[: n // 3, 30:33] + 1) % 4              # a real minor variant: the call
    for s in range(4):                                                    # 4 reads per slice share a second edit at 9..11
        rows[s * 1024 + 500: s * 1024 + 504, 9:12] = (rows[s * 1024 + 500: s * 1024 + 504, 9:12] + 2) % 4
    rows[2048 + 700: 2048 + 709, 45:48] = (rows[2048 + 700: 2048 + 709, 45:48] + 3) % 4      # 9 reads, one slice only
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    # phase against a hand-made table: the three edited codons
    import oracle_lib
    full = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(n_tests=1.0, alpha=1.0))
    keep = np.isin(full["col"], [9, 30, 45])
    table = full[keep]
    assert len(np.unique(table["col"])) == 3
    exp = oracle.phase(rows, table)
    c = capi.Juliet(0)
    c.upload_columns(msa.pack_columns(rows), n)
    got, _ = capi.phase_sharded_by_reads([c], table, 4)
    assert_same(got, exp, len(table))
    counts = sorted(got["hap_count"].tolist())
    assert 12 in counts and 9 not in counts and 4 not in counts
    assert got["summary"]["insufficient_reads"] == 9 + 4
    c.close()


def test_many_positions_take_the_multi_word_keys(oracle):
    """More than ten variant positions: the exporting run switches to multi-word keys like any other (jl_phase_groups_fetch
    re-runs it), and patterns of 14 positions merge the same way."""
    n, l = 6000, 300
    sp = synth.SynthParams(seed=91, minor_permille=(80, 70, 60, 50))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    rng = np.random.default_rng(3)
    for k in range(12):                                                   # twelve more edited codons, ~3 % of the reads each
        who = rng.choice(n, n // 30, replace=False)
        c0 = 3 * (5 + 7 * k)
        rows[who, c0:c0 + 3] = (rows[who, c0:c0 + 3] + 1 + k % 3) % 4
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    table = oracle.call(rows, genes, refseq=ref)
    assert len(np.unique(table["col"])) > 10
    exp = oracle.phase(rows, table)
    c = capi.Juliet(0)
    c.upload_columns(msa.pack_columns(rows), n)
    got, _ = capi.phase_sharded_by_reads([c], table, 3)
    assert_same(got, exp, len(table))
    c.close()


def test_slice_exchange_over_rccl_single_rank(oracle):
    """The RCCL form of the slice exchange (ncclSend / ncclRecv per position and rank; a rank's own slice by a device copy)
    with a one-rank communicator: everything but the wire."""
    import ctypes as C
    n, l = 5000, 300
    sp = synth.SynthParams(seed=41, minor_permille=(70, 60, 50, 40), partial_rate=0.1)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    win = capi.Juliet(0)
    win.upload_columns(msa.pack_columns(rows), n)
    win.pileup_async(genes, ref)
    win.call_async(capi.default_params())
    table = win.call_fetch()
    idbuf = np.zeros(128, dtype=np.uint8)
    assert win.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
    comm = C.c_void_p()
    win._chk(win.lib.jl_comm_create(win.h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
    pc = capi.Juliet(0)
    try:
        remapped, pos_global, vp = pc.xwin_assemble_slice_rccl(win, comm, [0], [l], table, sharding.read_slices(n, 1))
    finally:
        win.lib.jl_comm_destroy(comm)
    pc._shape(n, 3 * vp, pc.lib.jl_col_stride(n))
    pc.phase_groups_async(remapped)
    t = pc.phase_groups_fetch()
    patterns, counts, index = sharding.merge_groups([t])
    ph = sharding.select_haplotypes(patterns, counts, remapped, t["pos_cols"], 10, [t["summary"]])
    ph["read_hap"] = pc.phase_regroup(ph["hap_of_merged"][index[0]].astype(np.uint16), ph["summary"]["n_haplotypes"])
    ph["pos_cols"] = pos_global
    assert_same(ph, oracle.phase(rows, table), len(table))
    # an unaligned slice start is refused
    with pytest.raises(capi.JulietError):
        pc.xwin_assemble_slice_local([win], table, 100, 1000)
    pc.close()
    win.close()


def test_state_and_argument_errors_are_loud(oracle):
    n, l = 3000, 120
    sp = synth.SynthParams(seed=8, minor_permille=(70, 60, 50, 40))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    c = capi.Juliet(0)
    c.upload_columns(msa.pack_columns(rows), n)
    table = oracle.call(rows, genes, refseq=ref)
    with pytest.raises(capi.JulietError) as e:            # nothing exported yet
        c.phase_groups_fetch()
    assert "jl_phase_groups_async" in str(e.value)
    with pytest.raises(capi.JulietError):
        c.phase_regroup(np.zeros(1, dtype=np.uint16), 1)
    c.phase_async(table, 10)                               # a normal phase run does not export either
    c.phase_fetch()
    with pytest.raises(capi.JulietError):
        c.phase_groups_fetch()
    c.phase_groups_async(table)
    t = c.phase_groups_fetch()
    g = len(t["counts"])
    assert g >= 5 and int(t["counts"].sum()) == t["summary"]["insufficient_reads"] == n - t["summary"]["damaged_reads"]
    with pytest.raises(capi.JulietError) as e:            # a haplotype id outside the merged set
        c.phase_regroup(np.full(g, 7, dtype=np.uint16), 3)
    assert "haplotype 7 of 3" in str(e.value)
    with pytest.raises(capi.JulietError):                  # more haplotypes than have names
        c.phase_regroup(np.zeros(g, dtype=np.uint16), 703)
    # every group insufficient: all clean reads carry 0xFFFE, flagged ones 0xFFFF
    ids = c.phase_regroup(np.full(g, capi.HAP_INSUFFICIENT, dtype=np.uint16), 0)
    exp = oracle.phase(rows, table)
    assert ((ids == capi.HAP_DAMAGED) == (exp["read_hap"] == capi.HAP_DAMAGED)).all()
    assert (ids[ids != capi.HAP_DAMAGED] == capi.HAP_INSUFFICIENT).all()
    # the whole-path run afterwards is not disturbed by the export state
    c.run_async(genes, ref, capi.default_params(), None, True, 10, True)
    v = c.run_view() or c.run_fetch(True, True, cap_var=64)
    assert (np.asarray(v["phase"]["hap_count"]) == exp["hap_count"]).all()
    c.close()


def test_random_shapes_against_the_unsharded_oracle(oracle):
    """Random reads, window counts, shard counts, error rates (up to 200 variant positions = 20-word keys; samples where
    not one read is clean): the sharded run equals the unsharded oracle in everything."""
    rng = np.random.default_rng(int(os.environ.get("JL_TEST_SEED", "12345")))     # (JL_TEST_SEED: a soak with other draws)
    seen_many, seen_none = False, False
    for trial in range(12):
        n = int(rng.integers(300, 30000))
        l = int(rng.choice([120, 300, 600]))
        shards, world = int(rng.integers(1, 7)), int(rng.integers(1, 4))
        sp = synth.SynthParams(seed=int(rng.integers(1, 1 << 30)), minor_permille=tuple(int(x) for x in rng.integers(20, 120, 4)),
                               partial_rate=float(rng.choice([0.0, 0.1, 0.4])), sub_rate=float(rng.choice([1.75e-4, 5e-3])))
        ref = synth.reference(sp.seed, l)
        rows = synth.rows(sp, l, 0, n, ref)
        genes = np.array([(1, 3 * (l // 3) + 1)], dtype=capi.GENE)
        ctxs, merged = windows_of(rows, n, l, world, genes, ref)
        full = oracle.call(rows, genes, refseq=ref)
        assert len(merged) == len(full) and all((merged[k] == full[k]).all() for k in ("gene", "codon_pos", "col", "codon", "count", "coverage"))
        exp = oracle.phase(rows, full)
        got, _ = capi.phase_sharded_by_reads(ctxs, merged, shards)
        assert_same(got, exp, len(full))
        seen_many |= exp["summary"]["n_positions"] > 10
        seen_none |= exp["summary"]["n_positions"] > 0 and exp["summary"]["reported_reads"] + exp["summary"]["insufficient_reads"] == 0
        for c in ctxs:
            c.close()
    if "JL_TEST_SEED" not in os.environ:      # (what the default draws are known to contain)
        assert seen_many and seen_none


# ---------------------------------------------------------------------------------------------------------------------
# The whole sequence behind ONE call of the C ABI (jl_xwin_phase_sharded): tables -> merge -> plan -> packed column
# exchange -> grouping + export -> gather -> merge + selection -> per-read ids.

def _one_rank_comm(ctx):
    import ctypes as C
    idbuf = np.zeros(128, dtype=np.uint8)
    assert ctx.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
    comm = C.c_void_p()
    ctx._chk(ctx.lib.jl_comm_create(ctx.h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
    return comm


def _run_windows(rows, n, l, k_windows, genes, ref, whole_path=True):
    wb = sharding.window_bounds(l, k_windows)
    prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
    ctxs = []
    for b, e in wb:
        c = capi.Juliet(0)
        c.upload_columns(msa.pack_columns(rows[:, b:e]), n, win_begin=b)
        if whole_path:
            c.run_async(genes, ref, prm, None, False, 10, False)     # call only: the table lands in the pinned result block
        else:
            c.pileup_async(genes, ref)
            c.call_async(prm)
        ctxs.append(c)
    return ctxs, wb


def assert_session(res, full, exp):
    m = res["merged"]
    assert len(m) == len(full)
    for k in ("gene", "codon_pos", "col", "ref_codon", "codon", "count", "coverage", "expected"):   # integers: bit-exact
        assert (m[k] == full[k]).all(), k
    assert len(m) == 0 or np.abs(m["p_value"] - full["p_value"]).max() <= 1e-10                    # north_star's tolerance
    assert_same(res, exp, len(full))


@pytest.mark.parametrize("n,k_windows,with_comm,whole_path", [(7000, 1, False, True), (7000, 3, False, True), (7000, 3, True, True),
                                                                (7000, 8, False, False), (100_000, 4, True, True), (1000, 2, False, False)])
def test_session_equals_unsharded(oracle, n, k_windows, with_comm, whole_path):
    l = 900
    sp = synth.SynthParams(seed=31 + n, minor_permille=(70, 60, 50, 40), partial_rate=0.15)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    rows[: n // 20, 700:703] = (rows[: n // 20, 700:703] + 1) % 4          # haplotypes that span windows
    rows[n // 40: n // 16, 820:823] = (rows[n // 40: n // 16, 820:823] + 2) % 4
    ctxs, wb = _run_windows(rows, n, l, k_windows, genes, ref, whole_path)
    full = oracle.call(rows, genes, refseq=ref)
    exp = oracle.phase(rows, full)
    comm = _one_rank_comm(ctxs[0]) if with_comm else None
    xw = capi.Xwin(ctxs, [b for b, _ in wb], [e - b for b, e in wb], [0] * k_windows, [0, n], comm)
    try:
        for _ in range(3):                       # a session is a step loop: the same answer every time
            assert_session(xw.phase(10), full, exp)
    finally:
        xw.close()
        if comm is not None:
            ctxs[0].lib.jl_comm_destroy(comm)
    for c in ctxs:
        c.close()


def test_session_many_positions_and_many_groups(oracle):
    """More than ten positions (multi-word keys) and more groups than a block or the by-value table holds (1024): the
    session grows its blocks and runs the step again; the haplotypes of the exported groups travel through HBM."""
    n, l = 30000, 300
    sp = synth.SynthParams(seed=77, minor_permille=(80, 70, 60, 50))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    rng = np.random.default_rng(5)
    for k in range(13):                                                   # thirteen more edited codons, 35 % of the reads each
        who = rng.choice(n, int(n * 0.35), replace=False)
        c0 = 3 * (5 + 7 * k)
        rows[who, c0:c0 + 3] = (rows[who, c0:c0 + 3] + 1 + k % 3) % 4
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ctxs, wb = _run_windows(rows, n, l, 2, genes, ref)
    full = oracle.call(rows, genes, refseq=ref)
    exp = oracle.phase(rows, full)
    assert exp["summary"]["n_positions"] > 10
    for comm_on in (False, True):
        comm = _one_rank_comm(ctxs[0]) if comm_on else None
        xw = capi.Xwin(ctxs, [b for b, _ in wb], [e - b for b, e in wb], [0, 0], [0, n], comm)
        res = xw.phase(10)
        assert res["n_groups"] > 1024
        assert_session(res, full, exp)
        assert_session(xw.phase(10), full, exp)
        xw.close()
        if comm is not None:
            ctxs[0].lib.jl_comm_destroy(comm)
    for c in ctxs:
        c.close()


def test_session_without_variants_and_bad_layouts(oracle):
    n, l = 3000, 120
    ref = synth.reference(3, l)
    rows = np.tile(ref, (n, 1)).astype(np.uint8)                           # not one variant
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ctxs, wb = _run_windows(rows, n, l, 2, genes, ref)
    xw = capi.Xwin(ctxs, [b for b, _ in wb], [e - b for b, e in wb], [0, 0], [0, n])
    res = xw.phase(10)
    assert len(res["merged"]) == 0 and res["summary"]["n_positions"] == 0 and (res["read_hap"] == capi.HAP_DAMAGED).all()
    xw.close()
    with pytest.raises(capi.JulietError):      # slices must cover the reads
        capi.Xwin(ctxs, [b for b, _ in wb], [e - b for b, e in wb], [0, 0], [0, n - 1])
    with pytest.raises(capi.JulietError):      # the layout must describe the contexts
        capi.Xwin(ctxs, [0, 10], [e - b for b, e in wb], [0, 0], [0, n])
    with pytest.raises(capi.JulietError):      # one rank, two windows: both are local
        capi.Xwin(ctxs[:1], [b for b, _ in wb], [e - b for b, e in wb], [0, 0], [0, n])
    for c in ctxs:
        c.close()


def test_session_refuses_a_busy_communicator(oracle):
    """The cross-window calls issue their collectives from the calling thread: with an asynchronous exchange of the
    communicator still uncollected they are refused (JL_ERR_STATE), and work again once it is collected."""
    import ctypes as C
    n, l = 4000, 300
    sp = synth.SynthParams(seed=13, minor_permille=(70, 60, 50, 40))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ctxs, wb = _run_windows(rows, n, l, 1, genes, ref)
    win = ctxs[0]
    comm = _one_rank_comm(win)
    full = oracle.call(rows, genes, refseq=ref)
    exp = oracle.phase(rows, full)
    xw = capi.Xwin(ctxs, [0], [l], [0], [0, n], comm)
    win._chk(win.lib.jl_allgather_variants_async(win.h, comm))             # requested, not collected
    with pytest.raises(capi.JulietError) as e:
        xw.phase(10)
    assert e.value.status == -4 and "uncollected" in str(e.value)
    pc = capi.Juliet(0)
    with pytest.raises(capi.JulietError) as e:
        pc.xwin_assemble_slice_rccl(win, comm, [0], [l], full, [0, n])
    assert e.value.status == -4
    rows_out = np.zeros(128, dtype=capi.VARIANT)
    cnt = np.zeros(1, dtype=np.uint32)
    win._chk(win.lib.jl_allgather_variants(win.h, comm, rows_out.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), 128))
    assert int(cnt[0]) == len(full)
    assert_session(xw.phase(10), full, exp)
    pc.close()
    xw.close()
    win.lib.jl_comm_destroy(comm)
    win.close()


def test_allgather_groups_single_rank(oracle):
    """jl_allgather_groups with a one-rank communicator: the exported groups come back as the fetch gives them."""
    import ctypes as C
    n, l = 5000, 300
    sp = synth.SynthParams(seed=43, minor_permille=(70, 60, 50, 40), partial_rate=0.1)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    c = capi.Juliet(0)
    c.upload_columns(msa.pack_columns(rows), n)
    table = oracle.call(rows, genes, refseq=ref)
    c.phase_groups_async(table)
    t = c.phase_groups_fetch()
    g, vp = len(t["counts"]), len(t["pos_cols"])
    comm = _one_rank_comm(c)
    cap, stride = 256, 16
    pats = np.zeros((1, cap, stride), dtype=np.uint8)
    cnts = np.zeros((1, cap), dtype=np.uint32)
    ng = np.zeros(1, dtype=np.uint32)
    parts = np.zeros(1, dtype=capi.SUMMARY)
    npos = C.c_uint32()
    c._chk(c.lib.jl_allgather_groups(c.h, comm, cap, stride, pats.ctypes.data_as(C.c_void_p), cnts.ctypes.data_as(C.c_void_p),
                                     ng.ctypes.data_as(C.c_void_p), parts.ctypes.data_as(C.c_void_p), C.byref(npos)))
    assert int(ng[0]) == g and npos.value == vp
    assert (cnts[0, :g] == t["counts"]).all() and (pats[0, :g, :vp] == t["patterns"]).all()
    assert int(parts[0]["damaged_reads"]) == t["summary"]["damaged_reads"]
    with pytest.raises(capi.JulietError) as e:     # a block too small for the groups: loud, on every rank alike
        c.lib.jl_allgather_groups.restype = C.c_int
        c._chk(c.lib.jl_allgather_groups(c.h, comm, 2, stride, pats.ctypes.data_as(C.c_void_p), cnts.ctypes.data_as(C.c_void_p),
                                         ng.ctypes.data_as(C.c_void_p), parts.ctypes.data_as(C.c_void_p), C.byref(npos)))
    assert e.value.status == -5
    # jl_phase_regroup answers for every exported group: a shorter table is refused
    with pytest.raises(capi.JulietError) as e:
        c.phase_regroup(np.zeros(g - 1, dtype=np.uint16), 1)
    assert "exported" in str(e.value)
    c.lib.jl_comm_destroy(comm)
    c.close()


# ---------------------------------------------------------------------------------------------------------------------
# More than one rank on the one GPU of the box: the ranks are threads, the communicator is the in-process one
# (jl_comm_create_inproc: every exchange = device copies between the ranks' buffers between two barriers; RCCL refuses two
# ranks on one device).  Everything of the multi-rank sequence runs for real — the table gather of several ranks, the
# exchange schedule with its packed sends and receives, grouping per read slice, the group gather, merge + selection,
# per-read ids of each rank's slice — except RCCL's own wire.

def _ranks_in_threads(world, body):
    """body(rank) in one thread per rank; re-raises the first failure."""
    import threading
    errs = [None] * world
    outs = [None] * world

    def run(r):
        try:
            outs[r] = body(r)
        except BaseException as e:   # noqa: BLE001 - reported below
            errs[r] = e

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    for e in errs:
        if e is not None:
            raise e
    return outs


@pytest.mark.parametrize("n,world,k_windows", [(7000, 2, 2), (7000, 3, 5), (1000, 3, 3), (40_000, 2, 4), (300, 4, 4)])
def test_session_with_several_ranks_in_one_process(oracle, n, world, k_windows):
    import ctypes as C
    l = 900
    sp = synth.SynthParams(seed=57 + n + world, minor_permille=(70, 60, 50, 40), partial_rate=0.15)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    rows[: n // 20, 700:703] = (rows[: n // 20, 700:703] + 1) % 4          # haplotypes that span windows (and ranks)
    rows[n // 40: n // 16, 820:823] = (rows[n // 40: n // 16, 820:823] + 2) % 4
    rows[n // 30: n // 12, 100:103] = (rows[n // 30: n // 12, 100:103] + 3) % 4
    full = oracle.call(rows, genes, refseq=ref)
    exp = oracle.phase(rows, full)
    wb = sharding.window_bounds(l, k_windows)
    win_rank = [min(world - 1, k * world // k_windows) for k in range(k_windows)]   # consecutive windows per rank
    slice_begin = sharding.read_slices(n, world)     # n = 300, world = 4: ranks without reads
    slices = [(slice_begin[r], slice_begin[r + 1]) for r in range(world)]
    idbuf = np.frombuffer(np.random.default_rng(n * 7 + world).bytes(128), dtype=np.uint8).copy()
    prm = capi.default_params(n_tests=sharding.default_n_tests(genes))

    def body(rank):
        mine = [k for k in range(k_windows) if win_rank[k] == rank]
        ctxs = []
        for k in mine:
            b, e = wb[k]
            c = capi.Juliet(0)
            c.upload_columns(msa.pack_columns(rows[:, b:e]), n, win_begin=b)
            c.run_async(genes, ref, prm, None, False, 10, False)
            ctxs.append(c)
        comm = C.c_void_p()
        ctxs[0]._chk(ctxs[0].lib.jl_comm_create_inproc(ctxs[0].h, idbuf.ctypes.data_as(C.c_void_p), rank, world, C.byref(comm)))
        xw = capi.Xwin(ctxs, [b for b, _ in wb], [e - b for b, e in wb], win_rank, slice_begin, comm)
        try:
            res = [xw.phase(10) for _ in range(2)]        # a step loop: the same answer every time
        finally:
            xw.close()
            ctxs[0].lib.jl_comm_destroy(comm)
            for c in ctxs:
                c.close()
        return res

    outs = _ranks_in_threads(world, body)
    ids = np.full(n, 0xABCD, dtype=np.uint16)
    for rank, res in enumerate(outs):
        for r in res:
            m = r["merged"]
            assert len(m) == len(full)
            for k in ("gene", "codon_pos", "col", "ref_codon", "codon", "count", "coverage", "expected"):
                assert (m[k] == full[k]).all(), (rank, k)
            assert r["summary"] == exp["summary"], rank
            assert (r["pos_cols"] == exp["pos_cols"]).all() and (r["hap_count"] == exp["hap_count"]).all()
            assert (r["hap_pattern"] == exp["hap_pattern"]).all() and (r["hit"] == exp["hit"]).all() and (r["cooc"] == exp["cooc"]).all()
            b, cnt = r["slice"]
            assert (b, cnt) == (slices[rank][0], slices[rank][1] - slices[rank][0])
            ids[b:b + cnt] = r["read_hap"][:cnt]
    assert (ids == exp["read_hap"]).all()               # every read's id, each from the rank that owns its slice


@pytest.mark.parametrize("n,world", [(6000, 2), (5000, 3), (900, 4)])
def test_two_samples_take_turns_with_several_ranks_in_one_process(oracle, n, world):
    """The strong-scaling loop of bench.py (config3_strong) as it runs at N > 1, with the ranks as threads over the in-process
    transport: TWO samples (different reads of one reference), a window context and a cross-window session each on every
    rank, both sessions on the rank's ONE communicator; sample k + 1's call stage is enqueued before sample k's session is
    run, so its exchanges queue up behind a pileup that is already on the device.  Every step of every rank = the
    unsharded oracle of that step's sample (variants, haplotypes, hit, co-occurrence, every read's id)."""
    import ctypes as C
    l, reps = 600, 6
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    samples, expect = [], []
    for k in range(2):
        sp = synth.SynthParams(seed=900 + 31 * k + n + world, minor_permille=(70, 60, 50, 40), partial_rate=0.1)
        ref = synth.reference(900 + n + world, l)     # one reference, two samples
        rows = synth.rows(sp, l, 0, n, ref)
        rows[: n // 18, 40 + 300 * k:43 + 300 * k] = (rows[: n // 18, 40 + 300 * k:43 + 300 * k] + 1 + k) % 4     # haplotypes that span windows, other ones per sample
        rows[n // 30: n // 11, 520:523] = (rows[n // 30: n // 11, 520:523] + 2) % 4
        full = oracle.call(rows, genes, refseq=ref)
        samples.append(rows)
        expect.append((full, oracle.phase(rows, full)))
    assert not (expect[0][1]["read_hap"] == expect[1][1]["read_hap"]).all()      # (two samples, two answers)
    wb = sharding.window_bounds(l, world)
    slice_begin = sharding.read_slices(n, world)
    idbuf = np.frombuffer(np.random.default_rng(n + world).bytes(128), dtype=np.uint8).copy()
    prm = capi.default_params(n_tests=sharding.default_n_tests(genes))

    def body(rank):
        b, e = wb[rank]
        wins, xws = [], []
        comm = C.c_void_p()
        for k in range(2):
            c = capi.Juliet(0)
            c.upload_columns(msa.pack_columns(samples[k][:, b:e]), n, win_begin=b)
            wins.append(c)
        wins[0]._chk(wins[0].lib.jl_comm_create_inproc(wins[0].h, idbuf.ctypes.data_as(C.c_void_p), rank, world, C.byref(comm)))
        for k in range(2):
            xws.append(capi.Xwin([wins[k]], [x for x, _ in wb], [y - x for x, y in wb], list(range(world)), slice_begin, comm))
        out = []
        try:
            wins[0].run_async(genes, ref, prm, None, False, 10, False)
            for i in range(reps):
                k = i & 1
                if i + 1 < reps:
                    wins[k ^ 1].run_async(genes, ref, prm, None, False, 10, False)     # the next sample's pileup queues up behind this one's
                out.append(xws[k].phase(10))
        finally:
            for x in xws:
                x.close()
            wins[0].lib.jl_comm_destroy(comm)
            for c in wins:
                c.close()
        return out

    outs = _ranks_in_threads(world, body)
    for i in range(reps):
        full, exp = expect[i & 1]
        ids = np.full(n, 0xABCD, dtype=np.uint16)
        for rank in range(world):
            r = outs[rank][i]
            m = r["merged"]
            assert len(m) == len(full), (i, rank)
            for f in ("gene", "codon_pos", "col", "ref_codon", "codon", "count", "coverage", "expected"):
                assert (m[f] == full[f]).all(), (i, rank, f)
            assert r["summary"] == exp["summary"], (i, rank)
            assert (r["hap_count"] == exp["hap_count"]).all() and (r["hap_pattern"] == exp["hap_pattern"]).all()
            assert (r["hit"] == exp["hit"]).all() and (r["cooc"] == exp["cooc"]).all()
            b0, cnt = r["slice"]
            ids[b0:b0 + cnt] = r["read_hap"][:cnt]
        assert (ids == exp["read_hap"]).all(), i


def test_weak_scaling_exchange_with_two_ranks_in_one_process(oracle):
    """jl_allgather_variants (the call path's one collective) between two rank threads on one device: each rank's table
    arrives at both, through the asynchronous batch form and the blocking full-stride form."""
    import ctypes as C
    n, l, world = 6000, 300, 2
    genes = np.array([(1, world * l + 1)], dtype=capi.GENE)
    idbuf = np.frombuffer(np.random.default_rng(99).bytes(128), dtype=np.uint8).copy()
    tabs = [None, None]

    def body(rank):
        sp = synth.SynthParams(seed=300 + rank, minor_permille=(70, 60, 50, 40))
        ref_local = synth.reference(sp.seed, l)
        refseq = np.full(world * l, 4, dtype=np.uint8)
        refseq[rank * l:(rank + 1) * l] = ref_local
        c = capi.Juliet(0)
        c.alloc(n, l, win_begin=rank * l)
        c.synth_fill(sp, ref_local)
        c.sync()
        comm = C.c_void_p()
        c._chk(c.lib.jl_comm_create_inproc(c.h, idbuf.ctypes.data_as(C.c_void_p), rank, world, C.byref(comm)))
        out = []
        try:
            for _ in range(3):
                c.run_async(genes, refseq, capi.default_params(), None, True, 10, False)
                c._chk(c.lib.jl_allgather_variants_async(c.h, comm))
                rows_ = np.zeros(world * 128, dtype=capi.VARIANT)
                counts = np.zeros(world, dtype=np.uint32)
                c._chk(c.lib.jl_allgather_variants(c.h, comm, rows_.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), 128))
                own = c.run_view()["variants"].copy()
                out.append((rows_.copy(), counts.copy(), own))
        finally:
            c.lib.jl_comm_destroy(comm)
            c.close()
        return out

    outs = _ranks_in_threads(world, body)
    for rank in range(world):
        for rows_, counts, own in outs[rank]:
            for peer in range(world):
                theirs = outs[peer][0][2]
                assert counts[peer] == len(theirs) and len(theirs) >= 4
                got = rows_[peer * 128: peer * 128 + counts[peer]]
                for k in ("gene", "codon_pos", "col", "codon", "count", "coverage"):
                    assert (got[k] == theirs[k]).all(), (rank, peer, k)


def test_bound_exchange_with_two_ranks_in_one_process(oracle):
    """jl_group_exchange_bind: every group run carries the all-gather of its windows' table heads, in place in a pinned host
    region the run's kernels write into.  Two rank threads on one device, two windows each, three runs; each rank's collected
    tables = the tables both ranks read from their own result blocks."""
    import ctypes as C
    n, l, world, nw = 5000, 300, 2, 2
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    idbuf = np.frombuffer(np.random.default_rng(123).bytes(128), dtype=np.uint8).copy()

    def body(rank):
        ctxs = []
        for k in range(nw):
            sp = synth.SynthParams(seed=700 + 10 * rank + k, minor_permille=(70, 60, 50, 40))
            c = capi.Juliet(0)
            c.alloc(n + 500 * k, l)
            c.synth_fill(sp, synth.reference(77, l))
            c.sync()
            ctxs.append(c)
        comm = C.c_void_p()
        ctxs[0]._chk(ctxs[0].lib.jl_comm_create_inproc(ctxs[0].h, idbuf.ctypes.data_as(C.c_void_p), rank, world, C.byref(comm)))
        grp = capi.Group(ctxs)
        out = []
        try:
            grp.bind_exchange(comm)
            for _ in range(3):
                grp.run_async(genes, synth.reference(77, l), capi.default_params(), True, 10, True)
                rows_, counts = grp.exchange_collect(world)
                own = [c.run_view()["variants"].copy() for c in ctxs]
                out.append((rows_.copy(), counts.copy(), own))
            grp.bind_exchange(None)
        finally:
            grp.close()
            ctxs[0].lib.jl_comm_destroy(comm)
            for c in ctxs:
                c.close()
        return out

    outs = _ranks_in_threads(world, body)
    for rank in range(world):
        for rows_, counts, _own in outs[rank]:
            for peer in range(world):
                for k in range(nw):
                    theirs = outs[peer][0][2][k]
                    assert counts[k, peer] == len(theirs) and len(theirs) >= 4
                    got = rows_[k, peer, : counts[k, peer]]
                    for f in ("gene", "codon_pos", "col", "codon", "count", "coverage", "expected"):
                        assert (got[f] == theirs[f]).all(), (rank, peer, k, f)
