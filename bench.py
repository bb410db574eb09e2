#!/usr/bin/env python3
"""bench.py — aligned CCS reads/sec through juliet call+phase on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one resident batch of synthetic aligned CCS reads:
  column pileup + per-codon histograms -> Fisher's exact x Bonferroni -> variant table
  (-> the one RCCL all-gather of the variant table when N > 1) -> read x variant phasing -> results on the host.
Batches are independent windows; `--group G` of them (default 8) go through the path in ONE launch per stage
(jl_group_run_async: blockIdx.z = window), so a launch is G steps, and `--inflight` launches are kept in flight.
Every resident batch holds DIFFERENT reads (its own seed); every batch's results (variant table, haplotypes,
per-read ids) land in pinned host memory, are read each step, and one window per launch is compared with what the
same window gave when it was run alone before the timed region (a stale or mixed-up result block fails the run).

Workload at N=1: BASELINE.json configs[2] (= configs[1] with phasing on): 100k CCS reads x 3 kb reference.
N > 1 (`value`): reference windows shard independently — one 3 kb window x 100k reads per rank and batch (weak
scaling), the only exchange is the RCCL all-gather of the variant table, which rides with each launch as one device
operation (jl_group_exchange_bind; JL_BENCH_EXCHANGE=worker: the communicator's worker-thread form).
Every N also times BASELINE.json configs[3] as stated — ONE 10 kb reference whose 1M reads span all windows, split
into N column windows (strong scaling): call per window with the global Bonferroni factor, all-gather of the table,
the variant columns broadcast by their owners, phasing across windows — reported as `config3_strong` in the same line.

Prints ONE JSON line on rank 0 (see the driver contract in the task brief), including
  roofline     the dominant kernel (pileup) against the 8 TB/s HBM peak: algorithmic bytes = reads x columns x 3 / 8 — every
               cell of the resident matrix read exactly once, 3 bits per cell (seven symbols; the bit planes are THE resident
               format since round 4, DESIGN.md "Data layout") — so `frac` is physical: what crosses the HBM over the kernel's
               time (`traffic` from the PMC counters agrees within a few percent).  `step_frac` = the same bytes over the whole
               timed step.  (`frac_in_nibble_units`: the same time against SURVEY 8d's 4-bit cell, for comparison with rounds 1-3.)
  once_through a FRESH window per step: aligned records resident in HBM -> ingest (cigar expansion + plane split, three or four
               launches) -> pileup -> Fisher -> phasing -> results on the host; reads/s and the fraction of the HBM peak in
               record bytes read + plane bytes written + plane bytes read.
  once_through_qv  the same on the input the reference documents (`ccs --richQVs`, doc/JULIET.md:256-259): filtered bases keep
               their letter and carry a low quality, one quality byte per base resident beside the bases, min_qv = 20.
  end_to_end   the kept surface: juliet-synth writes a 100k x 3 kb rich-QV BAM, `juliet --timing -c cfg.json --mode-phasing
               --min-qv 20 in.bam out.json` runs three times as a child process (before this process opens a GPU context), the stage
               laps are parsed and the JSON is compared with the resident batch that holds the same reads; `--e2e-reads N`: a second size.
  cpu_baseline this repo's CPU restatement (oracle/, kind "port": the reference ships no source) on the host cores.

`--gpus N` (N > 1) without a launcher around it: this process starts the N ranks itself (python -m torch.distributed.run ... as a
child) and forwards rank 0's line; whatever happens to the ranks, ONE line goes out (launch_ranks; --run-timeout).
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

# Before anything loads the HIP runtime: by default it multiplexes a process's streams onto 4 hardware queues.  With
# N > 1 a step loop keeps five streams busy (4 launches in flight + the exchange), and the exchange then shares a
# queue with a launch: 0.0307 ms per step against 0.0291 with 8 queues (one-rank emulation); N = 1 is unaffected.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_READS = 100_000
N_COLS = 3000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
C3_READS, C3_COLS = 1_000_000, 10_000   # BASELINE.json configs[3]
C4_READS, C4_COLS = 10_000_000, 9719    # BASELINE.json configs[4]: full-HIV reference, deep coverage
# genes as on doc/img/juliet_target.png (5'LTR ... Protease; p6 / Protease overlap in different frames), continued with
# HIV-like ORFs in all three frames (tests/test_gpu_configs.py checks this layout against the oracle at full size)
HIV_GENES = [(1, 634), (790, 1186), (1186, 1879), (1879, 1921), (1921, 2086), (2086, 2134), (2134, 2292), (2253, 2550),
             (2550, 4230), (4230, 5096), (5041, 5619), (5559, 5850), (6062, 6310), (6225, 8795), (8797, 9417)]


def cpu_baseline(jl, genes, ref, expect, budget_s=12.0, threads=1, rows=None):
    """Oracle (CPU restatement) call+phase on the same reads, bounded to ~budget_s of wall time; also the checker of
    this window's device result (`expect`).
    threads = 1: the plain restatement (no thread option is documented for juliet, so this is the faithful stand-in);
    threads > 1: SURVEY.md §8d's "all host cores": OpenMP over columns / codon positions of a column-major copy of the
    matrix (made before the clock starts, as the GPU's resident matrix is), private counters, no merge; the per-read
    patterns of the phasing stage split over reads; its exact grouping (a sort of the clean reads) stays serial."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from minorseq_amd import msa

    orc = oracle_lib.load()
    if rows is None:
        rows = msa.unpack_columns(jl.download_columns(), jl.n_reads)
    orc.set_threads(threads)
    if threads > 1:
        orc.set_columns(rows)
    reps, t_total = 0, 0.0
    while reps < 1 or (t_total < budget_s and reps < 100):
        t0 = time.perf_counter()
        v = orc.call(rows, genes, refseq=ref)
        orc.pileup(rows)
        ph = orc.phase(rows, v)
        t_total += time.perf_counter() - t0
        reps += 1
    orc.set_threads(1)
    orc.set_columns(None)
    if expect is not None:   # the device's table and haplotypes of this window are the oracle's
        ok = (len(v) == len(expect["count"]) and (v["count"] == expect["count"]).all() and (v["col"] == expect["col"]).all()
              and (ph["hap_count"] == expect["hap_count"]).all() and ph["summary"] == expect["summary"])
        if not ok:
            raise SystemExit("bench.py: the device result of window 0 differs from the oracle's")
    res = {"value": jl.n_reads * reps / t_total, "unit": "reads/s", "cores": threads, "kind": "port",
           "sample": f"{reps} x the full workload ({jl.n_reads} reads x {jl.n_cols} columns), call+phase, "
                     f"oracle/juliet_oracle.c, {threads} thread(s), {t_total:.1f} s"}
    if threads > 1:
        # the two counting sweeps stream the by-column matrix once each (one byte per cell); what is left is the serial
        # grouping sort of the phasing stage and the Fisher tests
        res["host_sweep_GBps"] = 2.0 * jl.n_reads * jl.n_cols * reps / t_total / 1e9
        res["layout"] = "by-column uint8 matrix, OpenMP over columns / codon positions, private counters"
    return res, rows


def usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup's CPU quota (a GPU box hands a one-GPU job
    16 of its cores whatever os.cpu_count() says: 64 threads on them are slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 256))


def signature(out):
    """What a window's results are compared by in the step loop: a few small arrays, not the per-read ids."""
    v, ph = out["variants"], out["phase"]
    return dict(count=v["count"].copy(), col=v["col"].copy(), hap_count=ph["hap_count"].copy(), summary=dict(ph["summary"]))


def same(sig, out):
    v, ph = out["variants"], out["phase"]
    return (len(v) == len(sig["count"]) and (v["count"] == sig["count"]).all() and (v["col"] == sig["col"]).all()
            and len(ph["hap_count"]) == len(sig["hap_count"]) and (ph["hap_count"] == sig["hap_count"]).all()
            and ph["summary"] == sig["summary"])


def subprocess_errors():
    import subprocess
    return subprocess.SubprocessError


def launch_ranks(args, argv):
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as CHILD processes (the driver's
    own form: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...),
    forward rank 0's JSON line, return the children's exit code.  This process never imports torch and never touches a GPU
    (a process that has initialised the GPU must not exec or be replaced: the ranks are fresh children).  Whatever happens
    — a rank that dies, a collective that stalls, no GPU at all — ONE line goes out: the children's, or one that says why
    there is none."""
    import signal
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver supports dmabuf IPC only (RCCL across processes)
    env["JL_BENCH_LAUNCHED"] = "1"
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)   # (its own process group: killed as one)
    lines = []

    def reader():
        for raw in child.stdout:
            lines.append(raw.decode(errors="replace"))
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    deadline = args.run_timeout + 90.0       # the ranks' own watchdog fires first and still writes a line
    reason = None
    try:
        rc = child.wait(timeout=deadline)
    except subprocess.TimeoutExpired:
        reason = f"the ranks did not finish within {deadline:.0f} s; their process group was killed"
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)    # exactly the group started above
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = child.returncode if child.returncode is not None else 124
    th.join(timeout=5)
    line = None
    for ln in lines:
        t = ln.strip()
        if t.startswith("{"):
            try:
                if "metric" in json.loads(t):
                    line = t
            except ValueError:
                pass
    if line is None:
        line = json.dumps({"metric": "aligned CCS reads/sec through juliet call+phase", "value": None, "unit": "reads/s", "n_gpus": args.gpus,
                           "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                           "vs_baseline": None, "data": "synthetic",
                           "error": reason or f"the ranks ended with exit code {rc} and without a result line (their messages are on stderr)"})
        if rc == 0:
            rc = 1
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16000)   # 2000 launches of 8 windows, about 0.4 s timed
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--reads", type=int, default=N_READS)
    ap.add_argument("--cols", type=int, default=N_COLS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the configs[3] strong-scaling measurement")
    ap.add_argument("--no-config4", action="store_true", help="skip the configs[4] strong-scaling measurement (36.4 GB / N per GPU)")
    ap.add_argument("--no-once-through", action="store_true", help="skip the records -> ingest -> call+phase leg")
    ap.add_argument("--once-steps", type=int, default=240, help="timed steps of the once_through leg")
    ap.add_argument("--config3-timeout", type=float, default=180.0, help="N > 1: seconds the configs[3] + configs[4] measurements may take")
    ap.add_argument("--inflight", type=int, default=4,
                    help="launches in flight per GPU (each on its own stream, with its own captured graph)")
    ap.add_argument("--group", type=int, default=8,
                    help="batches (windows) per launch: 1 = one graph per batch (jl_run_async), G > 1 = group runs "
                         "(jl_group_run_async: one pileup / call / phase launch for G windows)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the BAM -> juliet CLI -> JSON leg")
    ap.add_argument("--e2e-reads", type=int, default=0, help="a second end_to_end size (reads x 3 kb), e.g. 1000000: the decode-bound regime")
    ap.add_argument("--run-timeout", type=float, default=900.0,
                    help="N > 1: seconds the whole run may take; then rank 0 writes the line with what was measured and every rank exits 3")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (decided before torch is imported or any GPU call is made)
        sys.exit(launch_ranks(args, sys.argv[1:]))

    # The contract is ONE JSON line on stdout.  Libraries print banners there (RCCL with NCCL_DEBUG=VERSION, gloo's
    # connection notice), so stdout is pointed at stderr for the duration and the line is written to the real one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # the CLI leg first: `juliet` is a process of its own with its own GPU context — before this process opens one
    e2e = e2e_big = None
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_end_to_end and (args.reads, args.cols) == (N_READS, N_COLS)
            and os.environ.get("JL_BENCH_FORCE_DIST") != "1"):
        try:
            e2e = end_to_end(args.reads, args.cols, 1000, 2)
            if args.e2e_reads:
                e2e_big = end_to_end(args.e2e_reads, args.cols, 1000, 2, reps=2)
                e2e_big.pop("_sig", None)
        except (OSError, ValueError, KeyError, subprocess_errors()) as exc:
            e2e = {"error": repr(exc)}

    import torch
    import torch.distributed as dist

    from minorseq_amd import capi, sharding, synth
    if os.environ.get("JL_LIB"):   # tuning aid: an alternative build of the library
        capi.load_library(os.environ["JL_LIB"])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # a launcher that narrows the visible devices per rank (HIP_VISIBLE_DEVICES) leaves one device, index 0
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.stderr.write(f"bench.py: rank {rank}: no GPU visible (the product path has no CPU fallback)\n")
        sys.stderr.flush()
        if world > 1:
            time.sleep(0.5)   # (the peers were started at the same moment: let them say so too before the launcher ends the group)
        raise SystemExit(4)
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    # JL_BENCH_FORCE_DIST=1 drives the N > 1 code path (process group, RCCL bootstrap, all-gather) with one rank
    distributed = world > 1 or os.environ.get("JL_BENCH_FORCE_DIST") == "1"

    # What the line holds so far: filled in as the legs complete.  At N > 1 ONE watchdog covers the whole run, armed before the
    # process group and the communicator are made — the first real ncclAllGather between two devices, the grouped send / recv
    # and the in-place all-gather in pinned host memory have never run on hardware: whatever stalls, rank 0 writes
    # the line with what was measured plus "error", and every rank leaves with a fresh exit (never a re-exec).
    out = {"metric": "aligned CCS reads/sec through juliet call+phase", "value": None, "unit": "reads/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u3 symbols as bit planes / u32 counts / f64 p-values", "data": "synthetic"}
    stage = {"at": "start"}
    left = threading.Lock()

    def leave(reason, code=3):
        if not left.acquire(blocking=False):
            return               # (another timer is already writing the line)
        out["error"] = f"{reason} (rank {rank}, at: {stage['at']})"
        if not args.no_config3:
            out.setdefault("config3_strong", {"error": reason, "scaling": "strong", "n_gpus": world})
            out.setdefault("config4_strong", {"error": reason, "scaling": "strong", "n_gpus": world})
        if rank == 0:
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        sys.stderr.write(f"bench.py: rank {rank}: {out['error']}\n")
        sys.stderr.flush()
        os._exit(code)           # the line is out, and the failure shows in the exit code too

    run_watchdog = threading.Timer(args.run_timeout, leave, (f"no result after --run-timeout {args.run_timeout:.0f} s",))
    run_watchdog.daemon = True
    if distributed:
        run_watchdog.start()
        _LEAVE[0] = leave        # (an exception on this rank: the peers are inside a collective — see the bottom of the file)
        # A launcher that ends the ranks because ONE of them failed sends SIGTERM: rank 0 still writes the line.  The main
        # thread may be inside a C call (a collective that waits for the failed peer) where no Python handler runs, so the
        # signal's number goes to a pipe (set_wakeup_fd: written by the C-level handler at once) that a thread waits on.
        import signal
        rfd, wfd = os.pipe()
        os.set_blocking(wfd, False)
        signal.signal(signal.SIGTERM, lambda *_: None)
        signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)

        def on_term():
            os.read(rfd, 1)
            leave("terminated by the launcher (SIGTERM: a peer rank failed or the job was cancelled)")
        threading.Thread(target=on_term, daemon=True).start()
        stage["at"] = "process group (gloo)"
        # control plane only (id broadcast, barrier, max-reduce of the timing): gloo.  The data-path collective is
        # RCCL through the C ABI (jl_allgather_variants).  A torch NCCL process group would add watchdog threads
        # that measurably slow the single-threaded step loop (0.071 vs 0.046 ms/step at world = 1).
        dist.init_process_group(os.environ.get("JL_BENCH_PG", "gloo"))

    n, l = args.reads, args.cols
    # window `rank` of a world*l reference; one ORF spans everything, so Bonferroni's n is global.
    # Every resident batch has its own reads (seed), so a result that lands in the wrong block cannot go unnoticed.
    ref_local = synth.reference(2 + rank, l)
    win_begin = rank * l
    G = max(1, args.group)
    if G > 32:
        raise SystemExit("bench.py: --group is at most 32 (a group launch carries its windows' argument blocks by value: JL_GROUP_WINDOWS_MAX)")
    n_flight = n_units = max(1, args.inflight)
    genes = np.array([(1, world * l + 1)], dtype=capi.GENE)
    refseq = np.full(world * l, 4, dtype=np.uint8)
    refseq[win_begin:win_begin + l] = ref_local
    prm = capi.default_params()
    ctxs, expected = [], {}
    for k in range(n_units * G):
        c = capi.Juliet(local_rank)
        c.alloc(n, l, win_begin=win_begin)
        c.synth_fill(synth.SynthParams(seed=1000 * (rank + 1) + k), ref_local)
        c.sync()
        ctxs.append(c)
    jl = ctxs[0]
    # launch unit u = contexts [u*G, (u+1)*G): one group object per unit (and, for a step count that is not a
    # multiple of G, one smaller group over the unit's first contexts, made on demand)
    units = [ctxs[u * G:(u + 1) * G] for u in range(n_units)]
    groups = [capi.Group(u) for u in units] if G > 1 else None
    partial_groups = {}
    handle_arrays = {}
    member_group, group_expect = {}, {}   # first member -> {count: group}; (first member, count) -> expected counts per window

    comm = None
    exchange = "RCCL all-gather of the variant table (jl_allgather_variants)"
    if distributed:
        # one RCCL communicator per rank (its own stream); the 128-byte id is made on rank 0 and broadcast.  A rank
        # whose communicator fails ends the run: a bench must not silently time a different exchange.
        idbuf = np.zeros(128, dtype=np.uint8)
        if rank == 0:
            assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
        t = torch.from_numpy(idbuf)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, 0)
        idbuf = t.cpu().numpy()
        comm = C.c_void_p()
        stage["at"] = "jl_comm_create (ncclCommInitRank)"
        jl._chk(jl.lib.jl_comm_create(jl.h, idbuf.ctypes.data_as(C.c_void_p), rank, world, C.byref(comm)))
        # room for the exchanges of one launch's windows, collected by ONE call (jl_allgather_variants_many); 128 rows per
        # rank and window is the stride the compact exchange itself carries (JL_PACK_MAX_VAR)
        XROWS = 128
        all_rows = np.zeros(G * world * XROWS, dtype=capi.VARIANT)
        all_counts = np.zeros(G * world, dtype=np.uint32)
        p_rows, p_counts = all_rows.ctypes.data_as(C.c_void_p), all_counts.ctypes.data_as(C.c_void_p)

    # The exchange of a launch of several windows rides with the launch (jl_group_exchange_bind: the run's kernels write the
    # table heads into the pinned region ONE all-gather works in, in place, on the group's stream); JL_BENCH_EXCHANGE=worker
    # takes the older form (gather kernel + all-gather + copy on the communicator's stream, issued by its worker thread).
    bound = comm is not None and G > 1 and os.environ.get("JL_BENCH_EXCHANGE", "bound") == "bound"
    gpending = {}                        # bound form: exchanges in flight per group
    if bound:
        stage["at"] = "jl_group_exchange_bind (trial all-gather in pinned host memory)"
        for grp_ in groups:
            grp_.bind_exchange(comm)
        exchange = "RCCL all-gather of the table heads, carried by the launch (jl_group_exchange_bind)"
    rccl_ranks = exchange_form = None
    if comm is not None:
        ci = [C.c_int(0) for _ in range(4)]
        jl._chk(jl.lib.jl_comm_info(comm, *[C.byref(x) for x in ci]))
        rccl_ranks = ci[0].value
        exchange_form = ("bound-host: in-place all-gather in pinned host memory" if ci[3].value == 1 else
                         "staged: device region + heads_to_host kernel" if ci[3].value == 0 else
                         "worker: gather kernel + all-gather + copy on the communicator's stream")
    stage["at"] = "set-up runs"
    pending = {id(c): 0 for c in ctxs}   # exchanges enqueued and not yet collected, per context
    state = dict(checked=0, gathered_rows=0)

    host = dict(launch=0.0, drain=0.0, collect=0.0, n=0)   # JL_BENCH_TRACE: host seconds inside the step loop's calls

    def launch(u, count=G):
        """Enqueue `count` steps (batches) of unit u: the whole path, results stored into pinned memory by the
        kernels; the all-gather (N > 1) is the only other device work of a step and is requested right behind it."""
        t_in = time.perf_counter()
        try:
            return launch_(u, count)
        finally:
            host["launch"] += time.perf_counter() - t_in
            host["n"] += 1

    def launch_(u, count=G):
        members = units[u][:count]
        if G == 1:
            members[0].run_async(genes, refseq, prm, None, True, 10, True)
        else:
            if count == G:
                grp = groups[u]
            else:
                grp = partial_groups.get((u, count))
                if grp is None:
                    grp = partial_groups[(u, count)] = capi.Group(members)
                    if bound:
                        grp.bind_exchange(comm)
            member_group.setdefault(id(members[0]), {})[count] = grp
            if (id(members[0]), count) not in group_expect and all(id(c) in expected for c in members):
                group_expect[(id(members[0]), count)] = (np.array([len(expected[id(c)]["count"]) for c in members], dtype=np.uint32),
                                                         np.array([len(expected[id(c)]["hap_count"]) for c in members], dtype=np.uint32))
            t_g = time.perf_counter()
            grp.run_async(genes, refseq, prm, True, 10, True)
            host["run_async"] = host.get("run_async", 0.0) + time.perf_counter() - t_g
        if bound:
            gpending[id(grp)] = gpending.get(id(grp), 0) + 1
        elif comm is not None:
            # the all-gathers of the launch's windows go out as one RCCL group (one collective launch)
            arr = handle_arrays.get((u, count))
            if arr is None:
                arr = handle_arrays[(u, count)] = (C.c_void_p * count)(*[c.h for c in members])
            rc = jl.lib.jl_allgather_variants_async_many(arr, count, comm)
            if rc:
                members[0]._chk(rc)
            for c in members:
                pending[id(c)] += 1
        return members

    def drain(c, k):
        t_in = time.perf_counter()
        try:
            drain_(c, k)
        finally:
            host["drain"] += time.perf_counter() - t_in

    def drain_(c, k):
        for _ in range(k):
            rc = c.lib.jl_allgather_variants(c.h, comm, p_rows, p_counts, XROWS)
            if rc:
                c._chk(rc)
            pending[id(c)] -= 1
            state["gathered_rows"] = int(all_counts[:world].sum())

    def drain_many(members, arr):
        """The oldest pending exchange of every member, one call (they were requested as one batch)."""
        t_in = time.perf_counter()
        rc = jl.lib.jl_allgather_variants_many(arr, len(members), comm, p_rows, p_counts, XROWS)
        if rc:
            for c in members:
                msg = c.lib.jl_last_error(c.h).decode()
                if msg:
                    raise capi.JulietError(rc, msg)
            raise capi.JulietError(rc, "jl_allgather_variants_many failed")
        for c in members:
            pending[id(c)] -= 1
        state["gathered_rows"] = int(all_counts[: len(members) * world].sum())
        host["drain"] += time.perf_counter() - t_in

    def collect(members, final=False, check=True, retire=True):
        last = None
        pick = state["checked"] % len(members)
        grp = member_group.get(id(members[0]), {}).get(len(members)) if G > 1 else None
        if grp is not None:
            # every window's result block in ONE call (jl_group_views: waits for each window's completion word in turn), the
            # counts of all of them compared at once; numpy views are built for the window that is checked and for the last
            exp_nv, exp_h = group_expect[(id(members[0]), len(members))]
            vw = grp.views()
            if not (vw["complete"].all() and (vw["n_variants"] == exp_nv).all() and (vw["n_haplotypes"] == exp_h).all()):
                raise SystemExit(f"bench.py: rank {rank}: a window's result block is incomplete or changed")
            for i in sorted({pick, len(members) - 1}):
                c = members[i]
                out = c.run_view() or c.run_fetch(True, True, cap_var=64)
                if check and i == pick and not same(expected[id(c)], out):
                    raise SystemExit(f"bench.py: rank {rank}: a window's results changed between runs (stale or mixed result block)")
                last = out
        for i, c in enumerate(members if grp is None else ()):
            # results are read in place: the kernels stored them into pinned host memory, completion is a sequence
            # word behind a system-scope fence (jl_run_view_get: counts, read categories, pointers into the block);
            # results too large for that block use the copying fetch.  numpy views are built for the window that is
            # checked and for the last one only (2 us of Python each).
            if i == pick or i == len(members) - 1:
                out = c.run_view() or c.run_fetch(True, True, cap_var=64)
                if check and i == pick and not same(expected[id(c)], out):
                    # one window per launch against what it gave when it ran alone, before the timed region
                    raise SystemExit(f"bench.py: rank {rank}: a window's results changed between runs (stale or mixed result block)")
                last = out
            else:
                rv = c.run_view_raw()
                if not rv.complete or rv.n_variants != len(expected[id(c)]["count"]) or rv.n_haplotypes != len(expected[id(c)]["hap_count"]):
                    raise SystemExit(f"bench.py: rank {rank}: a window's result block is incomplete or changed")
        # the exchange of these contexts' PREVIOUS step is collected now (their own is still crossing xGMI): every
        # step's all-gather is consumed, one cycle late, and its latency never stalls the launching thread.  The windows
        # of a launch were requested as one batch and are collected by one call.
        if bound and grp is not None:
            floor = 0 if final else 1
            while gpending.get(id(grp), 0) > floor:
                t_in = time.perf_counter()
                rc = jl.lib.jl_group_exchange_collect(grp.h, p_rows, p_counts, XROWS)
                if rc:
                    raise capi.JulietError(rc, jl.lib.jl_group_last_error(grp.h).decode())
                gpending[id(grp)] -= 1
                state["gathered_rows"] = int(all_counts[: len(members) * world].sum())
                host["drain"] += time.perf_counter() - t_in
        elif comm is not None:
            floor = 0 if final else 1
            while all(pending[id(c)] > floor for c in members):
                arr = handle_arrays.get(("m", id(members[0]), len(members)))
                if arr is None:
                    arr = handle_arrays[("m", id(members[0]), len(members))] = (C.c_void_p * len(members))(*[c.h for c in members])
                drain_many(members, arr)
            for c in members:   # (members whose exchanges were not requested together)
                if pending[id(c)] > floor:
                    drain(c, pending[id(c)] - floor)
        state["checked"] += 1
        if final:
            # The launch is complete (its completion words were read above); retiring its commands in the runtime now
            # (hipStreamSynchronize on the launch's stream, about 10 us of host time) overlaps with the launches still
            # running.  Left to the closing torch.cuda.synchronize() the same bookkeeping is done stream after stream
            # behind the last result: 10 us per stream that has run since the last fence (tools_tuning/sync_cost.py).
            if retire:
                members[0].sync()
        return last

    trace = [] if os.environ.get("JL_BENCH_TRACE") else None   # tuning aid: host time stamps of the timed steps

    def run_steps(k):
        """k steps; at most n_flight launches (G steps each) in flight, rotating over the n_units launch units; every
        step's results are read on the host."""
        last = None
        inflight = []   # (unit, members) in launch order
        done = 0
        u = 0
        while done < k:
            if len(inflight) == n_flight:
                last = collect(inflight.pop(0)[1])
            # (a step count that is no multiple of G: the short launch goes last.  First — so that its latency-bound stages
            # run beside the full launches' pileups — measured 36.8 against 25.3 us per step over 20 steps: tools_tuning/
            # short_region_order.sh, round 5)
            count = min(G, k - done)
            inflight.append((u, launch(u, count)))
            if trace is not None:
                trace.append(("launched", time.perf_counter()))
            done += count
            u = (u + 1) % n_units
        # the launches still in flight at the end are collected as they COMPLETE (a short launch issued last shares the chip
        # with the full ones in front of it and is done first: collected in launch order, its 14 us of host work — views,
        # check, stream retirement — came behind the last result instead of beside the other launches' kernels)
        newest = inflight[-1][1] if inflight else None
        while inflight:
            i_done = None
            t_poll = time.perf_counter()
            while i_done is None and len(inflight) > 1:      # (polling the completion words: 0.14 us each)
                i_done = next((i for i, (_, m) in enumerate(inflight) if all(c.run_done() for c in m)), None)
                if i_done is None and time.perf_counter() - t_poll > 20.0:
                    break            # a run that never stores its completion word: the in-order collect below waits on its stream and reports
            i_done = i_done or 0
            members = inflight.pop(i_done)[1]
            out = collect(members, final=True, retire=bool(inflight))   # (the closing fence retires the last launch's stream)
            if members is newest:
                last = out
            if trace is not None:
                trace.append(("collected+synced", time.perf_counter()))
        if comm is not None:
            for c in ctxs:
                drain(c, pending[id(c)])
        return last

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, before the warm-up steps: every window alone once (its expected results), then every launch unit (and the
    # smaller groups that step counts which are not a multiple of G will need) once, so that no graph capture or table
    # upload falls into the timed steps
    for c in ctxs:
        c.run_async(genes, refseq, prm, None, True, 10, True)
        expected[id(c)] = signature(c.run_view() or c.run_fetch(True, True, cap_var=64))
    if len({tuple(e["count"]) + tuple(e["hap_count"]) for e in expected.values()}) < max(2, len(ctxs) // 2):
        raise SystemExit("bench.py: the resident windows do not hold different reads")
    for u in range(n_units):
        collect(launch(u, G), final=True)
        for k in (args.warmup, args.steps):
            if G > 1 and k % G:
                collect(launch(u, k % G), final=True)
    if comm is not None:
        for c in ctxs:
            drain(c, pending[id(c)])
    fence()
    stage["at"] = "warm-up steps"
    run_steps(args.warmup)
    fence()
    stage["at"] = "timed steps"
    t0 = time.perf_counter()
    if trace is not None:
        del trace[:]
    last = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if trace is not None:
        sys.stderr.write(f"bench host time per launch (us): group run_async {1e6 * host.get('run_async', 0.0) / max(1, host['n']):.1f}, "
                         f"launch {1e6 * host['launch'] / max(1, host['n']):.1f}, "
                         f"drain {1e6 * host['drain'] / max(1, host['n']):.1f} over {host['n']} launches incl. warm-up\n")
    if trace:
        sys.stderr.write("bench trace (us after t0): " + ", ".join(f"{w} {1e6 * (t - t0):.1f}" for w, t in trace[:12]) +
                         f", fence done {1e6 * elapsed:.1f}\n")
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1000.0 * elapsed / args.steps
    table, ph = last["variants"], last["phase"]
    # (the headline figure is measured: whatever happens from here on, it is in the line)
    out.update({"metric": "aligned CCS reads/sec through juliet call+phase" + (" (weak scaling: independent windows per GPU; strong scaling "
                                                                                  "of one reference: config3_strong, config4_strong)" if world > 1 else ""),
                "value": world * n / (ms_per_step * 1e-3), "ms_per_step": ms_per_step})
    stage["at"] = "latency legs"

    # latency of ONE batch through the path (its own graph, nothing else on the GPU), for the record
    def one_batch():
        jl.run_async(genes, refseq, prm, None, True, 10, True)
        return jl.run_view()
    for _ in range(3):
        one_batch()
    fence()
    t1 = time.perf_counter()
    for _ in range(20):
        one_batch()
    fence()
    latency_ms = 1000.0 * (time.perf_counter() - t1) / 20
    # ... and the same loop inside the library (jl_time_run: launch -> completion word -> run view -> next launch, host clock):
    # what a C or C++ caller sees; the figure above has this interpreter between the runs (4-6 us: ctypes, the numpy views)
    latency_c_ms = jl.time_run(genes, refseq, prm, None, True, 10, True, reps=40)

    # the same for a window with SIXTEEN variant positions (the reference's own screenshots show 9-13 and more,
    # doc/JULIET.md:350, 362): beyond ten positions a pattern takes two key words — the two-word fused launch
    many_ms = many_c_ms = None
    if world == 1 and (n, l) == (N_READS, N_COLS):
        from minorseq_amd import msa
        mp = capi.Juliet(local_rank)
        rows = msa.unpack_columns(jl.download_columns(), n)
        rng = np.random.default_rng(3)
        for k in range(11):                       # eleven more edited codons, about 3 % of the reads each
            who = rng.choice(n, n // 30, replace=False)
            c0 = 3 * (100 + 61 * k)
            rows[who, c0:c0 + 3] = (rows[who, c0:c0 + 3] + 1 + k % 3) % 4
        mp.upload_rows(rows, win_begin=win_begin)
        del rows

        def many():
            mp.run_async(genes, refseq, prm, None, True, 10, True)
            return mp.run_view() or mp.run_fetch(True, True, cap_var=64)
        for _ in range(4):
            mv = many()
        fence()
        t2 = time.perf_counter()
        for _ in range(20):
            mv = many()
        fence()
        many_ms = 1000.0 * (time.perf_counter() - t2) / 20
        many_c_ms = mp.time_run(genes, refseq, prm, None, True, 10, True, reps=40)
        many_positions = int(mv["phase"]["summary"]["n_positions"])
        mp.close()

    # dominant kernel alone: HIP events on the stream it is launched on, around back-to-back launches that rotate
    # over the resident batches (no launch finds its windows in the 256 MiB Infinity Cache).  rocprofv3 sees exactly
    # these launches when bench.py runs with --kernel-only (profiles/README.md).
    if G > 1:
        run_steps(n_units * G)   # the timing hook reads each group's argument table: every group has run
        fence()
        t_pileup_ms, alg_bytes = capi.time_pileup_groups(groups, reps=max(20, args.steps // G))
        # (round 6: the launch the runs make — the pileup with the Fisher stage of its codons in the epilogue; JL_NO_FOLD_CALL=1: the plain one)
        kernel_name = "pileup_planes_group_kernel" if os.environ.get("JL_NO_FOLD_CALL") else "pileup_fold_group_kernel"
    else:
        alg_bytes = n * l * 3.0 / 8.0
        kernel_name = jl.lib.jl_pileup_kernel_name().decode()
        try:
            t_pileup_ms = capi.time_pileup_set(ctxs, reps=max(20, args.steps))
        except capi.JulietError:   # launch shapes that need zeroed counters (long columns): one launch per event pair
            t_pileup_ms = jl.time_pileup(reps=max(10, args.steps))
    achieved = alg_bytes / (t_pileup_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath) and (n, l) == (N_READS, N_COLS):   # the PMC passes were taken on the default workload
        try:
            tj = json.load(open(tpath))
            if tj.get("windows_per_launch", 1) == G:
                traffic = tj.get("pileup_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    step_bytes = n * l * 3.0 / 8.0
    # at N > 1 `value` is WEAK scaling (one independent 100k x 3kb window per GPU and batch); ONE reference split over
    # the GPUs (strong scaling) is config3_strong / config4_strong in the same line
    out.update({
        "config": {"workload": f"configs[2]: {n} CCS reads x {l} bp reference per GPU, pileup + Fisher-exact + phasing "
                               "(96% major + four 1% minor haplotypes, sub 1.75e-4, del 1.3e-3, N 2e-2); every resident "
                               "batch holds different reads",
                   "reads_per_gpu": n, "ref_columns_per_gpu": l,
                   "parallelism": f"window-sharded x{world}, {exchange}" if distributed
                   else "single GPU",
                   "rccl_ranks": rccl_ranks, "exchange_form": exchange_form,
                   "batches_per_launch": G, "launches_in_flight": n_flight, "resident_batches": len(ctxs),
                   "one_batch_latency_ms": latency_ms, "one_batch_latency_c_abi_ms": latency_c_ms,
                   "many_positions_latency_ms": many_ms, "many_positions_latency_c_abi_ms": many_c_ms, "many_positions": many_positions if many_ms is not None else None,
                   "variants_called": state["gathered_rows"] if comm is not None else len(table),
                   "haplotypes": ph["summary"]["n_haplotypes"],
                   "windows_verified_in_loop": state["checked"]},
        "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                       "command (a separate run; counters cannot be read from inside bench.py)" if traffic else None,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": t_pileup_ms,
                     "algorithmic_unit": "3 bits per cell (reads x columns x 3 / 8 bytes): every cell of the resident bit planes read once",
                     "layout": "bit planes, 3 bits per cell: the one resident format, written directly by every producer",
                     # the same time against SURVEY 8d's original 4-bit cell (what rounds 1-3 called `frac`)
                     "frac_in_nibble_units": alg_bytes * (4.0 / 3.0) / (t_pileup_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     # the same bytes over the whole step (launch gaps, Fisher, phasing, results on the host included)
                     # one window alone through the whole path (what `juliet in.bam out.json` does): its bytes over its latency
                     "one_batch_frac": step_bytes / (latency_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "step_achieved": step_bytes / (ms_per_step * 1e-3) / 1e9,
                     "step_frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
    })
    for g in (groups or []) + list(partial_groups.values()):
        g.close()
    if e2e is not None:
        e2e_verify(e2e, expected[id(jl)])
        out["end_to_end"] = e2e
        if e2e_big is not None:
            out["end_to_end_big"] = e2e_big
    if not args.no_once_through and (n, l) == (N_READS, N_COLS):
        try:
            out["once_through"] = once_through(capi, synth, torch, jl, genes, refseq, prm, expected[id(jl)], rank, local_rank, n, l,
                                               win_begin, args.once_steps)
        except (OSError, capi.JulietError) as exc:   # the generator binary is missing, or the leg failed: the main figures stand
            out["once_through"] = {"error": repr(exc)}
        try:
            out["once_through_qv"] = once_through(capi, synth, torch, jl, genes, refseq, prm, expected[id(jl)], rank, local_rank, n, l,
                                                  win_begin, args.once_steps, qv=True)
        except (OSError, capi.JulietError) as exc:
            out["once_through_qv"] = {"error": repr(exc)}
    if not args.no_config3:
        # The weak-scaling measurement above is complete.  The strong-scaling form adds two exchanges that have only ever
        # run with one rank on hardware: should it fail or stall on some rank, every rank leaves after
        # `--config3-timeout` seconds and the line still goes out with what was measured, the failure named in it.
        stage["at"] = "configs[3] / configs[4] strong scaling"
        watchdog = threading.Timer(args.config3_timeout, leave, (f"configs[3]/[4]: no result after {args.config3_timeout} s",))
        watchdog.daemon = True
        if distributed:
            watchdog.start()
        try:
            out["config3_strong"] = config3_strong(capi, sharding, synth, torch, dist, distributed, comm, rank, world, local_rank,
                                                   [c for c in ctxs[1:]])
            if not args.no_config4:
                # configs[4] the same way: 10M reads x 9719 columns (48.6 GB / N per rank) — the workload whose pileup is long
                # enough for the fixed costs of a step (launches, hand-offs, three small exchanges) to leave 1 -> 8 room
                out["config4_strong"] = config3_strong(capi, sharding, synth, torch, dist, distributed, comm, rank, world, local_rank,
                                                       [], reps=6, which=4)
        except Exception as exc:   # noqa: BLE001 — at N > 1 the other ranks are inside a collective: they leave by their timers
            if not distributed:
                raise
            leave(f"rank {rank}: {exc!r}")
        watchdog.cancel()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], rows_host = cpu_baseline(jl, genes, refseq, expected[id(jl)])
        ncores = usable_cores()
        if ncores > 1:
            out["cpu_baseline_all_cores"], _ = cpu_baseline(jl, genes, refseq, None, budget_s=6.0, threads=ncores, rows=rows_host)
    run_watchdog.cancel()
    if rank == 0 and left.acquire(blocking=False):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if comm is not None:
        jl.lib.jl_comm_destroy(comm)
    for c in ctxs:
        c.close()
    if distributed:
        dist.destroy_process_group()


def once_through(capi, synth, torch, jl0, genes, refseq, prm, expect, rank, local_rank, n, l, win_begin, steps, qv=False):
    """A FRESH window per step: aligned records (positions, cigars, BAM's packed bases — what a BAM decoder
    holds, resident in HBM as the brief prescribes for `value`) -> ingest into the bit planes (jl_records_window_async: three or four
    launches) -> pileup -> Fisher -> phasing -> results in pinned host memory, the whole chain enqueued on one stream per
    window, four windows in flight.  The reads are those of resident batch 0 (same seed), so every step's result is compared
    with that batch's; four copies of the records at different addresses and four window matrices rotate, so no step finds
    its records or its planes in the 256 MiB Infinity Cache."""
    # qv (`once_through_qv`): the input the reference documents (`ccs --richQVs`, doc/JULIET.md:256-259, 273-276) — a filtered base
    # keeps its LETTER and carries a low quality, so the cigars hold the true deletions and mismatches only (ten ops a read, not 127)
    # and the records bring one quality byte per base (+300 MB a window); the N's appear when the ingest applies min_qv = 20
    rec = synth.raw_records(1000 * (rank + 1), n, l, ref_seed=2 + rank, extra=("--rich-qv",) if qv else ())
    min_qv = 20 if qv else 0
    K = J = 4
    recs, wins, streams = [], [], []
    for k in range(K):
        c = capi.Juliet(local_rank)
        if qv:
            c.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"], rec["qual"], rec["qual_off"])
        else:
            c.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"])
        recs.append(c)
    for j in range(J):
        st = torch.cuda.Stream()
        streams.append(st)
        wins.append(capi.Juliet(local_rank, stream=st.cuda_stream))
    rec_keys = ("pos", "cigar", "cig_off", "seq4", "seq_off") + (("qual", "qual_off") if qv else ())
    record_bytes = sum(rec[k].nbytes for k in rec_keys)
    parts = {k: int(rec[k].nbytes) for k in rec_keys}
    ops_per_read = len(rec["cigar"]) / max(1, n)
    del rec

    def enqueue(j, s):
        wins[j].records_window(recs[s % K], l, win_begin, min_qv, wait=False)
        wins[j].run_async(genes, refseq, prm, None, True, 10, True)

    def check(j):
        out = wins[j].run_view() or wins[j].run_fetch(True, True, cap_var=64)
        if not same(expect, out):
            raise SystemExit("bench.py: once_through: a fresh window's results differ from the resident batch with the same reads")
        return out

    for rep in range(3):          # every (window, records) pair's first build allocates; graphs are captured on a configuration's 2nd run
        for j in range(J):
            enqueue(j, j + rep)
            check(j)
    torch.cuda.synchronize()
    # every CELL of a freshly built window against resident batch 0 (the same reads, filled by the generator on the device): the
    # per-step check above compares results, which a wrong cell in a column without a variant would pass
    if not (wins[0].download_columns() == jl0.download_columns()).all():
        raise SystemExit("bench.py: once_through: a freshly ingested window differs from the resident batch cell by cell")
    t0 = time.perf_counter()
    busy = [False] * J
    for s in range(steps):
        j = s % J
        if busy[j]:
            check(j)
        enqueue(j, s)
        busy[j] = True
    for j in range(J):
        if busy[(steps + j) % J]:
            check((steps + j) % J)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    # the ingest alone: torch events on the windows' (torch) streams around back-to-back builds, rotating as above
    reps = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(streams[0]):
        e0.record()
    for q in range(reps):
        wins[0].records_window(recs[q % K], l, win_begin, min_qv, wait=False)
    with torch.cuda.stream(streams[0]):
        e1.record()
    e1.synchronize()
    t_ing = e0.elapsed_time(e1) / reps
    plane_bytes = 3 * l * wins[0].plane_stride
    io = record_bytes + 2 * plane_bytes          # records read + planes written (ingest) + planes read (pileup)
    res = {"workload": f"a fresh window per step: records of {n} CCS reads x {l} bp (positions, cigars, 4-bit bases"
                       + (", one quality byte per base: the documented `ccs --richQVs` shape, filtered bases keep their letter, min_qv = 20"
                          if qv else "; filtered bases travel as N letters") + "; resident in HBM) -> "
                       "ingest into the bit planes (three launches for these reads, four when a read may be a long one) -> pileup + Fisher + phasing -> results on the host; 4 windows in flight, "
                       "every step's result verified",
           "min_qv": min_qv, "cigar_ops_per_read": ops_per_read, "record_bytes_by_array": parts,
           "value": n / t, "unit": "reads/s", "ms_per_step": 1000.0 * t, "steps": steps, "cells_verified_before_loop": n * l,
           "record_bytes": record_bytes, "plane_bytes": plane_bytes,
           "bytes_per_step": io, "frac": io / t / 1e9 / HBM_PEAK_GBS,
           "frac_records_plus_planes_once": (record_bytes + plane_bytes) / t / 1e9 / HBM_PEAK_GBS,
           "ingest_ms": t_ing, "ingest_frac": (record_bytes + plane_bytes) / (t_ing * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "ingest_kernels": "cigar_runs_kernel + ingest_planes_kernel, the second forms of both included (torch events on the window's stream, "
                             f"{reps} back-to-back builds rotating over 4 record copies)"}
    for c in wins + recs:
        c.close()
    return res


def end_to_end(n, l, seed, ref_seed, reps=3, min_qv=20, keep=None):
    """The kept surface (SURVEY 8d "end-to-end wall reported separately"; doc/JULIET.md:62-66): `juliet-synth` writes the n x l
    rich-QV BAM (what `ccs --richQVs` + an aligner leave: letters kept, dq / iq / sq tracks) and its target config, then
    `juliet --timing -c cfg.json --mode-phasing --min-qv 20 in.bam out.json` runs `reps` times as a CHILD process — it owns
    its GPU context, so this leg runs before this process opens any (a box allows few processes on a card, and the
    start-up of the HIP runtime is part of what a `juliet` user waits for).  Wall time by this process's clock around the child
    (min of the runs; all of them listed), the CLI's own stage laps of that run, and what the JSON says — compared later with the
    resident batch that holds the same reads.  Measurement only: the floor (runtime start-up, inflate) is outside SURVEY 8."""
    import re
    import subprocess
    import tempfile
    from minorseq_amd import msa
    bindir = os.path.join(ROOT, "minorseq_amd", "bin")
    tmp = tempfile.mkdtemp(prefix="jl_e2e_")
    bam, cfg, outj = os.path.join(tmp, "in.bam"), os.path.join(tmp, "cfg.json"), os.path.join(tmp, "out.json")
    try:
        t0 = time.perf_counter()
        subprocess.check_call([os.path.join(bindir, "juliet-synth"), "--reads", str(n), "--cols", str(l), "--seed", str(seed), "--ref-seed", str(ref_seed),
                               "--rich-qv", "-o", bam, "--config-out", cfg])
        t_gen = time.perf_counter() - t0
        cmd = [os.path.join(bindir, "juliet"), "--timing", "-c", cfg, "--mode-phasing", "--min-qv", str(min_qv), bam, outj]
        runs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
            wall = 1000.0 * (time.perf_counter() - t0)
            if p.returncode != 0:
                return {"error": f"juliet ended with exit code {p.returncode}: {p.stderr.decode(errors='replace')[-400:]}"}
            laps = {}
            for m in re.finditer(r"juliet: timing (\S.*?)\s+([0-9.]+) ms\s+\(at\s+([0-9.]+) ms\)", p.stderr.decode(errors="replace")):
                laps[m.group(1).strip()] = float(m.group(2))
                laps["_end"] = float(m.group(3))
            runs.append((wall, laps))
        wall, laps = min(runs, key=lambda r: r[0])
        j = json.load(open(outj))
        rows = []
        for gi, g in enumerate(j["genes"]):
            for vp in g["variant_positions"]:
                for aa in vp["variant_amino_acids"]:
                    for vc in aa["variant_codons"]:
                        rows.append((gi, vp["ref_position"], msa.codon_index(vc["codon"]), vc["count"]))
        rows.sort()
        hb = j.get("haplotype", {})
        device = sum(laps.get(k, 0.0) for k in ("device ingest", "plan + enqueue", "wait for the run + table", "column counts", "haplotypes + ids"))
        res = {"workload": f"juliet-synth --rich-qv BAM of {n} CCS reads x {l} bp (letters kept, dq / iq / sq tracks) -> `juliet --timing -c cfg.json --mode-phasing "
                           f"--min-qv {min_qv} in.bam out.json` as a child process, {reps} runs; wall = this process's clock around the child, min of the runs",
               "wall_ms": wall, "wall_ms_all": [r[0] for r in runs], "reads_per_s": n / (wall * 1e-3), "bam_bytes": os.path.getsize(bam),
               "in_process_ms": laps.get("_end"),      # the CLI's own clock, main() to the written JSON (the rest of wall_ms: exec, dynamic loading, exit)
               "decode_ms": laps.get("bam decode"), "context_ms": laps.get("context ready"), "upload_ms": laps.get("rest of the upload"),
               "device_ms": device, "emit_ms": laps.get("json / html"), "generate_bam_s": t_gen,
               "_sig": {"counts": [r[3] for r in rows], "hap_reads": [h["reads"] for h in hb.get("haplotypes", [])],
                        "reported_reads": hb.get("reported_reads"), "damaged_reads": hb.get("damaged_reads")}}
        return res
    finally:
        if keep is None:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)


def e2e_verify(e2e, sig):
    """The CLI's JSON against the resident batch that holds the same reads (filled on the device by the generator's cell
    function; the CLI's came through BAM decode + the QV ingest): variant counts in table order, haplotype read counts, categories."""
    if not e2e or "_sig" not in e2e:
        return
    g = e2e.pop("_sig")
    s = sig["summary"]
    ok = (g["counts"] == [int(x) for x in sig["count"]] and g["hap_reads"] == [int(x) for x in sig["hap_count"]]
          and g["reported_reads"] == s["reported_reads"] and g["damaged_reads"] == s["damaged_reads"])
    if not ok:
        raise SystemExit("bench.py: end_to_end: the CLI's JSON differs from the resident batch with the same reads")
    e2e["verified"] = f"{len(g['counts'])} variant rows and {len(g['hap_reads'])} haplotypes' read counts = the resident batch with the same reads"


def config3_strong(capi, sharding, synth, torch, dist, distributed, comm, rank, world, local_rank, free_ctxs, reps=12,
                   which=3):
    """BASELINE.json configs[3] as stated: 1M CCS reads x ONE 10 kb reference, split into `world` column windows (rank
    r holds the 1M reads' columns of window r: 5 GB / world).  `which` = 4: configs[4] the same way — 10M reads x the
    9719-column full-HIV reference with fifteen ORFs in three frames (48.6 GB / world per rank).  A step = call per window with the GLOBAL Bonferroni
    factor (jl_run_async, phasing off) and ONE call of the C ABI for the rest (jl_xwin_phase_sharded): all-gather of the
    variant table (RCCL) -> merge + plan -> one packed send per peer of the variant columns' read slices (RCCL; SURVEY 8e
    option A) -> every rank groups its 1M/world reads -> all-gather of the group tables (RCCL) -> merge + selection on the
    merged counts (C++) -> per-read ids of the rank's slice.  No Python, numpy or pickle between the stages; every
    hand-off to the host is a word in pinned memory.  `value` = 1M / t.
    The per-read ids stay on the device inside the loop (fetched once at the end): at 1e6 reads expanding them on the
    host would be most of a step."""
    # The weak-scaling batches stay allocated (4.8 + 5 GB of 288): device memory that was freed and is allocated again
    # runs slower on this driver — the pileup by 4 %, the phasing's random-access tables 3x (tools_tuning/
    # config3_stages.py runs both generations) — and a job that runs configs[3] alone never sees second-hand memory.
    if which == 4:
        n, l = C4_READS, C4_COLS
        sp = synth.SynthParams(seed=5)
        genes = np.array(HIV_GENES, dtype=capi.GENE)
    else:
        n, l = C3_READS, C3_COLS
        sp = synth.SynthParams(seed=4)
        genes = np.array([(1, 3 * (l // 3) + 1)], dtype=capi.GENE)
    ref = synth.reference(sp.seed, l)
    prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
    wb = sharding.window_bounds(l, world)
    b, e = wb[rank]
    # the reads are sharded for phasing (SURVEY 8e option A): rank r groups reads [sb[r], sb[r+1])
    sb = sharding.read_slices(n, world)
    # TWO samples (different reads of the same reference) with a session each, taking turns: sample k + 1's counting
    # launch is enqueued before sample k's table is waited for, so the latency-bound rest of a step (Fisher stage, the
    # exchanges, grouping, merge + selection on the host, ids) runs beside the next sample's pileup instead of behind an
    # idle device — the weak-scaling loop's "launches in flight" for the strong-scaling form.  Every step still takes one
    # whole sample through the whole path; the device never waits for the host between two pileups.
    wins, xws = [], []
    for k in range(2):
        w = capi.Juliet(local_rank)
        w.alloc(n, e - b, win_begin=b)
        w.synth_fill_window(synth.SynthParams(seed=sp.seed + 100 * k), ref)
        w.sync()
        wins.append(w)
        xws.append(capi.Xwin([w], [x for x, _ in wb], [y - x for x, y in wb], list(range(world)), sb, comm if world > 1 else None))
    win, xw = wins[0], xws[0]
    for w in wins:
        w.run_pileup_clock(True)     # two clock nodes around the pileup of every run: the pileup as it runs IN the loop

    def enqueue(k):
        wins[k].run_async(genes, ref, prm, None, False, 10, False)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    firsts = []
    for k in (0, 1, 0, 1):      # warm-up, and each sample's result for the checks in the loop
        enqueue(k)
        r = xws[k].phase_raw(10)
        firsts.append((r.n_variants, r.n_haplotypes, r.summary.reported_reads, r.summary.damaged_reads))
    assert firsts[0] == firsts[2] and firsts[1] == firsts[3]
    fence()
    t0 = time.perf_counter()
    per_step, in_loop = [], []
    enqueue(0)
    for i in range(reps):
        t_s = time.perf_counter()
        k = i & 1
        if i + 1 < reps:
            enqueue(k ^ 1)          # the next sample's pileup queues up behind this one's
        r = xws[k].phase_raw(10)    # returns once this sample's ids are enqueued
        if (r.n_variants, r.n_haplotypes, r.summary.reported_reads, r.summary.damaged_reads) != firsts[k]:
            raise SystemExit(f"bench.py: configs[{which}]: a step's result changed between runs")
        per_step.append(time.perf_counter() - t_s)
        in_loop.append(wins[k].run_pileup_interval())      # (this sample's run is over: its table was waited for)
    fence()
    t = (time.perf_counter() - t0) / reps
    t_median = sorted(per_step)[len(per_step) // 2]
    if distributed:
        tt = torch.tensor([t], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    # one sample alone, the device idle before and after (what a step cost before the samples took turns)
    solo, solo_pileup = [], []
    for _ in range(5):
        fence()
        t_s = time.perf_counter()
        enqueue(0)
        xws[0].phase_raw(10)
        torch.cuda.synchronize()
        solo.append(time.perf_counter() - t_s)
        solo_pileup.append(wins[0].run_pileup_ms())
    t_solo = sorted(solo)[len(solo) // 2]
    k_solo = sorted(solo_pileup)[len(solo_pileup) // 2]
    # the pileups of the loop on the device's time line: how long each took where it ran (two samples' pileups overlap, each
    # slower for it), and how long the device was WITHOUT any pileup between the first one's begin and the last one's end
    durs = [e_ - b_ for b_, e_ in in_loop]
    k_loop = sum(durs) / len(durs)
    k_loop_median = sorted(durs)[len(durs) // 2]
    iv = sorted(in_loop)
    covered, cur_b, cur_e = 0.0, iv[0][0], iv[0][1]
    for b_, e_ in iv[1:]:
        if b_ > cur_e:
            covered += cur_e - cur_b
            cur_b, cur_e = b_, e_
        else:
            cur_e = max(cur_e, e_)
    covered += cur_e - cur_b
    span = iv[-1][1] - iv[0][0]
    idle_per_step = (span - covered) / len(iv)          # device time without a pileup, per step
    busy_per_step = covered / len(iv)                   # device time with at least one pileup running, per step
    res = xw.phase(10, want_reads=True)
    s = res["summary"]
    assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == n      # doc/JULIET.md:378-379
    ids = res["read_hap"]
    assert len(ids) == sb[rank + 1] - sb[rank]
    if world == 1:   # every read's id is consistent with the categories
        assert int((ids == capi.HAP_DAMAGED).sum()) == s["damaged_reads"] and int((ids < s["n_haplotypes"]).sum()) == s["reported_reads"]
    t_k = win.time_pileup(reps=5)
    out = {"workload": f"configs[{which}]: {n} CCS reads x {l} bp reference ({len(genes)} ORF(s)) split into {world} column window(s), call per window + "
                       "jl_xwin_phase_sharded (all-gather of the table, packed column-slice exchange, grouping per read slice, "
                       "all-gather + C++ merge of the group tables, ids: SURVEY 8e option A); two samples take turns, the next "
                       "sample's pileup enqueued before this one's table is waited for",
           "value": n / t, "unit": "reads/s", "ms_per_step": 1000.0 * t, "ms_per_step_median": 1000.0 * t_median,
           "ms_per_step_one_sample_alone": 1000.0 * t_solo,
           "scaling": "strong", "n_gpus": world,
           "columns_per_gpu": int(e - b), "variants_called": int(len(res["merged"])), "variant_positions": int(s["n_positions"]),
           "haplotypes": int(s["n_haplotypes"]),
           "exchanges": "none (one window)" if world == 1 else "1 ncclAllGather of the variant tables + 1 group of packed ncclSend/ncclRecv (slice r of the owned columns to rank r, one message per peer) + 1 ncclAllGather of the group tables",
           # (3 bits per cell, as roofline.frac)
           "pileup_kernel_ms": t_k, "pileup_frac_of_hbm_peak": (n * (e - b) * 0.375) / (t_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
           # what does NOT shrink like the pileup when the columns are split over more GPUs (the Amdahl term of 1 -> 8 scaling):
           # device clock nodes around the pileup launch of every run (jl_run_pileup_ms) put the loop's pileups on one time line.
           # pileup_in_loop_ms: how long one took where it ran — the two samples' pileups OVERLAP (separate streams), each slower
           # for it, which is why subtracting the isolated kernel gave a negative residue in round 4.  exposed_ms = the device time
           # per step during which NO pileup was running (between the loop's first begin and last end): what the latency-bound
           # stages do not hide.  exposed_ms_by_host_clock = the host's step minus the device time with a pileup running per step
           # (the same thing seen from the host; nothing is clamped).  alone = one sample by itself.
           "pileup_in_loop_ms": k_loop, "pileup_in_loop_ms_median": k_loop_median,
           "pileup_busy_ms_per_step": busy_per_step, "exposed_ms": idle_per_step,
           "exposed_ms_by_host_clock": 1000.0 * t - busy_per_step,
           "pileup_in_run_ms_one_sample_alone": k_solo, "exposed_ms_one_sample_alone": 1000.0 * t_solo - k_solo}
    for x in xws:
        x.close()
    for w in wins:
        w.close()
    return out



_LEAVE = [None]   # main()'s leave(reason), once a whole-run watchdog exists (N > 1)

if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:   # noqa: BLE001
        # At N > 1 the other ranks are (or will be) inside a collective with this one: say what happened in the line (rank 0) and
        # leave with a fresh exit; the peers leave by the launcher's SIGTERM or by their watchdogs.
        if _LEAVE[0] is None:
            raise
        import traceback
        traceback.print_exc()
        _LEAVE[0](f"{exc!r}")
