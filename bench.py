#!/usr/bin/env python3
"""bench.py — aligned CCS reads/sec through juliet call+phase on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one resident batch of synthetic aligned CCS reads:
  column pileup + per-codon histograms -> Fisher's exact x Bonferroni -> variant table
  (-> the one RCCL all-gather of the variant table when N > 1) -> read x variant phasing -> results on the host.
Batches are independent windows; `--group G` of them (default 8) go through the path in ONE launch per stage
(jl_group_run_async: blockIdx.z = window), so a launch is G steps, and `--inflight` launches are kept in flight.
Every batch's results (variant table, haplotypes, per-read ids) land in pinned host memory and are read each step.
Workload at N=1: BASELINE.json configs[2] (= configs[1] with phasing on): 100k CCS reads x 3 kb reference.
N > 1: reference windows shard independently (one 3 kb window x 100k reads per rank, weak scaling), the only
exchange is the all-gather of the fixed-stride variant table.

Prints ONE JSON line on rank 0 (see the driver contract in the task brief), including
  roofline     the dominant kernel (pileup) against the 8 TB/s HBM peak: algorithmic bytes = reads x columns / 2
  cpu_baseline this repo's CPU restatement (oracle/, kind "port": the reference ships no source) on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
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


def cpu_baseline(jl, genes, ref, budget_s=12.0, threads=1, rows=None):
    """Oracle (CPU restatement) call+phase on the same reads, bounded to ~budget_s of wall time.
    threads = 1: the plain restatement (no thread option is documented for juliet, so this is the faithful stand-in);
    threads > 1: its two counting sweeps split over reads with OpenMP (SURVEY.md §8d "all host cores")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from minorseq_amd import msa

    orc = oracle_lib.load()
    if rows is None:
        rows = msa.unpack_columns(jl.download_columns(), jl.n_reads)
    orc.set_threads(threads)
    reps, t_total = 0, 0.0
    while reps < 1 or (t_total < budget_s and reps < 100):
        t0 = time.perf_counter()
        v = orc.call(rows, genes, refseq=ref)
        orc.pileup(rows)
        orc.phase(rows, v)
        t_total += time.perf_counter() - t0
        reps += 1
    orc.set_threads(1)
    return {"value": jl.n_reads * reps / t_total, "unit": "reads/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x the full workload ({jl.n_reads} reads x {jl.n_cols} columns), call+phase, "
                      f"oracle/juliet_oracle.c, {threads} thread(s), {t_total:.1f} s"}, rows


def main():
    # The contract is ONE JSON line on stdout.  Libraries print banners there (RCCL with NCCL_DEBUG=VERSION, gloo's
    # connection notice), so stdout is pointed at stderr for the duration and the line is written to the real one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16000)   # 2000 launches of 8 windows, about 0.4 s timed (fill and drain of the 4-deep pipeline and the clocks' ramp are inside the timed region)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--reads", type=int, default=N_READS)
    ap.add_argument("--cols", type=int, default=N_COLS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=4,
                    help="launches in flight per GPU (each on its own stream, with its own captured graph)")
    ap.add_argument("--group", type=int, default=8,
                    help="batches (windows) per launch: 1 = one graph per batch (jl_run_async), G > 1 = group runs "
                         "(jl_group_run_async: one pileup / call / phase launch for G windows)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from minorseq_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # a launcher that narrows the visible devices per rank (HIP_VISIBLE_DEVICES) leaves one device, index 0
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no GPU visible (the product path has no CPU fallback)")
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    # JL_BENCH_FORCE_DIST=1 drives the N > 1 code path (process group, RCCL bootstrap, all-gather) with one rank
    distributed = world > 1 or os.environ.get("JL_BENCH_FORCE_DIST") == "1"
    if distributed:
        # control plane only (id broadcast, barrier, max-reduce of the timing): gloo.  The data-path collective is
        # RCCL through the C ABI (jl_allgather_variants).  A torch NCCL process group would add watchdog threads
        # that measurably slow the single-threaded step loop (0.071 vs 0.046 ms/step at world = 1).
        dist.init_process_group(os.environ.get("JL_BENCH_PG", "gloo"))

    n, l = args.reads, args.cols
    # window `rank` of a world*l reference; one ORF spans everything, so Bonferroni's n is global.
    # `inflight` contexts hold one resident batch each (same workload); steps alternate between them so that
    # one batch's latency-bound tail (Fisher, phasing, result copy) overlaps the next batch's pileup stream.
    sp = synth.SynthParams(seed=2 + rank)
    ref_local = synth.reference(sp.seed, l)
    win_begin = rank * l
    G = max(1, args.group)
    if G > 32:
        raise SystemExit("bench.py: --group is at most 32 (a group launch carries its windows' argument blocks by value: JL_GROUP_WINDOWS_MAX)")
    n_units = max(1, args.inflight)
    ctxs = []
    for _ in range(n_units * G):
        c = capi.Juliet(local_rank)
        c.alloc(n, l, win_begin=win_begin)
        c.synth_fill(sp, ref_local)
        c.sync()
        ctxs.append(c)
    jl = ctxs[0]
    # launch unit u = contexts [u*G, (u+1)*G): one group object per unit (and, for a step count that is not a
    # multiple of G, one smaller group over the unit's first contexts, made on demand)
    units = [ctxs[u * G:(u + 1) * G] for u in range(n_units)]
    groups = [capi.Group(u) for u in units] if G > 1 else None
    partial_groups = {}
    handle_arrays = {}
    genes = np.array([(1, world * l + 1)], dtype=capi.GENE)
    refseq = np.full(world * l, 4, dtype=np.uint8)
    refseq[win_begin:win_begin + l] = ref_local
    prm = capi.default_params()

    comm = None
    exchange = "RCCL all-gather of the variant table (jl_allgather_variants)"
    if distributed and os.environ.get('JL_BENCH_NO_COMM') != '1':
        # one RCCL communicator per rank (its own stream); the 128-byte id is made on rank 0 and broadcast
        idbuf = np.zeros(128, dtype=np.uint8)
        if rank == 0:
            assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
        t = torch.from_numpy(idbuf)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, 0)
        idbuf = t.cpu().numpy()
        comm = C.c_void_p()
        try:
            jl._chk(jl.lib.jl_comm_create(jl.h, idbuf.ctypes.data_as(C.c_void_p), rank, world, C.byref(comm)))
            ok = 1
        except capi.JulietError as e:   # keep the run alive and say so: the exchange then goes through the process group
            print(f"[bench] rank {rank}: RCCL communicator failed ({e}); falling back to a torch.distributed all-gather",
                  file=sys.stderr, flush=True)
            comm, ok = None, 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                jl.lib.jl_comm_destroy(comm)
            comm = None
            exchange = "torch.distributed all_gather (RCCL communicator unavailable)"
        all_rows = np.zeros(world * capi.VARIANT_CAP, dtype=capi.VARIANT)
        all_counts = np.zeros(world, dtype=np.uint32)
        p_rows, p_counts = all_rows.ctypes.data_as(C.c_void_p), all_counts.ctypes.data_as(C.c_void_p)

    pending = {id(c): 0 for c in ctxs}   # exchanges enqueued and not yet collected, per context

    def launch(u, count=G):
        """Enqueue `count` steps (batches) of unit u: the whole path, results stored into pinned memory by the
        kernels; the all-gather (N > 1) is the only other device work of a step and is requested right behind it."""
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
            grp.run_async(genes, refseq, prm, True, 10, True)
        if comm is not None:
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
        for _ in range(k):
            rc = c.lib.jl_allgather_variants(c.h, comm, p_rows, p_counts, capi.VARIANT_CAP)
            if rc:
                c._chk(rc)
            pending[id(c)] -= 1

    def collect(members, final=False):
        last = None
        for c in members:
            # results are read in place: the kernels stored them into pinned host memory, completion is a sequence
            # word behind a system-scope fence (jl_run_view_get); results too large for that block use the copying fetch
            out = c.run_view() or c.run_fetch(True, True, cap_var=64)
            # the exchange of this context's PREVIOUS step is collected now (its own is still crossing xGMI): every
            # step's all-gather is consumed, one cycle late, and its latency never stalls the launching thread
            if comm is not None and pending[id(c)] > (0 if final else 1):
                drain(c, 1)
            if distributed and comm is None and os.environ.get('JL_BENCH_NO_COMM') != '1':
                from minorseq_amd import sharding
                tabs = sharding.allgather_tables(out["variants"])
                all_counts[:] = [len(t) for t in tabs]
            last = (out["variants"], out["phase"])
        return last

    def run_steps(k):
        """k steps; at most n_units launches (G steps each) in flight; every step's results are read on the host."""
        last = None
        inflight = []   # (unit, members) in launch order
        done = 0
        u = 0
        dbg = os.environ.get("JL_BENCH_DEBUG")
        tl = tc = 0
        while done < k:
            if len(inflight) == n_units:
                t_ = time.perf_counter_ns()
                last = collect(inflight.pop(0)[1])
                tc += time.perf_counter_ns() - t_
            count = min(G, k - done)
            t_ = time.perf_counter_ns()
            inflight.append((u, launch(u, count)))
            tl += time.perf_counter_ns() - t_
            done += count
            u = (u + 1) % n_units
        if dbg:
            print(f"[bench debug] {k} steps: launch {tl / 1e3:.0f} us, collect {tc / 1e3:.0f} us", file=sys.stderr)
        while inflight:
            last = collect(inflight.pop(0)[1], final=True)
        if comm is not None:
            for c in ctxs:
                drain(c, pending[id(c)])
        return last

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, before the warm-up steps: every launch unit (and the smaller groups that step counts which are not a
    # multiple of G will need) runs once, so that no graph capture or table upload falls into the timed steps
    for u in range(n_units):
        collect(launch(u, G), final=True)
        for k in (args.warmup, args.steps):
            if G > 1 and k % G:
                collect(launch(u, k % G), final=True)
    if comm is not None:
        for c in ctxs:
            drain(c, pending[id(c)])
    fence()
    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    table, ph = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1000.0 * elapsed / args.steps

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

    # dominant kernel alone: HIP events on the stream it is launched on, around back-to-back launches that rotate
    # over the resident batches (no launch finds its windows in the 256 MiB Infinity Cache)
    if G > 1:
        run_steps(n_units * G)   # the timing hook reads each group's argument table: every group has run
        fence()
        t_pileup_ms, alg_bytes = capi.time_pileup_groups(groups, reps=max(20, args.steps // G))
        kernel_name = "pileup_group_kernel"
    else:
        alg_bytes = n * l / 2.0
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

    n_var = int(all_counts.sum()) if distributed and os.environ.get('JL_BENCH_NO_COMM') != '1' else len(table)
    out = {
        "metric": "aligned CCS reads/sec through juliet call+phase",
        "value": world * n / (ms_per_step * 1e-3),
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u4 symbols / u32 counts / f64 p-values",
        "data": "synthetic",
        "config": {"workload": f"configs[2]: {n} CCS reads x {l} bp reference per GPU, pileup + Fisher-exact + phasing "
                               "(96% major + four 1% minor haplotypes, sub 1.75e-4, del 1.3e-3, N 2e-2)",
                   "reads_per_gpu": n, "ref_columns_per_gpu": l,
                   "parallelism": f"window-sharded x{world}, {exchange}" if distributed
                   else "single GPU",
                   "batches_per_launch": G, "launches_in_flight": n_units, "resident_batches": len(ctxs),
                   "one_batch_latency_ms": latency_ms,
                   "variants_called": n_var, "haplotypes": ph["summary"]["n_haplotypes"]},
        "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": t_pileup_ms},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], rows_host = cpu_baseline(jl, genes, refseq)
        ncores = min(os.cpu_count() or 1, 64)
        if ncores > 1:
            out["cpu_baseline_all_cores"], _ = cpu_baseline(jl, genes, refseq, budget_s=6.0, threads=ncores, rows=rows_host)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if comm is not None:
        jl.lib.jl_comm_destroy(comm)
    for g in (groups or []) + list(partial_groups.values()):
        g.close()
    for c in ctxs:
        c.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
