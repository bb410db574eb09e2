"""`python3 bench.py --gpus N` as a plain command (no torch.distributed.run around it) starts its own ranks as child
processes and ALWAYS ends with one JSON line: here without a GPU — both ranks say so, the launcher
reports it, nothing hangs, nothing is left running; on the GPU box with two ranks on the ONE device RCCL refuses, which is the
failure path of a real multi-GPU node (a rank that cannot make its communicator) exercised on hardware."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout, env_extra=None):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    t0 = time.time()
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, 9)
        raise
    return p.returncode, out.decode(), err.decode(), time.time() - t0


def _json_lines(out):
    return [json.loads(t) for t in out.splitlines() if t.strip().startswith("{")]


def _no_child_left(pgid):
    """no process of the launcher's group is alive any more (the group was the launcher's own session)"""
    try:
        os.killpg(pgid, 0)
        return False
    except ProcessLookupError:
        return True
    except PermissionError:
        return False


def test_launcher_without_a_gpu_reports_and_returns():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the no-GPU path is the CPU container's")
    rc, out, err, dt = _run(["--gpus", "2", "--steps", "20", "--warmup", "5"], timeout=120)
    assert rc != 0
    assert dt < 90, dt
    for r in (0, 1):
        assert f"rank {r}: no GPU visible" in err, err[-2000:]
    lines = _json_lines(out)
    assert len(lines) == 1, out
    line = lines[0]
    assert line["value"] is None and line["n_gpus"] == 2 and "error" in line
    assert line["steps"] == 20 and line["warmup"] == 5 and line["unit"] == "reads/s"


def test_one_rank_needs_no_launcher_and_fails_loudly_without_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    rc, out, err, dt = _run(["--steps", "20", "--warmup", "5"], timeout=120)
    assert rc != 0 and "no GPU visible" in err and not _json_lines(out)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_end_with_an_error_line():
    """Two ranks, one device: ncclCommInitRank refuses (or the run dies some other way) — either way ONE line comes out, it names
    the failure, the exit code is not 0 and the ranks are gone within the bound."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    rc, out, err, dt = _run(["--gpus", "2", "--steps", "16", "--warmup", "8", "--no-config3", "--no-once-through", "--no-cpu-baseline",
                             "--run-timeout", "150"], timeout=400)
    lines = _json_lines(out)
    assert len(lines) == 1, (out, err[-3000:])
    line = lines[0]
    assert line["n_gpus"] == 2
    if rc == 0:      # (an RCCL that accepts two ranks on a device: then the run must be a complete one)
        assert line["value"] and line["config"]["rccl_ranks"] == 2
    else:
        assert "error" in line, line
    assert dt < 380


@pytest.mark.gpu
def test_whole_run_watchdog_writes_the_line_and_exits_3():
    """One rank driving the N > 1 path (JL_BENCH_FORCE_DIST=1) with a --run-timeout that expires inside the later legs: the line
    holds what was measured so far (the headline figure) plus "error", the exit code is 3."""
    rc, out, err, dt = _run(["--steps", "64", "--warmup", "16", "--no-cpu-baseline", "--run-timeout", "2"], timeout=300,
                            env_extra={"JL_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29671", "RANK": "0",
                                       "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    lines = _json_lines(out)
    assert len(lines) == 1, (out, err[-3000:])
    line = lines[0]
    if rc == 0:
        pytest.skip(f"the whole run took less than the time-out ({dt:.1f} s)")
    assert rc == 3 and "error" in line and "run-timeout" in line["error"], line
    assert "config3_strong" in line
