"""`python3 bench.py --gpus N` must work WITHOUT a launcher: the GPU-free parent starts the N rank
processes itself, rank 0 prints ONE JSON line, `ranks_seen` is what a SUM all-reduce of 1.0 per rank returned.

CPU part (here): SMX_BENCH_DRY_RUN=1 replaces the GPU step by nothing -- launcher, rendezvous, barriers,
max-over-ranks and the one-line contract run for real over gloo.  GPU part: the real HIP step under two ranks
sharing cuda:0 over gloo (the one-GPU rehearsal of the RCCL run the driver makes on 8 GPUs).
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    p = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    return p, lines


@pytest.mark.timeout(300)
def test_self_launch_two_ranks_dry_run():
    p, lines = _run(["--gpus", "2", "--steps", "4", "--warmup", "1"],
                    {"SMX_BENCH_DRY_RUN": "1", "SMX_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, lines                      # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["steps"] == 4 and out["warmup"] == 1
    assert out["collective_backend"] == "gloo"
    assert out["launched_by"] == "bench.py launch_ranks"
    assert out["dry_run"] is True and out["value"] is None
    assert out["scaling"] == "weak" and out["higher_is_better"] is True


@pytest.mark.timeout(300)
def test_launcher_started_ranks_still_work_dry_run():
    """The driver's way: python -m torch.distributed.run ... bench.py --gpus 2."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"SMX_BENCH_DRY_RUN": "1", "SMX_BENCH_BACKEND": "gloo"})
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH,
                        "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["launched_by"] == "external launcher"


@pytest.mark.timeout(120)
def test_failing_rank_gives_nonzero_exit_and_no_line():
    """A rank that dies takes the run down with a non-zero status (the others are not left hanging)."""
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", "eager"],
                    {"SMX_BENCH_DRY_RUN": "1", "SMX_BENCH_BACKEND": "gloo", "SMX_BENCH_TEST_FAIL_RANK": "1"},
                    timeout=110)
    assert p.returncode != 0
    assert not [l for l in lines if l.startswith("{")]


def test_world_size_mismatch_is_refused():
    env = {"WORLD_SIZE": "2", "RANK": "0", "SMX_BENCH_CHILD": "1", "SMX_BENCH_DRY_RUN": "1"}
    p, _ = _run(["--gpus", "4"], env, timeout=60)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_self_launch_two_ranks_real_step_on_one_gpu(gpu):
    """Exactly the command of VERDICT r2 #1: no launcher, two ranks, the real HIP step (both ranks on cuda:0,
    gloo instead of RCCL because one device cannot host two RCCL ranks)."""
    p, lines = _run(["--gpus", "2", "--steps", "4", "--warmup", "1"],
                    {"SMX_BENCH_ONE_DEVICE": "1", "SMX_BENCH_BACKEND": "gloo"}, timeout=850)
    assert p.returncode == 0, p.stderr[-3000:]
    js = [l for l in lines if l.startswith("{")]
    assert len(js) == 1, lines
    out = json.loads(js[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 4
    assert out["config"]["grad_sync"] in ("overlap", "fused")
    assert out["config"]["global_batch"] == 128
    assert out["value"] > 1.0 and out["roofline"]["frac"] > 0.05
    assert "cpu_baseline" not in out and "other_configs" not in out          # N = 1 only


@pytest.mark.timeout(120)
def test_attempt_deadline_kills_a_rank_that_never_arrives():
    """one attempt is bounded (SMX_BENCH_ATTEMPT_S, default 240 s -- graph attempt + eager repeat fit
    the driver's 600 s), a rank that hangs before the rendezvous is killed with its session, the parent says which
    limit was hit and shows every rank's last stderr lines, exit status non-zero, no JSON line."""
    import time
    t0 = time.monotonic()
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"],
                    {"SMX_BENCH_DRY_RUN": "1", "SMX_BENCH_BACKEND": "gloo", "SMX_BENCH_TEST_SLEEP_RANK": "1",
                     "SMX_BENCH_ATTEMPT_S": "6"}, timeout=110)
    took = time.monotonic() - t0
    assert p.returncode != 0
    assert not [l for l in lines if l.startswith("{")]
    assert "hit its 6 s limit" in p.stderr, p.stderr[-2000:]
    assert "rank 1 exit" in p.stderr and "rank 0 exit" in p.stderr
    assert took < 60, took                                   # two bounded attempts, not the rendezvous timeout


@pytest.mark.timeout(120)
def test_rank_without_a_device_refuses_in_one_line():
    """Each rank checks torch.cuda.device_count() > LOCAL_RANK before any collective (no GPU initialisation) and exits
    2 with one stderr line; the launcher shows it and does not repeat the run with eager launches."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the devices")
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"SMX_BENCH_BACKEND": "gloo"}, timeout=110)
    assert p.returncode == 2, (p.returncode, p.stderr[-2000:])
    assert not [l for l in lines if l.startswith("{")]
    assert "needs cuda:1 but this process sees" in p.stderr
    assert "repeating with eager launches" not in p.stderr


@pytest.mark.timeout(300)
def test_line_carries_every_ranks_own_time():
    p, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"],
                    {"SMX_BENCH_DRY_RUN": "1", "SMX_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(lines[0])
    assert len(out["per_rank_ms"]) == 2 and max(out["per_rank_ms"]) == pytest.approx(out["ms_per_step"], rel=1e-3, abs=1e-5)
