"""CPU: bench.py's GPU-free supervisor (launch modes ranks / team / auto, heartbeat watchdog, fallback to a fresh team-mode
child) against FAKE workers -- bench.py's `fake_worker`, which walks through the phases without a GPU and fails or stalls where
the test says.  The real workers run in tests/test_gpu_bench_two_ranks.py and tests/test_gpu_bench_team.py."""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(extra_env, *argv, launcher=None, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("BENCH_")}
    env.update(BENCH_TEST_FAKE_WORKER="1", BENCH_TEST_NDEV="8")
    env.update(extra_env)
    cmd = (launcher or [sys.executable]) + [BENCH] + list(argv)
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, lines, time.time() - t0


def test_ranks_all_healthy_gives_one_line():
    out, lines, _ = run({}, "--gpus", "3")
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 3 and lines[0]["launch_mode"] == "ranks"
    assert "launch_fallback" not in lines[0] and lines[0]["launched_by"].startswith("bare python")


def test_a_rank_that_dies_before_the_first_collective_triggers_the_team_fallback():
    out, lines, _ = run({"BENCH_TEST_FAIL_BEFORE_COLLECTIVE": "1"}, "--gpus", "4")
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1, "exactly one line: the team-mode child's"
    j = lines[0]
    assert j["launch_mode"] == "team" and j["launch_fallback"]["from"] == "ranks" and "rank 1 exited with" in j["launch_fallback"]["reason"]
    assert "falling back to ONE fresh process" in out.stderr


def test_a_stalled_rank_is_named_with_its_phase_and_the_team_takes_over():
    out, lines, took = run({"BENCH_TEST_STALL_BEFORE_COLLECTIVE": "2", "BENCH_WATCHDOG_S": "2", "BENCH_WATCHDOG_IMPORT_S": "2"}, "--gpus", "3")
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["launch_mode"] == "team"
    why = lines[0]["launch_fallback"]["reason"]                        # names a stalled rank AND shows where every rank was
    assert "stalled: no heartbeat" in why and "2:'process-group-init'" in why and "0:'at-first-collective'" in why
    assert took < 60


def test_mode_ranks_never_falls_back():
    out, lines, _ = run({"BENCH_TEST_FAIL_BEFORE_COLLECTIVE": "0"}, "--gpus", "2", "--mode", "ranks")
    assert out.returncode != 0 and not lines
    assert "rank 0 exited with" in out.stderr and "no fallback" in out.stderr


def test_a_failure_after_the_first_collective_is_a_real_failure():
    out, lines, _ = run({"BENCH_TEST_FAIL_AFTER_COLLECTIVE": "1"}, "--gpus", "2")
    assert out.returncode != 0 and not lines
    assert "not a launch problem" in out.stderr


def test_mode_team_starts_one_child_only():
    out, lines, _ = run({"BENCH_TEST_FAIL_BEFORE_COLLECTIVE": "1"}, "--gpus", "8", "--mode", "team")   # no rank 1 exists in team mode
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["launch_mode"] == "team" and "launch_fallback" not in lines[0] and lines[0]["n_gpus"] == 8


def test_a_stalled_team_child_is_stopped_with_a_precise_error():
    out, lines, took = run({"BENCH_TEST_STALL_BEFORE_COLLECTIVE": "team", "BENCH_WATCHDOG_S": "2", "BENCH_WATCHDOG_IMPORT_S": "2"}, "--gpus", "2", "--mode", "team")
    assert out.returncode != 0 and not lines and took < 60
    assert "team-mode child stalled" in out.stderr and "process-group-init" in out.stderr


def _torchrun(nproc):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
            "--master-port", str(port)]


def test_under_torch_distributed_run_the_rank_processes_are_supervisors_too():
    out, lines, _ = run({}, "--gpus", "2", launcher=_torchrun(2), timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["launch_mode"] == "ranks" and lines[0]["launched_by"] == "torch.distributed.run"


def test_under_torch_distributed_run_a_dead_rank_still_ends_in_one_team_line():
    out, lines, _ = run({"BENCH_TEST_FAIL_BEFORE_COLLECTIVE": "1"}, "--gpus", "2", launcher=_torchrun(2), timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["launch_mode"] == "team" and "rank 1" in lines[0]["launch_fallback"]["reason"]


def test_one_visible_device_never_starts_n_rank_processes():
    """A one-GPU box: N RCCL ranks cannot work there, and starting them is not always recoverable (a process cap on the card kills
    the whole job) -- the plan is made before any worker exists: auto goes straight to the team, ranks refuses with the reason."""
    out, lines, _ = run({"BENCH_TEST_NDEV": "1", "BENCH_TEST_FAIL_BEFORE_COLLECTIVE": "0"}, "--gpus", "8")   # rank 0 would die if started
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["launch_mode"] == "team" and "1 device(s) visible for 8 ranks" in lines[0]["launch_fallback"]["reason"]
    assert "exited with" not in out.stderr                           # no rank process ever existed
    out, lines, _ = run({"BENCH_TEST_NDEV": "1"}, "--gpus", "8", "--mode", "ranks")
    assert out.returncode != 0 and not lines and "no rank process was started" in out.stderr
    out, lines, _ = run({"BENCH_TEST_NDEV": "4"}, "--gpus", "8")
    assert out.returncode != 0 and not lines and "4 device(s) visible for 8 ranks" in out.stderr
    out, lines, _ = run({"BENCH_TEST_NDEV": "1", "BENCH_BACKEND": "gloo"}, "--gpus", "2")   # validation over gloo: ranks share the device
    assert out.returncode == 0 and lines[0]["launch_mode"] == "ranks"


def test_under_torch_distributed_run_only_the_leader_plans():
    out, lines, _ = run({"BENCH_TEST_NDEV": "1"}, "--gpus", "2", launcher=_torchrun(2), timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["launch_mode"] == "team" and lines[0]["launched_by"] == "torch.distributed.run"


def test_a_long_silent_solve_is_not_a_stall_when_the_work_asked_for_explains_it():
    """One J.lsqr call of --lsqr K iterations is silent by construction: the limit of that phase grows with K (round-3 advisor finding:
    a healthy 120 s solve was reported as 'stalled')."""
    env = {"BENCH_TEST_LONG_SOLVE_S": "4", "BENCH_WATCHDOG_S": "2", "BENCH_WATCHDOG_IMPORT_S": "5", "BENCH_WATCHDOG_UNIT_S": "0.1"}
    out, lines, _ = run(env, "--gpus", "2", "--mode", "ranks", "--lsqr", "100")          # 100 iterations x 0.1 s = 10 s allowed
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    out, lines, _ = run(env, "--gpus", "2", "--mode", "ranks", "--lsqr", "10")           # 10 x 0.1 = 1 s -> the 2 s floor: a real stall
    assert out.returncode != 0 and not lines and "stalled: no heartbeat" in out.stderr and "'lsqr'" in out.stderr


def test_under_torch_distributed_run_the_leader_waits_for_every_rank():
    """The leader used to return as soon as ITS rank had exited 0 (and removed the heartbeat directory under the other ranks): a
    follower that is still working is waited for, and one that then fails makes the job fail."""
    port = socket.socket(); port.bind(("127.0.0.1", 0)); p = port.getsockname()[1]; port.close()
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(p)]
    out, lines, took = run({"BENCH_TEST_SLOW_RANK": "1", "BENCH_TEST_SLOW_RANK_S": "3"}, "--gpus", "2", "--mode", "ranks", launcher=launcher, timeout=180)
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    out, lines, took = run({"BENCH_TEST_SLOW_RANK": "1", "BENCH_TEST_SLOW_RANK_S": "3", "BENCH_TEST_SLOW_RANK_FAILS": "1"}, "--gpus", "2", "--mode", "ranks",
                           launcher=launcher, timeout=180)
    assert out.returncode != 0, "a rank failed after rank 0 had printed its line: the job must not report success"
