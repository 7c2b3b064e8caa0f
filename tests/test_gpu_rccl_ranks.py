"""GPU, MULTI-DEVICE: RCCL over xGMI with more than one rank -- runs by itself wherever >= 2 devices are visible.

The one-GPU test box skips these (RCCL refuses two ranks on one device; the same worker code runs there over gloo in
test_gpu_two_ranks_one_gpu.py).  On a multi-GPU node they exercise, with one process per GPU (tools/ranks_check.py):
`rowpart.for_device` over torch.distributed's "nccl" backend (pipelined and one-piece adjoint exchange, pipelined one-pass
LSQR step with the deferred ||u||^2) AND the C ABI's own communicator (jh_comm_init_rank / jh_comm_allreduce_sum /
jh_comm_allreduce_scalars / jh_lsqr_solve_partitioned): forward rows bit-exact vs the oracle, adjoint rel-l2 <= 1e-5
(src/Jets.jl:1045-1053 summed across ranks in another order), replicas bit-identical, distributed LSQR vs the fp64 CPU LSQR.

Rank counts: 2, and min(ndev, 4) -- not 8: a GPU box allows at most 6 of our processes on its cards at once and this pytest
process is one of them (JETS_TEST_MAX_RANKS overrides); the 8-rank flow is the driver's scaling run of bench.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ndev():
    import torch

    return torch.cuda.device_count()          # does not initialise the GPU


def _rank_counts():
    cap = int(os.environ.get("JETS_TEST_MAX_RANKS", "4"))
    n = _ndev()
    return sorted({2, max(2, min(n, cap))})


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_rccl_ranks(tmp_path, nranks):
    if _ndev() < 2:
        pytest.skip("one device visible: RCCL with >= 2 ranks needs >= 2 GPUs (same worker runs over gloo in test_gpu_two_ranks_one_gpu.py)")
    if nranks not in _rank_counts():
        pytest.skip(f"{nranks} ranks not scheduled on this box ({_ndev()} devices, cap {os.environ.get('JETS_TEST_MAX_RANKS', '4')})")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ranks_check.py"), str(tmp_path), "--ranks", str(nranks), "--backend", "nccl"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RANKS OK" in out.stdout


def test_bench_self_spawns_its_ranks_over_rccl():
    """`python bench.py --gpus 2` with no launcher: the parent starts the ranks itself (before any GPU call) and relays ONE line."""
    import json

    if _ndev() < 2:
        pytest.skip("one device visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--nblocks", "64",
                          "--edge", "128", "--lsqr", "10"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["multi_gpu"]["rccl_nranks"] == 2 and len(j["multi_gpu"]["per_rank"]) == 2
    assert j["lsqr"]["rel_err_vs_x_true"] < 1e-3


def test_bench_rccl_path_with_one_rank():
    """The bench's RCCL code path end to end on the one-GPU test box: a launcher-style environment with WORLD_SIZE=1 and
    BENCH_FORCE_DIST=1 makes rank 0 initialise torch.distributed's "nccl" backend, run the pipelined adjoint exchange, the
    multi-rank diagnostics and -- BENCH_LSQR_ABI=force -- jh_lsqr_solve_partitioned over the C ABI's own RCCL communicator.
    What N > 1 adds to this is only more ranks inside the same calls."""
    import json
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               BENCH_FORCE_DIST="1", BENCH_LSQR_ABI="force")
    env.pop("BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--nblocks", "24",
                          "--edge", "128", "--no-cpu-baseline", "--lsqr", "10"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    mg = j["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["rccl_nranks"] == 1 and mg["per_rank"][0]["rows"] == 24
    assert mg["allreduce"]["bytes"] == 128 ** 3 * 4 and mg["allreduce"]["ms_standalone"] > 0
    assert j["kernels"]["adjoint"]["kernel"].endswith("+allreduce")
    ls = j["lsqr"]
    assert ls["driver"].startswith("jh_lsqr_solve_partitioned") and ls["iterations"] == 10 and ls["rel_err_vs_x_true"] < 1e-3
