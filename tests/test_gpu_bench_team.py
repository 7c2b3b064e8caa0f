"""GPU: `bench.py --mode team` -- ONE worker process drives every member of the row partition through the C ABI's single-process
team (jh_comm_init_all, grouped ranged all-reduces).  On the one-GPU test box the 8 members are 8 contexts (streams) of that
device and the grouped sum is the device-side kernel; with >= 8 devices the very same command forms the team over RCCL
(ncclCommInitAll).  Checked: the partition, bit-identical replicas, the adjoint against the fp64 sum of the members' ordered
partial sums (rel l2 <= 1e-5: the sum order changes across members, src/Jets.jl:1045-1053) AND against the CPU oracle on element
slices regenerated from the counter generator (the bench itself never touches the oracle)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_team(tmp_path, nblocks, edge, members, extra=()):
    dump = str(tmp_path / "mt0.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(members), "--mode", "team", "--nblocks", str(nblocks), "--edge", str(edge),
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--check", "--dump", dump] + list(extra)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line"
    return json.loads(lines[0]), np.load(dump)


def oracle_slices(nblocks, edge, got, slices):
    """m~ = sum_i a_i .* (a_i .* m) on element slices: the products in Float32 as the kernels round them, the sum over the
    block rows in fp64 (the members' order differs from the sequential one: tolerance parity)."""
    from oracle import jets_oracle as jo

    n = edge ** 3
    for lo, cnt in slices:
        m = jo.rng_u01(np.float32, 2, 0, lo, cnt)
        acc = np.zeros(cnt, dtype=np.float64)
        for i in range(nblocks):
            a = jo.rng_u01(np.float32, 1, 0, i * n + lo, cnt)
            acc += (a * (a * m)).astype(np.float64)
        rel = np.linalg.norm(got[lo:lo + cnt].astype(np.float64) - acc) / np.linalg.norm(acc)
        assert rel <= 1e-5, (lo, cnt, rel)


@pytest.mark.parametrize("nblocks, rows", [(1024, [128] * 8), (1000, [125] * 8), (1003, [126, 126, 126, 125, 125, 125, 125, 125])])
def test_eight_members_through_team_mode(tmp_path, nblocks, rows):
    edge = 128
    j, mt0 = run_team(tmp_path, nblocks, edge, 8)
    assert j["n_gpus"] == 8 and j["launch_mode"].startswith("team") and j["unit"] == "pairs/s" and j["value"] > 0 and j["steps"] == 3
    mg = j["multi_gpu"]
    assert [r["rows"] for r in mg["per_rank"]] == rows and sum(rows) == nblocks
    assert all(r["fwd_ms"] > 0 and r["adj_ms"] > 0 and r["adj_kernel_ms"] > 0 for r in mg["per_rank"])
    assert mg["allreduce"]["bytes"] == edge ** 3 * 4 and mg["allreduce"]["chunks_in_adjoint"] == 4
    if mg["rccl_nranks"] is None:                                     # one device: members are streams of it -- flagged, not a scaling point
        assert j["valid_scaling_point"] is False and "ONE device" in mg["placement"]
    else:
        assert mg["rccl_nranks"] == 8
    ck = j["check"]
    assert ck["ok"] and ck["replicas_bit_identical"] and ck["adjoint_rel_l2_vs_fp64_sum_of_partials"] <= 1e-5
    # every member's slabs came from Jets.stream_pair (unprobed at this size: 1 GiB shards are below the probe's 4 GiB floor)
    assert len(j["config"]["placement"]) == 8 and all(p == {"probed": False} for p in j["config"]["placement"])
    n = edge ** 3
    assert mt0.shape == (n,) and mt0.dtype == np.float32
    oracle_slices(nblocks, edge, mt0, [(0, 4096), (n // 2 - 1000, 3000), (n - 4096, 4096), (3 * n // 4 + 16384 - 7, 64)])   # incl. a range boundary of the pipeline


def test_team_of_one_is_the_plain_pair(tmp_path):
    """--gpus 1 --mode team: a team of ONE through ncclCommInitAll -- same bits as the ordered single-context adjoint."""
    j, mt0 = run_team(tmp_path, 24, 64, 1)
    assert j["n_gpus"] == 1 and j["multi_gpu"]["rccl_nranks"] == 1 and j["check"]["ok"]
    assert j["check"]["adjoint_rel_l2_vs_fp64_sum_of_partials"] < 1e-7    # one member: the all-reduce adds nothing
    oracle_slices(24, 64, mt0, [(0, 64 ** 3)])
