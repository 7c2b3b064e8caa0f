"""CPU, world_size 2, gloo: the row partition + exchange step of the tall operator.

The sharding / collective logic of jets.jl_amd/rowpart.py is run by two processes over torch's gloo
backend with a TEST DOUBLE for the compute engine (numpy arrays + the CPU oracle).  The product
wiring (`rowpart.for_device`) uses the HIP path and RCCL; only the engine differs.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_rows_exactly_once():
    sys.path.insert(0, ROOT)
    from jets_jl_amd import rowpart

    for nrow in (1, 2, 7, 8, 1024, 1023):
        for world in (1, 2, 3, 4, 8):
            if world > nrow:
                continue
            parts = [rowpart.partition_rows(nrow, world, r) for r in range(world)]
            assert parts[0].first == 0 and sum(p.count for p in parts) == nrow
            for a, b in zip(parts, parts[1:]):
                assert a.first + a.count == b.first                       # contiguous, slab order preserved
            assert max(p.count for p in parts) - min(p.count for p in parts) <= 1
            for irow in (0, nrow // 2, nrow - 1):
                owner = parts[0].owner(irow)
                assert parts[owner].first <= irow < parts[owner].first + parts[owner].count
                assert parts[0].local_index(irow) == irow - parts[owner].first
    p = rowpart.partition_rows(1024, 8, 3)
    assert (p.first, p.count) == (384, 128)                                # config 4: 128 rows = 16 GiB of a per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nrow, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jets_jl_amd import rowpart
    from oracle import jets_oracle as jo

    dt = np.float32
    part = rowpart.partition_rows(nrow, world, rank)
    # this rank's slice of the global seeded vectors (index_base = first * n), exactly like bench.py
    a_loc = [jo.rng_u01(dt, 1, 0, (part.first + i) * n, n) for i in range(part.count)]
    d_loc = [jo.rng_u01(dt, 3, 0, (part.first + i) * n, n) for i in range(part.count)]
    m = jo.rng_u01(dt, 2, 0, 0, n)
    ops = [[jo.Block("diag", n, coeff=g)] for g in a_loc]

    comm = rowpart.Comm(as_tensor=torch.from_numpy)
    shard = rowpart.RowPartitionedOp(
        part, ops, comm,
        local_mul=lambda d, A, mm: jo.block_df(A, d, [mm]),
        local_mul_adj=lambda mm, A, d: jo.block_df_adj(A, [mm], d)[0],
        local_dot=lambda x, y: jo.barr_dot(x, y),
        local_norm=lambda x, p: jo.barr_norm(x, p),
    )
    fwd = shard.mul_([np.zeros(n, dtype=dt) for _ in range(part.count)], m)       # no communication
    mt = np.full(n, 9.0, dtype=dt)
    local_only = jo.block_df_adj(ops, [np.zeros(n, dtype=dt)], d_loc)[0].copy()
    shard.mul_adj_(mt, d_loc)                                                      # local ordered sum + all-reduce
    dotv = shard.dot_range(d_loc, fwd)
    nrm2, nrminf, nrm1 = shard.norm_range(d_loc, 2), shard.norm_range(d_loc, float("inf")), shard.norm_range(d_loc, 1)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), fwd=np.concatenate(fwd), mt=mt, local=local_only, dot=dotv,
             nrm=np.array([nrm2, nrminf, nrm1]), first=part.first, count=part.count)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nrow,n", [(6, 1000), (5, 257)])
def test_world_size_2_forward_is_local_and_adjoint_all_reduces(tmp_path, nrow, n):
    import torch.multiprocessing as mp

    from oracle import jets_oracle as jo

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, nrow, n, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]

    dt = np.float32
    a = [jo.rng_u01(dt, 1, 0, i * n, n) for i in range(nrow)]
    d = [jo.rng_u01(dt, 3, 0, i * n, n) for i in range(nrow)]
    m = jo.rng_u01(dt, 2, 0, 0, n)
    ops = [[jo.Block("diag", n, coeff=g)] for g in a]
    ref_fwd = np.concatenate(jo.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [m]))
    ref_adj = jo.block_df_adj(ops, [np.zeros(n, dtype=dt)], d)[0]

    # forward: each rank holds exactly its rows of the global result, bit for bit
    got_fwd = np.concatenate([res[r]["fwd"] for r in range(world)])
    assert got_fwd.tobytes() == ref_fwd.tobytes()
    assert int(res[0]["count"]) + int(res[1]["count"]) == nrow and int(res[1]["first"]) == int(res[0]["count"])
    # adjoint: replicas identical after the all-reduce, equal to the sum of the two ordered local sums,
    # and within the stated multi-GPU tolerance (rel l2 <= 1e-5) of the sequential reference
    assert res[0]["mt"].tobytes() == res[1]["mt"].tobytes()
    assert np.array_equal(res[0]["mt"], res[0]["local"] + res[1]["local"])
    rel = np.linalg.norm(res[0]["mt"].astype(np.float64) - ref_adj) / np.linalg.norm(ref_adj)
    assert rel <= 1e-5
    # range-side reductions are global
    flat_d = np.concatenate(d).astype(np.float64)
    assert float(res[0]["dot"]) == pytest.approx(float(np.dot(flat_d, ref_fwd.astype(np.float64))), rel=1e-5)
    assert float(res[0]["dot"]) == float(res[1]["dot"])
    assert res[0]["nrm"][0] == pytest.approx(np.linalg.norm(flat_d), rel=1e-5)
    assert res[0]["nrm"][1] == pytest.approx(np.abs(flat_d).max(), rel=1e-7)
    assert res[0]["nrm"][2] == pytest.approx(np.abs(flat_d).sum(), rel=1e-5)
