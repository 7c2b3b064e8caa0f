"""GPU: the row-partitioned device path with TWO processes sharing the one GPU of the test box.

RCCL refuses two ranks on one device, so the exchange step is staged through the host over gloo here (test-only
`HostStagedComm`); everything else is the product path: each rank builds ITS rows of the seeded operator on the
device (index_base slices of the counter generator, like bench.py), runs the HIP forward / adjoint / fused LSQR
halves, and the results are compared with the single-process operator and the CPU oracle.  What this leaves
uncovered is only RCCL itself with more than one rank (exercised with one rank in tests/test_gpu_lsqr.py).
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nrow, shape, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jets_jl_amd as J

    J.init(0)                                                     # both ranks on the one GPU
    dt = np.float32
    n = int(np.prod(shape))
    part = J.rowpart.partition_rows(nrow, world, rank)
    blk = J.JetSpace(dt, *shape)
    coeff = J.rand(J.JetBSpace([blk] * part.count), seed=1, stream=0, index_base=part.first * n)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0, index_base=part.first * n)

    class HostStagedComm:
        """Test-only exchange: device -> host -> gloo all-reduce -> device."""

        world, rank = dist.get_world_size(), dist.get_rank()

        def all_reduce_sum_(self, x, force=False):
            h = torch.from_numpy(x.to_numpy().ravel(order="F").copy())
            dist.all_reduce(h)
            x._upload(h.numpy())
            return x

        def all_reduce_scalars(self, values, op="sum"):
            t = torch.tensor(list(values), dtype=torch.float64)
            dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
            return t.tolist()

    shard = J.rowpart.for_device(part, A, comm=HostStagedComm())
    fwd = shard.mul_(J.zeros(J.range(A)), m)
    mt = shard.mul_adj_(J.rand(J.domain(A), seed=9, stream=rank), d)            # dirty, rank-dependent output buffer
    nrm = shard.norm_range(d, 2)
    dotv = shard.dot_range(d, fwd)
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    b = A * x_true
    res = J.lsqr(shard, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=15)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), fwd=fwd.to_numpy(), mt=mt.to_numpy().ravel(order="F"), nrm=nrm, dot=dotv,
             x=res.x.to_numpy().ravel(order="F"), r=np.array([h[1] for h in res.history]), first=part.first, count=part.count)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_single_process_operator(tmp_path, Jets, oracle):
    import torch.multiprocessing as mp

    world, nrow, shape = 2, 7, (32, 16, 8)
    n = int(np.prod(shape))
    mp.spawn(_worker, args=(world, _free_port(), nrow, shape, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    dt = np.float32
    ha = [oracle.rng_u01(dt, 1, 0, i * n, n) for i in range(nrow)]
    hm = oracle.rng_u01(dt, 2, 0, 0, n)
    hd = [oracle.rng_u01(dt, 3, 0, i * n, n) for i in range(nrow)]
    ops = [[oracle.Block("diag", n, coeff=g)] for g in ha]
    ref_fwd = np.concatenate(oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm]))
    ref_adj = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], hd)[0]
    # forward: the ranks' rows concatenate to the global result, bit for bit; no communication involved
    assert np.concatenate([res[0]["fwd"], res[1]["fwd"]]).tobytes() == ref_fwd.tobytes()
    assert (int(res[0]["count"]), int(res[1]["count"])) == (4, 3) and int(res[1]["first"]) == 4
    # adjoint: replicas identical, within the multi-GPU tolerance of the sequential reference
    assert res[0]["mt"].tobytes() == res[1]["mt"].tobytes()
    assert np.linalg.norm(res[0]["mt"].astype(np.float64) - ref_adj) <= 1e-5 * np.linalg.norm(ref_adj)
    flat_d = np.concatenate(hd).astype(np.float64)
    assert float(res[0]["nrm"]) == pytest.approx(np.linalg.norm(flat_d), rel=1e-6) and float(res[0]["nrm"]) == float(res[1]["nrm"])
    assert float(res[0]["dot"]) == pytest.approx(float(flat_d @ ref_fwd.astype(np.float64)), rel=1e-5)
    # LSQR on the partition == LSQR on the whole operator in one process (same device kernels), to fp32 round-off
    blk = Jets.JetSpace(dt, *shape)
    coeff = Jets.rand(Jets.JetBSpace([blk] * nrow), seed=1, stream=0)
    A = Jets.blockop([[Jets.JopDiagonal(c)] for c in coeff.arrays])
    x_true = Jets.rand(Jets.domain(A), seed=4, stream=0)
    one = Jets.lsqr(A, A * x_true, atol=0.0, btol=0.0, conlim=0.0, maxiter=15)
    x1 = one.x.to_numpy().ravel(order="F").astype(np.float64)
    assert res[0]["x"].tobytes() == res[1]["x"].tobytes()
    assert np.linalg.norm(res[0]["x"] - x1) <= 1e-5 * np.linalg.norm(x1)
    assert np.linalg.norm(x1 - x_true.to_numpy().ravel(order="F")) <= 1e-4 * np.linalg.norm(x1)
