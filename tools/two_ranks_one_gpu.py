#!/usr/bin/env python3
"""Two processes on ONE GPU drive the row-partitioned device path (exchange staged through the host over gloo; RCCL refuses
two ranks on one device).  Standalone: the launcher itself never touches the GPU.   python tools/two_ranks_one_gpu.py OUTDIR"""
import os
import socket
import sys
import time

import numpy as np

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



def _single(rank, nrow, shape, out_dir):
    """The same operator and solve in ONE process (run after the two ranks have exited)."""
    sys.path.insert(0, ROOT)
    import jets_jl_amd as J

    J.init(0)
    dt = np.float32
    blk = J.JetSpace(dt, *shape)
    coeff = J.rand(J.JetBSpace([blk] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    one = J.lsqr(A, A * x_true, atol=0.0, btol=0.0, conlim=0.0, maxiter=15)
    np.savez(os.path.join(out_dir, "single.npz"), x=one.x.to_numpy().ravel(order="F"), x_true=x_true.to_numpy().ravel(order="F"))


def check(out_dir, nrow, shape):
    """Compare the two ranks with the CPU oracle and with the single-process device run (test code: loads oracle/)."""
    sys.path.insert(0, ROOT)
    from oracle import jets_oracle as oracle

    n = int(np.prod(shape))
    res = [np.load(os.path.join(out_dir, f"r{r}.npz")) for r in range(2)]
    one = np.load(os.path.join(out_dir, "single.npz"))
    dt = np.float32
    ha = [oracle.rng_u01(dt, 1, 0, i * n, n) for i in range(nrow)]
    hm = oracle.rng_u01(dt, 2, 0, 0, n)
    hd = [oracle.rng_u01(dt, 3, 0, i * n, n) for i in range(nrow)]
    ops = [[oracle.Block("diag", n, coeff=g)] for g in ha]
    ref_fwd = np.concatenate(oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm]))
    ref_adj = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], hd)[0]
    # forward: the ranks' rows concatenate to the global result, bit for bit; no communication involved
    assert np.concatenate([res[0]["fwd"], res[1]["fwd"]]).tobytes() == ref_fwd.tobytes(), "forward"
    assert (int(res[0]["count"]), int(res[1]["count"])) == (4, 3) and int(res[1]["first"]) == 4, "partition"
    # adjoint: replicas identical, within the multi-GPU tolerance of the sequential reference
    assert res[0]["mt"].tobytes() == res[1]["mt"].tobytes(), "adjoint replicas differ"
    assert np.linalg.norm(res[0]["mt"].astype(np.float64) - ref_adj) <= 1e-5 * np.linalg.norm(ref_adj), "adjoint"
    flat_d = np.concatenate(hd).astype(np.float64)
    assert abs(float(res[0]["nrm"]) - np.linalg.norm(flat_d)) <= 1e-6 * np.linalg.norm(flat_d) and float(res[0]["nrm"]) == float(res[1]["nrm"]), "norm"
    want_dot = float(flat_d @ ref_fwd.astype(np.float64))
    assert abs(float(res[0]["dot"]) - want_dot) <= 1e-5 * abs(want_dot), "dot"
    # LSQR on the partition == LSQR on the whole operator in one process (same device kernels), to fp32 round-off
    x1 = one["x"].astype(np.float64)
    assert res[0]["x"].tobytes() == res[1]["x"].tobytes(), "LSQR replicas differ"
    assert np.linalg.norm(res[0]["x"] - x1) <= 1e-4 * np.linalg.norm(x1), "distributed LSQR vs single process"
    assert np.linalg.norm(x1 - one["x_true"]) <= 1e-3 * np.linalg.norm(x1), "LSQR vs x_true"


if __name__ == "__main__":
    import torch.multiprocessing as mp

    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/two_ranks"
    os.makedirs(out, exist_ok=True)
    nrow, shape = 7, (32, 16, 8)
    t0 = time.time()
    mp.spawn(_worker, args=(2, _free_port(), nrow, shape, out), nprocs=2, join=True)      # two device contexts at a time ...
    t1 = time.time()
    mp.spawn(_single, args=(nrow, shape, out), nprocs=1, join=True)                        # ... then one
    check(out, nrow, shape)
    print(f"two ranks on one GPU: {t1 - t0:.1f} s, single process {time.time() - t1:.1f} s", flush=True)
    print("TWO RANKS OK", flush=True)
