#!/usr/bin/env python3
"""W processes drive the row-partitioned device path and are checked against the CPU oracle (test infrastructure).

    python tools/ranks_check.py OUTDIR --ranks W --backend nccl     # one rank per GPU, RCCL over xGMI (needs >= W devices)
    python tools/ranks_check.py OUTDIR --ranks W --backend gloo     # W ranks on device 0, exchange staged through the host

The launcher itself never touches the GPU: it spawns the ranks, waits, then loads the CPU oracle and compares.
Per rank, on ITS rows of the seeded operator (index_base slices of the counter generator, like bench.py):
  A  rowpart.for_device over torch.distributed  -- forward, pipelined adjoint (JETS_AR_CHUNKS=4) and one-piece adjoint,
     range-side norm / dot, LSQR through the Python driver (pipelined one-pass step, deferred ||u||^2)
  B  (nccl only) the C ABI's own communicator: AbiComm -> jh_comm_init_rank / jh_comm_allreduce_sum /
     jh_comm_allreduce_scalars, and jh_lsqr_solve_partitioned (the whole distributed solve behind the ABI)
Checks (src/Jets.jl:1015-1031 forward rows independent, 1045-1053 adjoint = sum over rows):
  forward rows concatenate to the oracle's result BIT FOR BIT; adjoint rel-l2 <= 1e-5 vs the sequential oracle (the
  cross-rank sum order differs); every replica bit-identical; A and B agree to 1e-6; distributed LSQR vs the fp64 CPU LSQR
  of oracle/lsqr_ref.py on the whole operator.
"""
import argparse
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NROW_PER_RANK, SHAPE, LSQR_ITERS = 3, (64, 64, 20), 15      # 81 920 elements: three 64 KiB-aligned exchange ranges


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def nrow_for(world):
    return NROW_PER_RANK * world + 1          # uneven on purpose: rank 0 owns one row more


def _worker(rank, world, port, backend, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    device = rank if backend == "nccl" else 0
    if backend == "nccl":
        if torch.cuda.device_count() < world:
            raise SystemExit(f"--backend nccl needs {world} devices, {torch.cuda.device_count()} visible")
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import jets_jl_amd as J

    J.init(device)
    dt = np.float32
    nrow, shape = nrow_for(world), SHAPE
    n = int(np.prod(shape))
    part = J.rowpart.partition_rows(nrow, world, rank)
    blk = J.JetSpace(dt, *shape)
    coeff = J.rand(J.JetBSpace([blk] * part.count), seed=1, stream=0, index_base=part.first * n)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0, index_base=part.first * n)
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    out = dict(first=part.first, count=part.count)

    # ---- A: torch.distributed exchange (RCCL when backend == nccl), pipelined in 4 element ranges -------------------
    os.environ["JETS_AR_CHUNKS"] = "4"
    shard = J.rowpart.for_device(part, A)
    fwd = shard.mul_(J.zeros(J.range(A)), m)
    out["fwd"] = fwd.to_numpy()
    out["mt"] = shard.mul_adj_(J.rand(J.domain(A), seed=9, stream=rank), d).to_numpy().ravel(order="F")   # dirty, rank-dependent buffer
    os.environ["JETS_AR_CHUNKS"] = "1"
    whole = J.rowpart.for_device(part, A)                                                                  # one kernel + one all-reduce
    out["mt_whole"] = whole.mul_adj_(J.rand(J.domain(A), seed=8, stream=rank), d).to_numpy().ravel(order="F")
    out["nrm"] = shard.norm_range(d, 2)
    out["dot"] = shard.dot_range(d, fwd)
    b = A * x_true
    res = J.lsqr(shard, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=LSQR_ITERS)
    out["x"] = res.x.to_numpy().ravel(order="F")
    out["r"] = np.array([h[1] for h in res.history])

    # ---- B: the C ABI's own communicator (what a host without torch.distributed uses) --------------------------------
    if backend == "nccl":
        def exchange_id(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)       # ship rank 0's 128-byte id (any out-of-band channel would do)
            return box[0]

        comm = J.rowpart.AbiComm(world, rank, exchange_id=exchange_id)
        os.environ["JETS_AR_CHUNKS"] = "4"                        # the ABI's pipelined exchange: ranged all-reduces on its own stream + jh_comm_join
        shard_b = J.rowpart.for_device(part, A, comm=comm)
        out["mt_abi"] = shard_b.mul_adj_(J.rand(J.domain(A), seed=7, stream=rank), d).to_numpy().ravel(order="F")
        out["nrm_abi"] = shard_b.norm_range(d, 2)
        res_b = J.lsqr(shard_b, A * x_true, atol=0.0, btol=0.0, conlim=0.0, maxiter=LSQR_ITERS)      # jh_lsqr_solve_partitioned
        out["x_abi"] = res_b.x.to_numpy().ravel(order="F")
        # a rank-LOCAL solve while the communicator is alive must stay local (no hidden collective)
        if rank == 0:
            loc = J.lsqr(A, A * x_true, atol=0.0, btol=0.0, conlim=0.0, maxiter=3)
            out["x_local_itn"] = loc.itn
        dist.barrier()
        comm.close()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def check(out_dir, world, backend):
    """Compare the ranks with the CPU oracle (test code: loads oracle/)."""
    sys.path.insert(0, ROOT)
    from oracle import jets_oracle as oracle
    from oracle.lsqr_ref import lsqr_fp64

    nrow, shape = nrow_for(world), SHAPE
    n = int(np.prod(shape))
    res = [np.load(os.path.join(out_dir, f"r{r}.npz")) for r in range(world)]
    dt = np.float32
    ha = [oracle.rng_u01(dt, 1, 0, i * n, n) for i in range(nrow)]
    hm = oracle.rng_u01(dt, 2, 0, 0, n)
    hd = [oracle.rng_u01(dt, 3, 0, i * n, n) for i in range(nrow)]
    hx = oracle.rng_u01(dt, 4, 0, 0, n)
    ops = [[oracle.Block("diag", n, coeff=g)] for g in ha]
    ref_fwd = np.concatenate(oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm]))
    ref_adj = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], hd)[0]
    counts = [int(r["count"]) for r in res]
    assert sum(counts) == nrow and counts[0] == NROW_PER_RANK + 1 and all(c == NROW_PER_RANK for c in counts[1:]), f"partition {counts}"
    assert [int(r["first"]) for r in res] == list(np.cumsum([0] + counts[:-1])), "partition offsets"
    # forward: the ranks' rows concatenate to the global result, bit for bit; no communication involved
    assert np.concatenate([r["fwd"] for r in res]).tobytes() == ref_fwd.tobytes(), "forward rows differ from the oracle"
    # adjoint: replicas identical, within the multi-GPU tolerance of the sequential reference
    keys = ["mt", "mt_whole"] + (["mt_abi"] if backend == "nccl" else [])
    for key in keys:
        for r in res[1:]:
            assert r[key].tobytes() == res[0][key].tobytes(), f"{key}: replicas differ"
        err = np.linalg.norm(res[0][key].astype(np.float64) - ref_adj) / np.linalg.norm(ref_adj)
        assert err <= 1e-5, f"{key}: rel-l2 {err:.2e} vs the sequential oracle"
    # the pipelined and the one-piece exchange add the same per-rank partial sums element by element
    assert np.linalg.norm(res[0]["mt"].astype(np.float64) - res[0]["mt_whole"]) <= 1e-6 * np.linalg.norm(ref_adj), "pipelined vs whole"
    flat_d = np.concatenate(hd).astype(np.float64)
    for r in res:
        assert float(r["nrm"]) == float(res[0]["nrm"]) and abs(float(r["nrm"]) - np.linalg.norm(flat_d)) <= 1e-6 * np.linalg.norm(flat_d), "norm"
    want_dot = float(flat_d @ ref_fwd.astype(np.float64))
    assert abs(float(res[0]["dot"]) - want_dot) <= 1e-5 * abs(want_dot), "dot"
    # distributed LSQR vs the fp64 CPU LSQR on the WHOLE operator (oracle/lsqr_ref.py), same number of iterations
    a64 = np.stack(ha).astype(np.float64)
    b64 = (np.stack(ha) * hx[None, :]).astype(np.float64).ravel()        # b = A x_true: the Float32 products, like the device's
    xr, _ = lsqr_fp64(lambda v: (a64 * v[None, :]).ravel(), lambda u: (a64 * u.reshape(nrow, n)).sum(axis=0), b64, n, atol=0.0, btol=0.0,
                      conlim=0.0, maxiter=LSQR_ITERS)
    for key in ["x"] + (["x_abi"] if backend == "nccl" else []):
        for r in res[1:]:
            assert r[key].tobytes() == res[0][key].tobytes(), f"LSQR {key}: replicas differ"
        err = np.linalg.norm(res[0][key].astype(np.float64) - xr) / np.linalg.norm(xr)
        assert err <= 1e-4, f"LSQR {key}: rel-l2 {err:.2e} vs the fp64 CPU LSQR"
    assert res[0]["r"][-1] < 0.05 * res[0]["r"][0], "LSQR residual did not fall"
    if backend == "nccl":
        assert abs(float(res[0]["nrm_abi"]) - float(res[0]["nrm"])) <= 1e-12 * float(res[0]["nrm"]), "ABI scalar all-reduce"
        assert int(res[0]["x_local_itn"]) == 3, "rank-local solve under a live communicator"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    args = ap.parse_args()
    import torch.multiprocessing as mp

    os.makedirs(args.out, exist_ok=True)
    t0 = time.time()
    mp.spawn(_worker, args=(args.ranks, _free_port(), args.backend, args.out), nprocs=args.ranks, join=True)
    check(args.out, args.ranks, args.backend)
    print(f"{args.ranks} ranks over {args.backend}: {time.time() - t0:.1f} s", flush=True)
    print("RANKS OK", flush=True)


if __name__ == "__main__":
    main()
