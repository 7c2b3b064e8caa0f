#!/usr/bin/env python3
"""bench.py -- forward+adjoint mul! pairs/sec on a tall JopBlock (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): the configuration the metric is quoted on -- a 1024x1 tall JopBlock of
diagonal blocks, 256^3 Float32 per block (64 MiB/block; 64 GiB of coefficients + 64 GiB range
vector), synthetic U[0,1) data from the counter-based generator (seeds a=1, m=2, d=3), resident in
HBM before the timed region.  One step = one forward mul!(d, A, m) + one adjoint mul!(m, A', d).
With N GPUs the SAME operator is row-partitioned (1024/N block rows per rank, "strong" scaling);
the adjoint ends with one RCCL all-reduce of the 64 MiB domain vector.

Extras (not the metric): --fused-normal (the fused A'A kernel), --lsqr K (K LSQR iterations on b = A x_true).
BENCH_FORCE_DIST=1 under torch.distributed.run with one process exercises the RCCL path on a one-GPU box.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the library stream) and `cpu_baseline` (the CPU oracle, single thread, on a
bounded sample; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # the all-cores CPU baseline must not spin on barriers in a CPU-capped container

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nblocks", type=int, default=1024, help="global block rows of the tall operator")
    ap.add_argument("--edge", type=int, default=256, help="block is edge^3 Float32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-blocks", type=int, default=16)
    ap.add_argument("--cpu-pairs", type=int, default=4)
    ap.add_argument("--tune", type=str, default="", help="k=v,k=v kernel knobs (fwd_group, fwd_unroll, fwd_wg, adj_unroll, adj_depth, adj_wg, nt; 0 = automatic)")
    ap.add_argument("--fused-normal", action="store_true", help="also time the fused A'A kernel (extra field, not the metric)")
    ap.add_argument("--lsqr", type=int, default=0, help="also run this many LSQR iterations on b = A x_true (extra field, not the metric)")
    return ap.parse_args()


def cpu_baseline(edge: int, nblocks_full: int, sample_blocks: int, pairs: int) -> dict:
    """The CPU oracle's JetBlock_df!/df'! (reference loop structure incl. per-call temporaries,
    oracle/jets_oracle_body.inc) on a bounded sample, single thread."""
    import numpy as np
    from oracle import jets_oracle as jo

    n = edge ** 3
    a = [jo.rng_u01(np.float32, 1, 0, i * n, n) for i in range(sample_blocks)]
    m = jo.rng_u01(np.float32, 2, 0, 0, n)
    d = [jo.rng_u01(np.float32, 3, 0, i * n, n) for i in range(sample_blocks)]
    mt = np.zeros(n, dtype=np.float32)
    ops = [[jo.Block("diag", n, coeff=ai)] for ai in a]
    jo.block_df(ops, d, [m])
    jo.block_df_adj(ops, [mt], d)  # warm-up pair
    times = []
    for _ in range(pairs):
        t0 = time.perf_counter()
        jo.block_df(ops, d, [m])
        jo.block_df_adj(ops, [mt], d)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    bytes_pair = (4 * sample_blocks * n + 2 * n) * 4
    allcores = None
    try:   # separately labelled: NOT the reference's structure (it is single-threaded), same results bit for bit
        a2 = [np.empty(n, dtype=np.float32) for _ in range(sample_blocks)]     # untouched pages: first touched by the threads that use them
        d2 = [np.empty(n, dtype=np.float32) for _ in range(sample_blocks)]
        jo.fill_u01_omp_f32(a2, 1)
        jo.fill_u01_omp_f32(d2, 3)
        assert a2[-1][-7:].tobytes() == a[-1][-7:].tobytes()                 # the same values as the single-thread sample
        mt2 = np.zeros(n, dtype=np.float32)
        t0 = time.perf_counter()
        nt = jo.tall_diag_pair_omp_f32(a2, m, d2, mt2)
        if time.perf_counter() - t0 > 4 * med + 1.0:
            raise RuntimeError("OpenMP run slower than the single-thread run; skipped")
        assert mt2.tobytes() == mt.tobytes(), "all-cores adjoint differs from the single-thread loop"
        tt = []
        for _ in range(pairs):
            t0 = time.perf_counter()
            jo.tall_diag_pair_omp_f32(a2, m, d2, mt2)
            tt.append(time.perf_counter() - t0)
        tt.sort()
        mo = tt[len(tt) // 2]
        allcores = {"value": (1.0 / mo) * sample_blocks / nblocks_full, "unit": "pairs/s", "cores": nt, "kind": "port, OpenMP over element chunks (NUMA first-touch), rows in order inside a chunk",
                    "sample": f"same sample, median {mo:.4f} s/pair, {bytes_pair / mo / 1e9:.1f} GB/s algorithmic"}
    except Exception as e:
        allcores = {"value": None, "sample": f"failed: {e!r}"}
    return {
        "value": (1.0 / med) * sample_blocks / nblocks_full,
        "unit": "pairs/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sample_blocks} of {nblocks_full} block rows ({edge}^3 Float32 each), median of {pairs} pairs = {med:.3f} s/pair, "
                  f"{bytes_pair / med / 1e9:.1f} GB/s algorithmic; value = sample pairs/s x {sample_blocks}/{nblocks_full} (bandwidth-bound, linear in rows)",
        "build": "gcc -O2 -ftree-vectorize -ffp-contract=off (oracle/Makefile; BASELINE.md section 3)",
        "host_cores_available": os.cpu_count(),
        "all_cores_variant": allcores,
    }


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- one child process per GPU, the same
    environment torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) --
    BEFORE this process has made any GPU call (it never does: no torch import, no HIP), and relay the outcome: rank 0's ONE
    JSON line goes to our stdout as is, the exit code is non-zero if any rank failed.  No process that has touched the GPU
    is ever re-executed."""
    import signal
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GROUP_RANK="0", BENCH_SELF_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")               # dmabuf IPC: RCCL's intra-node transport needs it on this pool
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

    def stop_all(*_):
        for p in procs:
            if p.poll() is None:
                p.terminate()                                           # exact PIDs we started, nothing by pattern

    signal.signal(signal.SIGTERM, lambda *a: (stop_all(), sys.exit(143)))
    rc = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    stop_all()
            time.sleep(0.05)
    except KeyboardInterrupt:
        stop_all()
        rc = 130
    finally:
        deadline = time.time() + 15
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def main():
    args = parse_args()
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))                      # the parent never touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch

    force_dist = os.environ.get("BENCH_FORCE_DIST", "0") == "1"      # run the RCCL path even with one rank (validation)
    ndev = torch.cuda.device_count()                                  # (does not initialise the GPU)
    if ndev < 1:
        raise SystemExit("bench.py: no MI355X visible -- there is no CPU path to fall back to")
    backend = os.environ.get("BENCH_BACKEND", "nccl")                # "gloo": validation with several ranks on ONE GPU (RCCL refuses that)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if backend == "nccl" and world > 1 and ndev < local_world and ndev != 1:
        raise SystemExit(f"bench.py: {local_world} ranks on this node but only {ndev} devices visible (RCCL wants one device per rank)")
    if backend == "nccl" and world > 1 and ndev == 1 and os.environ.get("HIP_VISIBLE_DEVICES") is None and os.environ.get("ROCR_VISIBLE_DEVICES") is None:
        raise SystemExit(f"bench.py: --gpus {world} but ONE device visible: RCCL refuses several ranks on one device "
                         "(BENCH_BACKEND=gloo validates the multi-rank flow on a one-GPU box)")
    device = local_rank % ndev                                        # a launcher that narrows HIP_VISIBLE_DEVICES per rank leaves one device
    dist = None
    if world > 1 or (force_dist and "RANK" in os.environ):
        import torch.distributed as dist  # noqa: F811

        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)

    import jets_jl_amd as J

    J.init(device)
    if args.tune:
        J.tune(**{k: int(v) for k, v in (kv.split("=") for kv in args.tune.split(","))})

    edge, nblocks = args.edge, args.nblocks
    n = edge ** 3
    part = J.rowpart.partition_rows(nblocks, world, rank)
    nloc = part.count

    # ---- operator + vectors, resident in HBM ------------------------------------------------------
    blk = J.JetSpace("float32", edge, edge, edge)
    Rloc = J.JetBSpace([blk] * nloc)
    coeff = J.rand(Rloc, seed=1, stream=0, index_base=part.first * n)      # all diagonals of this rank: one slab
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0, index_base=part.first * n)
    mt = J.zeros(J.domain(A))
    shard = J.rowpart.for_device(part, A) if dist is not None else None
    J.synchronize()

    def forward():
        J.mul_(d, A, m)

    def adjoint():
        if shard is not None:
            shard.mul_adj_(mt, d, force_collective=force_dist)
        else:
            J.mul_(mt, A.H, d)

    # operator setup, outside the warm-up count: the first forwards of a large operator each try one grid walk (lazy autotune,
    # jh_blockop.hip: fwd_autotune_next -- no extra launches, no host sync; 16 calls until the choice is made) and the first
    # collective builds RCCL's channels -- with --warmup 0 neither may land in the timed region
    forward()
    adjoint()
    setup_forwards = 1
    while J.op_tune_get(A, "fwd_walk") == -1 and 0 < J.op_tune_get(A, "fwd_trials") and setup_forwards < 24:
        forward()
        J.synchronize()
        setup_forwards += 1
    for _ in range(args.warmup):
        forward()
        adjoint()

    ev = [[J.Event() for _ in range(3)] for _ in range(args.steps)]

    def fence():
        J.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        forward()
        ev[k][1].record()
        adjoint()
        ev[k][2].record()
    fence()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- multi-rank diagnostics, OUTSIDE the timed region: per-rank device times, the local adjoint kernel alone and the
    # stand-alone all-reduce, so that the exposed part of the exchange can be read off the line (adj_ms - adj_kernel_ms)
    multi = None
    if dist is not None:
        reps = max(3, min(args.steps, 10))
        f_ms = sum(ev[k][0].elapsed_ms(ev[k][1]) for k in range(args.steps)) / args.steps
        a_ms = sum(ev[k][1].elapsed_ms(ev[k][2]) for k in range(args.steps)) / args.steps
        J.mul_(mt, A.H, d)
        e0 = J.Event().record()
        for _ in range(reps):
            J.mul_(mt, A.H, d)                                       # this rank's rows, no exchange
        e1 = J.Event().record()
        k_ms = e0.elapsed_ms(e1) / reps
        fence()
        shard.comm.all_reduce_sum_(mt, force=force_dist)
        e0 = J.Event().record()
        for _ in range(reps):
            shard.comm.all_reduce_sum_(mt, force=force_dist)         # the whole 64 MiB domain vector in one piece, nothing to hide behind
        e1 = J.Event().record()
        ar_ms = e0.elapsed_ms(e1) / reps
        on_gpu = dist.get_backend() == "nccl"
        mine = torch.tensor([f_ms, a_ms, k_ms, ar_ms], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rows = [[float(x) for x in t.cpu()] for t in every]
        ar_bytes = n * 4
        ar_worst = max(r[3] for r in rows)
        multi = {
            "backend": dist.get_backend(), "rccl_nranks": world if on_gpu else None,
            "per_rank": [{"rank": r, "rows": J.rowpart.partition_rows(nblocks, world, r).count, "fwd_ms": rows[r][0], "adj_ms": rows[r][1],
                          "adj_kernel_ms": rows[r][2], "exposed_exchange_ms": rows[r][1] - rows[r][2]} for r in range(world)],
            "allreduce": {"bytes": ar_bytes, "chunks_in_adjoint": int(os.environ.get("JETS_AR_CHUNKS", "4")), "ms_standalone": ar_worst,
                          "busbw_GBps": (2.0 * (world - 1) / world) * ar_bytes / ar_worst / 1e6 if world > 1 and ar_worst > 0 else None,
                          "exposed_ms_max": max(r[1] - r[2] for r in rows)},
        }

    fwd_t = [ev[k][0].elapsed_ms(ev[k][1]) for k in range(args.steps)]
    adj_t = [ev[k][1].elapsed_ms(ev[k][2]) for k in range(args.steps)]
    pair_t = sorted(f + a for f, a in zip(fwd_t, adj_t))
    fwd_ms, adj_ms = sum(fwd_t) / args.steps, sum(adj_t) / args.steps
    s = 4
    fwd_bytes = (2 * nloc * n + n) * s        # read a, read m, write d      
    adj_bytes = (2 * nloc * n + n) * s        # read a, read d, write m
    pair_bytes_global = (4 * nblocks * n + 2 * n) * s
    kernels = {
        "forward": {"kernel": "k_tall_diag_fwd", "ms": fwd_ms, "bytes": fwd_bytes, "GBps": fwd_bytes / fwd_ms / 1e6},
        "adjoint": {"kernel": "k_tall_diag_adj" + ("+allreduce" if dist is not None else ""), "ms": adj_ms, "bytes": adj_bytes,
                    "GBps": adj_bytes / adj_ms / 1e6},
    }
    # a tall adjoint beyond 48 GiB is walked in launches of 512 rows (same bits, 2-3 % faster): report per LAUNCH
    adj_launches = max(1, J.tune_get("last_adj_launches")) if dist is None else 1
    kernels["adjoint"]["launches_per_call"] = adj_launches
    kernels["forward"]["launches_per_call"] = 1
    dom = max(kernels.values(), key=lambda kv: kv["ms"])
    traffic, traffic_round = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("nblocks") == nloc and tj.get("edge") == edge:
                key = dom["kernel"].split("+")[0]
                if key == "k_tall_diag_fwd":                            # PMC traffic is recorded per grid walk; the column-persistent
                    persistent = J.tune_get("last_fwd_rows_per_wg") >= nloc   # walk re-reads no m, like the row-concurrent one
                    key += "@walk%d" % (1 if persistent else J.tune_get("last_fwd_walk"))
                traffic = tj.get(key)
                if traffic is None:
                    key = key.split("@")[0]                              # counters taken over the walks the lazy autotune tried
                    traffic = tj.get(key)
                traffic_round = tj.get(key + "#round", "r01")
                if traffic is not None and dom["launches_per_call"] > 1 and traffic > 0.75 * dom["bytes"]:
                    traffic /= dom["launches_per_call"]                  # recorded when the whole call was one launch
        except Exception:
            traffic = None

    extra = {}
    if args.fused_normal and world == 1:
        y = J.zeros(J.domain(A))
        C = A.H @ A
        J.mul_(y, C, m)
        e0, e1 = J.Event().record(), None
        for _ in range(args.steps):
            J.mul_(y, C, m)
        e1 = J.Event().record()
        ms = e0.elapsed_ms(e1) / args.steps
        nb = (nloc * n + 2 * n) * s
        extra["fused_normal"] = {"ms": ms, "bytes": nb, "GBps": nb / ms / 1e6, "unfused_pair_ms": fwd_ms + adj_ms}

    if args.lsqr:
        x_true = J.rand(J.domain(A), seed=4, stream=0)
        J.mul_(d, A, x_true)                                   # b = A x_true (this rank's rows), in the range vector's storage
        # N > 1: the whole distributed loop behind the C ABI (jh_lsqr_solve_partitioned over the ABI's own RCCL communicator:
        # ranged steps, ranged all-reduces on the exchange stream, one host synchronisation per step) when every rank can set it
        # up; otherwise the Python driver over torch.distributed.  BENCH_LSQR_ABI=0 forces the latter.
        target, lsqr_driver = (shard if shard is not None else A), ("python driver over torch.distributed" if shard is not None else "jh_lsqr_solve")
        abi_mode = os.environ.get("BENCH_LSQR_ABI", "1")               # "force": also with ONE rank (validation / timing on a one-GPU box)
        if dist is not None and dist.get_backend() == "nccl" and (abi_mode == "force" or (world > 1 and abi_mode == "1")):
            import ctypes as _C

            from jets_jl_amd._ffi import lib as _lib

            ok = torch.tensor([1 if _lib.jh_comm_unique_id(_C.create_string_buffer(128)) == 0 else 0], device="cuda")   # loads librccl: no collective yet
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                def exchange_id(raw):
                    box = [raw]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]

                abi = J.rowpart.AbiComm(world, rank, exchange_id=exchange_id)
                if world == 1:
                    J.tune(force_dist=1)                             # run the exchange with the one-rank communicator
                target, lsqr_driver = J.rowpart.for_device(part, A, comm=abi), "jh_lsqr_solve_partitioned (C ABI communicator, pipelined exchange)"
        fence()
        t_l = time.perf_counter()
        res = J.lsqr(target, d, atol=0.0, btol=0.0, conlim=0.0, maxiter=args.lsqr, overwrite_b=True, force_maxiter=True)
        fence()
        t_l = time.perf_counter() - t_l
        if dist is not None:
            tt = torch.tensor([t_l], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_l = float(tt.item())
        err = (res.x - x_true).materialize()
        one_pass = os.environ.get("JETS_LSQR_FUSED_STEP", "1") != "0"
        # whole job per iteration.  one pass (jh_blockop_bidiag_step): read a, read u, write u = 3Nn, + v, w and the x/w/v updates
        # (~12n per rank); two halves: forward 3Nn+n, adjoint 2Nn+2n, updates ~8n per rank
        it_bytes = ((3 if one_pass else 5) * nblocks * n + (12 if one_pass else 11) * n * world) * s
        extra["lsqr"] = {"iterations": res.itn, "ms_per_iteration": 1e3 * t_l / max(res.itn, 1), "algorithmic_bytes_per_iteration": it_bytes,
                         "GBps": it_bytes * res.itn / t_l / 1e9, "rel_err_vs_x_true": float(J.norm(err)) / float(J.norm(x_true)),
                         "istop": res.istop, "r1norm_first_last": [res.history[0][1], res.history[-1][1]], "driver": lsqr_driver,
                         "schedule": ("one pass per iteration (jh_blockop_bidiag_step): u<-Av-(alpha/beta)u, ||u||^2 and A'u together; v<-A'u/beta-beta v on domain-sized vectors"
                                      if one_pass else "two fused halves: u<-Av-(alpha/beta)u with ||u||^2 (jh_blockop_mul_axpby), v<-A'u/beta-beta v with ||v||^2 (jh_blockop_mul_adj_axpby)")
                                     + "; u never normalised in memory"}

    if rank == 0:
        pairs_per_s = args.steps / elapsed
        out = {
            "metric": "fwd+adj mul! pairs/sec, 1024-block tall JopBlock (256^3 Float32 diagonal blocks)",
            "value": pairs_per_s,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_device": {"median": pair_t[len(pair_t) // 2], "min": pair_t[0], "max": pair_t[-1]},   # HIP events, this rank
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic U[0,1) Float32 from the counter-based generator (seeds a=1, m=2, d=3), generated on device",
            "config": {
                "workload": f"{nblocks}x1 tall JopBlock of diagonal JopLn, {edge}^3 Float32 blocks, fwd+adj mul! pair",
                "nblocks": nblocks, "block": [edge, edge, edge], "rows_per_gpu": nloc,
                "parallelism": f"row-partition x{world}" + (f" + {os.environ.get('BENCH_BACKEND', 'RCCL')} all-reduce({n * s / 2**20:.0f} MiB, pipelined in 4 chunks) in adjoint" if world > 1 else ""),
                "fwd_grid_walk": ("column-persistent (a workgroup streams every block row)" if J.tune_get("last_fwd_rows_per_wg") >= nloc
                                  else {0: "sequential row sweep", 1: "all rows concurrent"}.get(J.tune_get("last_fwd_walk"), "banded")),
                "fwd_walk_choice": {"candidate": J.op_tune_get(A, "fwd_walk"), "setup_forward_calls": setup_forwards},
                "tune": {k: J.tune_get(k) for k in ("fwd_group", "fwd_unroll", "fwd_wg", "fwd_order", "adj_unroll", "adj_depth", "adj_wg", "nt", "autotune")},
            },
            "achieved_GBps_pair": pair_bytes_global * pairs_per_s / 1e9,
            "roofline_frac_pair": pair_bytes_global * pairs_per_s / 1e9 / (HBM_PEAK_GBS * world),
            "roofline": {
                "bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": (None if traffic is None else
                                   {"measured_in_this_run": False, "file": "profiles/traffic_latest.json", "round": traffic_round,
                                    "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this same command (tools/prof_default_pmc.sh), "
                                           "(2*FETCH_SIZE + WRITE_SIZE) * 1024 B per launch of the named kernel (gfx950 corrections of MI355X_MICROARCH.md)"}),
                "bytes_per_launch": dom["bytes"] / dom["launches_per_call"], "ms_per_launch": dom["ms"] / dom["launches_per_call"],
                "launches_per_call": dom["launches_per_call"],
            },
            "kernels": kernels,
        }
        out.update(extra)
        if multi is not None:
            out["multi_gpu"] = multi
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(edge, nblocks, min(args.cpu_sample_blocks, nblocks), args.cpu_pairs)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
