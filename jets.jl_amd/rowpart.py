"""Row partition of a tall block operator across the GPUs of one node.

A tall JopBlock (many block rows x one block column, src/Jets.jl:926-933) shards naturally:
the forward needs no exchange (each range block depends only on m, src/Jets.jl:1015-1031) and the
adjoint is a sum over rows (1045-1053).  One process per GPU; rank g owns the contiguous rows
[first, first+count) -- the slab order of JetBSpace.indices is preserved, so a global block index
maps to (rank, local index) by integer arithmetic.  Domain-side vectors are replicated; the only
data-path collective is ONE all-reduce (RCCL over xGMI via torch.distributed's "nccl" backend) of
the domain vector after the local adjoint, and scalar all-reduces for range-side dot/norm.

The summation order across ranks differs from the sequential reference => tolerance parity at
world_size > 1 (bit-exact at world_size 1).

The compute engine is injected (`local_mul`, `local_mul_adj`, `as_tensor`) so the sharding and
collective logic can be exercised with world_size-2 gloo tests on CPU against a test double; the
product wiring (`for_device`) uses the HIP path only.
"""
from __future__ import annotations

import builtins
import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable

__all__ = ["RowPartition", "partition_rows", "Comm", "AbiComm", "RowPartitionedOp", "for_device"]


@dataclass(frozen=True)
class RowPartition:
    nrow: int        # global block rows
    world: int
    rank: int
    first: int       # first global row owned by this rank
    count: int       # rows owned

    def owner(self, irow: int) -> int:
        """Rank owning global row `irow`."""
        base, rem = divmod(self.nrow, self.world)
        cut = rem * (base + 1)
        return irow // (base + 1) if irow < cut else rem + (irow - cut) // builtins.max(base, 1)

    def local_index(self, irow: int) -> int:
        return irow - partition_rows(self.nrow, self.world, self.owner(irow)).first


def partition_rows(nrow: int, world: int, rank: int) -> RowPartition:
    """Contiguous chunks; the first nrow % world ranks get one extra row."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(nrow, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + builtins.min(rank, rem)
    return RowPartition(nrow, world, rank, first, count)


class Comm:
    """Thin wrapper over torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests)."""

    def __init__(self, as_tensor: Callable, stream_ctx: Callable | None = None):
        import torch.distributed as dist

        self._dist = dist
        self._as_tensor = as_tensor
        self._stream_ctx = stream_ctx
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self._views = {}   # id(vector) -> (vector, tensor): the solver all-reduces the same few vectors every iteration

    def all_reduce_sum_(self, x, force: bool = False):
        """In-place sum of a replicated domain vector over all ranks (the adjoint accumulate).
        `force` runs the collective even with one rank (exercises the RCCL path on a one-GPU box)."""
        if self.world == 1 and not (force and self._dist.is_initialized()):
            return x
        hit = self._views.get(id(x))
        if hit is None or hit[0] is not x:
            if len(self._views) > 64:
                self._views.clear()
            hit = (x, self._as_tensor(x))
            self._views[id(x)] = hit
        t = hit[1]
        if self._stream_ctx is not None:
            with self._stream_ctx():
                self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        else:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return x

    def all_reduce_scalars(self, values, op: str = "sum"):
        """Batched scalar all-reduce (range-side dot / norm^2 / extrema partials), fp64 on the host path."""
        import torch

        if self.world == 1:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64)
        backend = self._dist.get_backend()
        if backend == "nccl":
            t = t.cuda()
        red = {"sum": self._dist.ReduceOp.SUM, "max": self._dist.ReduceOp.MAX, "min": self._dist.ReduceOp.MIN}[op]
        self._dist.all_reduce(t, op=red)
        return t.cpu().tolist()

    def barrier(self):
        if self.world > 1:
            self._dist.barrier()


class AbiComm:
    """The same exchange step through the C ABI's own RCCL entry points (jh_comm_*, include/jetship.h) -- what a
    host without torch.distributed (the Julia binding) uses.  `exchange_id` ships rank 0's 128-byte id to the other
    ranks (MPI broadcast, a socket, a file); with one rank it is not needed."""

    def __init__(self, nranks: int = 1, rank: int = 0, exchange_id: Callable | None = None):
        import ctypes as C

        from ._ffi import lib, check
        from . import device as _device

        _device.init()
        self._C, self._lib, self._check = C, lib, check
        ident = C.create_string_buffer(128)
        if rank == 0:
            check(lib.jh_comm_unique_id(ident))
        if nranks > 1:
            if exchange_id is None:
                raise ValueError("exchange_id(bytes_or_None) -> bytes is required for more than one rank")
            ident = C.create_string_buffer(exchange_id(ident.raw if rank == 0 else None), 128)
        check(lib.jh_comm_init_rank(ident, nranks, rank))
        self.world, self.rank = nranks, rank

    def all_reduce_sum_(self, x, force: bool = False):
        if self.world == 1 and not force:
            return x
        self._check(self._lib.jh_comm_allreduce_sum(x.handle))
        return x

    def all_reduce_scalars(self, values, op: str = "sum"):
        vals = list(values)
        buf = (self._C.c_double * len(vals))(*vals)
        self._check(self._lib.jh_comm_allreduce_scalars(buf, len(vals), {"sum": 0, "max": 1, "min": 2}[op]))
        return list(buf)

    def barrier(self):
        self.all_reduce_scalars([0.0])

    def close(self):
        self._check(self._lib.jh_comm_destroy())


class RowPartitionedOp:
    """This rank's shard of a tall block operator plus the exchange step."""

    def __init__(self, part: RowPartition, local_op, comm: Comm, local_mul: Callable, local_mul_adj: Callable,
                 local_dot: Callable, local_norm: Callable, pipelined_adj: Callable | None = None,
                 pipelined_step: Callable | None = None, pipelined_normal: Callable | None = None):
        self.part, self.local_op, self.comm = part, local_op, comm
        self._mul, self._mul_adj, self._dot, self._norm = local_mul, local_mul_adj, local_dot, local_norm
        self._pipelined_adj = pipelined_adj   # optional: local adjoint and all-reduce pipelined chunk by chunk
        self._pipelined_step = pipelined_step  # optional: one-pass Golub-Kahan step, its w all-reduced chunk by chunk
        self._pipelined_normal = pipelined_normal  # optional: the fused A'A, its result all-reduced chunk by chunk

    def normal_mul_(self, y, m, tmp_local=None, force_collective: bool = False):
        """y = (A'A) m = sum over ALL rows A_i'(A_i m)  (JetComposite_df! over (A', A), src/Jets.jl:530-534, on a row partition): every
        rank's fused A_k'A_k m -- its coefficients read once, no range-side intermediate -- in element ranges
        (jh_blockop_normal_mul_range), the all-reduce of a finished range under the next range's kernel.  Operators without the
        fused kernel run forward then adjoint through `tmp_local` (a range vector of this rank's rows)."""
        if self._pipelined_normal is not None and (self.comm.world > 1 or force_collective):
            if self._pipelined_normal(y, self.local_op, m):
                return y
        if tmp_local is None:
            raise ValueError("normal_mul_: this operator has no fused A'A; pass tmp_local (a range vector of this rank's rows)")
        self.mul_(tmp_local, m)
        return self.mul_adj_(y, tmp_local, force_collective=force_collective)

    def bidiag_step_(self, u_local, v, w, alpha: float, beta: float, force_collective: bool = False):
        """u_local <- alpha*(A_local v) + beta*u_local ; w <- sum over ALL ranks of A_local' u_local, the all-reduce of a
        finished chunk of w overlapping the kernel of the next.  Returns the GLOBAL ||u||^2, or None when the local
        operator has no ranged one-pass kernel (the caller then runs the step in one piece)."""
        if self._pipelined_step is None or not (self.comm.world > 1 or force_collective):
            return None
        local = self._pipelined_step(u_local, v, w, alpha, beta)
        if local is None:
            return None
        if isinstance(local, tuple):                          # (value, True): already summed over the ranks (jh_comm_allreduce_normsq)
            return local[0]
        return self.comm.all_reduce_scalars([local], "sum")[0]

    def mul_(self, d_local, m):
        """d_local = A[rows of this rank] m   -- no communication."""
        return self._mul(d_local, self.local_op, m)

    def mul_adj_(self, m, d_local, force_collective: bool = False):
        """m = sum over ALL rows A_i' d_i  -- local ordered sum, then one all-reduce (or, with the device wiring,
        the two pipelined chunk by chunk: all-reduce of chunk k overlaps the kernel of chunk k+1)."""
        if self._pipelined_adj is not None and (self.comm.world > 1 or force_collective):
            if self._pipelined_adj(m, self.local_op, d_local):
                return m
        self._mul_adj(m, self.local_op, d_local)
        return self.comm.all_reduce_sum_(m, force=force_collective)

    def dot_range(self, x_local, y_local) -> float:
        return self.comm.all_reduce_scalars([float(self._dot(x_local, y_local))], "sum")[0]

    def norm_range(self, x_local, p: float = 2) -> float:
        if p == 2:
            return math.sqrt(self.comm.all_reduce_scalars([float(self._norm(x_local, 2)) ** 2], "sum")[0])
        if p == math.inf:
            return self.comm.all_reduce_scalars([float(self._norm(x_local, p))], "max")[0]
        if p == -math.inf:
            return self.comm.all_reduce_scalars([float(self._norm(x_local, p))], "min")[0]
        if p in (0, 1):
            return self.comm.all_reduce_scalars([float(self._norm(x_local, p))], "sum")[0]
        return self.comm.all_reduce_scalars([float(self._norm(x_local, p)) ** p], "sum")[0] ** (1.0 / p)


def for_device(part: RowPartition, local_op, comm=None) -> RowPartitionedOp:
    """Product wiring: HIP kernels for the local work, RCCL for the exchange -- through torch.distributed's "nccl"
    backend (default; the all-reduce is ordered against the library's HIP stream with torch.cuda.ExternalStream)
    or through the C ABI's own RCCL entry points when an AbiComm is passed."""
    from . import device as _device
    from .arrays import dot, norm
    from .jets import mul_, adjoint

    if comm is not None:
        if isinstance(comm, AbiComm):
            return _for_device_abi(part, local_op, comm)
        return RowPartitionedOp(part, local_op, comm, lambda d, A, m: mul_(d, A, m), lambda m, A, d: mul_(m, adjoint(A), d), dot, norm)
    import torch

    ext = torch.cuda.ExternalStream(_device.stream_handle(), device=torch.device("cuda", _device.init()))

    def as_tensor(x):
        # RCCL needs a contiguous tensor: hand it the flat 1-D view of the slab (shares memory with x)
        flat = x if len(x.shape) == 1 else x.reshape((x.length(),))
        t = torch.as_tensor(flat, device=torch.device("cuda", _device.init()))
        t._jets_owner = flat  # keep the view alive while torch holds the pointer
        return t

    def stream_ctx():
        return torch.cuda.stream(ext)

    comm = Comm(as_tensor, stream_ctx)

    import os

    import torch.distributed as dist

    from ._ffi import lib, check, JetsHipError
    from . import jetblock as _blk

    nchunks = int(os.environ.get("JETS_AR_CHUNKS", "4"))
    views = {}      # id(vector) -> (vector, flat torch view): a solver exchanges the same one or two vectors every iteration

    def tensor_of(x):
        hit = views.get(id(x))
        if hit is None or hit[0] is not x:
            if len(views) >= 8:                               # bounded: a view pins its vector (64 MiB at the headline size)
                views.clear()
            hit = views[id(x)] = (x, as_tensor(x))
        return hit[1]

    def native_of(A):
        if nchunks <= 1 or not dist.is_initialized() or not _blk.isblockop(A):
            return None
        jt = A.jet
        return _blk._native_op(jt.s.get("_native"), jt.s["ops"], jt.rng.eltype())

    def chunk_bounds(n):
        return _chunk_bounds(n, nchunks)

    def exchange(t, lo, cnt, works):
        with torch.cuda.stream(ext):                          # RCCL's stream waits for the library stream up to here
            works.append(dist.all_reduce(t[lo:lo + cnt], op=dist.ReduceOp.SUM, async_op=True))

    def join(works):
        with torch.cuda.stream(ext):
            for w in works:
                w.wait()                                      # the library stream waits for every chunk's all-reduce

    def pipelined_adj(m, A, d) -> bool:
        """Local adjoint in `nchunks` element ranges (jh_blockop_mul_adj_range); the all-reduce of a finished range
        runs on RCCL's stream while the kernel of the next range runs on the library stream.  Same values as the
        unpipelined path.  Returns False when the operator has no ranged kernel (then the caller does it in one piece)."""
        nat = native_of(A)
        if nat is None:
            return False
        t = tensor_of(m)
        works = []
        try:
            for lo, cnt in chunk_bounds(m.length()):
                check(lib.jh_blockop_mul_adj_range(nat.handle, m.handle, d.handle, lo, cnt))
                exchange(t, lo, cnt, works)
        except JetsHipError as e:
            if e.status == 4 and not works:                   # JH_ERR_UNSUPPORTED before anything was enqueued
                return False
            raise
        join(works)
        return True

    def pipelined_normal(y, A, m) -> bool:
        """The fused A'A in `nchunks` element ranges (jh_blockop_normal_mul_range), exchanged like the adjoint's."""
        nat = native_of(A)
        if nat is None:
            return False
        t = tensor_of(y)
        works = []
        try:
            for lo, cnt in chunk_bounds(y.length()):
                check(lib.jh_blockop_normal_mul_range(nat.handle, y.handle, m.handle, lo, cnt))
                exchange(t, lo, cnt, works)
        except JetsHipError as e:
            if e.status == 4 and not works:
                return False
            raise
        join(works)
        return True

    def pipelined_step(u, v, w, alpha, beta):
        """jh_blockop_bidiag_step in `nchunks` element ranges, enqueued back to back: every range adds its share of ||u||^2
        to a device-side accumulator (jh_normsq_reset / normsq == NULL / jh_normsq_read), so the host synchronises ONCE per
        step, after the last range, while the all-reduces of the finished ranges of w run under the later kernels.  Returns
        the local ||u||^2, or None when the operator has no ranged one-pass kernel."""
        nat = native_of(local_op)
        if nat is None:
            return None
        t = tensor_of(w)
        works = []
        out = C.c_double(0)
        try:
            check(lib.jh_normsq_reset())
            for lo, cnt in chunk_bounds(w.length()):
                check(lib.jh_blockop_bidiag_step_range(nat.handle, u.handle, v.handle, w.handle, float(alpha), float(beta), lo, cnt, None))
                exchange(t, lo, cnt, works)
        except JetsHipError as e:
            if e.status == 4 and not works:
                return None
            raise
        check(lib.jh_normsq_read(C.byref(out)))                # the one read-back (library stream: kernels only, not the exchange)
        join(works)
        return out.value

    return RowPartitionedOp(part, local_op, comm, lambda d, A, m: mul_(d, A, m), lambda m, A, d: mul_(m, adjoint(A), d), dot, norm,
                            pipelined_adj=pipelined_adj, pipelined_step=pipelined_step, pipelined_normal=pipelined_normal)


def _chunk_bounds(n: int, nchunks: int):
    step = -(-n // nchunks)
    step = -(-step // 16384) * 16384                          # chunk bounds on 64 KiB boundaries
    lo = 0
    while lo < n:
        cnt = builtins.min(step, n - lo)
        yield lo, cnt
        lo += cnt


def _for_device_abi(part: RowPartition, local_op, comm: AbiComm) -> RowPartitionedOp:
    """The same wiring over the C ABI's own communicator and exchange stream (jh_comm_allreduce_sum_range / jh_comm_join /
    jh_comm_allreduce_normsq): what a host without torch.distributed gets -- the pipelined adjoint and the pipelined one-pass
    step with ONE host synchronisation per step."""
    import os

    from ._ffi import lib, check, JetsHipError
    from . import jetblock as _blk
    from .arrays import dot, norm
    from .jets import mul_, adjoint

    nchunks = int(os.environ.get("JETS_AR_CHUNKS", "4"))

    def native_of(A):
        if nchunks <= 1 or not _blk.isblockop(A):             # (with one rank the caller only comes here when forced: validation)
            return None
        jt = A.jet
        return _blk._native_op(jt.s.get("_native"), jt.s["ops"], jt.rng.eltype())

    def pipelined_adj(m, A, d) -> bool:
        nat = native_of(A)
        if nat is None:
            return False
        done = 0
        try:
            for lo, cnt in _chunk_bounds(m.length(), nchunks):
                check(lib.jh_blockop_mul_adj_range(nat.handle, m.handle, d.handle, lo, cnt))
                check(lib.jh_comm_allreduce_sum_range(m.handle, lo, cnt))
                done += 1
        except JetsHipError as e:
            if e.status == 4 and done == 0:
                return False
            raise
        check(lib.jh_comm_join())
        return True

    def pipelined_normal(y, A, m) -> bool:
        nat = native_of(A)
        if nat is None:
            return False
        done = 0
        try:
            for lo, cnt in _chunk_bounds(y.length(), nchunks):
                check(lib.jh_blockop_normal_mul_range(nat.handle, y.handle, m.handle, lo, cnt))
                check(lib.jh_comm_allreduce_sum_range(y.handle, lo, cnt))
                done += 1
        except JetsHipError as e:
            if e.status == 4 and done == 0:
                return False
            raise
        check(lib.jh_comm_join())
        return True

    def pipelined_step(u, v, w, alpha, beta):
        nat = native_of(local_op)
        if nat is None:
            return None
        out = C.c_double(0)
        done = 0
        try:
            check(lib.jh_normsq_reset())
            for lo, cnt in _chunk_bounds(w.length(), nchunks):
                check(lib.jh_blockop_bidiag_step_range(nat.handle, u.handle, v.handle, w.handle, float(alpha), float(beta), lo, cnt, None))
                check(lib.jh_comm_allreduce_sum_range(w.handle, lo, cnt))
                done += 1
        except JetsHipError as e:
            if e.status == 4 and done == 0:
                return None
            raise
        check(lib.jh_comm_allreduce_normsq(C.byref(out)))     # global ||u||^2; kernels and ranged all-reduces are complete on return
        return (out.value, True)

    return RowPartitionedOp(part, local_op, comm, lambda d, A, m: mul_(d, A, m), lambda m, A, d: mul_(m, adjoint(A), d), dot, norm,
                            pipelined_adj=pipelined_adj, pipelined_step=pipelined_step, pipelined_normal=pipelined_normal)


# ------------------------------------------------------------------ ONE process, several contexts -------------------------
class TeamVec:
    """One vector per member of a single-process team: a RANGE-side TeamVec holds every member's rows, a DOMAIN-side one
    the members' replicas (kept identical by running the same deterministic updates on each)."""

    def __init__(self, members):
        self.members = list(members)

    def __len__(self):
        return len(self.members)

    def __getitem__(self, k):
        return self.members[k]

    def close(self):
        for x in self.members:
            x.close()


class Team:
    """SURVEY section 8e's single-process form: this process holds one context per GPU (`device.context_create` /
    `init(device)`), `jh_comm_init_all` makes them a team, and the exchange step is the members' all-reduces issued between
    jh_comm_group_begin / jh_comm_group_end.  The members may also be several contexts of ONE GPU (then the grouped sum is a
    device kernel: RCCL refuses two ranks on a device) -- which is how the one-GPU test box exercises this flow.

        team = Team([ctx0, ctx1, ...])
        with using_context(team.contexts[k]): A_k = blockop(rows of member k)      # built by the caller, member by member
        T = team.operator([A_0, A_1, ...])
        T.mul_(d, m); T.mul_adj_(m, d); lsqr(T, b)                                  # d, m, b: TeamVec
    """

    def __init__(self, contexts):
        import ctypes as C

        from ._ffi import lib, check

        self.contexts = [int(c) for c in contexts]
        arr = (C.c_int * len(self.contexts))(*self.contexts)
        check(lib.jh_comm_init_all(len(self.contexts), arr))
        self.world = len(self.contexts)
        self._lib, self._check = lib, check

    def each(self):
        """Iterate over (member index, context id) with that context current."""
        from . import device as _device

        for k, ctx in enumerate(self.contexts):
            _device.context_use(ctx)
            yield k, ctx

    def zeros(self, spaces) -> TeamVec:
        """One zero vector per member: `spaces` is one space (replicated, domain side) or one per member (range side)."""
        from .arrays import zeros

        per = spaces if isinstance(spaces, (list, tuple)) else [spaces] * self.world
        return TeamVec([zeros(per[k]) for k, _ in self.each()])

    def group(self):
        return _TeamGroup(self)

    def operator(self, local_ops) -> "TeamOp":
        return TeamOp(self, local_ops)

    def synchronize(self):
        from . import device as _device

        for _ in self.each():
            _device.synchronize()

    def close(self):
        from . import device as _device

        _device.context_use(self.contexts[0])
        self._check(self._lib.jh_comm_destroy())


class _TeamGroup:
    def __init__(self, team):
        self.team = team

    def __enter__(self):
        from . import device as _device

        _device.context_use(self.team.contexts[0])
        self.team._check(self.team._lib.jh_comm_group_begin())
        return self

    def __exit__(self, et, ev, tb):
        from . import device as _device

        _device.context_use(self.team.contexts[0])
        self.team._check(self.team._lib.jh_comm_group_end())
        return False


class TeamOp:
    """A tall block operator whose rows are spread over the members of a Team: member k holds `local_ops[k]` (built in
    its context).  Forward: every member's rows from its replica of m, no exchange (src/Jets.jl:1015-1031).  Adjoint:
    every member's ordered row sum range by range, the grouped all-reduce of a finished range running on the members'
    exchange streams while the next range computes (1045-1053 summed over members: tolerance parity, like any G > 1)."""

    def __init__(self, team: Team, local_ops):
        import os

        from . import jetblock as _blk

        if len(local_ops) != team.world:
            raise ValueError(f"{team.world} members, {len(local_ops)} operators")
        self.team, self.local_ops = team, list(local_ops)
        self.nchunks = builtins.max(1, int(os.environ.get("JETS_AR_CHUNKS", "4")))
        self.one_call = os.environ.get("JETS_TEAM_ONE_CALL", "1") != "0"    # the member loop in C (round 4); 0: spelled out here, call by call
        self._op_handles = None
        self._natives = []
        for A in self.local_ops:
            nat = None
            if _blk.isblockop(A):
                jt = A.jet
                nat = _blk._native_op(jt.s.get("_native"), jt.s["ops"], jt.rng.eltype())
            self._natives.append(nat)

    def domain(self):
        from .jets import domain

        return domain(self.local_ops[0])

    def ranges(self):
        from .jets import range_

        return [range_(A) for A in self.local_ops]

    def _handles(self, xs):
        import ctypes as C

        return (C.c_void_p * len(xs))(*[x.handle for x in xs])

    def _team_call(self, fn, outs: TeamVec, ins: TeamVec, *extra) -> bool:
        """The whole member loop behind ONE ABI call (jh_team_mul / jh_team_mul_adj / jh_team_normal_mul): False when a member has
        no native operator or the library says 'unsupported' before anything is enqueued (the callers then take the generic path)."""
        from ._ffi import check, JetsHipError

        if any(n is None for n in self._natives):
            return False
        if self._op_handles is None:
            self._op_handles = self._handles(self._natives)
        try:
            check(fn(self.team.world, self._op_handles, self._handles(outs.members), self._handles(ins.members), *extra))
        except JetsHipError as e:
            if e.status != 4:                                  # JH_ERR_UNSUPPORTED comes before anything is enqueued (every member alike)
                raise
            return False
        return True

    def mul_(self, d: TeamVec, m: TeamVec) -> TeamVec:
        from ._ffi import lib
        from .jets import mul_

        if self.one_call and self._team_call(lib.jh_team_mul, d, m):
            return d
        for k, _ in self.team.each():
            mul_(d[k], self.local_ops[k], m[k])
        return d

    def _ranged(self, m: TeamVec, enqueue_range) -> bool:
        """For every range of the domain: every member's kernel for it, then the members' all-reduces of it in one group."""
        from ._ffi import lib, check

        if any(n is None for n in self._natives):
            return False
        for lo, cnt in _chunk_bounds(m[0].length(), self.nchunks):
            for k, _ in self.team.each():
                enqueue_range(k, lo, cnt)
            with self.team.group():
                for k in builtins.range(self.team.world):
                    check(lib.jh_comm_allreduce_sum_range(m[k].handle, lo, cnt))
        for _ in self.team.each():
            check(lib.jh_comm_join())
        return True

    def mul_adj_(self, m: TeamVec, d: TeamVec) -> TeamVec:
        from ._ffi import lib, check, JetsHipError
        from .jets import mul_, adjoint

        if self.one_call and self._team_call(lib.jh_team_mul_adj, m, d, self.nchunks):
            return m
        try:
            if self._ranged(m, lambda k, lo, cnt: check(lib.jh_blockop_mul_adj_range(self._natives[k].handle, m[k].handle, d[k].handle, lo, cnt))):
                return m
        except JetsHipError as e:
            if e.status != 4:                                  # JH_ERR_UNSUPPORTED comes before anything is enqueued (every member alike)
                raise
        for k, _ in self.team.each():
            mul_(m[k], adjoint(self.local_ops[k]), d[k])
        with self.team.group():
            for k in builtins.range(self.team.world):
                check(lib.jh_comm_allreduce_sum(m[k].handle))
        return m

    def normal_mul_(self, y: TeamVec, m: TeamVec, tmp: TeamVec | None = None) -> TeamVec:
        """y = (A'A) m on every member's replica: the members' fused A_k'A_k m range by range, each range summed over the team under
        the next range's kernels; forward then adjoint through `tmp` (a range-side TeamVec) for operators without the fused kernel."""
        from ._ffi import lib, check, JetsHipError

        if self.one_call and self._team_call(lib.jh_team_normal_mul, y, m, self.nchunks):
            return y
        try:
            if self._ranged(y, lambda k, lo, cnt: check(lib.jh_blockop_normal_mul_range(self._natives[k].handle, y[k].handle, m[k].handle, lo, cnt))):
                return y
        except JetsHipError as e:
            if e.status != 4:
                raise
        if tmp is None:
            raise ValueError("normal_mul_: this operator has no fused A'A; pass tmp (a range-side TeamVec)")
        return self.mul_adj_(y, self.mul_(tmp, m))

    def bidiag_step_(self, u: TeamVec, v: TeamVec, w: TeamVec, alpha: float, beta: float):
        """One Golub-Kahan step on every member (jh_blockop_bidiag_step_range per range) with the ranged exchange of w;
        returns the GLOBAL ||u||^2 -- the host adds the members' deferred accumulators -- or None without a ranged kernel."""
        import ctypes as C

        from ._ffi import lib, check, JetsHipError

        if any(n is None for n in self._natives):
            return None
        for _ in self.team.each():
            check(lib.jh_normsq_reset())
        try:
            self._ranged(w, lambda k, lo, cnt: check(lib.jh_blockop_bidiag_step_range(
                self._natives[k].handle, u[k].handle, v[k].handle, w[k].handle, float(alpha), float(beta), lo, cnt, None)))
        except JetsHipError as e:
            if e.status != 4:
                raise
            return None
        total = 0.0
        out = C.c_double(0)
        for _ in self.team.each():
            check(lib.jh_normsq_read(C.byref(out)))          # synchronises this member's stream
            total += out.value
        return total
