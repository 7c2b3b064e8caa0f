"""Fused operator chains: JetComposite chains of any depth and JetSum terms that are chains, through ONE pass of a tall block operator.

The reference applies a composite stage by stage, right to left, each stage into a fresh `zeros(range(op_i))` (src/Jets.jl:524-540), and a
sum term by term through one temporary (630-655).  Around a tall device-native block operator A every other device-native stage is
elementwise -- `a *` (1159-1164), a diagonal on the domain, a diagonal on the range -- so maximal runs of such stages collapse into one
call of libjetship's chain kernels (include/jetship.h: jh_chain_*; jh_tall_chain.hip):

    W o A o M                 FORWARD     d_i = R(a_i .* P(m))
    M' o A' o W'              ADJOINT     m   = Q(sum_i conj(a_i) .* R(d_i))
    M' o A' o W o A o M       NORMAL      y   = Q(sum_i conj(a_i) .* R(a_i .* P(m)))        (weighted / preconditioned normal equations)

with the bits of the stage-by-stage chain.  Anything that is not device-native splits the chain: the planner fuses the runs on either side
and applies the opaque stage on its own, exactly as before.  The Julia twin of this file is the `_plan_chain` family in julia/JetsHIP.jl.
"""
from __future__ import annotations

import builtins
import ctypes as C
import itertools
from typing import Sequence

import numpy as np

from ._ffi import (lib, check, JetsHipError, scalar_flags, SCALAR_WIDE, SCALAR_COMPLEX, ChainStage, STAGE_SCALE, STAGE_DIAG, STAGE_CONJ, STAGE_ROWSUM,
                   CHAIN_FORWARD, CHAIN_ADJOINT, CHAIN_NORMAL)
from . import arrays as _arr
from . import jets as _j
from .jets import JopLn, JopNl, JopAdjoint, Jop, domain, range_, mul_, adjoint

MAX_STAGES = 4          # per side (jh_tall_chain.hip: JH_CHAIN_MAX_STAGES)
_UNSUPPORTED = 4        # JH_ERR_UNSUPPORTED
ENABLED = [True]        # tests / A-B timings: [False] sends every composite and sum down the stage-by-stage path of rounds 1-5
STATS = {"chain_calls": 0, "sum_terms_fused": 0, "bcast_calls": 0}   # how often a fused run was applied (tests assert that the fused path is the one that ran)


# ------------------------------------------------------------------------------ classification -----
class Stage:
    """One stage of a chain as the planner sees it.  kind: 'tall' (a tall native block operator or its adjoint), 'scale', 'diag'
    (coefficients in one device vector), 'rows' (a block-diagonal block operator: per-row coefficient arrays), 'identity', 'opaque'."""

    __slots__ = ("kind", "op", "R", "base", "nat", "adj", "a", "flags", "vec", "conj", "ptrs", "row_flags", "keep")

    def __init__(self, kind, op, R, **kw):
        self.kind, self.op, self.R = kind, op, R
        for k in self.__slots__[3:]:
            setattr(self, k, kw.get(k))

    def elementwise(self) -> bool:
        return self.kind in ("scale", "diag", "rows", "identity")

    def signature(self):
        if self.kind == "scale":
            return ("s", self.a, self.flags)
        if self.kind == "diag":
            return ("d", self.vec.ptr, self.vec.length(), bool(self.conj))
        if self.kind == "rows":
            return ("r", tuple(self.ptrs), bytes(self.row_flags), bool(self.conj))
        if self.kind == "tall":
            return ("t", self.nat.serial, bool(self.adj))
        return (self.kind,)


def _elem_bits(T) -> int:
    T = np.dtype(T)
    return T.itemsize // (2 if T.kind == "c" else 1)


def _blockdiag_rows(base):
    """(ptrs, row_flags, keep) when `base` is a square block operator that is block-DIAGONAL with diag / identity / zero children on its diagonal
    (JopZeroBlock everywhere else, src/Jets.jl:941-951) -- data weights written as `@blockop [W1 0; 0 W2]`; else None."""
    from . import jetblock as _b

    if not (isinstance(base, JopLn) and _b.isblockop(base)):
        return None
    ops = base.jet.s["ops"]
    n = ops.shape[0]
    if ops.shape[1] != n or n < 2:
        return None
    cell = base.jet.s.get("_native")
    memo = cell.__dict__ if cell is not None else {}
    hit = memo.get("blockdiag_rows")
    if hit is not None and hit[0] == _j.STATE_GEN[0]:
        return hit[1]
    res = None
    ptrs, flags, keep = [], bytearray(n), []
    ok = True
    for i in builtins.range(n):
        for jc in builtins.range(n):
            if i != jc and not _b.iszero(ops[i, jc]):
                ok = False
                break
        if not ok:
            break
        dsc = _b._native_desc(ops[i, i])
        if dsc is None or isinstance(ops[i, i], JopNl):
            ok = False
            break
        kind, adj, coeff, _scale = dsc
        if kind == "diag":
            ptrs.append(coeff.ptr)
            keep.append(coeff)
            flags[i] = 1 if adj else 0
        elif kind == "identity":
            ptrs.append(0)
        elif kind == "zero":
            ptrs.append(0)
            flags[i] = 2
        else:
            ok = False
            break
    if ok:
        res = (ptrs, bytes(flags), keep)
    memo["blockdiag_rows"] = (_j.STATE_GEN[0], res)
    return res


def classify(op: Jop, R) -> Stage:
    """What stage is `op` (already a JopLn or a JopAdjoint, as the reference's JetComposite_df!/df'! wrap it, src/Jets.jl:531, 537)?"""
    from . import jetblock as _b

    adj = isinstance(op, JopAdjoint)
    base = op.op if adj else op
    if isinstance(base, JopNl) and not adj:                      # a nonlinear stage of JetComposite_f! (src/Jets.jl:524-528): its f!
        jn = base.jet
        if jn.f is _b.square_f:
            return Stage("square_f", op, R)                      # d .= m.^2   (test/runtests.jl:19; benchmark/benchmarks.jl:55)
        if jn.f is _b.elementwise_f:
            return Stage("expr_f", op, R, vec=jn.s["f_expr"], keep=jn.s["params"])
        return Stage("opaque", op, R)
    if not isinstance(base, JopLn):
        return Stage("opaque", op, R)
    j = base.jet
    if j.df is _b.square_df and j.df_adj is _b.square_df_adj and j.mo is not None and j.mo.length() == domain(base).length():
        return Stage("square_df", op, R, vec=j.mo, conj=adj)      # dd .= 2 .* mo .* dm  /  conj.(2 .* mo) .* dd   (test/runtests.jl:20)
    if j.df is _b.identity_df:
        return Stage("identity", op, R)
    if j.df is _j.constdiag_df and j.df_adj is _j.constdiag_df_adj:
        a = j.s["a"]
        fl = scalar_flags(a)
        if _elem_bits(domain(base).eltype()) == 8:
            fl &= ~SCALAR_WIDE
        if (fl & SCALAR_COMPLEX) or complex(a).imag != 0.0:
            return Stage("opaque", op, R)                    # a Complex scalar: the full product (src/Jets.jl:1159) -- the typed lincomb
        return Stage("scale", op, R, a=float(complex(a).real), flags=fl)      # (conj(a) == a for a real a: 1160)
    if j.df is _b.diagonal_df and j.df_adj is _b.diagonal_df_adj:
        return Stage("diag", op, R, vec=j.s["diagonal"], conj=adj)
    if j.df is _b.elementwise_df and j.df_adj is _b.elementwise_df_adj and j.s["pointed"][0]:
        return Stage("diag", op, R, vec=j.s["diagonal"], conj=adj)          # the Jacobian of an elementwise nonlinear child about its point
    nat = _b._tall_native(base)
    if nat is not None and base.jet.s["ops"].shape[0] >= 2:
        return Stage("tall", op, R, base=base, nat=nat, adj=adj)
    rows = _blockdiag_rows(base)
    if rows is not None:
        return Stage("rows", op, R, ptrs=rows[0], row_flags=rows[1], keep=rows[2], conj=adj)
    return Stage("opaque", op, R)


# ------------------------------------------------------------------------------ the device handle --
class ChainHandle:
    """Owns a jh_chain (include/jetship.h) and keeps the coefficient arrays it borrows alive."""

    _serial = itertools.count(1)

    def __init__(self, tall: Stage, ctype: int, pre: Sequence[Stage], mid: Sequence[Stage], post: Sequence[Stage]):
        self._keep = [tall.nat]
        self._h = C.c_void_p()
        nrow = tall.base.jet.s["ops"].shape[0]
        nblk = domain(tall.base).length()
        es = np.dtype(domain(tall.base).eltype()).itemsize

        def pack(stages, range_side):
            arr = (ChainStage * max(1, len(stages)))()
            for k, st in enumerate(stages):
                g = arr[k]
                if st.kind == "scale":
                    g.kind, g.flags, g.a = STAGE_SCALE, st.flags, st.a
                    continue
                g.kind, g.flags, g.a = STAGE_DIAG, (STAGE_CONJ if st.conj else 0), 0.0
                if st.kind == "diag":
                    self._keep.append(st.vec)
                    n = nrow if range_side else 1
                    want = nblk * n
                    if st.vec.length() != want:
                        raise JetsHipError(_UNSUPPORTED, f"a diagonal of {st.vec.length()} elements on a side of {want}")
                    # (one pointer per block row of a slab: numpy, not a Python loop -- a composite built anew for every application pays this per call,
                    # 0.6 us per row as a list comprehension: 21 ms at 32768 rows where the pass itself takes 0.7)
                    ptrs = np.uint64(st.vec.ptr) + np.arange(n, dtype=np.uint64) * np.uint64(nblk * es)
                    g.coeff = ptrs.ctypes.data_as(C.POINTER(C.c_void_p))
                    self._keep.append(ptrs)
                else:                                        # 'rows': the diagonal of a block-diagonal block operator (range side only)
                    if not range_side or len(st.ptrs) != nrow:
                        raise JetsHipError(_UNSUPPORTED, "a block-diagonal operator that does not match the tall operator's rows")
                    ptrs = (C.c_void_p * nrow)(*[p or None for p in st.ptrs])
                    fl = (C.c_uint8 * nrow)(*st.row_flags)
                    g.coeff = C.cast(ptrs, C.POINTER(C.c_void_p))
                    g.row_flags = C.cast(fl, C.POINTER(C.c_uint8))
                    g.flags |= STAGE_ROWSUM                  # a block operator of several columns accumulates its rows into zeros (src/Jets.jl:1024, 1049): 0 + c .* x
                    self._keep += [ptrs, fl, st.keep]
            return arr

        a_pre, a_mid, a_post = pack(pre, False), pack(mid, True), pack(post, False)
        check(lib.jh_chain_create(tall.nat.handle, ctype, len(pre), a_pre, len(mid), a_mid, len(post), a_post, C.byref(self._h)))

    def apply(self, out, x, accumulate: int = 0):
        check(lib.jh_chain_apply(self._h, out.handle, x.handle, accumulate))
        return out

    def close(self):
        if self._h is not None and self._h.value:
            h, self._h = self._h, None
            lib.jh_chain_destroy(h)
        self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChainCache:
    """Plans and device handles of one combinator (it travels in the jet's state like the block jets' _NativeCell)."""

    def __init__(self):
        self.plans = {}
        self.handles = {}

    def __deepcopy__(self, memo):
        return ChainCache()

    def handle(self, key, make):
        h = self.handles.get(key)
        if h is None:
            if len(self.handles) >= 16:
                self.close_handles()
            h = self.handles[key] = make()
        return h

    def close_handles(self):
        for h in self.handles.values():
            if isinstance(h, ChainHandle):
                h.close()
        self.handles = {}
        self.plans = {}

    def close(self):
        self.close_handles()


# ------------------------------------------------------------------------------ planning -----------
def _active(stages):
    return [s for s in stages if s.kind != "identity"]            # d .= m: the same bits with or without the stage


def _segments(st: Sequence[Stage]):
    """Cut the stages (application order) into steps: ('chain', ctype, tall, pre, mid, post, first, last) for a fusable run, ('op', index) otherwise."""
    steps, i, n = [], 0, len(st)
    while i < n:
        j = i
        while j < n and st[j].elementwise():
            j += 1
        if j < n and st[j].kind == "tall":
            t = st[j]
            if not t.adj:                                     # E* A E* [A' E*]
                pre = _active(st[i:j])
                while len(pre) > MAX_STAGES:                  # the earliest stages run on their own
                    steps.append(("op", i))
                    i += 1
                    pre = _active(st[i:j])
                k = j + 1
                while k < n and st[k].elementwise() and len(_active(st[j + 1:k + 1])) <= MAX_STAGES:
                    k += 1
                mid = _active(st[j + 1:k])
                if k < n and st[k].kind == "tall" and st[k].adj and st[k].nat is t.nat:
                    l = k + 1
                    while l < n and st[l].elementwise() and len(_active(st[k + 1:l + 1])) <= MAX_STAGES:
                        l += 1
                    steps.append(("chain", CHAIN_NORMAL, t, pre, mid, _active(st[k + 1:l]), i, l - 1))
                    i = l
                elif pre or mid:
                    steps.append(("chain", CHAIN_FORWARD, t, pre, mid, [], i, k - 1))
                    i = k
                else:
                    steps.append(("op", j))
                    i = j + 1
            else:                                             # E* A' E*
                mid = _active(st[i:j])
                while len(mid) > MAX_STAGES:
                    steps.append(("op", i))
                    i += 1
                    mid = _active(st[i:j])
                l = j + 1
                while l < n and st[l].elementwise() and len(_active(st[j + 1:l + 1])) <= MAX_STAGES:
                    l += 1
                post = _active(st[j + 1:l])
                if mid or post:
                    steps.append(("chain", CHAIN_ADJOINT, t, [], mid, post, i, l - 1))
                    i = l
                else:
                    steps.append(("op", j))
                    i = j + 1
        else:                                                 # elementwise stages with no tall operator to lean on, or an opaque stage
            stop = max(j, i + 1)
            for q in builtins.range(i, stop):
                steps.append(("op", q))
            i = stop
    return steps


def _sides_ok(ctype, tall: Stage, pre, mid, post) -> bool:
    """Lengths: a diagonal before A / after A' lives on the domain, one after A / before A' on the range."""
    ndom = domain(tall.base).length()
    nrng = range_(tall.base).length()
    for s in list(pre) + list(post):
        if s.kind == "rows" or (s.kind == "diag" and s.vec.length() != ndom):
            return False
    for s in mid:
        if s.kind == "diag" and s.vec.length() != nrng:
            return False
    return True


def plan(stages: Sequence, cache: ChainCache | None, tag):
    """stages: [(op, R)] in application order -> the steps to run.  Cached per (tag, state generation, point generation)."""
    key = (tag, _j.STATE_GEN[0], _j.POINT_GEN[0], len(stages))
    if cache is not None:
        hit = cache.plans.get(key)
        if hit is not None:
            return hit
    st = [classify(op, R) for op, R in stages]
    steps = []
    for step in _segments(st):
        if step[0] == "op":
            steps.append(("op", st[step[1]]))
            continue
        _, ctype, tall, pre, mid, post, first, last = step
        if _sides_ok(ctype, tall, pre, mid, post):
            steps.append(("chain", ctype, tall, pre, mid, post, st[first:last + 1]))
        else:   # a diagonal of the wrong length for its side: not the planner's business -- the stages run one by one and raise what they raise
            steps += [("op", s) for s in st[first:last + 1]]
    steps = _merge_bcast(steps)
    if cache is not None:
        if len(cache.plans) >= 8:
            cache.plans.clear()
        cache.plans[key] = steps
    return steps


# ---- runs of elementwise stages with NO tall operator to lean on: one JIT-compiled broadcast (broadcast.py; src/Jets.jl:889-911) -------------
# The reference's own composition benchmark is such a chain: G = F o A o F o A with F: d .= m.^2 and A a diagonal on one plain space
# (benchmark/benchmarks.jl:33-38, 55-60, 73-80) -- four passes through three temporaries there, ONE pass here:
# d .= (a .* ((a .* m) .* (a .* m))) .* (a .* ((a .* m) .* (a .* m))), every operation rounded as the stages round it.
_BCAST_KINDS = ("scale", "diag", "identity", "square_f", "square_df", "expr_f")
_BCAST_MAX_CODE = 1500


def _stage_len(s: Stage) -> int:
    return s.R.length() if s.R is not None else -1


def _merge_bcast(steps):
    out, run = [], []

    def flush():
        active = [q for q in run if q.kind != "identity"]
        if len(active) >= 2:
            out.append(("bcast", list(run)))
        else:
            out.extend(("op", q) for q in run)
        run.clear()

    for step in steps:
        if step[0] == "op" and step[1].kind in _BCAST_KINDS and (not run or _stage_len(step[1]) == _stage_len(run[0])):
            run.append(step[1])
            continue
        flush()
        if step[0] == "op" and step[1].kind in _BCAST_KINDS:
            run.append(step[1])
        else:
            out.append(step)
    flush()
    return out


def _bcast_expr(run, x):
    """The run as ONE lazy elementwise expression over x (broadcast.BExpr): each stage wraps the previous one's expression."""
    import re

    from .broadcast import BExpr, bc, lazy

    e = lazy(x)
    for s in run:
        if s.kind == "identity":
            continue
        if s.kind == "scale":
            a = s.op.op.jet.s["a"] if isinstance(s.op, JopAdjoint) else s.op.jet.s["a"]
            e = BExpr.of(a) * e                                # (the scalar as given: its TYPE decides the arithmetic, broadcast._wide_mask / _real_mask)
        elif s.kind == "diag":
            c = lazy(s.vec)
            e = (bc.conj(c) if s.conj else c) * e
        elif s.kind == "square_f":
            e = e * e
        elif s.kind == "square_df":
            c = lazy(s.vec)
            c2 = c + c
            e = (bc.conj(c2) if s.conj else c2) * e
        else:                                                     # expr_f: the child's own expression over x0 and its parameters s0..
            fmt = s.vec.replace("{", "{{").replace("}", "}}")
            fmt = re.sub(r"\bx0\b", "{0}", fmt)
            for k in builtins.range(len(s.keep)):
                fmt = re.sub(r"\bs%d\b" % k, "{%d}" % (k + 1), fmt)
            e = BExpr._join("(" + fmt + ")", e, *s.keep)
        if len(e.code) > _BCAST_MAX_CODE:
            return None
    return e


def _chain_handle(cache: ChainCache | None, ctype, tall, pre, mid, post):
    """The ChainHandle of a fusable run, or None when the library declines (JH_ERR_UNSUPPORTED: more than two coefficient arrays on a side, ...)."""
    key = (ctype, tall.signature(), tuple(s.signature() for s in pre), tuple(s.signature() for s in mid), tuple(s.signature() for s in post))

    def make():
        try:
            return ChainHandle(tall, ctype, pre, mid, post)
        except JetsHipError as e:
            if e.status == _UNSUPPORTED:
                return "unsupported"
            raise

    h = cache.handle(key, make) if cache is not None else make()
    return None if h == "unsupported" else h


def has_chain(steps) -> bool:
    return any(s[0] in ("chain", "bcast") for s in steps)


def run(out, x, stages: Sequence, ws, cache: ChainCache | None, tag, accumulate: int = 0):
    """x -> stages -> out, fusing what can be fused.  Returns None when nothing in the chain fuses (the caller runs the plain chain).
    accumulate != 0 (a term of a sum): only when the WHOLE chain is one fused run -- else None."""
    if not ENABLED[0]:
        return None
    steps = plan(stages, cache, tag)
    if not has_chain(steps):
        return None
    if accumulate and (len(steps) != 1 or steps[0][0] != "chain"):
        return None
    ws = ws if ws is not None else _j._Workspace()
    try:
        cur = x
        for k, step in enumerate(steps):
            last = k == len(steps) - 1
            if step[0] == "op":
                s = step[1]
                dst = _j._zeroed_output(out, s.op) if last else ws.zeros(("chain", k), s.R, s.op)
                cur = mul_(dst, s.op, cur)
                continue
            if step[0] == "bcast":
                members = step[1]
                e = _bcast_expr(members, cur)
                if e is not None:
                    from .broadcast import assign_

                    dst = out if last else ws.zeros(("chain", k), members[-1].R, overwritten=True)
                    cur = assign_(dst, e)
                    STATS["bcast_calls"] += 1
                    continue
                for q, s in enumerate(members):                   # (an expression too long to be worth one kernel: its stages one by one)
                    fin = last and q == len(members) - 1
                    d2 = _j._zeroed_output(out, s.op) if fin else ws.zeros(("chain", k, q), s.R, s.op)
                    cur = mul_(d2, s.op, cur)
                continue
            _, ctype, tall, pre, mid, post, members = step
            h = _chain_handle(cache, ctype, tall, pre, mid, post)
            dst = out if last else ws.zeros(("chain", k), members[-1].R, overwritten=True)
            done = False
            if h is not None:
                try:
                    h.apply(dst, cur, accumulate if last else 0)
                    done = True
                    STATS["chain_calls"] += 1
                except JetsHipError as e:
                    if e.status != _UNSUPPORTED:
                        raise
                    if accumulate:
                        return None
            elif accumulate:
                return None
            if not done:                                      # the library declined this run: its stages one by one
                for q, s in enumerate(members):
                    fin = last and q == len(members) - 1
                    d2 = _j._zeroed_output(out, s.op) if fin else ws.zeros(("chain", k, q), s.R, s.op)
                    cur = mul_(d2, s.op, cur)
                continue
            cur = dst
        return out
    finally:
        ws.release()


# ------------------------------------------------------------------------------ sums ----------------
def _term_stages(op: Jop, transposed: bool):
    """The stages of ONE term of a JetSum in application order, or None when the term is not linear."""
    op = JopLn(op)
    if isinstance(op, JopAdjoint):
        return None                                           # (a sum's terms are stored un-adjointed: jops_sum wraps, src/Jets.jl:657-665)
    ops = op.jet.s["ops"] if op.jet.f is _j.JetComposite_f else (op,)
    if transposed:
        return [(adjoint(JopLn(o)), domain(JopLn(o))) for o in ops]                 # (A1 o A2)' = A2' o A1': right to left = ops in order
    return [(JopLn(o), range_(JopLn(o))) for o in reversed(ops)]


def _overwrites(op: Jop, transposed: bool) -> bool:
    """Does mul! of this (un-fused) term write every element of its output whatever the output held?  A sum reuses ONE temporary for all its terms
    (src/Jets.jl:632): a block operator with a zero block leaves that row of the temporary as the PREVIOUS term left it (1022), which a fused
    neighbour would not have written -- such sums keep the reference's loop."""
    from . import jetblock as _b

    if transposed:
        base = JopLn(op)
        if isinstance(base, JopLn) and base.jet.f in (_j.JetSum_f, _j.JetComposite_f):
            return True
        st = classify(adjoint(base), None)
        return st.kind in ("scale", "diag", "identity") or (st.kind == "tall")      # the tall adjoint zeroes m first (1042)
    return _b.overwrites_its_whole_range(JopLn(op)) or classify(JopLn(op), None).kind in ("scale", "diag", "identity")


def try_sum(out, x, ops: Sequence[Jop], sgns: Sequence[str], transposed: bool, ws, cache: ChainCache | None):
    """JetSum_df! / JetSum_df'! (src/Jets.jl:639-655) with terms that are chains: every term that is ONE fusable run adds itself to the output in
    its own last stage (jh_chain_apply(accumulate)); the other terms go through the temporary as in the reference.  None: nothing to fuse."""
    if not ENABLED[0]:
        return None
    plans = []
    for t, op in enumerate(ops):
        stages = _term_stages(op, transposed)
        if stages is None:
            return None
        steps = plan(stages, cache, ("sum", transposed, t))
        fused = len(steps) == 1 and steps[0][0] == "chain"
        if fused and steps[0][1] == CHAIN_NORMAL and not (steps[0][3] or steps[0][4] or steps[0][5]):
            fused = False      # a bare (A', A) term: the tuned fused A'A (jh_blockop_normal_mul) into the domain-sized temporary + one accumulate pass beats the
                               # general chain kernel on many rows of small blocks (4096 x 64^3: 0.62 against 1.47 ms) and equals it elsewhere
        mode = "chain" if fused else "generic"
        if not fused and len(steps) == 1 and steps[0][0] == "chain" and t == 0 and sgns[0] == _j.PLUS:
            mode = "normal_first"                             # `A'A + ...`: the fused A'A writes the output itself -- its rows are summed from +0, so it IS 0 + t (640 / 649)
        elif not fused:
            sts = [classify(o, R) for o, R in stages]
            scales = [st for st in sts if st.kind == "scale"]
            if all(st.kind in ("scale", "identity") for st in sts) and len(scales) <= 1 and all(st.flags == 0 for st in scales) and t > 0:
                mode = ("scale", scales[0].a if scales else 1.0)   # `... + lambda * I`: d = d +- T(lambda x) in ONE pass (no temporary, no second accumulate pass)
        if mode == "generic" and not _overwrites(op, transposed):
            return None
        plans.append((stages, mode, steps))
    if not any(m == "chain" or m == "normal_first" for _, m, _ in plans):
        return None
    ws = ws if ws is not None else _j._Workspace()
    try:
        started = False
        tmp = None
        for t, (op, sg) in enumerate(zip(ops, sgns)):
            stages, mode, steps = plans[t]
            sign = 1 if sg == _j.PLUS else -1
            if mode == "chain":
                r = run(out, x, stages, None, cache, ("sum", transposed, t), accumulate=(sign if started else 2 * sign))
                if r is not None:
                    started = True
                    STATS["sum_terms_fused"] += 1
                    continue
            elif mode == "normal_first":
                try:
                    steps[0][2].nat.normal_mul(out, x)
                    started = True
                    STATS["sum_terms_fused"] += 1
                    continue
                except JetsHipError as e:
                    if e.status != _UNSUPPORTED:
                        raise
            elif isinstance(mode, tuple) and started:
                # tmp .= a * x rounded in the element type, then d .= d +- tmp (634 / 643 / 652): k_lincomb rounds every product and adds left to right; 1 * d is d
                _arr.lincomb_(out, [1.0, sign * mode[1]], [out, x])
                STATS["sum_terms_fused"] += 1
                continue
            if not started:
                _arr.fill_(out, 0)                            # d .= 0 (640 / 649)
                started = True
            if tmp is None:
                R = domain(ops[0]) if transposed else range_(ops[0])
                tmp = ws.zeros("sumtmp", R)
            term = adjoint(JopLn(op)) if transposed else JopLn(op)
            _j._accumulate(sg, out, mul_(tmp, term, x))
        return out
    finally:
        ws.release()
