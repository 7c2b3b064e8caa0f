"""GPU checks of the single-process team (ONE process, one context per GPU) that ONE GPU can prove -- eight
contexts (streams) of this device stand in for eight GPUs:

 * the member loop of the tall operator's forward / adjoint / fused A'A runs behind ONE ABI call each (jh_team_mul / jh_team_mul_adj /
   jh_team_normal_mul; src/Jets.jl:1015-1031, 1045-1053, 530-534): the same bits as the loop spelled out call by call from the host
   language, and a forward + adjoint pair costs the host thread well under a millisecond to enqueue for 8 members (all members' launches
   come from that one thread; the projected pair at 8 GPUs is 5.7 ms of device time);
 * jh_cgls_solve_team enqueues EVERY member's first pass (the fused A'A and its inner product) before it waits for any of them: round 3
   waited for member k's scalar before member k + 1's kernel was launched, so M GPUs would have run that pass one after the other.
   Checked with event timestamps taken inside the library (knob cgls_trace): member k + 1's pass begins before member k's has finished.
"""
import gc
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M = 8


@pytest.fixture()
def team8(Jets):
    from jets_jl_amd import rowpart

    J = Jets
    home = J.context_current()[0]
    others = [J.context_create(0) for _ in range(M - 1)]
    J.context_use(home)
    team = rowpart.Team([home] + others)
    try:
        yield J, rowpart, team
    finally:
        team.close()
        gc.collect()
        J.context_use(home)
        for c in others:
            J.context_destroy(c)


def _members(J, rowpart, team, spc, rows_per_member, seed=71):
    n = spc.length()
    ops, ds, keep = [], [], []
    for k, _ in team.each():
        ck = J.rand(J.JetBSpace([spc] * rows_per_member), seed=seed, stream=0, index_base=k * rows_per_member * n)
        keep.append(ck)
        ops.append(J.blockop([[J.JopDiagonal(c)] for c in ck.arrays]))
        ds.append(J.rand(J.range(ops[-1]), seed=seed + 1, stream=0, index_base=k * rows_per_member * n))
    return ops, ds, keep


def test_team_applications_behind_one_call_match_the_spelled_out_loop_and_enqueue_fast(team8):
    J, rowpart, team = team8
    spc = J.JetSpace(np.float32, 64, 64, 64)                                 # 1 MiB blocks, 16 rows per member: 128 rows in all
    ops, ds, keep = _members(J, rowpart, team, spc, 16)
    T = team.operator(ops)
    assert T.one_call
    m = rowpart.TeamVec([J.rand(spc, seed=73, stream=0) for _ in team.each()])
    d = rowpart.TeamVec(ds)
    out = {}
    for one_call in (True, False):
        T.one_call = one_call
        mt, y = team.zeros(T.domain()), team.zeros(T.domain())
        T.mul_(d, m)
        T.mul_adj_(mt, d)
        T.normal_mul_(y, m)
        team.synchronize()
        out[one_call] = ([d[k].to_numpy().tobytes() for k in range(M)], [mt[k].to_numpy().tobytes() for k in range(M)], [y[k].to_numpy().tobytes() for k in range(M)])
    assert out[True] == out[False], "one ABI call per application and the spelled-out member loop disagree"
    assert len(set(out[True][1])) == 1 and len(set(out[True][2])) == 1, "the members' replicas differ"
    # A'A m == A'(A m): the ranged fused pass against forward-then-adjoint (same order of the members' sums: tolerance of the exchange only)
    a, b = np.frombuffer(out[True][1][0], dtype=np.float32).astype(np.float64), np.frombuffer(out[True][2][0], dtype=np.float32).astype(np.float64)
    assert np.linalg.norm(a - b) <= 1e-5 * np.linalg.norm(a)
    # host cost of enqueueing one pair for all 8 members (no synchronisation inside the bracket)
    T.one_call = True
    mt = team.zeros(T.domain())
    samples = []
    for _ in range(9):
        team.synchronize()
        t0 = time.perf_counter()
        T.mul_(d, m)
        T.mul_adj_(mt, d)
        samples.append(1e3 * (time.perf_counter() - t0))
    team.synchronize()
    med = sorted(samples)[len(samples) // 2]
    assert med < 1.0, f"enqueueing one forward + adjoint pair for {M} members took the host {med:.3f} ms (samples {[round(s, 3) for s in samples]})"
    for A in ops:
        J.close(A)


def test_cgls_team_enqueues_every_members_first_pass_before_it_waits(team8):
    J, rowpart, team = team8
    spc = J.JetSpace(np.float32, 128, 128, 128)                              # 8 MiB blocks, 24 rows per member: a pass of >= 30 us per member
    ops, ds, keep = _members(J, rowpart, team, spc, 24, seed=75)
    T = team.operator(ops)
    J.context_use(team.contexts[0])
    J.tune(cgls_trace=1)
    try:
        res = J.cgls(T, rowpart.TeamVec(ds), atol=0.0, btol=0.0, maxiter=3, overwrite_b=True, force_maxiter=True)
        J.context_use(team.contexts[0])
        overlaps = J.tune_get("last_cgls_overlaps")
    finally:
        J.context_use(team.contexts[0])
        J.tune(cgls_trace=0)
    assert res.itn == 3
    assert len({res.x[k].to_numpy().tobytes() for k in range(M)}) == 1, "the members' replicas of x differ"
    # member k + 1's pass began before member k's had finished, for (nearly) every consecutive pair: a host wait between the members'
    # launches gives 0.  (Two streams of ONE device may share a hardware queue, which orders their commands: not every pair need overlap here.)
    assert overlaps >= (M - 1) // 2, f"only {overlaps} of {M - 1} consecutive members' first passes overlap"
    for A in ops:
        J.close(A)
