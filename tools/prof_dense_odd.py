#!/usr/bin/env python3
"""For tools/prof_any.sh: the dense shapes that sat at 40-56 % of the roofline in round 5's bench text, under rocprofv3 with counters.
    python tools/prof_dense_odd.py CASE      CASE in blockdiag4095 | blockdiag1023 | wide255 | grid128 | grid256
    TAG=dense_odd_<case> REGEX='k_gemv|k_block|k_fold' bash tools/prof_any.sh tools/prof_dense_odd.py <case>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

case = sys.argv[1] if len(sys.argv) > 1 else "blockdiag4095"
J.init(0)
for kv in os.environ.get("JETS_TUNE", "").split(","):
    if "=" in kv:
        J.tune(**{kv.split("=")[0]: int(kv.split("=")[1])})


def dense(k, seed):
    return J.JopDense(J.rand(J.JetSpace("float32", k, k), seed=7, stream=seed))


if case.startswith("blockdiag"):
    k = int(case[9:])
    M = 8 if k > 2048 else 64
    spc = J.JetSpace("float32", k)
    A = J.blockop([[dense(k, i) if i == j else J.JopZeroBlock(spc, spc) for j in range(M)] for i in range(M)])
    nd = M
elif case.startswith("wide"):
    k = int(case[4:])
    nd = 2064
    A = J.blockop([[dense(k, j) for j in range(nd)]])
else:
    k = int(case[4:])
    M = 64 if k == 128 else 32
    A = J.blockop([[dense(k, i * M + j) for j in range(M)] for i in range(M)])
    nd = M * M
m = J.rand(J.domain(A), seed=2, stream=0)
d = J.zeros(J.range(A))
mt = J.zeros(J.domain(A))
by = nd * k * k * 4
print(f"ALGO k_gemv {by}")


def timed(fn, reps=12):
    fn(); fn(); J.synchronize()
    e0 = J.Event().record()
    for _ in range(reps):
        fn()
    e1 = J.Event().record()
    return e0.elapsed_ms(e1) / reps


tf = timed(lambda: J.mul_(d, A, m))
ta = timed(lambda: J.mul_(mt, A.H, d))
print(f"{case}: {nd} dense children of {k}^2 Float32 ({by / 2**20:.0f} MiB): forward {tf:7.3f} ms {by / tf / 1e6:6.0f} GB/s ({J.tune_get('last_launches')} launches) | "
      f"adjoint {ta:7.3f} ms {by / ta / 1e6:6.0f} GB/s", flush=True)
