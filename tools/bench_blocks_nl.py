#!/usr/bin/env python3
"""The reference's "Block, homogeneous" (2x3 JopBar) and "Block, heterogeneous" (2x3 JopBar / JopFoo) benchmark groups
(benchmark/benchmarks.jl:88-157) at MI355X scale: nonlinear mul! (JetBlock_f!), jacobian! (block point!), mul! with J and J'.
HIP-event timed; achieved GB/s over the UNIQUE bytes (every distinct block read once -- the two block rows share their
inputs -- outputs written once plus the read of the accumulation).  Default 512^3 blocks (512 MiB each) so that nothing
stays in the 256 MiB Infinity Cache.

    python tools/bench_blocks_nl.py [EDGE] > profiles/bench_blocks_nl_r01.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
J.init(0)
n = edge ** 3
s = 4
spc = J.JetSpace(np.float32, edge, edge, edge)
PEAK = 8000.0


def timeit(fn, reps=5):
    fn(); fn()
    e0 = J.Event().record()
    for _ in range(reps):
        fn()
    e1 = J.Event().record()
    return e0.elapsed_ms(e1) / reps


def row(name, nbytes, ms):
    gbs = nbytes / ms / 1e6
    print(f"{name:58s} {ms:9.3f} ms {nbytes / 1e9:8.2f} GB {gbs:8.1f} GB/s {100 * gbs / PEAK:5.1f} % of 8 TB/s", flush=True)


for title, kinds in (("Block, homogeneous   (2x3 JopBar)", [["b", "b", "b"], ["b", "b", "b"]]),
                     ("Block, heterogeneous (2x3 JopBar JopFoo JopBar)", [["b", "f", "b"], ["b", "f", "b"]])):
    x = J.rand(spc, seed=9, stream=0)
    F = J.blockop([[J.JopSquare(spc) if k == "b" else J.JopDiagonal(x) for k in r] for r in kinds])
    m, d = J.rand(J.domain(F), seed=2, stream=0), J.rand(J.range(F), seed=3, stream=0)
    mt = J.zeros(J.domain(F))
    dm = J.rand(J.domain(F), seed=4, stream=0)
    nb = sum(k == "b" for r in kinds for k in r)
    nf = 6 - nb
    print(f"# {title}, blocks of {edge}^3 Float32 ({n * s / 2**20:.0f} MiB)")
    # f!: read d (2), the 3 input blocks (+ the diagonal of JopFoo), write d (2)
    row("mul!(d, F, m)            [JetBlock_f!, one launch]", (4 + 3 + (1 if nf else 0)) * n * s, timeit(lambda: J.mul_(d, F, m)))
    row("jacobian!(F, m)          [block point!: pointers only]", 1, timeit(lambda: J.jacobian_(F, m), reps=50))
    Jm = J.jacobian_(F, m)
    # J dm: read d (2), dm (3), one coefficient block per column (mo_j or the diagonal: 3), write d (2)
    row("mul!(d, J, dm)           [Jacobian forward]", (4 + 3 + 3) * n * s, timeit(lambda: J.mul_(d, Jm, dm)))
    # J' dd: read dd (2), one coefficient block per column (3), write the 3 columns
    row("mul!(m, J', d)           [Jacobian adjoint]", (2 + 3 + 3) * n * s, timeit(lambda: J.mul_(mt, Jm.H, d)))
    del F, Jm, m, d, mt, x, dm
