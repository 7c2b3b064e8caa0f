#!/usr/bin/env python3
"""IEEE special values (signed zeros, infinities, NaN, denormals, the largest finite values) through every kernel family, against the
CPU oracle, element by element -- bit for bit except for the payload of a NaN.  Prints one line per check; exit status 1 on a mismatch.

    python tools/check_specials.py [elements per block]            (one MI355X; default 4120, 6291456 for the big-block routes)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np



def specials(rng, dt, n, frac=0.3):
    rt = np.float32 if dt in (np.float32, np.complex64) else np.float64
    fi = np.finfo(rt)
    pool = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, fi.tiny / 4, -fi.tiny / 8, fi.max, -fi.max, fi.tiny, 1.0, -1.0, fi.eps], dtype=rt)

    def one():
        x = rng.standard_normal(n).astype(rt)
        k = rng.random(n) < frac
        x[k] = rng.choice(pool, size=int(k.sum()))
        return x

    if np.dtype(dt).kind != "c":
        return one()
    out = np.empty(n, dtype=dt)
    out.real, out.imag = one(), one()
    return out


def same(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.kind == "c":
        rt = np.float32 if a.dtype == np.complex64 else np.float64
        a, b = a.view(rt), b.view(rt)
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return f"NaN in different places ({int(na.sum())} vs {int(nb.sum())})"
    it = np.uint32 if a.dtype == np.float32 else np.uint64
    d = a.view(it)[~na] != b.view(it)[~nb]
    return True if not d.any() else f"{int(d.sum())} non-NaN elements differ"




def close_or_better(a, b, rtol):
    """For the one kernel family whose sum is NOT in the reference's order (the dense adjoint: fp64 lanes + wave reduction, tolerance parity):
    a NaN of the device must be a NaN of the oracle (accumulating in fp64 only removes overflow-made Inf - Inf), finite pairs agree to rtol."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.kind == "c":
        rt = np.float32 if a.dtype == np.complex64 else np.float64
        a, b = a.view(rt), b.view(rt)
    if (np.isnan(a) & ~np.isnan(b)).any():
        return f"{int((np.isnan(a) & ~np.isnan(b)).sum())} NaNs the oracle does not have"
    both = np.isfinite(a) & np.isfinite(b)
    err = np.abs(a[both].astype(np.float64) - b[both].astype(np.float64))
    scale = np.abs(b[both].astype(np.float64)) + np.finfo(a.dtype).tiny
    return True if (err <= rtol * scale + 1e-30).all() else f"finite values differ by up to {float((err / scale).max()):.1e}"


def run_checks(J, jo, seed=5, n=4096 + 24, dtypes=(np.float32, np.float64, np.complex64, np.complex128)):
    """[(dtype name, what, True | mismatch text)] for every check; J = the product package, jo = the oracle module."""
    from jets_jl_amd._ffi import lib, check

    rng = np.random.default_rng(seed)
    results = []

    def report(dt, what, r):
        results.append((np.dtype(dt).name, what, r))

    with np.errstate(all="ignore"):
        for dt in dtypes:
            nrow = 6
            z = lambda k=1: [np.zeros(n, dtype=dt) for _ in range(k)]
            coeffs = [specials(rng, dt, n) for _ in range(nrow)]
            hm, hd = specials(rng, dt, n), [specials(rng, dt, n) for _ in range(nrow)]
            A = J.blockop([[J.JopDiagonal(J.from_numpy(c))] for c in coeffs])
            ops = [[jo.Block("diag", n, coeff=c)] for c in coeffs]
            m = J.from_numpy(hm)
            ref_d = jo.block_df(ops, z(nrow), [hm])
            report(dt, "tall forward", same((A * m).to_numpy(), np.concatenate(ref_d)))
            dd = J.from_numpy(np.concatenate(hd), J.range(A))
            report(dt, "tall adjoint", same((A.H * dd).to_numpy(), jo.block_df_adj(ops, z(), hd)[0]))
            report(dt, "fused A'A", same(((A.H @ A) * m).to_numpy(), jo.block_df_adj(ops, z(), ref_d)[0]))
            # rows of every elementwise kind, some adjointed
            spc = J.JetSpace(dt, n)
            sc = complex(specials(rng, dt, 1, 0.0)[0])
            sc = sc if np.dtype(dt).kind == "c" else sc.real
            kinds = [J.JopDiagonal(J.from_numpy(coeffs[0])), J.JopIdentity(spc), J.JopZeroBlock(spc, spc), J.JopDiagonal(J.from_numpy(coeffs[1])).H,
                     J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": sc}), J.JopDiagonal(J.from_numpy(coeffs[2]))]
            okinds = [jo.Block("diag", n, coeff=coeffs[0]), jo.Block("identity", n), jo.Block("zero", n, n), jo.Block("diag", n, coeff=coeffs[1], adjoint=True),
                      jo.Block("scale", n, scale=sc), jo.Block("diag", n, coeff=coeffs[2])]
            B = J.blockop([[k] for k in kinds])
            bops = [[k] for k in okinds]
            report(dt, "mixed rows forward", same((B * m).to_numpy(), np.concatenate(jo.block_df(bops, z(nrow), [hm]))))
            report(dt, "mixed rows adjoint", same((B.H * dd).to_numpy(), jo.block_df_adj(bops, z(), hd)[0]))
            # the one-pass step and the two fused halves
            for name, op, oo in (("all-diagonal", A, ops), ("mixed rows", B, bops)):
                from jets_jl_amd import jetblock
                h = jetblock._native_op(op.jet.s["_native"], op.jet.s["ops"], op.jet.rng.eltype()).handle
                for beta in (0.0, -0.5):
                    u = J.from_numpy(np.concatenate(hd), J.range(op))
                    w = J.zeros(J.domain(op))
                    out = C.c_double(0)
                    check(lib.jh_blockop_bidiag_step(h, u.handle, m.handle, w.handle, 0.75, beta, C.byref(out)))
                    tmp = jo.block_df(oo, z(nrow), [hm])
                    ref_u = jo.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [0.75, beta] if beta else [0.75], [tmp, hd] if beta else [tmp])
                    report(dt, f"step u ({name}, beta {beta})", same(u.to_numpy(), np.concatenate(ref_u)))
                    report(dt, f"step w ({name}, beta {beta})", same(w.to_numpy(), jo.block_df_adj(oo, z(), ref_u)[0]))
            # grid with a zero block
            g = [[specials(rng, dt, n) for _ in range(3)] for _ in range(3)]
            G = J.blockop([[J.JopZeroBlock(spc, spc) if (i, j) == (1, 1) else J.JopDiagonal(J.from_numpy(g[i][j])) for j in range(3)] for i in range(3)])
            gops = [[jo.Block("zero", n, n) if (i, j) == (1, 1) else jo.Block("diag", n, coeff=g[i][j]) for j in range(3)] for i in range(3)]
            hx = [specials(rng, dt, n) for _ in range(3)]
            x = J.from_numpy(np.concatenate(hx), J.domain(G))
            report(dt, "grid forward", same((G * x).to_numpy(), np.concatenate(jo.block_df(gops, z(3), hx))))
            report(dt, "grid adjoint", same((G.H * x).to_numpy(), np.concatenate(jo.block_df_adj(gops, z(3), hx))))
            # sum of three tall operators, + - +
            A2 = J.blockop([[J.JopDiagonal(J.from_numpy(c))] for c in coeffs[::-1]])
            A3 = J.blockop([[J.JopDiagonal(J.from_numpy(c))] for c in coeffs[1:] + coeffs[:1]])
            ops2, ops3 = [[jo.Block("diag", n, coeff=c)] for c in coeffs[::-1]], [[jo.Block("diag", n, coeff=c)] for c in coeffs[1:] + coeffs[:1]]
            S = A - A2 + A3
            r1, r2, r3 = jo.block_df(ops, z(nrow), [hm]), jo.block_df(ops2, z(nrow), [hm]), jo.block_df(ops3, z(nrow), [hm])
            ref = [(a - b) + c for a, b, c in zip(r1, r2, r3)]
            ref = [(np.zeros(n, dtype=dt) + a) for a in ref]
            report(dt, "sum forward (+ - +)", same((S * m).to_numpy(), np.concatenate([((np.zeros(n, dtype=dt) + a) - b) + c for a, b, c in zip(r1, r2, r3)])))
            # dense children
            nd = 96
            Md = specials(rng, dt, nd * nd, 0.1).reshape(nd, nd, order="F")
            hv = specials(rng, dt, nd, 0.1)
            D = J.blockop([[J.JopDense(J.from_numpy(np.asfortranarray(Md)))], [J.JopDense(J.from_numpy(np.asfortranarray(Md)))]]) if hasattr(J, "JopDense") else None
            if D is not None:
                dops = [[jo.Block("dense", nd, nd, coeff=np.asfortranarray(Md))], [jo.Block("dense", nd, nd, coeff=np.asfortranarray(Md))]]
                report(dt, "dense forward", same((D * J.from_numpy(hv)).to_numpy(), np.concatenate(jo.block_df(dops, [np.zeros(nd, dtype=dt) for _ in range(2)], [hv]))))
                h2 = [specials(rng, dt, nd, 0.1) for _ in range(2)]
                report(dt, "dense adjoint", close_or_better((D.H * J.from_numpy(np.concatenate(h2), J.range(D))).to_numpy(), jo.block_df_adj(dops, [np.zeros(nd, dtype=dt)], h2)[0],
                                                            1e-4 if dt in (np.float32, np.complex64) else 1e-11))
            # broadcast: a*x + b*y over block arrays
            R = J.range(A)
            X, Y = J.from_numpy(np.concatenate(hd), R), J.from_numpy(np.concatenate(ref_d), R)
            outv = J.zeros(R)
            J.lincomb_(outv, [0.75, -1.25], [X, Y])
            report(dt, "lincomb", same(outv.to_numpy(), np.concatenate(jo.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [0.75, -1.25], [hd, ref_d]))))
            J.broadcast_(outv, "s0*x0 + s1*x1", [X, Y], [0.75, -1.25])            # the compiled form of the same expression: real scalars stay real
            report(dt, "compiled broadcast, real scalars", same(outv.to_numpy(), np.concatenate(jo.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [0.75, -1.25], [hd, ref_d]))))
            if np.dtype(dt).kind == "c":
                cs = [0.75 + 0.5j, -1.25 - 2j]
                J.lincomb_(outv, cs, [X, Y])
                ref = np.concatenate(jo.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], cs, [hd, ref_d]))
                report(dt, "lincomb, complex scalars", same(outv.to_numpy(), ref))
                J.broadcast_(outv, "s0*x0 + s1*x1", [X, Y], cs)
                report(dt, "compiled broadcast, complex scalars", same(outv.to_numpy(), ref))
    return results


if __name__ == "__main__":
    import jets_jl_amd as J
    from oracle import jets_oracle as jo

    J.init(0)
    out = run_checks(J, jo, n=int(sys.argv[1]) if len(sys.argv) > 1 else 4096 + 24)   # e.g. 6291456: blocks of 24-48 MiB take the big-block routes
    for name, what, r in out:
        print(f"{name:10s} {what:38s} {'ok' if r is True else 'MISMATCH: ' + str(r)}", flush=True)
    sys.exit(1 if any(r is not True for _, _, r in out) else 0)
