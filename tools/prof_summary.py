#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel-trace --stats run + separate --pmc FETCH_SIZE / WRITE_SIZE passes)
into profiles/: a per-kernel table (markdown), the raw kernel_stats.csv, and traffic_latest.json that
bench.py reports as roofline.traffic.

HBM traffic per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 bytes: FETCH_SIZE/WRITE_SIZE are in KiB
and on gfx950 FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced read stream
(/opt/skills/guides/MI355X_MICROARCH.md, "HBM").

    python tools/prof_summary.py --round r01 --kt gpurun_out/prof_kt --fetch gpurun_out/prof_fetch \
        --write gpurun_out/prof_write --nblocks 1024 --edge 256
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    for key in ("k_tall_diag_bidiag_chain", "k_tall_diag_bidiag", "k_tall_diag_fwd_update", "k_tall_diag_adj_update", "k_tall_diag_fwd", "k_tall_diag_adj", "k_block_fwd_general", "k_block_adj_general", "k_uniform", "k_reduce_final",
                "k_reduce", "k_lincomb", "k_hadamard", "k_fill", "k_gemv", "k_sum_partials"):
        if key in name:
            return key
    return name.split("(")[0][-48:]


def counters(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r02")
    ap.add_argument("--kt", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--nblocks", type=int, default=1024)
    ap.add_argument("--edge", type=int, default=256)
    ap.add_argument("--cmd", default="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline")
    ap.add_argument("--walk", type=int, default=None, help="grid walk of the tall forward the profiled runs were pinned to (0/1)")
    ap.add_argument("--adj-launches", type=int, default=1, help="launches per tall adjoint call (a 128 GiB adjoint goes in 2 launches of 512 rows): algorithmic bytes per LAUNCH = per call / this")
    ap.add_argument("--tag", default="", help="suffix for the output file names (rocprof_<round><tag>_summary.md)")
    ap.add_argument("--merge", action="store_true", help="merge into an existing traffic_latest.json instead of replacing it")
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    ks = glob.glob(os.path.join(args.kt, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(ks, os.path.join(out_dir, f"rocprof_{args.round}{args.tag}_kernel_stats.csv"))
    stats = list(csv.DictReader(open(ks)))
    fetch, nf = counters(args.fetch, "FETCH_SIZE") if args.fetch else ({}, {})
    write, nw = counters(args.write, "WRITE_SIZE") if args.write else ({}, {})
    n = args.edge ** 3
    algo = {"k_tall_diag_fwd": (2 * args.nblocks * n + n) * 4, "k_tall_diag_adj": (2 * args.nblocks * n + n) * 4,
            "k_tall_diag_bidiag": (3 * args.nblocks * n + 2 * n) * 4, "k_tall_diag_bidiag_chain": (3 * args.nblocks * n + 2 * n) * 4,
            "k_tall_diag_fwd_update": (3 * args.nblocks * n + n) * 4, "k_tall_diag_adj_update": (2 * args.nblocks * n + 2 * n) * 4}
    algo["k_tall_diag_adj"] //= args.adj_launches
    lines = [f"# rocprofv3 summary, round {args.round}", "",
             f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- {args.cmd}` on one MI355X (gfx950);",
             "PMC: separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes of the same program (3 timed steps).",
             f"Workload: {args.nblocks}x1 tall JopBlock, {args.edge}^3 Float32 diagonal blocks.", "",
             "| kernel | calls | avg ms | % of GPU time | algorithmic bytes/launch | achieved GB/s (algorithmic / avg) | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM traffic/launch (2*FETCH+WRITE)*1024 | traffic / algorithmic |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    traffic = {"nblocks": args.nblocks, "edge": args.edge, "round": args.round,
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (gfx950 FETCH_SIZE x2 correction)"}
    for r in stats:
        k = short(r["Name"])
        avg_ms = float(r["AverageNs"]) / 1e6
        ab = algo.get(k)
        f, w = fetch.get(k), write.get(k)
        tr = (2 * f + w) * 1024 if (f is not None and w is not None) else None
        if tr is not None and k in algo:
            key = k + (f"@walk{args.walk}" if (k == "k_tall_diag_fwd" and args.walk is not None) else "")
            traffic[key] = tr
            traffic[key + "#round"] = args.round              # every entry says which round's counters it comes from
        lines.append("| {} | {} | {:.3f} | {} | {} | {} | {} | {} | {} | {} |".format(
            k, r["Calls"], avg_ms, r["Percentage"], f"{ab:,}" if ab else "-",
            f"{ab / avg_ms / 1e6:.1f}" if ab else "-", f"{f:,.0f}" if f is not None else "-",
            f"{w:,.0f}" if w is not None else "-", f"{tr:,.0f}" if tr is not None else "-",
            f"{tr / ab:.3f}" if (tr is not None and ab) else "-"))
    lines += ["", "Full kernel names and min/max/stddev: `rocprof_%s%s_kernel_stats.csv`." % (args.round, args.tag), ""]
    open(os.path.join(out_dir, f"rocprof_{args.round}{args.tag}_summary.md"), "w").write("\n".join(lines))
    tpath = os.path.join(out_dir, "traffic_latest.json")
    if args.merge and os.path.exists(tpath):
        old = json.load(open(tpath))
        old.update(traffic)
        traffic = old
    json.dump(traffic, open(tpath, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
