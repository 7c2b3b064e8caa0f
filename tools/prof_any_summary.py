#!/usr/bin/env python3
"""Summarise the passes of tools/prof_any.sh: per kernel INSTANTIATION (full template name) the kernel-trace average, the algorithmic
bytes the tool declared ("ALGO <regex> <bytes>" lines of its output, first match wins), HBM traffic per launch from the PMC passes
((2 * FETCH_SIZE + WRITE_SIZE) * 1024 B: FETCH_SIZE is KiB and counts half of a 16 B/lane read stream on gfx950,
/opt/skills/guides/MI355X_MICROARCH.md "HBM") and the L2 hit rate.  Markdown on stdout.

    python tools/prof_any_summary.py --tag gt32 --regex k_general_tile --cmd "python3 tools/bench_grid_mixed.py 32 32 128"
"""
import argparse
import collections
import csv
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def key(full: str) -> str:
    m = re.search(r"(k_\w+(?:<[^()]*>)?)", full)
    return (m.group(1) if m else full[:100]).replace("(anonymous namespace)::", "")


def counters(dirname, names):
    agg = {n: collections.defaultdict(list) for n in names}
    for f in glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in agg:
                agg[r["Counter_Name"]][key(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {n: {k: sum(v) / len(v) for k, v in d.items()} for n, d in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--regex", default="k_")
    ap.add_argument("--cmd", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    a = ap.parse_args()
    base = os.path.join(a.out, "pa_" + a.tag)
    log = open(base + "_kt.log", errors="replace").read().splitlines() if os.path.exists(base + "_kt.log") else []
    algo = []
    for ln in log:
        if ln.startswith("ALGO "):
            _, rx, by = ln.split(None, 2)
            algo.append((re.compile(rx), float(by)))
    fe = counters(base + "_fetch", ["FETCH_SIZE"])["FETCH_SIZE"]
    wr = counters(base + "_write", ["WRITE_SIZE"])["WRITE_SIZE"]
    l2 = counters(base + "_l2", ["TCC_HIT_sum", "TCC_MISS_sum"])
    print(f"# rocprofv3 of `{a.cmd}` (one MI355X)")
    print()
    print("Passes: `rocprofv3 --kernel-trace --stats` | `--pmc FETCH_SIZE` | `--pmc WRITE_SIZE` | `--pmc TCC_HIT_sum TCC_MISS_sum`, each its own run of the same program")
    print(f"(counter passes restricted to kernels matching `{a.regex}`). Traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per launch.")
    print()
    for ln in log:
        if "GB/s" in ln or "TB/s" in ln:
            print("    " + ln.strip())
    print()
    print("| kernel | calls | avg ms | algorithmic bytes | algorithmic GB/s | frac of 8 TB/s | HBM traffic / launch | traffic / algorithmic | L2 hit rate |")
    print("|---|---|---|---|---|---|---|---|---|")
    ks = glob.glob(os.path.join(base + "_kt", "**", "*_kernel_stats.csv"), recursive=True)
    rx = re.compile(a.regex)
    for r in (csv.DictReader(open(ks[0])) if ks else []):
        if not rx.search(r["Name"]):
            continue
        k = key(r["Name"])
        ms = float(r["AverageNs"]) / 1e6
        ab = next((b for (x, b) in algo if x.search(r["Name"])), None)
        t = (2 * fe[k] + wr.get(k, 0.0)) * 1024 if k in fe else None
        h, m = l2["TCC_HIT_sum"].get(k), l2["TCC_MISS_sum"].get(k)
        print("| `{}` | {} | {:.4f} | {} | {} | {} | {} | {} | {} |".format(
            k, r["Calls"], ms, f"{ab:,.0f}" if ab else "-", f"{ab / ms / 1e6:.0f}" if ab else "-", f"{ab / ms / 1e6 / 8000:.3f}" if ab else "-",
            f"{t / 1e6:,.1f} MB" if t is not None else "-", f"{t / ab:.3f}" if (t is not None and ab) else "-",
            f"{h / (h + m):.3f}" if (h is not None and m is not None and h + m > 0) else "-"))
    print()


if __name__ == "__main__":
    main()
