#!/usr/bin/env python3
"""Condense the output of tools/bench_grid_sparse.py: per grid and pattern the best forward | adjoint GB/s of every route.   python tools/sparse_summary.py FILE"""
import re
import sys
from collections import OrderedDict

rows = OrderedDict()
notes = {}
for ln in open(sys.argv[1]):
    m = re.match(r"(\d+ x \d+) (\w+)\s+of (\d+)\^3 .*general_list=(\d) bits (..): forward\s+[\d.]+ ms\s+([\d.]+) GB/s \| adjoint\s+[\d.]+ ms\s+([\d.]+) GB/s(.*)", ln)
    if not m:
        continue
    key = (m.group(1), m.group(3), m.group(2))
    r = rows.setdefault(key, {})
    f, a = float(m.group(6)), float(m.group(7))
    o = r.get(m.group(4), (0.0, 0.0))
    r[m.group(4)] = (max(o[0], f), max(o[1], a))
    if m.group(5) == "!=":
        notes[key] = "BITS DIFFER"
    if m.group(8).strip():
        notes[key] = m.group(8).strip()
print(f"{'grid':22s} " + " ".join(f"{'list=' + k:>13s}" for k in "0231") + "  note")
for key, r in rows.items():
    print(f"{key[0] + ' of ' + key[1] + '^3 ' + key[2]:22s} " + " ".join((f"{r[k][0]:6.0f}|{r[k][1]:6.0f}" if k in r else " " * 13) for k in "0231") + "  " + notes.get(key, ""))
