#!/usr/bin/env python3
"""Regenerates tools/README.md: one row per file under tools/ (and tools/micro/*.hip) with the first paragraph of its docstring / leading comment.
    python tools/make_readme.py"""
import ast
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def blurb(path):
    text = open(path, errors="replace").read()
    if path.endswith(".py"):
        try:
            doc = ast.get_docstring(ast.parse(text)) or ""
        except SyntaxError:
            doc = ""
    else:
        lines = []
        for ln in text.splitlines():
            s = ln.strip()
            if s.startswith("#!") or not s:
                if lines:
                    break
                continue
            if s.startswith("#") or s.startswith("//"):
                lines.append(s.lstrip("#/ ").strip())
            else:
                break
        doc = " ".join(lines)
    doc = doc.split("\n\n")[0]
    doc = re.sub(r"\s+", " ", doc).replace("|", "/").strip()
    return doc if len(doc) <= 260 else doc[:257] + "..."


rows = []
for name in sorted(os.listdir(HERE)):
    p = os.path.join(HERE, name)
    if os.path.isfile(p) and name.endswith((".py", ".sh")):
        rows.append((name, blurb(p)))
micro = os.path.join(HERE, "micro")
for name in sorted(os.listdir(micro)):
    if name.endswith(".hip"):
        rows.append(("micro/" + name, blurb(os.path.join(micro, name))))
out = ["# tools/ — benchmarks, experiments, fuzzers and profiling recipes", "",
       "Nothing here is on the product path. Everything runs through the C ABI (or, under `micro/`, stand-alone HIP harnesses); results that matter are "
       "committed under `profiles/` (see `profiles/README.md`). This file is generated: `python tools/make_readme.py`.", "",
       "| file | what it does |", "|---|---|"]
out += [f"| `{n}` | {b} |" for n, b in rows]
open(os.path.join(HERE, "README.md"), "w").write("\n".join(out) + "\n")
print(f"{len(rows)} rows")
