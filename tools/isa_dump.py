#!/usr/bin/env python3
"""Disassemble one kernel of libjetship.so: every code object of the fat binary through llvm-objdump, the kernel whose demangled name contains
the given text, reduced to its skeleton (loads, stores, waits, branches, lane spills) unless --full.

    python tools/isa_dump.py "k_general_tile<float, 1, 4, 1, 1, false, 4>" [--full] [--lib path]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SKELETON = re.compile(r"global_load|global_store|global_atomic|flat_|buffer_|scratch_|s_load|s_waitcnt|s_cbranch|s_branch|s_barrier|v_readlane|v_writelane|v_readfirstlane|s_endpgm|ds_")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--lib", default=os.path.join(ROOT, "jets.jl_amd", "libjetship.so"))
    a = ap.parse_args()
    data = open(a.lib, "rb").read()
    for name, off, size, _ in kr._elf_sections(data):
        if name != ".hip_fatbin":
            continue
        for co in kr._code_objects(data[off:off + size]):
            with tempfile.NamedTemporaryFile(suffix=".o") as f:
                f.write(co)
                f.flush()
                text = subprocess.run([OBJDUMP, "-d", "--demangle", f.name], capture_output=True, text=True).stdout
            on, n = False, 0
            for ln in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
                if m:
                    on = a.name in m.group(1).replace("(anonymous namespace)::", "")
                    if on:
                        print("==", m.group(1)[:200])
                        n = 0
                    continue
                if not on or not ln.strip():
                    continue
                n += 1
                body = re.sub(r"\s*//.*", "", ln).strip()
                body = re.sub(r"<.*", "", body).strip()
                if a.full or SKELETON.search(body):
                    print(f"{n:5d}  {body}")


if __name__ == "__main__":
    main()
