#!/usr/bin/env python3
"""Per-kernel resources of the gfx950 code objects inside libjetship.so (or any object file with a .hip_fatbin section):
VGPRs, SGPRs, LDS and -- the reason this exists -- scratch (`.private_segment_fixed_size`): a kernel that spills its registers to
scratch turns a streaming kernel into one that also streams its own spills.  Pure Python (ELF + clang offload bundle + the
NT_AMDGPU_METADATA msgpack note), no GPU and no LLVM tool needed.

    python tools/kernel_resources.py [path] [--spills] [--top N]
"""
import os
import struct
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(data):
    """(name, offset, size, type) of every section of a 64-bit little-endian ELF image."""
    assert data[:4] == b"\x7fELF" and data[4] == 2 and data[5] == 1, "not a 64-bit little-endian ELF"
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        secs.append((name, off, size, typ))
    stroff = secs[shstrndx][1]
    out = []
    for name, off, size, typ in secs:
        end = data.index(b"\0", stroff + name)
        out.append((data[stroff + name:end].decode(), off, size, typ))
    return out


def _code_objects(fatbin):
    """Every gfx950 code object (an ELF image) in a .hip_fatbin section: the section is a sequence of uncompressed clang offload
    bundles, one per translation unit."""
    pos = 0
    while True:
        pos = fatbin.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", fatbin, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", fatbin, p)
            triple = fatbin[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield fatbin[pos + off:pos + off + size]
        pos += len(MAGIC)


def _metadata(code_object):
    for name, off, size, typ in _elf_sections(code_object):
        if typ != 7:                                          # SHT_NOTE
            continue
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", code_object, p)
            p += 12
            nm = code_object[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = code_object[p:p + descsz]
            p += (descsz + 3) & ~3
            if nm == b"AMDGPU" and ntype == 32:               # NT_AMDGPU_METADATA
                return msgpack.unpackb(desc, raw=False, strict_map_key=False)
    return None


def kernels(path=None):
    """One dict per kernel: name, vgpr, sgpr, lds, scratch (bytes per lane), max_wg."""
    path = path or os.path.join(ROOT, "jets.jl_amd", "libjetship.so")
    data = open(path, "rb").read()
    fat = [s for s in _elf_sections(data) if s[0] == ".hip_fatbin"]
    if not fat:
        raise RuntimeError(f"{path}: no .hip_fatbin section")
    out = []
    for _name, off, size, _typ in fat:
        for co in _code_objects(data[off:off + size]):
            md = _metadata(co)
            if not md:
                continue
            for k in md.get("amdhsa.kernels", []):
                out.append({"name": k[".name"], "vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count", 0),
                            "lds": k.get(".group_segment_fixed_size", 0), "scratch": k.get(".private_segment_fixed_size", 0),
                            "vgpr_spills": k.get(".vgpr_spill_count", 0), "sgpr_spills": k.get(".sgpr_spill_count", 0),
                            "max_wg": k.get(".max_flat_workgroup_size", 0)})
    return out


def demangle(names):
    import subprocess
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.splitlines()
    except Exception:
        return list(names)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ks = kernels(args[0] if args else None)
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else None
    bad = [k for k in ks if k["scratch"] or k["vgpr_spills"] or k["sgpr_spills"]]
    print(f"{len(ks)} kernels, {len(bad)} with scratch / spills")
    show = bad if "--spills" in sys.argv else sorted(ks, key=lambda k: -k["vgpr"])[:top or 20]
    for k, nm in zip(show, demangle([k["name"] for k in show])):
        print(f"  scratch {k['scratch']:4d} B  vgpr {k['vgpr']:3d}  sgpr {k['sgpr']:3d}  lds {k['lds']:6d}  wg {k['max_wg']:4d}  {nm[:150]}")
    sys.exit(1 if bad and "--spills" in sys.argv else 0)
