"""CPU check of the BUILT library: no kernel of libjetship.so may spill registers to scratch memory.

A streaming kernel that keeps part of its tile in scratch streams its own spills as well; round 3 shipped 32 such instantiations
(all 1024-thread workgroups at the 128-VGPR cap, one of them a default forward-walk candidate of every large operator).  The
code objects inside the shared library carry per-kernel metadata (`.private_segment_fixed_size`, the bytes of scratch per lane);
tools/kernel_resources.py reads it without a GPU and without LLVM tools.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402

LIB = os.path.join(ROOT, "jets.jl_amd", "libjetship.so")


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.fail("libjetship.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    return kernel_resources.kernels(LIB)


def test_every_hot_kernel_family_is_in_the_library(kernels):
    names = " ".join(k["name"] for k in kernels)
    for family in ("k_tall_diag_fwd", "k_tall_diag_adj", "k_tall_diag_bidiag", "k_tall_diag_bidiag_chain", "k_tall_sum_fwd", "k_grid_tile",
                   "k_general_tile", "k_block_fwd_general", "k_reduce", "k_lincomb", "k_uniform", "k_gemv_rows_batched", "k_gemv_cols_fused"):
        assert family in names, f"no instantiation of {family} in {LIB}"
    assert len(kernels) > 500


def test_no_kernel_spills_to_scratch(kernels):
    bad = [k for k in kernels if k["scratch"] > 0]
    lines = [f"{k['scratch']} B/lane, {k['vgpr']} VGPRs, workgroup {k['max_wg']}: {n}" for k, n in zip(bad, kernel_resources.demangle([k["name"] for k in bad]))]
    assert not bad, "kernels with scratch (register spills or stack arrays):\n" + "\n".join(lines)


def test_sgpr_spill_census(kernels):
    """SGPRs spilled into VGPR lanes (never to scratch: test above).  Round 6 measured what they cost in the HBM-bound families -- nothing: halving
    the spills of the 16-term JetSum forward moved no figure, and the zero-spill (rolled) form of the chain kernels is 4-35 % SLOWER
    (profiles/ab_r06_sgpr_spills.md) -- so this is a census, not a ban: the families and their worst case are pinned so that a jump shows up."""
    import collections

    names = kernel_resources.demangle([k["name"] for k in kernels])
    worst = collections.defaultdict(int)
    for k, n in zip(kernels, names):
        fam = n.replace("void (anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
        worst[fam] = max(worst[fam], k["sgpr_spills"])
    spilling = {f: w for f, w in worst.items() if w}
    ceiling = {"k_chain_adj": 150, "k_chain_fwd": 24, "k_tall_sum_fwd": 100, "k_tall_sum_adj": 90, "k_tall_sum_fwd_few": 60, "k_tall_sum_adj_few": 24,
               "k_general_tile": 64, "k_grid_tile": 16, "k_tall_diag_bidiag": 110, "k_tall_diag_adj": 8, "k_tall_diag_bidiag_chain": 36, "k_tall_diag_fwd_update": 32, "k_lincomb": 104}
    unknown = sorted(set(spilling) - set(ceiling))
    assert not unknown, f"kernel families that spill SGPRs and are not in the census: {[(f, spilling[f]) for f in unknown]}"
    over = {f: (w, ceiling[f]) for f, w in spilling.items() if w > ceiling[f]}
    assert not over, f"SGPR spills above the recorded worst case (family: (now, recorded)): {over}"
    # the 9-16-term JetSum forward without Float64 scalars: 60-82 in round 5, the select masks gone in round 6
    plain = [k["sgpr_spills"] for k, n in zip(kernels, names) if "k_tall_sum_fwd<" in n and ", false, 1>" in n]
    assert plain and max(plain) <= 30, plain


def test_registers_fit_the_declared_workgroup(kernels):
    """512 VGPRs per lane and SIMD: a W-thread workgroup has W / 256 waves per SIMD, so at most 512 / (W / 256) registers per lane
    (128 at 1024 threads).  The compiler enforces it through __launch_bounds__; a kernel AT the cap is where spills start, so the
    count of those is reported to keep an eye on."""
    over = [k for k in kernels if k["max_wg"] >= 256 and k["vgpr"] + k["agpr"] > 512 // (k["max_wg"] // 256)]
    assert not over, [k["name"] for k in over]


def test_no_kernel_addresses_hbm_through_flat_instructions(tmp_path):
    """Every operand of the library's kernels lives in HBM, so every access should be a `global_*` (or scalar) instruction: a `flat_*`
    access goes through the address-space check, also waits on the LDS counter and cannot take the nontemporal / sc bits the streaming
    kernels rely on.  The compiler falls back to flat when it cannot know a pointer's address space -- pointers READ FROM THE BLOCK TABLE,
    as every dense child's matrix is: round 4's jh_dense code object held 612 flat loads / stores (17 in each of k_gemv_rows_batched,
    k_gemv_rows_wide_fused, k_gemv_rows_mixed).  Round 5 routes them through address_space(1) helpers (jh_dense.hip: ldg / ldg_nt / stg;
    jh_blockop_common.h: ld / st).  This test disassembles every gfx950 code object of the built library."""
    import re
    import shutil
    import subprocess

    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    data = open(LIB, "rb").read()
    total, per_kernel, nobj = 0, {}, 0
    for name, off, size, _typ in kernel_resources._elf_sections(data):
        if name != ".hip_fatbin":
            continue
        for co in kernel_resources._code_objects(data[off:off + size]):
            nobj += 1
            path = tmp_path / f"co{nobj}.co"
            path.write_bytes(co)
            text = subprocess.run([objdump, "-d", str(path)], capture_output=True, text=True, check=True).stdout
            assert "global_load" in text or "s_load" in text, "the disassembly is empty?"
            cur = None
            for ln in text.splitlines():
                mt = re.match(r"^[0-9a-f]+ <(\w+)>:", ln)
                if mt:
                    cur = mt.group(1)
                elif re.search(r"\bflat_(load|store|atomic)", ln):
                    total += 1
                    per_kernel[cur] = per_kernel.get(cur, 0) + 1
    assert nobj >= 5, "one code object per translation unit"
    worst = sorted(per_kernel.items(), key=lambda kv: -kv[1])[:8]
    assert total == 0, f"{total} flat accesses; worst kernels: {list(zip(kernel_resources.demangle([k for k, _ in worst]), [v for _, v in worst]))}"
