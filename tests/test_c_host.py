"""A C host of the drop-in boundary (tests/c_host/host_pair.c): include/jetship.h is a plain C header, libjetship.so
links into a C program with gcc -- no Python, no torch in the product path.

CPU (`not gpu`): the header compiles as C (gcc -std=c11 -pedantic) and the host program links against the library.
GPU: the program runs the BASELINE path at a small size and checks forward / adjoint / fused A'A bit for bit against the
oracle, reductions within 1e-5, the dot-product test, the error codes and the getblock!/setblock! round trip.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "host_pair.c")
LIBDIR = os.path.join(ROOT, "jets.jl_amd")
ORADIR = os.path.join(ROOT, "oracle")


def _build(tmp_path):
    subprocess.check_call(["make", "-C", ORADIR, "-s"])
    exe = str(tmp_path / "host_pair")
    cmd = ["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-I" + ORADIR, SRC, "-o", exe,
           "-L" + LIBDIR, "-ljetship", "-L" + ORADIR, "-ljets_oracle", "-lm",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + ORADIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_header_is_plain_c(tmp_path):
    probe = tmp_path / "probe.c"
    probe.write_text('#include "jetship.h"\nint main(void) { jh_block_desc b; (void)b; return jh_abi_version() < 0; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", str(probe),
                           "-o", str(tmp_path / "probe.o")])


def test_c_host_links_against_the_library(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    needed = subprocess.check_output(["readelf", "-d", exe], text=True)
    assert "libjetship.so" in needed and "python" not in needed.lower() and "torch" not in needed.lower()


@pytest.mark.gpu
def test_c_host_runs_the_path_bit_exact(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("C HOST OK"), out.stdout
