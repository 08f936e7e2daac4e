"""The boundary is a C ABI: a plain-C client (tests/abi_c/smoke.c) must compile against include/sirius_amd.h with a C
compiler, link against libsirius_amd.so without any C++ / Python / torch symbol, and (on a GPU box) run."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "abi_c", "smoke.c")
LIBDIR = os.path.join(ROOT, "sirius_amd", "csrc")


def _build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", SRC, "-I", os.path.join(ROOT, "include"),
                           "-L", LIBDIR, "-lsirius_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_c_client_builds_and_links(tmp_path):
    import sirius_amd.build as B
    B.build()
    _build(tmp_path)


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C-ABI OK" in r.stdout, (r.stdout, r.stderr)
