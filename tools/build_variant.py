#!/usr/bin/env python
"""Builds variants/<name>.so = the library with ONE patched source file (A/B runs through SRS_AMD_LIB, tools/ab_variants.sh).
usage: python tools/build_variant.py <name> <file.hip> <python-expression taking the source text `s` and returning the patched text>
The other objects come from the main build (sirius_amd/csrc/*.o must be current)."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sirius_amd", "csrc")
name, fname, expr = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, ROOT)
from sirius_amd.build import FLAGS, HIPCC      # noqa: E402
with tempfile.TemporaryDirectory() as snap:
    d = os.path.join(snap, "sirius_amd", "csrc")
    os.makedirs(d)
    os.makedirs(os.path.join(snap, "include"))
    for f in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cuh")) + glob.glob(os.path.join(CSRC, "*.inc")):
        shutil.copy2(f, d)
    for f in glob.glob(os.path.join(ROOT, "include", "*.h")):
        shutil.copy2(f, os.path.join(snap, "include"))
    p = os.path.join(d, fname)
    s = open(p).read()
    t = eval(expr, {"s": s})
    assert t != s, "the patch did not change the source"
    open(p, "w").write(t)
    obj = os.path.join(ROOT, "variants", f"{name}_{fname[:-4]}.o")
    subprocess.check_call([HIPCC] + FLAGS + ["-c", p, "-o", obj])
    objs = [obj if os.path.basename(o) == fname[:-4] + ".o" else o for o in sorted(glob.glob(os.path.join(CSRC, "*.o")))]
    out = os.path.join(ROOT, "variants", f"{name}.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lhiprtc"])
    print(out)
