#!/usr/bin/env python
"""Builds variants/<name>.so = the library with patched source files (A/B runs through SRS_AMD_LIB, tools/ab_step.sh).
usage: python tools/build_variant.py <name> <file> <python-expression taking the source text `s` and returning the patched text> [<file> <expr> ...]
       an expression of the form  @path  replaces the file with the contents of `path`.
A patched .hip file is recompiled; a patched header / .inc recompiles every .hip that includes it.  The other objects come from the
main build (sirius_amd/csrc/*.o must be current)."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sirius_amd", "csrc")
name, pairs = sys.argv[1], list(zip(sys.argv[2::2], sys.argv[3::2]))
sys.path.insert(0, ROOT)
from sirius_amd.build import FLAGS, HIPCC      # noqa: E402
os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
with tempfile.TemporaryDirectory() as snap:
    d = os.path.join(snap, "sirius_amd", "csrc")
    os.makedirs(d)
    os.makedirs(os.path.join(snap, "include"))
    for f in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cuh")) + glob.glob(os.path.join(CSRC, "*.inc")):
        shutil.copy2(f, d)
    for f in glob.glob(os.path.join(ROOT, "include", "*.h")):
        shutil.copy2(f, os.path.join(snap, "include"))
    rebuild = set()
    for fname, expr in pairs:
        p = os.path.join(d, fname)
        s = open(p).read()
        t = open(expr[1:]).read() if expr.startswith("@") else eval(expr, {"s": s})
        assert t != s, f"the patch did not change {fname}"
        open(p, "w").write(t)
        if fname.endswith(".hip"):
            rebuild.add(fname)
        else:                         # every .hip that includes it, directly or through a header that does
            stack, seen = [fname], set()
            while stack:
                h = stack.pop()
                for g in glob.glob(os.path.join(d, "*")):
                    b = os.path.basename(g)
                    if b in seen or not re.search(r'#include\s+"' + re.escape(h) + '"', open(g, errors="replace").read()):
                        continue
                    seen.add(b)
                    (rebuild.add if b.endswith(".hip") else stack.append)(b)
    objs = {os.path.basename(o): o for o in sorted(glob.glob(os.path.join(CSRC, "*.o")))}
    for fname in sorted(rebuild):
        obj = os.path.join(ROOT, "variants", f"{name}_{fname[:-4]}.o")
        subprocess.check_call([HIPCC] + FLAGS + ["-c", os.path.join(d, fname), "-o", obj])
        objs[fname[:-4] + ".o"] = obj
    out = os.path.join(ROOT, "variants", f"{name}.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + list(objs.values()) + ["-lhiprtc"])
    print(out, "recompiled:", sorted(rebuild))
