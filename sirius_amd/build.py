"""Builds sirius_amd/csrc/libsirius_amd.so with hipcc for gfx950 (in-tree, no JIT cache).

Sources are snapshotted into a temporary directory before hipcc runs (hipcc reads a .hip file
twice, once per host/device pass; compiling from a snapshot makes the object immune to edits that
land mid-compile) and objects are cached by content hash, not mtime."""
import concurrent.futures as cf
import glob
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.abspath(os.path.join(HERE, "..", "include"))
OUT = os.path.join(CSRC, "libsirius_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed", "-Wno-unused-result"]


def _digest(paths):
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    with tempfile.TemporaryDirectory(prefix="srs_build_") as snap:
        # snapshot: <snap>/sirius_amd/csrc/* and <snap>/include/* (capi.hip includes ../../include/...)
        s_csrc = os.path.join(snap, "sirius_amd", "csrc")
        s_inc = os.path.join(snap, "include")
        os.makedirs(s_csrc)
        os.makedirs(s_inc)
        for f in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cuh")) + glob.glob(os.path.join(CSRC, "*.inc")):
            shutil.copy2(f, s_csrc)
        for f in glob.glob(os.path.join(INCLUDE, "*.h")):
            shutil.copy2(f, s_inc)
        srcs = sorted(glob.glob(os.path.join(s_csrc, "*.hip")))
        hdrs = glob.glob(os.path.join(s_csrc, "*.h")) + glob.glob(os.path.join(s_csrc, "*.cuh")) + glob.glob(os.path.join(s_csrc, "*.inc")) + glob.glob(os.path.join(s_inc, "*.h"))
        objs, jobs = [], []
        for s in srcs:
            name = os.path.basename(s)[:-4]
            o = os.path.join(CSRC, name + ".o")
            tag = _digest([s] + hdrs)
            objs.append(o)
            stamp = o + ".sha"
            fresh = os.path.exists(o) and os.path.exists(stamp) and open(stamp).read().strip() == tag
            if force or not fresh:
                jobs.append((s, o, tag))

        def compile_one(job):
            s, o, tag = job
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {os.path.basename(s)}:\n{r.stderr[-4000:]}")
            with open(o + ".sha", "w") as f:
                f.write(tag)
            return o

        if jobs:
            with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
                list(ex.map(compile_one, jobs))
        link_tag = _digest([o + ".sha" for o in objs])
        lstamp = OUT + ".sha"
        if jobs or force or not os.path.exists(OUT) or not os.path.exists(lstamp) or open(lstamp).read().strip() != link_tag:
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
            with open(lstamp, "w") as f:
                f.write(link_tag)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
