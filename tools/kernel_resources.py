"""VGPRs / scratch / occupancy / LDS of every kernel of a .hip file, from hipcc's kernel-resource-usage remarks (no GPU needed).
usage: python tools/kernel_resources.py sirius_amd/csrc/rowprog.hip [filter]"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = os.path.dirname(os.path.abspath(src))
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.basename(src), "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], cwd=d, capture_output=True, text=True)
blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
K = {"VGPR": r"VGPRs", "AGPR": r"AGPRs", "scratch": r"ScratchSize \[bytes/lane\]", "occ": r"Occupancy \[waves/SIMD\]", "LDS": r"LDS Size \[bytes/block\]"}
for b in blocks:
    name = b.split(" [")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void srs::", "")
    if flt and flt not in dem:
        continue
    vals = {k: (re.search(pat + r": (\d+)", b) or [None, "?"])[1] for k, pat in K.items()}
    print(f"{dem[:78]:78s} " + " ".join(f"{k} {v:>5}" for k, v in vals.items()))
