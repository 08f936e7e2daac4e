"""PCIe-inclusive cost of the host-pointer entry points (what a Rust shim passing `&[F]` slices pays): one k=17 fold step
with every operand in host memory -- pageable numpy arrays vs page-locked buffers -- next to the device-resident step."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import sirius_amd as S

dev = torch.device("cuda", 0)
pri = bench.Side("primary", 17, 21, 0, 1, dev); sec = bench.Side("secondary", 17, 21, 0, 1, dev)
out = {}
for mode in ("device", "host_pageable", "host_pinned"):
    for s in (pri, sec):
        w = s.w
        if mode == "device":
            s.accW, s.accE, s.inW = (torch.from_numpy(w[k].view(np.int64)).to(dev) for k in ("W1", "E", "W2"))
        elif mode == "host_pageable":
            s.accW, s.accE, s.inW = w["W1"].copy(), w["E"].copy(), w["W2"].copy()
        else:
            pin = lambda a: torch.from_numpy(a.view(np.int64)).clone().pin_memory()
            s.accW, s.accE, s.inW = pin(w["W1"]), pin(w["E"]), pin(w["W2"])
        s.inC = np.zeros(8, dtype=np.uint64)
    def step():
        for side in (sec, pri):
            terms, commits = S.VanillaFS.commit_cross_terms(side.ck, side.S, side.u1c, side.u1u, side.accW, side.u2c, side.inW)
            S.RelaxedPlonkWitness(side.field, [side.accW], side.accE).fold([side.inW], terms, side.r)     # result dropped: fixed inputs
            other = pri if side is sec else sec
            other.ck.commit(other.inW)
    step(); step(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    out[mode + "_ms_per_step"] = round((time.perf_counter() - t) / 10 * 1e3, 3)
out["bytes_staged_per_step"] = 32 * (1 << 17) * (2 * (12 + 7) + (12 + 7) + 2 + (6 + 5) * 2)
print(json.dumps(out))
