"""One large witness commit (12 * 2^k trace-like scalars, bn256) x3 -- for kernel traces of the MSM alone."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 12 << k
ck = S.CommitmentKey.setup_synthetic(S.CURVE_BN256, n, seed=3)
g = torch.Generator(device="cuda").manual_seed(1)
v = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
v[:, 3] &= (1 << 60) - 1
v[torch.rand(n, device="cuda", generator=g) < 0.55] = 0
for _ in range(4):
    ck.commit(v)
torch.cuda.synchronize()
print("done")
