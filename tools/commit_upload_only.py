"""One streamed witness commit (12 * 2^k scalars from page-locked host memory, 55 % zero / 45 % uniform -- bench.py's witness; bn256) x4:
the chunked commit in slot mode alone, for kernel traces and PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S
from workloads import trace_like
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 12 << k
ck = S.CommitmentKey.setup_synthetic(S.CURVE_BN256, n, seed=3)
hb = S.HostBuffer(n)
hb.array[:] = trace_like(np.random.default_rng(1), n)
d = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
for _ in range(4):
    ck.commit_upload(hb.array, dev_copy=d)
torch.cuda.synchronize()
print("done", ck.msm_stats())
