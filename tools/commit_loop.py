"""Streamed commits of bench.py's 12 * 2^20 witness back to back for <seconds> (power / clock sampling from outside: tools/power_commit.sh).
usage: [SRS_AMD_LIB=variants/fracenv.so SRS_COMMIT_FRAC=...] python tools/commit_loop.py <seconds>"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S
from workloads import trace_like
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = 12 << 20
ck = S.CommitmentKey.setup_synthetic(S.CURVE_BN256, 1 << 24, seed=3)
hb = S.HostBuffer(n)
hb.array[:] = trace_like(np.random.default_rng(1), n)
d = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
for _ in range(4):
    ck.commit_upload(hb.array, dev_copy=d)
torch.cuda.synchronize()
print("loop start", flush=True)
t0 = time.perf_counter(); c = 0
while time.perf_counter() - t0 < secs:
    ck.commit_upload(hb.array, dev_copy=d); c += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{c} commits, {dt / c * 1e3:.3f} ms per commit", flush=True)
