"""Shader clock inside k_accum0s (variants/clk.so: an A/B build whose k_accum0s stamps clock64 / wall_clock64 per sampled workgroup) during
streamed commits of bench.py's witness under chunk schedules given in SRS_COMMIT_FRAC.  Prints per launch: threads, workgroup MHz
(mean / min / max over the sampled workgroups) and the mean workgroup lifetime.
usage: SRS_AMD_LIB=variants/clk.so [SRS_COMMIT_FRAC=...] python tools/clk_commit.py [pause_us]"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S
from sirius_amd import _lib
from workloads import trace_like
n = 12 << 20
ck = S.CommitmentKey.setup_synthetic(S.CURVE_BN256, 1 << 24, seed=3)
hb = S.HostBuffer(n)
hb.array[:] = trace_like(np.random.default_rng(1), n)
d = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
L = _lib.lib()
L.srs_dbg_clk_dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(1 << 16, dtype=np.uint64)
for _ in range(6):
    ck.commit_upload(hb.array, dev_copy=d)
torch.cuda.synchronize()
L.srs_dbg_clk_dump(None, 1)
t0 = time.perf_counter()
for _ in range(3):
    ck.commit_upload(hb.array, dev_copy=d)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
assert L.srs_dbg_clk_dump(buf.ctypes.data, 0) == 0
cnt = int(buf[0]); rec = buf[1:1 + 3 * min(cnt, 21000)].reshape(-1, 3)
tag = rec[:, 0] >> np.uint64(32)
print(f"commit {dt * 1e3:.3f} ms; {cnt} stamps")
seen = {}                                 # launches of the three commits with the same thread count merge (insertion order = launch order)
for t, c, w in zip(tag, rec[:, 1], rec[:, 2]):
    seen.setdefault(int(t), []).append((int(c), int(w)))
for t in seen:
    v = np.array(seen[t], dtype=np.float64)
    mhz = v[:, 0] / v[:, 1] * 100.0
    print(f"threads {t:8d}  workgroups sampled {len(v):4d}  MHz mean {mhz.mean():7.1f} min {mhz.min():7.1f} max {mhz.max():7.1f}  lifetime us mean {v[:, 1].mean() / 100:7.1f} max {v[:, 1].max() / 100:7.1f}")
