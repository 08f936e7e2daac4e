"""The support circuit's batched MSM alone (grumpkin, key 2^17: one trace of 3 * 2^15 scalars, 55 % zero, + two dense cross-term vectors of 2^15),
device-resident, for A/B of the small-MSM tunables (SRS_TEST_TUNING="msm_l0=3,msm_quad_max=16").
usage: python tools/small_msm_probe.py [reps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sirius_amd as S
from sirius_amd.workloads import rand_fe, trace_like
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(4)
ck = S.CommitmentKey.setup_synthetic(S.CURVE_GRUMPKIN, 1 << 17, seed=43)
dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
vs = [dev(trace_like(rng, 3 << 15)), dev(rand_fe(rng, 1 << 15)), dev(rand_fe(rng, 1 << 15))]
for _ in range(10):
    ck.commit_batch(vs)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    ck.commit_batch(vs)
torch.cuda.synchronize()
print("small batched MSM (3*2^15 + 2 x 2^15 grumpkin): %.1f us per call   env: %s" % ((time.perf_counter() - t) / reps * 1e6,
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SRS_TEST_TUNING"))), flush=True)
