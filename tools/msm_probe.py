"""Wall time of ck.commit on device-resident scalars over a range of sizes (one key of 2^24 bases); run once per setting of the
tunables (SRS_TEST_TUNING="msm_sort=2,..." of the Python mirror).  usage: python tools/msm_probe.py [log_key] [sizes...]"""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sirius_amd as S

log_key = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sizes = [int(x) for x in sys.argv[2:]] or [1 << 20, 3 << 20, 5 << 20, 12 << 20, 1 << 24]
ck = S.CommitmentKey.setup_synthetic(0, 1 << log_key, seed=3)
g = torch.Generator(device="cuda").manual_seed(1)
for n in sizes:
    for kind in ("uniform", "trace"):
        v = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
        v[:, 3] &= (1 << 60) - 1
        if kind == "trace":
            v[torch.rand(n, device="cuda", generator=g) < 0.55] = 0
        ck.commit(v); ck.commit(v)
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 5
        for _ in range(reps):
            c = ck.commit(v)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
        print(f"n={n:>9} {kind:8s} {ms:8.3f} ms  {n / ms / 1e3:8.1f} M scalars/s  tag={os.environ.get('PROBE_TAG', '')}", flush=True)
