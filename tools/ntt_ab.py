"""fft / ifft / coset_ifft timings at 2^k (device-resident), for A/B of the NTT kernels (SRS_NTT_MUL29=0|1 is read once per process).
usage: python tools/ntt_ab.py [k ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sirius_amd as S
from sirius_amd.workloads import rand_fe
ks = [int(x) for x in sys.argv[1:]] or [24, 22, 20, 16]
for k in ks:
    a = torch.from_numpy(rand_fe(np.random.default_rng(k), 1 << k).view(np.int64)).cuda()
    out = []
    for name in ("fft", "ifft", "coset_ifft"):
        fn = getattr(S.fft, name)
        fn(a); torch.cuda.synchronize()
        reps = 10 if k < 24 else 5
        t = time.perf_counter()
        for _ in range(reps):
            fn(a)
        torch.cuda.synchronize()
        out.append(f"{name} {(time.perf_counter() - t) / reps * 1e3:.3f} ms")
    print(f"mul29={os.environ.get('SRS_NTT_MUL29', '0')} 2^{k}: " + "  ".join(out), flush=True)
