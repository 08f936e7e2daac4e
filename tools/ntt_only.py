"""2^24-point fft x3 (for PMC passes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE -- python tools/ntt_only.py)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S
from workloads import rand_fe
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
a = torch.from_numpy(rand_fe(np.random.default_rng(1), 1 << k).view(np.int64)).cuda()
for _ in range(3):
    S.fft.fft(a)
torch.cuda.synchronize()
print("done")
