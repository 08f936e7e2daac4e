"""2^24-point transforms alone, for PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters, one pass per run):
2 x a 512 MiB device copy (the calibration: reads and writes exactly n * 32 B), then 3 x fft and 3 x ifft.  tools/pmc_ntt.py reads the CSVs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S
from workloads import rand_fe
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
a = torch.from_numpy(rand_fe(np.random.default_rng(1), 1 << k).view(np.int64)).cuda()
b = torch.empty_like(a)
S.fft.fft(a); S.fft.ifft(a)          # plans (tables) built outside the counted transforms
torch.cuda.synchronize()
for _ in range(2):
    b.copy_(a)
for _ in range(3):
    S.fft.fft(a)
for _ in range(3):
    S.fft.ifft(a)
torch.cuda.synchronize()
print("done")
