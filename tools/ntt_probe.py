#!/usr/bin/env python
"""2^k-point fft / ifft timing (BASELINE configs[4]) and parity of
the two against each other at a size the caller picks.  usage: python tools/ntt_probe.py [log_n] -> one line per transform"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S                  # noqa: E402
from workloads import rand_fe          # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log_n
a0 = torch.from_numpy(rand_fe(np.random.default_rng(5), n).view(np.int64)).cuda()
for name, fn in (("fft", S.fft.fft), ("ifft", S.fft.ifft), ("coset_fft", S.fft.coset_fft)):
    a = a0.clone()
    fn(a)
    digest = hashlib.sha256(a.cpu().numpy().tobytes()).hexdigest()[:16]
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(a); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print(f"2^{log_n} {name:9s} {dt * 1e3:7.3f} ms  {64.0 * n / dt / 1e9:7.1f} GB/s algorithmic = {64.0 * n / dt / 8e12:.4f} of 8 TB/s  sha256 {digest}")
