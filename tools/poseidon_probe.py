#!/usr/bin/env python
"""Host vs device random oracle, measured (VERDICT r02 item 9 / DESIGN.md 4.8): the ProtoGalaxy transcript of one prove at the Poseidon
configurations absorbs 4 + t betas, 32 coefficients of F and 256 of K and squeezes three challenges.  Times the host sponge of the
library (what the proves use) and the one-wavefront device sponge on the same buffers.  usage: python tools/poseidon_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S                                    # noqa: E402
from sirius_amd.field import MODULUS, ints_to_mont        # noqa: E402
import random                                             # noqa: E402

rnd = random.Random(5)
for n in (25, 57, 313):          # after delta (4 + 21), after F (+ 32), after K (+ 256): the three squeezes of ProtoGalaxy::prove at k = 20
    v = ints_to_mont(0, [rnd.randrange(MODULUS[0]) for _ in range(n)])
    perms = n // 4 + 1
    h = S.PoseidonHash(0, 5, 4, 10, 10)
    h.absorb_field(v)
    host = []
    for _ in range(20):
        h.reset(); h.absorb_field(v)
        t0 = time.perf_counter(); a = h.squeeze(255, 0); host.append(time.perf_counter() - t0)
    dev_wall, dev_kernel = [], []
    for _ in range(6):
        t0 = time.perf_counter(); b, ms = h.squeeze_device(255, 0); dev_wall.append(time.perf_counter() - t0); dev_kernel.append(ms)
    assert np.array_equal(a, b)
    print(f"{n:4d} elements = {perms:3d} permutations: host sponge {min(host) * 1e6:8.1f} us ({min(host) * 1e6 / perms:5.2f} us / permutation)   "
          f"device sponge kernel {min(dev_kernel) * 1e3:8.1f} us ({min(dev_kernel) * 1e3 / perms:6.2f} us / permutation), call wall {min(dev_wall) * 1e6:8.1f} us")
