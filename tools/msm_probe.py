"""GPU probe: wall time of CommitmentKey.commit at the Sangria k=17 shapes (device-resident scalars)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402  (inputs + spot check only)
import sirius_amd as S  # noqa: E402
from tests.conftest import seeded_scalars  # noqa: E402


def dev(a):
    return torch.from_numpy(a.view(np.int64)).cuda()


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-key", type=int, default=21)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    for cid in (0, 1):
        N = 1 << args.log_key
        t = time.perf_counter()
        bases = O.make_bases(cid, 42 + cid, N)
        t1 = time.perf_counter()
        ck = S.CommitmentKey(cid, bases)
        t2 = time.perf_counter()
        print(f"curve {cid}: bases {t1 - t:.2f}s (host oracle), ck_create 2^{args.log_key} {t2 - t1:.3f}s", flush=True)
        for kind in ("uniform", "trace"):
            for n in (1 << 17, 12 << 17, min(N, 1 << 20)):
                if n > N:
                    continue
                sc = seeded_scalars(O, cid, n, 5, kind)
                d = dev(sc)
                dt = timeit(lambda: ck.commit(d), args.reps)
                line = f"  commit n={n:>8} {kind:7s}: {dt * 1e3:8.3f} ms  {n / dt / 1e6:8.2f} Mscalars/s  {96 * n / dt / 1e9:7.2f} GB/s(alg)"
                if args.check and n <= (1 << 17):
                    ok = np.array_equal(ck.commit(d), O.msm(cid, sc, bases[:n]))
                    line += f"  parity={ok}"
                print(line, flush=True)
            vs = [dev(seeded_scalars(O, cid, 1 << 17, 100 + i, kind)) for i in range(6)]
            dt = timeit(lambda: ck.commit_batch(vs), args.reps)
            print(f"  commit_batch 6 x 2^17 {kind:7s}: {dt * 1e3:8.3f} ms  {6 * (1 << 17) / dt / 1e6:8.2f} Mscalars/s", flush=True)
        ck.close()


if __name__ == "__main__":
    main()
