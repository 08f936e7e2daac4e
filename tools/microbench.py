"""BASELINE config 5: 2^24-point MSM (bn256 G1) and 2^24-point NTT (Fr) on one MI355X, device-resident.
Prints one JSON line per kernel with the achieved fraction of the HBM roofline (algorithmic bytes:
MSM 96 B/scalar, NTT 64 B/element)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S  # noqa: E402
from workloads import rand_fe  # noqa: E402


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
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    n = 1 << args.log_n
    rng = np.random.default_rng(5)
    dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    # ---- NTT
    a = dev(rand_fe(rng, n))
    S.fft.fft(a)                                    # plan + twiddle tables (one-time)
    for name, fn in (("fft", S.fft.fft), ("ifft", S.fft.ifft), ("coset_fft", S.fft.coset_fft)):
        dt = timeit(lambda: fn(a), args.reps)
        print(json.dumps({"kernel": f"ntt_{name}", "log_n": args.log_n, "ms": round(dt * 1e3, 3), "elements_per_s": round(n / dt),
                          "roofline": {"bound": "hbm", "achieved": round(64 * n / dt / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(64 * n / dt / 8e12, 5)}}), flush=True)
    del a
    torch.cuda.empty_cache()
    # ---- MSM
    t0 = time.perf_counter()
    ck = S.CommitmentKey.setup_synthetic(S.CURVE_BN256, n, seed=1)
    t_key = time.perf_counter() - t0
    for kind in ("uniform", "trace"):
        sc = rand_fe(rng, n, zero_frac=0.55 if kind == "trace" else 0.0)
        d = dev(sc)
        dt = timeit(lambda: ck.commit(d), args.reps)
        print(json.dumps({"kernel": "msm_commit", "scalars": kind, "log_n": args.log_n, "ms": round(dt * 1e3, 3),
                          "scalars_per_s": round(n / dt), "key_setup_s": round(t_key, 2),
                          "roofline": {"bound": "hbm", "achieved": round(96 * n / dt / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(96 * n / dt / 8e12, 5)}}), flush=True)


if __name__ == "__main__":
    main()
