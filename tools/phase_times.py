"""Wall-clock split of the k=20 CycleFold step (bench.py's objects) by phase, with a device synchronisation between phases, and
of ProtoGalaxy::prove by stage through the step-wise calls.  Host view: where the time between kernels goes."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import sirius_amd as S  # noqa: E402
from sirius_amd import protogalaxy as PG  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--leaf-rows", default="compat")
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
D = bench.Dist(argparse.Namespace(emu=False, gpus=1, dist_backend="nccl"))
compat = a.leaf_rows == "compat"
pri, sup, _ = bench.build_cyclefold(S, D, a.k, a.k + 4, compat, 15)
for _ in range(3):
    bench.cyclefold_step(S, D, pri, sup, True)
T = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    T.setdefault(name, []).append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize()
    T.setdefault(name + " (+drain)", []).append((time.perf_counter() - t) * 1e3)
    return r


for _ in range(a.reps):
    timed("A  pri.prove", lambda: pri.prove(S, D, True))
    timed("B  sup.prove_incoming", lambda: sup.prove_incoming(S, True))
    timed("C  pri.witness_commit", lambda: pri.witness_commit(S, D))
med = lambda v: float(np.median(v))
for k, v in T.items():
    print(f"{k:34s} {med(v):8.3f} ms")
print("sum of phases", round(sum(med(v) for k, v in T.items() if k.endswith("(+drain)")), 3))
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(a.reps):
    bench.cyclefold_step(S, D, pri, sup, True)
torch.cuda.synchronize()
print("unsplit step", round((time.perf_counter() - t) / a.reps * 1e3, 3))

# ProtoGalaxy::prove by stage (step-wise calls: each returns its polynomial to the host)
pri.fold_done()
ctx, m = pri.ctx, pri.m
Tp = {}


def tp(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    Tp.setdefault(name, []).append((time.perf_counter() - t) * 1e3)
    return r


for _ in range(a.reps):
    ro = pri.ro.reset()
    def absorb0():
        ro.absorb_field(np.concatenate([pri.accC.reshape(2, 4), pri.inC.reshape(2, 4)]))
        ro.absorb_field(pri.betas)
        return ro.squeeze(128, 0)
    delta = tp("transcript: instances + betas -> delta", absorb0)
    pF = tp("compute_F", lambda: PG.compute_F(ctx, pri.betas, delta, pri.accW, reference_compat=compat))
    alpha = tp("transcript: F -> alpha", lambda: (ro.absorb_field(pF), ro.squeeze(255, 0))[1])
    bs = tp("beta_stroke", lambda: PG.beta_stroke(pri.betas, alpha, delta))
    pG = tp("compute_G", lambda: PG.compute_G(ctx, bs, [pri.accW, pri.inW], reference_compat=compat))
    pK = tp("compute_K_from_G", lambda: PG.compute_K_from_G(ctx, pG, PG.poly_eval(pF, alpha)))
    gamma = tp("transcript: K -> gamma", lambda: (ro.absorb_field(pK), ro.squeeze(255, 0))[1])
    tp("calculate_e + lagrange", lambda: (PG.calculate_e(pF, pK, gamma, alpha, ctx.lagrange_domain),
                                          PG.eval_lagrange_poly_for_cyclic_group(gamma, ctx.lagrange_domain)))
tot = 0
for k, v in Tp.items():
    print(f"  {k:40s} {med(v):8.3f} ms")
    tot += med(v)
print("  sum", round(tot, 3), " poly_F", len(pF), "poly_K", len(pK), "betas", len(pri.betas))
