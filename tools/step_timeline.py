"""Wall-clock split of one k=17 fold step by call (host view), to spot host-side overheads."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import sirius_amd as S

class A: k = 17; log_key = 21
dev = torch.device("cuda", 0)
pri = bench.Side("primary", 17, 21, 0, 1, dev); sec = bench.Side("secondary", 17, 21, 0, 1, dev)
for s in (pri, sec): s.inC = np.zeros(8, dtype=np.uint64)
bench.witness_commit(S, pri, None, 1, dev); bench.witness_commit(S, sec, None, 1, dev)
for _ in range(3): bench.fold_step(S, pri, sec, None, 1, dev)
T = {}
def timed(name, fn, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
    T[name] = T.get(name, 0) + (time.perf_counter() - t) * 1e3; return r
N = 10
for _ in range(N):
    for side, tag in ((sec, "sec"), (pri, "pri")):
        terms, commits = timed(f"{tag}.commit_cross_terms", S.VanillaFS.commit_cross_terms, side.ck, side.S, side.u1c, side.u1u, side.accW, side.u2c, side.inW)
        timed(f"{tag}.lincomb_W", S.point_lincomb, side.curve, side.accCW, side.inC.reshape(1, 8), side.r.reshape(1, 4))
        timed(f"{tag}.lincomb_E", S.point_lincomb, side.curve, side.accCE, commits, side.rpows)
        acc = timed(f"{tag}.fold", lambda: S.RelaxedPlonkWitness(side.field, [side.accW], side.accE).fold([side.inW], terms, side.r))
        side.accW, side.accE = acc.W[0], acc.E
        other = pri if side is sec else sec
        timed(f"{other is pri and 'pri' or 'sec'}.witness_commit", other.ck.commit, other.inW)
tot = 0
for k, v in T.items():
    print(f"{k:28s} {v / N:8.3f} ms"); tot += v / N
print(f"{'sum (serialised)':28s} {tot:8.3f} ms")
S.profile_enable(True); S.profile_reset()
for _ in range(N): bench.fold_step(S, pri, sec, None, 1, dev)
for name in ("msm_accum0", "rowprog_cross_terms"):
    p = S.profile_get(name); print(name, round(p["total_ms"] / N, 3), "ms/step", p["launches"] // N, "launches/step")
