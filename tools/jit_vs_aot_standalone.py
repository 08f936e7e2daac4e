"""Same cross-term program (MainGate<5>+<3>, k = 17) through the ahead-of-time kernel and through the hiprtc one, alone on
the device: back-to-back launches, then interleaved with a 2^17 commit on the same stream (what a fold step does)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
import sirius_amd as S
from workloads import gates_for, rand_fe

k, rows = 17, 1 << 17
gates, nfix, nadv = gates_for([5, 3])
rng = np.random.default_rng(1)
fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
W1, W2 = dev(rand_fe(rng, nadv * rows)), dev(rand_fe(rng, nadv * rows))
ck = S.CommitmentKey(S.CURVE_BN256, O.make_bases(S.CURVE_BN256, 7, rows))
sc = dev(rand_fe(rng, rows))
J = {"SRS_NO_SPEC": "1", "SRS_JIT_ALWAYS": "1"}
JI = dict(J, SRS_JIT_INLINE_MUL="1")      # the run-time compiled kernel with every multiplier inlined (= the ahead-of-time ISA)
for tag, env in (("aot", {}), ("jit", J), ("jit-inline-mul", JI), ("aot", {}), ("jit", J), ("jit-inline-mul", JI)):
    os.environ.update(env)
    St = S.PlonkStructure(0, k, [], fixed, nadv, gates)
    for e in env: os.environ.pop(e)
    nch = St.num_challenges
    u1c, u1u, u2c = rand_fe(rng, nch), rand_fe(rng, 1)[0], rand_fe(rng, nch)
    f = lambda: S.VanillaFS.commit_cross_terms(None, St, u1c, u1u, W1, u2c, W2)
    res = {}
    for mode in ("alone",):
        g = (lambda: (f(), ck.commit(sc))) if mode == "with_commit" else f
        g(); g(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): g()
        torch.cuda.synchronize()
        res[mode] = round((time.perf_counter() - t) / 20 * 1e3, 3)
    print(tag, res, flush=True)
    St.close()
