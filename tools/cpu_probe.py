"""How many host threads does the oracle (CPU port) actually scale to on this box?  Prints cgroup / affinity limits and the
time of one k=15 secondary cross-term evaluation + one 2^17 MSM for several OpenMP thread counts."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
from oracle import expr as OE, pyref as P  # noqa: E402
from workloads import make_structure_inputs  # noqa: E402

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|^CPU\\(s\\)'")
w = make_structure_inputs("secondary", 15, seed=1)
og = [OE.main_gate_expression(5, 0, 0, 0, w["num_fixed"])]
cg = OE.CompressedGates.new(og, OE.QueryIndexContext(0, w["num_fixed"], w["num_advice"], 0, 0))
progs = [OE.GraphEvaluator(t, P.MODULI[w["field"]]).export(w["field"], O) for t in cg.grouped().iter_from_first() if t is not None]
ch = np.concatenate([w["u1_challenges"].reshape(-1, 4), w["u1_u"].reshape(1, 4), w["u2_challenges"].reshape(-1, 4), O.ints_to_mont(w["field"], [1])])
bases = O.make_bases(1, 3, 1 << 17)
sc = w["W1"][: 1 << 17]
for th in (1, 8, 16, 32, 64, 128, 256):
    if th > 2 * (os.cpu_count() or 1):
        break
    t0 = time.perf_counter()
    for pr in progs:
        O.eval_program(w["field"], pr, [], w["fixed"], w["W1"], w["W2"], ch, th)
    t1 = time.perf_counter()
    O.msm(1, sc, bases, th)
    t2 = time.perf_counter()
    print(f"threads {th:4d}: cross terms k=15 {t1 - t0:7.3f} s   msm 2^17 {t2 - t1:7.3f} s", flush=True)
