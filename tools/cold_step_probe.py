"""How much slower is a CycleFold step (bench.py's headline step, k = 20) that starts after the device has idled -- what an IVC driver sees
when it synthesises the next witness on the host between steps.  usage: python tools/cold_step_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gc
import torch
import bench as B
import sirius_amd as S
sys.argv = ["bench.py", "--no-extras", "--no-cpu-baseline"]
args = B.parse()
D = B.Dist(args)
pri, sup, ks = B.build_cyclefold(S, D, 20, 24, True, 15)
pri.set_witness("bench")
gc.collect(); gc.disable()
def step():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    B.cyclefold_step(S, D, pri, sup, args.ro_challenge)
    pri.settle(); sup.settle(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for _ in range(8):
    step()
warm = [step() for _ in range(10)]
print("back to back: %.2f ms (min %.2f max %.2f)" % (sum(warm) / len(warm), min(warm), max(warm)))
for idle in (0.005, 0.02, 0.05, 0.2, 1.0, 3.0):
    ts = []
    for rep in range(3):
        time.sleep(idle)
        ts.append((step(), step(), step()))
    print("after %5.3f s idle: first step %s ms, second %s, third %s" % (idle, " ".join("%.2f" % t[0] for t in ts), " ".join("%.2f" % t[1] for t in ts),
                                                                         " ".join("%.2f" % t[2] for t in ts)))
# does a short burst of device work before the step bring the clocks back?  (what INTEGRATION.md 6c suggests a shim may do)
y = torch.zeros(64 << 20, dtype=torch.int64, device="cuda")      # 512 MB: one add_ ~0.2 ms
for burst in (2, 10, 40):
    ts = []
    for rep in range(3):
        time.sleep(1.0)
        t0 = time.perf_counter()
        for _ in range(burst):
            y.add_(1)
        torch.cuda.synchronize()
        b_ms = (time.perf_counter() - t0) * 1e3
        ts.append((b_ms, step()))
    print("after 1 s idle + a burst of %2d x add_ (%s ms): step %s ms" % (burst, " ".join("%.1f" % t[0] for t in ts), " ".join("%.2f" % t[1] for t in ts)))
