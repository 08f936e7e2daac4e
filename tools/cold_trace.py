import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/sirius_amd") else os.environ.get("GRAFT_REPO_ROOT")
sys.path.insert(0, ROOT)
import gc, torch
import bench as B
import sirius_amd as S
sys.argv = ["bench.py", "--no-extras", "--no-cpu-baseline"]
args = B.parse(); D = B.Dist(args)
pri, sup, ks = B.build_cyclefold(S, D, 20, 24, True, 15)
pri.set_witness("bench")
gc.collect(); gc.disable()
def step():
    B.cyclefold_step(S, D, pri, sup, args.ro_challenge); pri.settle(); sup.settle(); torch.cuda.synchronize()
for _ in range(12): step()
print("=== WARM", file=sys.stderr, flush=True); step()
time.sleep(1.0)
print("=== COLD", file=sys.stderr, flush=True); step()
print("=== SECOND", file=sys.stderr, flush=True); step()
