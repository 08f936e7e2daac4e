"""Per-step times of bench.py's device-resident CycleFold step after N host-fed steps (why does secondary.device_resident read 13 ms after
--steps 10 and 9.5 ms after --steps 20?).  usage: python tools/resident_probe.py <host steps before> [resident steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
import sirius_amd as S
n_before = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_res = int(sys.argv[2]) if len(sys.argv) > 2 else 14
sys.argv = ["bench.py", "--no-extras", "--no-cpu-baseline"]
args = B.parse()
D = B.Dist(args)
pri, sup, ks = B.build_cyclefold(S, D, 20, 24, True, 15)
pri.set_witness("bench")
for _ in range(n_before):
    B.cyclefold_step(S, D, pri, sup, args.ro_challenge)
if os.environ.get("PROBE_LEGS", "1") == "1":      # the legs bench.py runs between the headline and the resident steps
    B.cyclefold_step(S, D, pri, sup, args.ro_challenge, count=True) if False else None
    pri.compat = False
    for _ in range(7):
        B.cyclefold_step(S, D, pri, sup, args.ro_challenge)
    pri.compat = True
    pri.set_witness("survey")
    for _ in range(8):
        B.cyclefold_step(S, D, pri, sup, args.ro_challenge)
    pri.set_witness("bench")
pri.settle(); sup.settle(); torch.cuda.synchronize()
pri.set_resident(D, True)
st0 = pri.ck.msm_stats()
ts = []
for i in range(n_res):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    B.cyclefold_step(S, D, pri, sup, args.ro_challenge, resident=True)
    pri.settle(); sup.settle(); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("host steps before:", n_before, "resident step ms:", " ".join(f"{t:.2f}" for t in ts))
print("msm stats delta:", {k: pri.ck.msm_stats()[k] - st0[k] for k in st0})
