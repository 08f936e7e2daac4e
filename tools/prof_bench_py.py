"""cProfile of bench.py's k=20 CycleFold steps (host view: which Python-level calls the time between kernels goes to).
usage (GPU box): python tools/prof_bench_py.py"""
import cProfile, pstats, sys, os, io, argparse
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import sirius_amd as S
D = bench.Dist(argparse.Namespace(emu=False, gpus=1, dist_backend="nccl"))
pri, sup, _ = bench.build_cyclefold(S, D, 20, 24, True, 15)
for _ in range(3):
    bench.cyclefold_step(S, D, pri, sup, True)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    bench.cyclefold_step(S, D, pri, sup, True)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:7000])
