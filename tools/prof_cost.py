"""What do the HIP-event timers of the roofline cost the headline?  Runs a patched copy of bench.py with srs_profile_enable off in the
timed loop (PROF=0) or as shipped (PROF=1).  usage: PROF=0|1 python tools/prof_cost.py [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "bench.py")).read()
if os.environ.get("PROF", "1") == "0":
    old = "        S.profile_enable(True)\n        S.profile_sampling(PROF_SAMPLE)\n        S.profile_reset()\n        dt = timed("
    assert src.count(old) == 1
    src = src.replace(old, "        S.profile_enable(False)\n        S.profile_reset()\n        dt = timed(")
src = src.replace('ROOT = os.path.dirname(os.path.abspath(__file__))', 'ROOT = %r' % ROOT)
sys.argv = ["bench.py"] + sys.argv[1:]
exec(compile(src, os.path.join(ROOT, "bench.py"), "exec"), {"__name__": "__main__", "__file__": os.path.join(ROOT, "bench.py")})
