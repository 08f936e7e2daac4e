"""Per-step wall times of bench.py's HEADLINE loop (a patched copy of bench.py runs: every timed step is followed by a device synchronisation and
its time printed on stderr; the loop's own result is printed as well) -- looks for pauses of the harness inside the timed region.
usage: python tools/headline_steps.py [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "bench.py")).read()
old = "        dt = timed(D, lambda: cyclefold_step(S, D, pri, sup, args.ro_challenge, resident=resident), args.steps, after=lambda: (pri.settle(), sup.settle()))\n"
new = """        import torch as _t
        _ts = []
        def _one():
            _t0 = time.perf_counter()
            cyclefold_step(S, D, pri, sup, args.ro_challenge, resident=resident)
            _t.cuda.synchronize()
            _ts.append((time.perf_counter() - _t0) * 1e3)
        dt = timed(D, _one, args.steps, after=lambda: (pri.settle(), sup.settle()))
        print("headline per-step ms (with a synchronisation per step):", " ".join("%.2f" % x for x in _ts), file=sys.stderr)
"""
assert src.count(old) == 1
src = src.replace(old, new).replace('ROOT = os.path.dirname(os.path.abspath(__file__))', 'ROOT = %r' % ROOT)
if os.environ.get("COLLECT_LATE") == "1":      # A/B: the collection right before the timed region (35 ms of idle device) instead of before the warm-up
    old_call = "        gc_hold()                  # (the collector runs here, BEFORE the warm-up steps, and stays off through the timed ones)\n"
    assert src.count(old_call) == 1
    src = src.replace(old_call, "")
    src = src.replace("        gc_release()\n        if resident:\n            pri.set_resident(D, False)", "        if resident:\n            pri.set_resident(D, False)")
sys.argv = ["bench.py"] + sys.argv[1:]
exec(compile(src, os.path.join(ROOT, "bench.py"), "exec"), {"__name__": "__main__", "__file__": os.path.join(ROOT, "bench.py")})
