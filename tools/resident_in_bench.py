"""bench.py's own device-resident leg with per-step times on stderr (a patched COPY of bench.py is executed; diagnosis of the 13 ms that
`secondary.device_resident` reads after --steps 10).  usage: python tools/resident_in_bench.py [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "bench.py")).read()
old = "                dt_rs = timed(D, lambda: cyclefold_step(S, D, pri, sup, args.ro_challenge, resident=True), n_rs, after=lambda: (pri.settle(), sup.settle()))\n"
new = """                import torch as _t, gc as _gc
                _gcl = []
                def _cb(phase, info, _st=[0.0]):
                    if phase == "start": _st[0] = time.perf_counter()
                    else: _gcl.append((info["generation"], (time.perf_counter() - _st[0]) * 1e3))
                _gc.callbacks.append(_cb)
                _ts = []
                for _i in range(n_rs):
                    _t.cuda.synchronize(); _t0 = time.perf_counter()
                    cyclefold_step(S, D, pri, sup, args.ro_challenge, resident=True)
                    if os.environ.get("PROBE_SETTLE", "1") == "1":
                        pri.settle(); sup.settle()
                    _t.cuda.synchronize(); _ts.append((time.perf_counter() - _t0) * 1e3)
                print("resident leg per-step ms:", " ".join("%.2f" % x for x in _ts), file=sys.stderr)
                print("resident leg gc (generation, ms):", [(g, round(m, 2)) for g, m in _gcl if m > 0.5], "counts", _gc.get_count(), "objects", len(_gc.get_objects()), file=sys.stderr)
                dt_rs = timed(D, lambda: cyclefold_step(S, D, pri, sup, args.ro_challenge, resident=True), n_rs, after=lambda: (pri.settle(), sup.settle()))
                print("resident leg timed(): %.3f ms per step" % (dt_rs / n_rs * 1e3), file=sys.stderr)
"""
assert src.count(old) == 1
src = src.replace(old, new).replace('ROOT = os.path.dirname(os.path.abspath(__file__))', 'ROOT = %r' % ROOT)
sys.argv = ["bench.py"] + sys.argv[1:]
exec(compile(src, os.path.join(ROOT, "bench.py"), "exec"), {"__name__": "__main__", "__file__": os.path.join(ROOT, "bench.py")})
