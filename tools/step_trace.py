"""One step of bench.py as a compact timeline, from a rocprofv3 --kernel-trace [--memory-copy-trace] CSV pair.
usage: python tools/step_trace.py <kernel_trace.csv> [<memory_copy_trace.csv>] [step_from_end=2]
A step starts at a k_pg_F_leaves launch (ProtoGalaxy::prove opens every CycleFold step).  Prints, per kernel of that step: start offset,
duration, the idle gap before it (us), plus the copies' spans; then totals per kernel name."""
import collections
import csv
import sys


def short(n):
    return n.split("<")[0].split("(")[0].split("::")[-1][:30]


def load(path, name_key):
    out = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(name_key, "")))
    return sorted(out)


def main():
    args = [a for a in sys.argv[1:] if not a.isdigit()]
    back = int(([a for a in sys.argv[1:] if a.isdigit()] or ["2"])[0])
    ks = load(args[0], "Kernel_Name")
    cps = load(args[1], "Direction") if len(args) > 1 else []
    # (a prove may launch k_pg_F_leaves more than once back to back: a step starts at the first of a run)
    starts = [i for i, k in enumerate(ks) if "k_pg_F_leaves" in k[2] and (i == 0 or "k_pg_F_leaves" not in ks[i - 1][2])]
    a, b = starts[-back - 1], starts[-back]
    t0 = ks[a][0]
    t1 = ks[b][0]
    print(f"# step of {(t1 - t0) / 1e3:.1f} us, {b - a} kernels")
    prev_end = t0
    agg = collections.defaultdict(lambda: [0.0, 0])
    busy = 0.0
    for s, e, n in ks[a:b]:
        gap = (s - prev_end) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f}  {short(n)}")
        prev_end = max(prev_end, e)
        agg[short(n)][0] += (e - s) / 1e3
        agg[short(n)][1] += 1
        busy += (e - s) / 1e3
    print(f"# kernels busy {busy:.1f} us of {(t1 - t0) / 1e3:.1f}")
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"#   {n:32s} {t:8.1f} us  x{c}")
    for s, e, d in cps:
        if t0 <= s < t1 and (e - s) > 20000:
            print(f"# copy {d:14s} {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:.1f} us)")


if __name__ == "__main__":
    main()
