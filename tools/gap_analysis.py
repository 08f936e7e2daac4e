"""Idle time between kernels in a rocprofv3 --kernel-trace database (rocpd sqlite): which transitions leave the GPU waiting.
usage: python tools/gap_analysis.py gpurun_out/prof_x/x_results.db"""
import collections
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    names = [r[0] for r in rows]
    setup = [i for i, n in enumerate(names) if "k_table_step" in n or "k_normalize" in n or "k_gen_bases" in n]
    rs = rows[(max(setup) + 1) if setup else 0:]
    rs = [r for r in rs if r[1] >= rs[int(len(rs) * 0.4)][1]]          # skip warm-up
    span, busy = rs[-1][2] - rs[0][1], sum(r[2] - r[1] for r in rs)
    print(f"span {span / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms, idle {100 * (1 - busy / span):.1f} %")
    agg = collections.defaultdict(lambda: [0, 0])
    short = lambda n: n.split("<")[0].split("(")[0].split("::")[-1][:34]
    for a, b in zip(rs, rs[1:]):
        g = b[1] - a[2]
        if g > 0:
            k = (short(a[0]), short(b[0]))
            agg[k][0] += g
            agg[k][1] += 1
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"{k[0]:36s} -> {k[1]:30s} total {v[0] / 1e6:6.2f} ms  n={v[1]:4d}  avg {v[0] / v[1] / 1e3:7.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
