"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) result database as a table.
usage: python tools/prof_summary.py gpurun_out/prof_x/x_results.db > profiles/r01_x.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in rows:
        print(f"  {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name[:150]}")


if __name__ == "__main__":
    main(sys.argv[1])
