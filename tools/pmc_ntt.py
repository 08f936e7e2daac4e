"""HBM traffic and SQ counters of the NTT kernels that ship (k_ntt_pass_lazy / k_ntt_last_lazy) from rocprofv3 PMC passes of
tools/ntt_only.py: one run with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, optionally one with SQ counters.
usage: python tools/pmc_ntt.py <fetch.csv> <write.csv> [<sq.csv>] [log_n] > profiles/r05_pmc_ntt.json
FETCH_SIZE is calibrated on the run's own 512 MiB device copies (MI355X_MICROARCH.md, HBM section: on gfx950 it reports half the bytes of a
wide coalesced read; "calibrate on a known byte count"): the factor is written into the output."""
import csv
import json
import sys
from collections import defaultdict


def rows(path):
    with open(path, newline="") as f:
        return sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"]))


def kind(name):
    if "k_ntt_pass" in name:
        return "pass"
    if "k_ntt_last" in name:
        return "last"
    if "elementwise" in name or "copyBuffer" in name or "copy" in name.lower():
        return "copy"
    return None


def collect(path, counter):
    """-> {"copy": [values per dispatch], "ntt": [(kind, value) in dispatch order]}"""
    per = defaultdict(float)
    names = {}
    for r in rows(path):
        if r["Counter_Name"] != counter:
            continue
        d = int(r["Dispatch_Id"])
        per[d] += float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    copy, ntt = [], []
    for d in sorted(per):
        k = kind(names[d])
        if k == "copy":
            copy.append(per[d])
        elif k:
            ntt.append((k, per[d], names[d]))
    return copy, ntt


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    sq_csv = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3].endswith(".csv") else None
    log_n = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 24
    n = 1 << log_n
    fcopy, fntt = collect(fetch_csv, "FETCH_SIZE")
    wcopy, wntt = collect(write_csv, "WRITE_SIZE")
    # the two largest copies of the run are the calibration copies (n * 32 B each way); rocprofv3 prints KB (1024 B)
    big_f = sorted(fcopy)[-2:]
    big_w = sorted(wcopy)[-2:]
    nbytes = n * 32.0
    f_fac = nbytes / (sum(big_f) / len(big_f) * 1024.0) if big_f else 2.0
    w_fac = nbytes / (sum(big_w) / len(big_w) * 1024.0) if big_w else 1.0
    out = {"workload": f"tools/ntt_only.py: 3 x fft + 3 x ifft of 2^{log_n} points (lazy 9 x 29-bit tile kernels), separate --pmc FETCH_SIZE / WRITE_SIZE passes",
           "calibration": {"kernel": f"device copy of n * 32 B = {int(nbytes)} B (reads and writes exactly that)", "fetch_factor": round(f_fac, 4),
                           "write_factor": round(w_fac, 4), "fetch_kb_raw": big_f, "write_kb_raw": big_w}}
    # 6 transforms x 3 launches in dispatch order (the two plan-building transforms before the copies are dropped: the LAST 18 launches)
    per_t = 3 if log_n > 16 else 2
    fl, wl = fntt[-6 * per_t:], wntt[-6 * per_t:]
    res = {}
    for ti, tname in enumerate(("fft", "ifft")):
        seg_f = fl[ti * 3 * per_t:(ti + 1) * 3 * per_t]
        seg_w = wl[ti * 3 * per_t:(ti + 1) * 3 * per_t]
        kern = defaultdict(lambda: {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
        for (k, v, nm) in seg_f:
            kern[k]["launches"] += 1
            kern[k]["fetch_bytes"] += v * 1024.0 * f_fac
        for (k, v, nm) in seg_w:
            kern[k]["write_bytes"] += v * 1024.0 * w_fac
        tot = sum(v["fetch_bytes"] + v["write_bytes"] for v in kern.values()) / 3.0
        res[tname] = {"hbm_bytes": tot, "algorithmic_bytes": 64.0 * n, "traffic_over_algorithmic": tot / (64.0 * n), "bytes_per_element": tot / n,
                      "kernels": {k: {"launches_per_transform": v["launches"] / 3.0, "fetch_bytes_per_launch": v["fetch_bytes"] / v["launches"],
                                      "write_bytes_per_launch": v["write_bytes"] / v["launches"]} for k, v in kern.items()}}
    out["per_transform"] = res
    if sq_csv:
        sq = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        seen = set()
        for r in rows(sq_csv):
            k = kind(r["Kernel_Name"])
            if k in ("pass", "last"):
                sq[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if (r["Dispatch_Id"], k) not in seen:
                    seen.add((r["Dispatch_Id"], k))
                    cnt[k] += 1
        out["sq_counters_per_launch"] = {k: dict({c: v / cnt[k] for c, v in sq[k].items()}, launches=cnt[k]) for k in sq}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
