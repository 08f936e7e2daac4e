"""Generates tests/golden/*.json from the pure-Python big-int restatement (oracle/pyref.py).

The reference is Rust and cannot be built or imported here (no cargo, no network),
so the vectors are produced by an independent big-int implementation that is itself pinned by
the reference's known-answer tests (fft_simple_input_test, basic_lagrange_test, [r-1]G = -G);
those KAT constants are stored verbatim in kat.json.  Run:  python tests/golden/gen_golden.py
All integers are canonical (non-Montgomery) values serialised as hex strings.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyref as P  # noqa: E402

H = lambda v: hex(v)


def main():
    P.self_check()
    rnd = random.Random(0x5349524955530000)
    # ---- KATs copied from the reference's own tests
    kat = {
        "fft_simple_input_test": {"source": "src/fft.rs:241-260", "input": list(range(8)),
                                  "output": [str(v) for v in P.FFT_KAT]},
        "basic_lagrange_test": {"source": "src/polynomial/lagrange.rs:116-127", "X": 2, "log_n": 2,
                                "output": [str(v) for v in P.LAGRANGE_KAT]},
        "digest_consistency": {"source": "src/digest.rs:100-114", "scalar": str(P.FR - 1),
                               "expect": "-G on bn256"},
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

    # ---- field vectors
    field = {}
    for name, p in (("fr", P.FR), ("fq", P.FQ)):
        a = [0, 1, p - 1, P.R256 % p, (P.R256 * P.R256) % p, 2, (1 << 128) - 1] + [rnd.randrange(p) for _ in range(24)]
        b = [p - 1, p - 1, p - 1, 1, 7, 0, (1 << 128) - 1] + [rnd.randrange(p) for _ in range(24)]
        field[name] = {
            "p": H(p), "a": [H(x) for x in a], "b": [H(x) for x in b],
            "mul": [H(x * y % p) for x, y in zip(a, b)],
            "add": [H((x + y) % p) for x, y in zip(a, b)],
            "sub": [H((x - y) % p) for x, y in zip(a, b)],
            "inv_a": [H(pow(x, p - 2, p)) for x in a],
            "mont_a": [H(x * P.R256 % p) for x in a],
        }
    json.dump(field, open(os.path.join(HERE, "field.json"), "w"), indent=1)

    # ---- curve + MSM vectors (bases = small multiples of G so the fixture stays tiny)
    curves = {}
    for name, cv in (("bn256", P.BN256), ("grumpkin", P.GRUMPKIN)):
        ks = [rnd.randrange(1, cv.q) for _ in range(40)]
        pts = [cv.mul(k, cv.g) for k in ks]
        special = [0, 1, cv.q - 1, (1 << 128) - 1, 1 << 16, (1 << 16) - 1, 0x8000, 0x8001, (1 << 255) % cv.q]
        msms = []
        for n in (0, 1, 2, 3, 31, 32, 33, 40):
            for kind in ("random", "special"):
                if kind == "random":
                    sc = [rnd.randrange(cv.q) for _ in range(n)]
                else:
                    sc = [special[i % len(special)] for i in range(n)]
                r = cv.msm(sc, pts[:n])
                msms.append({"n": n, "kind": kind, "scalars": [H(s) for s in sc], "out": [H(r[0]), H(r[1])]})
        adds = []
        for i in range(6):
            a, b = pts[i], pts[i + 1]
            adds.append({"a": [H(a[0]), H(a[1])], "b": [H(b[0]), H(b[1])], "sum": [H(v) for v in cv.add(a, b)],
                         "dbl": [H(v) for v in cv.add(a, a)], "a_minus_a": [H(v) for v in cv.add(a, cv.neg(a))]})
        curves[name] = {"generator": [H(cv.g[0]), H(cv.g[1])], "base_scalars": [H(k) for k in ks],
                        "bases": [[H(p[0]), H(p[1])] for p in pts], "msm": msms, "add": adds,
                        "order_minus_one_times_g": [H(v) for v in cv.mul(cv.q - 1, cv.g)]}
    json.dump(curves, open(os.path.join(HERE, "curve_msm.json"), "w"), indent=1)

    # ---- NTT vectors k = 0..10, forward / inverse / coset (src/fft.rs:160-198)
    ntt = []
    for k in range(0, 11):
        v = [rnd.randrange(P.FR) for _ in range(1 << k)]
        if k == 3:
            v = list(range(8))
        rec = {"k": k, "input": [H(x) for x in v]}
        for fn in ("fft", "ifft", "coset_fft", "coset_ifft"):
            w = list(v)
            getattr(P, fn)(w)
            rec[fn] = [H(x) for x in w]
        ntt.append(rec)
    json.dump(ntt, open(os.path.join(HERE, "ntt.json"), "w"))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
