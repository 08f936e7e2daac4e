"""The lazy 9 x 29-bit arithmetic of the hot kernels (csrc/field29.cuh, csrc/curve29.cuh) checked on the HOST against Python
integers, with operands sitting AT the bounds the headers state: limbs up to 2^31 - 1, values up to 12 p, differences
with the offsets the curve code uses, exceptional cases of the group law (identity operands, Q = +-acc).  The device code is the
same source (SRS_HD functions); tests/emu/field29_check.cpp compiles it with g++ through the emulator headers."""
import os
import random
import subprocess

import pytest

from conftest import ROOT
from oracle import pyref as P

EMU = os.path.join(ROOT, "tests", "emu")
R261 = 1 << 261


@pytest.fixture(scope="module")
def calc():
    exe = os.path.join(EMU, "field29_check")
    src = os.path.join(EMU, "field29_check.cpp")
    deps = [src, os.path.join(EMU, "hipemu.h")] + [os.path.join(ROOT, "sirius_amd", "csrc", f) for f in ("field29.cuh", "curve29.cuh", "field.cuh", "curve.cuh")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["g++", "-std=c++20", "-O2", "-DSRS_EMU", "-I" + EMU, "-I" + os.path.join(ROOT, "sirius_amd", "csrc"), "-pthread",
                               "-Wno-unknown-pragmas", "-Wno-attributes", src, "-o", exe])
    proc = subprocess.Popen([exe], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)

    def ask(line):
        proc.stdin.write(line + "\n")
        proc.stdin.flush()
        out = proc.stdout.readline().strip()
        assert out and out != "unsupported", line[:80]
        return [int(x, 16) for x in out.split()]
    yield ask
    proc.stdin.close()
    proc.wait(timeout=10)


def limbs29(x, wide=0, rnd=None):
    """x -> 9 limbs of 29 bits (limb 8 takes the rest); wide: re-distribute so that limbs reach up to 2^(29+wide) - 1"""
    l = [(x >> (29 * i)) & ((1 << 29) - 1) for i in range(8)] + [x >> 232]
    if wide:
        for i in range(8):
            room = ((1 << (29 + wide)) - 1 - l[i]) >> 29
            t = min(room, l[i + 1])
            if rnd is not None and t:
                t = rnd.randrange(t + 1)
            l[i] += t << 29
            l[i + 1] -= t
    assert all(0 <= v < (1 << 32) for v in l) and sum(v << (29 * i) for i, v in enumerate(l)) == x
    return l


val = lambda l: sum(v << (29 * i) for i, v in enumerate(l))
hx = lambda l: " ".join(f"{v:x}" for v in l)
words = lambda x, n: [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
FIELDS = [("Fr", P.FR), ("Fq", P.FQ)]


def edge_values(p, rnd, kmax):
    out = []
    for k in range(kmax + 1):
        for r in (0, 1, p - 1, p // 2, rnd.randrange(p)):
            v = k * p + r
            if v < (kmax + 1) * p and v < (1 << 260):
                out.append(v)
    return out


@pytest.mark.parametrize("name,p", FIELDS)
def test_products_at_the_bounds(calc, name, p):
    rnd = random.Random(29)
    inv = pow(R261, -1, p)
    cases = []
    big = edge_values(p, rnd, 12)
    for a in big[::3]:
        for b in (0, 1, p - 1, 2 * p - 1, 12 * p - 1, rnd.randrange(12 * p)):
            cases.append((a, b))
    cases += [(rnd.randrange(12 * p), rnd.randrange(12 * p)) for _ in range(300)]
    for a, b in cases:
        assert a * b < R261 * p
        for wa in (0, 2):                                   # the wide operand: limbs < 2^31, the other normalised
            la, lb = limbs29(a, wa, rnd), limbs29(b)
            r = calc(f"mul {name} {hx(la)} {hx(lb)}")
            assert all(v < (1 << 29) for v in r[:8]) and val(r) < 2 * p and val(r) % p == a * b * inv % p, (a, b, wa)
        la = limbs29(a, 1, rnd)                             # squares: limbs < 2^30
        r = calc(f"sqr {name} {hx(la)}")
        if a * a < R261 * p:
            assert all(v < (1 << 29) for v in r[:8]) and val(r) < 2 * p and val(r) % p == a * a * inv % p, a
    # both operands with limbs < 2^30
    for _ in range(100):
        a, b = rnd.randrange(12 * p), rnd.randrange(12 * p)
        r = calc(f"mul {name} {hx(limbs29(a, 1, rnd))} {hx(limbs29(b, 1, rnd))}")
        assert val(r) < 2 * p and val(r) % p == a * b * inv % p
    # mul2: (a b + c d) / 2^261 with ONE reduction -- a, c with limbs < 2^30, b, d normalised, at the bounds madd_signed uses
    # (t < 12P, r < 8P; 6P - y <= 6P, ppp < 2P) and at the extreme all-ones limb patterns
    quads = [(12 * p - 1, 8 * p - 1, 6 * p, 2 * p - 1), (0, 0, 0, 0), (p - 1, p - 1, p - 1, p - 1), (12 * p - 1, 8 * p - 1, 0, 0)]
    quads += [(rnd.randrange(12 * p), rnd.randrange(8 * p), rnd.randrange(6 * p + 1), rnd.randrange(2 * p)) for _ in range(300)]
    for a, b, c, d in quads:
        assert a * b + c * d < R261 * p
        for wa, wc in ((0, 0), (1, 1), (0, 1), (1, 0)):
            r = calc(f"mul2 {name} {hx(limbs29(a, wa, rnd))} {hx(limbs29(b))} {hx(limbs29(c, wc, rnd))} {hx(limbs29(d))}")
            assert all(v < (1 << 29) for v in r[:8]) and val(r) < 2 * p and val(r) % p == (a * b + c * d) * inv % p, (a, b, c, d, wa, wc)
    ones30, ones29 = [(1 << 30) - 1] * 8 + [(1 << 22) - 1], [(1 << 29) - 1] * 8 + [(1 << 21) - 1]      # every limb at its bound
    r = calc(f"mul2 {name} {hx(ones30)} {hx(ones29)} {hx(ones30)} {hx(ones29)}")
    assert val(r) % p == 2 * val(ones30) * val(ones29) * inv % p


@pytest.mark.parametrize("name,p", FIELDS)
def test_lazy_sums_differences_and_reductions(calc, name, p):
    rnd = random.Random(31)
    vals = edge_values(p, rnd, 12)
    for a in vals[::2]:
        la = limbs29(a, 2, rnd)
        r = calc(f"norm {name} {hx(la)}")
        assert val(r) == a and all(v < (1 << 29) for v in r[:8])
        b = rnd.choice(vals)
        r = calc(f"add {name} {hx(limbs29(a, 1, rnd))} {hx(limbs29(b, 1, rnd))}")
        assert val(r) == a + b
    # a - b + CP p with the (CP, E) pairs of curve29.cuh and the sweep emitter; b below CP p with limbs < 2^(29 + E)
    for cp, e in ((1, 0), (2, 0), (3, 0), (5, 1), (6, 2), (7, 2), (8, 0), (10, 0), (13, 0), (31, 0), (3, 2), (12, 2), (6, 0), (8, 1)):
        for _ in range(40):
            b = rnd.choice([0, 1, p - 1, cp * p - 1, rnd.randrange(cp * p), max(0, cp * p - (1 << e) * (1 << 232))])
            lb = limbs29(b, e, rnd)
            top_ok = lb[8] <= ((cp * p) >> 232) - (1 << e)
            if not top_ok:
                continue
            a = rnd.randrange(12 * p)
            la = limbs29(a, 1, rnd)
            r = calc(f"sub {name} {cp} {e} {hx(la)} {hx(lb)}")
            assert val(r) == a - b + cp * p, (cp, e)
            r = calc(f"neg {name} {cp} {e} {hx(lb)}")
            assert val(r) == cp * p - b
    for v in [0, 1, p - 1, p, p + 1, 2 * p - 1, 2 * p, 3 * p, 4 * p - 1] + [rnd.randrange(4 * p) for _ in range(100)]:
        r = calc(f"canon {name} {hx(limbs29(v))}")
        assert sum(w << (32 * i) for i, w in enumerate(r)) == v % p
    # r05 reduce_lazy (the NTT's closing reduction): any normalised value below 2^263 -> the same residue below 2P, normalised
    top = (1 << 263) - 1
    for v in ([0, 1, p - 1, p, p + 1, 2 * p, 469 * p, 469 * p - 1, 470 * p + 5, 700 * p, top, top - p, (1 << 232) - 1, 1 << 232, (1 << 262) + 12345]
              + [k * p + d for k in (1, 7, 29, 117, 235, 468, 600) for d in (-1, 0, 1)] + [rnd.randrange(top) for _ in range(300)]
              + [rnd.randrange(1, 720) * p + rnd.choice([0, 1, p - 1, rnd.randrange(p)]) for _ in range(300)]):
        v = min(v, top)
        r = calc(f"redlazy {name} {hx(limbs29(v))}")
        assert val(r) % p == v % p and val(r) < 2 * p and all(x < (1 << 29) for x in r[:8]), hex(v)
    for v in [0, 1, (1 << 256) - 1, p, rnd.randrange(1 << 256)]:
        l = calc(f"unpack {name} {hx(words(v, 8))}")
        assert val(l) == v and all(x < (1 << 29) for x in l[:8])
        assert calc(f"pack {name} {hx(l)}") == words(v, 8)


@pytest.mark.parametrize("cname,cid", [("Bn256", 0), ("Grumpkin", 1)])
def test_group_law_on_the_lazy_form(calc, cname, cid):
    cv = P.CURVES[cid]
    p = cv.p
    rnd = random.Random(7 + cid)
    R256, R261p = (1 << 256) % p, R261 % p
    aff_abi = lambda Pt: words(Pt[0] * R256 % p, 8) + words(Pt[1] * R256 % p, 8)

    def tform(Pt):                                          # ABI affine -> table form through the library, checked
        w = calc(f"tform {cname} {hx(aff_abi(Pt))}")
        x = sum(v << (32 * i) for i, v in enumerate(w[:8]))
        y = sum(v << (32 * i) for i, v in enumerate(w[8:]))
        assert (x, y) == ((Pt[0] * R261p) % p, (Pt[1] * R261p) % p) or Pt == (0, 0)
        return w

    def point_of(packed):                                   # packed R'-form XYZZ -> affine integers (through to_xyzz, checked both ways)
        c = [sum(v << (32 * i) for i, v in enumerate(packed[8 * j: 8 * j + 8])) for j in range(4)]
        assert all(v < p for v in c)
        ir = pow(R261p, -1, p)
        x, y, zz, zzz = [(v * ir) % p for v in c]
        abi = calc(f"xyzz {cname} {hx(packed)}")
        ca = [sum(v << (32 * i) for i, v in enumerate(abi[8 * j: 8 * j + 8])) for j in range(4)]
        assert ca == [(v * R256) % p for v in (x, y, zz, zzz)]
        if zz == 0:
            return (0, 0)
        assert (zz * zz * zz - zzz * zzz) % p == 0
        return (x * pow(zz, -1, p) % p, y * pow(zzz, -1, p) % p)

    G = cv.g
    pts = [cv.mul(rnd.randrange(1, cv.q), G) for _ in range(12)]

    def chain(seq):                                         # seq of (point, negate): Ec29::madd(load(..)) AND madd_signed(load_raw(..))
        body = f"{cname} {len(seq)} " + " ".join(hx(tform(Q)) + f" {int(n)}" for Q, n in seq)
        a, b = calc("chain " + body), calc("chains " + body)
        if a != b:                                          # packed records are canonical except for the scale of (zz, zzz): compare points
            assert point_of(a) == point_of(b), seq
        return b
    # plain sums, long enough for the lazy bounds to reach their steady state
    seq = [(rnd.choice(pts), rnd.random() < 0.5) for _ in range(40)]
    want = (0, 0)
    for Q, n in seq:
        want = cv.add(want, cv.neg(Q) if n else Q)
    assert point_of(chain(seq)) == want
    # exceptional cases inside a chain: first addition onto the identity, Q = acc (doubling), Q = -acc (back to the identity),
    # identity table entries (skipped bases), and again after the identity
    A, B = pts[0], pts[1]
    cases = [
        [(A, False)], [(A, False), (A, False)], [(A, False), (A, True)], [(A, False), (A, True), (B, False)],
        [((0, 0), False)], [(A, False), ((0, 0), False), (B, True)], [(A, False), (A, False), (A, False), (A, False)],
        [(A, False), (B, False), (cv.add(A, B), True)], [(A, False), (B, False), (cv.add(A, B), False)],
    ]
    for seq in cases:
        want = (0, 0)
        for Q, n in seq:
            want = cv.add(want, cv.neg(Q) if n else Q)
        assert point_of(chain(seq)) == want, seq
    # full additions / doublings of packed partial sums (what the accumulation levels and the bucket reduction run on)
    S1, S2 = chain([(pts[2], False), (pts[3], False), (pts[4], True)]), chain([(pts[5], False), (pts[6], True)])
    P1, P2 = point_of(S1), point_of(S2)
    Z = chain([(A, False), (A, True)])                      # identity as a packed record
    assert point_of(calc(f"addp {cname} {hx(S1)} {hx(S2)}")) == cv.add(P1, P2)
    assert point_of(calc(f"addp {cname} {hx(S1)} {hx(S1)}")) == cv.add(P1, P1)             # equal operands -> doubling
    S1n = chain([(pts[2], True), (pts[3], True), (pts[4], False)])
    assert point_of(calc(f"addp {cname} {hx(S1)} {hx(S1n)}")) == (0, 0)                    # opposite operands
    assert point_of(calc(f"addp {cname} {hx(Z)} {hx(S2)}")) == P2 and point_of(calc(f"addp {cname} {hx(S2)} {hx(Z)}")) == P2
    assert point_of(calc(f"addp {cname} {hx(Z)} {hx(Z)}")) == (0, 0)
    assert point_of(calc(f"dblp {cname} {hx(S1)}")) == cv.add(P1, P1) and point_of(calc(f"dblp {cname} {hx(Z)}")) == (0, 0)
