"""Off-circuit Poseidon random oracle: the oracle restatement is pinned by the reference's known answer
(src/poseidon/poseidon_hash.rs:248-266); the library's host implementation (srs_poseidon_*, runs without a device) is checked
against the oracle on both fields of the bn256 / grumpkin cycle."""
import random

import numpy as np

from oracle import poseidon as OP
from oracle import pyref as P


def test_reference_known_answer():
    h = OP.PoseidonHash(OP.PASTA_FP, 3, 2, 4, 3).absorb_field_iter(range(5))
    assert h.squeeze(128) == 277726250230731218669330566268314254439
    assert h.squeeze(128) == 277726250230731218669330566268314254439       # squeezing restarts the state, keeps the buffer


def test_library_matches_oracle_on_both_fields():
    """srs_poseidon_* (host code) vs oracle/poseidon.py: field elements, points incl. the identity, exact / ragged / empty
    buffers, repeated squeezes, several (T, R_F, R_P)."""
    import oracle as O
    import sirius_amd as S
    rnd = random.Random(7)
    for field, p in ((0, P.FR), (1, P.FQ)):
        curve = 1 if field == 0 else 0                      # the curve whose base field is `field`: grumpkin / bn256
        for t, r_f, r_p in ((3, 4, 3), (3, 8, 56), (5, 10, 60), (2, 8, 56)):
            h = S.PoseidonHash(field, t, t - 1, r_f, r_p)
            o = OP.PoseidonHash(p, t, t - 1, r_f, r_p)
            assert O.mont_to_ints(1 - field, h.squeeze(128, 1 - field)) == [o.squeeze(128)]          # empty buffer
            for n in (1, t - 1, 2 * (t - 1) + 1, 7):
                vals = [rnd.randrange(p) for _ in range(n)]
                h.absorb_field(O.ints_to_mont(field, vals))
                o.absorb_field_iter(vals)
                for bits, of in ((128, 1 - field), (250, field)):
                    assert O.mont_to_ints(of, h.squeeze(bits, of)) == [o.squeeze(bits)]
            pts = O.make_bases(curve, 3, 2)
            pts[1] = 0                                       # the identity
            for pt in pts:
                h.absorb_point(curve, pt)
                xy = O.mont_to_ints(field, pt.reshape(2, 4))
                o.absorb_point(tuple(xy))
            assert O.mont_to_ints(field, h.squeeze(128, field)) == [o.squeeze(128)]
            first = h.reset().absorb_field(O.ints_to_mont(field, [5])).squeeze(128, field)
            assert O.mont_to_ints(field, first) == [OP.PoseidonHash(p, t, t - 1, r_f, r_p).absorb_field(5).squeeze(128)]
            h.close()
    import pytest
    with pytest.raises(S.SiriusAmdError):
        S.PoseidonHash(0, 3, 3, 4, 3)                        # RATE != T - 1
    hh = S.PoseidonHash(0)
    with pytest.raises(S.SiriusAmdError):
        hh.absorb_point(0, np.zeros(8, np.uint64))           # bn256 coordinates are Fq elements, this oracle is over Fr


def test_scalar_and_ifma_permutations_agree_with_the_oracle():
    """The host permutation has two implementations (csrc/poseidon.hip: 4 x 64-bit scalar code; csrc/poseidon_x86.hip: AVX-512 IFMA,
    chosen by cpuid).  Both are run against the oracle (one subprocess each; SRS_POSEIDON_SCALAR=1 forces the scalar one) on the
    transcript sizes of a ProtoGalaxy prove and on edge values (0, 1, p - 1) -- whatever CPU the suite runs on, the path it does
    not take by default is still covered when the CPU has IFMA, and the scalar path always is."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import random, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "import oracle as O, sirius_amd as S\n"
        "from oracle import poseidon as OP, pyref as P\n"
        "rnd = random.Random(11)\n"
        "for field, p in ((0, P.FR), (1, P.FQ)):\n"
        "    for t, r_f, r_p in ((5, 10, 10), (3, 4, 3), (8, 8, 9)):\n"
        "        h, o = S.PoseidonHash(field, t, t - 1, r_f, r_p), OP.PoseidonHash(p, t, t - 1, r_f, r_p)\n"
        "        for n in (25, 32, 256):\n"
        "            vals = [rnd.randrange(p) for _ in range(n - 3)] + [0, 1, p - 1]\n"
        "            h.absorb_field(O.ints_to_mont(field, vals)); o.absorb_field_iter(vals)\n"
        "            assert O.mont_to_ints(field, h.squeeze(253, field)) == [o.squeeze(253)], (field, t, n)\n"
        "print('ok')\n")
    for scalar in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, SRS_POSEIDON_SCALAR=scalar), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (scalar, r.stdout[-300:], r.stderr[-1500:])
