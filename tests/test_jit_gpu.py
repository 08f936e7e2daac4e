"""GPU parity: row programs compiled at run time with hiprtc (csrc/jit.hip) -- structures without an ahead-of-time kernel
and >= 2^14 rows.  Same results as the LDS interpreter and as the oracle, bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import expr as OE
from workloads import gates_for, rand_fe

pytestmark = pytest.mark.gpu


def _kind(S, St):
    from sirius_amd import _lib
    sid, fp = C.c_int(), C.c_uint64()
    _lib.lib().srs_structure_program_source(St._h, 0, None, 0, C.byref(fp), C.byref(sid))
    return sid.value


@pytest.mark.parametrize("field,gate_T", [(1, [3, 2]), (0, [2, 5, 2])])
def test_jit_cross_terms(srs, oracle, field, gate_T):
    O = oracle
    k, rows = 14, 1 << 14
    gates, nfix, nadv = gates_for(gate_T)
    og, fo, ao = [], 0, 0
    for T in gate_T:
        og.append(OE.main_gate_expression(T, 0, fo, ao, nfix)); fo += 2 * T + 5; ao += T + 2
    rng = np.random.default_rng(len(gate_T) + field)
    fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
    W1, W2 = rand_fe(rng, nadv * rows, 0.4), rand_fe(rng, nadv * rows)
    St = srs.PlonkStructure(field, k, [], fixed, nadv, gates)
    assert _kind(srs, St) == -2, "expected the run-time compiled kernel"
    with srs.tuning(no_jit=1):
        Si = srs.PlonkStructure(field, k, [], fixed, nadv, gates)
    assert _kind(srs, Si) == -1
    nch = St.num_challenges
    u1c, u1u, u2c = rand_fe(rng, nch), rand_fe(rng, 1)[0], rand_fe(rng, nch)
    tj, _ = srs.VanillaFS.commit_cross_terms(None, St, u1c, u1u, W1, u2c, W2)
    ti, _ = srs.VanillaFS.commit_cross_terms(None, Si, u1c, u1u, W1, u2c, W2)
    ch = srs.VanillaFS.cross_term_challenges(u1c, u1u, u2c, field)
    cg, exp = OE.cross_terms_oracle(O, field, og, 0, nfix, nadv, [], fixed, W1, W2, ch)
    assert len(tj) == len(ti) == len(exp) == cg.degree
    for a, b, c in zip(tj, ti, exp):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    St.close()
    Si.close()


def test_jit_lookup_structure(srs, oracle):
    """A structure with a vector lookup at 2^14 rows runs its (lookup-extended) cross-term program through hiprtc too."""
    from lookup_cases import run_lookup_case
    run_lookup_case(srs, oracle, "vector", 14)
