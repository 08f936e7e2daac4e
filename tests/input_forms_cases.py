"""Regression for the ctypes keep-alive bug of r01: host inputs that are NOT contiguous uint64 arrays -- strided
views, int64-typed arrays -- are converted by the binding into temporaries, and those temporaries must stay alive until the C call
has returned.  Every entry that takes more than one such input is driven with both forms, many times (fresh allocations of the
same size in between make a freed temporary likely to be overwritten), and compared with the plain contiguous call."""
import numpy as np


def _forms(a):
    """the same (n, 4) uint64 content as: a strided view (every other row of a larger buffer), an int64-typed array"""
    big = np.zeros((2 * a.shape[0], 4), dtype=np.uint64)
    big[::2] = a
    big[1::2] = 0xDEADBEEFDEADBEEF
    return [big[::2], a.view(np.int64).copy()]


def run_input_forms_case(S, O, field=0, n=3000, reps=12):
    from workloads import rand_fe
    rng = np.random.default_rng(31)
    w1, w2, e, t1, t2 = (rand_fe(rng, n) for _ in range(5))
    r = rand_fe(rng, 1)[0]
    want = S.RelaxedPlonkWitness(field, [w1], e).fold([w2], [t1, t2], r)
    assert np.array_equal(want.W[0], O.fold_w(field, w1, w2, r)) and np.array_equal(want.E, O.fold_e(field, e, [t1, t2], r))
    l, t, m = rand_fe(rng, n), rand_fe(rng, n), rand_fe(rng, n)
    St = type("LookupOnly", (), {"field": field, "lookup_coeff_2": S.PlonkStructure.lookup_coeff_2})()
    hs0, gs0 = St.lookup_coeff_2([l], [t], [m], r)
    num, den = rand_fe(rng, n), rand_fe(rng, n)
    den[::7] = 0
    inv0 = S.batch_invert_assigned(field, num, den)
    for rep in range(reps):
        for k in range(2):
            f = lambda a: _forms(a)[k]
            got = S.RelaxedPlonkWitness(field, [f(w1)], f(e)).fold([f(w2)], [f(t1), f(t2)], r)
            junk = [np.full((n, 4), rep * 17 + j, dtype=np.uint64) for j in range(6)]      # same-size allocations: reuse freed temporaries
            assert np.array_equal(got.W[0], want.W[0]) and np.array_equal(got.E, want.E), ("fold", k, rep)
            hs, gs = St.lookup_coeff_2([f(l)], [f(t)], [f(m)], r)
            assert np.array_equal(hs[0], hs0[0]) and np.array_equal(gs[0], gs0[0]), ("lookup_coeff_2", k, rep)
            inv = S.batch_invert_assigned(field, f(num), f(den))
            assert np.array_equal(inv, inv0), ("batch_invert_assigned", k, rep)
            del junk
