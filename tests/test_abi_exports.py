"""CPU: the C-ABI library loads and exports every symbol include/sirius_amd.h declares.
No compute entry is called here (there is no GPU in this container and no CPU fallback)."""
import ctypes
import os
import re

import numpy as np

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "sirius_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srs_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sirius_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib._prototypes()) == names, "python prototypes drifted from the header"


def test_host_only_entries():
    """Entry points that are pure host code may be exercised without a device."""
    import sirius_amd as S
    from sirius_amd import _lib
    lib = _lib.load()
    assert lib.srs_version().startswith(b"sirius_amd")
    assert lib.srs_scalar_field_of(0) == 0 and lib.srs_scalar_field_of(1) == 1
    import oracle as O
    for f in (0, 1):
        one, two = O.ints_to_mont(f, [1])[0], O.ints_to_mont(f, [2])[0]
        assert lib.srs_layout_selftest(f, one.ctypes.data, two.ctypes.data) == 0
        assert lib.srs_layout_selftest(f, two.ctypes.data, one.ctypes.data) == _lib.ERR_LAYOUT
    # host-side group helpers agree with the oracle
    for cid in (0, 1):
        b = O.make_bases(cid, 5, 6)
        assert np.array_equal(S.point_sum(cid, b[:2]), O.point_add(cid, b[0], b[1]))
        s = O.ints_to_mont(O.SCALAR_FIELD[cid], [0xDEADBEEF12345])[0]
        assert np.array_equal(S.point_mul(cid, s, b[3]), O.point_mul(cid, s, b[3]))
        assert np.array_equal(S.point_sum(cid, np.zeros((0, 8), np.uint64)), np.zeros(8, np.uint64))


def test_tunables_table():
    """srs_tuning_set / _get / _reset / _name (host code): the ONE table of run-time tunables that replaced the environment switches (r06).
    Every name listed in include/sirius_amd.h is known, unknown names are refused, values round-trip, the context manager restores."""
    import sirius_amd as S
    from sirius_amd import _lib
    names = S.tuning.names()
    header = open(os.path.join(ROOT, "include", "sirius_amd.h")).read()
    listed = re.search(r"Names \(srs_tuning_name.*?:(.*?)\.\s+Process-wide", header, flags=re.S).group(1)
    assert sorted(names) == sorted(n.strip() for n in re.sub(r"[*\n]", " ", listed).split(",")), (names, listed)
    assert all(S.tuning.get(n) is None for n in names)
    with S.tuning(msm_sort=2, commit_chunks=3):
        assert S.tuning.get("msm_sort") == 2 and S.tuning.get("commit_chunks") == 3
        with S.tuning(msm_sort=1):
            assert S.tuning.get("msm_sort") == 1
        assert S.tuning.get("msm_sort") == 2
    assert S.tuning.get("msm_sort") is None and S.tuning.get("commit_chunks") is None
    try:
        S.tuning.set("no_such_tunable", 1)
    except S.SiriusAmdError as e:
        assert e.rc == _lib.ERR_INVALID and "no_such_tunable" in str(e)
    else:
        raise AssertionError("unknown tunable accepted")
    S.tuning.set("msm_l0", 5)
    _lib.lib().srs_tuning_reset()
    assert S.tuning.get("msm_l0") is None
    # the library itself reads a fixed, short list of environment variables
    src = "".join(open(os.path.join(ROOT, "sirius_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "sirius_amd", "csrc"))
                  if f.endswith((".hip", ".h", ".cuh")) )
    assert sorted(set(re.findall(r'getenv\("(\w+)"\)', src))) == ["ROCM_PATH", "SRS_HOST_TRACE", "SRS_JIT_DUMP", "SRS_MSM_WIDE", "SRS_NO_JIT", "SRS_POSEIDON_SCALAR"]


def test_lagrange_values_on_and_off_the_domain():
    """srs_lagrange_eval (host code, one batched inversion): iter_eval_lagrange_poly_for_cyclic_group (src/polynomial/lagrange.rs:50-75)
    for challenges off the domain, ON the domain (X = w^i: the unit vector, the branch the reference special-cases) and X = 0."""
    import oracle as O
    from oracle import pyref as P
    from sirius_amd import protogalaxy as PG
    m = lambda v: O.ints_to_mont(O.FR, list(v))
    for log_n in (0, 1, 2, 4):
        n = 1 << log_n
        w = pow(P.FR_ROOT_OF_UNITY, 1 << (P.FR_S - log_n), P.FR) if log_n else 1
        for X in (0, 1, w, pow(w, n - 1, P.FR), 12345678901234567890, P.FR - 1, 7):
            got = O.mont_to_ints(O.FR, PG.eval_lagrange_poly_for_cyclic_group(m([X])[0], log_n))
            assert got == P.eval_lagrange_poly_for_cyclic_group(X, log_n), (log_n, X)


def test_async_instance_fold_matches_the_blocking_one():
    """srs_point_lincomb_async / srs_job_wait (host workers): same points as srs_point_lincomb, any wait order, inputs may be
    released right after submission, a job can be waited for once."""
    import pytest
    import sirius_amd as S
    from sirius_amd import _lib
    import oracle as O
    from oracle import pyref as P
    lib = _lib.load()
    rng = np.random.default_rng(3)
    for cid in (0, 1):
        sf = O.SCALAR_FIELD[cid]
        b = O.make_bases(cid, 9, 12)
        jobs, expect = [], []
        for n in (0, 1, 6, 11):
            sc = O.ints_to_mont(sf, [int(rng.integers(1, 1 << 62)) ** 4 % P.MODULI[sf] for _ in range(n)]) if n else np.zeros((0, 4), np.uint64)
            acc = None if n == 1 else b[11]
            pts = b[:n].copy()
            expect.append(S.point_lincomb(cid, acc, pts, sc))
            jobs.append(S.point_lincomb_async(cid, acc, pts, sc))
            pts[:] = 0; sc[:] = 0          # inputs were copied at submission
        for j in (2, 0, 3, 1):             # any order
            assert np.array_equal(jobs[j].wait(), expect[j])
            assert np.array_equal(jobs[j].wait(), expect[j])       # the python handle is idempotent ...
        # n = 6 against the oracle's msm as an independent check
        sc = O.ints_to_mont(sf, list(range(3, 9)))
        assert np.array_equal(S.point_lincomb_async(cid, None, b[:6], sc).wait(), O.msm(cid, sc, b[:6]))
    out, job = np.zeros(8, np.uint64), ctypes.c_uint64()
    assert lib.srs_point_lincomb_async(0, None, None, None, 0, 1, out.ctypes.data, ctypes.byref(job)) == 0
    assert lib.srs_job_wait(job.value) == 0
    assert lib.srs_job_wait(job.value) == _lib.ERR_INVALID     # ... the C job is not
    assert lib.srs_job_wait(1 << 60) == _lib.ERR_INVALID
    assert lib.srs_point_lincomb_async(0, None, None, None, 0, 1, None, ctypes.byref(job)) == _lib.ERR_INVALID


def test_async_instance_fold_from_several_threads():
    """The job queue is shared by every caller: four host threads submit and join their own jobs concurrently."""
    import threading
    import sirius_amd as S
    import oracle as O
    from sirius_amd import _lib
    _lib.load()
    b = O.make_bases(0, 21, 8)
    sc = O.ints_to_mont(O.FR, [3, 5, 7, 11, 13, 17, 19, 23])
    expect = [S.point_lincomb(0, b[j % 8], b[:1 + j % 7], sc[:1 + j % 7]) for j in range(14)]
    errors = []

    def run(seed):
        try:
            order = np.random.default_rng(seed).permutation(14)
            for _ in range(6):
                jobs = {int(j): S.point_lincomb_async(0, b[j % 8], b[:1 + j % 7], sc[:1 + j % 7]) for j in order}
                for j in reversed(order):
                    if not np.array_equal(jobs[int(j)].wait(), expect[int(j)]):
                        errors.append((seed, int(j)))
        except Exception as e:      # surfaced in the main thread
            errors.append((seed, repr(e)))

    th = [threading.Thread(target=run, args=(s,)) for s in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors[:3]


def test_run_time_compilation_path_compiles_without_a_device():
    """hiprtc + the device headers embedded in the library accept the emitted program form (called and inlined multipliers,
    column loads, uniforms) for both fields -- a header that stops compiling under hiprtc would otherwise only show up as a
    silent fall-back to the interpreter on the GPU box."""
    from sirius_amd import _lib
    lib = _lib.load()
    log = ctypes.create_string_buffer(1 << 16)
    size = ctypes.c_size_t()
    rc = lib.srs_jit_selfcheck(ctypes.byref(size), log, len(log))
    assert rc == 0, log.value.decode(errors="replace")[-3000:]
    assert size.value > 4096          # a real gfx950 code object


def test_univariate_eval_kats():
    """UnivariatePoly::eval known answers of the reference (src/polynomial/univariate.rs:197-262) through srs_poly_eval
    (host code: runs without a device)."""
    import oracle as O
    from sirius_amd import protogalaxy as PG
    m = lambda v: O.ints_to_mont(O.FR, list(v))
    ev = lambda coeffs, x: O.mont_to_ints(O.FR, PG.poly_eval(m(coeffs) if coeffs else np.zeros((0, 4), np.uint64), m([x])[0]))[0]
    assert ev([5], 10) == 5                                   # test_constant_polynomial
    assert ev([3, 2], 4) == 11                                # test_linear_polynomial
    assert ev([3, 2, 1], 2) == 11                             # test_quadratic_polynomial
    assert ev([5, 1, 2, 3, 4], 2) == 5 + 2 + 8 + 24 + 64      # test_high_degree_polynomial
    assert ev([], 1) == 0                                     # test_zero_polynomial


def test_no_device_fails_loudly():
    """Without a GPU a compute entry must return SRS_ERR_DEVICE, never a CPU result."""
    import torch
    if torch.cuda.is_available():
        return
    import sirius_amd as S
    import oracle as O
    import pytest
    with pytest.raises(S.SiriusAmdError) as e:
        S.CommitmentKey(0, O.make_bases(0, 1, 4))
    assert e.value.rc == 5


def test_host_group_entries_vs_eip196(oracle):
    """The library's HOST group arithmetic (srs_point_sum / srs_point_mul / srs_point_lincomb: the instance folds of
    RelaxedPlonkInstance::fold, accumulator.rs:201-264) against the EIP-196 known answers.  No device needed."""
    import eip196_cases as E
    import sirius_amd as S
    E.check_adder(oracle, lambda a, b: S.point_sum(0, np.stack([a, b])), lambda k, p: S.point_mul(0, k, p),
                  lambda s, b: S.point_lincomb(0, None, b, s))
