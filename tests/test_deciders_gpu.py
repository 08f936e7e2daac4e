"""GPU parity: the Sangria deciders' copy-constraint check (sparse matvec) and witness-commitment check
(src/nifs/sangria/mod.rs:385-474, src/polynomial/sparse.rs:7-19) against plain big-int restatements."""
import numpy as np
import pytest

from oracle import pyref as P
from workloads import rand_fe

pytestmark = pytest.mark.gpu


def _sparse_reference(O, field, n, rows, cols, vals, Z):
    """sparse::matrix_multiply as written (sparse.rs:7-19): result[row] += value * Z[col], serial."""
    p = P.MODULI[field]
    zi, vi = O.mont_to_ints(field, Z), O.mont_to_ints(field, vals)
    out = [0] * n
    for r, c, v in zip(rows, cols, vi):
        out[int(r)] = (out[int(r)] + v * zi[int(c)]) % p
    return O.ints_to_mont(field, out)


def _permutation_case(S, O, field, k, num_advice, num_io):
    """Copy constraints as cycles over Z = instances || advice cells (PermutationData::matrix shape): the matrix maps every
    cell to the next cell of its cycle; a witness satisfies the check iff every cycle holds one value."""
    rng = np.random.default_rng(k * 11 + num_advice)
    rows_n = 1 << k
    n = num_io + num_advice * rows_n
    perm = np.arange(n)
    cells = rng.permutation(n)
    cycles, at = [], 0
    while at < n - 4:                                  # a few hundred cycles of length 2..5, the rest fixed points
        ln = int(rng.integers(2, 6))
        cyc = cells[at:at + ln]
        at += ln + int(rng.integers(0, 8))
        cycles.append(cyc)
        for i in range(ln):
            perm[cyc[i]] = cyc[(i + 1) % ln]
    one = O.ints_to_mont(field, [1])[0]
    M = S.SparseMatrix(field, n, np.arange(n), perm, np.broadcast_to(one, (n, 4)))
    Z = rand_fe(rng, n)
    for cyc in cycles:
        Z[cyc] = Z[cyc[0]]
    assert S.VanillaFS.is_sat_permutation(M, Z) == 0
    assert np.array_equal(M.matrix_multiply(Z), Z[perm])
    # break three cycles: each broken cell and the cell mapped onto it disagree -> 2 mismatching rows per break
    Zb = Z.copy()
    broken = [cycles[1][0], cycles[5][1], cycles[9][0]]
    for c in broken:
        Zb[c] = rand_fe(rng, 1)[0]
    exp = int(np.count_nonzero(np.any(Zb[perm] != Zb, axis=1)))
    assert exp >= 3 and S.VanillaFS.is_sat_permutation(M, Zb) == exp
    M.close()
    return n


def _general_sparse_case(S, O, field, n, nnz, seed):
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, n, size=nnz).astype(np.uint64)
    cols = rng.integers(0, n, size=nnz).astype(np.uint64)
    rows[: nnz // 8] = 3                                # a heavy row and duplicates of (row, col)
    cols[: nnz // 16] = 5
    vals = rand_fe(rng, nnz)
    Z = rand_fe(rng, n)
    M = S.SparseMatrix(field, n, rows, cols, vals)
    assert np.array_equal(M.matrix_multiply(Z), _sparse_reference(O, field, n, rows, cols, vals, Z))
    M.close()
    with pytest.raises(S.SiriusAmdError) as e:          # sparse.rs:15-17 "invalid matrix multiply"
        S.SparseMatrix(field, n, rows, np.where(np.arange(nnz) == 2, n, cols).astype(np.uint64), vals)
    assert "invalid matrix multiply" in str(e.value)


def _witness_commit_case(S, O, curve, n0, n1):
    field = O.SCALAR_FIELD[curve]
    rng = np.random.default_rng(n0 + n1)
    ck = S.CommitmentKey.setup_synthetic(curve, max(n0, n1), seed=3)
    W = [rand_fe(rng, n0, 0.5), rand_fe(rng, n1, 0.5)]
    E = rand_fe(rng, n1 // 2)
    cw = np.stack([ck.commit(w) for w in W])
    ce = ck.commit(E)
    assert S.VanillaFS.is_sat_witness_commit(ck, W, cw, E, ce) == (0, False)
    assert S.VanillaFS.is_sat_witness_commit(ck, W, cw) == (0, False)          # PlonkStructure::is_sat has no E
    W[1][7] = rand_fe(rng, 1)[0]
    assert S.VanillaFS.is_sat_witness_commit(ck, W, cw, E, ce) == (1, False)
    E[0] = rand_fe(rng, 1)[0]
    assert S.VanillaFS.is_sat_witness_commit(ck, W, cw[::-1].copy(), E, ce) == (2, True)
    with pytest.raises(S.SiriusAmdError) as e:
        S.VanillaFS.is_sat_witness_commit(ck, [rand_fe(rng, max(n0, n1) + 1)], cw[:1])
    assert e.value.rc == 1
    ck.close()


def test_permutation_check(srs, oracle):
    _permutation_case(srs, oracle, 0, 8, 3, 2)
    _permutation_case(srs, oracle, 1, 6, 7, 4)


def test_permutation_check_k17_device_resident(srs, oracle):
    """Primary-circuit size: Z = 2 instances || 12 * 2^17 advice cells resident in HBM."""
    import torch
    O = oracle
    field, n = 0, 2 + 12 * (1 << 17)
    rng = np.random.default_rng(5)
    perm = np.arange(n)
    pairs = rng.permutation(n)[: 200000].reshape(-1, 2)
    perm[pairs[:, 0]], perm[pairs[:, 1]] = pairs[:, 1], pairs[:, 0]
    one = O.ints_to_mont(field, [1])[0]
    M = srs.SparseMatrix(field, n, np.arange(n), perm, np.broadcast_to(one, (n, 4)))
    Z = rand_fe(rng, n)
    Z[pairs[:, 1]] = Z[pairs[:, 0]]
    Zd = torch.from_numpy(Z.view(np.int64)).cuda()
    assert srs.VanillaFS.is_sat_permutation(M, Zd) == 0
    assert torch.equal(M.matrix_multiply(Zd), Zd[torch.from_numpy(perm).cuda()])
    Zd[int(pairs[17, 0])] = 0
    assert srs.VanillaFS.is_sat_permutation(M, Zd) == 2
    M.close()


def test_general_sparse_matvec(srs, oracle):
    _general_sparse_case(srs, oracle, 0, 300, 2000, 1)
    _general_sparse_case(srs, oracle, 1, 64, 64, 2)


def test_witness_commit_check(srs, oracle):
    _witness_commit_case(srs, oracle, 0, 3000, 1000)
    _witness_commit_case(srs, oracle, 1, 500, 2048)
