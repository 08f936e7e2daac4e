"""Host-side mirror of the Sangria prover types this library accelerates.

  PlonkStructure                      <- src/plonk/mod.rs:127-157 (the slice the row programs need)
  VanillaFS.commit_cross_terms        <- src/nifs/sangria/mod.rs:102-158
  RelaxedPlonkWitness.fold            <- src/nifs/sangria/accumulator.rs:364-404
  PlonkStructure.eval_gates           <- deciders: src/plonk/mod.rs:304-361, src/nifs/sangria/mod.rs:334-383
Witness vectors are numpy (n,4) uint64 (host) or torch CUDA int64/uint64 tensors (resident in HBM).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .commitment import _buf, _is_torch, _stream
from .expression import LookupArguments, serialize_gates


def _alloc_like(ref, rows):
    if _is_torch(ref):
        import torch
        return torch.empty((rows, 4), dtype=ref.dtype, device=ref.device)
    return np.zeros((rows, 4), dtype=np.uint64)


def _cat_rounds(W):
    """PlonkWitness::W (a list of round vectors) -> the concatenated vector the library addresses; a single
    vector passes through."""
    if not isinstance(W, (list, tuple)):
        return W
    if len(W) == 1:
        return W[0]
    if _is_torch(W[0]):
        import torch
        return torch.cat([w.reshape(-1, 4) for w in W], dim=0)
    return np.ascontiguousarray(np.concatenate([np.asarray(w, dtype=np.uint64).reshape(-1, 4) for w in W], axis=0))


class PlonkStructure:
    """k, selectors, fixed_columns, num_advice_columns, gates, lookup arguments  (src/plonk/mod.rs:127-157).

    `gates` are the custom gates; `lookups` = [(input_expressions, table_expressions), ...] (or None).  As
    ConstraintSystemMetainfo::build does (src/table/constraint_system_metainfo.rs:27-97), the lookup expressions are
    appended to `self.gates` and `round_sizes` is derived."""

    def __init__(self, field, k, selectors, fixed_columns, num_advice_columns, gates, lookups=None):
        self.field, self.k, self.num_advice_columns = field, k, num_advice_columns
        rows = 1 << k
        sel = [np.ascontiguousarray(s, dtype=np.uint8) for s in selectors]
        fix = [np.ascontiguousarray(f, dtype=np.uint64) for f in fixed_columns]
        assert all(s.shape == (rows,) for s in sel) and all(f.shape == (rows, 4) for f in fix)
        self.num_selectors, self.num_fixed = len(sel), len(fix)
        selp = (C.c_void_p * max(len(sel), 1))(*[s.ctypes.data for s in sel])
        fixp = (C.c_void_p * max(len(fix), 1))(*[f.ctypes.data for f in fix])
        h = C.c_void_p()
        self.lookup_arguments = LookupArguments(lookups) if lookups else None
        if self.lookup_arguments is None:
            self.gates = list(gates)
            self.num_lookups, self.has_vector_lookup = 0, False
            self.round_sizes = [num_advice_columns * rows]
            words = serialize_gates(self.gates, field)
            L.check(L.lib().srs_structure_create(field, k, len(sel), len(fix), num_advice_columns, selp, fixp, L.SPACE_HOST,
                                                 words.ctypes.data, len(words), len(self.gates), C.byref(h)))
        else:
            la = self.lookup_arguments
            self.gates = list(gates) + la.to_expressions(len(sel), len(fix), num_advice_columns)
            self.num_lookups, self.has_vector_lookup = la.num_lookups, la.has_vector_lookup
            nl = la.num_lookups
            self.round_sizes = ([num_advice_columns * rows, 3 * nl * rows, 2 * nl * rows] if la.has_vector_lookup
                                else [(num_advice_columns + 3 * nl) * rows, 2 * nl * rows])
            words = serialize_gates(self.gates, field)
            lwords = serialize_gates(la.lookup_polys + la.table_polys, field)
            L.check(L.lib().srs_structure_create_lookup(field, k, len(sel), len(fix), num_advice_columns, selp, fixp,
                                                        L.SPACE_HOST, words.ctypes.data, len(words), len(self.gates), nl,
                                                        1 if la.has_vector_lookup else 0, lwords.ctypes.data, len(lwords),
                                                        C.byref(h)))
        self._h = h
        self.num_challenges = L.lib().srs_structure_num_challenges(h)
        self.num_cross_terms = L.lib().srs_structure_num_cross_terms(h)
        self.num_witness_columns = L.lib().srs_structure_num_witness_columns(h)

    @property
    def rows(self):
        return 1 << self.k

    def set_shard(self, rank, world):
        """Multi-GPU: cross terms only on the rows of this rank's key stripes (srs_structure_set_shard)."""
        L.check(L.lib().srs_structure_set_shard(self._h, rank, world))

    def upload_shard_halo(self, witness_host, dev_copy, reference_compat=False):
        """After a sharded `commit_upload` only this rank's key stripes of the witness are resident: bring up the rows the rank's
        sharded kernels read beyond them (rotation halo; row 0 of every column for reference_compat) -- srs_structure_upload_shard_halo."""
        a = np.ascontiguousarray(witness_host, dtype=np.uint64).reshape(-1, 4)
        dp = dev_copy.data_ptr() if _is_torch(dev_copy) else dev_copy.ctypes.data
        L.check(L.lib().srs_structure_upload_shard_halo(self._h, a.ctypes.data, dp, a.shape[0], int(bool(reference_compat)), _stream()))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().srs_structure_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval_gates(self, W, challenges, homogeneous=False):
        """Per-row gate value; homogeneous=False: compressed gate, challenges = U.challenges;
        homogeneous=True: challenges = U.challenges || U.u."""
        W = _cat_rounds(W)
        addr, space, n, keep = _buf(W, 4)
        assert n == self.num_witness_columns * self.rows
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)
        out = _alloc_like(W, self.rows)
        oaddr = out.data_ptr() if _is_torch(out) else out.ctypes.data
        L.check(L.lib().srs_eval_gates(self._h, 1 if homogeneous else 0, addr, ch.ctypes.data, ch.shape[0], space, _stream(), oaddr))
        return out


    def is_sat_gates(self, W, challenges, E=None):
        """Mismatch count of the deciders' gate check: E is None -> compressed gate == 0 per row
        (PlonkStructure::is_sat, src/plonk/mod.rs:329-346, challenges = U.challenges); otherwise homogeneous
        gate == E[row] (is_sat_accumulation, src/nifs/sangria/mod.rs:352-376, challenges = U.challenges || U.u)."""
        W = _cat_rounds(W)
        addr, space, n, keep = _buf(W, 4)
        assert n == self.num_witness_columns * self.rows
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)
        eaddr = None
        if E is not None:
            eaddr, espace, en, ekeep = _buf(E, 4)
            assert espace == space and en == self.rows
        cnt = C.c_size_t()
        L.check(L.lib().srs_is_sat_gates(self._h, 0 if E is None else 1, addr, ch.ctypes.data, ch.shape[0], eaddr, space,
                                         _stream(), C.byref(cnt)))
        return cnt.value

    # ---- lookup arguments (src/plonk/lookup.rs) ----
    def lookup_coeff_1(self, advice, r):
        """Arguments::evaluate_coefficient_1 (lookup.rs:319-341) -> (ls, ts, ms), each a list of num_lookups vectors.
        advice: the advice columns (num_advice * 2^k, column-major); r: the challenge the vector lookups are
        compressed with (r1 in run_sps_protocol_3, 0 in run_sps_protocol_2, plonk/mod.rs:515,618)."""
        addr, space, n, keep = _buf(advice, 4)
        assert n == self.num_advice_columns * self.rows and self.num_lookups > 0
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        outs = [[_alloc_like(advice, self.rows) for _ in range(self.num_lookups)] for _ in range(3)]
        ptrs = [(C.c_void_p * self.num_lookups)(*[(v.data_ptr() if _is_torch(v) else v.ctypes.data) for v in vs]) for vs in outs]
        L.check(L.lib().srs_lookup_coeff_1(self._h, addr, r.ctypes.data, space, _stream(), ptrs[0], ptrs[1], ptrs[2]))
        return tuple(outs)

    def lookup_coeff_2(self, ls, ts, ms, r):
        """ArgumentCoefficient1::evaluate_coefficient_2 (lookup.rs:350-365) -> (hs, gs)."""
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        hs, gs = [], []
        for l, t, m in zip(ls, ts, ms):
            # every keepalive stays bound until the C call below has returned (a converted temporary must outlive it)
            (al, space, n, keep_l), (at, s2, n2, keep_t), (am, s3, n3, keep_m) = _buf(l, 4), _buf(t, 4), _buf(m, 4)
            assert space == s2 == s3 and n == n2 == n3
            h, g = _alloc_like(l, n), _alloc_like(l, n)
            L.check(L.lib().srs_lookup_coeff_2(self.field, al, at, am, r.ctypes.data, n, space, _stream(),
                                               h.data_ptr() if _is_torch(h) else h.ctypes.data,
                                               g.data_ptr() if _is_torch(g) else g.ctypes.data))
            hs.append(h)
            gs.append(g)
        return hs, gs

    def is_sat_log_derivative(self, W):
        """Number of lookups violating sum_row (h_i - g_i) == 0 (PlonkStructure::is_sat_log_derivative,
        src/plonk/mod.rs:366-398); 0 <=> satisfied."""
        W = _cat_rounds(W)
        addr, space, n, keep = _buf(W, 4)
        assert n == self.num_witness_columns * self.rows
        cnt = C.c_size_t()
        L.check(L.lib().srs_is_sat_log_derivative(self._h, addr, space, _stream(), C.byref(cnt)))
        return cnt.value


def batch_invert_assigned(field, numerators, denominators, has_denominator=None):
    """`util::batch_invert_assigned` (src/util/mod.rs:119-153) for one flattened column of `Assigned<F>` cells:
    numerators[i] * denominators[i]^-1 where has_denominator[i] (all cells when None), numerators[i] otherwise; 1/0 := 0."""
    an, space, n, keep_n = _buf(numerators, 4)        # distinct names: both conversions must outlive the C call
    ad, s2, n2, keep_d = _buf(denominators, 4)
    assert space == s2 and n == n2
    out = _alloc_like(numerators, n)
    hp, keep = None, None
    if has_denominator is not None:
        if _is_torch(has_denominator):
            assert has_denominator.dtype.itemsize == 1 and has_denominator.numel() == n
            keep = has_denominator.contiguous()
            hp = keep.data_ptr()
        else:
            keep = np.ascontiguousarray(has_denominator, dtype=np.uint8)
            assert keep.shape == (n,)
            hp = keep.ctypes.data
    L.check(L.lib().srs_batch_invert_assigned(field, an, ad, hp, n, space, _stream(),
                                              out.data_ptr() if _is_torch(out) else out.ctypes.data))
    return out


class SparseMatrix:
    """`SparseMatrix<F>` = Vec<(row, col, value)> of an n x n matrix (src/polynomial/sparse.rs:5), device resident.
    The reference builds it from the copy constraints (PermutationData::matrix, src/plonk/permutation.rs)."""

    def __init__(self, field, n, rows, cols, values):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        values = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        assert rows.shape == cols.shape == (values.shape[0],)
        self.field, self.n = field, n
        h = C.c_void_p()
        L.check(L.lib().srs_sparse_create(field, n, rows.ctypes.data, cols.ctypes.data, values.ctypes.data, rows.shape[0], C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            L.lib().srs_sparse_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def matrix_multiply(self, Z):
        """sparse::matrix_multiply (sparse.rs:7-19)."""
        addr, space, n, keep = _buf(Z, 4)
        assert n == self.n
        y = _alloc_like(Z, n)
        L.check(L.lib().srs_sparse_matvec(self._h, addr, space, _stream(), y.data_ptr() if _is_torch(y) else y.ctypes.data))
        return y


class VanillaFS:
    """Sangria NIFS prover pieces (src/nifs/sangria/mod.rs)."""

    @staticmethod
    def is_sat_permutation(P, Z):
        """Mismatch count of the copy-constraint check: #{row : (P Z)[row] != Z[row]}  (src/nifs/sangria/mod.rs:425-452)."""
        addr, space, n, keep = _buf(Z, 4)
        assert n == P.n
        cnt = C.c_size_t()
        L.check(L.lib().srs_is_sat_permutation(P._h, addr, space, _stream(), C.byref(cnt)))
        return cnt.value

    @staticmethod
    def is_sat_witness_commit(ck, W, W_commitments, E=None, E_commitment=None):
        """-> (number of rounds with ck.commit(W[i]) != W_commitments[i], E mismatch flag)  (src/nifs/sangria/mod.rs:455-474)."""
        bufs = [_buf(w, 4) for w in W]
        space = bufs[0][1] if bufs else L.SPACE_HOST
        assert all(b[1] == space for b in bufs)
        wp = (C.c_void_p * max(len(bufs), 1))(*[b[0] for b in bufs])
        wn = (C.c_size_t * max(len(bufs), 1))(*[b[2] for b in bufs])
        cm = np.ascontiguousarray(W_commitments, dtype=np.uint64).reshape(-1, 8)
        assert cm.shape[0] == len(bufs)
        ea = en = ec = None
        keep = None
        if E is not None:
            ea, espace, en, keep = _buf(E, 4)
            assert espace == space
            ec = np.ascontiguousarray(E_commitment, dtype=np.uint64).reshape(8)
        bad, ebad = C.c_size_t(), C.c_int()
        L.check(L.lib().srs_is_sat_witness_commit(ck._h, wp, wn, len(bufs), cm.ctypes.data, ea, en or 0,
                                                  ec.ctypes.data if ec is not None else None, space, _stream(),
                                                  C.byref(bad), C.byref(ebad)))
        return bad.value, bool(ebad.value)

    @staticmethod
    def cross_term_challenges(U1_challenges, U1_u, U2_challenges, field):
        """concat_vec!(U1.challenges, [U1.u], U2.challenges, [DEFAULT_u = 1])  (src/nifs/sangria/mod.rs:113-118)."""
        from .field import to_mont
        parts = [np.asarray(U1_challenges, dtype=np.uint64).reshape(-1, 4), np.asarray(U1_u, dtype=np.uint64).reshape(1, 4),
                 np.asarray(U2_challenges, dtype=np.uint64).reshape(-1, 4), to_mont(field, 1).reshape(1, 4)]
        return np.ascontiguousarray(np.concatenate(parts, axis=0))

    @staticmethod
    def commit_cross_terms(ck, S, U1_challenges, U1_u, W1, U2_challenges, W2, want_terms=True):
        """-> (cross_terms: list of d vectors, cross_term_commits: (d, 8) affine points).
        With ck=None only the evaluation half runs (commits = None)."""
        ch = VanillaFS.cross_term_challenges(U1_challenges, U1_u, U2_challenges, S.field)
        W1, W2 = _cat_rounds(W1), _cat_rounds(W2)
        a1, space, n1, k1 = _buf(W1, 4)
        a2, space2, n2, k2 = _buf(W2, 4)
        assert space == space2 and n1 == n2 == S.num_witness_columns * S.rows
        d = S.num_cross_terms
        terms = [_alloc_like(W1, S.rows) for _ in range(d)] if (want_terms or ck is None) else None
        tp = None
        if terms is not None:
            tp = (C.c_void_p * max(d, 1))(*[(t.data_ptr() if _is_torch(t) else t.ctypes.data) for t in terms])
        if ck is None:
            L.check(L.lib().srs_cross_terms(S._h, a1, a2, ch.ctypes.data, ch.shape[0], space, _stream(), tp))
            return terms, None
        commits = np.zeros((d, 8), dtype=np.uint64)
        L.check(L.lib().srs_commit_cross_terms(S._h, ck._h, a1, a2, ch.ctypes.data, ch.shape[0], space, _stream(), tp,
                                               commits.ctypes.data))
        return terms, commits


def sangria_prove(ck, S, U1_challenges, U1_u, W1, U2_challenges, W2, E, W_commitments, E_commitment, r=None, ro=None,
                  incoming=False, incoming_host=None, u2_tail=None):
    """`VanillaFS::prove` (src/nifs/sangria/mod.rs:253-277) as one library call (srs_sangria_prove) on device-resident traces:
    W1 and E are folded IN PLACE.  r: the challenge (when `ro` is None) -- otherwise squeezed from `ro` (a PoseidonHash over the
    curve's base field that has absorbed pp_digest, U1, U2) after the cross-term commitments.
    incoming=True (srs_sangria_prove_incoming): W2 is a freshly synthesised trace without a commitment -- `incoming_host` (if
    given) is uploaded into W2, its commitment is computed in the same batched MSM as the cross terms' and returned as
    `incoming_commitment`; W_commitments = U1's only; `ro` holds pp_digest and U1, the call absorbs the new commitment, `u2_tail`
    (oracle-field elements) and the cross-term commitments.
    -> dict(terms, commits, r, W_commitment: PendingPoint, E_commitment: PendingPoint[, incoming_commitment])."""
    from .commitment import PendingPoint
    ch = VanillaFS.cross_term_challenges(U1_challenges, U1_u, U2_challenges, S.field)
    a1, sp1, n1, k1 = _buf(W1, 4)
    a2, sp2, n2, k2 = _buf(W2, 4)
    ae, spe, ne, ke = _buf(E, 4)
    assert n1 == n2 == S.num_witness_columns * S.rows and ne == S.rows
    d = S.num_cross_terms
    terms = [_alloc_like(W1, S.rows) for _ in range(d)]
    tp = (C.c_void_p * max(d, 1))(*[(t.data_ptr() if _is_torch(t) else t.ctypes.data) for t in terms])
    commits = np.zeros((d, 8), dtype=np.uint64)
    rr = np.zeros(4, dtype=np.uint64) if r is None else np.ascontiguousarray(r, dtype=np.uint64).reshape(4).copy()
    wc = np.zeros((2, 8), dtype=np.uint64)
    wc[: (1 if incoming else 2)] = np.ascontiguousarray(W_commitments, dtype=np.uint64).reshape(-1, 8)[: (1 if incoming else 2)]
    ec = np.ascontiguousarray(E_commitment, dtype=np.uint64).reshape(8)
    folded = np.zeros((2, 8), dtype=np.uint64)
    jobs = (C.c_uint64 * 2)()
    lib = L.lib()
    roh = None if ro is None else ro._h
    if incoming:
        hw = None
        if incoming_host is not None:
            hw = np.ascontiguousarray(incoming_host, dtype=np.uint64).reshape(-1, 4)
            assert hw.shape[0] == n2
        tail = None if u2_tail is None else np.ascontiguousarray(u2_tail, dtype=np.uint64).reshape(-1, 4)
        L.check(lib.srs_sangria_prove_incoming(S._h, ck._h, roh, ch.ctypes.data, ch.shape[0], a1, a2, None if hw is None else hw.ctypes.data,
                                               None if tail is None else tail.ctypes.data, 0 if tail is None else tail.shape[0], ae, _stream(),
                                               rr.ctypes.data, tp, commits.ctypes.data, wc.ctypes.data, ec.ctypes.data, folded.ctypes.data, jobs))
    else:
        L.check(lib.srs_sangria_prove(S._h, ck._h, roh, ch.ctypes.data, ch.shape[0], a1, a2, ae, _stream(), rr.ctypes.data, tp,
                                      commits.ctypes.data, wc.ctypes.data, ec.ctypes.data, folded.ctypes.data, jobs))
    out = dict(terms=terms, commits=commits, r=rr, W_commitment=PendingPoint(jobs[0], folded[0], lib), E_commitment=PendingPoint(jobs[1], folded[1], lib),
               _keep=(folded, wc, ec))
    if incoming:
        out["incoming_commitment"] = wc[1].copy()
    return out


class RelaxedPlonkWitness:
    """{ W: Vec<Vec<F>>, E: Box<[F]> }  (src/nifs/sangria/accumulator.rs:485-489)."""

    def __init__(self, field, W, E):
        self.field, self.W, self.E = field, list(W), E

    def fold(self, W2, cross_terms, r):
        """W'[j] = W[j] + r*W2[j];  E' = E + sum_k r^(k+1) T_k  (src/nifs/sangria/accumulator.rs:364-404)."""
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        newW = []
        for w1, w2 in zip(self.W, W2):
            a1, space, n, keep1 = _buf(w1, 4)             # distinct names: both conversions must outlive the C call
            a2, space2, n2, keep2 = _buf(w2, 4)
            assert space == space2 and n == n2
            out = _alloc_like(w1, n)
            L.check(L.lib().srs_fold_witness(self.field, out.data_ptr() if _is_torch(out) else out.ctypes.data, a1, a2,
                                             r.ctypes.data, n, space, _stream()))
            newW.append(out)
        ae, space, n, keep_e = _buf(self.E, 4)
        tb = [_buf(t, 4) for t in cross_terms]
        assert all(b[1] == space and b[2] == n for b in tb)
        tp = (C.c_void_p * max(len(tb), 1))(*[b[0] for b in tb])
        E = _alloc_like(self.E, n)
        L.check(L.lib().srs_fold_error(self.field, E.data_ptr() if _is_torch(E) else E.ctypes.data, ae, tp, len(tb),
                                       r.ctypes.data, n, space, _stream()))
        return RelaxedPlonkWitness(self.field, newW, E)
