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
from .expression import serialize_gates


def _alloc_like(ref, rows):
    if _is_torch(ref):
        import torch
        return torch.empty((rows, 4), dtype=ref.dtype, device=ref.device)
    return np.zeros((rows, 4), dtype=np.uint64)


class PlonkStructure:
    """k, selectors, fixed_columns, num_advice_columns, gates  (no lookups, src/plonk/mod.rs:127-157)."""

    def __init__(self, field, k, selectors, fixed_columns, num_advice_columns, gates):
        self.field, self.k, self.num_advice_columns = field, k, num_advice_columns
        rows = 1 << k
        sel = [np.ascontiguousarray(s, dtype=np.uint8) for s in selectors]
        fix = [np.ascontiguousarray(f, dtype=np.uint64) for f in fixed_columns]
        assert all(s.shape == (rows,) for s in sel) and all(f.shape == (rows, 4) for f in fix)
        self.num_selectors, self.num_fixed = len(sel), len(fix)
        words = serialize_gates(gates, field)
        selp = (C.c_void_p * max(len(sel), 1))(*[s.ctypes.data for s in sel])
        fixp = (C.c_void_p * max(len(fix), 1))(*[f.ctypes.data for f in fix])
        h = C.c_void_p()
        L.check(L.lib().srs_structure_create(field, k, len(sel), len(fix), num_advice_columns, selp, fixp, L.SPACE_HOST,
                                             words.ctypes.data, len(words), len(gates), C.byref(h)))
        self._h = h
        self.num_challenges = L.lib().srs_structure_num_challenges(h)
        self.num_cross_terms = L.lib().srs_structure_num_cross_terms(h)

    @property
    def rows(self):
        return 1 << self.k

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
        addr, space, n, keep = _buf(W, 4)
        assert n == self.num_advice_columns * self.rows
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)
        out = _alloc_like(W, self.rows)
        oaddr = out.data_ptr() if _is_torch(out) else out.ctypes.data
        L.check(L.lib().srs_eval_gates(self._h, 1 if homogeneous else 0, addr, ch.ctypes.data, ch.shape[0], space, _stream(), oaddr))
        return out


    def is_sat_gates(self, W, challenges, E=None):
        """Mismatch count of the deciders' gate check: E is None -> compressed gate == 0 per row
        (PlonkStructure::is_sat, src/plonk/mod.rs:329-346, challenges = U.challenges); otherwise homogeneous
        gate == E[row] (is_sat_accumulation, src/nifs/sangria/mod.rs:352-376, challenges = U.challenges || U.u)."""
        addr, space, n, keep = _buf(W, 4)
        assert n == self.num_advice_columns * self.rows
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)
        eaddr = None
        if E is not None:
            eaddr, espace, en, ekeep = _buf(E, 4)
            assert espace == space and en == self.rows
        cnt = C.c_size_t()
        L.check(L.lib().srs_is_sat_gates(self._h, 0 if E is None else 1, addr, ch.ctypes.data, ch.shape[0], eaddr, space,
                                         _stream(), C.byref(cnt)))
        return cnt.value


class VanillaFS:
    """Sangria NIFS prover pieces (src/nifs/sangria/mod.rs)."""

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
        a1, space, n1, k1 = _buf(W1, 4)
        a2, space2, n2, k2 = _buf(W2, 4)
        assert space == space2 and n1 == n2 == S.num_advice_columns * S.rows
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


class RelaxedPlonkWitness:
    """{ W: Vec<Vec<F>>, E: Box<[F]> }  (src/nifs/sangria/accumulator.rs:485-489)."""

    def __init__(self, field, W, E):
        self.field, self.W, self.E = field, list(W), E

    def fold(self, W2, cross_terms, r):
        """W'[j] = W[j] + r*W2[j];  E' = E + sum_k r^(k+1) T_k  (src/nifs/sangria/accumulator.rs:364-404)."""
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        newW = []
        for w1, w2 in zip(self.W, W2):
            a1, space, n, _ = _buf(w1, 4)
            a2, space2, n2, _ = _buf(w2, 4)
            assert space == space2 and n == n2
            out = _alloc_like(w1, n)
            L.check(L.lib().srs_fold_witness(self.field, out.data_ptr() if _is_torch(out) else out.ctypes.data, a1, a2,
                                             r.ctypes.data, n, space, _stream()))
            newW.append(out)
        ae, space, n, _ = _buf(self.E, 4)
        tb = [_buf(t, 4) for t in cross_terms]
        assert all(b[1] == space and b[2] == n for b in tb)
        tp = (C.c_void_p * max(len(tb), 1))(*[b[0] for b in tb])
        E = _alloc_like(self.E, n)
        L.check(L.lib().srs_fold_error(self.field, E.data_ptr() if _is_torch(E) else E.ctypes.data, ae, tp, len(tb),
                                       r.ctypes.data, n, space, _stream()))
        return RelaxedPlonkWitness(self.field, newW, E)
