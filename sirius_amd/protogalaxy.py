"""Host-side mirror of the ProtoGalaxy prover polynomials (reference src/nifs/protogalaxy/).

  PolyContext                      <- poly/mod.rs:205-269
  compute_F / compute_G / compute_K_from_G / evaluate_e_from_trace / calculate_e
                                   <- poly/mod.rs:68-203,308-425,475-509; mod.rs:571-640,748-764
  fold_witness                     <- mod.rs:176-210
  lagrange.iter_eval_lagrange_poly_for_cyclic_group, UnivariatePoly.eval  <- lagrange.rs:50-75, univariate.rs:67-75
`reference_compat=True` (default) reproduces the reference bit for bit including its row-index quirk
(every leaf evaluated at row 0, src/plonk/mod.rs:714)."""
import ctypes as C

import numpy as np

from . import _lib as L
from .commitment import _buf, _is_torch, _stream
from .plonk import _alloc_like


class _Ctx(C.Structure):
    _fields_ = [("count_of_evaluation_with_padding", C.c_size_t), ("betas_count", C.c_size_t),
                ("fft_points_count_F", C.c_size_t), ("fft_points_count_G", C.c_size_t),
                ("instances_to_fold", C.c_size_t), ("lagrange_domain", C.c_size_t),
                ("fft_log_domain_size_K", C.c_uint32)]


class PolyContext:
    def __init__(self, S, traces_len):
        c = _Ctx()
        L.check(L.lib().srs_pg_context_new(S._h, traces_len, C.byref(c)))
        self.S = S
        for name, _ in _Ctx._fields_:
            setattr(self, name, getattr(c, name))

    def fft_points_count_K(self):
        return 1 << self.fft_log_domain_size_K


def _k_len(ctx):
    """Coefficients of K.  A domain "log" above F::S = 28 (quirk Q2 with 32 points of G or >= 8 instances) is refused by the
    library with rc 3 -- the reference's `assert!(k <= F::S)` (src/fft.rs:13) -- before anything is written: no buffer then."""
    return 1 << ctx.fft_log_domain_size_K if ctx.fft_log_domain_size_K <= 28 else 0


def _fe(a):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)


def compute_F(ctx, betas, delta, W, challenges=(), reference_compat=True):
    betas, delta, ch = _fe(betas), _fe(delta), _fe(np.zeros((0, 4), np.uint64) if len(challenges) == 0 else challenges)
    addr, space, n, keep = _buf(W, 4)
    out = np.zeros((ctx.fft_points_count_F, 4), dtype=np.uint64)
    L.check(L.lib().srs_pg_compute_F(ctx.S._h, betas.ctypes.data, betas.shape[0], delta.ctypes.data, addr, ch.ctypes.data,
                                     ch.shape[0], space, 1 if reference_compat else 0, _stream(), out.ctypes.data))
    return out


def beta_stroke(betas, alpha, delta):
    """`PolyChallenges::iter_beta_stroke` (poly/mod.rs:432-462): betas[i] + alpha * delta^(2^i)."""
    b, a, d = _fe(betas), _fe(alpha), _fe(delta)
    out = np.zeros_like(b)
    L.check(L.lib().srs_pg_beta_stroke(b.ctypes.data, b.shape[0], a.ctypes.data, d.ctypes.data, out.ctypes.data))
    return out


def compute_G(ctx, betas_stroke, Ws, challenges_list=None, reference_compat=True):
    bs = _fe(betas_stroke)
    bufs = [_buf(w, 4) for w in Ws]
    spaces = {b[1] for b in bufs}
    assert len(spaces) == 1
    J = len(bufs)
    chs = [_fe(c) for c in (challenges_list if challenges_list is not None else [np.zeros((0, 4), np.uint64)] * J)]
    wp = (C.c_void_p * J)(*[b[0] for b in bufs])
    cp = (C.c_void_p * J)(*[c.ctypes.data for c in chs])
    out = np.zeros((ctx.fft_points_count_G, 4), dtype=np.uint64)
    L.check(L.lib().srs_pg_compute_G(ctx.S._h, bs.ctypes.data, bs.shape[0], wp, cp, chs[0].shape[0], J, spaces.pop(),
                                     1 if reference_compat else 0, _stream(), out.ctypes.data))
    return out


def compute_K_from_G(ctx, poly_G, poly_F_in_alpha):
    g, fa = _fe(poly_G), _fe(poly_F_in_alpha)
    out = np.zeros((_k_len(ctx), 4), dtype=np.uint64)
    L.check(L.lib().srs_pg_compute_K_from_G(g.ctypes.data, g.shape[0], fa.ctypes.data, ctx.instances_to_fold,
                                            ctx.fft_log_domain_size_K, _stream(), out.ctypes.data))
    return out


def evaluate_e_from_trace(ctx, betas, W, challenges=(), reference_compat=True):
    betas, ch = _fe(betas), _fe(np.zeros((0, 4), np.uint64) if len(challenges) == 0 else challenges)
    addr, space, n, keep = _buf(W, 4)
    out = np.zeros(4, dtype=np.uint64)
    L.check(L.lib().srs_pg_evaluate_e(ctx.S._h, betas.ctypes.data, betas.shape[0], addr, ch.ctypes.data, ch.shape[0], space,
                                      1 if reference_compat else 0, _stream(), out.ctypes.data))
    return out


def calculate_e(poly_F, poly_K, gamma, alpha, log_n):
    f, k, g, a = _fe(poly_F), _fe(poly_K), _fe(gamma), _fe(alpha)
    out = np.zeros(4, dtype=np.uint64)
    L.check(L.lib().srs_pg_calculate_e(f.ctypes.data, f.shape[0], k.ctypes.data, k.shape[0], g.ctypes.data, a.ctypes.data, log_n, out.ctypes.data))
    return out


def eval_lagrange_poly_for_cyclic_group(X, log_n):
    x = _fe(X)
    out = np.zeros((1 << log_n, 4), dtype=np.uint64)
    L.check(L.lib().srs_lagrange_eval(x.ctypes.data, log_n, out.ctypes.data))
    return out


def poly_eval(coeffs, x):
    c, xx = _fe(coeffs), _fe(x)
    out = np.zeros(4, dtype=np.uint64)
    L.check(L.lib().srs_poly_eval(c.ctypes.data, c.shape[0], xx.ctypes.data, out.ctypes.data))
    return out


def fold_witness(field, Ws, lagrange_for_gamma, out=None, shard=None, structure=None, reference_compat=True):
    """W' = sum_j L_j(gamma) * W_j   (ProtoGalaxy::fold_witness).  shard = (rank, world): device vectors, only the rank's
    block-cyclic stripes of `out` are computed (srs_fold_lincomb_sharded; `out` may be one of the inputs).  structure = a
    row-sharded PlonkStructure: its stripes AND the halo rows its kernels read beyond them (srs_structure_fold_sharded)."""
    bufs = [_buf(w, 4) for w in Ws]
    spaces = {b[1] for b in bufs}
    assert len(spaces) == 1 and len({b[2] for b in bufs}) == 1
    coefs = _fe(lagrange_for_gamma)[: len(bufs)].copy()
    wp = (C.c_void_p * len(bufs))(*[b[0] for b in bufs])
    if out is None:
        out = _alloc_like(Ws[0], bufs[0][2])
    if structure is not None:
        L.check(L.lib().srs_structure_fold_sharded(structure._h, out.data_ptr() if _is_torch(out) else out.ctypes.data, wp, coefs.ctypes.data,
                                                   len(bufs), int(bool(reference_compat)), _stream()))
        return out
    if shard is not None and shard[1] > 1:
        L.check(L.lib().srs_fold_lincomb_sharded(field, out.data_ptr() if _is_torch(out) else out.ctypes.data, wp, coefs.ctypes.data,
                                                 len(bufs), bufs[0][2], shard[0], shard[1], _stream()))
        return out
    L.check(L.lib().srs_fold_lincomb(field, out.data_ptr() if _is_torch(out) else out.ctypes.data, wp, coefs.ctypes.data,
                                     len(bufs), bufs[0][2], spaces.pop(), _stream()))
    return out


def prove(ctx, betas, delta, Ws, ro=None, alpha=None, gamma=None, challenges_list=None, reference_compat=True, fold=True):
    """`ProtoGalaxy::prove` (src/nifs/protogalaxy/mod.rs:400-481) as one library call (srs_pg_prove) on device-resident
    witnesses Ws = [accumulator, incoming...].  With `ro` (a PoseidonHash over Fr holding the transcript so far) alpha and gamma
    are squeezed inside; otherwise they are given.  fold=False: W is None and fold_witness(field, Ws, lagrange) is the caller's to
    queue later.  -> dict(alpha, gamma, poly_F, poly_K, betas_stroke, e, lagrange, W)."""
    bs = _fe(betas)
    bufs = [_buf(w, 4) for w in Ws]
    assert all(b[1] == L.SPACE_DEVICE for b in bufs) or not _is_torch(Ws[0]) or not Ws[0].is_cuda, "device-resident witnesses"
    J = len(bufs)
    chs = [_fe(c) for c in (challenges_list if challenges_list is not None else [np.zeros((0, 4), np.uint64)] * J)]
    wp = (C.c_void_p * J)(*[b[0] for b in bufs])
    cp = (C.c_void_p * J)(*[c.ctypes.data for c in chs])
    ag = np.zeros((2, 4), dtype=np.uint64)
    if ro is None:
        ag[0], ag[1] = _fe(alpha).reshape(4), _fe(gamma).reshape(4)
    out = dict(poly_F=np.zeros((ctx.fft_points_count_F, 4), np.uint64), poly_K=np.zeros((_k_len(ctx), 4), np.uint64),
               betas_stroke=np.zeros((ctx.betas_count, 4), np.uint64), e=np.zeros(4, np.uint64), lagrange=np.zeros((J, 4), np.uint64),
               W=_alloc_like(Ws[0], bufs[0][2]) if fold else None)
    d = _fe(delta)
    Wf = out["W"]
    L.check(L.lib().srs_pg_prove(ctx.S._h, None if ro is None else ro._h, bs.ctypes.data, bs.shape[0], d.ctypes.data, wp, cp, chs[0].shape[0], J,
                                 1 if reference_compat else 0, _stream(), ag.ctypes.data, out["poly_F"].ctypes.data, out["poly_K"].ctypes.data,
                                 out["betas_stroke"].ctypes.data, out["e"].ctypes.data, out["lagrange"].ctypes.data,
                                 None if Wf is None else (Wf.data_ptr() if _is_torch(Wf) else Wf.ctypes.data)))
    out["alpha"], out["gamma"] = ag[0].copy(), ag[1].copy()
    return out
