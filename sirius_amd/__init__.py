"""sirius_amd -- MI355X-native folding-prover hot path for snarkify/sirius.

Host-side mirror of the reference interfaces this library replaces (names follow the reference):
  CommitmentKey.commit           <- src/commitment.rs:81-90
  fft.{fft,ifft,coset_fft,coset_ifft} <- src/fft.rs:160-198
  plonk.PlonkStructure / VanillaFS.commit_cross_terms / RelaxedPlonkWitness.fold
                                 <- src/plonk/mod.rs:127-157, src/nifs/sangria/mod.rs:102-158, accumulator.rs:364-404
Arrays are numpy uint64 `(n, 4)` field elements / `(n, 8)` affine points (Montgomery, LE limbs)
or torch uint64/int64 CUDA tensors of the same shape for data already resident in HBM.
"""
from . import _lib
from ._lib import (CURVE_BN256, CURVE_GRUMPKIN, FIELD_FQ, FIELD_FR, SiriusAmdError)  # noqa: F401
from .commitment import CommitmentKey, HostBuffer, PendingPoint, concatenate_with_padding, TooLongInput, point_lincomb, point_lincomb_async, point_mul, point_sum  # noqa: F401
from . import fft  # noqa: F401,E402
from . import distributed, expression, field, plonk, poseidon, protogalaxy  # noqa: F401,E402
from .poseidon import PoseidonHash  # noqa: F401,E402
from .plonk import PlonkStructure, RelaxedPlonkWitness, SparseMatrix, VanillaFS, batch_invert_assigned, sangria_prove  # noqa: F401,E402


def init_thread(device):
    """bind the CALLING host thread to a device (srs_init_thread; -1: back to the process's device)"""
    _lib.check(_lib.lib().srs_init_thread(int(device)))


def profile_enable(on=True):
    _lib.lib().srs_profile_enable(1 if on else 0)


def profile_reset():
    _lib.lib().srs_profile_reset()


def profile_sampling(every=1):
    """events on every `every`-th timed launch of a name only (srs_profile_sampling)"""
    _lib.lib().srs_profile_sampling(int(every))


def profile_get(name):
    """-> dict(total_ms, launches, units) or None."""
    import ctypes as _C
    ms, n, u = _C.c_double(), _C.c_uint64(), _C.c_uint64()
    if _lib.lib().srs_profile_get(name.encode(), _C.byref(ms), _C.byref(n), _C.byref(u)) != 0:
        return None
    return dict(total_ms=ms.value, launches=n.value, units=u.value)


class tuning:
    """`with sirius_amd.tuning(msm_sort=2, commit_chunks=3): ...` -- run-time tunables of the library (srs_tuning_set, csrc/tuning.h):
    each selects among code paths that are the default for SOME input size; none changes a result.  Restored on exit."""
    UNSET = -(1 << 63)

    def __init__(self, **kv):
        self.kv = kv
        self.old = {}

    @staticmethod
    def names():
        out, i = [], 0
        while True:
            n = _lib.lib().srs_tuning_name(i)
            if n is None:
                return out
            out.append(n.decode())
            i += 1

    @staticmethod
    def get(name):
        import ctypes as _C
        v = _C.c_int64()
        _lib.check(_lib.lib().srs_tuning_get(name.encode(), _C.byref(v)))
        return None if v.value == tuning.UNSET else v.value

    @staticmethod
    def set(name, value):
        _lib.check(_lib.lib().srs_tuning_set(name.encode(), tuning.UNSET if value is None else int(value)))

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = tuning.get(k)
            tuning.set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tuning.set(k, v)
        return False
