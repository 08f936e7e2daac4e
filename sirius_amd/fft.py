"""`fft`, `ifft`, `coset_fft`, `coset_ifft` -- mirror of reference src/fft.rs:160-198.

In place like the reference (`a: &mut [F]`): numpy arrays / torch tensors are modified and returned.
Asserts of the reference surface as exceptions: NotPowerOfTwo (src/fft.rs:161,169) and
KTooLarge (src/fft.rs:13)."""
import numpy as np

from . import _lib as L
from .commitment import _is_torch, _stream


class NotPowerOfTwo(AssertionError):
    pass


class KTooLarge(AssertionError):
    pass


def _run(a, inverse, coset):
    if _is_torch(a):
        assert a.is_contiguous() and a.shape[-1] == 4 and a.element_size() == 8 and a.dim() == 2
        addr, space, n = a.data_ptr(), (L.SPACE_DEVICE if a.is_cuda else L.SPACE_HOST), a.shape[0]
    else:
        assert isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags.c_contiguous and a.ndim == 2 and a.shape[1] == 4
        addr, space, n = a.ctypes.data, L.SPACE_HOST, a.shape[0]
    rc = L.lib().srs_ntt(L.FIELD_FR, addr, n, inverse, coset, space, _stream())
    if rc == L.ERR_NOT_POW2:
        raise NotPowerOfTwo("a.len().is_power_of_two()")
    if rc == L.ERR_K_TOO_LARGE:
        raise KTooLarge(L.lib().srs_last_error().decode())
    L.check(rc)
    return a


def fft(a): return _run(a, 0, 0)
def ifft(a): return _run(a, 1, 0)
def coset_fft(a): return _run(a, 0, 1)
def coset_ifft(a): return _run(a, 1, 1)
