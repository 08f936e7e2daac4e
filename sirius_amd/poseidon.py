"""`PoseidonHash` -- host-side mirror of the reference's off-circuit random oracle (src/poseidon/poseidon_hash.rs:108-237,
trait ROTrait src/poseidon/mod.rs).  Host code inside the library: usable without a device."""
import ctypes as C

import numpy as np

from . import _lib as L


class PoseidonHash:
    def __init__(self, field, t=3, rate=2, r_f=4, r_p=3):
        h = C.c_void_p()
        L.check(L.lib().srs_poseidon_new(field, t, rate, r_f, r_p, C.byref(h)))
        self._h, self.field = h, field

    def close(self):
        if getattr(self, "_h", None):
            L.lib().srs_poseidon_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        """a fresh oracle with the same constants (the reference constructs a new RO per use)"""
        L.lib().srs_poseidon_reset(self._h)
        return self

    def absorb_field(self, v):
        """one element or an (n, 4) array of elements of the oracle's field (Montgomery)"""
        a = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        L.check(L.lib().srs_poseidon_absorb_field(self._h, a.ctypes.data, a.shape[0]))
        return self

    absorb_field_iter = absorb_field

    def absorb_point(self, curve, point):
        p = np.ascontiguousarray(point, dtype=np.uint64).reshape(8)
        L.check(L.lib().srs_poseidon_absorb_point(self._h, curve, p.ctypes.data))
        return self

    def squeeze(self, num_bits, out_field):
        out = np.zeros(4, dtype=np.uint64)
        L.check(L.lib().srs_poseidon_squeeze(self._h, num_bits, out_field, out.ctypes.data))
        return out

    def squeeze_device(self, num_bits, out_field):
        """the same squeeze with the sponge run on the device (one wavefront) -> (value, kernel milliseconds); slower than the
        host code by construction (a permutation is a chain) -- kept for the measured comparison"""
        out = np.zeros(4, dtype=np.uint64)
        ms = C.c_double()
        L.check(L.lib().srs_poseidon_squeeze_device(self._h, num_bits, out_field, out.ctypes.data, C.byref(ms)))
        return out, ms.value
