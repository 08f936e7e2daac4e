"""`CommitmentKey` -- mirror of reference src/commitment.rs:29-90 on top of the C-ABI."""
import ctypes as C

import numpy as np

from . import _lib as L


class TooLongInput(ValueError):
    """commitment::Error::TooLongInput { input_len, limit }  (src/commitment.rs:23-27)."""

    def __init__(self, input_len, limit):
        super().__init__(f"Can't commit too long input: input len: {input_len}, but limit is {limit}")
        self.input_len, self.limit = input_len, limit


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _buf(x, last):
    """-> (address, space, n_rows, keepalive)."""
    if _is_torch(x):
        assert x.is_contiguous() and x.shape[-1] == last and x.element_size() == 8, (x.shape, x.dtype)
        space = L.SPACE_DEVICE if x.is_cuda else L.SPACE_HOST
        return x.data_ptr(), space, x.numel() // last, x
    a = np.ascontiguousarray(x, dtype=np.uint64)
    assert a.shape[-1] == last, a.shape
    return a.ctypes.data, L.SPACE_HOST, a.size // last, a


_torch_cuda = None      # torch.cuda once it is known to be usable (is_available() costs ~50-90 us per call on ROCm)


def _stream():
    """The caller's current torch stream (the kernels are enqueued behind the caller's own work), or the default stream."""
    global _torch_cuda
    if _torch_cuda is None:
        try:
            import torch
            if not torch.cuda.is_available():
                return None
            _torch_cuda = torch.cuda
        except Exception:
            return None
    return _torch_cuda.current_stream().cuda_stream


class CommitmentKey:
    """Device-resident Pedersen commitment key (window-expanded in HBM)."""

    def __init__(self, curve, bases, rank=0, world=1):
        addr, space, n, keep = _buf(bases, 8)
        self.curve, self._len = curve, n
        self.rank, self.world = rank, world
        h = C.c_void_p()
        L.check(L.lib().srs_ck_create_sharded(curve, addr, n, space, rank, world, C.byref(h)))
        self._h = h

    @classmethod
    def setup_synthetic(cls, curve, n, seed=0, rank=0, world=1):
        """Device-generated key P_i = [h(seed, i)]G for tests/benches -- NOT CommitmentKey::setup
        (src/commitment.rs:55-79, hash_to_curve is un-vendored third-party code)."""
        self = cls.__new__(cls)
        self.curve, self._len, self.rank, self.world = curve, n, rank, world
        h = C.c_void_p()
        L.check(L.lib().srs_ck_setup_synthetic(curve, n, seed, rank, world, C.byref(h)))
        self._h = h
        return self

    @staticmethod
    def setup_uniform_bytes(label, first, count):
        """The 32-byte chunks `first .. first + count` of CommitmentKey::setup's SHAKE256(label) stream (src/commitment.rs:61-67)
        as a (count, 32) uint8 array."""
        out = np.zeros((count, 32), dtype=np.uint8)
        L.check(L.lib().srs_ck_setup_uniform_bytes(bytes(label), len(label), first, count, out.ctypes.data))
        return out

    @classmethod
    def setup(cls, curve, k, label):
        """`CommitmentKey::setup` (src/commitment.rs:55-79): NOT available -- the chunk -> point map is halo2curves'
        hash_to_curve (third party, unpinned); raises SiriusAmdError(rc = ERR_UNSUPPORTED).  Use load_from_file / __init__."""
        assert k < 32
        h = C.c_void_p()
        L.check(L.lib().srs_ck_setup(curve, k, bytes(label), len(label), C.byref(h)))
        raise AssertionError("srs_ck_setup returned OK")

    @classmethod
    def create_multi(cls, curve, bases, n_devices=0):
        """One process, several GPUs (srs_ck_create_multi): the library spreads the key over `n_devices` shards (0 = all
        visible devices; more shards than devices are folded onto the devices round-robin) and partitions the scalars of
        every commit the same way.  The handle behaves like an ordinary key: commits return the full commitment."""
        addr, space, n, keep = _buf(bases, 8)
        self = cls.__new__(cls)
        self.curve, self._len, self.rank, self.world = curve, n, 0, 1
        h = C.c_void_p()
        L.check(L.lib().srs_ck_create_multi(curve, addr, n, space, n_devices, C.byref(h)))
        self._h = h
        return self

    @classmethod
    def setup_synthetic_multi(cls, curve, n, seed=0, n_devices=0):
        self = cls.__new__(cls)
        self.curve, self._len, self.rank, self.world = curve, n, 0, 1
        h = C.c_void_p()
        L.check(L.lib().srs_ck_setup_synthetic_multi(curve, n, seed, n_devices, C.byref(h)))
        self._h = h
        return self

    @property
    def num_shards(self):
        return L.lib().srs_ck_num_shards(self._h)

    def msm_stats(self):
        """Diagnostics of the MSM engine (srs_ck_msm_stats): dict(slot_sets, hot_sets, redo, other_sets)."""
        out = (C.c_uint64 * 4)()
        L.check(L.lib().srs_ck_msm_stats(self._h, out))
        return dict(slot_sets=int(out[0]), hot_sets=int(out[1]), redo=int(out[2]), other_sets=int(out[3]))

    def shard_stats(self, shard):
        """dict(h2d_bytes, peer_bytes, streamed_commits, device) of one shard of a multi-device key (srs_ck_shard_stats)."""
        out = (C.c_uint64 * 4)()
        L.check(L.lib().srs_ck_shard_stats(self._h, shard, out))
        return dict(h2d_bytes=int(out[0]), peer_bytes=int(out[1]), streamed_commits=int(out[2]), device=int(out[3]))

    def has_wide_table(self):
        """True when the key holds the second (20-bit-window) table (srs_ck_has_wide_table)."""
        return bool(L.lib().srs_ck_has_wide_table(self._h))

    @classmethod
    def load_from_file(cls, curve, file_path, k, rank=0, world=1):
        """`CommitmentKey::load_from_file` + the on-curve validation of `load_or_setup_cache`
        (src/commitment.rs:112-160): raw `[C; 2^k]` dump; IOError on a short/missing file,
        ValueError("Wrong file in cache, some ptr out of curve") on invalid points."""
        self = cls.__new__(cls)
        self.curve, self._len, self.rank, self.world = curve, 1 << k, rank, world
        h = C.c_void_p()
        rc = L.lib().srs_ck_load_file(curve, str(file_path).encode(), k, rank, world, C.byref(h))
        if rc == L.ERR_IO:
            raise IOError(L.lib().srs_last_error().decode())
        if rc == L.ERR_INVALID_DATA:
            raise ValueError(L.lib().srs_last_error().decode())
        L.check(rc)
        self._h = h
        return self

    def save_to_file(self, file_path):
        """`CommitmentKey::save_to_file` (src/commitment.rs:99-104)."""
        rc = L.lib().srs_ck_save_file(self._h, str(file_path).encode())
        if rc == L.ERR_IO:
            raise IOError(L.lib().srs_last_error().decode())
        L.check(rc)

    @classmethod
    def load_or_setup_cache(cls, curve, cache_folder, label, k, setup=None):
        """`load_or_setup_cache` (src/commitment.rs:137-170): `{cache}/{label}/{k}.bin`.  When the file is
        missing the reference runs `setup` (SHAKE256 + hash_to_curve, un-vendored third-party code); here the
        caller supplies `setup(k) -> CommitmentKey` (e.g. setup_synthetic) and the result is saved."""
        import os
        path = os.path.join(str(cache_folder), label, f"{k}.bin")
        if os.path.exists(path):
            return cls.load_from_file(curve, path, k)
        if setup is None:
            raise FileNotFoundError(path)
        key = setup(k)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        key.save_to_file(path)
        return key

    def count_off_curve(self):
        n = C.c_size_t()
        L.check(L.lib().srs_ck_count_off_curve(self._h, C.byref(n)))
        return n.value

    def bases(self):
        """This rank's bases as (local_len, 8) uint64."""
        n = L.lib().srs_ck_local_len(self._h)
        out = np.zeros((n, 8), dtype=np.uint64)
        L.check(L.lib().srs_ck_get_bases(self._h, out.ctypes.data))
        return out

    def __len__(self):
        return self._len

    def close(self):
        if getattr(self, "_h", None):
            L.lib().srs_ck_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def commit(self, v, repr=L.REPR_MONT):
        """sum v[i] * ck[i] -> affine (8,) uint64.  Raises TooLongInput like the reference."""
        addr, space, n, keep = _buf(v, 4)
        if n > self._len:
            raise TooLongInput(n, self._len)
        out = np.zeros(8, dtype=np.uint64)
        L.check(L.lib().srs_commit(self._h, addr, n, space, repr, _stream(), out.ctypes.data))
        return out

    def commit_upload(self, v_host, dev_copy=None, repr=L.REPR_MONT):
        """`ck.commit(&W)` for a witness fresh from the host that also leaves a device copy (srs_commit_upload): chunked
        upload overlapped with the MSM of the chunks already in HBM.  v_host: numpy (n, 4) uint64 (ideally page-locked,
        `HostBuffer`); dev_copy: torch CUDA tensor of the same shape or None."""
        a = np.ascontiguousarray(v_host, dtype=np.uint64)
        assert a.shape[-1] == 4
        n = a.size // 4
        if n > self._len:
            raise TooLongInput(n, self._len)
        dptr = None
        if dev_copy is not None:
            assert _is_torch(dev_copy) and dev_copy.is_contiguous() and dev_copy.numel() == 4 * n
            dptr = dev_copy.data_ptr()
        out = np.zeros(8, dtype=np.uint64)
        L.check(L.lib().srs_commit_upload(self._h, a.ctypes.data, n, dptr, repr, _stream(), out.ctypes.data))
        return out

    def commit_upload_columns(self, columns, pad_size, dev_copy=None, repr=L.REPR_MONT):
        """`ck.commit(&concatenate_with_padding(advice, pad_size))` (run_sps_protocol_*, src/plonk/mod.rs:441-447) straight from
        the per-column host vectors: the concatenated, zero-padded witness is assembled in HBM (srs_commit_upload_columns),
        groups of columns go up while the MSM of the previous group runs.  dev_copy: torch tensor of concat_len elements."""
        cols = [np.ascontiguousarray(c, dtype=np.uint64).reshape(-1, 4) for c in columns]
        lens = (C.c_size_t * max(len(cols), 1))(*[c.shape[0] for c in cols])
        ptrs = (C.c_void_p * max(len(cols), 1))(*[c.ctypes.data for c in cols])
        n = L.lib().srs_concat_len(lens, len(cols), pad_size)
        if n > self._len:
            raise TooLongInput(n, self._len)
        dptr = None
        if dev_copy is not None:
            assert _is_torch(dev_copy) and dev_copy.is_contiguous() and dev_copy.numel() == 4 * n
            dptr = dev_copy.data_ptr()
        out = np.zeros(8, dtype=np.uint64)
        L.check(L.lib().srs_commit_upload_columns(self._h, ptrs, lens, len(cols), pad_size, dptr, repr, _stream(), out.ctypes.data))
        return out

    def commit_batch(self, vs, repr=L.REPR_MONT):
        """[commit(v) for v in vs] in one set of launches (cross-term commits, sangria/mod.rs:151-154)."""
        bufs = [_buf(v, 4) for v in vs]
        if not bufs:
            return np.zeros((0, 8), dtype=np.uint64)
        spaces = {b[1] for b in bufs}
        assert len(spaces) == 1, "all vectors of a batch must live in the same memory space"
        for b in bufs:
            if b[2] > self._len:
                raise TooLongInput(b[2], self._len)
        ptrs = (C.c_void_p * len(bufs))(*[b[0] for b in bufs])
        ns = (C.c_size_t * len(bufs))(*[b[2] for b in bufs])
        out = np.zeros((len(bufs), 8), dtype=np.uint64)
        L.check(L.lib().srs_commit_batch(self._h, ptrs, ns, len(bufs), spaces.pop(), repr, _stream(), out.ctypes.data))
        return out


def point_sum(curve, points):
    a = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros(8, dtype=np.uint64)
    L.check(L.lib().srs_point_sum(curve, a.ctypes.data, a.shape[0], out.ctypes.data))
    return out


def point_lincomb(curve, acc, points, scalars, repr=L.REPR_MONT):
    """acc + sum scalars[i] * points[i] on the host (RelaxedPlonkInstance::fold, accumulator.rs:201-264)."""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    assert pts.shape[0] == sc.shape[0]
    a = None if acc is None else np.ascontiguousarray(acc, dtype=np.uint64).reshape(8)
    out = np.zeros(8, dtype=np.uint64)
    L.check(L.lib().srs_point_lincomb(curve, None if a is None else a.ctypes.data, pts.ctypes.data, sc.ctypes.data,
                                      pts.shape[0], repr, out.ctypes.data))
    return out


class PendingPoint:
    """Result of point_lincomb_async: .wait() -> the (8,) affine point (blocks until the host workers are done)."""

    def __init__(self, job, out, lib):
        self._job, self._out, self._lib = job, out, lib      # the job lives in the library instance that accepted it

    def wait(self):
        if self._job is not None:
            L.check(self._lib.srs_job_wait(self._job))
            self._job = None
        return self._out

    def __del__(self):       # never leave a job writing into freed memory
        try:
            if self._job is not None:
                self._lib.srs_job_wait(self._job)
        except Exception:
            pass


def point_lincomb_async(curve, acc, points, scalars, repr=L.REPR_MONT):
    """point_lincomb on the library's host workers, off the calling thread (the instance fold overlaps the next commitment)."""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    assert pts.shape[0] == sc.shape[0]
    a = None if acc is None else np.ascontiguousarray(acc, dtype=np.uint64).reshape(8)
    out = np.zeros(8, dtype=np.uint64)
    job = C.c_uint64()
    lib = L.lib()
    L.check(lib.srs_point_lincomb_async(curve, None if a is None else a.ctypes.data, pts.ctypes.data, sc.ctypes.data,
                                        pts.shape[0], repr, out.ctypes.data, C.byref(job)))
    return PendingPoint(job.value, out, lib)


def point_mul(curve, scalar, p, repr=L.REPR_MONT):
    s = np.ascontiguousarray(scalar, dtype=np.uint64).reshape(4)
    q = np.ascontiguousarray(p, dtype=np.uint64).reshape(8)
    out = np.zeros(8, dtype=np.uint64)
    L.check(L.lib().srs_point_mul(curve, s.ctypes.data, repr, q.ctypes.data, out.ctypes.data))
    return out


class HostBuffer:
    """Page-locked host memory from the library (srs_host_alloc) viewed as a numpy (n, 4) uint64 array: uploads from it
    are asynchronous to the caller.  Keep the object alive while `.array` is in use."""

    def __init__(self, n_fe):
        p = C.c_void_p()
        L.check(L.lib().srs_host_alloc(max(n_fe, 1) * 32, C.byref(p)))
        self._p = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n_fe, 4))

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            L.lib().srs_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def concatenate_with_padding(columns, pad_size, out):
    """`util::concatenate_with_padding` (src/util/mod.rs:214-218) from host columns into the device vector `out`
    (torch tensor of concat_len elements; host memory under the CPU emulator)."""
    cols = [np.ascontiguousarray(c, dtype=np.uint64).reshape(-1, 4) for c in columns]
    lens = (C.c_size_t * max(len(cols), 1))(*[c.shape[0] for c in cols])
    ptrs = (C.c_void_p * max(len(cols), 1))(*[c.ctypes.data for c in cols])
    n = L.lib().srs_concat_len(lens, len(cols), pad_size)
    assert out.numel() == 4 * n and out.is_contiguous()
    L.check(L.lib().srs_concat_with_padding(out.data_ptr(), ptrs, lens, len(cols), pad_size, _stream()))
    return out
