//! `CommitmentKey::commit` (sirius src/commitment.rs:81-90) and its witness-sized variants.
//!
//! In sirius, `CommitmentKey<C>` gains `gpu: OnceCell<GpuKey>` next to `ck: Box<[C]>` (`Deref<Target = [C]>` untouched) and
//! `commit` becomes:
//! ```ignore
//! pub fn commit(&self, v: &[C::Scalar]) -> Result<C, Error> {
//!     if v.len() > self.ck.len() { return Err(Error::TooLongInput { input_len: v.len(), limit: self.ck.len() }); }
//!     if !gpu_ready::<C>() { return Ok(best_multiexp(v, &self.ck[..v.len()]).to_affine()); }
//!     Ok(self.gpu.get_or_try_init(|| GpuKey::create(&self.ck))?.commit(v)?)
//! }
//! ```
use std::mem::MaybeUninit;
use std::ptr;

use crate::sys::*;
use crate::{check, GpuCurve, ShimError};

/// Device mirror of a `CommitmentKey` (window-expanded table in HBM), `Drop` frees it.
pub struct GpuKey<C: GpuCurve> { pub(crate) raw: *mut srs_ck, len: usize, _c: std::marker::PhantomData<C> }
unsafe impl<C: GpuCurve> Send for GpuKey<C> {}       // one driver thread per key at a time (include/sirius_amd.h: thread-compatible)

impl<C: GpuCurve> GpuKey<C> {
    /// `srs_ck_create`: once per key; `n_devices` = 0 keeps the key on the process's GPU, N > 0 spreads it over N GPUs inside the
    /// library (`srs_ck_create_multi`) -- the handle behaves the same either way.
    pub fn create(bases: &[C], n_devices: i32) -> Result<Self, ShimError> {
        let mut raw: *mut srs_ck = ptr::null_mut();
        let p = bases.as_ptr() as *const srs_affine;
        check(unsafe {
            if n_devices > 0 { srs_ck_create_multi(C::CURVE, p, bases.len(), SRS_SPACE_HOST, n_devices, &mut raw) }
            else { srs_ck_create(C::CURVE, p, bases.len(), SRS_SPACE_HOST, &mut raw) }
        })?;
        Ok(Self { raw, len: bases.len(), _c: Default::default() })
    }

    /// `load_from_file` + the on-curve check of `load_or_setup_cache` (src/commitment.rs:112-160)
    pub fn load_file(path: &std::path::Path, k: usize) -> std::io::Result<Self> {
        let c = std::ffi::CString::new(path.to_str().expect("utf-8 path")).unwrap();
        let mut raw: *mut srs_ck = ptr::null_mut();
        match unsafe { srs_ck_load_file(C::CURVE, c.as_ptr(), k, 0, 1, &mut raw) } {
            SRS_OK => Ok(Self { raw, len: 1 << k, _c: Default::default() }),
            SRS_ERR_INVALID_DATA => Err(std::io::Error::new(std::io::ErrorKind::InvalidData, "Wrong file in cache, some ptr out of curve")),
            _ => Err(std::io::Error::new(std::io::ErrorKind::Other, crate::last_error())),
        }
    }

    pub fn len(&self) -> usize { self.len }

    /// the body of `commit` after the reference's own length check
    pub fn commit(&self, v: &[C::ScalarExt]) -> Result<C, ShimError> {
        let mut out = MaybeUninit::<C>::uninit();
        check(unsafe { srs_commit(self.raw, v.as_ptr() as *const srs_fe, v.len(), SRS_SPACE_HOST, SRS_REPR_MONT, ptr::null_mut(),
                                  out.as_mut_ptr() as *mut srs_affine) })?;
        Ok(unsafe { out.assume_init() })
    }

    /// the d cross-term commitments over one base prefix as one set of launches (sangria/mod.rs:151-154)
    pub fn commit_batch(&self, vs: &[&[C::ScalarExt]]) -> Result<Vec<C>, ShimError> {
        let ptrs: Vec<*const srs_fe> = vs.iter().map(|v| v.as_ptr() as *const srs_fe).collect();
        let lens: Vec<usize> = vs.iter().map(|v| v.len()).collect();
        let mut out = vec![C::identity(); vs.len()];
        check(unsafe { srs_commit_batch(self.raw, ptrs.as_ptr(), lens.as_ptr(), vs.len(), SRS_SPACE_HOST, SRS_REPR_MONT, ptr::null_mut(),
                                        out.as_mut_ptr() as *mut srs_affine) })?;
        Ok(out)
    }

    /// `ck.commit(&W1)` of run_sps_protocol_* (src/plonk/mod.rs:441-447) for a witness fresh from synthesis: streamed upload
    /// overlapped with the MSM; `dev_copy` (from `DeviceVec`) keeps the vector in HBM as the incoming trace of the next prove.
    pub fn commit_upload(&self, v: &[C::ScalarExt], dev_copy: Option<&mut DeviceVec<C::ScalarExt>>) -> Result<C, ShimError> {
        let mut out = MaybeUninit::<C>::uninit();
        let d = dev_copy.map(|d| { assert!(d.len >= v.len()); d.raw as *mut srs_fe }).unwrap_or(ptr::null_mut());
        check(unsafe { srs_commit_upload(self.raw, v.as_ptr() as *const srs_fe, v.len(), d, SRS_REPR_MONT, ptr::null_mut(),
                                         out.as_mut_ptr() as *mut srs_affine) })?;
        Ok(unsafe { out.assume_init() })
    }

    /// `ck.commit(&concatenate_with_padding(advice, 2^k))` straight from the per-column vectors (src/util/mod.rs:214-218)
    pub fn commit_upload_columns(&self, columns: &[Vec<C::ScalarExt>], pad_size: usize, dev_copy: Option<&mut DeviceVec<C::ScalarExt>>)
        -> Result<C, ShimError> {
        let ptrs: Vec<*const srs_fe> = columns.iter().map(|c| c.as_ptr() as *const srs_fe).collect();
        let lens: Vec<usize> = columns.iter().map(|c| c.len()).collect();
        let n = unsafe { srs_concat_len(lens.as_ptr(), lens.len(), pad_size) };
        let d = dev_copy.map(|d| { assert!(d.len >= n); d.raw as *mut srs_fe }).unwrap_or(ptr::null_mut());
        let mut out = MaybeUninit::<C>::uninit();
        check(unsafe { srs_commit_upload_columns(self.raw, ptrs.as_ptr(), lens.as_ptr(), lens.len(), pad_size, d, SRS_REPR_MONT,
                                                 ptr::null_mut(), out.as_mut_ptr() as *mut srs_affine) })?;
        Ok(unsafe { out.assume_init() })
    }

    /// number of shards of a multi-device key (1 for an ordinary key)
    pub fn num_shards(&self) -> usize { unsafe { srs_ck_num_shards(self.raw) as usize } }

    /// (bytes uploaded over the shard's own link, bytes forwarded to the process's device, streamed commits, device ordinal) of one shard
    /// of a multi-device key: `commit_upload` sends every shard ITS stripes of the witness only (diagnostics, no reference counterpart)
    pub fn shard_stats(&self, shard: usize) -> Result<[u64; 4], ShimError> {
        let mut out = [0u64; 4];
        check(unsafe { srs_ck_shard_stats(self.raw, shard as i32, out.as_mut_ptr()) })?;
        Ok(out)
    }

    /// whether the key holds the optional second (20-bit-window) table; without it every MSM takes the 16-bit windows
    pub fn has_wide_table(&self) -> bool { unsafe { srs_ck_has_wide_table(self.raw) != 0 } }
}
impl<C: GpuCurve> Drop for GpuKey<C> { fn drop(&mut self) { unsafe { srs_ck_free(self.raw) } } }

/// A vector of field elements in HBM (`srs_dev_alloc`): accumulators, cross terms and the incoming witness live here between calls.
pub struct DeviceVec<F> { pub(crate) raw: *mut std::ffi::c_void, pub(crate) len: usize, _f: std::marker::PhantomData<F> }
impl<F: Copy> DeviceVec<F> {
    pub fn new(len: usize) -> Result<Self, ShimError> {
        let mut raw = ptr::null_mut();
        check(unsafe { srs_dev_alloc(len * std::mem::size_of::<F>(), &mut raw) })?;
        Ok(Self { raw, len, _f: Default::default() })
    }
    pub fn from_host(v: &[F]) -> Result<Self, ShimError> {
        let d = Self::new(v.len())?;
        check(unsafe { srs_upload(d.raw, v.as_ptr() as *const _, v.len() * std::mem::size_of::<F>(), ptr::null_mut()) })?;
        Ok(d)
    }
    pub fn to_host(&self, out: &mut [F]) -> Result<(), ShimError> {
        assert_eq!(out.len(), self.len);
        check(unsafe { srs_download(out.as_mut_ptr() as *mut _, self.raw as *const _, self.len * std::mem::size_of::<F>(), ptr::null_mut()) })
    }
    pub fn len(&self) -> usize { self.len }
}
impl<F> Drop for DeviceVec<F> { fn drop(&mut self) { unsafe { srs_dev_free(self.raw) } } }
