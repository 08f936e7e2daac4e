//! `fft::fft / ifft / coset_fft / coset_ifft` (sirius src/fft.rs:160-198): the asserts stay in sirius (`is_power_of_two`,
//! `k <= F::S`), then for `F = bn256::Fr`:
//! ```ignore
//! pub fn fft<F: PrimeField>(a: &mut [F]) {
//!     assert!(a.len().is_power_of_two());
//!     if sirius_amd_shim::fft::ntt(a, false, false) { return; }      // bn256::Fr on a ready device
//!     let log_n = a.len().ilog2(); best_fft(a, get_omega_or_inv(log_n, false), log_n)
//! }
//! ```
use std::ptr;

use halo2_proofs::halo2curves::ff::PrimeField;

use crate::sys::*;

/// in place, natural order in and out; `true` if the device did it.  rc 2 / 3 are the reference's two panics, re-raised.
pub fn ntt<F: PrimeField>(a: &mut [F], inverse: bool, coset: bool) -> bool {
    if crate::field_id::<F>() != Some(SRS_FIELD_FR) || !crate::gpu_ready::<halo2_proofs::halo2curves::bn256::G1Affine>() { return false; }
    match unsafe { srs_ntt(SRS_FIELD_FR, a.as_mut_ptr() as *mut srs_fe, a.len(), inverse as i32, coset as i32, SRS_SPACE_HOST, ptr::null_mut()) } {
        SRS_OK => true,
        SRS_ERR_NOT_POW2 => panic!("assertion failed: a.len().is_power_of_two()"),                  // src/fft.rs:161,169
        SRS_ERR_K_TOO_LARGE => panic!("{}", crate::last_error()),                                   // src/fft.rs:13
        _ => false,                                                                                // device trouble: CPU body
    }
}
