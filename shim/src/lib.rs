//! Safe wrappers over `libsirius_amd.so` (C-ABI: `include/sirius_amd.h`) for the bodies of the sirius functions listed in
//! INTEGRATION.md.  sirius keeps `StepCircuit` / `IVC` / `nifs`; each replaced body becomes a short delegation into this crate,
//! and every curve this crate does not know (pasta in sirius' unit tests) keeps the CPU body (`gpu_ready::<C>()` is false).
//!
//! `sys.rs` is GENERATED from the header (tools/gen_rust_sys.py); the other modules are hand-written against it.
pub mod sys;

pub mod commit;
pub mod fft;
pub mod protogalaxy;
pub mod sangria;

use std::ffi::CStr;
use std::os::raw::c_int;

use halo2_proofs::halo2curves::{bn256, grumpkin, CurveAffine};
use halo2_proofs::halo2curves::ff::{Field, PrimeField};
use once_cell::sync::OnceCell;

/// rc of a library call that is not one of the reference's own error cases
#[derive(Debug)]
pub struct ShimError { pub rc: c_int, pub message: String }

pub fn last_error() -> String {
    unsafe { CStr::from_ptr(sys::srs_last_error()) }.to_string_lossy().into_owned()
}
pub(crate) fn check(rc: c_int) -> Result<(), ShimError> {
    if rc == sys::SRS_OK { Ok(()) } else { Err(ShimError { rc, message: last_error() }) }
}

/// Curves the library implements (src/lib.rs:29-48 of sirius: C1 = bn256::G1Affine, C2 = grumpkin::G1Affine).
pub trait GpuCurve: CurveAffine { const CURVE: c_int; const SCALAR_FIELD: c_int; }
impl GpuCurve for bn256::G1Affine { const CURVE: c_int = sys::SRS_CURVE_BN256; const SCALAR_FIELD: c_int = sys::SRS_FIELD_FR; }
impl GpuCurve for grumpkin::G1Affine { const CURVE: c_int = sys::SRS_CURVE_GRUMPKIN; const SCALAR_FIELD: c_int = sys::SRS_FIELD_FQ; }

/// Once per process and curve: a gfx950 device is bound AND halo2curves' in-memory layout is what the library assumes
/// (Montgomery 4 x u64 little-endian limbs; affine = x || y) -- checked from the raw bytes of `F::ONE`, `F::from(2)` and
/// `C::generator()`.  `false` => the caller keeps the reference's CPU body.
pub fn gpu_ready<C: GpuCurve>() -> bool {
    static READY: [OnceCell<bool>; 2] = [OnceCell::new(), OnceCell::new()];
    *READY[C::CURVE as usize].get_or_init(|| unsafe {
        let one = C::ScalarExt::ONE;
        let two = C::ScalarExt::from(2u64);
        let g = C::generator();
        sys::srs_init(-1) == sys::SRS_OK
            && sys::srs_layout_selftest(C::SCALAR_FIELD, &one as *const _ as *const sys::srs_fe, &two as *const _ as *const sys::srs_fe) == sys::SRS_OK
            && sys::srs_layout_selftest_point(C::CURVE, &g as *const _ as *const sys::srs_affine) == sys::SRS_OK
    })
}

/// `PrimeField` -> the header's field id, for the NTT / folds (only the two fields of the cycle exist in the library)
pub fn field_id<F: PrimeField>() -> Option<c_int> {
    let m = F::MODULUS.to_ascii_lowercase();
    if m.ends_with("43e1f593f0000001") { Some(sys::SRS_FIELD_FR) } else if m.ends_with("3c208c16d87cfd47") { Some(sys::SRS_FIELD_FQ) } else { None }
}
