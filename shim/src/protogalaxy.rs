//! ProtoGalaxy (sirius src/nifs/protogalaxy/): `poly::compute_F` (poly/mod.rs:68-203), `poly::compute_G` (:308-425),
//! `compute_K_from_G` (:475-509), `evaluate_e_from_trace` (mod.rs:571-640), `ProtoGalaxy::fold_witness` (mod.rs:176-210).
//! `reference_compat = 1` reproduces the reference's leaf rows (`index & 2^k`, src/plonk/mod.rs:714) and is what a drop-in passes:
//! the polynomials enter the transcript.
use std::ptr;

use crate::sangria::GpuStructure;
use crate::sys::*;
use crate::{check, ShimError};

pub const REFERENCE_COMPAT: i32 = 1;

pub fn context(s: &GpuStructure, traces_len: usize) -> Result<srs_pg_context, ShimError> {
    let mut ctx = srs_pg_context::default();
    check(unsafe { srs_pg_context_new(s.raw, traces_len, &mut ctx) })?;
    Ok(ctx)
}

/// compute_F: `w` = the accumulator's witness rounds concatenated, `challenges` = its challenges -> fft_points_count_F coefficients
pub fn compute_f<F: Copy + Default>(s: &GpuStructure, ctx: &srs_pg_context, betas: &[F], delta: &F, w: &[F], challenges: &[F]) -> Result<Vec<F>, ShimError> {
    let mut poly = vec![F::default(); ctx.fft_points_count_F];
    check(unsafe { srs_pg_compute_F(s.raw, betas.as_ptr() as *const srs_fe, betas.len(), delta as *const F as *const srs_fe, w.as_ptr() as *const srs_fe,
                                    challenges.as_ptr() as *const srs_fe, challenges.len(), SRS_SPACE_HOST, REFERENCE_COMPAT, ptr::null_mut(),
                                    poly.as_mut_ptr() as *mut srs_fe) })?;
    Ok(poly)
}

/// compute_G: ws[0] = accumulator, ws[1..] = incoming traces (`FoldedWitness::new` is never materialised) -> fft_points_count_G coefficients
pub fn compute_g<F: Copy + Default>(s: &GpuStructure, ctx: &srs_pg_context, betas_stroke: &[F], ws: &[&[F]], challenges: &[&[F]]) -> Result<Vec<F>, ShimError> {
    assert!(ws.len() == challenges.len() && ws.len() == ctx.instances_to_fold);
    let wp: Vec<*const srs_fe> = ws.iter().map(|w| w.as_ptr() as *const srs_fe).collect();
    let cp: Vec<*const srs_fe> = challenges.iter().map(|c| c.as_ptr() as *const srs_fe).collect();
    let mut poly = vec![F::default(); ctx.fft_points_count_G];
    check(unsafe { srs_pg_compute_G(s.raw, betas_stroke.as_ptr() as *const srs_fe, betas_stroke.len(), wp.as_ptr(), cp.as_ptr(), challenges[0].len(),
                                    ws.len(), SRS_SPACE_HOST, REFERENCE_COMPAT, ptr::null_mut(), poly.as_mut_ptr() as *mut srs_fe) })?;
    Ok(poly)
}

/// compute_K_from_G: 2^fft_log_domain_size_K coefficients (the "log" is a count -- 256 at the Poseidon configurations)
pub fn compute_k_from_g<F: Copy + Default>(ctx: &srs_pg_context, poly_g: &[F], poly_f_in_alpha: &F) -> Result<Vec<F>, ShimError> {
    let mut poly = vec![F::default(); 1usize << ctx.fft_log_domain_size_K];
    check(unsafe { srs_pg_compute_K_from_G(poly_g.as_ptr() as *const srs_fe, poly_g.len(), poly_f_in_alpha as *const F as *const srs_fe,
                                           ctx.instances_to_fold, ctx.fft_log_domain_size_K, ptr::null_mut(), poly.as_mut_ptr() as *mut srs_fe) })?;
    Ok(poly)
}

/// evaluate_e_from_trace
pub fn evaluate_e<F: Copy + Default>(s: &GpuStructure, betas: &[F], w: &[F], challenges: &[F]) -> Result<F, ShimError> {
    let mut e = F::default();
    check(unsafe { srs_pg_evaluate_e(s.raw, betas.as_ptr() as *const srs_fe, betas.len(), w.as_ptr() as *const srs_fe, challenges.as_ptr() as *const srs_fe,
                                     challenges.len(), SRS_SPACE_HOST, REFERENCE_COMPAT, ptr::null_mut(), &mut e as *mut F as *mut srs_fe) })?;
    Ok(e)
}

/// ProtoGalaxy::fold_witness, one round vector: out = sum_j coefs[j] * ws[j]
pub fn fold_witness<F: Copy>(field: i32, out: &mut [F], ws: &[&[F]], coefs: &[F]) -> Result<(), ShimError> {
    assert_eq!(ws.len(), coefs.len());
    let wp: Vec<*const srs_fe> = ws.iter().map(|w| w.as_ptr() as *const srs_fe).collect();
    check(unsafe { srs_fold_lincomb(field, out.as_mut_ptr() as *mut srs_fe, wp.as_ptr(), coefs.as_ptr() as *const srs_fe, ws.len(), out.len(),
                                    SRS_SPACE_HOST, ptr::null_mut()) })
}
