//! Sangria: `VanillaFS::commit_cross_terms` (sirius src/nifs/sangria/mod.rs:102-158), `RelaxedPlonkWitness::fold`
//! (src/nifs/sangria/accumulator.rs:364-404), the group half of `RelaxedPlonkInstance::fold` (:201-264).
use std::ptr;

use crate::commit::GpuKey;
use crate::sys::*;
use crate::{check, GpuCurve, ShimError};

/// Device mirror of a `PlonkStructure` (src/plonk/mod.rs:127-157): gates compiled to row programs, fixed columns and selectors in
/// HBM.  Built once (`PlonkStructure.gpu: OnceCell<GpuStructure>`) from the gates serialised into the header's postfix word stream
/// (`SRS_EX_*`: a 20-line recursive walk over `Expression<F>`).
pub struct GpuStructure { pub(crate) raw: *mut srs_structure }
unsafe impl Send for GpuStructure {}

impl GpuStructure {
    #[allow(clippy::too_many_arguments)]
    pub fn create(field: i32, k: u32, selectors: &[Vec<u8>], fixed: &[*const srs_fe], num_advice: usize, gate_words: &[u64], num_gates: usize)
        -> Result<Self, ShimError> {
        let sel: Vec<*const u8> = selectors.iter().map(|s| s.as_ptr()).collect();
        let mut raw: *mut srs_structure = ptr::null_mut();
        check(unsafe { srs_structure_create(field, k, sel.len(), fixed.len(), num_advice, sel.as_ptr(), fixed.as_ptr(), SRS_SPACE_HOST,
                                            gate_words.as_ptr(), gate_words.len(), num_gates, &mut raw) })?;
        Ok(Self { raw })
    }
    pub fn num_cross_terms(&self) -> usize { unsafe { srs_structure_num_cross_terms(self.raw) } }
}
impl Drop for GpuStructure { fn drop(&mut self) { unsafe { srs_structure_free(self.raw) } } }

/// `commit_cross_terms`: `challenges` = concat_vec!(U1.challenges, [U1.u], U2.challenges, [DEFAULT_u]) (:113-118); W1 / W2 the
/// round vectors concatenated.  Returns (cross_terms, cross_term_commits); rc 1 / 7 map to the reference's two error variants.
pub fn commit_cross_terms<C: GpuCurve>(s: &GpuStructure, ck: &GpuKey<C>, w1: &[C::ScalarExt], w2: &[C::ScalarExt], challenges: &[C::ScalarExt],
                                        row_size: usize) -> Result<(Vec<Box<[C::ScalarExt]>>, Vec<C>), ShimError> {
    use halo2_proofs::halo2curves::ff::Field;
    let d = s.num_cross_terms();
    let mut terms: Vec<Box<[C::ScalarExt]>> = (0..d).map(|_| vec![C::ScalarExt::ZERO; row_size].into_boxed_slice()).collect();
    let t_ptrs: Vec<*mut srs_fe> = terms.iter_mut().map(|t| t.as_mut_ptr() as *mut srs_fe).collect();
    let mut commits = vec![C::identity(); d];
    check(unsafe { srs_commit_cross_terms(s.raw, ck.raw, w1.as_ptr() as *const srs_fe, w2.as_ptr() as *const srs_fe,
                                          challenges.as_ptr() as *const srs_fe, challenges.len(), SRS_SPACE_HOST, ptr::null_mut(),
                                          t_ptrs.as_ptr(), commits.as_mut_ptr() as *mut srs_affine) })?;
    Ok((terms, commits))
}

/// `RelaxedPlonkWitness::fold`, one round vector: out = w1 + r * w2
pub fn fold_witness<F: Copy>(field: i32, out: &mut [F], w1: &[F], w2: &[F], r: &F) -> Result<(), ShimError> {
    assert!(out.len() == w1.len() && w1.len() == w2.len());
    check(unsafe { srs_fold_witness(field, out.as_mut_ptr() as *mut srs_fe, w1.as_ptr() as *const srs_fe, w2.as_ptr() as *const srs_fe,
                                    r as *const F as *const srs_fe, out.len(), SRS_SPACE_HOST, ptr::null_mut()) })
}

/// `RelaxedPlonkWitness::fold`, the error vector: out = e + sum_k r^(k+1) * t[k]
pub fn fold_error<F: Copy>(field: i32, out: &mut [F], e: &[F], cross_terms: &[Box<[F]>], r: &F) -> Result<(), ShimError> {
    let tp: Vec<*const srs_fe> = cross_terms.iter().map(|t| t.as_ptr() as *const srs_fe).collect();
    check(unsafe { srs_fold_error(field, out.as_mut_ptr() as *mut srs_fe, e.as_ptr() as *const srs_fe, tp.as_ptr(), tp.len(),
                                  r as *const F as *const srs_fe, out.len(), SRS_SPACE_HOST, ptr::null_mut()) })
}

/// group half of `RelaxedPlonkInstance::fold`: acc + sum_i scalars[i] * points[i]  (host code in the library)
pub fn point_lincomb<C: GpuCurve>(acc: Option<&C>, points: &[C], scalars: &[C::ScalarExt]) -> Result<C, ShimError> {
    assert_eq!(points.len(), scalars.len());
    let mut out = C::identity();
    check(unsafe { srs_point_lincomb(C::CURVE, acc.map(|a| a as *const C as *const srs_affine).unwrap_or(ptr::null()),
                                     points.as_ptr() as *const srs_affine, scalars.as_ptr() as *const srs_fe, points.len(), SRS_REPR_MONT,
                                     &mut out as *mut C as *mut srs_affine) })?;
    Ok(out)
}

/// Device-resident accumulator of one Sangria-folded circuit: the running (W, E) and the incoming trace buffer live in HBM
/// (`srs_dev_alloc`), the cross-term vectors too; only commitments and the challenge cross the boundary.
pub struct ResidentTraces { pub acc_w: *mut srs_fe, pub inc_w: *mut srs_fe, pub acc_e: *mut srs_fe, pub t: Vec<*mut srs_fe> }

/// What `prove_incoming` hands back: the fresh trace's commitment (for `PlonkTrace.u.W_commitments[0]`), the cross-term commitments,
/// the challenge, and the two library jobs computing the folded W / E commitments (`srs_job_wait`).
/// `folded` is written by the library's workers until the jobs have been waited for, so it is private: `finish` waits and is the
/// only way to read it, and dropping the value early (a `?` return between `prove_incoming` and `finish`) waits too -- the Box is
/// never freed under a running job.
pub struct ProvedIncoming<C: GpuCurve> { pub incoming: C, pub cross_term_commits: Vec<C>, pub r: C::ScalarExt, folded: Box<[C; 2]>, jobs: [u64; 2], waited: bool }

impl<C: GpuCurve> ProvedIncoming<C> {
    /// waits for the two instance-fold jobs: (folded W commitment, folded E commitment)
    pub fn finish(mut self) -> Result<[C; 2], ShimError> {
        self.waited = true;
        let (a, b) = unsafe { (srs_job_wait(self.jobs[0]), srs_job_wait(self.jobs[1])) };
        check(a)?;
        check(b)?;
        Ok(*self.folded)
    }
}
impl<C: GpuCurve> Drop for ProvedIncoming<C> {
    fn drop(&mut self) {
        if !self.waited {
            unsafe { srs_job_wait(self.jobs[0]); srs_job_wait(self.jobs[1]); }
        }
    }
}

/// `VanillaFS::prove` (src/nifs/sangria/mod.rs:253-277) for a trace that has just been synthesised and is not committed yet
/// (CyclefoldIVC::next's support circuit, src/ivc/cyclefold/incrementally_verifiable_computation/mod.rs:255-300): upload, cross terms,
/// ONE batched MSM for the trace's commitment and the cross terms', challenge, folds in place.  ONLY for structures without
/// challenges (the library refuses others: U2's challenges depend on the commitment, src/plonk/mod.rs:465-495).  `ro`: the transcript holding pp_digest
/// and U1; `u2_tail`: what `generate_challenge` (:162-179) absorbs of U2 after its W commitment.  `acc_commitments` = (U1.W, U1.E).
#[allow(clippy::too_many_arguments)]
pub fn prove_incoming<C: GpuCurve>(s: &GpuStructure, ck: &GpuKey<C>, ro: *mut srs_poseidon, challenges: &[C::ScalarExt], tr: &ResidentTraces,
                                   incoming_host: &[C::ScalarExt], u2_tail: &[C::Base], acc_commitments: (&C, &C), stream: *mut std::ffi::c_void)
    -> Result<ProvedIncoming<C>, ShimError> {
    use halo2_proofs::halo2curves::ff::Field;
    let d = s.num_cross_terms();
    let mut commits = vec![C::identity(); d];
    let mut w_commitments = [*acc_commitments.0, C::identity()];
    let mut folded = Box::new([C::identity(); 2]);
    let mut r = C::ScalarExt::ZERO;
    let mut jobs = [0u64; 2];
    check(unsafe { srs_sangria_prove_incoming(s.raw, ck.raw, ro, challenges.as_ptr() as *const srs_fe, challenges.len(), tr.acc_w, tr.inc_w,
                                              incoming_host.as_ptr() as *const srs_fe, u2_tail.as_ptr() as *const srs_fe, u2_tail.len(), tr.acc_e,
                                              stream, &mut r as *mut C::ScalarExt as *mut srs_fe, tr.t.as_ptr(), commits.as_mut_ptr() as *mut srs_affine,
                                              w_commitments.as_mut_ptr() as *mut srs_affine, acc_commitments.1 as *const C as *const srs_affine,
                                              folded.as_mut_ptr() as *mut srs_affine, jobs.as_mut_ptr()) })?;
    Ok(ProvedIncoming { incoming: w_commitments[1], cross_term_commits: commits, r, folded, jobs, waited: false })
}
