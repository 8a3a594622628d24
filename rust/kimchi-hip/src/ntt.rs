//! The evaluation-domain transforms of kimchi's prover on the device.
//!
//! kimchi calls inherent ark-poly methods (`Evaluations::interpolate`, `DensePolynomial::evaluate_over_domain_by_ref`:
//! kimchi/src/prover.rs:289,377,907,1163, circuits/constraints.rs:490-495), so there is no trait to implement; the seam is
//! the crate itself.  `rust/ark-poly-patch/` holds the two functions to splice into a fork of ark-poly 0.5
//! (`Radix2EvaluationDomain::{fft_in_place, ifft_in_place}`), wired in with `[patch.crates-io]` like the arkworks patches
//! the workspace already carries (proof-systems/Cargo.toml:151-152).  That fork binds `kimchi-hip-sys` DIRECTLY (this crate
//! depends on poly-commitment -> ark-poly: calling it from the fork would be a dependency cycle); the helpers below are the
//! same calls for a prover restructured around batches of columns.
use ark_ff::PrimeField;
use kimchi_hip_sys as sys;

use crate::ok;

/// Field ids of the library: 0 = Fp (scalar field of Vesta), 1 = Fq.  Decided from the modulus, so generic code can ask.
pub fn field_id<F: PrimeField>() -> Option<i32> {
    const FP_LOW: u64 = 0x992d30ed00000001;
    const FQ_LOW: u64 = 0x8c46eb2100000001;
    if F::MODULUS_BIT_SIZE != 255 {
        return None;
    }
    match F::MODULUS.as_ref()[0] {
        FP_LOW => Some(sys::KH_FIELD_FP),
        FQ_LOW => Some(sys::KH_FIELD_FQ),
        _ => None,
    }
}

/// Below this size the PCIe round trip of a host-buffer transform costs more than ark-poly's CPU butterflies.
pub const MIN_LOG_SIZE: u32 = 12;

/// In-place NTT / iNTT of `batch` vectors of 2^log2_n elements, natural order in and out, the iNTT scaled by 1/n:
/// exactly `Radix2EvaluationDomain::{fft_in_place, ifft_in_place}`.  Returns false if the field or the size is not one the
/// device handles (the caller then falls through to the CPU code).
pub fn transform_in_place<F: PrimeField>(data: &mut [F], log2_n: u32, inverse: bool) -> bool {
    let Some(fid) = field_id::<F>() else { return false };
    if log2_n < MIN_LOG_SIZE || log2_n > 28 || data.len() % (1usize << log2_n) != 0 {
        return false;
    }
    let batch = data.len() >> log2_n;
    ok(unsafe { sys::kh_ntt(fid, data.as_mut_ptr() as *mut u64, log2_n, inverse as i32, batch) });
    true
}

/// `DensePolynomial::evaluate_over_domain_by_ref(d_{n << log2_blowup})` for `batch` polynomials of degree < n: the zero-padded
/// part of the input never exists (the blow-up index is a virtual first digit of the transform).
pub fn low_degree_extension<F: PrimeField>(coeffs: &[F], log2_n: u32, log2_blowup: u32, out: &mut [F]) -> bool {
    let Some(fid) = field_id::<F>() else { return false };
    let n = 1usize << log2_n;
    if coeffs.len() % n != 0 || out.len() != coeffs.len() << log2_blowup {
        return false;
    }
    ok(unsafe { sys::kh_lde(fid, coeffs.as_ptr() as *const u64, log2_n, log2_blowup, out.as_mut_ptr() as *mut u64, coeffs.len() / n) });
    true
}
