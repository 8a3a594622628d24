//! `GpuSrs<G>` and `GpuOpeningProof<G, FULL_ROUNDS>`: the reference's own plug-in seam
//! (`poly_commitment::SRS<G>`, `poly_commitment::OpenProof<G, FULL_ROUNDS>`, poly-commitment/src/lib.rs:61-298)
//! implemented over `libkimchi_hip.so`, by delegation exactly as `kzg::PairingSRS` delegates to `ipa::SRS`
//! (poly-commitment/src/kzg.rs:250-345).  `ProverProof<G, GpuOpeningProof<G, R>, R>` and
//! `ProverIndex<R, G, GpuSrs<G>>` are then the reference's prover with every MSM on the device
//! (kimchi/src/prover.rs:140-145, prover_index.rs:26-57; the test framework takes a custom SRS factory:
//! kimchi/src/tests/framework.rs:278-319).
//!
//! What runs where:
//!   * `commit_non_hiding`, `commit_evaluations_non_hiding`                -> `kh_commit_*` (MSMs over resident window tables)
//!   * `mask_custom`, `mask`, `commit*`, `commit_evaluations*`              -> compositions, blinding on the host (one
//!     fixed-base scalar multiplication per chunk), as `ipa.rs:605-635,686-748`
//!   * `get_lagrange_basis*`                                                -> the CPU cache of the inner SRS (what the verifier
//!     and the index read); the device computes its own copy of the same basis once (`kh_srs_compute_lagrange`)
//!   * `OpenProof::open`                                                    -> host transcript, device rounds (`kh_ipa_*`)
//!   * `prover::GpuProver::create`                                         -> `kh_prove`: the WHOLE of `ProverProof::create` in one call (no lookups / recursion)
//!   * `OpenProof::verify`                                                  -> `ipa::SRS::verify` of the inner SRS (batch verifier MSM:
//!     `kh_ipa_verify_msm` is available to a caller that restructures `verify`; not needed for proving)
//!
//! Not compiled in the repository that ships this file (no Rust toolchain in that image); the FFI surface is checked
//! against the C header by tests/test_rust_bindings.py.
#![allow(clippy::type_complexity)]

use ark_ec::{AffineRepr, CurveGroup};
use ark_ff::{BigInteger, Field, One, PrimeField, UniformRand, Zero};
use ark_poly::{univariate::DensePolynomial, EvaluationDomain, Evaluations, Radix2EvaluationDomain as D};
use core::ops::Deref;
use groupmap::GroupMap;
use kimchi_hip_sys as sys;
use kimchi::curve::KimchiCurve;
use mina_poseidon::{sponge::ScalarChallenge, FqSponge};
use poly_commitment::{
    commitment::{combined_inner_product, shift_scalar, BatchEvaluationProof, BlindedCommitment, CommitmentCurve, EndoCurve, PolyComm},
    error::CommitmentError,
    ipa::{self, endos, OpeningProof},
    utils::{combine_polys, DensePolynomialOrEvaluations},
    OpenProof, SRS,
};
use rand_core::{CryptoRng, RngCore};
use std::{ffi::CStr, sync::Arc};

pub mod ntt;
pub mod prover;

/// Non-zero status -> panic with the library's message: the trait methods return values, and the reference itself
/// unwraps at these sites (poly-commitment/src/ipa.rs:649-659).
#[track_caller]
fn ok(rc: i32) {
    if rc != sys::KH_OK {
        let msg = unsafe { CStr::from_ptr(sys::kh_last_error()) }.to_string_lossy().into_owned();
        panic!("libkimchi_hip: {msg} (status {rc})");
    }
}

/// Curves the library knows: 0 = Vesta (coordinates Fq, scalars Fp), 1 = Pallas.
pub trait HipCurve: CommitmentCurve + EndoCurve {
    const CURVE_ID: i32;
}
impl HipCurve for mina_curves::pasta::Vesta {
    const CURVE_ID: i32 = sys::KH_CURVE_VESTA;
}
impl HipCurve for mina_curves::pasta::Pallas {
    const CURVE_ID: i32 = sys::KH_CURVE_PALLAS;
}

/// `&[Fp]` -> `*const u64`: ark-ff's `Fp256<MontBackend<_, 4>>` is four little-endian u64 Montgomery limbs, and the
/// reference performs the same reinterpretation with size / alignment asserts (kimchi/src/cached_prover_index.rs:502-539).
fn limbs<F: PrimeField>(v: &[F]) -> *const u64 {
    debug_assert_eq!(core::mem::size_of::<F>(), 32);
    v.as_ptr() as *const u64
}
fn limbs_mut<F: PrimeField>(v: &mut F) -> *mut u64 {
    debug_assert_eq!(core::mem::size_of::<F>(), 32);
    v as *mut F as *mut u64
}

/// `Affine { x, y, infinity }` has no guaranteed field order or padding: copy x || y field-wise into the packed 64-byte
/// record of the wire format, infinity out of band.
fn pack<G: CommitmentCurve>(pts: &[G]) -> (Vec<u64>, Vec<u8>)
where
    G::BaseField: PrimeField,
{
    let mut xy = vec![0u64; 8 * pts.len()];
    let mut inf = vec![0u8; pts.len()];
    for (i, p) in pts.iter().enumerate() {
        match p.to_coordinates() {
            None => inf[i] = 1,
            Some((x, y)) => {
                // the in-memory (Montgomery) limbs, not into_bigint(): the wire format is ark-ff's representation
                let xs = unsafe { core::slice::from_raw_parts(&x as *const G::BaseField as *const u64, 4) };
                let ys = unsafe { core::slice::from_raw_parts(&y as *const G::BaseField as *const u64, 4) };
                xy[8 * i..8 * i + 4].copy_from_slice(xs);
                xy[8 * i + 4..8 * i + 8].copy_from_slice(ys);
            }
        }
    }
    (xy, inf)
}
fn unpack<G: CommitmentCurve>(xy: &[u64], inf: &[u8]) -> Vec<G>
where
    G::BaseField: PrimeField,
{
    inf.iter()
        .enumerate()
        .map(|(i, &is_inf)| {
            if is_inf != 0 {
                G::zero()
            } else {
                let mut x = G::BaseField::zero();
                let mut y = G::BaseField::zero();
                unsafe {
                    core::ptr::copy_nonoverlapping(xy[8 * i..].as_ptr(), &mut x as *mut G::BaseField as *mut u64, 4);
                    core::ptr::copy_nonoverlapping(xy[8 * i + 4..].as_ptr(), &mut y as *mut G::BaseField as *mut u64, 4);
                }
                G::of_coordinates(x, y)
            }
        })
        .collect()
}

/// The device half of an SRS: window tables of `g` (and of every Lagrange basis used so far) resident in HBM.
struct DevHandle(*mut sys::kh_srs_t);
unsafe impl Send for DevHandle {} // every kh_* entry point is thread-safe (include/kimchi_hip.h)
unsafe impl Sync for DevHandle {}
impl Drop for DevHandle {
    fn drop(&mut self) {
        unsafe { sys::kh_srs_free(self.0) }
    }
}

#[derive(Clone)]
pub struct GpuSrs<G: HipCurve> {
    /// the reference SRS: owns `g`, `h`, the CPU Lagrange-basis cache (verifier, index, serialisation)
    pub inner: Arc<ipa::SRS<G>>,
    dev: Arc<DevHandle>,
}

impl<G: HipCurve> core::fmt::Debug for GpuSrs<G> {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "GpuSrs {{ size: {}, device: {} }}", self.inner.g.len(), unsafe { sys::kh_srs_device(self.dev.0) })
    }
}

impl<G: HipCurve> GpuSrs<G>
where
    G::BaseField: PrimeField,
{
    /// Uploads `inner.g` to the calling thread's current device (`kh_set_device`; default: the first device initialised)
    /// and expands it to the MSM window tables.  The blinding base is taken from `inner.h`.
    pub fn new(inner: ipa::SRS<G>) -> Self {
        let (xy, _) = pack(&inner.g);
        let mut h = core::ptr::null_mut();
        ok(unsafe { sys::kh_srs_create(G::CURVE_ID, xy.as_ptr(), inner.g.len(), &mut h) });
        let (hxy, _) = pack(&[inner.h]);
        ok(unsafe { sys::kh_srs_set_blinding_base(h, hxy.as_ptr()) });
        Self { inner: Arc::new(inner), dev: Arc::new(DevHandle(h)) }
    }

    /// Once per (SRS, domain), like the reference's cache (ipa.rs:780-801): the device derives the basis from its own copy
    /// of `g` with a group iNTT (same points as `ipa::SRS::lagrange_basis`, ipa.rs:1065-1172).
    fn ensure_basis(&self, log2_domain: u32) {
        if unsafe { sys::kh_srs_lagrange_chunks(self.dev.0, log2_domain) } == 0 {
            ok(unsafe { sys::kh_srs_compute_lagrange(self.dev.0, log2_domain) });
        }
    }

    fn chunks(&self, xy: Vec<u64>, inf: Vec<u8>, count: usize) -> PolyComm<G> {
        PolyComm::new(unpack::<G>(&xy[..8 * count], &inf[..count]))
    }

    /// the library's handle (prover.rs: `kh_prover_index_new` takes it)
    pub(crate) fn handle(&self) -> *mut sys::kh_srs_t {
        self.dev.0
    }
}

/// ONE large MSM over several GPUs (BASELINE config 4): the basis is split by point range, shard r lives on device r
/// (`kh_set_device` + `kh_srs_create`), every shard reduces its slice with the full single-GPU pipeline and the partial
/// sums are folded on the host.  No collective, no torch: `kh_msm_sharded` does the whole thing.
pub struct GpuShardedMsm<G: HipCurve> {
    shards: Vec<DevHandle>,
    _g: core::marker::PhantomData<G>,
}

impl<G: HipCurve> GpuShardedMsm<G>
where
    G::BaseField: PrimeField,
{
    /// `devices[r]` gets `g[r * len / R .. (r + 1) * len / R)`.
    pub fn new(g: &[G], devices: &[i32]) -> Self {
        let r_ = devices.len();
        let mut shards = Vec::with_capacity(r_);
        let prev = unsafe { sys::kh_get_device() };
        for (r, dev) in devices.iter().enumerate() {
            let (lo, hi) = (r * g.len() / r_, (r + 1) * g.len() / r_);
            ok(unsafe { sys::kh_set_device(*dev) });
            let (xy, _) = pack(&g[lo..hi]);
            let mut h = core::ptr::null_mut();
            ok(unsafe { sys::kh_srs_create(G::CURVE_ID, xy.as_ptr(), hi - lo, &mut h) });
            shards.push(DevHandle(h));
        }
        if prev >= 0 {
            ok(unsafe { sys::kh_set_device(prev) });
        }
        Self { shards, _g: core::marker::PhantomData }
    }

    /// `VariableBaseMSM::msm(g, scalars)` over all the shards.
    pub fn msm(&self, scalars: &[G::ScalarField]) -> G {
        let hs: Vec<*mut sys::kh_srs_t> = self.shards.iter().map(|d| d.0).collect();
        let mut xy = vec![0u64; 8];
        let mut inf = vec![0u8; 1];
        ok(unsafe { sys::kh_msm_sharded(hs.as_ptr(), hs.len(), limbs(scalars), scalars.len(), 1, xy.as_mut_ptr(), inf.as_mut_ptr()) });
        unpack::<G>(&xy, &inf)[0]
    }
}

impl<G: HipCurve> SRS<G> for GpuSrs<G>
where
    G::BaseField: PrimeField,
{
    fn max_poly_size(&self) -> usize {
        self.inner.g.len()
    }

    fn blinding_commitment(&self) -> G {
        self.inner.h
    }

    fn mask_custom(&self, com: PolyComm<G>, blinders: &PolyComm<G::ScalarField>) -> Result<BlindedCommitment<G>, CommitmentError> {
        if com.len() != blinders.len() {
            return Err(CommitmentError::BlindersDontMatch(blinders.len(), com.len()));
        }
        let (xy, inf) = pack(&com.chunks);
        let mut out = vec![0u64; 8 * com.len()];
        let mut oinf = vec![0u8; com.len()];
        ok(unsafe {
            sys::kh_mask_custom(self.dev.0, xy.as_ptr(), inf.as_ptr(), com.len(), limbs(&blinders.chunks), blinders.len(), out.as_mut_ptr(), oinf.as_mut_ptr())
        });
        Ok(BlindedCommitment { commitment: PolyComm::new(unpack::<G>(&out, &oinf)), blinders: blinders.clone() })
    }

    fn mask(&self, comm: PolyComm<G>, rng: &mut (impl RngCore + CryptoRng)) -> BlindedCommitment<G> {
        let blinders = comm.map(|_| G::ScalarField::rand(rng));
        self.mask_custom(comm, &blinders).unwrap()
    }

    fn commit_non_hiding(&self, plnm: &DensePolynomial<G::ScalarField>, num_chunks: usize) -> PolyComm<G> {
        let n = self.inner.g.len();
        let cap = num_chunks.max(plnm.coeffs.len().div_ceil(n)).max(1);
        let (mut xy, mut inf, mut cnt) = (vec![0u64; 8 * cap], vec![0u8; cap], 0usize);
        ok(unsafe { sys::kh_commit_non_hiding(self.dev.0, limbs(&plnm.coeffs), plnm.coeffs.len(), num_chunks, xy.as_mut_ptr(), inf.as_mut_ptr(), &mut cnt) });
        self.chunks(xy, inf, cnt)
    }

    fn commit(&self, plnm: &DensePolynomial<G::ScalarField>, num_chunks: usize, rng: &mut (impl RngCore + CryptoRng)) -> BlindedCommitment<G> {
        self.mask(self.commit_non_hiding(plnm, num_chunks), rng)
    }

    fn commit_custom(
        &self,
        plnm: &DensePolynomial<G::ScalarField>,
        num_chunks: usize,
        blinders: &PolyComm<G::ScalarField>,
    ) -> Result<BlindedCommitment<G>, CommitmentError> {
        self.mask_custom(self.commit_non_hiding(plnm, num_chunks), blinders)
    }

    fn commit_evaluations_non_hiding(&self, domain: D<G::ScalarField>, plnm: &Evaluations<G::ScalarField, D<G::ScalarField>>) -> PolyComm<G> {
        let k = domain.log_size_of_group;
        self.ensure_basis(k);
        let cap = (unsafe { sys::kh_srs_lagrange_chunks(self.dev.0, k) }).max(1) as usize;
        let (mut xy, mut inf, mut cnt) = (vec![0u64; 8 * cap], vec![0u8; cap], 0usize);
        ok(unsafe {
            sys::kh_commit_evaluations_non_hiding(self.dev.0, k, limbs(&plnm.evals), plnm.evals.len(), xy.as_mut_ptr(), inf.as_mut_ptr(), &mut cnt)
        });
        self.chunks(xy, inf, cnt)
    }

    fn commit_evaluations(
        &self,
        domain: D<G::ScalarField>,
        plnm: &Evaluations<G::ScalarField, D<G::ScalarField>>,
        rng: &mut (impl RngCore + CryptoRng),
    ) -> BlindedCommitment<G> {
        self.mask(self.commit_evaluations_non_hiding(domain, plnm), rng)
    }

    fn commit_evaluations_custom(
        &self,
        domain: D<G::ScalarField>,
        plnm: &Evaluations<G::ScalarField, D<G::ScalarField>>,
        blinders: &PolyComm<G::ScalarField>,
    ) -> Result<BlindedCommitment<G>, CommitmentError> {
        self.mask_custom(self.commit_evaluations_non_hiding(domain, plnm), blinders)
    }

    fn create(depth: usize) -> Self {
        Self::new(<ipa::SRS<G> as SRS<G>>::create(depth))
    }

    fn get_lagrange_basis(&self, domain: D<G::ScalarField>) -> impl Deref<Target = Vec<PolyComm<G>>> + '_ {
        self.inner.get_lagrange_basis(domain)
    }

    fn get_lagrange_basis_from_domain_size(&self, domain_size: usize) -> impl Deref<Target = Vec<PolyComm<G>>> + '_ {
        self.inner.get_lagrange_basis_from_domain_size(domain_size)
    }

    fn size(&self) -> usize {
        self.inner.g.len()
    }
}

/// `ipa::OpeningProof` produced with the folding rounds on the device.
#[derive(Clone, Debug)]
pub struct GpuOpeningProof<G: HipCurve, const FULL_ROUNDS: usize>(pub OpeningProof<G, FULL_ROUNDS>);

// `KimchiCurve<FULL_ROUNDS>` (kimchi/src/curve.rs:20-38; both Pasta curves implement it) is an impl-level bound, not one on the
// trait methods: `verify` needs `other_curve_sponge_params()` to build the placeholder sponges it swaps in (see there).
impl<G: HipCurve + KimchiCurve<FULL_ROUNDS>, const FULL_ROUNDS: usize> OpenProof<G, FULL_ROUNDS> for GpuOpeningProof<G, FULL_ROUNDS>
where
    G::BaseField: PrimeField,
{
    type SRS = GpuSrs<G>;

    /// `SRS::open` (poly-commitment/src/ipa.rs:823-1061).  The transcript, the RNG and the Schnorr tail are the
    /// reference's, line for line; the two MSMs and the folds of every round run on the device, which keeps `a`, `b` and the
    /// challenge tensor resident and never folds the basis (L_j, R_j are MSMs over the original tables; equal group
    /// elements have equal affine coordinates, so the proof bytes are the reference's).
    fn open<EFqSponge, RNG, Dom: EvaluationDomain<<G as AffineRepr>::ScalarField>>(
        srs: &Self::SRS,
        group_map: &<G as CommitmentCurve>::Map,
        // `poly_commitment::PolynomialsToCombine<G, Dom>` spelled out: the alias is private to the crate (poly-commitment/src/lib.rs:249,
        // `type`, not `pub type`), so an implementor outside it has to write the slice type itself
        plnms: &[(DensePolynomialOrEvaluations<'_, <G as AffineRepr>::ScalarField, Dom>, PolyComm<<G as AffineRepr>::ScalarField>)],
        elm: &[<G as AffineRepr>::ScalarField],
        polyscale: <G as AffineRepr>::ScalarField,
        evalscale: <G as AffineRepr>::ScalarField,
        mut sponge: EFqSponge,
        rng: &mut RNG,
    ) -> Self
    where
        EFqSponge: Clone + FqSponge<<G as AffineRepr>::BaseField, G, <G as AffineRepr>::ScalarField, FULL_ROUNDS>,
        RNG: RngCore + CryptoRng,
    {
        let (_endo_q, endo_r) = endos::<G>();
        let n = srs.inner.g.len();
        assert!(n.is_power_of_two(), "the device opening needs a power-of-two SRS");
        let _ctx = PrivateContext::enter(); // the reference opens from rayon workers: each gets its own streams and pipeline slots for the rounds
        let rounds = n.trailing_zeros() as usize;

        // ipa.rs:851-888: p = sum_i polyscale^i p_i and the combined blinder; b_init[j] = sum_i evalscale^i elm_i^j
        let (p, blinding_factor) = combine_polys::<G, Dom>(plnms, polyscale, n);
        let mut b_init = vec![G::ScalarField::zero(); n];
        let mut scale = G::ScalarField::one();
        for e in elm {
            let mut t = G::ScalarField::one();
            for b in b_init.iter_mut() {
                *b += scale * t;
                t *= e;
            }
            scale *= evalscale;
        }
        // ipa.rs:891-913
        let cip = p.coeffs.iter().zip(b_init.iter()).map(|(a, b)| *a * b).fold(G::ScalarField::zero(), |acc, x| acc + x);
        sponge.absorb_fr(&[shift_scalar::<G>(cip)]);
        let u_base: G = {
            let t = sponge.challenge_fq();
            let (x, y) = group_map.to_group(t);
            G::of_coordinates(x, y)
        };
        let mut a = p.coeffs;
        a.resize(n, G::ScalarField::zero());

        let (uxy, _) = pack(&[u_base]);
        let mut st = core::ptr::null_mut();
        ok(unsafe { sys::kh_ipa_begin(srs.dev.0, limbs(&a), a.len(), limbs(&b_init), b_init.len(), uxy.as_ptr(), &mut st) });
        struct Guard(*mut sys::kh_ipa_t);
        impl Drop for Guard {
            fn drop(&mut self) {
                unsafe { sys::kh_ipa_free(self.0) }
            }
        }
        let _guard = Guard(st);

        let (mut lr, mut blinders, mut chals, mut chal_invs) = (vec![], vec![], vec![], vec![]);
        for _ in 0..rounds {
            // ipa.rs:940-941: the RNG is drawn in the reference's order
            let rand_l = <G::ScalarField as UniformRand>::rand(rng);
            let rand_r = <G::ScalarField as UniformRand>::rand(rng);
            let (mut out, mut inf) = ([0u64; 16], [0u8; 2]);
            ok(unsafe { sys::kh_ipa_round_lr(st, limbs(&[rand_l]), limbs(&[rand_r]), out.as_mut_ptr(), inf.as_mut_ptr()) });
            let pts = unpack::<G>(&out, &inf);
            let (l, r) = (pts[0], pts[1]);
            lr.push((l, r));
            blinders.push((rand_l, rand_r));
            sponge.absorb_g(&[l]); // ipa.rs:966-967
            sponge.absorb_g(&[r]);
            let u_pre = sponge.challenge(); // ipa.rs:972: ScalarChallenge::new(sponge.challenge()), 128 bits
            let pre = u_pre.into_bigint();
            let chal = [pre.as_ref()[0], pre.as_ref()[1]];
            let (mut u, mut u_inv) = (G::ScalarField::zero(), G::ScalarField::zero());
            ok(unsafe { sys::kh_ipa_round_fold(st, chal.as_ptr(), limbs_mut(&mut u), limbs_mut(&mut u_inv)) });
            debug_assert_eq!(u, ScalarChallenge::new(u_pre).to_field(&endo_r));
            chals.push(u);
            chal_invs.push(u_inv);
        }
        let (mut a0, mut b0) = (G::ScalarField::zero(), G::ScalarField::zero());
        let (mut sg, mut sg_inf) = ([0u64; 8], 0u8);
        ok(unsafe { sys::kh_ipa_finish(st, limbs_mut(&mut a0), limbs_mut(&mut b0), sg.as_mut_ptr(), &mut sg_inf) });
        let g0 = unpack::<G>(&sg, &[sg_inf])[0];

        // ipa.rs:1021-1059, unchanged
        let r_prime = blinders
            .iter()
            .zip(chals.iter().zip(chal_invs.iter()))
            .map(|((rand_l, rand_r), (u, u_inv))| ((*rand_l) * u_inv) + (*rand_r * u))
            .fold(blinding_factor, |acc, x| acc + x);
        let d = <G::ScalarField as UniformRand>::rand(rng);
        let r_delta = <G::ScalarField as UniformRand>::rand(rng);
        let delta = ((g0.into_group() + (u_base.mul(b0))).into_affine().mul(d) + srs.inner.h.mul(r_delta)).into_affine();
        sponge.absorb_g(&[delta]);
        let c = ScalarChallenge::new(sponge.challenge()).to_field(&endo_r);
        let z1 = a0 * c + d;
        let z2 = r_prime * c + r_delta;
        GpuOpeningProof(OpeningProof { delta, lr, z1, z2, sg: g0 })
    }

    fn verify<EFqSponge, RNG>(
        srs: &Self::SRS,
        group_map: &G::Map,
        batch: &mut [BatchEvaluationProof<G, EFqSponge, Self, FULL_ROUNDS>],
        rng: &mut RNG,
    ) -> bool
    where
        EFqSponge: FqSponge<G::BaseField, G, G::ScalarField, FULL_ROUNDS>,
        RNG: RngCore + CryptoRng,
    {
        // The verifier is not on the proving path: hand the batch to the inner SRS (`ipa::SRS::verify`, ipa.rs:301-502).  Its element
        // type differs (`opening: &OpeningProof` instead of `&GpuOpeningProof`), so every element is rebuilt BY MOVING its parts:
        // `EFqSponge` is not `Clone` here (lib.rs:289-297 of poly-commitment: only `FqSponge`), so the sponge is swapped against a
        // fresh one (`FqSponge::new`, poseidon/src/sponge.rs:16) and -- like the vectors -- put back afterwards in the state the inner
        // verifier left it in, which is what the reference's own `verify` does to the caller's batch.
        // The parts go back from a drop guard, so a panic inside the inner verifier (it `assert!`s on malformed batches) does not leave the
        // caller's batch holding placeholder sponges and emptied vectors.
        struct Restore<'s, 'a, G, S, const R: usize>
        where
            G: HipCurve + KimchiCurve<R>,
            G::BaseField: PrimeField,
            S: FqSponge<G::BaseField, G, G::ScalarField, R>,
        {
            batch: &'s mut [BatchEvaluationProof<'a, G, S, GpuOpeningProof<G, R>, R>],
            inner: Vec<BatchEvaluationProof<'a, G, S, OpeningProof<G, R>, R>>,
        }
        impl<'s, 'a, G, S, const R: usize> Drop for Restore<'s, 'a, G, S, R>
        where
            G: HipCurve + KimchiCurve<R>,
            G::BaseField: PrimeField,
            S: FqSponge<G::BaseField, G, G::ScalarField, R>,
        {
            fn drop(&mut self) {
                for (b, i) in self.batch.iter_mut().zip(self.inner.drain(..)) {
                    b.sponge = i.sponge;
                    b.evaluations = i.evaluations;
                    b.evaluation_points = i.evaluation_points;
                }
            }
        }
        let mut guard: Restore<'_, '_, G, EFqSponge, FULL_ROUNDS> = Restore { inner: Vec::with_capacity(batch.len()), batch };
        for k in 0..guard.batch.len() {
            let b = &mut guard.batch[k];
            let opening: &GpuOpeningProof<G, FULL_ROUNDS> = b.opening; // `&'a Self` is `Copy`: not a borrow of `batch`
            let moved = BatchEvaluationProof {
                sponge: core::mem::replace(&mut b.sponge, EFqSponge::new(G::other_curve_sponge_params())),
                evaluations: core::mem::take(&mut b.evaluations),
                evaluation_points: core::mem::take(&mut b.evaluation_points),
                polyscale: b.polyscale,
                evalscale: b.evalscale,
                opening: &opening.0,
                combined_inner_product: b.combined_inner_product,
            };
            guard.inner.push(moved);
        }
        let accepted = srs.inner.verify(group_map, &mut guard.inner, rng);
        drop(guard); // puts sponge / evaluations / evaluation_points back, in the state the inner verifier left them in
        accepted
    }
}

/// combined_inner_product is re-exported for callers that assemble `BatchEvaluationProof`s (verifier.rs:491-520).
pub use combined_inner_product as combined_inner_product_of_evaluations;

/// Several prover threads in one process: while a `PrivateContext` is alive the calling thread works on a library context of its own (own main
/// stream, MSM pipeline slots, workspaces, lock) instead of the device's shared one (include/kimchi_hip.h, `kh_private_context_begin`), so independent
/// provers do not queue their vector steps on one stream.  `GpuProver::create` needs none (`kh_prove` takes one itself); `open` below takes one when
/// the thread has none yet.  Not `Send`: the context belongs to the thread that began it.
pub struct PrivateContext {
    began: bool,
    _not_send: core::marker::PhantomData<*mut ()>,
}
impl PrivateContext {
    pub fn enter() -> Self {
        let began = unsafe { sys::kh_private_context_active() } == 0;
        if began {
            ok(unsafe { sys::kh_private_context_begin() });
        }
        PrivateContext { began, _not_send: core::marker::PhantomData }
    }
}
impl Drop for PrivateContext {
    fn drop(&mut self) {
        if self.began {
            ok(unsafe { sys::kh_private_context_end() });
        }
    }
}

/// One process, several GPUs: bind the calling thread (e.g. a rayon worker) to `device` before creating a `GpuSrs`;
/// the handle then runs on that device from any thread (include/kimchi_hip.h, "device").
pub fn set_device(device: i32) {
    ok(unsafe { sys::kh_set_device(device) });
}

#[allow(dead_code)]
fn _assert_traits<G: HipCurve>()
where
    G::BaseField: PrimeField,
{
    fn is_srs<G: CommitmentCurve, S: SRS<G>>() {}
    is_srs::<G, GpuSrs<G>>();
    let _ = BigInteger::num_bits; // (keeps the ark-ff import set identical to the reference's ipa.rs)
    let _ = <G::ScalarField as Field>::ONE;
    let _ = CurveGroup::into_affine;
}
