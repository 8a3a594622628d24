//! `GpuProver`: `ProverProof::create` (kimchi/src/prover.rs:147-1515) as ONE call into `libkimchi_hip.so` (`kh_prove`, the host loop
//! in C++ over the library's own entry points), (this wrapper: circuits without lookups and previous challenges; `kh_prover_index_attach_lookup` / `kh_prove_full` take those).  The index columns are built once on the
//! device from the reference's own `ConstraintSystem` (gates, wiring, shifts), the witness goes in as the 15 columns the reference
//! takes, and what comes back is the reference's `ProverProof` value -- byte-identical to what `ProverProof::create` produces for
//! the same random stream (tests/test_gpu_native_prover.py pins `kh_prove` on the oracle prover, which is pinned on
//! kimchi/src/tests/and.rs:404-731).  Circuits outside that scope keep using `ProverProof::create` with `GpuSrs` /
//! `GpuOpeningProof` / the ark-poly patch (lib.rs, ntt.rs).
//!
//! Not compiled in the repository that ships this file (no Rust toolchain in that image); the FFI surface is checked against the C
//! header by tests/test_rust_bindings.py.
use crate::{limbs, ok, pack, unpack, GpuSrs, HipCurve};
use ark_ff::{PrimeField, Zero};
use kimchi::{
    circuits::{
        constraints::ConstraintSystem,
        gate::GateType,
        wires::{COLUMNS, PERMUTS},
    },
    curve::KimchiCurve,
    proof::{PointEvaluations, ProofEvaluations, ProverCommitments, ProverProof},
};
use kimchi_hip_sys as sys;
use poly_commitment::{commitment::PolyComm, ipa::OpeningProof};
use rand_core::{CryptoRng, RngCore};

const LIB: [GateType; 5] = [GateType::Poseidon, GateType::CompleteAdd, GateType::VarBaseMul, GateType::EndoMul, GateType::EndoMulScalar];
/// the optional gate selectors in the column order of `ProofEvaluations` (proof.rs:95-106)
const OPTIONAL: [(GateType, &str); 6] = [
    (GateType::RangeCheck0, "RangeCheck0"),
    (GateType::RangeCheck1, "RangeCheck1"),
    (GateType::ForeignFieldAdd, "ForeignFieldAdd"),
    (GateType::ForeignFieldMul, "ForeignFieldMul"),
    (GateType::Xor16, "Xor16"),
    (GateType::Rot64, "Rot64"),
];

/// A device buffer of field elements (freed with the prover).
struct DevCols(*mut u64);
impl Drop for DevCols {
    fn drop(&mut self) {
        unsafe { sys::kh_dev_free(self.0 as *mut core::ffi::c_void) };
    }
}

/// Owns a `kh_proof_t` between `kh_prove` and the end of the conversion (like the `kh_ipa_t` guard of `open`, lib.rs).
struct ProofGuard(*mut sys::kh_proof_t);
impl Drop for ProofGuard {
    fn drop(&mut self) {
        unsafe { sys::kh_proof_free(self.0) }
    }
}

pub struct GpuProver<G: HipCurve> {
    index: *mut sys::kh_prover_index_t,
    optional: Vec<usize>, // indices into OPTIONAL of the gate types the circuit has
    num_chunks: usize,
    _cols: [DevCols; 3], // d1, coefficient forms, d8: the index keeps pointers into them
    _srs: GpuSrs<G>,
}
unsafe impl<G: HipCurve> Send for GpuProver<G> {}
unsafe impl<G: HipCurve> Sync for GpuProver<G> {}
impl<G: HipCurve> Drop for GpuProver<G> {
    fn drop(&mut self) {
        unsafe { sys::kh_prover_index_free(self.index) }
    }
}

fn gate_id(name: &str) -> i32 {
    let n = unsafe { sys::kh_gate_count() };
    (0..n)
        .find(|&g| unsafe { std::ffi::CStr::from_ptr(sys::kh_gate_name(g)) }.to_str() == Ok(name))
        .expect("gate kernel")
}

impl<G: HipCurve> GpuProver<G>
where
    G::BaseField: PrimeField,
{
    /// `cs`: the constraint system of `ProverIndex::cs`; `digest`: `ProverIndex::verifier_index_digest` (prover_index.rs:130-146).
    pub fn new(cs: &ConstraintSystem<G::ScalarField>, srs: GpuSrs<G>, digest: G::BaseField) -> Self {
        assert!(matches!(cs.lookup_constraint_system.try_get_or_err(), Ok(None)) && cs.prev_challenges == 0, "kh_prove: no lookups / recursion; use ProverProof::create");
        let n = cs.domain.d1.size as usize;
        let log2_n = cs.domain.d1.log_size_of_group;
        let optional: Vec<usize> = (0..6).filter(|&k| cs.gates.iter().any(|g| g.typ == OPTIONAL[k].0)).collect();
        let ncol = COLUMNS + 2 + PERMUTS + 5 + optional.len();
        // ---- d1 columns: coefficients | generic selector | sid | sigma | five selectors | optional selectors (kimchi_hip.h, kh_prover_index_new)
        let mut d1 = vec![G::ScalarField::zero(); ncol * n];
        let one = <G::ScalarField as ark_ff::One>::one();
        for (row, gate) in cs.gates.iter().enumerate() {
            for (c, v) in gate.coeffs.iter().take(COLUMNS).enumerate() {
                d1[c * n + row] = *v;
            }
            let sel = |t: GateType| if gate.typ == t { one } else { G::ScalarField::zero() };
            d1[COLUMNS * n + row] = sel(GateType::Generic);
            for (c, cell) in gate.wires.iter().enumerate() {
                d1[(COLUMNS + 2 + c) * n + row] = cs.shift[cell.col] * cs.sid[cell.row]; // Shifts::cell_to_field
            }
            for (k, t) in LIB.iter().enumerate() {
                d1[(COLUMNS + 2 + PERMUTS + k) * n + row] = sel(*t);
            }
            for (k, &o) in optional.iter().enumerate() {
                d1[(COLUMNS + 2 + PERMUTS + 5 + k) * n + row] = sel(OPTIONAL[o].0);
            }
        }
        d1[(COLUMNS + 1) * n..(COLUMNS + 2) * n].copy_from_slice(&cs.sid);
        let zk = cs.zk_rows as usize;
        for row in n + 2 - zk..n - 1 {
            for c in 0..PERMUTS {
                d1[(COLUMNS + 2 + c) * n + row] = G::ScalarField::zero(); // constraints.rs:523-530
            }
        }
        // ---- coefficient forms (+ x, permutation vanishing polynomial) and 8x extensions, computed on the device
        let field = if G::CURVE_ID == sys::KH_CURVE_VESTA { sys::KH_FIELD_FP } else { sys::KH_FIELD_FQ };
        let alloc = |elems: usize| {
            let mut p: *mut core::ffi::c_void = core::ptr::null_mut();
            ok(unsafe { sys::kh_dev_alloc(&mut p, 32 * elems) });
            DevCols(p as *mut u64)
        };
        let (b1, bc, b8) = (alloc(ncol * n), alloc((ncol + 2) * n), alloc((ncol + 2) * 8 * n));
        let mut tail = vec![G::ScalarField::zero(); 2 * n];
        tail[1] = one; // x
        let w = cs.domain.d1.group_gen;
        let (a, b, c) = (w.pow([(n - zk) as u64]), w.pow([(n - zk + 1) as u64]), w.pow([(n - 1) as u64]));
        tail[n] = -(a * b * c); // (x - a)(x - b)(x - c): permutation_vanishing_polynomial (permutation.rs:107-118)
        tail[n + 1] = a * b + a * c + b * c;
        tail[n + 2] = -(a + b + c);
        tail[n + 3] = one;
        unsafe {
            ok(sys::kh_dev_upload(b1.0 as *mut _, d1.as_ptr() as *const _, 32 * ncol * n));
            ok(sys::kh_dev_copy(bc.0 as *mut _, b1.0 as *const _, 32 * ncol * n));
            ok(sys::kh_ntt_dev(field, bc.0, log2_n, 1, ncol));
            ok(sys::kh_dev_upload(bc.0.add(4 * ncol * n) as *mut _, tail.as_ptr() as *const _, 32 * 2 * n));
            ok(sys::kh_lde_dev(field, bc.0, log2_n, 3, b8.0, ncol + 2));
            ok(sys::kh_srs_compute_lagrange(srs.handle(), log2_n));
        }
        let live = LIB.iter().enumerate().filter(|(_, t)| cs.gates.iter().any(|g| g.typ == **t)).fold(0u32, |m, (k, _)| m | (1 << k));
        let opt_ids: Vec<i32> = optional.iter().map(|&o| gate_id(OPTIONAL[o].1)).collect();
        let mut index = core::ptr::null_mut();
        let (zk32, pub32) = (cs.zk_rows as u32, cs.public as u32);
        ok(unsafe {
            sys::kh_prover_index_new(srs.handle(), log2_n, zk32, pub32, b1.0, bc.0, b8.0, opt_ids.as_ptr(), opt_ids.len(), live, limbs(&cs.shift), limbs(core::slice::from_ref(&digest)), &mut index)
        });
        let max_poly_size = poly_commitment::SRS::max_poly_size(&srs);
        GpuProver { index, optional, num_chunks: if n < max_poly_size { 1 } else { n / max_poly_size }, _cols: [b1, bc, b8], _srs: srs }
    }

    /// `ProverProof::create::<EFqSponge, EFrSponge, _>(group_map, witness, &[], index, rng)`: the blinders and zero-knowledge rows are
    /// drawn from `rng` in the reference's order (`ScalarField::rand` per element), so a seeded `rng` reproduces the reference's bytes.
    pub fn create<const FULL_ROUNDS: usize>(
        &self,
        witness: &[Vec<G::ScalarField>; COLUMNS],
        rng: &mut (impl RngCore + CryptoRng),
    ) -> ProverProof<G, OpeningProof<G, FULL_ROUNDS>, FULL_ROUNDS>
    where
        G: KimchiCurve<FULL_ROUNDS>,
    {
        let rows = witness[0].len();
        let flat: Vec<G::ScalarField> = witness.iter().flat_map(|c| c.iter().copied()).collect();
        let need = unsafe { sys::kh_prove_randomness_count(self.index, 1) };
        let rnd: Vec<G::ScalarField> = (0..need).map(|_| <G::ScalarField as ark_ff::UniformRand>::rand(rng)).collect();
        let mut proof = core::ptr::null_mut();
        ok(unsafe { sys::kh_prove(self.index, limbs(&flat), rows, core::ptr::null(), limbs(&rnd), need, sys::KH_PROVE_CHECK as u32, &mut proof) });
        let _free = ProofGuard(proof); // `ok()` / a slice conversion below may panic: the library's proof object is released on every path
        let section = |s: i32| {
            let (mut l, mut f, mut k) = (core::ptr::null(), core::ptr::null(), 0usize);
            ok(unsafe { sys::kh_proof_section(proof, s, &mut l, &mut f, &mut k) });
            (l, f, k)
        };
        let points = |s: i32| -> Vec<G> {
            let (l, f, k) = section(s);
            unpack::<G>(unsafe { core::slice::from_raw_parts(l, 8 * k) }, unsafe { core::slice::from_raw_parts(f, k) })
        };
        let elems = |s: i32| -> Vec<G::ScalarField> {
            let (l, _, k) = section(s);
            unsafe { core::slice::from_raw_parts(l as *const G::ScalarField, k) }.to_vec() // Montgomery limbs = ark-ff's representation
        };
        let nch = self.num_chunks;
        let comm = |v: &[G]| PolyComm { chunks: v.to_vec() };
        let w = points(sys::KH_PROOF_W_COMM);
        let e = elems(sys::KH_PROOF_EVALS);
        let ev = |j: usize| PointEvaluations { zeta: e[2 * j * nch..(2 * j + 1) * nch].to_vec(), zeta_omega: e[(2 * j + 1) * nch..(2 * j + 2) * nch].to_vec() };
        let opt = |t: usize| self.optional.iter().position(|&o| o == t).map(|k| ev(43 + k));
        let pe = elems(sys::KH_PROOF_PUBLIC_EVALS);
        let lr = points(sys::KH_PROOF_LR);
        let z12 = elems(sys::KH_PROOF_Z1_Z2);
        let out = ProverProof {
            commitments: ProverCommitments {
                w_comm: core::array::from_fn(|i| comm(&w[i * nch..(i + 1) * nch])),
                z_comm: comm(&points(sys::KH_PROOF_Z_COMM)),
                t_comm: comm(&points(sys::KH_PROOF_T_COMM)),
                lookup: None,
            },
            proof: OpeningProof {
                lr: lr.chunks(2).map(|p| (p[0], p[1])).collect(),
                delta: points(sys::KH_PROOF_DELTA)[0],
                z1: z12[0],
                z2: z12[1],
                sg: points(sys::KH_PROOF_SG)[0],
            },
            evals: ProofEvaluations {
                public: Some(PointEvaluations { zeta: pe[..nch].to_vec(), zeta_omega: pe[nch..].to_vec() }),
                z: ev(0),
                generic_selector: ev(1),
                poseidon_selector: ev(2),
                complete_add_selector: ev(3),
                mul_selector: ev(4),
                emul_selector: ev(5),
                endomul_scalar_selector: ev(6),
                w: core::array::from_fn(|i| ev(7 + i)),
                coefficients: core::array::from_fn(|i| ev(22 + i)),
                s: core::array::from_fn(|i| ev(37 + i)),
                range_check0_selector: opt(0),
                range_check1_selector: opt(1),
                foreign_field_add_selector: opt(2),
                foreign_field_mul_selector: opt(3),
                xor_selector: opt(4),
                rot_selector: opt(5),
                lookup_aggregation: None,
                lookup_table: None,
                lookup_sorted: core::array::from_fn(|_| None),
                runtime_lookup_table: None,
                runtime_lookup_table_selector: None,
                xor_lookup_selector: None,
                lookup_gate_lookup_selector: None,
                range_check_lookup_selector: None,
                foreign_field_mul_lookup_selector: None,
            },
            ft_eval1: elems(sys::KH_PROOF_FT_EVAL1)[0],
            prev_challenges: vec![],
        };
        let _ = pack::<G>; // (the wire format helpers are shared with lib.rs)
        out
    }
}
