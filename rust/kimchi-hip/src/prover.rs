//! `GpuProver`: `ProverProof::create_recursive` (kimchi/src/prover.rs:187-1515) as ONE call into `libkimchi_hip.so` (`kh_prove_full`, the host loop
//! in C++ over the library's own entry points) for everything the reference's entry takes: generic / library / optional gates, public inputs, any
//! number of chunks, LOOKUPS into fixed tables, RUNTIME TABLES and PREVIOUS CHALLENGES.  The index columns are built once on the device from the
//! reference's own `ConstraintSystem` (gates, wiring, shifts; with lookups: the pattern selectors, table columns and table ids of its
//! `LookupConstraintSystem`, lookup/index.rs:159-194), the witness goes in as the 15 columns the reference takes, and what comes back is the
//! reference's `ProverProof` value -- byte-identical to what `ProverProof::create_recursive` produces for the same random stream
//! (tests/test_gpu_native_prover.py and tests/test_gpu_prover_parity.py pin `kh_prove*` on the oracle prover, which is pinned on
//! kimchi/src/tests/and.rs:404-731 -- a lookup proof; tests/test_gpu_proof_fixtures.py at 2^16 gates).
//!
//! STATUS: EXPERIMENTAL, NEVER COMPILED.  The image this repository is built in has no Rust toolchain; what is checked is (a) every `sys::` call
//! against the C header (names, arity: tests/test_rust_bindings.py), (b) every `use` path against the reference's sources for visibility and the
//! trait signatures token for token (tests/test_rust_reference_surface.py), (c) the field names of the reference structs built below
//! (`ProofEvaluations`, `ProverCommitments`, `LookupCommitments`, `RecursionChallenge`, `RuntimeTable`: the same test).  Treat it as the shim a
//! maintainer starts from, with `cargo check` against the proof-systems workspace as the first step.
use crate::{limbs, ok, pack, unpack, GpuSrs, HipCurve};
use ark_ff::{batch_inversion, Field, One, PrimeField, Zero};
use kimchi::{
    circuits::{
        constraints::ConstraintSystem,
        gate::GateType,
        lookup::{index::LookupConstraintSystem, runtime_tables::RuntimeTable},
        wires::{COLUMNS, PERMUTS},
    },
    curve::KimchiCurve,
    prover_index::ProverIndex,
    proof::{LookupCommitments, PointEvaluations, ProofEvaluations, ProverCommitments, ProverProof, RecursionChallenge},
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

/// What `create` needs to know about the lookup argument of the index to lay the library's sections out as the reference's structs.
struct LookupShape {
    patterns: [bool; 4],                // xor, lookup, range_check, ffmul (the library's pattern ids 0..3, lookups.rs:57-66)
    max_per_row: usize,                 // sorted columns = max_per_row + 1
    runtime: Option<Vec<(i32, usize)>>, // (id, len) of the configured runtime tables, in order (runtime_tables.rs:15-20)
}

pub struct GpuProver<G: HipCurve> {
    index: *mut sys::kh_prover_index_t,
    optional: Vec<usize>, // indices into OPTIONAL of the gate types the circuit has
    num_chunks: usize,
    lookup: Option<LookupShape>,
    prev_challenges: usize,
    _cols: Vec<DevCols>, // d1, coefficient forms, d8 (+ the lookup columns): the index keeps pointers into them
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
    /// From the reference's own `ProverIndex` instantiated with `GpuSrs<G>` as its SRS type (kimchi/src/prover_index.rs:26-57): the constraint system
    /// (`cs`), the SRS handle (`srs`) and the verifier-index digest the index caches (`verifier_index_digest`: `compute_verifier_index_digest`,
    /// prover_index.rs:103-123, must have run -- the prover absorbs it first, prover.rs:259-262).  `max_poly_size` is the SRS's own (prover_index.rs:86-87).
    pub fn from_index<const FULL_ROUNDS: usize>(index: &ProverIndex<FULL_ROUNDS, G, GpuSrs<G>>) -> Self
    where
        G: KimchiCurve<FULL_ROUNDS>,
    {
        let digest = index.verifier_index_digest.expect("ProverIndex::compute_verifier_index_digest has not run");
        assert_eq!(index.max_poly_size, poly_commitment::SRS::max_poly_size(&*index.srs), "the index was built for another SRS");
        Self::new(&index.cs, (*index.srs).clone(), digest)
    }

    /// `cs`: the constraint system of `ProverIndex::cs`; `digest`: `ProverIndex::verifier_index_digest` (prover_index.rs:130-146).
    pub fn new(cs: &ConstraintSystem<G::ScalarField>, srs: GpuSrs<G>, digest: G::BaseField) -> Self {
        // the lookup constraint system is built lazily (constraints.rs:230, 239-240); a build error is the caller's, as in prover.rs:386-390
        let lcs: Option<&LookupConstraintSystem<G::ScalarField>> = cs.lookup_constraint_system.try_get_or_err().expect("lookup constraint system").as_ref();
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
        let mut cols = vec![b1, bc, b8];
        let lookup = lcs.map(|lcs| Self::attach_lookup(index, lcs, cs, field, log2_n, n, zk, &mut cols));
        let max_poly_size = poly_commitment::SRS::max_poly_size(&srs);
        GpuProver { index, optional, num_chunks: if n < max_poly_size { 1 } else { n / max_poly_size }, lookup, prev_challenges: cs.prev_challenges, _cols: cols, _srs: srs }
    }

    /// `kh_prover_index_attach_lookup` (+ `_attach_runtime_tables`) from the reference's `LookupConstraintSystem`: it keeps the pattern selectors,
    /// the table columns and the table ids as evaluations over d8 (lookup/index.rs:159-194) -- every 8th value is the d1 column; coefficient forms
    /// and the library's own d8 copies are made on the device.  The three row-set atoms of the lookup constraints (expr.rs:883-893) are evaluated
    /// here on d8: VanishesOnZeroKnowledgeAndPreviousRows is the index's precomputation, UnnormalizedLagrangeBasis(i) = (x^n - 1) / (x - w^i),
    /// with the value n w^-i where x = w^i.
    #[allow(clippy::too_many_arguments)]
    fn attach_lookup(
        index: *mut sys::kh_prover_index_t,
        lcs: &LookupConstraintSystem<G::ScalarField>,
        cs: &ConstraintSystem<G::ScalarField>,
        field: i32,
        log2_n: u32,
        n: usize,
        zk: usize,
        cols: &mut Vec<DevCols>,
    ) -> LookupShape {
        let d1_of = |e8: &[G::ScalarField]| -> Vec<G::ScalarField> { (0..n).map(|i| e8[8 * i]).collect() };
        let sels = [&lcs.lookup_selectors.xor, &lcs.lookup_selectors.lookup, &lcs.lookup_selectors.range_check, &lcs.lookup_selectors.ffmul];
        let patterns: Vec<i32> = (0..4).filter(|&k| sels[k as usize].is_some()).collect();
        // ---- selector block: the pattern selectors, then (if any) the runtime-table selector
        let mut sel1: Vec<G::ScalarField> = Vec::new();
        for &k in &patterns {
            sel1.extend(d1_of(&sels[k as usize].as_ref().unwrap().evals));
        }
        if let Some(rs) = &lcs.runtime_selector {
            sel1.extend(d1_of(&rs.evals));
        }
        let nsel = sel1.len() / n;
        // ---- table block: the concatenated table columns, then (if any) the table-id column
        let mut tab1: Vec<G::ScalarField> = Vec::new();
        for t8 in &lcs.lookup_table8 {
            tab1.extend(d1_of(&t8.evals));
        }
        if let Some(ids8) = &lcs.table_ids8 {
            tab1.extend(d1_of(&ids8.evals));
        }
        // ---- the atoms on d8
        let n8 = 8 * n;
        let w8 = cs.domain.d8.group_gen;
        let w = cs.domain.d1.group_gen;
        let mut x8 = Vec::with_capacity(n8);
        let mut acc = G::ScalarField::one();
        for _ in 0..n8 {
            x8.push(acc);
            acc *= w8;
        }
        let nf = G::ScalarField::from(n as u64);
        let unnormalized_lagrange = |i: usize| -> Vec<G::ScalarField> {
            let a = w.pow([i as u64]);
            let mut den: Vec<G::ScalarField> = x8.iter().map(|x| *x - a).collect();
            batch_inversion(&mut den); // zeros (x = a) stay zero
            let z8 = w8.pow([n as u64]); // x^n on d8 takes the eight values z8^k
            let mut zk_ = G::ScalarField::one();
            let xn: Vec<G::ScalarField> = (0..8)
                .map(|_| {
                    let v = zk_;
                    zk_ *= z8;
                    v
                })
                .collect();
            (0..n8).map(|k| if k == 8 * i { nf * a.inverse().unwrap() } else { (xn[k % 8] - G::ScalarField::one()) * den[k] }).collect()
        };
        let mut atoms: Vec<G::ScalarField> = cs.precomputations().vanishes_on_zero_knowledge_and_previous_rows.evals.clone();
        atoms.extend(unnormalized_lagrange(0));
        atoms.extend(unnormalized_lagrange(n - zk - 1));
        // ---- to the device; coefficient forms and 8x extensions of the selector block there
        let alloc = |elems: usize| {
            let mut p: *mut core::ffi::c_void = core::ptr::null_mut();
            ok(unsafe { sys::kh_dev_alloc(&mut p, 32 * elems.max(1)) });
            DevCols(p as *mut u64)
        };
        let (s1, sc, s8, t1, a8) = (alloc(nsel * n), alloc(nsel * n), alloc(nsel * n8), alloc(tab1.len()), alloc(3 * n8));
        unsafe {
            ok(sys::kh_dev_upload(s1.0 as *mut _, sel1.as_ptr() as *const _, 32 * nsel * n));
            ok(sys::kh_dev_copy(sc.0 as *mut _, s1.0 as *const _, 32 * nsel * n));
            ok(sys::kh_ntt_dev(field, sc.0, log2_n, 1, nsel));
            ok(sys::kh_lde_dev(field, sc.0, log2_n, 3, s8.0, nsel));
            ok(sys::kh_dev_upload(t1.0 as *mut _, tab1.as_ptr() as *const _, 32 * tab1.len()));
            ok(sys::kh_dev_upload(a8.0 as *mut _, atoms.as_ptr() as *const _, 32 * 3 * n8));
        }
        let col = |base: &DevCols, k: usize, len: usize| unsafe { base.0.add(4 * k * len) as *const u64 };
        let np = patterns.len();
        let (p1, pc, p8): (Vec<_>, Vec<_>, Vec<_>) = ((0..np).map(|k| col(&s1, k, n)).collect(), (0..np).map(|k| col(&sc, k, n)).collect(), (0..np).map(|k| col(&s8, k, n8)).collect());
        let ntab = lcs.lookup_table8.len();
        let tcols: Vec<*const u64> = (0..ntab).map(|k| col(&t1, k, n)).collect();
        let ids = if lcs.table_ids8.is_some() { col(&t1, ntab, n) } else { core::ptr::null() };
        let at: Vec<*const u64> = (0..3).map(|k| col(&a8, k, n8)).collect();
        ok(unsafe { sys::kh_prover_index_attach_lookup(index, patterns.as_ptr(), np, p1.as_ptr(), pc.as_ptr(), p8.as_ptr(), tcols.as_ptr(), ntab, ids, at.as_ptr()) });
        let runtime = match (&lcs.runtime_selector, &lcs.runtime_tables, lcs.runtime_table_offset) {
            (Some(_), Some(specs), Some(offset)) => {
                let length: usize = specs.iter().map(|t| t.len).sum();
                ok(unsafe { sys::kh_prover_index_attach_runtime_tables(index, col(&s1, np, n), col(&sc, np, n), col(&s8, np, n8), offset, length) });
                Some(specs.iter().map(|t| (t.id, t.len)).collect())
            }
            _ => None,
        };
        cols.extend([s1, sc, s8, t1, a8]);
        let mut present = [false; 4];
        for &k in &patterns {
            present[k as usize] = true;
        }
        LookupShape { patterns: present, max_per_row: lcs.configuration.lookup_info.max_per_row, runtime }
    }

    /// `ProverProof::create::<EFqSponge, EFrSponge, _>(group_map, witness, &[], index, rng)`.
    pub fn create<const FULL_ROUNDS: usize>(
        &self,
        witness: &[Vec<G::ScalarField>; COLUMNS],
        rng: &mut (impl RngCore + CryptoRng),
    ) -> ProverProof<G, OpeningProof<G, FULL_ROUNDS>, FULL_ROUNDS>
    where
        G: KimchiCurve<FULL_ROUNDS>,
    {
        self.create_recursive::<FULL_ROUNDS>(witness, &[], Vec::new(), rng)
    }

    /// `ProverProof::create_recursive::<EFqSponge, EFrSponge, _>(group_map, witness, runtime_tables, index, prev_challenges, None, rng)`
    /// (prover.rs:187-195): the blinders and zero-knowledge rows are drawn from `rng` in the reference's order (`ScalarField::rand` per
    /// element), so a seeded `rng` reproduces the reference's bytes.  `runtime_tables` in the configured order, as the reference demands
    /// (prover.rs:397-420: anything else is `ProverError::RuntimeTablesInconsistent`, here a panic with that name).
    pub fn create_recursive<const FULL_ROUNDS: usize>(
        &self,
        witness: &[Vec<G::ScalarField>; COLUMNS],
        runtime_tables: &[RuntimeTable<G::ScalarField>],
        prev_challenges: Vec<RecursionChallenge<G>>,
        rng: &mut (impl RngCore + CryptoRng),
    ) -> ProverProof<G, OpeningProof<G, FULL_ROUNDS>, FULL_ROUNDS>
    where
        G: KimchiCurve<FULL_ROUNDS>,
    {
        let rows = witness[0].len();
        let flat: Vec<G::ScalarField> = witness.iter().flat_map(|c| c.iter().copied()).collect();
        assert_eq!(prev_challenges.len(), self.prev_challenges, "the index was built for another number of previous challenges");
        // ---- runtime tables: the second column of the configured rows, concatenated
        let configured = self.lookup.as_ref().and_then(|l| l.runtime.clone()).unwrap_or_default();
        let given: Vec<(i32, usize)> = runtime_tables.iter().map(|t| (t.id, t.data.len())).collect();
        assert!(given == configured, "RuntimeTablesInconsistent");
        let runtime_values: Vec<G::ScalarField> = runtime_tables.iter().flat_map(|t| t.data.iter().copied()).collect();
        // ---- previous challenges: all challenges / all commitment chunks concatenated, with their counts
        let prev_chals: Vec<G::ScalarField> = prev_challenges.iter().flat_map(|p| p.chals.iter().copied()).collect();
        let prev_rounds: Vec<u32> = prev_challenges.iter().map(|p| p.chals.len() as u32).collect();
        let prev_points: Vec<G> = prev_challenges.iter().flat_map(|p| p.comm.chunks.iter().copied()).collect();
        let prev_chunks: Vec<usize> = prev_challenges.iter().map(|p| p.comm.chunks.len()).collect();
        let (prev_xy, prev_inf) = pack(&prev_points);
        let need = unsafe { sys::kh_prove_randomness_count(self.index, 1) };
        let rnd: Vec<G::ScalarField> = (0..need).map(|_| <G::ScalarField as ark_ff::UniformRand>::rand(rng)).collect();
        let mut proof = core::ptr::null_mut();
        ok(unsafe {
            sys::kh_prove_full(
                self.index,
                limbs(&flat),
                rows,
                core::ptr::null(),
                limbs(&rnd),
                need,
                sys::KH_PROVE_CHECK as u32,
                limbs(&prev_chals),
                prev_rounds.as_ptr(),
                prev_xy.as_ptr(),
                prev_inf.as_ptr(),
                prev_chunks.as_ptr(),
                prev_challenges.len(),
                limbs(&runtime_values),
                runtime_values.len(),
                &mut proof,
            )
        });
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
        // the lookup argument's polynomials follow (kimchi_hip.h, kh_prover_index_attach_lookup / _attach_runtime_tables): sorted x (max_per_row + 1),
        // aggregation, combined table, [runtime table, runtime selector], one selector per pattern in the order xor, lookup, range_check, ffmul
        let lk0 = 43 + self.optional.len();
        let nsorted = self.lookup.as_ref().map_or(0, |l| l.max_per_row + 1);
        let has_rt = self.lookup.as_ref().map_or(false, |l| l.runtime.is_some());
        let lk = |j: usize| self.lookup.as_ref().map(|_| ev(lk0 + j));
        let sel0 = lk0 + nsorted + 2 + if has_rt { 2 } else { 0 };
        let pattern_sel = |k: usize| {
            self.lookup.as_ref().and_then(|l| if l.patterns[k] { Some(ev(sel0 + l.patterns[..k].iter().filter(|&&p| p).count())) } else { None })
        };
        let lookup_comm = self.lookup.as_ref().map(|_| {
            let sorted = points(sys::KH_PROOF_LOOKUP_SORTED_COMM);
            let runtime = points(sys::KH_PROOF_LOOKUP_RUNTIME_COMM);
            LookupCommitments {
                sorted: sorted.chunks(nch).map(comm).collect(),
                aggreg: comm(&points(sys::KH_PROOF_LOOKUP_AGGREG_COMM)),
                runtime: if has_rt { Some(comm(&runtime)) } else { None },
            }
        });
        let pe = elems(sys::KH_PROOF_PUBLIC_EVALS);
        let lr = points(sys::KH_PROOF_LR);
        let z12 = elems(sys::KH_PROOF_Z1_Z2);
        let out = ProverProof {
            commitments: ProverCommitments {
                w_comm: core::array::from_fn(|i| comm(&w[i * nch..(i + 1) * nch])),
                z_comm: comm(&points(sys::KH_PROOF_Z_COMM)),
                t_comm: comm(&points(sys::KH_PROOF_T_COMM)),
                lookup: lookup_comm,
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
                lookup_aggregation: lk(nsorted),
                lookup_table: lk(nsorted + 1),
                lookup_sorted: core::array::from_fn(|j| if j < nsorted { lk(j) } else { None }),
                runtime_lookup_table: if has_rt { lk(nsorted + 2) } else { None },
                runtime_lookup_table_selector: if has_rt { lk(nsorted + 3) } else { None },
                xor_lookup_selector: pattern_sel(0),
                lookup_gate_lookup_selector: pattern_sel(1),
                range_check_lookup_selector: pattern_sel(2),
                foreign_field_mul_lookup_selector: pattern_sel(3),
            },
            ft_eval1: elems(sys::KH_PROOF_FT_EVAL1)[0],
            prev_challenges,
        };
        let _ = pack::<G>; // (the wire format helpers are shared with lib.rs)
        out
    }
}
