"""The Rust crates under rust/ cannot be compiled in this image (no toolchain).  Beyond the FFI names (tests/test_rust_bindings.py) two
classes of error can still be caught by reading the reference's sources:

  * every `use` path into a crate of o1-labs/proof-systems must resolve to an item that is `pub` all the way down (round 3 imported
    `poly_commitment::PolynomialsToCombine`, a PRIVATE alias -- poly-commitment/src/lib.rs:249 -- which is E0603 for an outside crate);
  * every method of `impl SRS<G> for GpuSrs<G>` and `impl OpenProof<..> for GpuOpeningProof<..>` must have the trait's signature token for
    token (after the spelling differences an impl is allowed: `mut` on a parameter, the name of a type parameter, the private alias written out),
    and must not ask more of its type parameters than the trait does (round 3 cloned an `EFqSponge` that is only `FqSponge` in `verify`).

Reads /root/reference (present in the build container, absent on the GPU box: skipped there)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is not on this machine")

CRATES = {"poly_commitment": "poly-commitment/src", "kimchi": "kimchi/src", "mina_poseidon": "poseidon/src", "mina_curves": "curves/src",
          "groupmap": "groupmap/src", "o1_utils": "utils/src"}
RUST_FILES = [os.path.join(ROOT, "rust", "kimchi-hip", "src", f) for f in ("lib.rs", "ntt.rs", "prover.rs")]


def _strip(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def _use_leaves(src: str):
    """all (crate, [segments]) leaves of the `use` trees of a file"""
    out = []

    def expand(prefix, tree):
        tree = tree.strip()
        if not tree:
            return
        if "{" not in tree:
            for part in tree.split(","):
                part = part.strip()
                if not part:
                    continue
                part = re.sub(r"\s+as\s+\w+$", "", part)
                segs = prefix + [s for s in part.split("::") if s]
                if segs and segs[-1] == "self":
                    segs = segs[:-1]
                out.append(segs)
            return
        # split at top-level commas
        depth, cur, parts = 0, "", []
        for ch in tree:
            if ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur); cur = ""
            else:
                cur += ch
        parts.append(cur)
        for part in parts:
            part = part.strip()
            if not part:
                continue
            m = re.match(r"^([\w:]*?)(?:::)?\{(.*)\}$", part, flags=re.S)
            if m:
                expand(prefix + [s for s in m.group(1).split("::") if s], m.group(2))
            else:
                expand(prefix, part)

    for m in re.finditer(r"^\s*(?:pub\s+)?use\s+([^;]+);", _strip(src), flags=re.M):
        expand([], m.group(1))
    return [(s[0], s[1:]) for s in out if s and s[0] in CRATES]


def _module_file(base: str, mods):
    """source file of crate-relative module path `mods` (lib.rs for the root)"""
    d = os.path.join(REF, base)
    if not mods:
        return os.path.join(d, "lib.rs")
    p = os.path.join(d, *mods)
    for cand in (p + ".rs", os.path.join(p, "mod.rs")):
        if os.path.exists(cand):
            return cand
    return None


def _declares_pub(src: str, name: str):
    """'pub' / 'private' / None: how `name` is declared at the top level of a module's source"""
    src = _strip(src)
    decl = r"(?:unsafe\s+)?(?:async\s+)?(?:const\s+)?(?:fn|struct|enum|trait|type|const|static|mod|union)\s+%s\b" % re.escape(name)
    if re.search(r"^\s*pub\s+" + decl, src, flags=re.M):
        return "pub"
    for m in re.finditer(r"^\s*pub\s+use\s+([^;]+);", src, flags=re.M):
        if re.search(r"\b%s\b" % re.escape(name), m.group(1)):
            return "pub"
    if re.search(r"^\s*(?:pub\s*\([^)]*\)\s+)?" + decl, src, flags=re.M):
        return "private"
    if re.search(r"^\s*(?:pub\s*\([^)]*\)\s+)?use\s+[^;]*\b%s\b" % re.escape(name), src, flags=re.M):
        return "private"
    return None


def _resolve(crate: str, segs):
    """None if crate::segs names a public item, else a description of what is wrong"""
    base = CRATES[crate]
    mods = []
    for k, seg in enumerate(segs):
        f = _module_file(base, mods)
        if f is None:
            return f"no source file for module {'::'.join([crate] + mods)}"
        how = _declares_pub(open(f).read(), seg)
        if how != "pub":
            # an associated item / enum variant of the previous segment (GateType::Poseidon): fine if the previous segment was a public type
            if k > 0 and how is None and _module_file(base, mods) is not None and not os.path.exists(os.path.join(REF, base, *mods, seg + ".rs")):
                return None
            return f"`{seg}` in {os.path.relpath(f, REF)} is {how or 'not declared'}"
        nxt = _module_file(base, mods + [seg])
        if nxt is not None and re.search(r"^\s*pub\s+mod\s+%s\b" % re.escape(seg), _strip(open(f).read()), flags=re.M):
            mods.append(seg)
        else:
            return None if k == len(segs) - 1 or True else None           # an item: whatever follows is an associated name
    return None


def test_the_checker_catches_round_3s_private_import():
    assert _resolve("poly_commitment", ["PolynomialsToCombine"]) is not None
    assert _resolve("poly_commitment", ["OpenProof"]) is None
    assert _resolve("poly_commitment", ["utils", "DensePolynomialOrEvaluations"]) is None
    assert _resolve("poly_commitment", ["combine", "combine_one_endo"]) is not None          # `mod combine` is private


def test_every_use_path_resolves_to_a_public_item():
    bad = []
    for f in RUST_FILES:
        for crate, segs in _use_leaves(open(f).read()):
            why = _resolve(crate, segs)
            if why:
                bad.append(f"{os.path.basename(f)}: {crate}::{'::'.join(segs)} -- {why}")
    assert not bad, "\n".join(bad)


# ---------------------------------------------------------------------------------------------------------------- signatures
def _block(src: str, header_re: str) -> str:
    """body (between the braces) of the first item whose header matches"""
    m = re.search(header_re, src, flags=re.S)
    assert m, header_re
    i = src.index("{", m.end() - 1) if src[m.end() - 1] != "{" else m.end() - 1
    depth, j = 0, i
    while True:
        if src[j] == "{":
            depth += 1
        elif src[j] == "}":
            depth -= 1
            if depth == 0:
                return src[i + 1:j]
        j += 1


def _signatures(body: str):
    """{name: signature text up to the body / semicolon} of the fns at depth 0 of a trait / impl body"""
    out, depth, i = {}, 0, 0
    body = _strip(body)
    while i < len(body):
        ch = body[i]
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        elif depth == 0 and body.startswith("fn ", i) and (i == 0 or not body[i - 1].isalnum()):
            j, d2 = i, 0
            while not (d2 == 0 and body[j] in ";{"):
                d2 += body[j] in "(<["
                d2 -= body[j] in ")>]"
                if body.startswith("->", j):
                    d2 += 1                                   # the `>` of `->` is not a closing bracket
                j += 1
            sig = body[i:j]
            out[re.match(r"fn\s+(\w+)", sig).group(1)] = sig
            i = j
            continue
        i += 1
    return out


ALIAS = "&[(DensePolynomialOrEvaluations<'_, <G as AffineRepr>::ScalarField, D>, PolyComm<<G as AffineRepr>::ScalarField>)]"


def _norm(sig: str) -> str:
    sig = re.sub(r"#\[[^\]]*\]", "", sig)
    sig = re.sub(r"\bmut\s+(\w+\s*:)", r"\1", sig)                      # `mut sponge: T` is the impl's business
    sig = re.sub(r"\bDom\b", "D", sig)                                    # the name of a type parameter is free
    sig = sig.replace("PolynomialsToCombine<G, D>", ALIAS)                # the private alias, written out (its one lifetime elided)
    sig = re.sub(r"\s+", "", sig)
    sig = sig.replace(",)", ")").replace(",>", ">")                       # trailing commas
    return sig.rstrip(",")


def _ref_trait(name: str) -> str:
    src = _strip(open(os.path.join(REF, "poly-commitment/src/lib.rs")).read())
    return _block(src, r"pub\s+trait\s+%s\b[^{]*\{" % name)


def _our_impl(trait_re: str) -> str:
    src = _strip(open(os.path.join(ROOT, "rust", "kimchi-hip", "src", "lib.rs")).read())
    return _block(src, r"impl<[^{]*?\b%s[^{]*\{" % trait_re)


@pytest.mark.parametrize("trait,impl_re", [("SRS", r"SRS<G>\s+for\s+GpuSrs<G>"), ("OpenProof", r"OpenProof<G,\s*FULL_ROUNDS>\s+for\s+GpuOpeningProof")])
def test_trait_method_signatures_equal_the_references(trait, impl_re):
    want, got = _signatures(_ref_trait(trait)), _signatures(_our_impl(impl_re))
    assert set(want) == set(got), set(want) ^ set(got)
    for name in want:
        assert _norm(got[name]) == _norm(want[name]), f"{trait}::{name}\n  ours: {_norm(got[name])}\n  ref : {_norm(want[name])}"


def test_verify_uses_nothing_the_trait_does_not_grant():
    """`verify`'s sponge is `FqSponge` only: no `.clone()` on it (round 3's second compile error), and the batch is handed to the inner SRS."""
    body = _block(_our_impl(r"OpenProof<G,\s*FULL_ROUNDS>\s+for\s+GpuOpeningProof"), r"fn\s+verify\b.*?\)\s*->\s*bool\s*where[^{]*\{")
    assert "sponge.clone()" not in body and ".sponge.clone" not in body
    assert "srs.inner.verify(" in body and "other_curve_sponge_params" in body


def test_proof_handles_are_freed_on_every_path():
    src = _strip(open(os.path.join(ROOT, "rust", "kimchi-hip", "src", "prover.rs")).read())
    assert re.search(r"impl\s+Drop\s+for\s+ProofGuard", src) and "kh_proof_free" in _block(src, r"impl\s+Drop\s+for\s+ProofGuard\s*\{")
    assert len(re.findall(r"sys::kh_proof_free", src)) == 1               # only the guard frees


# ---------------------------------------------------------------------------------------------------------------- struct literals
def _struct_fields(path: str, name: str):
    """names of the `pub` fields of `pub struct name` in a reference source file"""
    body = _block(_strip(open(os.path.join(REF, path)).read()), r"pub\s+struct\s+%s\b[^{;]*\{" % name)
    clean, i = "", 0
    while i < len(body):                                     # drop #[...] attributes (their arguments may nest brackets and hold strings)
        if body.startswith("#[", i):
            depth, i = 1, i + 2
            while depth:
                depth += body[i] == "["
                depth -= body[i] == "]"
                i += 1
        else:
            clean += body[i]; i += 1
    body = clean
    out, depth, cur = [], 0, ""
    for ch in body + ",":
        if ch in "<([{":
            depth += 1
        elif ch in ">)]}":
            depth -= 1
        if ch == "," and depth == 0:
            m = re.match(r"\s*pub\s+(\w+)\s*:", cur)
            if m:
                out.append(m.group(1))
            cur = ""
        else:
            cur += ch
    return out


def _literal_fields(src: str, name: str):
    """field names of the first struct literal `name { ... }` in our source (a field may be shorthand: `prev_challenges,`)"""
    m = re.search(r"\b%s\s*\{" % name, src)
    assert m, name
    depth, j = 0, m.end() - 1
    while True:
        depth += src[j] == "{"
        depth -= src[j] == "}"
        if depth == 0:
            break
        j += 1
    body, out, depth, cur = src[m.end():j], [], 0, ""
    for ch in body + ",":
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            mm = re.match(r"\s*(\w+)\s*(?::|$)", cur.strip())
            if mm and cur.strip():
                out.append(mm.group(1))
            cur = ""
        else:
            cur += ch
    return out


@pytest.mark.parametrize("name,path", [("ProverProof", "kimchi/src/proof.rs"), ("ProverCommitments", "kimchi/src/proof.rs"), ("ProofEvaluations", "kimchi/src/proof.rs"),
                                       ("LookupCommitments", "kimchi/src/proof.rs"), ("PointEvaluations", "kimchi/src/proof.rs"),
                                       ("OpeningProof", "poly-commitment/src/ipa.rs"), ("BlindedCommitment", "poly-commitment/src/commitment.rs")])
def test_struct_literals_name_exactly_the_references_fields(name, path):
    src = "".join(_strip(open(f).read()) for f in RUST_FILES)
    got, want = _literal_fields(src, name), _struct_fields(path, name)
    assert want, name
    assert sorted(got) == sorted(want), (name, sorted(set(got) ^ set(want)))


def test_fields_read_from_reference_values_exist():
    """`x.field` accesses on reference types in prover.rs: the fields must be public in the reference"""
    for path, struct, fields in [("kimchi/src/circuits/lookup/index.rs", "LookupConstraintSystem", ["lookup_table8", "table_ids8", "lookup_selectors", "runtime_selector",
                                                                                                "runtime_tables", "runtime_table_offset", "configuration"]),
                                 ("kimchi/src/circuits/lookup/index.rs", "LookupSelectors", ["xor", "lookup", "range_check", "ffmul"]),
                                 ("kimchi/src/circuits/lookup/lookups.rs", "LookupInfo", ["max_per_row"]),
                                 ("kimchi/src/circuits/lookup/runtime_tables.rs", "RuntimeTable", ["id", "data"]),
                                 ("kimchi/src/circuits/lookup/runtime_tables.rs", "RuntimeTableSpec", ["id", "len"]),
                                 ("kimchi/src/proof.rs", "RecursionChallenge", ["chals", "comm"]),
                                 ("kimchi/src/circuits/constraints.rs", "ConstraintSystem", ["public", "prev_challenges", "domain", "gates", "zk_rows", "sid", "shift", "lookup_constraint_system"]),
                                 ("kimchi/src/circuits/domain_constant_evaluation.rs", "DomainConstantEvaluations", ["vanishes_on_zero_knowledge_and_previous_rows"])]:
        have = _struct_fields(path, struct)
        for f in fields:
            assert f in have, (struct, f, have)
    src = _strip(open(RUST_FILES[2]).read())
    # GpuProver::from_index reads the reference's ProverIndex (kimchi/src/prover_index.rs:26-57): every `index.<field>` must be one of its pub fields,
    # and the three the shim relies on must be there with the types it assumes (Arc<ConstraintSystem>, Arc<Srs>, usize, Option<G::BaseField>)
    pix = _struct_fields("kimchi/src/prover_index.rs", "ProverIndex")
    reads = set(m.group(1) for m in re.finditer(r"\bindex\.(\w+)\b(?!\s*\()", src))
    assert {"cs", "srs", "max_poly_size", "verifier_index_digest"} <= reads, reads
    for f in reads:
        assert f in pix, ("ProverIndex", f, pix)
    ref = _strip(open(os.path.join(REF, "kimchi/src/prover_index.rs")).read())
    for decl in (r"pub\s+cs\s*:\s*Arc\s*<\s*ConstraintSystem\s*<\s*G::ScalarField\s*>\s*>", r"pub\s+srs\s*:\s*Arc\s*<\s*Srs\s*>", r"pub\s+max_poly_size\s*:\s*usize",
                 r"pub\s+verifier_index_digest\s*:\s*Option\s*<\s*G::BaseField\s*>"):
        assert re.search(decl, ref), decl
    assert re.search(r"pub\s+struct\s+ProverIndex\s*<\s*const\s+FULL_ROUNDS\s*:\s*usize\s*,\s*G\s*:\s*KimchiCurve\s*<\s*FULL_ROUNDS\s*>\s*,\s*Srs\s*>", ref)
    for m in re.finditer(r"\blcs\.(\w+)\b(?!\s*\()", src):
        assert m.group(1) in _struct_fields("kimchi/src/circuits/lookup/index.rs", "LookupConstraintSystem"), m.group(1)
    for m in re.finditer(r"\bcs\.(\w+)\b(?!\s*\()", src):
        assert m.group(1) in _struct_fields("kimchi/src/circuits/constraints.rs", "ConstraintSystem"), m.group(1)
