"""The Rust side of the drop-in boundary exists as source files (rust/kimchi-hip-sys, rust/kimchi-hip) but cannot be compiled
in this image.  What CAN be checked: the `extern "C"` block of the -sys crate declares exactly the functions
include/kimchi_hip.h declares, with the same arity and the same pointer-ness / const-ness / integer width of every
parameter and return value (an independent parser, not the generator's), every constant has the header's value, and the
safe crate only calls functions that exist with the number of arguments they take."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    src = open(os.path.join(ROOT, "include", "kimchi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"([A-Za-z_][\w \*]*?)\b(kh_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", src):
        args = " ".join(args.split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        out[name] = (ret.strip(), params)
    consts = dict(re.findall(r"#define\s+(KH_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", src))
    for body in re.findall(r"enum\s*\{([^}]*)\}", src):
        consts.update(dict(re.findall(r"(KH_[A-Z0-9_]+)\s*=\s*(-?\d+)", body)))
    return out, consts


def _shape_c(decl):
    """(pointer levels, const-ness of the pointee chain, base type) of a C parameter / return type."""
    decl = re.sub(r"\b[a-z_][a-z_0-9]*\s*(\[\d*\])?$", lambda m: "*" if m.group(1) else "", decl.strip()) if not decl.strip().endswith("*") and " " in decl.strip() else decl
    toks = decl.replace("*", " * ").split()
    base = [t for t in toks if t not in ("const", "*")][0]
    levels = toks.count("*")
    first_const = toks[0] == "const"
    return levels, first_const, base


def _shape_rs(t):
    levels = t.count("*")
    first = re.findall(r"\*(const|mut)", t)
    base = t.split()[-1]
    return levels, (first[-1] == "const") if first else False, base       # the innermost pointer's mutability describes the base


MAP = {"int": "c_int", "unsigned": "c_uint", "size_t": "usize", "uint64_t": "u64", "uint8_t": "u8", "uint32_t": "u32", "float": "f32", "double": "f64", "void": "c_void", "char": "c_char"}


def test_sys_crate_matches_the_header():
    hdr, consts = _header()
    rs = open(os.path.join(ROOT, "rust", "kimchi-hip-sys", "src", "lib.rs")).read()
    block = rs[rs.index('extern "C" {'):]
    fns = {m.group(1): (m.group(2), m.group(3)) for m in re.finditer(r"pub fn (kh_[a-z_0-9]+)\((.*?)\)(?: -> ([^;]+))?;", block)}
    assert sorted(fns) == sorted(hdr), (set(fns) ^ set(hdr))
    for name, (ret, params) in hdr.items():
        rparams = [p.split(":", 1)[1].strip() for p in fns[name][0].split(", ") if p.strip()]
        assert len(rparams) == len(params), (name, params, rparams)
        for cp, rp in zip(params, rparams):
            # pointer depth and base type
            m = re.match(r"(.*?)([A-Za-z_]\w*)\s*(\[\d*\])?$", cp)
            ctype = m.group(1).strip() + (" *" if m.group(3) else "")
            cl = ctype.count("*"); cbase = [t for t in ctype.replace("*", " ").split() if t != "const"][0]
            assert rp.count("*") == cl, (name, cp, rp)
            assert rp.split()[-1] == MAP.get(cbase, cbase), (name, cp, rp)
            if cl >= 1:
                innermost = re.findall(r"\*(const|mut)", rp)[-1]          # mutability of the data itself
                assert (innermost == "const") == ctype.startswith("const"), (name, cp, rp)
        rret = fns[name][1]
        if ret == "void":
            assert rret is None, name
        else:
            assert rret is not None and rret.strip().split()[-1] == MAP.get(ret.replace("const", "").replace("*", "").strip(), ret) and rret.count("*") == ret.count("*"), (name, ret, rret)
    rconst = dict(re.findall(r"pub const (KH_[A-Z0-9_]+): c_int = (-?\d+);", rs))
    assert rconst == consts


def test_generated_file_is_current():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "tools", "gen_rust_sys.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    assert open(os.path.join(ROOT, "rust", "kimchi-hip-sys", "src", "lib.rs")).read() == gen.render()


def test_safe_crate_calls_exist_with_the_right_arity():
    hdr, _ = _header()
    for f in ("lib.rs", "ntt.rs", "prover.rs"):
        src = open(os.path.join(ROOT, "rust", "kimchi-hip", "src", f)).read()
        src = re.sub(r"//.*", "", src)
        for m in re.finditer(r"sys::(kh_[a-z_0-9]+)\s*\(", src):
            name = m.group(1)
            assert name in hdr, f"{f}: sys::{name} is not in the header"
            depth, i, args, cur = 1, m.end(), 0, ""
            while depth:
                ch = src[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == "," and depth == 1:
                    args += 1 if cur.strip() else 0; cur = ""; i += 1; continue
                cur += ch; i += 1
            args += 1 if cur.strip(" )\n\t") else 0
            assert args == len(hdr[name][1]), (f, name, args, len(hdr[name][1]))
    # the trait surface: every method of poly_commitment::SRS (lib.rs:61-241) is implemented
    src = open(os.path.join(ROOT, "rust", "kimchi-hip", "src", "lib.rs")).read()
    for method in ("max_poly_size", "blinding_commitment", "mask_custom", "mask", "commit_non_hiding", "commit", "commit_custom", "commit_evaluations_non_hiding",
                   "commit_evaluations", "commit_evaluations_custom", "create", "get_lagrange_basis", "get_lagrange_basis_from_domain_size", "size"):
        assert re.search(r"fn %s\b" % method, src), method
    assert re.search(r"impl<G: HipCurve.*?, const FULL_ROUNDS: usize> OpenProof<G, FULL_ROUNDS> for GpuOpeningProof<G, FULL_ROUNDS>", src)


def _path_deps(cargo_toml: str):
    """names of the `path = ...` / plain dependencies a Cargo.toml (or a patch adding lines to one) declares"""
    return set(re.findall(r"^\+?\s*([a-z][a-z0-9_-]*)\s*=\s*(?:\{[^}]*\}|\"[^\"]*\")", cargo_toml, flags=re.M))


def test_no_dependency_cycle():
    """kimchi-hip-sys <- ark-poly fork <- poly-commitment <- kimchi-hip: the fork must bind the -sys crate only.  Builds the crate graph
    from rust/*/Cargo.toml, the Cargo.toml hunk + `use` lines of the ark-poly patch and the reference's known edge
    poly-commitment -> ark-poly (poly-commitment/Cargo.toml), and fails on any cycle."""
    g = {}
    for crate in ("kimchi-hip-sys", "kimchi-hip"):
        toml = open(os.path.join(ROOT, "rust", crate, "Cargo.toml")).read()
        deps = toml[toml.index("[dependencies]"):] if "[dependencies]" in toml else ""
        deps = re.sub(r"^#.*$", "", deps, flags=re.M)
        g[crate] = _path_deps(deps) - {"version", "edition", "path"}
    patch = open(os.path.join(ROOT, "rust", "ark-poly-patch", "radix2_fft_in_place.patch")).read()
    hunk = patch[patch.index("+++ b/poly/Cargo.toml"):patch.index("--- a/poly/src")]
    added = _path_deps("\n".join(l for l in hunk.splitlines() if l.startswith("+") and not l.startswith("+#") and not l.startswith("+++")))
    used = set(re.findall(r"^\+use ([a-z_]+)", patch, flags=re.M)) | set(re.findall(r"\b(kimchi_hip(?:_sys)?)::", patch))
    assert {u.replace("_", "-") for u in used if u.startswith("kimchi")} <= added, (used, added)       # every kimchi crate the patch uses is declared
    g["ark-poly"] = added
    g["poly-commitment"] = {"ark-poly"}                                     # /root/reference/poly-commitment/Cargo.toml: ark-poly.workspace = true
    assert g["kimchi-hip-sys"] == set(), g["kimchi-hip-sys"]
    assert "kimchi-hip" not in g["ark-poly"] and g["ark-poly"] == {"kimchi-hip-sys"}

    def reach(a, seen=()):
        for b in g.get(a, ()):
            assert b not in seen + (a,), f"dependency cycle through {a} -> {b}"
            reach(b, seen + (a,))
    for c in g:
        reach(c)
    # the forward transform must not zero-extend on the host before the call (the padding would cross PCIe): no resize to self.size() in fft_in_place
    fwd = re.sub(r"//.*", "", patch[patch.index("fn fft_in_place"):patch.index("fn ifft_in_place")])
    assert "resize" not in fwd and "kimchi_hip_dispatch::forward" in fwd
    assert "TypeId" not in patch.replace("`TypeId::of::<T>()`", "")        # DomainCoeff has no 'static bound
    assert "sys::kh_lde(" in patch and "sys::kh_ntt(" in patch


def test_rust_sources_are_delimiter_balanced():
    """no compiler here: at least every bracket of the Rust files closes (comments, strings, char literals and lifetimes stripped)"""
    for f in ("kimchi-hip/src/lib.rs", "kimchi-hip/src/prover.rs", "kimchi-hip/src/ntt.rs", "kimchi-hip-sys/src/lib.rs"):
        s = open(os.path.join(ROOT, "rust", f)).read()
        s = re.sub(r"//[^\n]*", "", s); s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
        s = re.sub(r'"(?:\\.|[^"\\])*"', '""', s); s = re.sub(r"'(?:\\.|[^'\\])'", "''", s)
        stack, pairs = [], {")": "(", "]": "[", "}": "{"}
        for ch in s:
            if ch in "([{":
                stack.append(ch)
            elif ch in ")]}":
                assert stack and stack.pop() == pairs[ch], f
        assert not stack, f
