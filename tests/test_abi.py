"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/kimchi_hip.h declares; and it fails loudly (no CPU fallback)
when no GPU is present.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def khip():
    import __graft_entry__ as ge
    ge.build()                      # no-op when libkimchi_hip.so is up to date
    import proof_systems_amd.khip as k
    return k


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "kimchi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kh_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(khip):
    decl = _declared_functions()
    assert len(decl) >= 20
    lib = ctypes.CDLL(khip.LIB_PATH)
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in kimchi_hip.h but not exported"
    assert sorted(khip.SYMBOLS) == decl


def test_binding_declares_the_argument_types_of_every_wide_argument(khip):
    """ctypes passes an undeclared Python int as a 32-bit C int: a function with a size_t / uint64_t / pointer parameter that the binding calls without
    `argtypes` gets garbage in the upper halves (round 6: kh_msm_submit_host asked hipMalloc for 6.6 EB).  Every function of the header that has such a
    parameter and that the binding CALLS must have its argument list declared, with the header's arity."""
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "kimchi_hip.h")).read(), flags=re.S)
    decls = {m.group(1): m.group(2) for m in re.finditer(r"\b(kh_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", src)}
    py = open(os.path.join(ROOT, "proof_systems_amd", "khip.py")).read()
    called = set(re.findall(r"_lib\.(kh_[a-z_0-9]+)\(", py))
    lib = khip.raw()
    for name in sorted(called):
        args = decls[name].strip()
        if args in ("", "void"):
            continue
        wide = any(t in args for t in ("size_t", "uint64_t", "*", "["))
        at = getattr(lib, name).argtypes
        if wide:
            assert at is not None, f"{name}({args}): argtypes not declared in proof_systems_amd/khip.py"
        if at is not None:
            assert len(at) == len([a for a in args.split(",") if a.strip()]), f"{name}: {len(at)} argtypes for ({args})"


def test_no_oracle_in_product():
    """The product path must never route through the oracle or any CPU fallback."""
    pkg = os.path.join(ROOT, "proof_systems_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cuh", ".inc", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pasta_ref" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_fails_loudly_without_gpu(khip):
    if khip.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(khip.KhError) as e:
        khip.init(0)
    assert "no CPU fallback" in str(e.value)
    import numpy as np
    with pytest.raises(khip.KhError):
        khip.ntt(khip.FP, np.zeros((4, 4), np.uint64), 2)


def test_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/kimchi_hip.h must compile as strict C99 (a cgo / bindgen / ctypes consumer
    sees exactly this file)."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "kimchi_hip.h"\nint main(void) { return KH_OK + KH_TOK_LOAD - KH_TOK_LOAD + KH_SCAN_ADD; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-fsyntax-only", str(src)])
