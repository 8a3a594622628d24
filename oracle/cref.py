"""ctypes loader for the C oracle (oracle/pasta_ref.c -> oracle/_build/libpasta_ref.so).

TEST INFRASTRUCTURE ONLY (see oracle/pasta.py header).  All field elements on
this interface are numpy uint64 arrays of 4 little-endian limbs in Montgomery
form (the ark-ff in-memory layout), points are x||y = 8 limbs.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpasta_ref.so")
_lib = None

U64P = C.POINTER(C.c_uint64)
U8P = C.POINTER(C.c_uint8)


def _cpu_has(flag: str) -> bool:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pasta_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        # BMI2 / ADX: the hosts an MI355X sits in have both (the product's host code requires them too); a build machine without them falls back
        fast = ["-mbmi2", "-madx"] if _cpu_has("bmi2") and _cpu_has("adx") else []
        subprocess.check_call(["gcc", "-O3"] + fast + ["-fPIC", "-shared", "-pthread", "-o", _SO, src])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ko_init()
    return _lib


def _p64(a):
    return a.ctypes.data_as(U64P)


def _p8(a):
    return None if a is None else a.ctypes.data_as(U8P)


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def field_op(field: int, op: str, a, b=None):
    ops = {"mul": 0, "add": 1, "sub": 2, "to_mont": 3, "from_mont": 4, "inv": 5, "sqr": 6}
    a = _c64(a).reshape(-1, 4)
    out = np.empty_like(a)
    bb = None
    if b is not None:
        bb = _c64(b).reshape(-1, 4)
        assert bb.shape == a.shape
    rc = lib().ko_field_op(field, ops[op], _p64(a), _p64(bb) if bb is not None else None, _p64(out), C.c_size_t(a.shape[0]))
    assert rc == 0
    return out


def msm(curve: int, xy, scalars, inf=None, scalars_mont=True, threads=1, naive=False):
    xy = _c64(xy).reshape(-1, 8)
    sc = _c64(scalars).reshape(-1, 4)
    n = min(xy.shape[0], sc.shape[0])
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.zeros(8, dtype=np.uint64)
    oinf = C.c_uint8(0)
    if naive:
        rc = lib().ko_msm_naive(curve, _p64(xy), _p8(inf), _p64(sc), C.c_size_t(n), int(scalars_mont), _p64(out), C.byref(oinf))
    else:
        rc = lib().ko_msm(curve, _p64(xy), _p8(inf), _p64(sc), C.c_size_t(n), int(scalars_mont), int(threads), _p64(out), C.byref(oinf))
    assert rc == 0
    return out, bool(oinf.value)


def last_threads() -> int:
    """threads actually used by the last msm() call"""
    return lib().ko_last_threads()


def ntt(field: int, data, log2_n: int, inverse: bool, threads: int = 1):
    d = np.array(data, dtype=np.uint64, copy=True).reshape(-1, 1 << log2_n, 4)
    rc = lib().ko_ntt(field, _p64(d), log2_n, int(inverse), C.c_size_t(d.shape[0]), threads)
    assert rc == 0
    return d


def lde(field: int, coeffs, log2_n: int, log2_blowup: int, threads: int = 1):
    c = _c64(coeffs).reshape(-1, 1 << log2_n, 4)
    out = np.empty((c.shape[0], 1 << (log2_n + log2_blowup), 4), dtype=np.uint64)
    rc = lib().ko_lde(field, _p64(c), log2_n, log2_blowup, _p64(out), C.c_size_t(c.shape[0]), threads)
    assert rc == 0
    return out


def srs_generate(curve: int, start: int, count: int, threads: int = 1):
    out = np.empty((count, 8), dtype=np.uint64)
    rc = lib().ko_srs_generate(curve, C.c_size_t(start), C.c_size_t(count), _p64(out), threads)
    assert rc == 0
    return out


def srs_h(curve: int):
    out = np.empty(8, dtype=np.uint64)
    lib().ko_srs_h(curve, _p64(out))
    return out


def compress(curve: int, xy, inf=None):
    xy = _c64(xy).reshape(-1, 8)
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.empty((xy.shape[0], 33), dtype=np.uint8)
    lib().ko_compress(curve, _p64(xy), _p8(inf), C.c_size_t(xy.shape[0]), out.ctypes.data_as(U8P))
    return out


def point_add(curve: int, p, q, p_inf=False, q_inf=False):
    p = _c64(p).reshape(8); q = _c64(q).reshape(8)
    out = np.zeros(8, dtype=np.uint64); oinf = C.c_uint8(0)
    lib().ko_point_add(curve, _p64(p), int(p_inf), _p64(q), int(q_inf), _p64(out), C.byref(oinf))
    return out, bool(oinf.value)


def point_mul(curve: int, p, scalar, scalar_mont=True, p_inf=False):
    p = _c64(p).reshape(8); s = _c64(scalar).reshape(4)
    out = np.zeros(8, dtype=np.uint64); oinf = C.c_uint8(0)
    lib().ko_point_mul(curve, _p64(p), int(p_inf), _p64(s), int(scalar_mont), _p64(out), C.byref(oinf))
    return out, bool(oinf.value)


def lagrange_basis(curve: int, g_xy, log2_n: int, chunk: int = 0):
    g = _c64(g_xy).reshape(-1, 8)
    n = 1 << log2_n
    out = np.zeros((n, 8), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    rc = lib().ko_lagrange_basis(curve, _p64(g), C.c_size_t(g.shape[0]), log2_n, chunk, _p64(out), inf.ctypes.data_as(U8P))
    assert rc == 0
    return out, inf


# ---- int <-> limb helpers -------------------------------------------------
def ints_to_limbs(vals) -> np.ndarray:
    out = np.empty((len(vals), 4), dtype=np.uint64)
    m = (1 << 64) - 1
    for i, v in enumerate(vals):
        out[i, 0] = v & m; out[i, 1] = (v >> 64) & m; out[i, 2] = (v >> 128) & m; out[i, 3] = (v >> 192) & m
    return out


def limbs_to_ints(a) -> list:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]
