"""ctypes binding of include/kimchi_hip.h (numpy uint64 limb arrays in, numpy out).

No fallback: if libkimchi_hip.so has not been built this module raises ImportError, and
every call raises KhError when the library reports a failure (e.g. no GPU).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KH_LIB") or os.path.join(_HERE, "libkimchi_hip.so")      # KH_LIB: another build of the library (same-box A/B of two builds)

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")

_lib = C.CDLL(LIB_PATH)

U64P = C.POINTER(C.c_uint64)
U8P = C.POINTER(C.c_uint8)
VESTA, PALLAS = 0, 1
FP, FQ = 0, 1
BASIS_G = -1
MSM_SLOTS = 4          # KH_MSM_SLOTS: jobs kh_msm_submit accepts before kh_msm_wait

# every symbol include/kimchi_hip.h declares
SYMBOLS = [
    "kh_sponge_new", "kh_sponge_clone", "kh_sponge_free", "kh_sponge_absorb_g", "kh_sponge_absorb", "kh_sponge_absorb_fr", "kh_sponge_challenge",
    "kh_sponge_challenge_field", "kh_sponge_squeeze_field", "kh_sponge_digest",
    "kh_group_map_to_group", "kh_dev_copy", "kh_dev_memset_zero", "kh_dev_fill_elements", "kh_set_phase_timers", "kh_ipa_open",
    "kh_device_count", "kh_init", "kh_set_device", "kh_get_device", "kh_trim", "kh_srs_device", "kh_last_error", "kh_srs_create", "kh_msm_set_wide_min_n", "kh_msm_set_sort_staging", "kh_ntt_set_max_logr", "kh_debug_rebase_points", "kh_debug_glv_split", "kh_msm_submit_host", "kh_counter", "kh_srs_set_wide_tables", "kh_srs_has_wide_tables", "kh_srs_free", "kh_srs_size",
    "kh_srs_set_lagrange", "kh_srs_compute_lagrange", "kh_srs_get_lagrange", "kh_srs_lagrange_chunks",
    "kh_msm", "kh_msm_batch", "kh_msm_points", "kh_ntt", "kh_lde",
    "kh_dev_alloc", "kh_dev_free", "kh_dev_upload", "kh_dev_download", "kh_dev_upload_2d", "kh_dev_upload_2d_unordered",
    "kh_msm_batch_dev", "kh_ntt_dev", "kh_lde_dev", "kh_coset_ntt_dev", "kh_sync", "kh_last_timings",
    "kh_debug_field_op", "kh_debug_point_op", "kh_srs_generate", "kh_srs_h",
    "kh_msm_sharded", "kh_msm_sharded_dev", "kh_gate_count", "kh_gate_name", "kh_gate_num_constants", "kh_gate_evaluations_dev", "kh_gate_constants", "kh_srs_curve", "kh_lookup_sorted", "kh_private_context_begin", "kh_private_context_end", "kh_private_context_active", "kh_comm_unique_id", "kh_comm_init", "kh_comm_free", "kh_comm_world_size", "kh_comm_rank", "kh_comm_allgather_points",
    "kh_msm_allreduce",
    "kh_prover_index_new", "kh_prover_index_attach_lookup", "kh_prover_index_free", "kh_prove_randomness_count", "kh_prove", "kh_prove_recursive", "kh_prove_full", "kh_prover_index_attach_runtime_tables", "kh_proof_section", "kh_proof_phase_seconds", "kh_proof_free",
    "kh_commit_non_hiding", "kh_commit_evaluations_non_hiding", "kh_srs_set_blinding_base",
    "kh_srs_get_blinding_base", "kh_mask_custom", "kh_domain_generator", "kh_msm_points_batch", "kh_msm_submit", "kh_msm_wait",
    "kh_ipa_fold_scalars", "kh_inner_product", "kh_ipa_fold_points", "kh_ipa_fold_points_endo", "kh_endos", "kh_scalar_challenge_to_field",
    "kh_polycomm_multi_scalar_mul", "kh_expr_evaluations_dev", "kh_field_scan_dev", "kh_batch_inversion_dev", "kh_divide_by_linear_dev", "kh_divide_by_linear_async_dev", "kh_check_equal_dev", "kh_b_poly_coefficients", "kh_batch_dlog_accumulator_generate", "kh_batch_dlog_accumulator_check", "kh_ipa_verify_msm",
    "kh_ipa_begin", "kh_ipa_begin_dev", "kh_combine_polys_dev", "kh_poly_lincomb_dev", "kh_b_init_dev", "kh_evaluate_chunks_dev", "kh_evaluate_chunks_batch_dev", "kh_divide_by_vanishing_poly_dev", "kh_ipa_rounds_left", "kh_ipa_round_lr", "kh_ipa_round_fold", "kh_ipa_finish", "kh_ipa_free", "kh_points_sum", "kh_points_add", "kh_srs_create_device", "kh_srs_create_device_range", "kh_srs_get_g",
]

_lib.kh_last_error.restype = C.c_char_p
_lib.kh_set_device.argtypes = [C.c_int]
_lib.kh_ipa_open.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, U64P, U64P, C.c_void_p, U64P, C.c_size_t, U64P, U8P, U64P, U8P, U64P, U64P, U64P, U8P]
_lib.kh_sponge_new.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
_lib.kh_sponge_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
_lib.kh_sponge_free.argtypes = [C.c_void_p]
_lib.kh_sponge_free.restype = None
_lib.kh_sponge_absorb_g.argtypes = [C.c_void_p, U64P, U8P, C.c_size_t]
_lib.kh_sponge_absorb.argtypes = [C.c_void_p, U64P, C.c_size_t]
_lib.kh_sponge_absorb_fr.argtypes = [C.c_void_p, U64P, C.c_size_t]
_lib.kh_sponge_challenge.argtypes = [C.c_void_p, U64P]
_lib.kh_sponge_challenge_field.argtypes = [C.c_void_p, U64P]
_lib.kh_sponge_squeeze_field.argtypes = [C.c_void_p, U64P]
_lib.kh_sponge_digest.argtypes = [C.c_void_p, U64P]
_lib.kh_srs_device.argtypes = [C.c_void_p]
_lib.kh_srs_size.restype = C.c_size_t
_lib.kh_srs_size.argtypes = [C.c_void_p]
_lib.kh_srs_free.restype = None
_lib.kh_srs_free.argtypes = [C.c_void_p]
_lib.kh_srs_create.argtypes = [C.c_int, U64P, C.c_size_t, C.POINTER(C.c_void_p)]
_lib.kh_msm_set_wide_min_n.argtypes = [C.c_size_t]
_lib.kh_msm_set_sort_staging.argtypes = [C.c_uint, C.c_uint]
_lib.kh_ntt_set_max_logr.argtypes = [C.c_uint]
_lib.kh_counter.argtypes = [C.c_char_p]
_lib.kh_counter.restype = C.c_uint64
_lib.kh_srs_set_wide_tables.argtypes = [C.c_void_p, C.c_int]
_lib.kh_srs_has_wide_tables.argtypes = [C.c_void_p]
_lib.kh_dev_fill_elements.argtypes = [C.c_void_p, U64P, C.c_size_t]
_lib.kh_srs_set_lagrange.argtypes = [C.c_void_p, C.c_uint, C.c_uint, U64P, U8P, C.c_size_t]
_lib.kh_srs_compute_lagrange.argtypes = [C.c_void_p, C.c_uint]
_lib.kh_srs_get_lagrange.argtypes = [C.c_void_p, C.c_uint, C.c_uint, U64P, U8P]
_lib.kh_srs_lagrange_chunks.argtypes = [C.c_void_p, C.c_uint]
_lib.kh_msm.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_size_t, U64P, C.c_size_t, C.c_int, U64P, U8P]
_lib.kh_msm_batch.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_size_t, U64P, C.c_size_t, C.c_size_t, C.c_int, U64P, U8P]
_lib.kh_msm_batch_dev.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, U64P, U8P]
_lib.kh_msm_points.argtypes = [C.c_int, U64P, U8P, U64P, C.c_size_t, C.c_int, U64P, U8P]
_lib.kh_ipa_fold_scalars.argtypes = [C.c_int, U64P, U64P, U64P, C.c_size_t, U64P]
_lib.kh_inner_product.argtypes = [C.c_int, U64P, U64P, C.c_size_t, U64P]
_lib.kh_ipa_fold_points.argtypes = [C.c_int, U64P, U64P, U64P, C.c_size_t, U64P, U8P]
_lib.kh_ipa_fold_points_endo.argtypes = [C.c_int, U64P, U64P, U64P, C.c_size_t, U64P, U8P]
_lib.kh_endos.argtypes = [C.c_int, U64P, U64P]
_lib.kh_scalar_challenge_to_field.argtypes = [C.c_int, U64P, U64P]
_lib.kh_expr_evaluations_dev.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t,
                                         U64P, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint, C.c_int, C.c_void_p]
_lib.kh_field_scan_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
_lib.kh_batch_inversion_dev.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
_lib.kh_divide_by_linear_dev.argtypes = [C.c_int, C.c_void_p, C.c_size_t, U64P, C.c_void_p, U64P]
_lib.kh_divide_by_linear_async_dev.argtypes = [C.c_int, C.c_void_p, C.c_size_t, U64P, C.c_void_p, C.c_void_p]
_lib.kh_check_equal_dev.argtypes = [C.c_void_p, C.c_size_t, U64P, C.c_void_p, C.c_uint]
_lib.kh_polycomm_multi_scalar_mul.argtypes = [C.c_int, U64P, U8P, C.POINTER(C.c_size_t), C.c_size_t, U64P, U64P, U8P, C.POINTER(C.c_size_t)]
_lib.kh_b_poly_coefficients.argtypes = [C.c_int, U64P, C.c_uint, C.c_size_t, U64P]
_lib.kh_batch_dlog_accumulator_generate.argtypes = [C.c_void_p, C.c_size_t, U64P, C.c_size_t, U64P, U8P]
_lib.kh_batch_dlog_accumulator_check.argtypes = [C.c_void_p, U64P, U8P, C.c_size_t, U64P, C.c_size_t, U64P, C.POINTER(C.c_int)]
_lib.kh_ipa_verify_msm.argtypes = [C.c_void_p, U64P, C.c_size_t, U64P, C.c_size_t, U64P, U8P, U64P, C.c_size_t, C.POINTER(C.c_int)]
_lib.kh_ipa_begin.argtypes = [C.c_void_p, U64P, C.c_size_t, U64P, C.c_size_t, U64P, C.POINTER(C.c_void_p)]
_lib.kh_ipa_begin_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, U64P, C.POINTER(C.c_void_p)]
_lib.kh_combine_polys_dev.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t, U64P, C.c_size_t,
                                      C.c_void_p, C.POINTER(C.c_size_t)]
_lib.kh_poly_lincomb_dev.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), U64P, C.c_size_t, C.c_void_p, C.c_size_t]
_lib.kh_b_init_dev.argtypes = [C.c_int, U64P, C.c_size_t, U64P, C.c_size_t, C.c_void_p]
_lib.kh_evaluate_chunks_dev.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, U64P, C.c_size_t, U64P]
_lib.kh_evaluate_chunks_batch_dev.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t,
                                              U64P, C.c_size_t, U64P]
_lib.kh_divide_by_vanishing_poly_dev.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_void_p]
_lib.kh_ipa_rounds_left.argtypes = [C.c_void_p]
_lib.kh_ipa_round_lr.argtypes = [C.c_void_p, U64P, U64P, U64P, U8P]
_lib.kh_ipa_round_fold.argtypes = [C.c_void_p, U64P, U64P, U64P]
_lib.kh_ipa_finish.argtypes = [C.c_void_p, U64P, U64P, U64P, U8P]
_lib.kh_ipa_free.argtypes = [C.c_void_p]
_lib.kh_ipa_free.restype = None
_lib.kh_srs_create_device.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
_lib.kh_srs_create_device_range.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
_lib.kh_srs_get_g.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, U64P]
_lib.kh_points_sum.argtypes = [C.c_int, U64P, U8P, C.c_size_t, U64P, U8P]
_lib.kh_points_add.argtypes = [C.c_int, U64P, U8P, U64P, U8P, C.c_size_t, U64P, U8P]
_lib.kh_msm_submit.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
_lib.kh_msm_wait.argtypes = [C.c_uint64, U64P, U8P]
_lib.kh_msm_submit_host.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_size_t, U64P, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
_lib.kh_msm_points_batch.argtypes = [C.c_int, U64P, U8P, U64P, C.c_size_t, C.c_size_t, C.c_int, U64P, U8P]
_lib.kh_ntt.argtypes = [C.c_int, U64P, C.c_uint, C.c_int, C.c_size_t]
_lib.kh_lde.argtypes = [C.c_int, U64P, C.c_uint, C.c_uint, U64P, C.c_size_t]
_lib.kh_ntt_dev.argtypes = [C.c_int, C.c_void_p, C.c_uint, C.c_int, C.c_size_t]
_lib.kh_lde_dev.argtypes = [C.c_int, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_size_t]
_lib.kh_coset_ntt_dev.argtypes = [C.c_int, C.c_void_p, C.c_uint, U64P, C.c_void_p, C.c_size_t]
_lib.kh_dev_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
_lib.kh_dev_free.argtypes = [C.c_void_p]
_lib.kh_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
_lib.kh_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
_lib.kh_dev_upload_2d.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
_lib.kh_dev_upload_2d_unordered.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
_lib.kh_last_timings.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
_lib.kh_debug_field_op.argtypes = [C.c_int, C.c_int, U64P, U64P, U64P, C.c_size_t]
_lib.kh_debug_point_op.argtypes = [C.c_int, C.c_int, U64P, U8P, U64P, U8P, U64P, U8P, C.c_size_t]


_lib.kh_srs_generate.argtypes = [C.c_int, C.c_size_t, C.c_size_t, U64P, C.c_int]
_lib.kh_srs_h.argtypes = [C.c_int, U64P]


_lib.kh_commit_non_hiding.argtypes = [C.c_void_p, U64P, C.c_size_t, C.c_size_t, U64P, U8P, C.POINTER(C.c_size_t)]
_lib.kh_commit_evaluations_non_hiding.argtypes = [C.c_void_p, C.c_uint, U64P, C.c_size_t, U64P, U8P, C.POINTER(C.c_size_t)]
_lib.kh_srs_set_blinding_base.argtypes = [C.c_void_p, U64P]
_lib.kh_srs_get_blinding_base.argtypes = [C.c_void_p, U64P]
_lib.kh_mask_custom.argtypes = [C.c_void_p, U64P, U8P, C.c_size_t, U64P, C.c_size_t, U64P, U8P]
_lib.kh_domain_generator.argtypes = [C.c_int, C.c_uint, U64P]
E_BLINDERS = -5


def _declare_remaining_argtypes():
    """ctypes passes an undeclared Python int as a 32-bit C int: every function whose argument list is not declared above gets it from the header's own
    prototype (include/kimchi_hip.h): int / unsigned / size_t / uint64_t by value, every pointer or array as void* (which takes ctypes pointers, byref(),
    arrays, addresses and None alike).  Round 6: kh_msm_submit_host, called with a bare Python int for its size_t n, asked hipMalloc for 6.6 EB."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "kimchi_hip.h")
    try:
        src = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    except OSError:
        return
    by_value = {"int": C.c_int, "unsigned": C.c_uint, "size_t": C.c_size_t, "uint64_t": C.c_uint64, "uint32_t": C.c_uint32, "double": C.c_double, "float": C.c_float}
    for m in re.finditer(r"\b(kh_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", src):
        name, args = m.group(1), " ".join(m.group(2).split())
        fn = getattr(_lib, name, None)
        if fn is None or fn.argtypes is not None or args in ("", "void"):
            continue
        types = []
        for a in args.split(","):
            a = a.strip()
            if "*" in a or "[" in a:
                types.append(C.c_void_p)
                continue
            base = [t for t in a.split()[:-1] if t != "const"]
            if len(base) != 1 or base[0] not in by_value:
                types = None
                break
            types.append(by_value[base[0]])
        if types is not None:
            fn.argtypes = types


_declare_remaining_argtypes()


class KhError(RuntimeError):
    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


def _check(rc: int):
    if rc != 0:
        raise KhError(f"kimchi_hip error {rc}: {_lib.kh_last_error().decode()}", rc)


def _p64(a):
    return a.ctypes.data_as(U64P)


def _p8(a):
    return None if a is None else a.ctypes.data_as(U8P)


def _c64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a if shape is None else a.reshape(shape)


def raw():
    return _lib


def device_count() -> int:
    return _lib.kh_device_count()


def init(device: int = -1, timers: bool = True):
    """kh_init; timers: switch the per-phase HIP events behind last_timings() on for the device's shared context (the library's default is off: they cost
    the stream a few microseconds each; this binding's users are tools and tests that read them)"""
    _check(_lib.kh_init(device))
    if timers:
        _check(_lib.kh_set_phase_timers(1))


def set_device(device: int):
    """This THREAD's current device from now on (kh_set_device); handles keep running on the device they were created on."""
    _check(_lib.kh_set_device(device))


def get_device() -> int:
    return _lib.kh_get_device()


def trim():
    _check(_lib.kh_trim())


def set_phase_timers(on: bool):
    """per-phase HIP events behind last_timings() on the calling thread's current context (kh_set_phase_timers)"""
    _check(_lib.kh_set_phase_timers(int(bool(on))))


def set_sort_staging(entries: int = 28672, max_passes: int = 2):
    """The wide path's staged scatter (kh_msm_set_sort_staging): entries per pass in LDS (0 = direct scatter) and the most passes a partition may take."""
    _check(_lib.kh_msm_set_sort_staging(entries, max_passes))


def glv_split(scalar_field: int, scalars):
    """kh_debug_glv_split: for canonical scalars (n, 4) u64 -> (|k1|, |k2|) as Python ints and their signs, k = k1 + k2 lambda (mod r)."""
    a = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = a.shape[0]
    out = np.zeros((n, 10), np.uint32)
    _lib.kh_debug_glv_split.argtypes = [C.c_int, U64P, C.c_size_t, C.POINTER(C.c_uint32)]
    _check(_lib.kh_debug_glv_split(scalar_field, _p64(a), n, out.ctypes.data_as(C.POINTER(C.c_uint32))))
    res = []
    for row in out:
        m1 = sum(int(row[t]) << (32 * t) for t in range(4)); m2 = sum(int(row[4 + t]) << (32 * t) for t in range(4))
        res.append((-m1 if row[8] else m1, -m2 if row[9] else m2))
    return res


def set_wide_min_n(n: int):
    """Bases of >= n points created from now on get the 20-bit-window table set, single MSMs of >= n scalars take it (kh_msm_set_wide_min_n; 0 = never)."""
    _check(_lib.kh_msm_set_wide_min_n(n))


class Srs:
    """Device-resident SRS bases (the g half of ipa::SRS<G>, poly-commitment/src/ipa.rs:53-75)."""

    def __init__(self, curve: int, g_xy=None, depth: int = 0, start: int = 0):
        """Upload the bases g_xy, or (g_xy is None) run SRS::create on the device for g[start, start+depth)."""
        self.curve = curve
        self._h = C.c_void_p()
        if g_xy is None:
            self.n = depth
            _check(_lib.kh_srs_create_device_range(curve, start, depth, C.byref(self._h)))
        else:
            g = _c64(g_xy, (-1, 8))
            self.n = g.shape[0]
            _check(_lib.kh_srs_create(curve, _p64(g), self.n, C.byref(self._h)))

    @classmethod
    def create(cls, curve: int, depth: int, start: int = 0):
        """SRS::create(depth) (ipa.rs:751-778) on the device; start > 0 gives the slice g[start, start+depth)."""
        return cls(curve, None, depth, start)

    def get_g(self, offset: int = 0, count: int = None):
        count = self.n - offset if count is None else count
        out = np.zeros((count, 8), dtype=np.uint64)
        _check(_lib.kh_srs_get_g(self._h, offset, count, _p64(out)))
        return out

    def debug_rebase_points(self, coef):
        """kh_debug_rebase_points: the folded basis sum_q coef[q] g[q N + i] (csrc/rebase.hip), affine; raises if an output was the identity."""
        c = _c64(coef, (-1, 4))
        N = self.n // c.shape[0]
        out = np.zeros((N, 8), dtype=np.uint64); fail = C.c_uint32(0)
        _check(_lib.kh_debug_rebase_points(self._h, _p64(c), c.shape[0], _p64(out), C.byref(fail)))
        if fail.value:
            raise KhError("rebase: an output is the point at infinity")
        return out

    def has_wide_tables(self) -> bool:
        return bool(_lib.kh_srs_has_wide_tables(self._h))

    def set_wide_tables(self, on: bool):
        """Build (True) or give back (False) the optional 20-bit-window table set of this handle (kh_srs_set_wide_tables)."""
        _check(_lib.kh_srs_set_wide_tables(self._h, int(on)))

    def close(self):
        if self._h:
            _lib.kh_srs_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lagrange(self, log2_domain: int, xy, inf=None, chunk: int = 0):
        xy = _c64(xy, (-1, 8))
        if inf is not None:
            inf = np.ascontiguousarray(inf, dtype=np.uint8)
        _check(_lib.kh_srs_set_lagrange(self._h, log2_domain, chunk, _p64(xy), _p8(inf), xy.shape[0]))

    def compute_lagrange(self, log2_domain: int):
        _check(_lib.kh_srs_compute_lagrange(self._h, log2_domain))

    def get_lagrange(self, log2_domain: int, chunk: int = 0):
        n = 1 << log2_domain
        xy = np.zeros((n, 8), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        _check(_lib.kh_srs_get_lagrange(self._h, log2_domain, chunk, _p64(xy), _p8(inf)))
        return xy, inf

    def lagrange_chunks(self, log2_domain: int) -> int:
        return _lib.kh_srs_lagrange_chunks(self._h, log2_domain)

    # ---- the SRS trait surface (poly-commitment/src/lib.rs:61-241) over the kernels
    def max_poly_size(self) -> int:
        return self.n

    def blinding_commitment(self):
        out = np.zeros(8, dtype=np.uint64)
        _check(_lib.kh_srs_get_blinding_base(self._h, _p64(out)))
        return out

    def set_blinding_base(self, h_xy):
        h = _c64(h_xy, (8,))
        _check(_lib.kh_srs_set_blinding_base(self._h, _p64(h)))

    def commit_non_hiding(self, coeffs, num_chunks: int):
        """SRS::commit_non_hiding (ipa.rs:638-683) -> (chunks x 8 limbs, inf flags)."""
        c = _c64(coeffs, (-1, 4))
        cap = max(num_chunks, -(-c.shape[0] // self.n), 1)
        out = np.zeros((cap, 8), dtype=np.uint64)
        inf = np.zeros(cap, dtype=np.uint8)
        cnt = C.c_size_t(0)
        _check(_lib.kh_commit_non_hiding(self._h, _p64(c), c.shape[0], num_chunks, _p64(out), _p8(inf), C.byref(cnt)))
        return out[:cnt.value], inf[:cnt.value]

    def commit_evaluations_non_hiding(self, log2_domain: int, evals):
        """SRS::commit_evaluations_non_hiding (ipa.rs:706-728)."""
        e = _c64(evals, (-1, 4))
        cap = max(1, self.lagrange_chunks(log2_domain))
        out = np.zeros((cap, 8), dtype=np.uint64)
        inf = np.zeros(cap, dtype=np.uint8)
        cnt = C.c_size_t(0)
        _check(_lib.kh_commit_evaluations_non_hiding(self._h, log2_domain, _p64(e), e.shape[0], _p64(out), _p8(inf), C.byref(cnt)))
        return out[:cnt.value], inf[:cnt.value]

    def mask_custom(self, com_xy, com_inf, blinders):
        """SRS::mask_custom (ipa.rs:605-622); raises KhError(code=E_BLINDERS) on BlindersDontMatch."""
        com = _c64(com_xy, (-1, 8)); bl = _c64(blinders, (-1, 4))
        ci = np.ascontiguousarray(com_inf, dtype=np.uint8)
        out = np.zeros((com.shape[0], 8), dtype=np.uint64)
        inf = np.zeros(com.shape[0], dtype=np.uint8)
        _check(_lib.kh_mask_custom(self._h, _p64(com), _p8(ci), com.shape[0], _p64(bl), bl.shape[0], _p64(out), _p8(inf)))
        return out, inf

    def commit_custom(self, coeffs, num_chunks: int, blinders):
        """SRS::commit_custom (ipa.rs:686-693) = mask_custom(commit_non_hiding(..), blinders)."""
        com, inf = self.commit_non_hiding(coeffs, num_chunks)
        return self.mask_custom(com, inf, blinders)

    def msm(self, scalars, basis: int = BASIS_G, chunk: int = 0, offset: int = 0, mont: bool = True):
        sc = _c64(scalars, (-1, 4))
        out = np.zeros(8, dtype=np.uint64)
        inf = np.zeros(1, dtype=np.uint8)
        _check(_lib.kh_msm(self._h, basis, chunk, offset, _p64(sc), sc.shape[0], int(mont), _p64(out), _p8(inf)))
        return out, bool(inf[0])

    def msm_batch(self, scalars, basis: int = BASIS_G, chunk: int = 0, offset: int = 0, mont: bool = True):
        sc = _c64(scalars)
        assert sc.ndim == 3 and sc.shape[2] == 4
        k, n = sc.shape[0], sc.shape[1]
        out = np.zeros((k, 8), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.uint8)
        _check(_lib.kh_msm_batch(self._h, basis, chunk, offset, _p64(sc), n, k, int(mont), _p64(out), _p8(inf)))
        return out, inf

    def msm_submit(self, scalars_dev: int, n: int, k: int, basis: int = BASIS_G, chunk: int = 0, offset: int = 0, mont: bool = True):
        """Enqueue k MSMs (device-resident scalars) on a free pipeline slot; returns a ticket for msm_wait."""
        t = C.c_uint64(0)
        _check(_lib.kh_msm_submit(self._h, basis, chunk, offset, C.c_void_p(scalars_dev), n, k, int(mont), C.byref(t)))
        return (t.value, k)

    def msm_submit_host(self, scalars, k: int = 1, basis: int = BASIS_G, chunk: int = 0, offset: int = 0, mont: bool = True):
        """kh_msm_submit_host: the same pipeline from HOST scalars (k x n x 4 limbs); the buffer is the caller's again when this returns."""
        sc = _c64(scalars, (k, -1, 4))
        t = C.c_uint64(0)
        _check(_lib.kh_msm_submit_host(self._h, basis, chunk, offset, _p64(sc), sc.shape[1], k, int(mont), C.byref(t)))
        return (t.value, k)

    @staticmethod
    def msm_wait(ticket):
        t, k = ticket
        out = np.zeros((k, 8), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.uint8)
        _check(_lib.kh_msm_wait(t, _p64(out), _p8(inf)))
        return out, inf

    def msm_batch_dev(self, scalars_dev: int, n: int, k: int, basis: int = BASIS_G, chunk: int = 0, offset: int = 0, mont: bool = True):
        out = np.zeros((k, 8), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.uint8)
        _check(_lib.kh_msm_batch_dev(self._h, basis, chunk, offset, C.c_void_p(scalars_dev), n, k, int(mont), _p64(out), _p8(inf)))
        return out, inf


def msm_sharded(shards, scalars, mont: bool = True):
    """kh_msm_sharded: ONE MSM over a basis sharded by point range over the Srs handles `shards` (each on its own device): the slices
    of the host scalars go to their shards' devices concurrently, the partial sums are folded on the host."""
    sc = _c64(scalars, (-1, 4))
    hs = (C.c_void_p * len(shards))(*[s._h for s in shards])
    out = np.zeros(8, dtype=np.uint64); inf = np.zeros(1, dtype=np.uint8)
    _check(_lib.kh_msm_sharded(hs, len(shards), _p64(sc), sc.shape[0], int(mont), _p64(out), _p8(inf)))
    return out, bool(inf[0])


def msm_sharded_dev(shards, bufs, counts, mont: bool = True):
    """kh_msm_sharded_dev: bufs[r] = DevBuf on shard r's device holding its counts[r] scalars."""
    hs = (C.c_void_p * len(shards))(*[s._h for s in shards])
    ps = (C.c_void_p * len(shards))(*[C.c_void_p(b.ptr) for b in bufs])
    cs = (C.c_size_t * len(shards))(*counts)
    out = np.zeros(8, dtype=np.uint64); inf = np.zeros(1, dtype=np.uint8)
    _check(_lib.kh_msm_sharded_dev(hs, len(shards), ps, cs, int(mont), _p64(out), _p8(inf)))
    return out, bool(inf[0])


def srs_generate(curve: int, start: int, count: int, threads: int = 0):
    """SRS::create(depth).g[start:start+count] (host-side hash-to-curve, ipa.rs:751-778)."""
    out = np.empty((count, 8), dtype=np.uint64)
    _check(_lib.kh_srs_generate(curve, start, count, _p64(out), threads))
    return out


def srs_h(curve: int):
    out = np.empty(8, dtype=np.uint64)
    _check(_lib.kh_srs_h(curve, _p64(out)))
    return out


def msm_points(curve: int, xy, scalars, inf=None, mont: bool = True):
    xy = _c64(xy, (-1, 8))
    sc = _c64(scalars, (-1, 4))
    n = min(xy.shape[0], sc.shape[0])
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.zeros(8, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    _check(_lib.kh_msm_points(curve, _p64(xy), _p8(inf), _p64(sc), n, int(mont), _p64(out), _p8(oinf)))
    return out, bool(oinf[0])


def msm_points_batch(curve: int, xy, scalars, inf=None, mont: bool = True):
    """k independent MSMs: xy (k, n, 8), scalars (k, n, 4), inf (k, n) or None."""
    xy = _c64(xy); sc = _c64(scalars)
    assert xy.ndim == 3 and sc.ndim == 3 and xy.shape[:2] == sc.shape[:2]
    k, n = xy.shape[0], xy.shape[1]
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.zeros((k, 8), dtype=np.uint64)
    oinf = np.zeros(k, dtype=np.uint8)
    _check(_lib.kh_msm_points_batch(curve, _p64(xy), _p8(inf), _p64(sc), n, k, int(mont), _p64(out), _p8(oinf)))
    return out, oinf


def counter(name: str) -> int:
    """Process-wide event counter of the library (kh_counter): spread_retry, fused_retry, graph_replay, graph_capture, rebase_launch, rebase_switch, rebase_abandon, rebased_rounds."""
    return int(_lib.kh_counter(name.encode()))


def set_ntt_max_logr(v: int):
    """Largest sub-transform of an NTT pass = 2^v points (kh_ntt_set_max_logr: 4..10, 0 = default); same results under every setting."""
    _check(_lib.kh_ntt_set_max_logr(v))


def ntt(field: int, data, log2_n: int, inverse: bool = False, in_place: bool = False):
    """in_place: transform `data` itself (a C-contiguous uint64 array) -- what ark-poly's ifft_in_place does to the caller's Vec"""
    if in_place:
        assert data.dtype == np.uint64 and data.flags["C_CONTIGUOUS"]
        d = data.reshape(-1, 1 << log2_n, 4)
    else:
        d = np.array(data, dtype=np.uint64, copy=True).reshape(-1, 1 << log2_n, 4)
    _check(_lib.kh_ntt(field, _p64(d), log2_n, int(inverse), d.shape[0]))
    return d


def lde(field: int, coeffs, log2_n: int, log2_blowup: int, out=None):
    c = _c64(coeffs, (-1, 1 << log2_n, 4))
    if out is None:
        out = np.zeros((c.shape[0], 1 << (log2_n + log2_blowup), 4), dtype=np.uint64)
    assert out.dtype == np.uint64 and out.flags["C_CONTIGUOUS"] and out.size == c.shape[0] * (4 << (log2_n + log2_blowup))
    _check(_lib.kh_lde(field, _p64(c), log2_n, log2_blowup, _p64(out), c.shape[0]))
    return out


class DevBuf:
    """A raw HBM allocation (kh_dev_alloc) for the device-resident entry points."""

    def __init__(self, nbytes: int):
        self.nbytes = nbytes
        p = C.c_void_p()
        _check(_lib.kh_dev_alloc(C.byref(p), nbytes))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(_lib.kh_dev_upload(C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return self

    def download(self, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(_lib.kh_dev_download(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            _check(_lib.kh_dev_free(C.c_void_p(self.ptr)))
            self.ptr = None

    def view(self, offset_bytes: int):
        """A non-owning alias of this allocation starting `offset_bytes` in (anything with a .ptr is accepted as a column)."""
        return DevView(self.ptr + offset_bytes)

    def upload_at(self, offset_bytes: int, arr):
        arr = np.ascontiguousarray(arr)
        assert offset_bytes + arr.nbytes <= self.nbytes
        _check(_lib.kh_dev_upload(C.c_void_p(self.ptr + offset_bytes), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def upload_2d(self, offset_bytes: int, dst_pitch: int, arr2d):
        """arr2d: (rows, ...) C-contiguous; row r goes to offset_bytes + r * dst_pitch."""
        a = np.ascontiguousarray(arr2d)
        rows = a.shape[0]; width = a.nbytes // rows
        assert offset_bytes + (rows - 1) * dst_pitch + width <= self.nbytes
        _check(_lib.kh_dev_upload_2d(C.c_void_p(self.ptr + offset_bytes), dst_pitch, a.ctypes.data_as(C.c_void_p), width, width, rows))

    def download_at(self, offset_bytes: int, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        _check(_lib.kh_dev_download(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset_bytes), out.nbytes))
        return out

    def zero(self):
        _check(_lib.kh_dev_memset_zero(C.c_void_p(self.ptr), self.nbytes))
        return self

    def fill_elements(self, offset_elems: int, value, count: int):
        """count 32-byte records from element offset_elems on set to `value` (4 limbs), queued on the main stream (kh_dev_fill_elements)"""
        assert (offset_elems + count) * 32 <= self.nbytes
        _check(_lib.kh_dev_fill_elements(C.c_void_p(self.ptr + 32 * offset_elems), _p64(_c64(value, (4,))), count))
        return self


class DevView:
    def __init__(self, ptr: int):
        self.ptr = ptr

    def view(self, offset_bytes: int):
        return DevView(self.ptr + offset_bytes)


def dev_copy(dst_ptr: int, src_ptr: int, nbytes: int):
    _check(_lib.kh_dev_copy(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes))


def dev_memset_zero(dst_ptr: int, nbytes: int):
    _check(_lib.kh_dev_memset_zero(C.c_void_p(dst_ptr), C.c_size_t(nbytes)))


def group_map_to_group(curve: int, t):
    out = np.zeros(8, dtype=np.uint64)
    _check(_lib.kh_group_map_to_group(curve, _p64(_c64(t, (4,))), _p64(out)))
    return out


def ntt_dev(field: int, buf: DevBuf, log2_n: int, inverse: bool, batch: int):
    _check(_lib.kh_ntt_dev(field, C.c_void_p(buf.ptr), log2_n, int(inverse), batch))


def coset_ntt_dev(field: int, src: DevBuf, log2_n: int, shift, dst: DevBuf, batch: int):
    _check(_lib.kh_coset_ntt_dev(field, C.c_void_p(src.ptr), log2_n, _p64(_c64(shift, (4,))), C.c_void_p(dst.ptr), batch))


def lde_dev(field: int, src: DevBuf, log2_n: int, log2_blowup: int, dst: DevBuf, batch: int):
    _check(_lib.kh_lde_dev(field, C.c_void_p(src.ptr), log2_n, log2_blowup, C.c_void_p(dst.ptr), batch))


def points_sum(curve: int, xy, inf=None):
    """Host-side sum of a few affine points (fold of per-GPU partial MSM results)."""
    xy = _c64(xy, (-1, 8))
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.zeros(8, dtype=np.uint64); oinf = np.zeros(1, dtype=np.uint8)
    _check(_lib.kh_points_sum(curve, _p64(xy), _p8(inf), xy.shape[0], _p64(out), _p8(oinf)))
    return out, bool(oinf[0])


def points_add(curve: int, a_xy, a_inf, b_xy, b_inf):
    """Pairwise host sums a_j + b_j of affine points (the second half of a masking whose blinding points were computed ahead)."""
    a_xy = _c64(a_xy, (-1, 8)); b_xy = _c64(b_xy, (-1, 8))
    n = a_xy.shape[0]
    a_inf = None if a_inf is None else np.ascontiguousarray(a_inf, dtype=np.uint8)
    b_inf = None if b_inf is None else np.ascontiguousarray(b_inf, dtype=np.uint8)
    out = np.zeros((n, 8), dtype=np.uint64); oinf = np.zeros(n, dtype=np.uint8)
    _check(_lib.kh_points_add(curve, _p64(a_xy), _p8(a_inf), _p64(b_xy), _p8(b_inf), n, _p64(out), _p8(oinf)))
    return out, oinf


def ipa_fold_scalars(field: int, lo, hi, u):
    lo = _c64(lo, (-1, 4)); hi = _c64(hi, (-1, 4)); u = _c64(u, (4,))
    out = np.zeros_like(lo)
    _check(_lib.kh_ipa_fold_scalars(field, _p64(lo), _p64(hi), _p64(u), lo.shape[0], _p64(out)))
    return out


def inner_product(field: int, a, b):
    a = _c64(a, (-1, 4)); b = _c64(b, (-1, 4))
    out = np.zeros(4, dtype=np.uint64)
    _check(_lib.kh_inner_product(field, _p64(a), _p64(b), a.shape[0], _p64(out)))
    return out


def ipa_fold_points(curve: int, g_lo, g_hi, u):
    g_lo = _c64(g_lo, (-1, 8)); g_hi = _c64(g_hi, (-1, 8)); u = _c64(u, (4,))
    out = np.zeros_like(g_lo); inf = np.zeros(g_lo.shape[0], dtype=np.uint8)
    _check(_lib.kh_ipa_fold_points(curve, _p64(g_lo), _p64(g_hi), _p64(u), g_lo.shape[0], _p64(out), _p8(inf)))
    return out, inf


def ipa_fold_points_endo(curve: int, g_lo, g_hi, chal: int):
    """combine_one_endo: chal is the 128-bit prechallenge as a Python int."""
    g_lo = _c64(g_lo, (-1, 8)); g_hi = _c64(g_hi, (-1, 8))
    c = np.array([chal & (2**64 - 1), (chal >> 64) & (2**64 - 1)], dtype=np.uint64)
    out = np.zeros_like(g_lo); inf = np.zeros(g_lo.shape[0], dtype=np.uint8)
    _check(_lib.kh_ipa_fold_points_endo(curve, _p64(g_lo), _p64(g_hi), _p64(c), g_lo.shape[0], _p64(out), _p8(inf)))
    return out, inf


SCAN_ADD, SCAN_MUL = 0, 1


def field_scan_dev(field: int, op: int, buf, n: int, reverse: bool = False, offset: int = 0):
    """In-place inclusive scan of n elements starting `offset` elements into the DevBuf."""
    _check(_lib.kh_field_scan_dev(field, op, int(reverse), C.c_void_p(buf.ptr + 32 * offset), n))


def batch_inversion_dev(field: int, buf, n: int, offset: int = 0):
    _check(_lib.kh_batch_inversion_dev(field, C.c_void_p(buf.ptr + 32 * offset), n))


def divide_by_linear_dev(field: int, f, length: int, a, q):
    rem = np.zeros(4, dtype=np.uint64)
    _check(_lib.kh_divide_by_linear_dev(field, C.c_void_p(f.ptr), length, _p64(_c64(a, (4,))), C.c_void_p(q.ptr if q is not None else 0), _p64(rem)))
    return rem


def divide_by_linear_async_dev(field: int, f, length: int, a, q, rem_dev):
    """kh_divide_by_linear_async_dev: quotient into q, remainder into the DevBuf rem_dev (4 limbs); nothing waits."""
    _check(_lib.kh_divide_by_linear_async_dev(field, C.c_void_p(f.ptr), length, _p64(_c64(a, (4,))), C.c_void_p(q.ptr if q is not None else 0),
                                              C.c_void_p(rem_dev.ptr if rem_dev is not None else 0)))


def check_equal_dev(v, n: int, expect, flags, bit: int, offset: int = 0):
    """kh_check_equal_dev: flags (a DevBuf holding a uint32) |= 1 << bit when any of the n elements at v (+ offset elements) differs from expect (None: zero)."""
    e = None if expect is None else _c64(expect, (4,))
    _check(_lib.kh_check_equal_dev(C.c_void_p(v.ptr + 32 * offset), n, _p64(e) if e is not None else None, C.c_void_p(flags.ptr), bit))


TOK_CONST, TOK_CELL, TOK_DUP, TOK_POW, TOK_ADD, TOK_MUL, TOK_SUB, TOK_STORE, TOK_LOAD = range(9)


def expr_evaluations_dev(field: int, tokens, cols, col_len, constants, rows: int, out, stride: int = 1, next_shift: int = 8, accumulate: bool = False,
                         out_offset: int = 0):
    """tokens: list of (opcode, arg); cols: list of DevBuf; constants: (k, 4) Montgomery limbs; out: DevBuf receiving `rows`
    elements starting `out_offset` elements in."""
    tk = np.ascontiguousarray(np.array(tokens, dtype=np.uint32).reshape(-1, 2))
    m = len(cols)
    ptrs = (C.c_void_p * max(m, 1))(*[C.c_void_p(c.ptr) for c in cols])
    lens = (C.c_size_t * max(m, 1))(*col_len)
    cs = _c64(constants, (-1, 4))
    _check(_lib.kh_expr_evaluations_dev(field, tk.ctypes.data_as(C.POINTER(C.c_uint32)), tk.shape[0], ptrs, lens, m, _p64(cs), cs.shape[0],
                                        rows, stride, next_shift, int(accumulate), C.c_void_p(out.ptr + 32 * out_offset)))


_GATE_IDS = {}


def gate_ids():
    """{name: id} of the compiled kernels (kh_gate_evaluations_dev): the gate library's GateType names, "Generic", "Permutation"."""
    if not _GATE_IDS:
        _lib.kh_gate_name.restype = C.c_char_p
        _GATE_IDS.update({_lib.kh_gate_name(g).decode(): g for g in range(_lib.kh_gate_count())})
    return _GATE_IDS


def gate_num_constants(gate: int) -> int:
    return _lib.kh_gate_num_constants(C.c_int(gate))


def gate_constants(field: int, gate: int, alpha=None, endo=None, params=None):
    """kh_gate_constants: the (k, 4) constants table of a compiled gate kernel for one proof (host only)."""
    k = gate_num_constants(gate)
    out = np.zeros((k, 4), dtype=np.uint64)
    a = _c64(alpha, (4,)) if alpha is not None else None
    e = _c64(endo, (4,)) if endo is not None else None
    ps = _c64(params, (-1, 4)) if params is not None else None
    _check(_lib.kh_gate_constants(C.c_int(field), C.c_int(gate), _p64(a) if a is not None else None, _p64(e) if e is not None else None,
                                  _p64(ps) if ps is not None else None, C.c_size_t(ps.shape[0] if ps is not None else 0), _p64(out)))
    return out


def gate_evaluations_dev(field: int, gate: int, cols, col_len: int, constants, rows: int, out, stride: int = 1, next_shift: int = 8, accumulate: bool = False,
                         out_offset: int = 0):
    """kh_gate_evaluations_dev: cols = 31 DevBuf (witness 0..14, coefficients 15..29, the gate's selector); constants (k, 4) Montgomery limbs."""
    assert len(cols) == 31
    ptrs = (C.c_void_p * 31)(*[C.c_void_p(c.ptr) for c in cols])
    cs = _c64(constants, (-1, 4))
    _check(_lib.kh_gate_evaluations_dev(C.c_int(field), C.c_int(gate), ptrs, C.c_size_t(col_len), _p64(cs), C.c_size_t(cs.shape[0]), C.c_size_t(rows),
                                        C.c_uint(stride), C.c_uint(next_shift), C.c_int(int(accumulate)), C.c_void_p(out.ptr + 32 * out_offset)))


def lookup_sorted(table, lookup_rows: int, values, max_per_row: int):
    """kh_lookup_sorted: table (>= lookup_rows, 4) limbs, values (max_per_row, stride, 4) limbs -> (max_per_row + 1, lookup_rows + 1, 4).
    Raises ValueError(row) for a value that is not in the table."""
    t = _c64(table, (-1, 4)); v = _c64(values, (max_per_row, -1, 4))
    out = np.zeros((max_per_row + 1, lookup_rows + 1, 4), dtype=np.uint64)
    bad = C.c_size_t(0)
    rc = _lib.kh_lookup_sorted(_p64(t), C.c_size_t(lookup_rows), _p64(v), C.c_size_t(v.shape[1]), C.c_size_t(max_per_row), _p64(out), C.byref(bad))
    if rc != 0 and bad.value != C.c_size_t(-1).value:
        raise ValueError(bad.value)
    _check(rc)
    return out


class Comm:
    """kh_comm_*: the in-library RCCL communicator of the one-process-per-GPU deployment (no torch).  Comm.unique_id() on one rank, the bytes to
    the others, Comm(world, rank, id) on every rank."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _check(_lib.kh_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, world: int, rank: int, uid: bytes):
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128)(*uid)
        _check(_lib.kh_comm_init(C.c_int(world), C.c_int(rank), buf, C.byref(self._h)))
        self.world, self.rank = world, rank

    def allgather_points(self, xy, inf):
        xy = _c64(xy, (-1, 8)); i8 = np.ascontiguousarray(inf, dtype=np.uint8).reshape(-1)
        k = xy.shape[0]
        out = np.zeros((self.world * k, 8), dtype=np.uint64); oinf = np.zeros(self.world * k, dtype=np.uint8)
        _check(_lib.kh_comm_allgather_points(self._h, _p64(xy), _p8(i8), C.c_size_t(k), _p64(out), _p8(oinf)))
        return out, oinf

    def msm_allreduce(self, shard, scalars, mont: bool = True):
        sc = _c64(scalars, (-1, 4))
        out = np.zeros(8, dtype=np.uint64); inf = np.zeros(1, dtype=np.uint8)
        _check(_lib.kh_msm_allreduce(self._h, shard._h, _p64(sc), C.c_size_t(sc.shape[0]), C.c_int(int(mont)), _p64(out), _p8(inf)))
        return out, bool(inf[0])

    def free(self):
        if self._h:
            _lib.kh_comm_free(self._h); self._h = C.c_void_p()


PROVE_CHECK, PROVE_ALL_GATES, PROVE_SHARED_CONTEXT, PROVE_EAGER_CHECK = 1, 2, 4, 8
PROOF_SECTIONS = {"w_comm": 0, "z_comm": 1, "t_comm": 2, "public_comm": 3, "evals": 4, "public_evals": 5, "ft_eval1": 6, "lr": 7, "delta": 8, "z1_z2": 9, "sg": 10,
                  "challenges": 11, "lookup_sorted_comm": 12, "lookup_aggreg_comm": 13, "lookup_runtime_comm": 14}
LOOKUP_PATTERN_IDS = {"Xor": 0, "Lookup": 1, "RangeCheck": 2, "ForeignFieldMul": 3}
PROOF_PHASES = ("witness_upload", "witness_commit", "z", "quotient", "evaluations", "opening")


class NativeProverIndex:
    """kh_prover_index_new: the C++ prover's view of device-resident index columns (they must outlive the handle)."""

    def __init__(self, srs, log2_n: int, zk_rows: int, public: int, d1, dc, d8, optional_gate_ids, live_mask: int, shifts, digest):
        self._h = C.c_void_p()
        opt = (C.c_int * max(len(optional_gate_ids), 1))(*optional_gate_ids)
        sh = _c64(shifts, (7, 4)); dg = _c64(digest, (4,))
        _check(_lib.kh_prover_index_new(srs._h, C.c_uint(log2_n), C.c_uint(zk_rows), C.c_uint(public), C.c_void_p(d1.ptr), C.c_void_p(dc.ptr), C.c_void_p(d8.ptr),
                                        opt, C.c_size_t(len(optional_gate_ids)), C.c_uint(live_mask), _p64(sh), _p64(dg), C.byref(self._h)))
        self._keep = (srs, d1, dc, d8)

    def attach_lookup(self, patterns, sel_d1, sel_c, sel_d8, table_cols, table_ids, atoms8):
        """kh_prover_index_attach_lookup: patterns = names in the reference's order; per pattern three DevBufs; table columns / ids / atoms: DevBufs."""
        m = len(patterns)
        arr = lambda bufs: (C.c_void_p * max(len(bufs), 1))(*[C.c_void_p(b.ptr) for b in bufs])
        ids = (C.c_int * m)(*[LOOKUP_PATTERN_IDS[q] for q in patterns])
        _check(_lib.kh_prover_index_attach_lookup(self._h, ids, C.c_size_t(m), arr(sel_d1), arr(sel_c), arr(sel_d8), arr(table_cols), C.c_size_t(len(table_cols)),
                                                  C.c_void_p(table_ids.ptr) if table_ids is not None else None, arr(atoms8)))
        self._keep += (sel_d1, sel_c, sel_d8, table_cols, table_ids, atoms8)

    def attach_runtime_tables(self, sel_d1, sel_c, sel_d8, offset: int, length: int):
        _check(_lib.kh_prover_index_attach_runtime_tables(self._h, C.c_void_p(sel_d1.ptr), C.c_void_p(sel_c.ptr), C.c_void_p(sel_d8.ptr), C.c_size_t(offset), C.c_size_t(length)))
        self._keep += (sel_d1, sel_c, sel_d8)

    def randomness_count(self, witness_on_host: bool) -> int:
        _lib.kh_prove_randomness_count.restype = C.c_size_t
        return _lib.kh_prove_randomness_count(self._h, C.c_int(int(witness_on_host)))

    def prove(self, witness=None, witness_dev=None, randomness=None, flags: int = PROVE_CHECK, prev=(), runtime=None):
        """kh_prove_full (runtime: (k, 4) limbs, the runtime tables' second column).  witness: (15, rows, 4) limbs on the host, or witness_dev: DevBuf with the padded columns.  randomness:
        (k, 4) limbs in the reference's draw order, or None (the library draws from the OS).  prev: [(chals (k, 4) limbs, (xy (chunks, 8), inf
        (chunks,)))].  Returns ({section: limbs[, flags]}, {phase: seconds})."""
        pr = C.c_void_p()
        w = _c64(witness, (15, -1, 4)) if witness is not None else None
        rnd = _c64(randomness, (-1, 4)) if randomness is not None else None
        m = len(prev)
        chals = _c64(np.concatenate([np.asarray(c, dtype=np.uint64).reshape(-1, 4) for c, _ in prev]), (-1, 4)) if m else None
        rounds = (C.c_uint * max(m, 1))(*[np.asarray(c).reshape(-1, 4).shape[0] for c, _ in prev])
        cxy = np.ascontiguousarray(np.concatenate([np.asarray(cm[0], dtype=np.uint64).reshape(-1, 8) for _, cm in prev])) if m else None
        cinf = np.ascontiguousarray(np.concatenate([np.asarray(cm[1], dtype=np.uint8).reshape(-1) for _, cm in prev])) if m else None
        cch = (C.c_size_t * max(m, 1))(*[np.asarray(cm[1]).reshape(-1).shape[0] for _, cm in prev])
        rtv = _c64(runtime, (-1, 4)) if runtime is not None else None
        _check(_lib.kh_prove_full(self._h, _p64(w) if w is not None else None, C.c_size_t(w.shape[1] if w is not None else 0),
                                       C.c_void_p(witness_dev.ptr) if witness_dev is not None else None, _p64(rnd) if rnd is not None else None,
                                       C.c_size_t(rnd.shape[0] if rnd is not None else 0), C.c_uint(flags), _p64(chals) if m else None, rounds,
                                       _p64(cxy) if m else None, _p8(cinf) if m else None, cch, C.c_size_t(m), _p64(rtv) if rtv is not None else None,
                                       C.c_size_t(rtv.shape[0] if rtv is not None else 0), C.byref(pr)))
        try:
            out = {}
            for name, sid in PROOF_SECTIONS.items():
                lp = C.POINTER(C.c_uint64)(); fp = C.POINTER(C.c_uint8)(); cnt = C.c_size_t(0)
                _check(_lib.kh_proof_section(pr, C.c_int(sid), C.byref(lp), C.byref(fp), C.byref(cnt)))
                k = cnt.value
                if fp:                                    # points
                    out[name] = (np.ctypeslib.as_array(lp, shape=(k, 8)).copy() if k else np.zeros((0, 8), np.uint64),
                                 np.ctypeslib.as_array(fp, shape=(k,)).copy() if k else np.zeros(0, np.uint8))
                else:
                    out[name] = np.ctypeslib.as_array(lp, shape=(k, 4)).copy() if k else np.zeros((0, 4), np.uint64)
            ph = (C.c_double * 6)()
            _lib.kh_proof_phase_seconds(pr, ph, C.c_size_t(6))
            return out, dict(zip(PROOF_PHASES, list(ph)))
        finally:
            _lib.kh_proof_free(pr)

    def free(self):
        if self._h:
            _lib.kh_prover_index_free(self._h); self._h = C.c_void_p()


def polycomm_multi_scalar_mul(curve: int, comms, scalars):
    """comms: list of (chunks (k_i, 8) limbs, inf (k_i,) flags); scalars (m, 4).  Returns (chunks, inf)."""
    m = len(comms)
    counts = [np.asarray(c[0]).reshape(-1, 8).shape[0] for c in comms]
    xy = np.concatenate([np.asarray(c[0], dtype=np.uint64).reshape(-1, 8) for c in comms]) if m else np.zeros((0, 8), np.uint64)
    inf = np.concatenate([np.asarray(c[1], dtype=np.uint8).reshape(-1) for c in comms]) if m else np.zeros(0, np.uint8)
    xy = np.ascontiguousarray(xy); inf = np.ascontiguousarray(inf)
    sc = _c64(scalars, (-1, 4))
    width = max(counts + [1])
    out = np.zeros((width, 8), dtype=np.uint64); oinf = np.zeros(width, dtype=np.uint8)
    cnt = C.c_size_t(0)
    nc = (C.c_size_t * max(m, 1))(*counts)
    _check(_lib.kh_polycomm_multi_scalar_mul(curve, _p64(xy), _p8(inf), nc, m, _p64(sc), _p64(out), _p8(oinf), C.byref(cnt)))
    return out[:cnt.value], oinf[:cnt.value]


def combine_polys_dev(field: int, polys, lens, num_chunks, polyscale, srs_length: int, out):
    """polys: list of DevBuf; out: DevBuf of srs_length elements.  Returns the length of the combined polynomial."""
    m = len(polys)
    ptrs = (C.c_void_p * max(m, 1))(*[C.c_void_p(p.ptr) for p in polys])
    ls = (C.c_size_t * max(m, 1))(*lens); cs = (C.c_size_t * max(m, 1))(*num_chunks)
    ol = C.c_size_t(0)
    _check(_lib.kh_combine_polys_dev(field, ptrs, ls, cs, m, _p64(_c64(polyscale, (4,))), srs_length, C.c_void_p(out.ptr), C.byref(ol)))
    return ol.value


def poly_lincomb_dev(field: int, polys, lens, scalars, out, out_len: int):
    m = len(polys)
    ptrs = (C.c_void_p * max(m, 1))(*[C.c_void_p(p.ptr) for p in polys])
    ls = (C.c_size_t * max(m, 1))(*lens)
    _check(_lib.kh_poly_lincomb_dev(field, ptrs, ls, _p64(_c64(scalars, (-1, 4))), m, C.c_void_p(out.ptr), out_len))


def b_init_dev(field: int, elm, evalscale, padded_len: int, out):
    e = _c64(elm, (-1, 4))
    _check(_lib.kh_b_init_dev(field, _p64(e), e.shape[0], _p64(_c64(evalscale, (4,))), padded_len, C.c_void_p(out.ptr)))


def evaluate_chunks_dev(field: int, coeffs, length: int, chunk_size: int, num_chunks: int, points):
    pts = _c64(points, (-1, 4))
    out = np.zeros((pts.shape[0], num_chunks, 4), dtype=np.uint64)
    _check(_lib.kh_evaluate_chunks_dev(field, C.c_void_p(coeffs.ptr), length, chunk_size, num_chunks, _p64(pts), pts.shape[0], _p64(out)))
    return out


def evaluate_chunks_batch_dev(field: int, polys, lens, num_chunks, chunk_size: int, points):
    """Returns a list of (npts, num_chunks[j], 4) arrays, one per polynomial."""
    pts = _c64(points, (-1, 4)); m = len(polys)
    ptrs = (C.c_void_p * max(m, 1))(*[C.c_void_p(p.ptr) for p in polys])
    ls = (C.c_size_t * max(m, 1))(*lens); cs = (C.c_size_t * max(m, 1))(*num_chunks)
    out = np.zeros((pts.shape[0] * sum(num_chunks), 4), dtype=np.uint64)
    _check(_lib.kh_evaluate_chunks_batch_dev(field, ptrs, ls, cs, m, chunk_size, _p64(pts), pts.shape[0], _p64(out)))
    res, pos = [], 0
    for c in num_chunks:
        res.append(out[pos:pos + pts.shape[0] * c].reshape(pts.shape[0], c, 4)); pos += pts.shape[0] * c
    return res


def divide_by_vanishing_poly_dev(field: int, f, length: int, log2_n: int, q, r):
    _check(_lib.kh_divide_by_vanishing_poly_dev(field, C.c_void_p(f.ptr), length, log2_n, C.c_void_p(q.ptr if q is not None else 0), C.c_void_p(r.ptr)))


def b_poly_coefficients(field: int, chals, rounds: int):
    """chals: (k * rounds, 4) Montgomery limbs -> (k, 2^rounds, 4)."""
    ch = _c64(chals, (-1, 4))
    k = ch.shape[0] // rounds if rounds else 1
    out = np.zeros((k, 1 << rounds, 4), dtype=np.uint64)
    _check(_lib.kh_b_poly_coefficients(field, _p64(ch), rounds, k, _p64(out)))
    return out


def batch_dlog_accumulator_generate(srs, num_comms: int, chals):
    ch = _c64(chals, (-1, 4))
    out = np.zeros((num_comms, 8), dtype=np.uint64); inf = np.zeros(num_comms, dtype=np.uint8)
    _check(_lib.kh_batch_dlog_accumulator_generate(srs._h, num_comms, _p64(ch), ch.shape[0], _p64(out), _p8(inf)))
    return out, inf


def batch_dlog_accumulator_check(srs, comms, chals, r, inf=None) -> bool:
    cm = _c64(comms, (-1, 8)); ch = _c64(chals, (-1, 4)); r = _c64(r, (4,))
    if inf is None:
        inf = np.zeros(cm.shape[0], dtype=np.uint8)
    ok = C.c_int(0)
    _check(_lib.kh_batch_dlog_accumulator_check(srs._h, _p64(cm), _p8(inf), cm.shape[0], _p64(ch), ch.shape[0], _p64(r), C.byref(ok)))
    return bool(ok.value)


def ipa_verify_msm(srs, chals, sg_weights, extra_xy, extra_scalars, extra_inf=None) -> bool:
    ch = _c64(chals, (-1, 4)); w = _c64(sg_weights, (-1, 4)); ex = _c64(extra_xy, (-1, 8)); es = _c64(extra_scalars, (-1, 4))
    if extra_inf is None:
        extra_inf = np.zeros(ex.shape[0], dtype=np.uint8)
    z = C.c_int(0)
    _check(_lib.kh_ipa_verify_msm(srs._h, _p64(ch), ch.shape[0], _p64(w), w.shape[0], _p64(ex), _p8(np.ascontiguousarray(extra_inf, dtype=np.uint8)),
                                  _p64(es), ex.shape[0], C.byref(z)))
    return bool(z.value)


def _chal_limbs(chal: int):
    return np.array([chal & (2**64 - 1), (chal >> 64) & (2**64 - 1)], dtype=np.uint64)


def scalar_challenge_to_field(curve: int, chal: int):
    out = np.zeros(4, dtype=np.uint64)
    _check(_lib.kh_scalar_challenge_to_field(curve, _p64(_chal_limbs(chal)), _p64(out)))
    return out


class IpaOpening:
    """The folding loop of SRS::open (ipa.rs:929-1007) on device-resident vectors; see include/kimchi_hip.h."""

    def __init__(self, srs, a, b, u_base, a_len: int = None, b_len: int = None):
        """a, b: host limb arrays, or DevBuf objects with a_len / b_len elements (kh_ipa_begin_dev)."""
        u_base = _c64(u_base, (8,))
        h = C.c_void_p()
        if isinstance(a, DevBuf):
            _check(_lib.kh_ipa_begin_dev(srs._h, C.c_void_p(a.ptr), a_len, C.c_void_p(b.ptr), b_len, _p64(u_base), C.byref(h)))
        else:
            a = _c64(a, (-1, 4)); b = _c64(b, (-1, 4))
            _check(_lib.kh_ipa_begin(srs._h, _p64(a), a.shape[0], _p64(b), b.shape[0], _p64(u_base), C.byref(h)))
        self._h = h

    def rounds_left(self) -> int:
        return _lib.kh_ipa_rounds_left(self._h)

    def round_lr(self, rand_l, rand_r):
        rand_l = _c64(rand_l, (4,)); rand_r = _c64(rand_r, (4,))
        xy = np.zeros((2, 8), dtype=np.uint64); inf = np.zeros(2, dtype=np.uint8)
        _check(_lib.kh_ipa_round_lr(self._h, _p64(rand_l), _p64(rand_r), _p64(xy), _p8(inf)))
        return xy, inf

    def round_fold(self, chal: int):
        u = np.zeros(4, dtype=np.uint64); ui = np.zeros(4, dtype=np.uint64)
        _check(_lib.kh_ipa_round_fold(self._h, _p64(_chal_limbs(chal)), _p64(u), _p64(ui)))
        return u, ui

    def finish(self):
        a0 = np.zeros(4, dtype=np.uint64); b0 = np.zeros(4, dtype=np.uint64)
        sg = np.zeros(8, dtype=np.uint64); inf = np.zeros(1, dtype=np.uint8)
        _check(_lib.kh_ipa_finish(self._h, _p64(a0), _p64(b0), _p64(sg), _p8(inf)))
        return a0, b0, sg, bool(inf[0])

    def free(self):
        if self._h:
            _lib.kh_ipa_free(self._h); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def ipa_open(srs, a_dev, b_dev, n: int, combined_inner_product, blinding_factor, sponge, blinders):
    """kh_ipa_open: the rounds and the Schnorr tail of SRS::open in one native call.  Returns (lr_xy (k, 2, 8), lr_inf (k, 2),
    delta_xy, delta_inf, z1, z2, sg_xy, sg_inf)."""
    bl = _c64(blinders, (-1, 4))
    k = (bl.shape[0] - 2) // 2
    lr = np.zeros((k, 2, 8), dtype=np.uint64); lri = np.zeros((k, 2), dtype=np.uint8)
    delta = np.zeros(8, dtype=np.uint64); dinf = np.zeros(1, dtype=np.uint8)
    z1 = np.zeros(4, dtype=np.uint64); z2 = np.zeros(4, dtype=np.uint64)
    sg = np.zeros(8, dtype=np.uint64); sginf = np.zeros(1, dtype=np.uint8)
    _check(_lib.kh_ipa_open(srs._h, C.c_void_p(a_dev.ptr), n, C.c_void_p(b_dev.ptr), n, _p64(_c64(combined_inner_product, (4,))), _p64(_c64(blinding_factor, (4,))),
                            sponge._h, _p64(bl), bl.shape[0], _p64(lr), _p8(lri), _p64(delta), _p8(dinf), _p64(z1), _p64(z2), _p64(sg), _p8(sginf)))
    return lr, lri, delta, bool(dinf[0]), z1, z2, sg, bool(sginf[0])


def endos(curve: int):
    """(endo_q, endo_r) as Montgomery limb arrays; host-only."""
    q = np.zeros(4, dtype=np.uint64); r = np.zeros(4, dtype=np.uint64)
    _check(_lib.kh_endos(curve, _p64(q), _p64(r)))
    return q, r


def domain_generator(field: int, log2_n: int):
    out = np.zeros(4, dtype=np.uint64)
    _check(_lib.kh_domain_generator(field, log2_n, _p64(out)))
    return out


class Sponge:
    """kh_sponge_*: the host-side Fiat-Shamir sponges (FQ: DefaultFqSponge of `curve`, FR: DefaultFrSponge of `curve`)."""
    FQ, FR = 0, 1

    def __init__(self, kind: int, curve: int, _h=None):
        self.kind, self.curve = kind, curve
        self._h = C.c_void_p()
        if _h is None:
            _check(_lib.kh_sponge_new(kind, curve, C.byref(self._h)))
        else:
            self._h = _h

    def clone(self):
        h = C.c_void_p()
        _check(_lib.kh_sponge_clone(self._h, C.byref(h)))
        return Sponge(self.kind, self.curve, h)

    def absorb_g(self, xy, inf=None):
        xy = _c64(xy, (-1, 8))
        i8 = None if inf is None else np.ascontiguousarray(inf, dtype=np.uint8)
        _check(_lib.kh_sponge_absorb_g(self._h, _p64(xy), _p8(i8), xy.shape[0]))

    def absorb(self, x):
        x = _c64(x, (-1, 4))
        _check(_lib.kh_sponge_absorb(self._h, _p64(x), x.shape[0]))

    def absorb_fr(self, x):
        x = _c64(x, (-1, 4))
        _check(_lib.kh_sponge_absorb_fr(self._h, _p64(x), x.shape[0]))

    def challenge(self) -> int:
        c = np.zeros(2, dtype=np.uint64)
        _check(_lib.kh_sponge_challenge(self._h, _p64(c)))
        return int(c[0]) | (int(c[1]) << 64)

    def challenge_field(self):
        out = np.zeros(4, dtype=np.uint64)
        _check(_lib.kh_sponge_challenge_field(self._h, _p64(out)))
        return out

    def squeeze_field(self):
        out = np.zeros(4, dtype=np.uint64)
        _check(_lib.kh_sponge_squeeze_field(self._h, _p64(out)))
        return out

    def digest(self):
        out = np.zeros(4, dtype=np.uint64)
        _check(_lib.kh_sponge_digest(self._h, _p64(out)))
        return out

    def free(self):
        if self._h:
            _lib.kh_sponge_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync():
    _check(_lib.kh_sync())


def last_timings():
    names = (C.c_char_p * 32)()
    ms = (C.c_float * 32)()
    n = _lib.kh_last_timings(names, ms, 32)
    return [(names[i].decode(), float(ms[i])) for i in range(n)]


def debug_field_op(field: int, op: str, a, b=None):
    ops = {"mul": 0, "add": 1, "sub": 2, "to_mont": 3, "from_mont": 4, "sqr": 5, "neg": 6,
           "mul29": 7, "sqr29": 8, "pack29": 9, "mul29_32x": 10}      # field29.cuh (nine 29-bit limbs, lazy reduction)
    a = _c64(a, (-1, 4))
    bb = None if b is None else _c64(b, (-1, 4))
    out = np.zeros_like(a)
    _check(_lib.kh_debug_field_op(field, ops[op], _p64(a), None if bb is None else _p64(bb), _p64(out), a.shape[0]))
    return out


def debug_point_op(curve: int, op: int, p, q, p_inf=None, q_inf=None):
    p = _c64(p, (-1, 8)); q = _c64(q, (-1, 8))
    n = p.shape[0]
    pi = None if p_inf is None else np.ascontiguousarray(p_inf, dtype=np.uint8)
    qi = None if q_inf is None else np.ascontiguousarray(q_inf, dtype=np.uint8)
    out = np.zeros((n, 8), dtype=np.uint64)
    oinf = np.zeros(n, dtype=np.uint8)
    _check(_lib.kh_debug_point_op(curve, op, _p64(p), _p8(pi), _p64(q), _p8(qi), _p64(out), _p8(oinf), n))
    return out, oinf
