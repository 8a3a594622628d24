"""proof_systems_amd -- MI355X-native MSM + NTT hot path for Kimchi.

The product is the C-ABI shared library ``libkimchi_hip.so`` (include/kimchi_hip.h),
built from ``csrc/`` by ``__graft_entry__.build()``.  This package is only the thin
ctypes binding used by the tests and bench.py; there is no CPU fallback: importing
``proof_systems_amd.khip`` raises if the HIP library is missing.
"""
