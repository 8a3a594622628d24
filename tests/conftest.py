import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / at round end)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library and the C oracle are built in-tree (git-ignored).  If a checkout arrives without
    them, build before the first test (hipcc cross-compiles without a GPU); a no-op when up to date."""
    lib = os.path.join(ROOT, "proof_systems_amd", "libkimchi_hip.so")
    ora = os.path.join(ROOT, "oracle", "_build", "libpasta_ref.so")
    exe = os.path.join(ROOT, "tests", "cpp", "test_mirror")
    exe2 = os.path.join(ROOT, "tests", "cpp", "test_prove")
    if not (os.path.exists(lib) and os.path.exists(ora) and os.path.exists(exe) and os.path.exists(exe2)):
        import __graft_entry__ as ge
        ge.build()
    yield


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)
