"""csrc/host_ec.hpp's field inversion (batched division steps, ~8x faster than a^(p-2); two sequential inversions sit in every opening round)
against its definition, both Pasta fields: tests/cpp/test_host_inv.cpp, built with g++ here -- host code only, no GPU, no library."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inversion_equals_fermat(tmp_path):
    exe = str(tmp_path / "test_host_inv")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "test_host_inv.cpp"), "-o", exe])
    out = subprocess.run([exe, "30000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300).stdout.decode()
    assert "HOST_INV_OK" in out and "MISMATCH" not in out, out
