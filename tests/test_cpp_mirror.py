"""The C++ host mirror of the reference API (include/pandrs_hip.hpp) — the compiled-language stand-in for
the Rust shim, since the reference is Rust and this image has no Rust toolchain — replaying the reference's
own tests (tests/cpp/reference_like_tests.cpp) over libpandrs_hip.so with no Python in the process."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(out):
    import __graft_entry__ as g
    g.build()
    libdir = os.path.join(ROOT, "pandrs_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_like_tests.cpp"), "-L" + libdir, "-lpandrs_hip",
                           "-L" + os.path.join(ROOT, "oracle"), "-lpandrs_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", out])


def test_cpp_mirror_compiles_and_fails_loudly_without_gpu():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "reference_like_tests")
        _build(exe)
        r = subprocess.run([exe], capture_output=True, text=True)
        if r.returncode != 0:
            assert r.returncode == 1 and "no HIP device available" in r.stderr, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_mirror_replays_the_reference_tests():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "reference_like_tests")
        _build(exe)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "15 tests, 0 failed checks" in r.stdout
