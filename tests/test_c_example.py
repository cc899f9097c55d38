"""The drop-in boundary from plain C: examples/groupby_c_abi.c includes only include/pandrs_hip.h and
links libpandrs_hip.so (system ROCm runtime, no Python / torch in the process)."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(out, source="groupby_c_abi.c"):
    import __graft_entry__ as g
    g.build()
    libdir = os.path.join(ROOT, "pandrs_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", source), "-L" + libdir, "-lpandrs_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", out])


def test_c_example_compiles_and_fails_loudly_without_gpu():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "groupby_c_abi")
        _build(exe)
        r = subprocess.run([exe], capture_output=True, text=True)
        if r.returncode == 0:
            assert "C ABI example: OK" in r.stdout      # a GPU is present
        else:
            assert r.returncode == 1 and "no HIP device available" in r.stderr


@pytest.mark.gpu
def test_c_example_runs_on_gpu():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "groupby_c_abi")
        _build(exe)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "C ABI example: OK" in r.stdout and "groups: 3" in r.stdout


def test_exceptions_never_cross_the_c_abi():
    """examples/exception_firewall.c (plain C, no GPU needed): the host code of an entry point throws std::bad_alloc, a
    std::exception, a foreign exception and a real oversized std::vector::resize; the caller sees PANDRS_HIP_ERR_OUT_OF_MEMORY /
    PANDRS_HIP_ERR_COMPUTATION with pandrs_hip_last_error() set, never an abort (include/pandrs_hip.h conventions,
    /root/reference/src/core/error.rs:6)."""
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "exception_firewall")
        _build(exe, "exception_firewall.c")
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "exception firewall: OK" in r.stdout and "UNEXPECTED" not in r.stdout
