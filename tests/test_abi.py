"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports
every symbol include/pandrs_hip.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from pandrs_amd import _lib
    return _lib


def test_header_symbols_all_exported(lib):
    text = open(os.path.join(ROOT, "include", "pandrs_hip.h")).read()
    declared = set(re.findall(r"^(?:int32_t|const char \*)\s*(pandrs_hip_[a-z0-9_]+)\s*\(", text, re.M))
    assert len(declared) >= 32
    handle = lib.load()
    for name in sorted(declared):
        assert hasattr(handle, name), "libpandrs_hip.so lacks %s" % name
    assert declared == set(lib.SYMBOLS), "ctypes table and header disagree: %s" % (declared ^ set(lib.SYMBOLS))
    assert handle.pandrs_hip_abi_version() == 1


def test_struct_layouts_match_header(lib):
    import ctypes as C
    assert C.sizeof(lib.Column) == 24 and C.sizeof(lib.AggSpec) == 8
    assert C.sizeof(lib.Config) == 32
    assert C.sizeof(lib.Timings) == 8 + 8 * lib.MAX_PHASES + 6 * 8
    assert C.sizeof(lib.ColumnStats) == 11 * 8


def test_no_cpu_fallback_without_gpu(lib):
    import ctypes as C
    handle = lib.load()
    n = C.c_int32(-1)
    assert handle.pandrs_hip_device_count(C.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is present")
    import pandrs_amd as pa
    with pytest.raises(pa.PandrsHipError) as e:
        pa.Context(0)
    assert e.value.status == lib.ERR_NOT_INITIALIZED


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under pandrs_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pandrs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower(), "%s mentions the oracle" % os.path.join(dirpath, f)


def test_every_set_option_name_is_documented_in_the_header():
    """pandrs_hip.h documented a third of the option names the tests use.  The header's list and the
    strcmp chain of pandrs_hip_ctx_set_option must name the same options."""
    import re
    capi = open(os.path.join(ROOT, "pandrs_amd", "csrc", "capi.hip")).read()
    body = capi[capi.index("int32_t pandrs_hip_ctx_set_option("):]
    body = body[:body.index('on_exception("pandrs_hip_ctx_set_option")')]
    accepted = set(re.findall(r'std::strcmp\(name, "(\w+)"\)', body))
    header = open(os.path.join(ROOT, "include", "pandrs_hip.h")).read()
    doc = header[header.index("Tuning / testing knobs"):header.index("int32_t pandrs_hip_ctx_set_option(")]
    documented = set(re.findall(r'"(\w+)"', doc))
    assert accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))


def test_every_entry_point_has_the_exception_firewall():
    """Every extern "C" function the header declares is defined as a function-try-block whose handler maps the exception to a
    status (common.hpp on_exception): std::bad_alloc -> OUT_OF_MEMORY, anything else -> COMPUTATION.  (pandrs_hip_last_error
    returns a pointer into a fixed thread-local buffer and cannot throw.)"""
    header = open(os.path.join(ROOT, "include", "pandrs_hip.h")).read()
    declared = set(re.findall(r"^int32_t\s*(pandrs_hip_[a-z0-9_]+)\s*\(", header, re.M))
    guarded, defined = set(), set()
    for f in ("capi.hip", "dist.hip"):
        src = open(os.path.join(ROOT, "pandrs_amd", "csrc", f)).read()
        src = src[src.index('extern "C" {'):]
        defined |= set(re.findall(r"^int32_t\s+(pandrs_hip_\w+)\s*\(", src, re.M))
        guarded |= set(re.findall(r'catch \(\.\.\.\) \{ return pandrs::on_exception\("(pandrs_hip_\w+)"\); \}', src))
        for m in re.finditer(r"^int32_t\s+(pandrs_hip_\w+)\s*\([^{;]*\)\s*(try)?\s*\{", src, re.M):
            assert m.group(2) == "try", "%s in %s is not a function-try-block" % (m.group(1), f)
    assert declared == defined == guarded, (declared ^ defined, defined ^ guarded)


def test_exceptions_become_status_codes(lib):
    handle = lib.load()
    assert handle.pandrs_hip_ctx_set_option(None, b"test_throw", 1) == lib.ERR_OUT_OF_MEMORY
    assert b"bad_alloc" in handle.pandrs_hip_last_error()
    assert handle.pandrs_hip_ctx_set_option(None, b"test_throw", 2) == lib.ERR_COMPUTATION
    assert b"C++ exception" in handle.pandrs_hip_last_error()
    assert handle.pandrs_hip_ctx_set_option(None, b"test_throw", 3) == lib.ERR_COMPUTATION
    assert handle.pandrs_hip_ctx_set_option(None, b"test_throw", 4) in (lib.ERR_OUT_OF_MEMORY, lib.ERR_COMPUTATION)
    assert handle.pandrs_hip_ctx_set_option(None, b"test_throw", 0) == 0
