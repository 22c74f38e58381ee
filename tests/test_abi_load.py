"""The C-ABI library loads on a CPU-only host and exports every declared symbol."""

import ctypes
import os
import re

from specdec_hip import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            text = open(os.path.join(inc, fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_header_symbol():
    lib = _abi.load()
    declared = _declared_symbols()
    assert declared, "no sd_* declarations found in include/*.h"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    # and the binding table covers the header exactly
    assert declared == set(_abi.SIGNATURES), declared ^ set(_abi.SIGNATURES)


def test_abi_version_and_error_string():
    lib = _abi.load()
    assert lib.sd_abi_version() == _abi.SD_ABI_VERSION
    assert isinstance(_abi.last_error(), str)


def test_argument_validation_without_gpu():
    """Validation errors are reported through the return code + sd_last_error, before
    any device work (so this is safe on a host without a GPU)."""
    lib = _abi.load()
    rc = lib.sd_kv_append(None, None, None, None, None, 0, 3, 1, 1, 8, 1, 4, None)
    assert rc != 0 and "elem_size" in _abi.last_error()
    rc = lib.sd_verify_prefix(None, 99, None, _abi.SD_I64, ctypes.c_void_p(8), None, None,
                              1, 1, 10, 10, 10, None, 0, None)
    assert rc != 0 and "NULL" in _abi.last_error() or "dtype" in _abi.last_error()
    assert lib.sd_verify_prefix_workspace(2, 4, 1000) >= 2 * 4 * 8
