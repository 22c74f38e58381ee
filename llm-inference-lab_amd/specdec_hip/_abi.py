"""ctypes binding of the gfx950 C-ABI library (include/specdec_hip.h).

The library is the product; there is no CPU fallback behind it. If it cannot be
loaded this module raises, and every op built on it raises with it.
"""

from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path

# torch must be loaded BEFORE the library: the PyTorch-ROCm wheel bundles its own
# libamdhip64.so.7 and the library has to bind to that same runtime instance (one
# HIP context, torch's streams valid in our launches). Loading ours first pulls in
# /opt/rocm's copy and the process ends up with two runtimes ("no ROCm-capable
# device is detected" on the first launch).
import torch  # noqa: F401

PKG_DIR = Path(__file__).resolve().parent.parent
LIB_PATH = PKG_DIR / "lib" / "libspecdec_hip.so"

# enum sd_dtype (include/specdec_hip.h)
SD_F32, SD_F16, SD_BF16, SD_I32, SD_I64, SD_U8, SD_FP8_E4M3 = range(7)
SD_ABI_VERSION = 1

_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_size = ctypes.c_size_t

# symbol -> (restype, argtypes); every symbol declared in include/specdec_hip.h
SIGNATURES = {
    "sd_abi_version": (_c_int, []),
    "sd_last_error": (ctypes.c_char_p, []),
    "sd_verify_prefix_workspace": (_c_size, [_c_int, _c_int, _c_int]),
    "sd_verify_prefix": (
        _c_int,
        [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
         _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_void_p, _c_size, _c_void_p],
    ),
    "sd_sample_token": (
        _c_int,
        [_c_void_p, _c_int, _c_i64, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, ctypes.c_float, _c_int, ctypes.c_float,
         ctypes.c_uint64, _c_void_p, ctypes.c_uint32, _c_void_p, _c_void_p, _c_void_p],
    ),
    "sd_kv_append": (
        _c_int,
        [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int,
         _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p],
    ),
    "sd_kv_concat": (
        _c_int,
        [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
         _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_void_p],
    ),
    "sd_kv_append_masked": (
        _c_int,
        [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
         _c_void_p, _c_void_p,
         _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_void_p],
    ),
    # ---- decoder forward + step loop ----
    "sd_packed_bytes": (_c_size, [_c_void_p]),
    "sd_pack_weights": (_c_int, [_c_void_p, _c_void_p, _c_size, _c_void_p]),
    "sd_model_create": (_c_int, [_c_void_p, _c_void_p]),
    "sd_model_destroy": (_c_int, [_c_void_p]),
    "sd_quantize_fp8_rows": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "sd_model_pass_tokens": (_c_int, [_c_void_p]),
    "sd_model_hidden_rows": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p]),
    "sd_model_workspace_bytes": (_c_size, [_c_void_p]),
    "sd_model_kv_bytes": (_c_size, [_c_void_p, _c_int, _c_int]),
    "sd_model_bind": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_size]),
    "sd_model_kv_pool_bytes": (_c_size, [_c_void_p, _c_int, _c_int]),
    "sd_model_bind_paged": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_size]),
    "sd_model_forward": (
        _c_int,
        [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int,
         _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p],
    ),
    "sd_model_probe_gemv": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p,
                                     ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]),
    "sd_model_persist_tokens": (_c_int, [_c_void_p]),
    "sd_model_engine_status": (_c_int, [_c_void_p, ctypes.POINTER(ctypes.c_uint32), _c_void_p]),
    "sd_model_status_word": (ctypes.POINTER(ctypes.c_uint32), [_c_void_p]),
    "sd_model_engine_status_clear": (_c_int, [_c_void_p, _c_void_p]),
    "sd_model_set_persist_tokens": (_c_int, [_c_void_p, _c_int]),
    "sd_model_set_length_hint": (_c_int, [_c_void_p, _c_int]),
    "sd_model_persist_active": (_c_int, [_c_void_p, _c_int]),
    "sd_model_debug_rows": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "sd_model_probe_forward": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, ctypes.POINTER(ctypes.c_float),
                                        ctypes.POINTER(ctypes.c_double), _c_void_p, _c_size]),
    "sd_specdec_create": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "sd_specdec_destroy": (_c_int, [_c_void_p]),
    "sd_specdec_set_row": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "sd_specdec_set_sampling": (_c_int, [_c_void_p, _c_int, ctypes.c_float, _c_int, ctypes.c_float, ctypes.c_uint64,
                                         _c_void_p, _c_size, _c_void_p, _c_void_p]),
    "sd_specdec_set_adaptive": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_double, _c_void_p]),
    "sd_specdec_set_adaptive_row": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_double), _c_void_p]),
    "sd_specdec_set_medusa": (_c_int, [_c_void_p, _c_int, ctypes.POINTER(_c_void_p), _c_int]),
    "sd_specdec_eagle_bytes": (_c_size, [_c_int, _c_int, _c_int]),
    "sd_specdec_set_eagle": (_c_int, [_c_void_p, ctypes.c_float, _c_void_p, _c_size]),
    "sd_specdec_reset_eagle": (_c_int, [_c_void_p, _c_void_p]),
    "sd_packed_head_bytes": (_c_size, [_c_int, _c_int, _c_int]),
    "sd_pack_head": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_size, _c_void_p]),
    "sd_specdec_step": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int]),
    "sd_specdec_sync": (_c_int, [_c_void_p, _c_void_p]),
    "sd_specdec_launches": (ctypes.c_long, [_c_void_p]),
    "sd_specdec_wait": (_c_int, [_c_void_p, ctypes.c_long]),
    "sd_specdec_invalidate": (_c_int, [_c_void_p]),
    "sd_specdec_record": (ctypes.POINTER(ctypes.c_int32), [_c_void_p]),
    "sd_specdec_record_ints": (_c_int, [_c_void_p]),
}

_lock = threading.Lock()
_lib = None


class HipLibraryError(RuntimeError):
    """The C-ABI library is missing, stale, or returned an error code."""


def lib_path() -> Path:
    return Path(os.environ.get("SPECDEC_HIP_LIB", str(LIB_PATH)))


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """dlopen libspecdec_hip.so and bind every declared symbol (idempotent)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not path.exists() and build_if_missing and "SPECDEC_HIP_LIB" not in os.environ:
            from importlib import util as _u

            spec = _u.spec_from_file_location("_specdec_hip_build", PKG_DIR / "build.py")
            mod = _u.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build()
        if not path.exists():
            raise HipLibraryError(
                f"{path} not found: build it with `python llm-inference-lab_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        try:
            lib = ctypes.CDLL(str(path))
        except OSError as e:  # missing libamdhip64 etc.
            raise HipLibraryError(f"cannot load {path}: {e}") from e
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise HipLibraryError(f"{path} does not export {name}; rebuild it") from e
            fn.restype = restype
            fn.argtypes = argtypes
        got = lib.sd_abi_version()
        if got != SD_ABI_VERSION:
            raise HipLibraryError(f"{path}: ABI version {got}, binding expects {SD_ABI_VERSION}")
        _lib = lib
        return lib


def last_error() -> str:
    return (load().sd_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    """Raise on a non-zero return code, carrying sd_last_error()."""
    if rc != 0:
        raise HipLibraryError(f"{what} failed (rc={rc}): {last_error()}")
