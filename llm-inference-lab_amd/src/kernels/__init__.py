"""`kernels` — the drop-in boundary of the speculative-decoding hot path on MI355X.

Module surface of the reference (src/kernels/__init__.py:84-184):
`get_verify_prefix(device=None)`, `get_kv_append(device=None)`,
`get_kernel_info()`, `log_kernel_status()`, attributes `verify_prefix`,
`kv_append`, `registry`. The ops are the hand-written gfx950 kernels behind the
C-ABI library, registered at priority 100 for device "cuda" — the native device
string of PyTorch-ROCm, so callers that test `tensor.device.type == "cuda"`
(policies.py:123-126) reach them unchanged.

There is deliberately NO CPU implementation registered: on a host without the
library or without a GPU, lookups for "cpu" return None and calling an op with CPU
tensors raises. A silent fallback would void every parity claim.
"""

from __future__ import annotations

import logging
import os
from typing import Optional

import torch

from specdec_hip import _abi
from specdec_hip.ops import (  # noqa: F401
    kv_append_hip,
    kv_append_inplace_hip,
    kv_append_with_mask_hip,
    kv_concat_hip,
    verify_prefix_hip,
)

from . import registry as _registry_module  # noqa: F401  (eager: aliasable as kernels.registry)
from .registry import KernelRegistry, registry

logger = logging.getLogger(__name__)

# The reference reads this flag to skip Triton (kernels/__init__.py:18). There is
# no PyTorch backend to force here; the flag is recognised and refused loudly.
FORCE_PYTORCH_BACKEND = os.getenv("SPECDEC_FORCE_PYTORCH_BACKEND", "0").lower() in ("1", "true", "yes")

_HIP_PRIORITY = 100
_registered = False


def _register_kernels() -> None:
    """Register the HIP ops once (the class-level table survives re-imports)."""
    global _registered
    have = {e["function"] for op in ("verify_prefix", "kv_append", "kv_append_with_mask")
            for e in KernelRegistry._kernels.get(op, [])}
    if verify_prefix_hip not in have:
        registry.register("verify_prefix", verify_prefix_hip, priority=_HIP_PRIORITY, device="cuda")
    if kv_append_hip not in have:
        registry.register("kv_append", kv_append_hip, priority=_HIP_PRIORITY, device="cuda")
    if kv_append_with_mask_hip not in have:
        registry.register("kv_append_with_mask", kv_append_with_mask_hip,
                          priority=_HIP_PRIORITY, device="cuda")
    _registered = True


_register_kernels()
if FORCE_PYTORCH_BACKEND:
    logger.warning(
        "SPECDEC_FORCE_PYTORCH_BACKEND is set, but this build has no PyTorch backend: "
        "the HIP kernels stay registered"
    )


def _default_device() -> str:
    # PyTorch-ROCm reports the GPU as "cuda"; there is no "mps" on this platform.
    return "cuda" if torch.cuda.is_available() else "cpu"


def get_verify_prefix(device: Optional[str] = None):
    """Best verify_prefix implementation for `device` (None on a device without one)."""
    return registry.get_best("verify_prefix", device or _default_device())


def get_kv_append(device: Optional[str] = None):
    """Best kv_append implementation for `device` (None on a device without one)."""
    return registry.get_best("kv_append", device or _default_device())


def get_kv_append_with_mask(device: Optional[str] = None):
    return registry.get_best("kv_append_with_mask", device or _default_device())


# Bound once at import for the GPU device, as the reference binds them
# (kernels/__init__.py:111-112). They raise on CPU tensors.
verify_prefix = registry.get_best("verify_prefix", "cuda")
kv_append = registry.get_best("kv_append", "cuda")
kv_append_with_mask = registry.get_best("kv_append_with_mask", "cuda")


def _backend_of(fn_name: str) -> str:
    low = fn_name.lower()
    if low in ("none", "unknown"):
        return "unknown"
    if "hip" in low:
        return "hip"
    if "ref" in low:
        return "torch"
    return fn_name


def get_kernel_info():
    """Backend names per op. Keys as the reference (kernels/__init__.py:116-156);
    the HIP ops report "hip" (the reference's own allowed set has no such name)."""
    device = _default_device()
    status = registry.get_status("cuda")
    lib_ok = True
    try:
        _abi.load()
    except Exception:  # reported, never swallowed into a fallback
        lib_ok = False
    return {
        "verify_backend": _backend_of(status.get("verify_prefix", "unknown")),
        "kv_append_backend": _backend_of(status.get("kv_append", "unknown")),
        "verify_available": verify_prefix is not None and lib_ok,
        "kv_append_available": kv_append is not None and lib_ok,
        "device": device,
        "library": str(_abi.lib_path()),
        "library_loaded": lib_ok,
    }


def log_kernel_status() -> None:
    info = get_kernel_info()
    logger.info("Using verify backend: %s", info["verify_backend"])
    logger.info("Kernel backends: verify=%s, kv_append=%s", info["verify_backend"], info["kv_append_backend"])
    if not info["library_loaded"]:
        logger.error("HIP library %s could not be loaded; ops will raise", info["library"])


log_kernel_status()

__all__ = [
    "verify_prefix",
    "kv_append",
    "kv_append_with_mask",
    "get_verify_prefix",
    "get_kv_append",
    "get_kv_append_with_mask",
    "get_kernel_info",
    "log_kernel_status",
    "registry",
    "KernelRegistry",
]
