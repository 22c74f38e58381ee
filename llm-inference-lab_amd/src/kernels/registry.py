"""Kernel registry: priority- and device-filtered lookup of op implementations.

Same surface as the reference registry (src/kernels/registry.py:11-123):
`KernelRegistry.register(op_name, impl, priority, device="auto")`,
`get_best(op_name, device)`, `list_available(op_name, device)`,
`get_status(device)`, the class-level table `_kernels` and the module singleton
`registry`. An entry matches a query when its device equals the queried device or
is "auto"; among matches the highest priority wins and, at equal priority, the
earliest registration (stable order, as a sort-by-priority of the reference gives).
"""

from __future__ import annotations

import logging
from typing import Any, Callable, Dict, List, Optional

logger = logging.getLogger(__name__)


class KernelRegistry:
    """Class-level table: op name -> entries sorted by descending priority."""

    _kernels: Dict[str, List[Dict[str, Any]]] = {}

    @classmethod
    def register(cls, op_name: str, impl: Callable, priority: int, device: str = "auto") -> None:
        entries = cls._kernels.setdefault(op_name, [])
        entries.append({"function": impl, "priority": priority, "device": device})
        entries.sort(key=lambda e: e["priority"], reverse=True)  # stable
        logger.debug(
            "registered %s -> %s (priority=%s, device=%s)",
            op_name, getattr(impl, "__name__", repr(impl)), priority, device,
        )

    @classmethod
    def _matching(cls, op_name: str, device: str) -> List[Dict[str, Any]]:
        return [e for e in cls._kernels.get(op_name, []) if e.get("device") in (device, "auto")]

    @classmethod
    def get_best(cls, op_name: str, device: str) -> Optional[Callable]:
        if op_name not in cls._kernels:
            return None
        matches = cls._matching(op_name, device)
        if not matches:
            logger.warning("No kernels available for %s on device %s", op_name, device)
            return None
        fn = matches[0].get("function")
        return fn if callable(fn) else None

    @classmethod
    def list_available(cls, op_name: str, device: str) -> list:
        return [
            {"name": e["function"].__name__, "priority": e["priority"], "device": e["device"]}
            for e in cls._matching(op_name, device)
        ]

    @classmethod
    def get_status(cls, device: str) -> Dict[str, str]:
        status: Dict[str, str] = {}
        for op_name in cls._kernels:
            best = cls.get_best(op_name, device)
            status[op_name] = best.__name__ if callable(best) else "none"
        return status


registry = KernelRegistry()
