"""K controllers (reference: src/specdec/policies/controllers.py:38-173).

`FixedKController(k=4)` and `AdaptiveKController`: the adaptive rule looks at the mean of
the last four reported acceptance rates and moves K by `step_size` when that mean leaves
the band target±0.1, clamped to [min_k, max_k]; histories are kept to `window_size`."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, List


class KController(ABC):
    @abstractmethod
    def get_k(self, step: int, context: Dict[str, Any]) -> int: ...

    @abstractmethod
    def get_info(self) -> Dict[str, Any]: ...


class FixedKController(KController):
    def __init__(self, k: int = 4):
        self.k = k
        self.name = "fixed_k"

    def get_k(self, step: int, context: Dict[str, Any]) -> int:
        return self.k

    def get_info(self) -> Dict[str, Any]:
        return {"controller": self.name, "k": self.k}


class AdaptiveKController(KController):
    _BAND = 0.1
    _MIN_HISTORY = 4

    def __init__(self, initial_k: int = 4, min_k: int = 1, max_k: int = 8, step_size: int = 1,
                 window_size: int = 32, target_acceptance_rate: float = 0.7, per_row: bool = False):
        # per_row (not in the reference): every batch row carries its own instance of this rule,
        # evaluated on the device inside the captured step (sd_specdec_set_adaptive); this object holds the parameters
        self.per_row = bool(per_row)
        self.initial_k, self.min_k, self.max_k = initial_k, min_k, max_k
        self.step_size, self.window_size = step_size, window_size
        self.target_acceptance_rate = target_acceptance_rate
        self.name = "adaptive_k"
        self.current_k = initial_k
        self.acceptance_history: List[float] = []
        self.k_history: List[int] = []

    def _recent(self):
        h = self.acceptance_history
        return sum(h[-self._MIN_HISTORY:]) / self._MIN_HISTORY if len(h) >= self._MIN_HISTORY else None

    def get_k(self, step: int, context: Dict[str, Any]) -> int:
        if "acceptance_rate" in context:
            self.acceptance_history.append(context["acceptance_rate"])
            del self.acceptance_history[: max(0, len(self.acceptance_history) - self.window_size)]
        recent = self._recent()
        if recent is not None:
            if recent > self.target_acceptance_rate + self._BAND:
                self.current_k = min(self.current_k + self.step_size, self.max_k)
            elif recent < self.target_acceptance_rate - self._BAND:
                self.current_k = max(self.current_k - self.step_size, self.min_k)
        self.k_history.append(self.current_k)
        del self.k_history[: max(0, len(self.k_history) - self.window_size)]
        return self.current_k

    def fork(self) -> "AdaptiveKController":
        """A fresh controller with the same parameters (one per batch row)."""
        return AdaptiveKController(self.initial_k, self.min_k, self.max_k, self.step_size, self.window_size,
                                   self.target_acceptance_rate)

    def get_info(self) -> Dict[str, Any]:
        return {
            "controller": self.name, "current_k": self.current_k, "min_k": self.min_k, "max_k": self.max_k,
            "step_size": self.step_size, "window_size": self.window_size,
            "target_acceptance_rate": self.target_acceptance_rate, "recent_acceptance_rate": self._recent(),
        }


def create_controller(controller_type: str, **kwargs: Any) -> KController:
    if controller_type == "fixed":
        return FixedKController(kwargs.get("k", 4))
    if controller_type == "adaptive":
        return AdaptiveKController(
            initial_k=kwargs.get("initial_k", 4), min_k=kwargs.get("min_k", 1), max_k=kwargs.get("max_k", 8),
            step_size=kwargs.get("step_size", 1), window_size=kwargs.get("window_size", 32),
            target_acceptance_rate=kwargs.get("target_acceptance_rate", 0.7), per_row=kwargs.get("per_row", False),
        )
    raise ValueError(f"Unknown controller: {controller_type}. Available: ['fixed', 'adaptive']")
