"""Verification scheduling (reference: src/scheduler/)."""
from .speculative_scheduler import SpeculativeScheduler, create_speculative_scheduler  # noqa: F401
