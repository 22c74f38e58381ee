"""Speculative decoding on MI355X — the `src/specdec` pipeline surface of the reference
(`SpeculativePipeline` with `generate` / `generate_batch`; src/specdec/__init__.py:13-18).
`SpecDecRunner` is an alias: BASELINE.json's north_star uses that name for this class."""

from .core.pipeline import SpeculativePipeline
from .models.hip_lm import HipLM, create_hip_lm

SpecDecRunner = SpeculativePipeline

__all__ = ["SpeculativePipeline", "SpecDecRunner", "HipLM", "create_hip_lm"]
