"""Token-id range check + clamp (reference: src/specdec/utils/token_validation.py:15-85).

One validation point: ids outside [0, vocab) are reported and clamped. The HIP forward
applies the same clamp in its embedding gather (csrc/misc.hip), so an invalid id can
never index past the embedding table on the device either."""

from __future__ import annotations

import logging
from typing import Any, Optional

import torch

logger = logging.getLogger(__name__)


def validate_and_clamp_tokens(input_ids: Optional[torch.Tensor], vocab_size: int, name: str = "input",
                              strict: bool = False) -> Optional[torch.Tensor]:
    if input_ids is None or input_ids.numel() == 0:
        return input_ids
    bad = (input_ids >= vocab_size) | (input_ids < 0)
    if not bool(bad.any()):
        return input_ids
    logger.error("[%s] Input ID out of bounds detected! Min: %s, Max: %s, Vocab_size: %s, Invalid_count: %s/%s",
                 name, int(input_ids.min()), int(input_ids.max()), vocab_size, int(bad.sum()), input_ids.numel())
    return input_ids.clamp(min=0, max=vocab_size - 1)


def get_vocab_size(model: Any) -> Optional[int]:
    """Vocabulary size of a wrapper: model config first, tokenizer info second
    (reference token_validation.py:81-85 semantics)."""
    for attr in ("vocab_size",):
        v = getattr(model, attr, None)
        if isinstance(v, int):
            return v
    inner = getattr(model, "_model", None)
    cfg = getattr(inner, "config", None)
    v = getattr(cfg, "vocab_size", None) or getattr(cfg, "vocab", None)
    if isinstance(v, int):
        return v
    try:
        v = model.get_tokenizer_info().get("vocab_size")
        return int(v) if v is not None else None
    except Exception:
        return None
