"""Seeding (reference: src/specdec/utils/deterministic.py:16-60; seed 1234 default).

`SPECDEC_DETERMINISTIC=1` seeds every generator the host logic can touch. It does not
force greedy decoding (neither does the reference): that is `do_sample=False`. The HIP
kernels are deterministic by construction (fixed reduction orders, no float atomics)."""

from __future__ import annotations

import os
import random

import numpy as np
import torch


def set_deterministic_mode(seed: int = 1234) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def ensure_deterministic(seed: int = 1234) -> bool:
    """Seed when SPECDEC_DETERMINISTIC is set; returns whether it was."""
    on = os.getenv("SPECDEC_DETERMINISTIC", "0").lower() in ("1", "true", "yes")
    if on:
        set_deterministic_mode(seed)
    return on
