"""`SpeculativeScheduler` — dispatches the verify pass and applies the policy
(reference: src/scheduler/speculative_scheduler.py:43-512; same entry points:
`schedule_verification(base_model, draft_tokens, input_ids, ...) -> (base_tokens,
base_logits, info)` and `apply_acceptance_policy`).

The reference verifies by asking the base model for K greedy tokens from the same prefix
(`generate_tokens(T=1.0, do_sample=False)`, :192-199, 310-316, 347-353) on a second CUDA
stream. A model wrapper that offers `verify_tokens` (HipLM) is instead asked to score
(last, d_1..d_K) in ONE forward; the returned tokens/logits satisfy the same contract
(position i holds the target's greedy token and logits after the prefix + d_1..d_i),
which for the accepted prefix and the bonus position is what the autoregressive pass
yields. Wrappers without it get the reference behaviour."""

from __future__ import annotations

import logging
import os
import time
from typing import Any, Dict, Optional, Tuple

import torch

from kernels import get_kernel_info

logger = logging.getLogger(__name__)


class SpeculativeScheduler:
    def __init__(self, device: str = "cuda", enable_multi_stream: Optional[bool] = None,
                 enable_batched_verification: bool = True):
        self.device = device
        if enable_multi_stream is None:
            enable_multi_stream = os.getenv("SPECDEC_PARALLEL_STREAMS", "1").lower() in ("1", "true", "yes")
        self.use_event_sync = os.getenv("SPECDEC_SYNC_MODE", "event").lower() == "event"
        self.enable_multi_stream = bool(enable_multi_stream) and device == "cuda" and torch.cuda.is_available()
        self.enable_batched_verification = enable_batched_verification
        # The reference creates a verification stream and an event here and overlaps its base pass with drafting
        # (:123-190); that pass does not read the draft tokens. The one-pass verify does, so there is nothing to overlap
        # (measured: profiles/round2_row_group_concurrency.md) and no stream or event is created — the attributes stay for
        # callers that look at them.
        self.verification_stream = None
        self.default_stream = None
        self.verify_ready_event = None
        self.kernels_available = True
        self.kernel_info = get_kernel_info()
        self.metrics = {"total_proposed": 0, "total_accepted": 0, "total_steps": 0,
                        "verification_time_ms": 0.0, "draft_time_ms": 0.0, "overlap_time_ms": 0.0}

    def schedule_verification(self, base_model, draft_tokens: torch.Tensor, input_ids: torch.Tensor,
                              temperature: float = 0.7, do_sample: bool = True, **kwargs
                              ) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, Any]]:
        k = draft_tokens.shape[1]
        t0 = time.time()
        if hasattr(base_model, "verify_tokens"):
            ids, logits = base_model.verify_tokens(input_ids, draft_tokens)
            base_tokens, base_logits = ids[:, :k], logits  # [B,k], [B,k+1,V] (bonus position included)
            method = "parallel_verify"
        else:
            base_tokens, base_logits = base_model.generate_tokens(input_ids, max_new_tokens=k, temperature=1.0,
                                                                  do_sample=False, **kwargs)
            method = "autoregressive"
        if torch.cuda.is_available() and base_logits.is_cuda:
            torch.cuda.current_stream().synchronize()
        ms = (time.time() - t0) * 1e3
        self.metrics["verification_time_ms"] += ms
        self.metrics["total_steps"] += 1
        return base_tokens, base_logits, {"verification_time_ms": ms, "method": method,
                                          "multi_stream": self.enable_multi_stream}

    def apply_acceptance_policy(self, policy, draft_tokens, base_tokens, draft_logits, base_logits
                                ) -> Tuple[int, Dict[str, Any]]:
        accepted_len, info = policy.accept_tokens(draft_tokens, base_tokens, draft_logits, base_logits)
        self.metrics["total_proposed"] += draft_tokens.shape[1]
        self.metrics["total_accepted"] += accepted_len
        return accepted_len, info

    def get_metrics(self) -> Dict[str, Any]:
        m = dict(self.metrics)
        m["acceptance_rate"] = m["total_accepted"] / max(m["total_proposed"], 1)
        return m

    def reset_metrics(self) -> None:
        for k in self.metrics:
            self.metrics[k] = 0 if isinstance(self.metrics[k], int) else 0.0


def create_speculative_scheduler(device: str = "cuda", **kwargs) -> SpeculativeScheduler:
    return SpeculativeScheduler(device=device, **kwargs)
