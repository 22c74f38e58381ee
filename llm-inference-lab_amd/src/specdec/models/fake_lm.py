"""`FakeLM` — the reference's weight-less test double behind `implementation: fake`
(/root/reference/src/specdec/models/fake_lm.py:19-146; created by the pipeline at pipeline.py:466-499), restated so that the
reference's own `configs/specdec.yaml` (`implementation: fake`, base `gpt2`, draft `distilgpt2`) runs on this build unchanged.

What the double is: a `LanguageModel` whose k next tokens are a pure function of the input ids — token i is
`(hash(tuple(ids)) + i) mod vocab`, moved on by one when it lands on a special id (pad 0, eos 1, bos 2, unk 3) — and whose
"logits" are noise (`torch.randn`), so the exact-match policy sees the target's argmax disagree with every proposal and each step
falls back to one target token. It exercises the step loop, the policies, the controllers and — here — the registry's device ops
(the noise logits live on the GPU: `verify_prefix` runs as the HIP kernel), without a model. It is not a compute fallback: the
pipeline's model path is `implementation: hip`.

Pinned by tests/golden/fake_pipeline_golden.json (the reference pipeline run from its own YAML): the token
function row for row, and `generate` / `generate_batch` traces — tokens, proposed, accepted, steps."""

from __future__ import annotations

import logging
import random
from typing import Any, Dict, Optional, Tuple

import torch

from ..utils.interfaces import LanguageModel

logger = logging.getLogger(__name__)

PAD, EOS, BOS, UNK = 0, 1, 2, 3


class FakeLM(LanguageModel):
    def __init__(self, model_name: str = "fake-model", vocab_size: int = 1000, device: str = "cuda", seed: Optional[int] = None):
        self._name, self.vocab_size, self._device = model_name, int(vocab_size), device
        if seed is not None:            # the double seeds the global generators, as the reference's does
            random.seed(seed)
            torch.manual_seed(seed)

    # ---- LanguageModel ---------------------------------------------------------------------------------------------------
    def generate_tokens(self, input_ids: torch.Tensor, max_new_tokens: int, temperature: float = 0.7, do_sample: bool = True,
                        **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """(ids [1, k] int64, noise logits [1, k, V]); temperature / do_sample / row hints are ignored, as in the reference."""
        ids = input_ids if input_ids.dim() == 2 else input_ids.unsqueeze(0)
        base = hash(tuple(int(x) for x in ids[0].tolist()))     # (a tuple of ints: not subject to PYTHONHASHSEED)
        toks = []
        for i in range(int(max_new_tokens)):
            t = (base + i) % self.vocab_size
            if t in (PAD, EOS, BOS, UNK):
                t = (t + 1) % self.vocab_size
            toks.append(t)
        dev = self._device if self._device != "auto" else "cuda"
        out = torch.tensor([toks], dtype=torch.long, device=dev)
        logits = torch.randn(1, int(max_new_tokens), self.vocab_size, device=dev)
        return out, logits

    def get_tokenizer_info(self) -> Dict[str, Any]:
        return {"model_name": self._name, "vocab_size": self.vocab_size, "pad_token_id": PAD, "eos_token_id": EOS, "bos_token_id": BOS,
                "unk_token_id": UNK}

    def encode(self, text: str) -> torch.Tensor:
        """3-5 ids from the TEXT's hash (process-dependent unless PYTHONHASHSEED is fixed — as in the reference; callers that need
        reproducible prompts pass ids)."""
        h = hash(text)
        n = min(max(3, len(text) // 2), 5)
        return torch.tensor([[(h + i) % self.vocab_size for i in range(n)]], dtype=torch.long)

    def decode(self, token_ids: Any) -> str:
        if isinstance(token_ids, torch.Tensor):
            if token_ids.numel() == 0:
                return ""
            token_ids = token_ids.flatten().tolist()
        elif token_ids and isinstance(token_ids[0], (list, tuple)):
            token_ids = list(token_ids[0])
        if not token_ids:
            return ""
        return "fake_text_" + "_".join(str(int(t)) for t in list(token_ids)[:3])

    @property
    def device(self) -> str:
        return self._device

    @property
    def model_name(self) -> str:
        return self._name


def create_fake_lm(model_name: str = "fake-model", vocab_size: int = 1000, device: str = "cuda", seed: Optional[int] = None) -> FakeLM:
    return FakeLM(model_name=model_name, vocab_size=vocab_size, device=device, seed=seed)
