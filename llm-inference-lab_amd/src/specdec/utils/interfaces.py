"""`LanguageModel` — the model-wrapper interface of the pipeline
(reference: src/specdec/utils/interfaces.py:14-138; same method names and meanings)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, Tuple

import torch


class LanguageModel(ABC):
    @abstractmethod
    def generate_tokens(self, input_ids: torch.Tensor, max_new_tokens: int, temperature: float = 0.7,
                        do_sample: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (ids [B, k] int64, logits [B, k, V])"""

    @abstractmethod
    def get_tokenizer_info(self) -> Dict[str, Any]: ...

    @abstractmethod
    def encode(self, text: str) -> torch.Tensor: ...

    @abstractmethod
    def decode(self, token_ids: Any) -> str: ...

    @property
    @abstractmethod
    def device(self) -> str: ...

    @property
    @abstractmethod
    def model_name(self) -> str: ...

    @property
    def model(self) -> Any:
        return getattr(self, "_model", None)

    @property
    def tokenizer(self) -> Any:
        return getattr(self, "_tokenizer", None)

    def supports_kv_append(self) -> bool:
        return False

    def get_kv_cache(self) -> Any:
        return None

    def append_kv_cache(self, kv_chunk: Any) -> None:
        return None

    def clear_kv_cache(self) -> None:
        return None
