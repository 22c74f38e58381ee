"""`KVCache` — per-layer (key, value) container with prefix slicing
(reference: src/specdec/cache/kv_types.py:15-156; keys/values are [B, H, L, D]).

This is the interchange type of the `LanguageModel` KV hooks (`get_kv_cache`,
`append_kv_cache`) and of the registry op `kv_append`. The HIP engine's own cache is a
preallocated buffer that is appended to in place; `HipLM.get_kv_cache()` exposes it as
views in this type."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import torch


@dataclass
class KVCache:
    past_key_values: Tuple[Tuple[torch.Tensor, torch.Tensor], ...]
    seq_len: int
    dtype: torch.dtype
    device: torch.device

    @classmethod
    def from_hf_output(cls, past_key_values) -> "KVCache":
        if not past_key_values or len(past_key_values) == 0:
            raise ValueError("Cannot create KVCache from empty past_key_values")
        k0 = past_key_values[0][0]
        return cls(past_key_values=past_key_values, seq_len=k0.shape[2], dtype=k0.dtype, device=k0.device)

    def slice_prefix(self, length: int) -> "KVCache":
        if length > self.seq_len:
            raise ValueError(f"Cannot slice length {length} from cache with seq_len {self.seq_len}")
        kv = tuple((k[:, :, :length, :], v[:, :, :length, :]) for k, v in self.past_key_values)
        return KVCache(past_key_values=kv, seq_len=length, dtype=self.dtype, device=self.device)

    def to(self, device: torch.device) -> "KVCache":
        if self.device == device:
            return self
        kv = tuple((k.to(device), v.to(device)) for k, v in self.past_key_values)
        return KVCache(past_key_values=kv, seq_len=self.seq_len, dtype=self.dtype, device=device)

    def get_num_layers(self) -> int:
        return len(self.past_key_values)

    def get_shapes(self):
        if len(self.past_key_values) == 0:
            return ((), ())
        k, v = self.past_key_values[0]
        return (tuple(k.shape), tuple(v.shape))


def validate_kv_compatibility(base_cache: KVCache, new_cache: KVCache) -> None:
    if base_cache.get_num_layers() != new_cache.get_num_layers():
        raise ValueError(f"Layer count mismatch: base={base_cache.get_num_layers()}, new={new_cache.get_num_layers()}")
    if base_cache.dtype != new_cache.dtype:
        raise ValueError(f"Dtype mismatch: base={base_cache.dtype}, new={new_cache.dtype}")
    (bk, _), (nk, _) = base_cache.get_shapes(), new_cache.get_shapes()
    if bk[0] != nk[0] or bk[1] != nk[1] or bk[3] != nk[3]:
        raise ValueError(f"Shape mismatch (excluding seq_len): base={base_cache.get_shapes()}, new={new_cache.get_shapes()}")
