"""Host helpers for ragged batches of token sequences
(reference: src/specdec/core/sequence_utils.py:15-184 — same names and return values).

The HIP path itself never pads: every row keeps its own length on the device
(`cur_len[b]`) and the kernels index the KV cache per row. These helpers exist for
callers of the reference API (and its tests) that hand padded batches around."""

from __future__ import annotations

from typing import List, Tuple

import torch


def pad_sequences(sequences: List[torch.Tensor], pad_token_id: int, device: torch.device
                  ) -> Tuple[torch.Tensor, torch.Tensor, List[int]]:
    """RIGHT-pad 1-D sequences to a [B, max_len] tensor; mask is 1 on real tokens."""
    if not sequences:
        empty = torch.empty(0, 0, dtype=torch.long, device=device)
        return empty, empty.clone(), []
    lengths = [int(s.shape[0]) for s in sequences]
    width = max(lengths)
    batch = torch.full((len(sequences), width), pad_token_id, dtype=sequences[0].dtype, device=device)
    mask = torch.zeros((len(sequences), width), dtype=torch.long, device=device)
    for i, (s, n) in enumerate(zip(sequences, lengths)):
        batch[i, :n] = s.to(device)
        mask[i, :n] = 1
    return batch.contiguous(), mask, lengths


def unpad_sequences(batch_tensor: torch.Tensor, attention_mask: torch.Tensor) -> List[torch.Tensor]:
    lengths = attention_mask.sum(dim=1).tolist()
    return [batch_tensor[i, : int(n)] for i, n in enumerate(lengths)]


def unpad_append_repad(sequences: List[torch.Tensor], tokens_to_append: List[torch.Tensor], pad_token_id: int,
                       device: torch.device) -> Tuple[torch.Tensor, torch.Tensor, List[int]]:
    if len(sequences) != len(tokens_to_append):
        raise ValueError(f"Number of sequences ({len(sequences)}) must match number of token lists to append "
                         f"({len(tokens_to_append)})")
    grown = [torch.cat([s, t], dim=0) if t.shape[0] > 0 else s for s, t in zip(sequences, tokens_to_append)]
    return pad_sequences(grown, pad_token_id, device)


def create_position_ids(sequence_lengths: List[int], max_length: int, device: torch.device) -> torch.Tensor:
    """positions 0..len-1 on the real tokens of each row, 0 on padding."""
    pos = torch.arange(max_length, device=device, dtype=torch.long).unsqueeze(0).repeat(len(sequence_lengths), 1)
    lens = torch.tensor(sequence_lengths, device=device, dtype=torch.long).unsqueeze(1)
    return torch.where(pos < lens, pos, torch.zeros_like(pos))
