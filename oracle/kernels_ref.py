"""CPU restatement of the registry ops (reference: src/kernels/reference.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). Vectorised torch-CPU integer /
byte arithmetic; the reference's Python double loops are restated as closed forms.
"""

from __future__ import annotations

from typing import Tuple

import torch


def verify_prefix_oracle(logits: torch.Tensor, draft_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference.py:13-56 — argmax over V, equality with the draft ids, longest prefix.

    accept_len[b] = number of leading k with argmax(logits[b,k]) == draft_ids[b,k];
    accepted_mask[b,k] = 1 for k < accept_len[b] (prefix only: the reference loop
    breaks at the first mismatch, :47-53).
    """
    assert logits.dim() == 3 and draft_ids.dim() == 2
    assert logits.shape[:2] == draft_ids.shape
    B, K, _ = logits.shape
    logits = logits.detach().cpu()
    ids = draft_ids.detach().cpu().to(torch.int64)
    if K == 0:
        return torch.zeros(B, dtype=torch.int32), torch.zeros((B, 0), dtype=torch.uint8)
    pred = torch.argmax(logits.float() if logits.dtype != torch.float64 else logits, dim=-1)  # :36
    match = (pred == ids).to(torch.int64)  # :39
    prefix = torch.cumprod(match, dim=1)  # 1 while the prefix is unbroken
    return prefix.sum(dim=1).to(torch.int32), prefix.to(torch.uint8)


def argmax_oracle(logits: torch.Tensor) -> torch.Tensor:
    """torch.argmax(-1) on CPU: first index of the maximum, NaN counts as maximum."""
    return torch.argmax(logits.detach().cpu().float(), dim=-1)


def kv_append_oracle(base_k, base_v, new_k, new_v):
    """reference.py:59-93 — concatenation along the sequence axis (dim 2)."""
    for t in (base_k, base_v, new_k, new_v):
        assert t.dim() == 4
    assert base_k.shape[0] == new_k.shape[0], "Batch size mismatch"  # :85
    assert base_k.shape[1] == new_k.shape[1], "Num heads mismatch"  # :86
    assert base_k.shape[3] == new_k.shape[3], "Head dim mismatch"  # :87
    bk, bv, nk, nv = (t.detach().cpu() for t in (base_k, base_v, new_k, new_v))
    B, H, L, D = bk.shape
    K = nk.shape[2]
    out_k = bk.new_empty((B, H, L + K, D))
    out_v = bv.new_empty((B, H, L + K, D))
    out_k[:, :, :L] = bk
    out_k[:, :, L:] = nk
    out_v[:, :, :L] = bv
    out_v[:, :, L:] = nv
    return out_k, out_v


def kv_append_with_mask_oracle(base_k, base_v, draft_k, draft_v, accepted_mask, accept_len, offset: int = 0):
    """reference.py:96-159 — zero-init [B,H,L+K,D], copy base, then for each row the
    draft rows at the set mask positions are packed to L, L+1, … until accept_len[b]
    rows have been written (:140-157). A zero accept_len row keeps base + zeros.
    `offset` is accepted and unused, as in the reference.
    """
    bk, bv, dk, dv = (t.detach().cpu() for t in (base_k, base_v, draft_k, draft_v))
    mask = accepted_mask.detach().cpu()
    alen = accept_len.detach().cpu()
    B, H, L, D = bk.shape
    K = dk.shape[2]
    out_k = torch.zeros((B, H, L + K, D), dtype=bk.dtype)
    out_v = torch.zeros((B, H, L + K, D), dtype=bv.dtype)
    out_k[:, :, :L] = bk
    out_v[:, :, :L] = bv
    for b in range(B):
        want = int(alen[b])
        if want == 0:
            continue
        src = torch.nonzero(mask[b] != 0).flatten()[: max(want, 1)]
        n = src.numel()
        if n:
            out_k[b, :, L : L + n] = dk[b].index_select(1, src)
            out_v[b, :, L : L + n] = dv[b].index_select(1, src)
    return out_k, out_v
