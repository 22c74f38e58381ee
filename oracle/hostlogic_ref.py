"""CPU restatement of the reference's host-side pieces of the step loop.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). Pinned by tests/golden/hostlogic_golden.json
(outputs of the reference's own functions on seeded inputs)."""

from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch


def _leading_true(flags: Sequence[bool]) -> int:
    n = 0
    for f in flags:
        if not f:
            break
        n += 1
    return n


# ---- policies (src/specdec/policies/policies.py) -------------------------------------------
def longest_prefix_accept(draft_ids, base_ids, base_logits=None) -> int:
    """:156-197 — compare the draft ids with argmax(base_logits) when logits exist, else with
    the base ids; count the leading matches."""
    d = np.asarray(draft_ids).reshape(-1)
    ref = np.asarray(base_logits.argmax(-1)).reshape(-1) if base_logits is not None else np.asarray(base_ids).reshape(-1)
    n = min(len(d), len(ref))
    return _leading_true((d[:n] == ref[:n]).tolist())


def conf_threshold_accept(draft_ids, draft_logits, tau: float) -> int:
    """:213-270 — leading positions whose max softmax probability (of the DRAFT) is >= tau."""
    p = torch.softmax(draft_logits.float(), -1).amax(-1).reshape(-1)
    return _leading_true((p[: np.asarray(draft_ids).size] >= tau).tolist())


def topk_agree_accept(draft_ids, base_logits, k: int) -> int:
    """:272-329 — leading positions whose draft token lies in the base's top-k."""
    ids = torch.as_tensor(np.asarray(draft_ids)).reshape(-1)
    top = torch.topk(base_logits.reshape(-1, base_logits.shape[-1])[: ids.numel()], k, -1).indices
    return _leading_true((top == ids.unsqueeze(-1)).any(-1).tolist())


def typical_accept(draft_ids, base_logits, p: float) -> int:
    """:331-396 — leading positions where the base's probability of the draft token is >= p."""
    ids = torch.as_tensor(np.asarray(draft_ids)).reshape(-1).long()
    probs = torch.softmax(base_logits.reshape(-1, base_logits.shape[-1])[: ids.numel()].float(), -1)
    return _leading_true((probs.gather(-1, ids.unsqueeze(-1)).squeeze(-1) >= p).tolist())


def rejection_accept(draft_ids, draft_logits, base_logits, uniforms, temperature: float = 1.0):
    """Speculative-sampling acceptance (not in the reference: the product's opt-in `rejection` policy, restated in
    numpy float64): accept d_i while u_i < p_i(d_i) / q_i(d_i); -> (accepted_len, distribution of the next token:
    normalise(max(0, p - q)) at the first rejection, p_K after a full acceptance, None if base_logits has no K-th row)."""
    d = np.asarray(draft_ids).reshape(-1)
    n = d.size
    t = temperature if temperature > 0 else 1.0

    def sm(x):
        x = np.asarray(x, dtype=np.float64) / t
        e = np.exp(x - x.max(-1, keepdims=True))
        return e / e.sum(-1, keepdims=True)

    bl = np.asarray(base_logits, dtype=np.float64).reshape(-1, np.asarray(base_logits).shape[-1])
    p, q = sm(bl[:n]), sm(np.asarray(draft_logits, dtype=np.float64).reshape(-1, bl.shape[-1])[:n])
    a = 0
    for i in range(n):
        if float(uniforms[i]) < p[i, d[i]] / q[i, d[i]]:
            a += 1
        else:
            break
    if a < n:
        r = np.maximum(p[a] - q[a], 0.0)
        nxt = r / r.sum() if r.sum() > 0 else p[a]
    else:
        nxt = sm(bl[n]) if bl.shape[0] > n else None
    return a, nxt


# ---- adaptive K (src/specdec/policies/controllers.py:63-141) -------------------------------------
class AdaptiveKOracle:
    def __init__(self, initial_k=4, min_k=1, max_k=8, step_size=1, window_size=32, target_acceptance_rate=0.7):
        self.k, self.lo, self.hi, self.step, self.win, self.target = initial_k, min_k, max_k, step_size, window_size, target_acceptance_rate
        self.hist: List[float] = []

    def get_k(self, ctx) -> int:
        if "acceptance_rate" in ctx:
            self.hist = (self.hist + [ctx["acceptance_rate"]])[-self.win:]
        if len(self.hist) >= 4:
            recent = sum(self.hist[-4:]) / 4
            if recent > self.target + 0.1:
                self.k = min(self.k + self.step, self.hi)
            elif recent < self.target - 0.1:
                self.k = max(self.k - self.step, self.lo)
        return self.k

    def recent(self) -> Optional[float]:
        return sum(self.hist[-4:]) / 4 if len(self.hist) >= 4 else None


# ---- greedy bonus token (src/specdec/core/pipeline.py:48-147, do_sample=False) ----------------------
def bonus_token_greedy(logits: torch.Tensor, temperature: float, top_p: Optional[float], top_k: Optional[int], vocab: int) -> int:
    """T-scale, top-k mask, top-p (nucleus) mask, argmax, clamp. Under greedy decoding the filters
    can only remove non-maximal entries, but they are restated in full."""
    x = logits.clone().float().reshape(-1)
    if temperature > 0 and temperature != 1.0:
        x = x / temperature
    if top_k is not None and top_k > 0:
        kth = torch.topk(x, min(top_k, x.numel())).values[-1]
        keep = torch.zeros_like(x, dtype=torch.bool)
        keep[torch.topk(x, min(top_k, x.numel())).indices] = True
        x = torch.where(keep, x, torch.full_like(x, float("-inf")))
        del kth
    if top_p is not None and top_p < 1.0:
        srt, idx = torch.sort(x, descending=True)
        cum = torch.cumsum(torch.softmax(srt, -1), -1)
        drop = cum > top_p
        drop[0] = False
        mask = torch.zeros_like(drop)
        mask[idx] = drop
        x = x.masked_fill(mask, float("-inf"))
    return max(0, min(int(x.argmax()), vocab - 1))


# ---- sequences / token validation (sequence_utils.py:15-184, token_validation.py:15-78) -------------------
def pad_right(seqs: Sequence[Sequence[int]], pad: int):
    n = max((len(s) for s in seqs), default=0)
    batch = [list(s) + [pad] * (n - len(s)) for s in seqs]
    mask = [[1] * len(s) + [0] * (n - len(s)) for s in seqs]
    pos = [list(range(len(s))) + [0] * (n - len(s)) for s in seqs]
    return batch, mask, [len(s) for s in seqs], pos


def clamp_tokens(ids, vocab: int):
    return np.clip(np.asarray(ids), 0, vocab - 1).tolist()
