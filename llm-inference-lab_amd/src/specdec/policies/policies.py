"""Acceptance policies (reference: src/specdec/policies/policies.py:34-425).

`accept_tokens(proposed, base, proposed_logits, base_logits) -> (accepted_len, info)`.

`LongestPrefixPolicy` is the hot one: with target logits on the GPU it calls the HIP
`verify_prefix` op through the kernel registry (argmax over V + ballot prefix scan,
csrc/verify_prefix.hip) and reads back one int — the reference dispatches the same way
(policies.py:121-142) but silently falls back to PyTorch when the kernel throws; here a
kernel failure propagates. With token ids only (no logits) the comparison is host
integer logic on K ids, as in the reference (:183-197). CPU logits are refused: this
build has no CPU compute path.

The three logit-threshold policies are opt-in (`create_policy`): each needs one
per-position scalar (max softmax, top-k membership, p(target = draft token)), computed
with device tensor ops on [B,K,V] and then scanned for the first failing position.
"""

from __future__ import annotations

import logging
from abc import ABC, abstractmethod
from typing import Any, Dict, Optional, Tuple

import torch

from kernels import get_kernel_info, get_verify_prefix

logger = logging.getLogger(__name__)


class AcceptancePolicy(ABC):
    @abstractmethod
    def accept_tokens(self, proposed_tokens: torch.Tensor, base_tokens: torch.Tensor,
                      proposed_logits: Optional[torch.Tensor] = None, base_logits: Optional[torch.Tensor] = None,
                      **kwargs: Any) -> Tuple[int, Dict[str, Any]]: ...

    @abstractmethod
    def get_info(self) -> Dict[str, Any]: ...


def _first_false(flags) -> int:
    """Length of the leading run of True in a 1-D sequence of bools."""
    n = 0
    for f in flags:
        if not bool(f):
            break
        n += 1
    return n


class LongestPrefixPolicy(AcceptancePolicy):
    def __init__(self):
        self.name = "longest_prefix"
        self.kernels_available = True
        self.kernel_info = get_kernel_info()
        logger.info("Using verify backend: %s", self.kernel_info.get("verify_backend", "unknown"))

    def accept_tokens(self, proposed_tokens, base_tokens, proposed_logits=None, base_logits=None, **kwargs):
        k = proposed_tokens.shape[1]
        if base_logits is not None:
            if base_logits.device.type != "cuda":
                raise RuntimeError(
                    "LongestPrefixPolicy: target logits are on the CPU; this build verifies on the GPU only "
                    "(pass token ids without logits for a host-side id comparison)")
            verify = get_verify_prefix("cuda")
            if verify is None:
                raise RuntimeError("verify_prefix is not registered for device 'cuda'")
            ids = proposed_tokens.to(base_logits.device)
            accept_len, _mask = verify(base_logits[:, :k, :], ids)
            accepted_len = int(accept_len[0].item())  # batch of one per call, as the reference (:131)
            backend = self.kernel_info["verify_backend"]
        else:
            # ids only: longest common prefix of two short id rows (host integers)
            n = min(k, base_tokens.shape[1])
            a = proposed_tokens[:, :n].long().cpu()
            b = base_tokens[:, :n].long().cpu()
            accepted_len = _first_false((a == b).all(dim=0).tolist())
            backend = "host-ids"
        return accepted_len, {
            "policy": self.name, "accepted_len": accepted_len, "proposed_len": k,
            "base_len": base_tokens.shape[1], "verify_backend": backend,
        }

    def get_info(self) -> Dict[str, Any]:
        return {"policy": self.name}


class ConfidenceThresholdPolicy(AcceptancePolicy):
    """Accept while the draft's own max softmax probability stays >= tau (:213-270)."""

    def __init__(self, tau: float = 0.5):
        self.tau = tau
        self.name = "conf_threshold"

    def accept_tokens(self, proposed_tokens, base_tokens, proposed_logits=None, base_logits=None, **kwargs):
        if proposed_logits is None:
            return LongestPrefixPolicy().accept_tokens(proposed_tokens, base_tokens)
        conf = torch.softmax(proposed_logits.float(), dim=-1).amax(dim=-1)[0]  # [K]
        n = proposed_tokens.shape[1]
        accepted_len = _first_false((conf[:n] >= self.tau).tolist())
        return accepted_len, {
            "policy": self.name, "tau": self.tau, "accepted_len": accepted_len, "proposed_len": n,
            "min_confidence": float(conf[:accepted_len].min()) if accepted_len > 0 else 0.0,
        }

    def get_info(self):
        return {"policy": self.name, "tau": self.tau}


class TopKAgreementPolicy(AcceptancePolicy):
    """Accept while the draft token is inside the target's top-k (:272-329)."""

    def __init__(self, k: int = 5):
        self.k = k
        self.name = "topk_agree"

    def accept_tokens(self, proposed_tokens, base_tokens, proposed_logits=None, base_logits=None, **kwargs):
        if proposed_logits is None or base_logits is None:
            return LongestPrefixPolicy().accept_tokens(proposed_tokens, base_tokens)
        n = proposed_tokens.shape[1]
        topk = torch.topk(base_logits[0, :n], self.k, dim=-1).indices  # [K, k]
        hit = (topk == proposed_tokens[0, :n].to(topk.device).unsqueeze(-1)).any(dim=-1)
        accepted_len = _first_false(hit.tolist())
        return accepted_len, {"policy": self.name, "k": self.k, "accepted_len": accepted_len, "proposed_len": n}

    def get_info(self):
        return {"policy": self.name, "k": self.k}


class TypicalAcceptancePolicy(AcceptancePolicy):
    """Accept while the target's probability of the draft token stays >= p (:331-396)."""

    def __init__(self, p: float = 0.9):
        self.p = p
        self.name = "typical"

    def accept_tokens(self, proposed_tokens, base_tokens, proposed_logits=None, base_logits=None, **kwargs):
        if proposed_logits is None or base_logits is None:
            return LongestPrefixPolicy().accept_tokens(proposed_tokens, base_tokens)
        n = proposed_tokens.shape[1]
        probs = torch.softmax(base_logits[0, :n].float(), dim=-1)  # [K, V]
        idx = proposed_tokens[0, :n].to(probs.device).long()
        p_tok = probs.gather(-1, idx.unsqueeze(-1)).squeeze(-1)
        accepted_len = _first_false((p_tok >= self.p).tolist())
        if accepted_len > 0:
            # the reference reports min over the [a x a] cross-indexed block (:384-392)
            min_p = float(probs[:accepted_len][:, idx[:accepted_len]].min())
        else:
            min_p = 0.0
        return accepted_len, {"policy": self.name, "p": self.p, "accepted_len": accepted_len,
                              "proposed_len": n, "min_probability": min_p}

    def get_info(self):
        return {"policy": self.name, "p": self.p}


class RejectionSamplingPolicy(AcceptancePolicy):
    """Speculative-sampling acceptance (opt-in; NOT in the reference, whose four policies are deterministic — SURVEY
    section 8 f1): draft token d_i, drawn from the draft distribution q_i, is accepted with probability
    min(1, p_i(d_i) / q_i(d_i)) where p_i is the target distribution at that position GIVEN the draft tokens before it (the
    K+1-token parallel verify pass provides exactly that); the first rejected position is re-drawn from
    normalise(max(0, p - q)), a fully accepted step draws a bonus token from p_K. The emitted sequence is then distributed
    exactly as the target's own sampling (Leviathan et al. 2023, Chen et al. 2023).

    `temperature` shapes both distributions (softmax(logits / T)); the uniforms come from a private generator seeded with
    `seed` (or are passed in: `uniforms=`), so a run is reproducible and a CPU restatement (oracle/hostlogic_ref.py:
    rejection_accept) replays it. All arithmetic is float64 on the tensors' device."""

    def __init__(self, temperature: float = 1.0, seed: int = 0):
        self.temperature = float(temperature)
        self.seed = int(seed)
        self.name = "rejection"
        self._gen = torch.Generator().manual_seed(self.seed)

    def reseed(self, seed: Optional[int] = None) -> None:
        self._gen = torch.Generator().manual_seed(self.seed if seed is None else int(seed))

    def uniforms(self, n: int) -> torch.Tensor:
        return torch.rand(n, generator=self._gen, dtype=torch.float64)

    def distributions(self, logits: torch.Tensor) -> torch.Tensor:
        t = self.temperature if self.temperature > 0 else 1.0
        return torch.softmax(logits.double() / t, dim=-1)

    def accept_tokens(self, proposed_tokens, base_tokens, proposed_logits=None, base_logits=None, **kwargs):
        if proposed_logits is None or base_logits is None:
            raise ValueError("rejection sampling needs the draft and the target logits")
        n = proposed_tokens.shape[1]
        u = kwargs.get("uniforms")
        u = self.uniforms(n) if u is None else torch.as_tensor(u, dtype=torch.float64)
        p = self.distributions(base_logits[0, :n])                      # [K, V] target, conditioned on the draft prefix
        q = self.distributions(proposed_logits[0, :n])
        idx = proposed_tokens[0, :n].to(p.device).long().unsqueeze(-1)
        ratio = (p.gather(-1, idx) / q.gather(-1, idx)).squeeze(-1)     # q(d) > 0: d was drawn from q
        accepted_len = _first_false((u.to(ratio.device) < ratio).tolist())
        if accepted_len < n:
            resid = (p[accepted_len] - q[accepted_len]).clamp_min(0.0)
            total = resid.sum()
            nxt = resid / total if float(total) > 0 else p[accepted_len]
        else:                                                            # bonus: the target's distribution after all K drafts
            nxt = self.distributions(base_logits[0, n]) if base_logits.shape[1] > n else None
        return accepted_len, {"policy": self.name, "accepted_len": accepted_len, "proposed_len": n, "temperature": self.temperature,
                              "ratios": ratio.tolist(), "next_distribution": nxt}

    @staticmethod
    def draw(dist: torch.Tensor, u: float) -> int:
        """Inverse-CDF draw in index order."""
        c = torch.cumsum(dist.double(), dim=-1)
        i = int(torch.searchsorted(c, torch.tensor(float(u) * float(c[-1]), dtype=torch.float64, device=c.device), right=True))
        return min(i, dist.numel() - 1)

    def get_info(self):
        return {"policy": self.name, "temperature": self.temperature, "seed": self.seed}


def create_policy(policy_name: str, **kwargs: Any) -> AcceptancePolicy:
    makers = {
        "longest_prefix": LongestPrefixPolicy,
        "conf_threshold": lambda: ConfidenceThresholdPolicy(kwargs.get("tau", 0.5)),
        "topk_agree": lambda: TopKAgreementPolicy(kwargs.get("k", 5)),
        "typical": lambda: TypicalAcceptancePolicy(kwargs.get("p", 0.9)),
        "rejection": lambda: RejectionSamplingPolicy(kwargs.get("temperature", 1.0), kwargs.get("seed", 0)),
    }
    if policy_name not in makers:
        raise ValueError(f"Unknown policy: {policy_name}. Available: {list(makers.keys())}")
    return makers[policy_name]()
