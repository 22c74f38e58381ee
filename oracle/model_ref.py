"""CPU restatement of the decoder forward the reference gets from HF transformers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference never implements a transformer: `HFWrapper` calls
`AutoModelForCausalLM` (src/specdec/models/hf_wrappers.py:87-141, forward at
:417/:478/:684/:772; `transformers>=4.30,<5`, pyproject.toml:41). This file restates
the published algorithms of `LlamaForCausalLM` and `GPT2LMHeadModel`
(transformers/models/llama/modeling_llama.py, models/gpt2/modeling_gpt2.py):
RMSNorm / LayerNorm, rotary embedding with the "llama3" frequency scaling, grouped-
query causal attention, SwiGLU / gelu_new MLP, (un)tied lm_head. It is pinned against
the real transformers implementation on tiny random models by
tests/golden/make_golden.py -> tests/golden/hf_*.npz (tests/test_oracle_model.py).

Two precisions:
  * "fp32"  — everything in fp32: the mode pinned against HF fp32.
  * "bf16"  — bf16 weights and bf16 activations BETWEEN ops with fp32 accumulation
              inside each op, rounding at the points where the gfx950 kernels round
              (csrc/gemv.hip epilogues, csrc/attention.hip). This is the parity
              oracle of the GPU path and the model of the `cpu_baseline`.
"""

from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

ARCH_LLAMA, ARCH_GPT2 = 0, 1


def _bf(x: torch.Tensor) -> torch.Tensor:
    """Round to bf16 and come back to fp32 (round-to-nearest-even, as v_cvt_pk_bf16_f32)."""
    return x.to(torch.bfloat16).to(torch.float32)


def _inv_freq(cfg) -> torch.Tensor:
    # modeling_rope_utils._compute_default_rope_parameters / _compute_llama3_parameters
    D = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    sc = cfg.rope_scaling
    if sc and sc.get("rope_type", sc.get("type")) == "llama3":
        factor, low, high = float(sc["factor"]), float(sc["low_freq_factor"]), float(sc["high_freq_factor"])
        old = float(sc["original_max_position_embeddings"])
        wavelen = 2 * math.pi / inv
        out = torch.where(wavelen > old / low, inv / factor, inv)
        smooth = (old / wavelen - low) / (high - low)
        mid = (1 - smooth) * out / factor + smooth * out
        is_mid = ~(wavelen < old / high) & ~(wavelen > old / low)
        inv = torch.where(is_mid, mid, out)
    return inv.float()


def hf_sampling_probs(logits: torch.Tensor, temperature: float, top_k: Optional[int] = 50, top_p: Optional[float] = None) -> torch.Tensor:
    """The distribution transformers' `generate(do_sample=True, temperature=T)` draws a token from (the library the reference
    calls at hf_wrappers.py:232; pinned by tests/golden/pipeline_sampled_golden.json, generated with transformers 5.15):
    TemperatureLogitsWarper (scores / T) -> TopKLogitsWarper (top_k, the library's sampling default 50 when the model's
    generation config leaves it unset: everything below the k-th largest score to -inf, ties with it kept) -> TopPLogitsWarper
    (only for top_p < 1: ascending sort, drop the tokens whose cumulative probability is <= 1 - top_p, keep at least one) ->
    softmax. logits [B][V]."""
    s = logits.float() / temperature
    if top_k:
        kth = torch.topk(s, min(int(top_k), s.shape[-1]))[0][..., -1, None]
        s = s.masked_fill(s < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_idx = torch.sort(s, descending=False)
        remove = sorted_logits.softmax(dim=-1).cumsum(dim=-1) <= (1.0 - top_p)
        remove[..., -1:] = False
        s = s.masked_fill(remove.scatter(1, sorted_idx, remove), float("-inf"))
    return torch.softmax(s, dim=-1)


class OracleLM:
    """Decoder forward on CPU with an explicit KV cache (list of per-layer (k, v))."""

    def __init__(self, weights, precision: str = "bf16", threads: Optional[int] = None):
        assert precision in ("fp32", "bf16")
        self.w = weights
        self.cfg = weights.config
        self.round = precision == "bf16"
        if threads:
            torch.set_num_threads(threads)
        self._f32 = {}
        if self.cfg.arch == ARCH_LLAMA:
            inv = _inv_freq(self.cfg)
            ang = torch.outer(torch.arange(self.cfg.max_pos, dtype=torch.float32), inv)
            self.cos, self.sin = ang.cos(), ang.sin()  # [P][D/2]

    # weights as fp32 matrices (values are exactly the stored bf16/fp32 values)
    def _m(self, key, t: torch.Tensor) -> torch.Tensor:
        got = self._f32.get(key)
        if got is None:
            got = t.detach().to("cpu", torch.float32)
            self._f32[key] = got
        return got

    def _r(self, x):
        return _bf(x) if self.round else x

    # ---- norms --------------------------------------------------------------------
    def _rmsnorm(self, x, w):
        # LlamaRMSNorm.forward: weight * (x * rsqrt(mean(x^2)+eps)).to(input_dtype)
        var = x.pow(2).mean(-1, keepdim=True)
        xn = self._r(x * torch.rsqrt(var + self.cfg.norm_eps))
        return self._r(xn * w)

    def _layernorm(self, x, w, b):
        mean = x.mean(-1, keepdim=True)
        var = (x * x).mean(-1, keepdim=True) - mean * mean
        var = var.clamp_min(0.0)
        return self._r((x - mean) * torch.rsqrt(var + self.cfg.norm_eps) * w + b)

    def _norm(self, x, key, w, b):
        if self.cfg.arch == ARCH_LLAMA:
            return self._rmsnorm(x, self._m(key + ".w", w))
        return self._layernorm(x, self._m(key + ".w", w), self._m(key + ".b", b))

    # ---- one forward ---------------------------------------------------------------
    def forward(self, tokens: torch.Tensor, past: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None,
                positions: Optional[torch.Tensor] = None, need_logits: bool = True):
        """tokens int64 [B][L] (new tokens only when `past` is given).
        Returns (logits fp32 [B][L][V] or None, new_past)."""
        cfg, W = self.cfg, self.w
        B, L = tokens.shape
        Hq, Hkv, D, d = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.d_model
        G = Hq // Hkv
        P0 = 0 if not past else past[0][0].shape[2]
        if positions is None:
            positions = torch.arange(P0, P0 + L).unsqueeze(0).expand(B, L)
        tok = tokens.clamp(0, cfg.vocab - 1)  # validate_and_clamp_tokens, token_validation.py:15-78
        x = self._m("tok_emb", W.tok_emb)[tok]  # [B][L][d]
        if cfg.arch == ARCH_GPT2:
            x = self._r(x + self._m("pos_emb", W.pos_emb)[positions.clamp(0, cfg.max_pos - 1)])
        new_past = []
        scale = 1.0 / math.sqrt(D)
        for li, lw in enumerate(W.layers):
            k0 = f"l{li}."
            xn = self._norm(x, k0 + "n1", lw.attn_norm_w, lw.attn_norm_b)
            qkv = xn @ self._m(k0 + "wqkv", lw.wqkv).t()
            if lw.bqkv is not None:
                qkv = qkv + self._m(k0 + "bqkv", lw.bqkv)
            q = qkv[..., : Hq * D].view(B, L, Hq, D)
            k = qkv[..., Hq * D : (Hq + Hkv) * D].view(B, L, Hkv, D)
            v = qkv[..., (Hq + Hkv) * D :].view(B, L, Hkv, D)
            if cfg.arch == ARCH_LLAMA:
                # apply_rotary_pos_emb with rotate_half: pairs (i, i + D/2)
                cos = self.cos[positions].unsqueeze(2)  # [B][L][1][D/2]
                sin = self.sin[positions].unsqueeze(2)

                def rope(t):
                    a, b = t[..., : D // 2], t[..., D // 2 :]
                    return torch.cat([a * cos - b * sin, b * cos + a * sin], -1)

                q, k = rope(q), rope(k)
            q, k, v = self._r(q), self._r(k), self._r(v)
            k = k.permute(0, 2, 1, 3)  # [B][Hkv][L][D]
            v = v.permute(0, 2, 1, 3)
            if past:
                k = torch.cat([past[li][0], k], 2)
                v = torch.cat([past[li][1], v], 2)
            new_past.append((k, v))
            S = k.shape[2]
            qh = q.permute(0, 2, 1, 3).reshape(B, Hkv, G, L, D) * scale
            att = torch.einsum("bhgld,bhsd->bhgls", qh, k)
            key_pos = torch.arange(S).view(1, 1, 1, 1, S)
            q_pos = (P0 + torch.arange(L)).view(1, 1, 1, L, 1)
            att = att.masked_fill(key_pos > q_pos, float("-inf"))
            att = torch.softmax(att, -1)
            o = torch.einsum("bhgls,bhsd->bhgld", att, v)  # [B][Hkv][G][L][D]
            o = self._r(o.permute(0, 3, 1, 2, 4).reshape(B, L, Hq * D))
            y = o @ self._m(k0 + "wo", lw.wo).t()
            if lw.bo is not None:
                y = y + self._m(k0 + "bo", lw.bo)
            x = self._r(x + y)
            xn = self._norm(x, k0 + "n2", lw.mlp_norm_w, lw.mlp_norm_b)
            up = xn @ self._m(k0 + "wup", lw.w_up).t()
            if cfg.arch == ARCH_LLAMA:
                g, u = up[..., : cfg.d_ff], up[..., cfg.d_ff :]
                act = self._r(g / (1.0 + torch.exp(-g)) * u)
            else:
                if lw.b_up is not None:
                    up = up + self._m(k0 + "bup", lw.b_up)
                act = self._r(0.5 * up * (1.0 + torch.tanh(0.7978845608028654 * (up + 0.044715 * up * up * up))))
            y = act @ self._m(k0 + "wdown", lw.w_down).t()
            if lw.b_down is not None:
                y = y + self._m(k0 + "bdown", lw.b_down)
            x = self._r(x + y)
        self.last_hidden = x            # residual stream before the final norm [B][L][d] (Medusa heads read it)
        if not need_logits:
            return None, new_past
        xn = self._norm(x, "nf", W.final_norm_w, W.final_norm_b)
        logits = self._r(xn @ self._m("lm_head", W.lm_head).t())
        return logits, new_past

    def head_tokens(self, hidden_row: torch.Tensor, heads: torch.Tensor) -> List[int]:
        """argmax head_i(final_norm(h)) for every head: bf16-rounded logits as the lm_head's."""
        xn = self._norm(hidden_row.view(1, 1, -1), "nf", self.w.final_norm_w, self.w.final_norm_b)
        out = []
        for i in range(heads.shape[0]):
            key = f"medusa{i}"
            out.append(int(self._r(xn @ self._m(key, heads[i]).t()).view(-1).argmax()))
        return out

    # ---- greedy generation, the semantics of HFWrapper._generate_tokens_async -------
    def generate_tokens(self, input_ids: torch.Tensor, max_new_tokens: int, reprefill: bool = False,
                        do_sample: bool = False, temperature: float = 1.0, top_k: Optional[int] = 50, top_p: Optional[float] = None,
                        eos_token_id: Optional[int] = None):
        """hf_wrappers.py:272-627 under greedy decoding: `max_new_tokens` forwards, each
        taking argmax of the last position's logits; returns (ids [B][k], logits [B][k][V]).
        reprefill=True re-feeds the whole prefix every token, as the reference does with
        KV append off (pipeline.py:1838); the result is the same, the cost is not.
        do_sample=True: with KV append off the reference hands the call to transformers' `generate` (hf_wrappers.py:208-232), whose
        sampling is restated in hf_sampling_probs; one torch.multinomial draw on torch's GLOBAL generator per position. `generate`
        ends a (one-row) call at the first eos_token_id it produces: fewer than max_new_tokens ids come back, and no further draws
        are made."""
        cur = input_ids.clone()
        ids, lg = [], []
        past = None
        for _ in range(max_new_tokens):
            if reprefill or past is None:
                logits, past = self.forward(cur)
                if reprefill:
                    past = None
            else:
                logits, past = self.forward(cur[:, -1:], past)
            last = logits[:, -1, :]
            if do_sample:
                nxt = torch.multinomial(hf_sampling_probs(last, temperature, top_k, top_p), num_samples=1)[:, 0]
            else:
                nxt = torch.argmax(last, dim=-1)
            ids.append(nxt)
            lg.append(last)
            cur = torch.cat([cur, nxt.unsqueeze(1)], 1)
            if eos_token_id is not None and bool((nxt == eos_token_id).all()):
                break
        return torch.stack(ids, 1), torch.stack(lg, 1)
