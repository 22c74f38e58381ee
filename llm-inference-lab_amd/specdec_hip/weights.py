"""Model configuration + weight containers for the HIP decoder forward.

Weights live in torch device tensors (PyTorch-ROCm is the allocator); the C-ABI
library borrows their pointers. Layout is what csrc/gemv.hip streams: every linear
weight is [out_features][in_features] row-major bf16, q/k/v fused into one matrix
(q rows, k rows, v rows) and gate/up fused (gate rows, then up rows).

Sources of weights:
  * `from_hf_state_dict` — a transformers Llama / GPT-2 state dict (what the
    reference's HFWrapper loads, hf_wrappers.py:80-141), e.g. from a local
    checkpoint directory (`load_checkpoint_dir`);
  * `synthetic_llama` — architecture-exact random init for benchmarks without
    checkpoints, with a draft/target pair construction whose greedy
    acceptance is controlled and whose argmax margins are large.
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

ARCH_LLAMA, ARCH_GPT2 = 0, 1


@dataclass
class ModelConfig:
    arch: int = ARCH_LLAMA
    n_layers: int = 2
    d_model: int = 64
    n_heads: int = 2
    n_kv_heads: int = 2
    head_dim: int = 32
    d_ff: int = 128
    vocab: int = 256
    max_pos: int = 2048
    norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    rope_scaling: Optional[dict] = None  # HF "llama3" scaling dict
    tie_embeddings: bool = True
    eos_token_id: Optional[int] = None
    name: str = "tiny"

    @property
    def n_params_matmul(self) -> int:
        """Weights streamed by one forward (lm_head counted once, embedding gather ignored)."""
        per_layer = (
            (self.n_heads + 2 * self.n_kv_heads) * self.head_dim * self.d_model
            + self.d_model * self.n_heads * self.head_dim
            + (2 if self.arch == ARCH_LLAMA else 1) * self.d_ff * self.d_model
            + self.d_model * self.d_ff
        )
        return self.n_layers * per_layer + self.vocab * self.d_model

    @property
    def kv_bytes_per_token(self) -> int:
        return 2 * self.n_layers * self.n_kv_heads * self.head_dim * 2


# public model-card shapes (re-derived from config.json when a checkpoint is given)
LLAMA_3_2_1B = ModelConfig(
    arch=ARCH_LLAMA, n_layers=16, d_model=2048, n_heads=32, n_kv_heads=8, head_dim=64, d_ff=8192,
    vocab=128256, max_pos=4096, norm_eps=1e-5, rope_theta=500000.0,
    rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                  "original_max_position_embeddings": 8192, "rope_type": "llama3"},
    tie_embeddings=True, eos_token_id=128001, name="llama-3.2-1b",
)
LLAMA_3_2_3B = ModelConfig(
    arch=ARCH_LLAMA, n_layers=28, d_model=3072, n_heads=24, n_kv_heads=8, head_dim=128, d_ff=8192,
    vocab=128256, max_pos=4096, norm_eps=1e-5, rope_theta=500000.0,
    rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                  "original_max_position_embeddings": 8192, "rope_type": "llama3"},
    tie_embeddings=True, eos_token_id=128001, name="llama-3.2-3b",
)
LLAMA_3_8B = ModelConfig(
    arch=ARCH_LLAMA, n_layers=32, d_model=4096, n_heads=32, n_kv_heads=8, head_dim=128, d_ff=14336,
    vocab=128256, max_pos=4096, norm_eps=1e-5, rope_theta=500000.0, rope_scaling=None,
    tie_embeddings=False, eos_token_id=128001, name="llama-3-8b",
)
GPT2_SMALL = ModelConfig(
    arch=ARCH_GPT2, n_layers=12, d_model=768, n_heads=12, n_kv_heads=12, head_dim=64, d_ff=3072,
    vocab=50257, max_pos=1024, norm_eps=1e-5, tie_embeddings=True, eos_token_id=50256, name="gpt2",
)
DISTILGPT2 = ModelConfig(
    arch=ARCH_GPT2, n_layers=6, d_model=768, n_heads=12, n_kv_heads=12, head_dim=64, d_ff=3072,
    vocab=50257, max_pos=1024, norm_eps=1e-5, tie_embeddings=True, eos_token_id=50256, name="distilgpt2",
)


@dataclass
class LayerWeights:
    attn_norm_w: torch.Tensor
    wqkv: torch.Tensor
    wo: torch.Tensor
    mlp_norm_w: torch.Tensor
    w_up: torch.Tensor
    w_down: torch.Tensor
    attn_norm_b: Optional[torch.Tensor] = None
    bqkv: Optional[torch.Tensor] = None
    bo: Optional[torch.Tensor] = None
    mlp_norm_b: Optional[torch.Tensor] = None
    b_up: Optional[torch.Tensor] = None
    b_down: Optional[torch.Tensor] = None


@dataclass
class ModelWeights:
    config: ModelConfig
    tok_emb: torch.Tensor
    lm_head: torch.Tensor
    final_norm_w: torch.Tensor
    layers: List[LayerWeights]
    final_norm_b: Optional[torch.Tensor] = None
    pos_emb: Optional[torch.Tensor] = None
    rope_cos: Optional[torch.Tensor] = None
    rope_sin: Optional[torch.Tensor] = None
    meta: Dict[str, object] = field(default_factory=dict)

    def tensors(self):
        for name in ("tok_emb", "lm_head", "final_norm_w", "final_norm_b", "pos_emb", "rope_cos", "rope_sin"):
            t = getattr(self, name)
            if t is not None:
                yield name, t
        for i, l in enumerate(self.layers):
            for name, t in vars(l).items():
                if t is not None:
                    yield f"layers.{i}.{name}", t

    def to(self, device) -> "ModelWeights":
        def mv(t):
            return None if t is None else t.to(device)

        seen: Dict[int, torch.Tensor] = {}

        def mv_shared(t):  # keep tok_emb / lm_head aliasing
            if t is None:
                return None
            key = t.data_ptr()
            if key not in seen:
                seen[key] = t.to(device)
            return seen[key]

        return ModelWeights(
            config=self.config, tok_emb=mv_shared(self.tok_emb), lm_head=mv_shared(self.lm_head),
            final_norm_w=mv(self.final_norm_w), final_norm_b=mv(self.final_norm_b), pos_emb=mv(self.pos_emb),
            rope_cos=mv(self.rope_cos), rope_sin=mv(self.rope_sin),
            layers=[LayerWeights(**{k: mv(v) for k, v in vars(l).items()}) for l in self.layers],
            meta=dict(self.meta),
        )

    def matmul_bytes(self) -> int:
        return self.config.n_params_matmul * 2


# ------------------------------------------------------------------------------- RoPE
def rope_inv_freq(cfg: ModelConfig) -> torch.Tensor:
    """inv_freq[D/2] in fp32; HF default rope + the "llama3" frequency scaling
    (transformers modeling_rope_utils: _compute_default_rope_parameters /
    _compute_llama3_parameters)."""
    D = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    sc = cfg.rope_scaling
    if sc and sc.get("rope_type", sc.get("type")) == "llama3":
        factor = float(sc["factor"])
        low, high = float(sc["low_freq_factor"]), float(sc["high_freq_factor"])
        old_len = float(sc["original_max_position_embeddings"])
        low_wavelen, high_wavelen = old_len / low, old_len / high
        wavelen = 2 * math.pi / inv
        scaled = torch.where(wavelen > low_wavelen, inv / factor, inv)
        smooth = (old_len / wavelen - low) / (high - low)
        smoothed = (1 - smooth) * scaled / factor + smooth * scaled
        is_medium = ~(wavelen < high_wavelen) & ~(wavelen > low_wavelen)
        inv = torch.where(is_medium, smoothed, scaled)
    return inv.float()


def rope_tables(cfg: ModelConfig, device="cpu"):
    inv = rope_inv_freq(cfg)
    pos = torch.arange(cfg.max_pos, dtype=torch.float32)
    ang = torch.outer(pos, inv)  # [max_pos][D/2], fp32 as HF computes it
    return ang.cos().contiguous().to(device), ang.sin().contiguous().to(device)


# ---------------------------------------------------------------- HF state dict -> ours
def config_from_hf(hf_cfg) -> ModelConfig:
    mt = getattr(hf_cfg, "model_type", "llama")
    if mt == "gpt2":
        d = hf_cfg.n_embd
        return ModelConfig(
            arch=ARCH_GPT2, n_layers=hf_cfg.n_layer, d_model=d, n_heads=hf_cfg.n_head,
            n_kv_heads=hf_cfg.n_head, head_dim=d // hf_cfg.n_head,
            d_ff=hf_cfg.n_inner if hf_cfg.n_inner is not None else 4 * d, vocab=hf_cfg.vocab_size,
            max_pos=hf_cfg.n_positions, norm_eps=hf_cfg.layer_norm_epsilon, tie_embeddings=True,
            eos_token_id=hf_cfg.eos_token_id, name="gpt2-like",
        )
    head_dim = getattr(hf_cfg, "head_dim", None) or hf_cfg.hidden_size // hf_cfg.num_attention_heads
    rs = getattr(hf_cfg, "rope_scaling", None)
    theta = getattr(hf_cfg, "rope_theta", None)
    rp = getattr(hf_cfg, "rope_parameters", None)  # transformers >= 5 spelling
    if rp:
        theta = rp.get("rope_theta", theta)
        if rp.get("rope_type", "default") != "default":
            rs = dict(rp)
    if rs and rs.get("rope_type", rs.get("type", "default")) == "default":
        rs = None   # transformers >= 5 reports the plain rope as a "default" scaling dict: same frequencies, one spelling here
    eos = hf_cfg.eos_token_id
    if isinstance(eos, (list, tuple)):
        eos = eos[0]
    return ModelConfig(
        arch=ARCH_LLAMA, n_layers=hf_cfg.num_hidden_layers, d_model=hf_cfg.hidden_size,
        n_heads=hf_cfg.num_attention_heads, n_kv_heads=hf_cfg.num_key_value_heads, head_dim=head_dim,
        d_ff=hf_cfg.intermediate_size, vocab=hf_cfg.vocab_size,
        max_pos=min(int(hf_cfg.max_position_embeddings), 8192), norm_eps=hf_cfg.rms_norm_eps,
        rope_theta=float(theta or 10000.0), rope_scaling=rs,
        tie_embeddings=bool(getattr(hf_cfg, "tie_word_embeddings", False)), eos_token_id=eos, name="llama-like",
    )


def from_hf_state_dict(cfg: ModelConfig, sd: Dict[str, torch.Tensor], dtype=torch.bfloat16, device="cpu") -> ModelWeights:
    """Re-pack an HF Llama / GPT-2 state dict into the fused row-major layout."""

    def g(name):
        return sd[name].detach().to(dtype=dtype, device=device).contiguous()

    layers: List[LayerWeights] = []
    if cfg.arch == ARCH_LLAMA:
        p = "model." if any(k.startswith("model.") for k in sd) else ""
        for i in range(cfg.n_layers):
            b = f"{p}layers.{i}."
            wqkv = torch.cat([g(b + "self_attn.q_proj.weight"), g(b + "self_attn.k_proj.weight"), g(b + "self_attn.v_proj.weight")], 0).contiguous()
            w_up = torch.cat([g(b + "mlp.gate_proj.weight"), g(b + "mlp.up_proj.weight")], 0).contiguous()
            layers.append(LayerWeights(
                attn_norm_w=g(b + "input_layernorm.weight"), wqkv=wqkv, wo=g(b + "self_attn.o_proj.weight"),
                mlp_norm_w=g(b + "post_attention_layernorm.weight"), w_up=w_up, w_down=g(b + "mlp.down_proj.weight"),
            ))
        tok = g(p + "embed_tokens.weight")
        head = tok if (cfg.tie_embeddings or "lm_head.weight" not in sd) else g("lm_head.weight")
        cos, sin = rope_tables(cfg, device)
        return ModelWeights(cfg, tok, head, g(p + "norm.weight"), layers, rope_cos=cos, rope_sin=sin)
    # GPT-2: Conv1D stores [in][out]; transpose to [out][in]
    p = "transformer." if any(k.startswith("transformer.") for k in sd) else ""

    def gt(name):
        return sd[name].detach().t().to(dtype=dtype, device=device).contiguous()

    for i in range(cfg.n_layers):
        b = f"{p}h.{i}."
        layers.append(LayerWeights(
            attn_norm_w=g(b + "ln_1.weight"), attn_norm_b=g(b + "ln_1.bias"),
            wqkv=gt(b + "attn.c_attn.weight"), bqkv=g(b + "attn.c_attn.bias"),
            wo=gt(b + "attn.c_proj.weight"), bo=g(b + "attn.c_proj.bias"),
            mlp_norm_w=g(b + "ln_2.weight"), mlp_norm_b=g(b + "ln_2.bias"),
            w_up=gt(b + "mlp.c_fc.weight"), b_up=g(b + "mlp.c_fc.bias"),
            w_down=gt(b + "mlp.c_proj.weight"), b_down=g(b + "mlp.c_proj.bias"),
        ))
    tok = g(p + "wte.weight")
    return ModelWeights(cfg, tok, tok, g(p + "ln_f.weight"), layers, final_norm_b=g(p + "ln_f.bias"), pos_emb=g(p + "wpe.weight"))


def load_checkpoint_dir(path: str, device="cpu", dtype=torch.bfloat16) -> ModelWeights:
    """Load a local HF checkpoint directory (config.json + *.safetensors). Local paths
    only: nothing is fetched by name (the reference's from_pretrained-by-name,
    hf_wrappers.py:87,115, needs the network)."""
    import json

    from safetensors import safe_open

    with open(os.path.join(path, "config.json")) as f:
        raw = json.load(f)

    class _C:  # attribute view of config.json
        def __init__(self, d):
            self.__dict__.update(d)

        def __getattr__(self, k):
            return None

    cfg = config_from_hf(_C(raw))
    sd: Dict[str, torch.Tensor] = {}
    for fn in sorted(os.listdir(path)):
        if fn.endswith(".safetensors"):
            with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as f:
                for k in f.keys():
                    sd[k] = f.get_tensor(k)
    if not sd:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    return from_hf_state_dict(cfg, sd, dtype=dtype, device=device)


# ---------------------------------------------------------------------- synthetic init
def _randn(shape, std, gen, device, dtype):
    # generate in chunks on the target device to bound peak memory for 128256 x 4096
    out = torch.empty(shape, dtype=dtype, device=device)
    rows = shape[0]
    step = max(1, (1 << 24) // max(1, math.prod(shape[1:])))
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        blk = torch.randn((r1 - r0, *shape[1:]), generator=gen, device=device, dtype=torch.float32) * std
        out[r0:r1] = blk.to(dtype)
    return out


def synthetic_llama(cfg: ModelConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16,
                    layer_gain: float = 0.05, successor_mult: int = 7919, successor_add: int = 1,
                    embed_from: Optional[ModelWeights] = None, flip_fraction: float = 0.0,
                    flip_seed: int = 99) -> ModelWeights:
    """Architecture-exact random weights for a Llama-shaped model.

    Construction (documented in DESIGN.md §synthetic weights): the output table
    E_out ~ N(0,1) and the INPUT table is a row permutation of it,
    E_in[t] = E_out[succ(t)] with succ(t) = (successor_mult*t + successor_add) mod V,
    so the residual stream starts at the embedding of a definite next token; every
    layer is a full random transformer block whose output projections are scaled by
    `layer_gain`, i.e. it perturbs but does not drown that signal. The greedy
    continuation is therefore a long non-repeating walk with a large argmax margin
    (robust to accumulation order — required for bit-identical token ids between the
    GPU path and the CPU oracle), while all weights are streamed and all arithmetic
    is executed exactly as for a trained checkpoint.

    A draft is built with `embed_from=target` (its tables are the leading d_model
    columns of the target's) and `flip_fraction`: for that fraction of tokens the
    draft's successor differs from the target's, which sets the acceptance rate by
    construction instead of leaving it at ~0 for unrelated random models.
    """
    assert cfg.arch == ARCH_LLAMA
    gen = torch.Generator(device=device).manual_seed(seed)
    d, ff, V = cfg.d_model, cfg.d_ff, cfg.vocab
    Hq, Hkv, D = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim
    if embed_from is not None:
        src = embed_from.lm_head
        assert src.shape[0] == V and src.shape[1] >= d
        e_out = src[:, :d].contiguous().to(device=device, dtype=dtype)
        # keep the per-row norm comparable to a fresh N(0,1) table
    else:
        e_out = _randn((V, d), 1.0, gen, device, dtype)
    tok = torch.arange(V, device=device, dtype=torch.int64)
    succ = (tok * successor_mult + successor_add) % V
    if flip_fraction > 0.0:
        fg = torch.Generator(device="cpu").manual_seed(flip_seed)
        flip = (torch.rand(V, generator=fg) < flip_fraction).to(device)
        succ = torch.where(flip, (succ * 31 + 17) % V, succ)
    e_in = e_out.index_select(0, succ).contiguous()
    layers = []
    for _ in range(cfg.n_layers):
        layers.append(LayerWeights(
            attn_norm_w=torch.ones(d, dtype=dtype, device=device),
            wqkv=_randn(((Hq + 2 * Hkv) * D, d), 1.0 / math.sqrt(d), gen, device, dtype),
            wo=_randn((d, Hq * D), layer_gain / math.sqrt(Hq * D), gen, device, dtype),
            mlp_norm_w=torch.ones(d, dtype=dtype, device=device),
            w_up=_randn((2 * ff, d), 1.0 / math.sqrt(d), gen, device, dtype),
            w_down=_randn((d, ff), layer_gain / math.sqrt(ff), gen, device, dtype),
        ))
    cos, sin = rope_tables(cfg, device)
    return ModelWeights(
        cfg, e_in, e_out, torch.ones(d, dtype=dtype, device=device), layers, rope_cos=cos, rope_sin=sin,
        meta={"synthetic": True, "seed": seed, "layer_gain": layer_gain, "flip_fraction": flip_fraction,
              "successor": (successor_mult, successor_add)},
    )


def synthetic_gpt2(cfg: ModelConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16, layer_gain: float = 0.05,
                   successor_mult: int = 7919, successor_add: int = 1, embed_from: Optional[ModelWeights] = None,
                   flip_fraction: float = 0.0, flip_seed: int = 99) -> ModelWeights:
    """GPT-2-shaped counterpart of `synthetic_llama` (LayerNorm with biases, learned positions,
    biased linears, gelu_new MLP): the same successor construction on untied in/out tables."""
    assert cfg.arch == ARCH_GPT2
    gen = torch.Generator(device=device).manual_seed(seed)
    d, ff, V = cfg.d_model, cfg.d_ff, cfg.vocab
    if embed_from is not None:
        e_out = embed_from.lm_head[:, :d].contiguous().to(device=device, dtype=dtype)
    else:
        e_out = _randn((V, d), 1.0, gen, device, dtype)
    tok = torch.arange(V, device=device, dtype=torch.int64)
    succ = (tok * successor_mult + successor_add) % V
    if flip_fraction > 0.0:
        fg = torch.Generator(device="cpu").manual_seed(flip_seed)
        flip = (torch.rand(V, generator=fg) < flip_fraction).to(device)
        succ = torch.where(flip, (succ * 31 + 17) % V, succ)
    e_in = e_out.index_select(0, succ).contiguous()

    def vec(std):
        return _randn((d,), std, gen, device, dtype)

    layers = []
    for _ in range(cfg.n_layers):
        layers.append(LayerWeights(
            attn_norm_w=(1.0 + vec(0.02).float()).to(dtype), attn_norm_b=vec(0.02),
            wqkv=_randn((3 * d, d), 1.0 / math.sqrt(d), gen, device, dtype), bqkv=_randn((3 * d,), 0.02, gen, device, dtype),
            wo=_randn((d, d), layer_gain / math.sqrt(d), gen, device, dtype), bo=vec(0.01),
            mlp_norm_w=(1.0 + vec(0.02).float()).to(dtype), mlp_norm_b=vec(0.02),
            w_up=_randn((ff, d), 1.0 / math.sqrt(d), gen, device, dtype), b_up=_randn((ff,), 0.02, gen, device, dtype),
            w_down=_randn((d, ff), layer_gain / math.sqrt(ff), gen, device, dtype), b_down=vec(0.01),
        ))
    return ModelWeights(
        cfg, e_in, e_out, (1.0 + vec(0.02).float()).to(dtype), layers, final_norm_b=vec(0.02),
        pos_emb=_randn((cfg.max_pos, d), 0.05, gen, device, dtype),
        meta={"synthetic": True, "seed": seed, "layer_gain": layer_gain, "flip_fraction": flip_fraction,
              "successor": (successor_mult, successor_add)},
    )


def random_init(cfg: ModelConfig, seed: int = 0, std: float = 0.02, device="cpu", dtype=torch.bfloat16) -> ModelWeights:
    """Plain random initialisation of the architecture `cfg` names, as `transformers` initialises an untrained
    model (`initializer_range` 0.02: every Linear / embedding weight ~ N(0, std), norm weights 1) — except that
    biases and the norm offsets are drawn ~ N(0, std) too instead of zero, so that every operand of the forward
    carries information. No successor structure and no damped layers: argmax margins are whatever a random
    network has, which is what a numerical (logit-level) comparison wants. `synthetic_llama` is the
    token-parity counterpart."""
    gen = torch.Generator(device=device).manual_seed(seed)
    d, ff, V = cfg.d_model, cfg.d_ff, cfg.vocab
    Hq, Hkv, D = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim
    llama = cfg.arch == ARCH_LLAMA

    def w(*shape):
        return _randn(shape, std, gen, device, dtype)

    def ones_plus(n):
        return (1.0 + _randn((n,), std, gen, device, torch.float32)).to(dtype)

    tok = w(V, d)
    head = tok if cfg.tie_embeddings else w(V, d)
    layers = []
    for _ in range(cfg.n_layers):
        if llama:
            layers.append(LayerWeights(attn_norm_w=ones_plus(d), wqkv=w((Hq + 2 * Hkv) * D, d), wo=w(d, Hq * D),
                                       mlp_norm_w=ones_plus(d), w_up=w(2 * ff, d), w_down=w(d, ff)))
        else:
            layers.append(LayerWeights(attn_norm_w=ones_plus(d), attn_norm_b=w(d), wqkv=w(3 * d, d), bqkv=w(3 * d),
                                       wo=w(d, d), bo=w(d), mlp_norm_w=ones_plus(d), mlp_norm_b=w(d),
                                       w_up=w(ff, d), b_up=w(ff), w_down=w(d, ff), b_down=w(d)))
    meta = {"synthetic": True, "seed": seed, "init": f"normal(0, {std})"}
    if llama:
        cos, sin = rope_tables(cfg, device)
        return ModelWeights(cfg, tok, head, ones_plus(d), layers, rope_cos=cos, rope_sin=sin, meta=meta)
    return ModelWeights(cfg, tok, head, ones_plus(d), layers, final_norm_b=w(d), pos_emb=w(cfg.max_pos, d), meta=meta)


# ------------------------------------------------------------------------------- Medusa heads
@dataclass
class MedusaHeads:
    """K persistent vocabulary-sized heads over the target's last hidden state: head i proposes the token
    i+1 positions after the last emitted one. weights: bf16 [K][V][d_model]."""
    weights: torch.Tensor
    meta: Dict[str, object] = field(default_factory=dict)

    @property
    def n_heads(self) -> int:
        return int(self.weights.shape[0])


def synthetic_medusa_heads(target: ModelWeights, n_heads: int, flip_fraction: float = 0.2, flip_seed: int = 7) -> MedusaHeads:
    """Heads for a `synthetic_llama` target. The target's hidden state before the lm_head points at E_out[t]
    where t is its next token, and its greedy walk is t -> succ(t); head i is therefore the output table with
    its rows moved along the walk, head_i[succ^i(t)] = E_out[t], so that argmax head_i(h) = succ^i(t). For a
    `flip_fraction` of the tokens (per head) the row goes to another token: those proposals are wrong, which
    sets the acceptance rate by construction (as `synthetic_llama(..., flip_fraction)` does for a draft model)."""
    mult, add = target.meta.get("successor", (7919, 1))
    e_out = target.lm_head
    V, d = e_out.shape
    dev = e_out.device
    tok = torch.arange(V, device=dev, dtype=torch.int64)
    heads = torch.empty((n_heads, V, d), dtype=e_out.dtype, device=dev)
    dest = tok.clone()
    fg = torch.Generator(device="cpu").manual_seed(flip_seed)
    for i in range(n_heads):
        dest = (dest * mult + add) % V                       # succ^(i+1)(t) for every t
        flip = (torch.rand(V, generator=fg) < flip_fraction).to(dev)
        wrong = (dest * 31 + 17) % V
        target_row = torch.where(flip, wrong, dest)
        heads[i].zero_()
        heads[i].index_copy_(0, target_row, e_out)           # collisions: the later source row wins (still one hot spot each)
    return MedusaHeads(heads, meta={"synthetic": True, "flip_fraction": flip_fraction})
