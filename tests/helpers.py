"""Shared test helpers: golden HF models -> ModelWeights, tiny synthetic pairs."""

import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from specdec_hip import weights as W

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_hf_golden(name: str, dtype=torch.float32):
    """-> (ModelWeights on CPU in `dtype`, tokens int64 [B][L], HF logits fp32 [B][L][V])"""
    z = np.load(os.path.join(GOLD, f"hf_{name}_tiny.npz"))
    with open(os.path.join(GOLD, f"hf_{name}_tiny.json")) as f:
        meta = json.load(f)
    hf_cfg = SimpleNamespace(**meta["config"])
    for k in ("head_dim", "rope_scaling", "rope_parameters", "rope_theta", "n_inner", "tie_word_embeddings"):
        if not hasattr(hf_cfg, k):
            setattr(hf_cfg, k, None)
    cfg = W.config_from_hf(hf_cfg)
    sd = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("__")}
    mw = W.from_hf_state_dict(cfg, sd, dtype=dtype, device="cpu")
    if cfg.arch == W.ARCH_LLAMA:
        mw.rope_cos, mw.rope_sin = W.rope_tables(cfg, "cpu")
    return mw, torch.from_numpy(z["__tokens"]), torch.from_numpy(z["__logits"])


TINY_TARGET = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=3, d_model=128, n_heads=4, n_kv_heads=2, head_dim=32,
                            d_ff=256, vocab=1000, max_pos=512, rope_theta=500000.0,
                            rope_scaling={"factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                          "original_max_position_embeddings": 64, "rope_type": "llama3"},
                            tie_embeddings=False, name="tiny-target")
TINY_DRAFT = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=2, d_model=64, n_heads=2, n_kv_heads=1, head_dim=32,
                           d_ff=128, vocab=1000, max_pos=512, rope_theta=500000.0, rope_scaling=TINY_TARGET.rope_scaling,
                           tie_embeddings=False, name="tiny-draft")


def tiny_pair(flip_fraction=0.25, layer_gain=0.05, seed=0):
    tgt = W.synthetic_llama(TINY_TARGET, seed=seed, device="cpu", layer_gain=layer_gain)
    drf = W.synthetic_llama(TINY_DRAFT, seed=seed + 1, device="cpu", layer_gain=layer_gain,
                            embed_from=tgt, flip_fraction=flip_fraction)
    return drf, tgt


def synthetic_prompts(n, length, vocab, seed=1234):
    """ids uniform in [4, V), generator seed 1234+i, equal length."""
    out = []
    for i in range(n):
        g = torch.Generator().manual_seed(seed + i)
        out.append(torch.randint(4, vocab, (length,), generator=g, dtype=torch.int64))
    return torch.stack(out, 0)
