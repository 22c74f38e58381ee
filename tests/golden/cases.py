"""Seeded input builders shared by make_golden.py (reference side) and the tests.

Inputs are regenerated from the seed on both sides (numpy PCG64 streams are stable
across platforms); only the reference's OUTPUTS are stored in the fixtures, plus an
input checksum that guards against generator drift.
"""

from __future__ import annotations

import numpy as np
import torch

# (name, B, K, V, dtype, pattern)
VERIFY_CASES = [
    ("ref_test_1x1x100", 1, 1, 100, "f32", "random"),
    ("ref_test_1x2x1000", 1, 2, 1000, "f32", "random"),
    ("ref_test_2x3x5000", 2, 3, 5000, "f32", "planted"),
    ("ref_test_4x4x10000", 4, 4, 10000, "f32", "planted"),
    ("gpt2_1x2x50257", 1, 2, 50257, "f32", "planted"),
    ("gpt2_1x2x50257_bf16", 1, 2, 50257, "bf16", "planted"),
    ("llama_8x4x128256_bf16", 8, 4, 128256, "bf16", "planted"),
    ("llama_8x8x128256_bf16", 8, 8, 128256, "bf16", "planted"),
    ("llama_1x4x128256_f32", 1, 4, 128256, "f32", "planted"),
    ("f16_3x5x4099", 3, 5, 4099, "f16", "planted"),
    ("ties_2x4x777", 2, 4, 777, "f32", "ties"),
    ("ties_bf16_2x4x8200", 2, 4, 8200, "bf16", "ties"),
    ("neginf_2x3x513", 2, 3, 513, "f32", "neginf"),
    ("int32_ids_2x3x1000", 2, 3, 1000, "f32", "planted_i32"),
    ("k1_5x1x333", 5, 1, 333, "f32", "planted"),
    ("unaligned_3x3x1001_bf16", 3, 3, 1001, "bf16", "planted"),
    ("wide_k_1x70x300", 1, 70, 300, "f32", "planted_wide"),
]

_DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def build_verify_case(name, B, K, V, dtype, pattern, seed):
    """-> logits [B,K,V] (dtype), draft_ids [B,K] (int64 or int32)."""
    rng = np.random.default_rng(seed)
    logits = torch.from_numpy(rng.standard_normal((B, K, V), dtype=np.float32))
    ids = torch.from_numpy(rng.integers(0, V, size=(B, K), dtype=np.int64))
    if pattern.startswith("planted"):
        # row b accepts a prefix of length (b * 3 + 1) % (K + 1); the next position is a
        # planted mismatch; later positions are planted matches again (prefix-only mask!)
        for b in range(B):
            want = (b * 3 + 1) % (K + 1)
            for k in range(K):
                tgt = int(ids[b, k])
                if k == want:
                    tgt = (tgt + 1) % V
                logits[b, k, tgt] = 30.0 + k
    elif pattern == "ties":
        # exact ties: the maximum appears several times; lowest index must win
        for b in range(B):
            for k in range(K):
                lo = int(rng.integers(0, V // 2))
                hi = int(rng.integers(V // 2, V))
                logits[b, k, lo] = 25.0
                logits[b, k, hi] = 25.0
                ids[b, k] = lo if (b + k) % 3 else hi  # hi => mismatch
    elif pattern == "neginf":
        logits[0, 0, :] = float("-inf")  # argmax of an all -inf row is index 0
        ids[0, 0] = 0
        logits[0, 1, :] = float("-inf")
        logits[0, 1, V - 1] = -1e30
        ids[0, 1] = V - 1
        logits[1, 0, 7] = float("inf")
        ids[1, 0] = 7
    logits = logits.to(_DT[dtype])
    if pattern == "planted_i32":
        ids = ids.to(torch.int32)
    return logits, ids


# (name, B, H, L, K, D, dtype)
KV_CASES = [
    ("ref_test_basic", 1, 2, 3, 2, 4, "f32"),
    ("ref_test_single", 1, 2, 5, 1, 8, "f32"),
    ("ref_test_heads", 1, 8, 10, 5, 16, "f32"),
    ("llama1b_b2", 2, 8, 37, 5, 64, "bf16"),
    ("llama3b_b1", 1, 8, 130, 5, 128, "bf16"),
    ("gpt2_f16", 1, 12, 9, 3, 64, "f16"),
    ("odd_dim_f32", 2, 3, 4, 2, 5, "f32"),
    ("odd_dim_bf16", 2, 3, 4, 2, 7, "bf16"),
    ("empty_base", 2, 2, 0, 3, 8, "f32"),
    ("empty_new", 1, 2, 6, 0, 8, "bf16"),
    ("negative_accept_len", 3, 2, 4, 4, 8, "f32"),   # invalid input: the reference loop still writes ONE row per negative entry
]


def build_kv_case(name, B, H, L, K, D, dtype, seed):
    rng = np.random.default_rng(seed)

    def mk(*shape):
        return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32)).to(_DT[dtype])

    base_k, base_v = mk(B, H, L, D), mk(B, H, L, D)
    new_k, new_v = mk(B, H, K, D), mk(B, H, K, D)
    # masks: row-dependent, including non-prefix masks and zero accept
    mask = torch.zeros((B, K), dtype=torch.uint8)
    alen = torch.zeros(B, dtype=torch.int32)
    for b in range(B):
        bits = rng.integers(0, 2, size=K)
        if b == 0 and K:
            bits[:] = 1
        mask[b] = torch.from_numpy(bits.astype(np.uint8))
        alen[b] = int(rng.integers(0, K + 1)) if b else K
    if name == "negative_accept_len":
        mask[:] = torch.tensor([[0, 1, 1, 0], [1, 0, 1, 1], [0, 0, 0, 0]], dtype=torch.uint8)
        alen[:] = torch.tensor([-1, -3, -2], dtype=torch.int32)
    return base_k, base_v, new_k, new_v, mask, alen


def checksum(t: torch.Tensor) -> float:
    x = t.detach().cpu()
    if x.dtype.is_floating_point:
        x = torch.nan_to_num(x.double(), nan=7.0, posinf=11.0, neginf=-13.0)
    return float(x.double().sum())


# ---------------------------------------------------------------- G8 pipeline pairs
def g8_pairs(dtype=torch.float32):
    """The tiny Llama draft/target pairs of the pipeline goldens, rebuilt from seeds
    (torch CPU generator; a weight checksum in the fixture guards against drift)."""
    from specdec_hip import weights as W  # weight containers/builders only

    rs = {"factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
          "original_max_position_embeddings": 64, "rope_type": "llama3"}
    tcfg = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=2, d_model=64, n_heads=2, n_kv_heads=1, head_dim=32, d_ff=128,
                         vocab=160, max_pos=256, rope_theta=500000.0, tie_embeddings=False, eos_token_id=2,
                         rope_scaling=rs, name="g8-target")
    dcfg = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=1, d_model=64, n_heads=2, n_kv_heads=1, head_dim=32, d_ff=128,
                         vocab=160, max_pos=256, rope_theta=500000.0, tie_embeddings=False, eos_token_id=2,
                         rope_scaling=rs, name="g8-draft")
    out = {}
    tgt = W.synthetic_llama(tcfg, seed=11, dtype=dtype, layer_gain=0.05, successor_mult=37, successor_add=5)
    drf = W.synthetic_llama(dcfg, seed=12, dtype=dtype, layer_gain=0.05, successor_mult=37, successor_add=5,
                            embed_from=tgt, flip_fraction=0.3)
    out["structured"] = (drf, tgt)
    # successor(t) = t: immediate repeats, the reference's de-duplication heuristics fire
    tgt2 = W.synthetic_llama(tcfg, seed=21, dtype=dtype, layer_gain=0.05, successor_mult=1, successor_add=0)
    drf2 = W.synthetic_llama(dcfg, seed=22, dtype=dtype, layer_gain=0.05, successor_mult=1, successor_add=0,
                             embed_from=tgt2, flip_fraction=0.3)
    out["repeating"] = (drf2, tgt2)
    return out


def policy_pair(dtype=torch.float32):
    """A G8-shaped pair whose output tables are scaled by 0.12: the greedy token still wins by ~5 logits (argmax is
    robust to bf16 / summation order) but its softmax probability is ~0.9 and the runners-up matter, so the
    logit-threshold policies (typical, topk_agree, conf_threshold) accept and reject differently from exact match."""
    from specdec_hip import weights as W

    rs = {"factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
          "original_max_position_embeddings": 64, "rope_type": "llama3"}
    kw = dict(arch=W.ARCH_LLAMA, d_model=64, n_heads=2, n_kv_heads=1, head_dim=32, d_ff=128, vocab=160, max_pos=256,
              rope_theta=500000.0, tie_embeddings=False, eos_token_id=2, rope_scaling=rs)
    tgt = W.synthetic_llama(W.ModelConfig(n_layers=2, name="pol-target", **kw), seed=31, dtype=dtype, layer_gain=0.05,
                            successor_mult=37, successor_add=5)
    tgt.lm_head = (tgt.lm_head.float() * 0.12).to(dtype)
    drf = W.synthetic_llama(W.ModelConfig(n_layers=1, name="pol-draft", **kw), seed=32, dtype=dtype, layer_gain=0.05,
                            successor_mult=37, successor_add=5, embed_from=tgt, flip_fraction=0.3)
    return drf, tgt


POLICY_RUNS = [("typical", {"p": 0.9}), ("typical", {"p": 0.5}), ("topk_agree", {"k": 20}), ("topk_agree", {"k": 2}),
               ("conf_threshold", {"tau": 0.9}), ("conf_threshold", {"tau": 0.6})]


def weights_checksum(mw) -> float:
    return float(sum(checksum(t) for name, t in mw.tensors() if not name.startswith("rope_")))


def build_policy_case(K, V, seed):
    """draft logits, base logits [1,K,V] fp32 and token ids [1,K] for the policy goldens: the
    base agrees with the draft on a random prefix, so every policy sees accepts and rejects."""
    rng = np.random.default_rng(seed)
    dl = torch.from_numpy(rng.standard_normal((1, K, V)).astype(np.float32)) * 2
    bl = torch.from_numpy(rng.standard_normal((1, K, V)).astype(np.float32)) * 2
    d_ids = dl.argmax(-1)
    agree = int(rng.integers(0, K + 1))
    for k in range(K):
        if k < agree:
            bl[0, k, int(d_ids[0, k])] = 9.0 + rng.random()
    b_ids = bl.argmax(-1)
    return dl, bl, d_ids, b_ids
