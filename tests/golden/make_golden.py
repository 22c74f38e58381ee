"""Generate golden fixtures by importing the REFERENCE (build container only).

    cd /tmp && python /root/repo/tests/golden/make_golden.py

Requires /root/reference (read-only). It never runs on the GPU box; the fixtures it
writes (tests/golden/*.json / *.npz) are data: inputs-by-seed and the reference's
outputs. No reference source text is stored.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SPECDEC_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, REF)

import cases  # noqa: E402


def gen_kernels():
    from kernels.reference import kv_append_ref, kv_append_with_mask_ref, verify_prefix_ref

    out = {"verify": {}, "kv": {}}
    for i, case in enumerate(cases.VERIFY_CASES):
        name = case[0]
        seed = 1000 + i
        logits, ids = cases.build_verify_case(*case, seed=seed)
        # the reference computes on whatever dtype it is given (argmax on bf16 works on CPU)
        alen, mask = verify_prefix_ref(logits, ids)
        pred = torch.argmax(logits, dim=-1)
        out["verify"][name] = {
            "seed": seed,
            "accept_len": alen.tolist(),
            "mask": mask.tolist(),
            "argmax": pred.tolist(),
            "logits_checksum": cases.checksum(logits),
            "ids_checksum": cases.checksum(ids),
        }
    arrays = {}
    for i, case in enumerate(cases.KV_CASES):
        name = case[0]
        seed = 2000 + i
        bk, bv, nk, nv, mask, alen = cases.build_kv_case(*case, seed=seed)
        ok, ov = kv_append_ref(bk, bv, nk, nv)
        mk, mv = kv_append_with_mask_ref(bk, bv, nk, nv, mask, alen)
        out["kv"][name] = {
            "seed": seed,
            "inputs_checksum": [cases.checksum(t) for t in (bk, bv, nk, nv, mask, alen)],
            "concat_checksum": [cases.checksum(ok), cases.checksum(ov)],
            "masked_checksum": [cases.checksum(mk), cases.checksum(mv)],
        }
        # full outputs for the small cases, as float32 arrays (exact for bf16/f16 values)
        if ok.numel() <= 40000:
            arrays[f"{name}/concat_k"] = ok.float().numpy()
            arrays[f"{name}/concat_v"] = ov.float().numpy()
            arrays[f"{name}/masked_k"] = mk.float().numpy()
            arrays[f"{name}/masked_v"] = mv.float().numpy()
    with open(os.path.join(HERE, "kernels_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "kernels_golden.npz"), **arrays)
    print("kernels goldens:", len(out["verify"]), "verify cases,", len(out["kv"]), "kv cases")


def _tiny_hf_models():
    """Tiny random HF models (fp32) — the third-party forward the reference calls."""
    import transformers

    torch.manual_seed(1234)
    lcfg = transformers.LlamaConfig(
        vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
        num_attention_heads=2, num_key_value_heads=1, head_dim=32, max_position_embeddings=256,
        rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=False,
        rope_scaling={"factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                      "original_max_position_embeddings": 64, "rope_type": "llama3"},
        attention_bias=False, mlp_bias=False,
    )
    llama = transformers.LlamaForCausalLM(lcfg).eval()
    gcfg = transformers.GPT2Config(vocab_size=96, n_positions=128, n_embd=64, n_layer=2, n_head=2,
                                   activation_function="gelu_new", resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    gpt2 = transformers.GPT2LMHeadModel(gcfg).eval()
    # default init leaves biases at zero: randomise them so the bias paths are pinned too
    with torch.no_grad():
        for n, p_ in gpt2.named_parameters():
            if n.endswith("bias"):
                p_.normal_(0.0, 0.1)
    return (lcfg, llama), (gcfg, gpt2)


def gen_hf():
    """Pin for oracle/model_ref.py: logits of transformers' own Llama / GPT-2 forward."""
    import transformers

    (lcfg, llama), (gcfg, gpt2) = _tiny_hf_models()
    rng = np.random.default_rng(77)
    for name, cfg, model in (("llama", lcfg, llama), ("gpt2", gcfg, gpt2)):
        toks = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=(2, 70), dtype=np.int64))
        with torch.no_grad():
            logits = model(input_ids=toks).logits.float()
        arrays = {"__tokens": toks.numpy(), "__logits": logits.numpy()}
        for k, v in model.state_dict().items():
            arrays[k] = v.detach().float().numpy()
        np.savez_compressed(os.path.join(HERE, f"hf_{name}_tiny.npz"), **arrays)
        meta = {"transformers": transformers.__version__, "torch": torch.__version__,
                "config": {k: v for k, v in cfg.to_dict().items()
                           if isinstance(v, (int, float, str, bool, dict, type(None)))}}
        with open(os.path.join(HERE, f"hf_{name}_tiny.json"), "w") as f:
            json.dump(meta, f, indent=1, default=str)
        print("hf golden:", name, tuple(logits.shape))


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["kernels"]
    if "kernels" in which:
        gen_kernels()
    if "hf" in which:
        gen_hf()
