"""Device-selected draft forward 0 (engine.hip enqueue_step / misc.hip accept_kernel) against the CPU oracle.

With one sequence and a draft whose 1- and 2-token passes are persistent launches, the captured step holds draft forward 0 in
two forms — the 2-token pass over (prev, last) and the 1-token pass over `last` — and accept_kernel writes which one the next
step runs: prev's K/V are missing from the draft cache only after a step whose k proposals were all accepted (bonus emit mode)
or after the host has set the row. The oracle (oracle/pipeline_ref.py, a restatement of the reference loops,
/root/reference/src/specdec/core/pipeline.py:984-1275, 1984-3733) runs the draft model itself, so tokens AND the proposed /
accepted counters AND the per-step k trace pin the proposals, not only the emitted target tokens. The tiny pair of
tests/helpers.py is not eligible for the persistent launch (head_dim 32); this pair is (head_dim 64, dimensions in whole 128s, an even number of SwiGLU pairs per workgroup)."""

import pytest
import torch

from helpers import synthetic_prompts
from oracle.model_ref import OracleLM
from oracle.pipeline_ref import OraclePipeline
from specdec_hip import weights as W

pytestmark = pytest.mark.gpu

TGT = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=3, d_model=256, n_heads=4, n_kv_heads=2, head_dim=64, d_ff=512,
                    vocab=2048, max_pos=1024, rope_theta=500000.0, tie_embeddings=False, name="select-target")
DRF = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=2, d_model=256, n_heads=4, n_kv_heads=1, head_dim=64, d_ff=512,
                    vocab=2048, max_pos=1024, rope_theta=500000.0, tie_embeddings=False, name="select-draft")


def _pair(flip):
    tgt = W.synthetic_llama(TGT, seed=0, device="cpu", layer_gain=0.05)
    drf = W.synthetic_llama(DRF, seed=1, device="cpu", layer_gain=0.05, embed_from=tgt, flip_fraction=flip)
    return drf, tgt


def _pipe(drf, tgt, controller="fixed", params=None):
    from src.specdec import HipLM, SpeculativePipeline

    d = HipLM(drf.to("cuda"))
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda")), draft_lm=d, controller=controller, controller_params=params or {"k": 4}, seed=1234)
    return pipe, d


def _oracle(drf, tgt):
    return OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4, eos_token_id=tgt.config.eos_token_id)


@pytest.mark.parametrize("flip", [0.0, 0.3, 0.9])
def test_fixed_k_one_row_proposals_equal_the_oracle(flip, monkeypatch):
    """flip 0.0: every step fully accepted (always the 2-token form); 0.9: almost never (the 1-token form); 0.3: a mix."""
    drf, tgt = _pair(flip)
    prompts = synthetic_prompts(1, 9, TGT.vocab, seed=3).tolist()
    want = _oracle(drf, tgt).generate_batch(prompts, 48)[0]
    for select in (True, False):
        if not select:
            monkeypatch.setenv("SPECDEC_NO_FWD0_SELECT", "1")
        pipe, d = _pipe(drf, tgt)
        got = pipe.generate_batch(prompts, max_tokens=48, do_sample=False)[0]
        assert got["generated_tokens"] == want["generated_tokens"], (flip, select)
        assert (got["proposed"], got["accepted"]) == (want["proposed"], want["accepted"]), (flip, select)
    monkeypatch.delenv("SPECDEC_NO_FWD0_SELECT")


def test_the_pair_is_served_by_persistent_launches():
    from specdec_hip.engine import HipModel

    drf, _ = _pair(0.3)
    assert HipModel(drf.to("cuda"), batch=1, l_max=128).persist_tokens >= 2, "the draft of this file must be eligible, or the tests above check nothing new"


def test_adaptive_k_and_draft_emit_mode_one_row():
    """Per-row adaptive K on the device (k < K: a step is 'fully accepted' at the row's own k) and generate() (draft-token emit
    mode: prev is always an earlier input) — tokens, counters and the k trace equal the oracle's."""
    drf, tgt = _pair(0.35)
    params = {"initial_k": 2, "min_k": 1, "max_k": 4, "step_size": 1, "target_acceptance_rate": 0.6}
    pipe, _ = _pipe(drf, tgt, "adaptive", dict(params, per_row=True))
    prompts = synthetic_prompts(1, 9, TGT.vocab, seed=5).tolist()
    got = pipe.generate_batch(prompts, max_tokens=48, do_sample=False)[0]
    o = _oracle(drf, tgt)
    want = o.generate_batch(prompts, 48, per_row_k=params)[0]
    assert got["generated_tokens"] == want["generated_tokens"]
    assert (got["proposed"], got["accepted"]) == (want["proposed"], want["accepted"])
    assert got["k_trace"] == o.k_trace[0]
    pipe2, _ = _pipe(drf, tgt)
    single = pipe2.generate(prompts[0], max_tokens=32, do_sample=False)
    ref = _oracle(drf, tgt).generate(prompts[0], 32)
    assert single["generated_tokens"] == ref["generated_tokens"]
    assert (single["proposed"], single["accepted"]) == (ref["proposed"], ref["accepted"])
