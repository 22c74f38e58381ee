"""BASELINE configs 4 and 5 through the pipeline, against the oracle loop, at the 8B LAYER shapes.

config 4: Llama-3-8B target + Llama-3.2-1B draft, K=4, 32 prompts over 8 GPUs = 4 rows per GPU. config 5: Llama-3-8B +
Medusa-lite draft, K=4, fp8 weights. The CPU oracle cannot run 32 + 16 layers in test time, so the pair here has the
real layer dimensions (d 4096 / 32 q heads / 8 kv heads / D 128 / d_ff 14336 for the target, d 2048 / D 64 / d_ff 8192
for the draft), the real vocabulary size where stated, and TWO layers per model: every kernel instantiation the full
models run (D=128 and D=64 attention, the d_ff=14336 down projection through the MASK + aliased-partials GEMV at <= 5
tokens and through gemm_skinny at 20 tokens, the fp8 tile streams) runs here and the emitted tokens, the counters and
the sequences must equal the oracle's, token for token. The token comparison uses the successor-structured synthetic
weights (argmax margins that summation order cannot flip); logit-level parity on un-engineered weights of the same shapes
is tests/test_hip_fullshape_parity_gpu.py. The step loop mirrored: src/specdec/core/pipeline.py:1984-3733 (generate_batch),
:984-1275 (generate), drafting through src/specdec/modes/medusa.py:104-186."""

import dataclasses
import os

import pytest
import torch

from helpers import synthetic_prompts
from oracle import fp8_ref
from oracle.model_ref import OracleLM
from oracle.pipeline_ref import OraclePipeline
from specdec_hip import weights as W

pytestmark = pytest.mark.gpu

VOCAB = 16384


def _pair(vocab=VOCAB):
    tcfg = dataclasses.replace(W.LLAMA_3_8B, n_layers=2, vocab=vocab, max_pos=512, name="8b-shape-2L")
    dcfg = dataclasses.replace(W.LLAMA_3_2_1B, n_layers=2, vocab=vocab, max_pos=512, tie_embeddings=False, name="1b-shape-2L")
    tgt = W.synthetic_llama(tcfg, seed=40, device="cuda", layer_gain=0.25)
    drf = W.synthetic_llama(dcfg, seed=41, device="cuda", layer_gain=0.25, embed_from=tgt, flip_fraction=0.25)
    return drf, tgt


@pytest.mark.parametrize("wd", ["bf16", "fp8"])
def test_config4_per_gpu_shape_k4_batch4(wd):
    """4 rows per GPU, K=4: 20-token verify passes (gemm_skinny, 2 token groups) over the 8B layer shapes, 4-row draft
    passes; bf16 and fp8 storage."""
    from src.specdec import HipLM, SpeculativePipeline

    if wd == "fp8" and not os.environ.get("SPECDEC_RUN_SLOW"):
        # config 4 is a bf16 configuration; its fp8 twin costs another ~27 s of CPU-oracle forwards. fp8 storage under multi-token
        # passes at full size stays in the default run: tests/test_full_size_gpu.py (3B + 1B, 40-token verify),
        # tests/test_hip_fullshape_parity_gpu.py (8B shapes), config 5 below
        pytest.skip("fp8 twin of config 4: SPECDEC_RUN_SLOW=1 runs it")
    drf, tgt = _pair()
    prompts = synthetic_prompts(4, 24, VOCAB).tolist()
    pipe = SpeculativePipeline(base_lm=HipLM(tgt, weight_dtype=wd), draft_lm=HipLM(drf, weight_dtype=wd),
                               controller="fixed", controller_params={"k": 4}, seed=1234)
    got = pipe.generate_batch(prompts, max_tokens=8, do_sample=False)
    t_cpu, d_cpu = tgt.to("cpu"), drf.to("cpu")
    if wd == "fp8":
        t_cpu, d_cpu = fp8_ref.dequantized(t_cpu), fp8_ref.dequantized(d_cpu)
    want = OraclePipeline(OracleLM(t_cpu, "bf16"), OracleLM(d_cpu, "bf16"), k=4, eos_token_id=tgt.config.eos_token_id).generate_batch(prompts, 8)
    for b in range(4):
        assert got[b]["generated_tokens"] == want[b]["generated_tokens"], (wd, b)
        assert (got[b]["proposed"], got[b]["accepted"]) == (want[b]["proposed"], want[b]["accepted"])
        assert got[b]["sequence"] == want[b]["sequence"]
    acc = sum(r["accepted"] for r in got) / sum(r["proposed"] for r in got)
    assert 0.3 < acc <= 1.25, acc
    # batch 1 of the same pair: 5-token verify passes = gemv.hip's MASK variant with the partials aliased onto x (d_ff 14336)
    one = pipe.generate_batch([prompts[2]], max_tokens=8, do_sample=False)[0]
    assert one["generated_tokens"] == want[2]["generated_tokens"]


@pytest.mark.parametrize("mode", ["tied", "heads"])
def test_config5_medusa_fp8_k4(mode):
    """fp8 weight storage, K=4, no draft model. "tied": Medusa-lite as the reference's draftor defines it under greedy
    decoding (heads tied to the lm_head: K copies of the target's next token), generate(); "heads": K persistent
    vocabulary-sized heads, fp8 like the lm_head (the useful variant, not in the reference), generate_batch()."""
    from src.specdec import HipLM, SpeculativePipeline

    _, tgt = _pair()
    tq = fp8_ref.dequantized(tgt.to("cpu"))
    lm = OracleLM(tq, "bf16")
    eos = tgt.config.eos_token_id
    prompts = synthetic_prompts(2, 16, VOCAB).tolist()
    if mode == "tied":
        pipe = SpeculativePipeline(base_lm=HipLM(tgt, weight_dtype="fp8"), draft_model="none", draft_mode="medusa",
                                   controller="fixed", controller_params={"k": 4}, seed=1234)
        got = pipe.generate(prompts[0], max_tokens=10, do_sample=False)
        want = OraclePipeline(lm, None, k=4, eos_token_id=eos, draft_mode="medusa_tied").generate(prompts[0], 10)
        assert got["generated_tokens"] == want["generated_tokens"]
        assert (got["proposed"], got["accepted"], got["steps"]) == (want["proposed"], want["accepted"], want["steps"])
        return
    heads = W.synthetic_medusa_heads(tgt, 4, flip_fraction=0.2)
    pipe = SpeculativePipeline(base_lm=HipLM(tgt, weight_dtype="fp8"), draft_model="none", draft_mode="medusa", medusa_heads=heads,
                               controller="fixed", controller_params={"k": 4}, seed=1234)
    got = pipe.generate_batch(prompts, max_tokens=12, do_sample=False)
    hq = torch.stack([fp8_ref.quantize_rows(h)[0].float() * fp8_ref.quantize_rows(h)[1][:, None] for h in heads.weights.cpu()])
    want = OraclePipeline(lm, None, k=4, eos_token_id=eos, draft_mode="medusa_heads", medusa_heads=hq).generate_batch(prompts, 12)
    for b in range(2):
        assert got[b]["generated_tokens"] == want[b]["generated_tokens"], b
        assert (got[b]["proposed"], got[b]["accepted"]) == (want[b]["proposed"], want[b]["accepted"])
