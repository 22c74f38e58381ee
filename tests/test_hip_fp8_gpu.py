"""fp8 weight storage (BASELINE config 5's dtype): the device quantiser bit for bit against
oracle/fp8_ref.py, the forward against the oracle over the dequantised weights, and the step loop."""

import pytest
import torch

from helpers import load_hf_golden, synthetic_prompts, tiny_pair
from oracle import fp8_ref
from oracle.model_ref import OracleLM
from oracle.pipeline_ref import OraclePipeline

pytestmark = pytest.mark.gpu


def _rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)


@pytest.mark.parametrize("N,K,scale", [(7, 64, 1.0), (300, 3072, 0.02), (128, 8192, 5.0), (1000, 36, 1e-3)])
def test_quantizer_is_bit_exact(N, K, scale):
    from specdec_hip.ops import quantize_fp8_rows_hip

    g = torch.Generator().manual_seed(N + K)
    w = (torch.randn(N, K, generator=g) * scale).to(torch.bfloat16)
    w[0] = 0                         # all-zero row: scale 1
    w[1, :8] = torch.tensor([448.0, -448.0, 1e-9, -1e-9, 0.0, 3.0, 17.0, 0.0078125]).to(torch.bfloat16) * scale
    q, s = quantize_fp8_rows_hip(w.cuda())
    want_q, want_s = fp8_ref.quantize_rows(w)
    assert torch.equal(s.cpu(), want_s)
    assert torch.equal(q.cpu().view(torch.uint8), want_q.view(torch.uint8))


@pytest.mark.parametrize("name,L", [("llama", None), ("gpt2", None), ("llama", 5), ("llama", 40), ("gpt2", 64)])
def test_fp8_forward_matches_oracle_over_dequantized_weights(name, L):
    """L = None: the golden prompt; 5: one gemv.hip pass; 40 / 64: the multi-token kernel streaming fp8."""
    from specdec_hip.engine import HipModel

    mw, toks, _ = load_hf_golden(name, dtype=torch.bfloat16)
    if L is not None:
        toks = torch.randint(0, mw.config.vocab, (2, L), generator=torch.Generator().manual_seed(L))
    B, L = toks.shape
    want8, _ = OracleLM(fp8_ref.dequantized(mw), precision="bf16").forward(toks)
    want16, _ = OracleLM(mw, precision="bf16").forward(toks)
    hm = HipModel(mw.to("cuda"), batch=B, l_max=128, weight_dtype="fp8")
    assert hm.pass_tokens in (64, 128)
    ids, logits = hm.forward(toks.to(torch.int32).cuda(), torch.zeros(B, dtype=torch.int32, device="cuda"), 0, want_logits=True)
    got = logits.float().cpu()
    e8, gap = _rel_err(got, want8), _rel_err(want8, want16)
    assert e8 < 0.03, e8
    assert gap > 0.0 and e8 < max(0.5 * gap, 0.01), (e8, gap)     # closer to the fp8 oracle than fp8 is to bf16
    assert torch.equal(ids.cpu().long(), got.argmax(-1))


@pytest.mark.parametrize("k,batch", [(4, 1), (2, 3)])
def test_fp8_step_loop_matches_oracle_tokens(k, batch):
    """Target and draft both streamed as fp8: same tokens and counters as the oracle loop over the
    dequantised weights (large argmax margins by construction: the quantisation does not flip a token
    between the two implementations, and the run still equals the bf16 continuation here)."""
    from src.specdec import HipLM, SpeculativePipeline

    drf, tgt = tiny_pair(flip_fraction=0.25)
    prompts = synthetic_prompts(batch, 14, tgt.config.vocab).tolist()
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda"), weight_dtype="fp8"), draft_lm=HipLM(drf.to("cuda"), weight_dtype="fp8"),
                               controller="fixed", controller_params={"k": k}, seed=1234)
    got = pipe.generate_batch(prompts, max_tokens=20, do_sample=False)
    oracle = OraclePipeline(OracleLM(fp8_ref.dequantized(tgt), "bf16"), OracleLM(fp8_ref.dequantized(drf), "bf16"), k=k,
                            eos_token_id=tgt.config.eos_token_id)
    want = oracle.generate_batch(prompts, 20)
    for b in range(batch):
        assert got[b]["generated_tokens"] == want[b]["generated_tokens"], (k, b)
        assert (got[b]["proposed"], got[b]["accepted"]) == (want[b]["proposed"], want[b]["accepted"])


def test_fp8_refusals():
    from specdec_hip import _abi, weights as W
    from specdec_hip.engine import HipModel

    cfg = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=1, d_model=96, n_heads=3, n_kv_heads=1, head_dim=32, d_ff=160, vocab=200,
                        max_pos=64, rope_theta=10000.0, rope_scaling=None, tie_embeddings=False, name="odd")
    mw = W.synthetic_llama(cfg, seed=0, device="cuda")
    HipModel(mw, batch=1, l_max=32)                                  # bf16 covers the shape
    with pytest.raises(_abi.HipLibraryError, match="fp8 storage does not cover"):
        HipModel(mw, batch=1, l_max=32, weight_dtype="fp8")
    with pytest.raises(ValueError):
        HipModel(mw, batch=1, l_max=32, weight_dtype="int4")
