"""Prompt prefill as GEMMs (csrc/prefill_gemm.hip: passes of >= 96 tokens of a Llama model with bf16 weights and dense KV — one
library GEMM per matrix product of a <= 512-position chunk, this repo's RMSNorm / fused-epilogue / attention kernels around it)
against (a) the CPU oracle and (b) the 128-token passes of the decode-shaped kernels it replaces for prompts.

Reference being replaced: the first full-prefix forward, /root/reference/src/specdec/models/hf_wrappers.py:417.
Tolerances: the bf16 rounding points are those of every other path (the GEMM's fp32 products go through the same fused epilogues);
fp32 sums differ in order. Against the oracle: the full-shape bound of tests/test_hip_fullshape_parity_gpu.py (max error < 3 % of the
logit range, rms < 1.5 %). Between the two device paths: K / V rows and residual rows within the per-element bound of
tests/test_hip_persist_gpu.py, next-token ids equal wherever the top-2 margin exceeds twice the logit difference."""

import dataclasses

import pytest
import torch

from helpers import synthetic_prompts
from oracle.model_ref import OracleLM
from specdec_hip import weights as W
from test_hip_persist_gpu import TOY, TOY128, _close, _dev, _shape_1b, _shape_3b

pytestmark = pytest.mark.gpu


def _model(mw, l_max, batch=1):
    from specdec_hip.engine import HipModel

    return HipModel(mw.to("cuda") if mw.tok_emb.device.type != "cuda" else mw, batch=batch, l_max=l_max)


@pytest.mark.parametrize("cfg,L", [(TOY, 200), (TOY128, 333), (TOY, 700)], ids=["toy-200", "toy128-333", "toy-700-two-chunks"])
def test_gemm_prefill_matches_the_oracle(cfg, L, monkeypatch):
    """Logits of the position after an L-token prompt absorbed by the GEMM path, against the bf16 oracle's full-prefix forward."""
    mw = W.synthetic_llama(cfg, seed=3, device="cpu", layer_gain=0.05)
    seq = synthetic_prompts(1, L + 1, cfg.vocab, seed=7)
    want, _ = OracleLM(mw, "bf16").forward(seq)
    hm = _model(mw, L + 64)
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    hm.forward(_dev(seq[:, :L]), zero, 0, skip_head=True)                       # GEMM path (L >= 96, no logits asked for)
    pos = torch.tensor([L], dtype=torch.int32, device="cuda")
    ids, got = hm.forward(_dev(seq[:, L:]), pos, 0, want_logits=True)
    w = want[0, L].float()
    g = got[0, 0].float().cpu()
    rng = (w.max() - w.min()).item()
    err = (g - w).abs()
    assert err.max().item() < 0.03 * rng and err.pow(2).mean().sqrt().item() < 0.015 * rng, (err.max().item() / rng, rng)
    top2 = w.topk(2).values
    if (top2[0] - top2[1]).item() > 2 * err.max().item():
        assert int(ids[0, 0]) == int(w.argmax())


@pytest.mark.parametrize("shape,L", [(_shape_1b(2), 300), (_shape_3b(2), 640)], ids=["1b-2l-300", "3b-2l-640-two-chunks"])
def test_gemm_prefill_matches_the_128_token_passes_at_production_shapes(shape, L, monkeypatch):
    """The same prompt through both prefill paths at the real layer dimensions (plain random weights): the caches they leave, the
    residual rows of the last positions, the ids of the prompt positions and the next position's logits."""
    mw = W.random_init(dataclasses.replace(shape, vocab=32000), seed=11, device="cuda")
    seq = synthetic_prompts(1, L + 1, 32000, seed=9)
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    pos = torch.tensor([L], dtype=torch.int32, device="cuda")
    res = {}
    for path in ("gemm", "passes"):
        if path == "passes":
            monkeypatch.setenv("SPECDEC_NO_GEMM_PREFILL", "1")
        hm = _model(mw, L + 64)
        ids_p, _ = hm.forward(_dev(seq[:, :L]), zero, 0)                          # ids of every prompt position, no logits
        hid = hm.hidden_rows(min(L, 128) if path == "gemm" else (L - 1) % 128 + 1)
        k, v = hm.kv_view()
        ids_n, lg = hm.forward(_dev(seq[:, L:]), pos, 0, want_logits=True)
        res[path] = (ids_p.cpu(), hid.float().cpu(), k[:, :, :, :L].float().cpu(), v[:, :, :, :, :L].float().cpu(), ids_n.cpu(), lg.float().cpu())
    monkeypatch.delenv("SPECDEC_NO_GEMM_PREFILL")
    a, b = res["gemm"], res["passes"]
    n = min(a[1].shape[0], b[1].shape[0])
    _close(a[1][-n:], b[1][-n:], "residual rows of the last positions")
    _close(a[2], b[2], "K rows of the prompt")
    _close(a[3], b[3], "V rows of the prompt")
    _close(a[5], b[5], "logits of the next position", floor=b[5].abs().max().item() / 8)
    # ids: equal wherever the 128-token path's own top-2 margin is clear of the difference between the two paths
    band = (a[5] - b[5]).abs().max().item()
    top2 = b[5][0, 0].topk(2).values
    if (top2[0] - top2[1]).item() > 2 * band:
        assert torch.equal(a[4], b[4])
    agree = (a[0] == b[0]).float().mean().item()
    assert agree > 0.97, f"only {agree:.3f} of the prompt positions' ids agree between the two prefill paths"


def test_gemm_prefill_feeds_the_step_loop(monkeypatch):
    """A 400-token prompt through the pipeline: the session's prefill takes the GEMM path for both models and the decoded tokens
    are the oracle's (greedy, K = 4) — the caches the GEMM path leaves are the caches the captured step continues from."""
    from oracle.pipeline_ref import OraclePipeline
    from src.specdec import HipLM, SpeculativePipeline

    tcfg = dataclasses.replace(TOY, max_pos=1024)
    dcfg = dataclasses.replace(TOY, n_layers=1, d_model=128, n_heads=2, n_kv_heads=1, d_ff=256, max_pos=1024, name="persist-toy-draft")
    tgt = W.synthetic_llama(tcfg, seed=3, device="cpu", layer_gain=0.05)
    drf = W.synthetic_llama(dcfg, seed=4, device="cpu", layer_gain=0.05, embed_from=tgt, flip_fraction=0.25)
    prompt = synthetic_prompts(1, 400, tcfg.vocab, seed=21)[0].tolist()
    want = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4).generate_batch([prompt], 24)[0]
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda")), draft_lm=HipLM(drf.to("cuda")), controller="fixed", controller_params={"k": 4}, seed=1234)
    got = pipe.generate_batch([prompt], max_tokens=24, do_sample=False)[0]
    assert got["generated_tokens"] == want["generated_tokens"]
    assert (got["proposed"], got["accepted"]) == (want["proposed"], want["accepted"])
