"""Full-DEPTH logit parity against the bf16 oracle, with the error as a function of depth and a divergence report.

tests/test_hip_fullshape_parity_gpu.py checks every kernel instantiation at the real layer dimensions on two layers. Here the
whole models run — Llama-3.2-1B x 16 layers, Llama-3.2-3B x 28, Llama-3-8B x 32 in bf16 and with fp8 weight storage, vocabulary
128256 — on plain random weights (specdec_hip.weights.random_init: N(0, 0.02), no successor structure, no damped layers, i.e.
no margin engineering), so that (a) the growth of the device-vs-oracle error over the layers is measured, not assumed, and
(b) the first position at which greedy decoding on the device leaves the oracle's token sequence is REPORTED together with the
oracle's own top-2 logit margin there ("hard parts": a flip between near-ties is a property of bf16
arithmetic under a different summation order; a flip at a clear margin is a bug).

What the reference compares at this boundary: LongestPrefixPolicy.accept_tokens on argmax(base_logits),
/root/reference/src/specdec/policies/policies.py:156-180 — an argmax flip in the target changes the accepted prefix.

Tolerances, stated: at every depth d the logits of three new positions behind a 12-token prefix satisfy
  RMS(device - oracle) <= (0.008 + 0.0022 d) x RMS(oracle)   and   max|device - oracle| <= 4 x that bound x RMS(oracle)
i.e. a per-layer relative error budget of 0.22 % on top of 0.8 % for the head. Measured (round 3, printed by the test): the
error grows like a random walk, ~sqrt(d): 1B 0.8 % (2 layers) -> 2.0 % (16); 3B 0.8 % -> 3.2 % (28); 8B 0.9 % -> 4.5 % (32),
fp8 storage 4.7 %; the largest single logit error 9 % / 19 % / 26 % of RMS(logits). A wrong layer is O(1).
Greedy run: tokens must agree up to the first position where the oracle's top-2 margin is below 6 x the measured RMS logit
error of the full-depth model; a divergence at a larger margin fails. (Why 6 x e_rms and not a multiple of the MAX error: a flip
between tokens a and b needs |err_a - err_b| > margin with both errors of the size of e_rms — the difference of two of them has
standard deviation sqrt(2) e_rms, so 6 e_rms is a > 4 sigma event; the largest error over 128256 x 3 logits is a 5-6 sigma
outlier of the same distribution and a band built on it (round 3: 4 x e_max = 0.75 / 1.02 x RMS(logits) at 3B / 8B) admitted
every margin that occurs. Bands now: 1B 0.12, 3B 0.19, 8B 0.27 x RMS(logits).)
Run time: the CPU oracle's forwards dominate, so the depth sweep is {2, L} (round 3 also ran L / 2: the sqrt(d) growth is on
file in profiles/round3 logs and in DESIGN section 4) and the greedy runs are as long as the first divergence seen in round 3
needs (1B: token 7, 3B: token 14)."""

import dataclasses
import os

import pytest
import torch

from oracle import fp8_ref
from oracle.model_ref import OracleLM
from specdec_hip import weights as W

pytestmark = pytest.mark.gpu

MODELS = {
    # name: (config, weight dtype, tokens of the greedy run)
    "llama-3.2-1b-16L": (W.LLAMA_3_2_1B, "bf16", 12),
    "llama-3.2-3b-28L": (W.LLAMA_3_2_3B, "bf16", 10),
    "llama-3-8b-32L-bf16": (W.LLAMA_3_8B, "bf16", 6),
    "llama-3-8b-32L-fp8": (W.LLAMA_3_8B, "fp8", 4),
}
PREFIX, NEW = 12, 3
# Llama-3-8B x 32 layers in bf16: the same kernels, depth and oracle cost as the fp8 case below it (which adds the fp8 stream); kept
# out of the default run so that the driver's `-m gpu` step stays well inside its limit (round 3: 602 of 900 s)
SLOW = {"llama-3-8b-32L-bf16"}


def _truncate(mw: W.ModelWeights, depth: int) -> W.ModelWeights:
    """The first `depth` layers of `mw` (same tensors) under the same embedding, final norm and head."""
    out = dataclasses.replace(mw, config=dataclasses.replace(mw.config, n_layers=depth), layers=list(mw.layers[:depth]), meta={})
    return out


def _rel(got, want):
    d = (got - want).double()
    rms_w = want.double().pow(2).mean().sqrt().item()
    return d.abs().max().item() / rms_w, d.pow(2).mean().sqrt().item() / rms_w


@pytest.mark.parametrize("name", list(MODELS))
def test_full_depth_logits_and_first_divergence(name):
    from specdec_hip.engine import HipModel

    if name in SLOW and not os.environ.get("SPECDEC_RUN_SLOW"):
        pytest.skip(f"{name}: ~45 s of CPU-oracle forwards; the fp8 variant of the same model runs by default (SPECDEC_RUN_SLOW=1 runs both)")

    base_cfg, wdt, n_greedy = MODELS[name]
    cfg = dataclasses.replace(base_cfg, max_pos=512)
    L = cfg.n_layers
    mw_dev = W.random_init(cfg, seed=17, device="cuda")
    mw_cpu = mw_dev.to("cpu")
    if wdt == "fp8":
        mw_cpu = fp8_ref.dequantized(mw_cpu)
    g = torch.Generator().manual_seed(23)
    seq = torch.randint(4, cfg.vocab, (1, PREFIX + NEW), generator=g)

    # ---- (a) error against depth
    # (the CPU oracle's forwards are the slow part of this test; two layers at these dimensions are tests/test_hip_fullshape_parity_gpu.py's
    #  subject, so the shallow point is kept for the smallest model only)
    depths = [2, L] if L <= 16 else [L]
    rows = []
    for d in depths:
        lm = OracleLM(_truncate(mw_cpu, d), "bf16")
        _, past = lm.forward(seq[:, :PREFIX], need_logits=False)
        want, _ = lm.forward(seq[:, PREFIX:], past)
        hm = HipModel(_truncate(mw_dev, d), batch=1, l_max=128, weight_dtype=wdt)
        zero = torch.zeros(1, dtype=torch.int32, device="cuda")
        hm.forward(seq[:, :PREFIX].to(torch.int32).cuda(), zero, 0, skip_head=True)
        pos = torch.tensor([PREFIX], dtype=torch.int32, device="cuda")
        _, got = hm.forward(seq[:, PREFIX:].to(torch.int32).cuda(), pos, 0, want_logits=True)
        assert hm.engine_status() == 0
        e_max, e_rms = _rel(got.float().cpu()[0], want[0])
        rows.append((d, e_max, e_rms))
        del hm
        if d != L:
            del lm   # (the full-depth oracle, with its fp32 copies of the weights, also serves the greedy run below)
    print(f"\n[fulldepth] {name}: logits of {NEW} positions behind a {PREFIX}-token prefix, error relative to RMS(oracle logits)")
    for d, e_max, e_rms in rows:
        bound = 0.008 + 0.0022 * d
        print(f"[fulldepth]   depth {d:3d}: rms {e_rms:.4f}  max {e_max:.4f}   (bounds {bound:.4f} / {4 * bound:.4f})")
        assert e_rms <= bound and e_max <= 4 * bound, (name, d, e_rms, e_max)
    full_max, full_rms = rows[-1][1], rows[-1][2]
    band = 6 * full_rms   # see the module docstring

    # ---- (b) greedy decode: first divergence and the oracle's top-2 margin there
    prompt = seq[:, :PREFIX]
    want_ids, want_logits = lm.generate_tokens(prompt, n_greedy)
    rms_logit = want_logits.double().pow(2).mean().sqrt().item()
    hm = HipModel(mw_dev, batch=1, l_max=256, weight_dtype=wdt)
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    hm.forward(prompt[:, :-1].to(torch.int32).cuda(), zero, 0, skip_head=True)
    cur = prompt[:, -1:].to(torch.int32).cuda()
    pos = torch.tensor([PREFIX - 1], dtype=torch.int32, device="cuda")
    got = []
    for _ in range(n_greedy):
        cur, _ = hm.forward(cur, pos, 0)       # 1-token passes: the persistent launch where the model is eligible
        got.append(int(cur[0, 0]))
        pos = pos + 1
    assert hm.engine_status() == 0
    want = want_ids[0].tolist()
    first = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
    top2 = want_logits[0].topk(2, dim=-1).values
    margins = ((top2[:, 0] - top2[:, 1]) / rms_logit).tolist()       # oracle's top-2 margin per position, in units of RMS(logits)
    if first is None:
        print(f"[fulldepth]   greedy: all {n_greedy} tokens equal the oracle's; smallest oracle top-2 margin on the way "
              f"{min(margins):.4f} x RMS(logits) (device rms / max logit error {full_rms:.4f} / {full_max:.4f}, flip band {band:.4f})")
    else:
        print(f"[fulldepth]   greedy: first divergence at token {first} of {n_greedy}: device {got[first]} vs oracle {want[first]}, "
              f"oracle top-2 margin there {margins[first]:.4f} x RMS(logits) (device rms / max logit error {full_rms:.4f} / {full_max:.4f}, flip band {band:.4f}); "
              f"smallest margin before it {min(margins[:first], default=float('nan')):.4f}")
        # a flip is admissible only between near-ties: margin within 6 x the measured RMS logit error of this model
        assert margins[first] <= band, (name, first, margins[first], full_rms)
        # and the token the device picked must itself be a near-tie of the oracle's maximum (bf16 logits tie many ways)
        gap = (want_logits[0, first].max() - want_logits[0, first, got[first]]).item() / rms_logit
        assert gap <= band, (name, first, gap, full_rms)


def test_full_depth_8b_specdec_is_the_targets_greedy_continuation():
    """BASELINE configs 4 / 5 at full depth (8B target, 32 layers, untied 1.05 GB head; 1B draft; K = 4, 4 rows): the loop's
    output is the target's own greedy continuation, row for row (the full-size property test of tests/test_full_size_gpu.py,
    which runs the 3B + 1B pair)."""
    from src.specdec import HipLM, SpeculativePipeline
    from helpers import synthetic_prompts

    tgt = W.synthetic_llama(W.LLAMA_3_8B, seed=0, device="cuda")
    drf = W.synthetic_llama(W.LLAMA_3_2_1B, seed=1, device="cuda", embed_from=tgt, flip_fraction=0.2)
    target_lm, draft_lm = HipLM(tgt), HipLM(drf)
    prompts = synthetic_prompts(4, 32, target_lm.vocab_size).tolist()
    pipe = SpeculativePipeline(base_lm=target_lm, draft_lm=draft_lm, controller="fixed", controller_params={"k": 4}, seed=1234)
    got = pipe.generate_batch(prompts, max_tokens=24, do_sample=False)
    for b in range(2):
        ids, _ = target_lm.generate_tokens(torch.tensor([prompts[b]]), 29, do_sample=False)
        g = got[b]["generated_tokens"]
        assert len(g) >= 24 and g == ids[0].tolist()[: len(g)], b
    assert sum(r["accepted"] for r in got) / sum(r["proposed"] for r in got) > 0.25
