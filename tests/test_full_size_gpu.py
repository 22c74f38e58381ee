"""BASELINE.json's full sizes (Llama-3.2-3B target + 1B draft, bf16) through size-independent properties —
the CPU oracle cannot run these shapes in test time (bench.py's cpu_baseline leg re-checks a bounded sample).

Property (greedy speculative decoding): whatever the draft proposes, the emitted sequence is the target's
own greedy continuation; with the bonus rule of generate_batch every step emits accept_len + 1 of them.
So the pipeline's tokens must equal those of the SAME target decoded one token at a time through
`generate_tokens` (M = 1 forwards: another kernel path — 1-token GEMV passes vs 5/40-token verify passes),
at batch 1 (config 2) and batch 8 (config 3), for every K."""

import pytest
import torch

from helpers import synthetic_prompts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_pair():
    from specdec_hip import weights as W
    from src.specdec import HipLM

    tgt = W.synthetic_llama(W.LLAMA_3_2_3B, seed=0, device="cuda")
    drf = W.synthetic_llama(W.LLAMA_3_2_1B, seed=1, device="cuda", embed_from=tgt, flip_fraction=0.2)
    return HipLM(drf), HipLM(tgt)


def _reference_greedy(target_lm, prompts, n):
    out = []
    for p in prompts:
        ids, _ = target_lm.generate_tokens(torch.tensor([p]), n, do_sample=False)
        out.append(ids[0].tolist())
    return out


@pytest.mark.parametrize("k,batch", [(4, 1), (4, 8), (1, 8), (8, 8), (2, 3), (2, 8)])
def test_specdec_output_is_the_targets_greedy_continuation(full_pair, k, batch):
    from src.specdec import SpeculativePipeline

    draft_lm, target_lm = full_pair
    V = target_lm.vocab_size
    prompts = synthetic_prompts(batch, 32, V).tolist()
    n = 40
    pipe = SpeculativePipeline(base_lm=target_lm, draft_lm=draft_lm, controller="fixed", controller_params={"k": k}, seed=1234)
    got = pipe.generate_batch(prompts, max_tokens=n, do_sample=False)
    want = _reference_greedy(target_lm, prompts[: min(batch, 3)], n + k + 1)
    for b, w in enumerate(want):
        g = got[b]["generated_tokens"]
        assert len(g) >= n
        assert g == w[: len(g)], (k, batch, b)
    for r in got:
        # counters: every step proposes k and emits accept_len + 1 tokens (bonus counted, pipeline.py:3433)
        steps = r["batch_metrics"]["total_steps"]
        assert r["proposed"] % k == 0 and r["proposed"] // k <= steps
        assert r["accepted"] == len(r["generated_tokens"])
        assert r["proposed"] // k <= r["accepted"] <= (k + 1) * (r["proposed"] // k)
    acc = sum(r["accepted"] for r in got) / sum(r["proposed"] for r in got)
    assert acc > 1.0 / k   # the draft shares 80 % of the successors: more than the bonus token alone is accepted
    # rows are independent: row 1 of the batch equals the same prompt decoded alone
    if batch > 1:
        alone = pipe.generate_batch([prompts[1]], max_tokens=n, do_sample=False)[0]
        assert alone["generated_tokens"] == got[1]["generated_tokens"]


def test_kv_append_is_position_exact_at_full_size(full_pair):
    """Round trip through the fused in-place KV append: decoding 24 tokens one by one, then re-prefilling
    the same 24 tokens in one multi-token pass into another cache row-set, must give the same next token
    and the same logits up to accumulation order (1 % of the range) — the cache written by 1-token passes
    and by a 24-token pass is interchangeable."""
    _, target_lm = full_pair
    V = target_lm.vocab_size
    prompt = synthetic_prompts(1, 16, V)[0].tolist()
    ids, logits = target_lm.generate_tokens(torch.tensor([prompt]), 24, do_sample=False)
    seq = prompt + ids[0].tolist()
    eng = target_lm.new_engine(1, 128)
    toks = torch.tensor([seq[:-1]], dtype=torch.int32, device="cuda")
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    got_ids, got_logits = eng.forward(toks, zero, 0, want_logits=True)
    assert eng.pass_tokens in (64, 128)
    # position i of the one-pass prefill predicts seq[i+1]
    assert got_ids[0, len(prompt) - 1:].cpu().tolist() == seq[len(prompt):]
    a, b = got_logits[0, -1].float().cpu(), logits[0, -1].float().cpu()
    assert (a - b).abs().max().item() / b.abs().max().item() < 0.01


def test_fp8_storage_at_full_size_multi_token_passes():
    """fp8 weight storage on the 3B + 1B pair, batch 8 (40-token verify passes through the multi-token kernel
    streaming fp8): the output is the fp8 target's own greedy continuation (1-token gemv.hip passes)."""
    from specdec_hip import weights as W
    from src.specdec import HipLM, SpeculativePipeline

    tgt = W.synthetic_llama(W.LLAMA_3_2_3B, seed=0, device="cuda")
    drf = W.synthetic_llama(W.LLAMA_3_2_1B, seed=1, device="cuda", embed_from=tgt, flip_fraction=0.2)
    target_lm, draft_lm = HipLM(tgt, weight_dtype="fp8"), HipLM(drf, weight_dtype="fp8")
    assert target_lm.new_engine(1, 64).pass_tokens in (64, 128)
    prompts = synthetic_prompts(8, 32, target_lm.vocab_size).tolist()
    pipe = SpeculativePipeline(base_lm=target_lm, draft_lm=draft_lm, controller="fixed", controller_params={"k": 4}, seed=1234)
    got = pipe.generate_batch(prompts, max_tokens=32, do_sample=False)
    want = _reference_greedy(target_lm, prompts[:2], 40)
    for b, w in enumerate(want):
        g = got[b]["generated_tokens"]
        assert len(g) >= 32 and g == w[: len(g)], b


def test_device_selected_first_draft_forward_changes_nothing(full_pair, monkeypatch):
    """One sequence with a persistent draft: the captured step holds draft forward 0 twice — the 2-token pass over (prev, last)
    and the 1-token pass over `last` — and the device runs the one the previous step's accept length calls for
    (engine.hip enqueue_step, misc.hip accept_kernel: prev's K/V are missing from the draft cache only after a fully accepted
    step). The proposals must not depend on it: same tokens, same step count, same proposed / accepted counters as with the
    selection switched off (always the 2-token pass), over a run that contains fully and partly accepted steps."""
    from src.specdec import SpeculativePipeline

    draft_lm, target_lm = full_pair
    prompts = synthetic_prompts(1, 32, target_lm.vocab_size, seed=11).tolist()
    k, runs = 4, {}
    for select in (True, False):
        if not select:
            monkeypatch.setenv("SPECDEC_NO_FWD0_SELECT", "1")
        pipe = SpeculativePipeline(base_lm=target_lm, draft_lm=draft_lm, controller="fixed", controller_params={"k": k}, seed=1234)
        r = pipe.generate_batch(prompts, max_tokens=96, do_sample=False)[0]
        runs[select] = (r["generated_tokens"], r["batch_metrics"]["total_steps"], r["proposed"], r["accepted"])
    monkeypatch.delenv("SPECDEC_NO_FWD0_SELECT")
    assert runs[True] == runs[False]
    _, steps, proposed, accepted = runs[True]
    assert steps < accepted < steps * (k + 1), "the run must mix fully and partly accepted steps"


def test_persistent_draft_proposes_what_the_launch_path_proposes(full_pair, monkeypatch):
    """The output of greedy speculative decoding is the target's continuation whatever the draft proposes, so the tests above
    cannot see a persistent draft forward (csrc/persist.hip) that proposed worse tokens — only acceptance would drop. Here the
    PROPOSALS are pinned at full size: the same run with the draft on the persistent launch (default: 1- and 2-token passes of
    the 1B model) and with every draft pass on the launch path (SPECDEC_PERSIST_MAX_T=0, read when an engine binds its cache)
    must agree on tokens, steps, proposed and accepted, and on every step's accept length — a different proposal anywhere changes
    an accept length. What the reference compares at this boundary: LongestPrefixPolicy.accept_tokens,
    /root/reference/src/specdec/policies/policies.py:156-180."""
    from src.specdec import SpeculativePipeline

    draft_lm, target_lm = full_pair
    prompts = synthetic_prompts(1, 32, target_lm.vocab_size, seed=5).tolist()
    k, runs = 4, {}
    for persist in (True, False):
        if not persist:
            monkeypatch.setenv("SPECDEC_PERSIST_MAX_T", "0")
        pipe = SpeculativePipeline(base_lm=target_lm, draft_lm=draft_lm, controller="fixed", controller_params={"k": k}, seed=1234)
        sess = pipe.start_session(prompts, max_tokens=96, emit_mode=0)
        assert bool(sess.rt["draft"].persist_active(1)) == persist
        while sess.any_active() and sess.advance():
            pass
        sess.finish()
        r = sess.rows[0]
        runs[persist] = (list(r.generated), r.steps, r.proposed, r.accepted, [c[3] for c in r.counters])
    monkeypatch.delenv("SPECDEC_PERSIST_MAX_T")
    assert runs[True] == runs[False]
    lens = runs[True][4]
    assert 0 < sum(1 for a in lens if a == k) < len(lens), "the run must mix fully and partly accepted steps"
