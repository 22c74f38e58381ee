"""Wrapper-level pieces on the GPU: HipLM (LanguageModel contract), the scheduler, the HIP-backed
policy, and BASELINE config 0 (GPT-2 target + DistilGPT2 draft shapes, K=2, batch 1)."""

import json
import os

import pytest
import torch

import cases
from helpers import synthetic_prompts, tiny_pair
from oracle.model_ref import OracleLM
from oracle.pipeline_ref import OraclePipeline

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "hostlogic_golden.json")


def test_longest_prefix_policy_on_gpu_logits_matches_reference():
    from src.specdec.policies.policies import create_policy

    with open(GOLD) as f:
        gold = json.load(f)
    pol = create_policy("longest_prefix")
    for row in gold["policies"]:
        dl, bl, d_ids, b_ids = cases.build_policy_case(row["K"], row["V"], row["seed"])
        a, info = pol.accept_tokens(d_ids.cuda(), b_ids.cuda(), dl.cuda(), bl.cuda())
        assert a == row["longest_prefix"][0] and info["verify_backend"] == "hip"
        # logits longer than the proposal (K+1 positions, as the parallel verify returns them)
        bl2 = torch.cat([bl, torch.randn(1, 1, row["V"])], 1)
        assert pol.accept_tokens(d_ids.cuda(), b_ids.cuda(), dl.cuda(), bl2.cuda())[0] == row["longest_prefix"][0]


def test_hiplm_generate_tokens_contract_and_prefix_reuse():
    """generate_tokens(input_ids, k) -> (ids [B,k] int64, logits [B,k,V]); greedy ids equal the
    oracle's (hf_wrappers.py:272-627 semantics); a longer input that extends the cached prefix
    only computes the new suffix and gives the same continuation."""
    from src.specdec import HipLM

    drf, tgt = tiny_pair()
    lm = HipLM(tgt.to("cuda"))
    olm = OracleLM(tgt, "bf16")
    ids = synthetic_prompts(2, 10, tgt.config.vocab)
    got_ids, got_logits = lm.generate_tokens(ids, 6, temperature=1.0, do_sample=False)
    want_ids, want_logits = olm.generate_tokens(ids, 6)
    assert got_ids.dtype == torch.int64 and got_ids.shape == (2, 6) and got_logits.shape == (2, 6, tgt.config.vocab)
    assert torch.equal(got_ids.cpu(), want_ids)
    rel = (got_logits.float().cpu() - want_logits).abs().max() / want_logits.abs().max()
    assert rel < 0.03
    longer = torch.cat([ids, want_ids[:, :3]], 1)
    again, _ = lm.generate_tokens(longer, 3, do_sample=False)
    assert torch.equal(again.cpu(), want_ids[:, 3:6])
    info = lm.get_tokenizer_info()
    assert info["vocab_size"] == tgt.config.vocab and lm.device == "cuda" and lm.supports_kv_append()
    kv = lm.get_kv_cache()
    assert kv is not None and kv.get_num_layers() == tgt.config.n_layers
    assert kv.past_key_values[0][0].shape[1:] == (tgt.config.n_kv_heads, kv.seq_len, tgt.config.head_dim)
    assert kv.past_key_values[0][1].shape == kv.past_key_values[0][0].shape
    assert lm.decode(lm.encode("5 6 7")) == "5 6 7"
    lm.clear_kv_cache()
    assert lm.get_kv_cache() is None


def test_hiplm_kv_hook_on_paged_engine_equals_dense():
    """get_kv_cache() of a paged engine gathers the rows' pages into dense [B,H,n,D] tensors: equal, bit for bit, to the
    views a dense engine returns for the same tokens (the paged forward is bit-identical to the dense one); page-crossing
    lengths; clear_kv_cache() hands the pages back to the pool."""
    from src.specdec import HipLM

    drf, tgt = tiny_pair()
    ids = synthetic_prompts(2, 37, tgt.config.vocab)       # 37 + 6 positions: crosses a 32-position page
    dense = HipLM(tgt.to("cuda"), batch=2)
    paged = HipLM(tgt.to("cuda"), batch=2, kv_page_len=32)
    d_ids, _ = dense.generate_tokens(ids, 6, do_sample=False)
    p_ids, _ = paged.generate_tokens(ids, 6, do_sample=False)
    assert torch.equal(d_ids, p_ids)
    kd, kp = dense.get_kv_cache(), paged.get_kv_cache()
    assert kd.seq_len == kp.seq_len and kp.get_num_layers() == tgt.config.n_layers
    for (k0, v0), (k1, v1) in zip(kd.past_key_values, kp.past_key_values):
        assert k1.shape == k0.shape and v1.shape == v0.shape
        assert torch.equal(k0, k1) and torch.equal(v0, v1)
    assert paged._model.pages_in_use() > 0
    paged.clear_kv_cache()
    assert paged._model.pages_in_use() == 0 and paged.get_kv_cache() is None


def test_scheduler_parallel_verify_matches_autoregressive_contract():
    """schedule_verification returns base tokens/logits such that position i is the target's
    greedy prediction after prefix + d_1..d_i: on the accepted prefix identical to K
    autoregressive base steps (speculative_scheduler.py:192-199)."""
    from src.scheduler import create_speculative_scheduler
    from src.specdec import HipLM
    from src.specdec.policies.policies import create_policy

    drf, tgt = tiny_pair(flip_fraction=0.25)
    base, draft = HipLM(tgt.to("cuda")), HipLM(drf.to("cuda"))
    sched = create_speculative_scheduler(device="cuda")
    pol = create_policy("longest_prefix")
    o_base, o_draft = OracleLM(tgt, "bf16"), OracleLM(drf, "bf16")
    ids = synthetic_prompts(1, 9, tgt.config.vocab, seed=77)
    for _ in range(4):
        d_ids, d_logits = draft.generate_tokens(ids, 4, do_sample=False)
        b_ids, b_logits, info = sched.schedule_verification(base, d_ids, ids, temperature=1.0, do_sample=False)
        assert info["method"] == "parallel_verify" and b_logits.shape[1] == 5
        a, pinfo = sched.apply_acceptance_policy(pol, d_ids, b_ids, d_logits, b_logits)
        want_d, _ = o_draft.generate_tokens(ids, 4)
        want_b, _ = o_base.generate_tokens(ids, 4)
        assert torch.equal(d_ids.cpu(), want_d)
        want_a = 0
        while want_a < 4 and int(want_d[0, want_a]) == int(want_b[0, want_a]):
            want_a += 1
        assert a == want_a and pinfo["verify_backend"] == "hip"
        assert torch.equal(b_ids[0, :a].cpu(), want_b[0, :a])
        nxt = want_b[:, : a + 1] if a < 4 else want_b
        ids = torch.cat([ids, nxt], 1)
    assert sched.get_metrics()["total_proposed"] == 16


def test_config0_gpt2_distilgpt2_shapes_k2_batch1():
    """BASELINE.json configs[0]: GPT-2 target + DistilGPT2 draft (shapes; synthetic weights), K=2,
    batch 1 — generate_batch and generate equal the oracle token for token."""
    from specdec_hip import weights as W
    from src.specdec import HipLM, SpeculativePipeline

    tgt = W.synthetic_gpt2(W.GPT2_SMALL, seed=5)
    drf = W.synthetic_gpt2(W.DISTILGPT2, seed=6, embed_from=tgt, flip_fraction=0.3)
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda")), draft_lm=HipLM(drf.to("cuda")),
                               controller="fixed", controller_params={"k": 2}, seed=1234)
    prompt = synthetic_prompts(1, 12, W.GPT2_SMALL.vocab)[0].tolist()
    oracle = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=2, eos_token_id=W.GPT2_SMALL.eos_token_id)
    got = pipe.generate_batch([prompt], max_tokens=16, do_sample=False)[0]
    want = oracle.generate_batch([prompt], 16)[0]
    assert got["generated_tokens"] == want["generated_tokens"]
    assert (got["proposed"], got["accepted"]) == (want["proposed"], want["accepted"])
    assert 0 < got["accepted"] and len(set(got["generated_tokens"])) > 8
    gs, ws = pipe.generate(prompt, max_tokens=12, do_sample=False), oracle.generate(prompt, 12)
    assert gs["generated_tokens"] == ws["generated_tokens"] and gs["steps"] == ws["steps"]
