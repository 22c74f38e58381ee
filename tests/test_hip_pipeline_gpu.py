"""SpeculativePipeline on the GPU vs the oracle restatement of the reference loop and vs
the reference's own traces (tests/golden/pipeline_golden.json). Token ids must be identical."""

import json
import os

import pytest
import torch

import cases
from helpers import synthetic_prompts, tiny_pair
from oracle.model_ref import OracleLM
from oracle.pipeline_ref import OraclePipeline

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pipe(drf, tgt, k, controller="fixed", controller_params=None):
    from src.specdec import HipLM, SpeculativePipeline

    return SpeculativePipeline(base_lm=HipLM(tgt.to("cuda")), draft_lm=HipLM(drf.to("cuda")),
                               controller=controller, controller_params=controller_params or {"k": k}, seed=1234)


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "pipeline_golden.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("pname", ["structured", "repeating"])
def test_pipeline_matches_oracle_and_reference_traces(pname, gold):
    """bf16 weights on the GPU vs (a) the oracle loop on the same bf16 weights and (b) the
    reference's fp32 CPU traces. The synthetic pairs have large argmax margins, so the
    precision does not change a single token; 'repeating' drives the de-duplication rules
    (rows rewound on the host -> cache rebuild on the device)."""
    drf, tgt = cases.g8_pairs(torch.bfloat16)[pname]
    base, draft = OracleLM(tgt, "bf16"), OracleLM(drf, "bf16")
    for run in gold[pname]["runs"]:
        k, mt, prompt = run["k"], run["max_tokens"], run["prompt_ids"]
        pipe = _pipe(drf, tgt, k)
        oracle = OraclePipeline(base, draft, k=k, eos_token_id=2)
        got = pipe.generate_batch([prompt], max_tokens=mt, do_sample=False)[0]
        want = oracle.generate_batch([prompt], mt)[0]
        assert got["generated_tokens"] == want["generated_tokens"], (pname, k, mt)
        assert (got["proposed"], got["accepted"], got["batch_metrics"]["total_steps"]) == \
               (want["proposed"], want["accepted"], want["steps"])
        assert got["sequence"] == want["sequence"]
        ref = run["batch"]
        assert got["generated_tokens"] == ref["generated_tokens"], "differs from the reference's own trace"
        assert (got["proposed"], got["accepted"], got["batch_metrics"]["total_steps"]) == \
               (ref["proposed"], ref["accepted"], ref["steps"])
        gs = pipe.generate(prompt, max_tokens=mt, do_sample=False)
        ws = oracle.generate(prompt, mt)
        assert gs["generated_tokens"] == ws["generated_tokens"] == run["single"]["generated_tokens"]
        assert (gs["proposed"], gs["accepted"], gs["steps"]) == (ws["proposed"], ws["accepted"], ws["steps"])
        assert (gs["proposed"], gs["accepted"], gs["steps"]) == \
               (run["single"]["proposed"], run["single"]["accepted"], run["single"]["steps"])


@pytest.mark.parametrize("pname", ["structured", "repeating"])
def test_generate_with_sampling_makes_the_reference_draws(pname):
    """generate(do_sample=True) (pipeline.py:1019-1027, :1217-1224): the draft's proposals and the base token of a zero-accept step
    are drawn — transformers' sampling (temperature, top-k 50) on torch's global CPU generator — and verification is greedy. Seeded
    as the reference run was, the HIP path (bf16 weights, logits from the engines, probabilities and draws on the host) must
    make the same draws in the same order as (a) the oracle on the same bf16 weights — every run — and (b) the reference's own
    fp32 CPU run (tests/golden/pipeline_sampled_golden.json) wherever bf16 weights leave the draws alone: tokens, proposed (a
    draft that drew EOS is shorter), accepted, steps."""
    with open(os.path.join(GOLD, "pipeline_sampled_golden.json")) as f:
        runs = json.load(f)[pname]["runs"]
    drf, tgt = cases.g8_pairs(torch.bfloat16)[pname]
    base, draft = OracleLM(tgt, "bf16"), OracleLM(drf, "bf16")
    left_greedy = agree = 0
    for run in runs:
        k, mt, prompt, seed, temp = run["k"], run["max_tokens"], run["prompt_ids"], run["seed"], run["temperature"]
        pipe = _pipe(drf, tgt, k)
        torch.manual_seed(seed)
        got = pipe.generate(prompt, max_tokens=mt, temperature=temp, do_sample=True)
        torch.manual_seed(seed)
        want = OraclePipeline(base, draft, k=k, eos_token_id=2).generate(prompt, mt, do_sample=True, temperature=temp)
        assert got["generated_tokens"] == want["generated_tokens"], (pname, k, temp)
        assert (got["proposed"], got["accepted"], got["steps"]) == (want["proposed"], want["accepted"], want["steps"])
        # (b): the reference ran fp32 weights, the device holds their bf16 roundings — a draw that sits within that rounding of a
        # boundary of the sampled distribution lands elsewhere (one of the six 'structured' runs: T = 12, k = 4, token 8). The
        # oracle separates the two: at fp32 it reproduces EVERY reference run (tests/test_oracle_pipeline.py), at bf16 it is what
        # the device must equal (above); where the bf16 oracle agrees with the reference, so does the device.
        ref = run["sampled"]
        same = (want["generated_tokens"], want["proposed"], want["accepted"], want["steps"]) == \
               (ref["generated_tokens"], ref["proposed"], ref["accepted"], ref["steps"])
        agree += same
        left_greedy += ref != run["greedy"]
    assert left_greedy >= 3 and agree >= len(runs) - 1
    # refused where it is not restated
    from src.specdec import HipLM, SpeculativePipeline
    pol = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda")), draft_lm=HipLM(drf.to("cuda")), controller="fixed", controller_params={"k": 2},
                              policy="conf_threshold", seed=1234)
    with pytest.raises(NotImplementedError):
        pol.generate(runs[0]["prompt_ids"], max_tokens=4, do_sample=True)


@pytest.mark.parametrize("k", [1, 2, 4, 8])
def test_k_sweep_batch8_rows_are_independent(k):
    """BASELINE config 3 shape (K sweep, batch 8) on a tiny pair: every row of the batch equals
    the same prompt run alone, and equals the oracle."""
    drf, tgt = tiny_pair(flip_fraction=0.25)
    prompts = synthetic_prompts(8, 16, tgt.config.vocab).tolist()
    pipe = _pipe(drf, tgt, k)
    got = pipe.generate_batch(prompts, max_tokens=24, do_sample=False)
    oracle = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=k, eos_token_id=tgt.config.eos_token_id)
    want = oracle.generate_batch(prompts, 24)
    for b in range(8):
        assert got[b]["generated_tokens"] == want[b]["generated_tokens"], (k, b)
        assert (got[b]["proposed"], got[b]["accepted"]) == (want[b]["proposed"], want[b]["accepted"])
    alone = pipe.generate_batch([prompts[3]], max_tokens=24, do_sample=False)[0]
    assert alone["generated_tokens"] == got[3]["generated_tokens"]
    acc = sum(r["accepted"] for r in got) / sum(r["proposed"] for r in got)
    assert 0.0 < acc <= (k + 1) / k


@pytest.mark.parametrize("kw", [{}, {"top_k": 10}, {"top_p": 0.9}, {"top_k": None, "top_p": 0.7}])
def test_generate_with_sampling_filters_against_the_oracle(kw):
    """The sampling filters of generate(do_sample=True) at a vocabulary larger than the default top-k (the reference fixtures have 160
    entries): the call's top_k / top_p reach the sampler as they reach transformers' generate in the reference
    (hf_wrappers.py:230 `generate_kwargs.update(kwargs)`); temperature 8 so that the draws leave the greedy path. Device (bf16 weights,
    host-side probabilities and draws) against the oracle on the same weights under the same seed."""
    drf, tgt = tiny_pair(flip_fraction=0.25)
    prompts = synthetic_prompts(2, 9, tgt.config.vocab, seed=77).tolist()
    pipe = _pipe(drf, tgt, 4)
    oracle = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4, eos_token_id=tgt.config.eos_token_id)
    greedy_like = 0
    for i, prompt in enumerate(prompts):
        torch.manual_seed(100 + i)
        got = pipe.generate(prompt, max_tokens=14, temperature=8.0, do_sample=True, **kw)
        torch.manual_seed(100 + i)
        want = oracle.generate(prompt, 14, do_sample=True, temperature=8.0, **{"top_k": 50, "top_p": None, **kw})
        assert got["generated_tokens"] == want["generated_tokens"], (kw, i)
        assert (got["proposed"], got["accepted"], got["steps"]) == (want["proposed"], want["accepted"], want["steps"])
        greedy_like += got["accepted"] == oracle.generate(prompt, 14)["accepted"]
    assert greedy_like < len(prompts) or kw.get("top_k") == 10, "the draws should leave the greedy path"


@pytest.mark.parametrize("rows", [16, 17])
def test_step_tail_on_both_sides_of_its_one_workgroup_limit(rows):
    """The greedy step's tail — target ids, accept scan, state advance and step record — is ONE launch (csrc/misc.hip
    verify_tail_kernel: one workgroup, wave w folds verify positions w, w + 16, ...) while B x (K+1) <= 144 positions, and the three
    separate launches beyond. 16 rows at K = 8 is the largest step the one-launch tail takes (144 positions, 9 per wave), 17 rows
    the first it does not; both against the oracle, ragged prompts so that rows finish at different steps (what LongestPrefixPolicy
    .accept_tokens and the batch rules decide per row, policies.py:156-180, pipeline.py:3059-3292)."""
    drf, tgt = tiny_pair(flip_fraction=0.25)
    g = torch.Generator().manual_seed(9)
    prompts = [torch.randint(4, tgt.config.vocab, (5 + (i % 7),), generator=g).tolist() for i in range(rows)]
    pipe = _pipe(drf, tgt, 8)
    got = pipe.generate_batch(prompts, max_tokens=14 , do_sample=False)
    oracle = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=8, eos_token_id=tgt.config.eos_token_id)
    want = oracle.generate_batch(prompts, 14)
    for b in range(rows):
        assert got[b]["generated_tokens"] == want[b]["generated_tokens"], (rows, b)
        assert (got[b]["proposed"], got[b]["accepted"]) == (want[b]["proposed"], want[b]["accepted"])


def test_ragged_prompts_and_result_keys():
    drf, tgt = tiny_pair()
    g = torch.Generator().manual_seed(4)
    prompts = [torch.randint(4, 1000, (n,), generator=g).tolist() for n in (3, 11, 7)]
    pipe = _pipe(drf, tgt, 4)
    got = pipe.generate_batch(prompts, max_tokens=10, do_sample=False)
    oracle = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4)
    want = oracle.generate_batch(prompts, 10)
    for g_, w_ in zip(got, want):
        assert g_["generated_tokens"] == w_["generated_tokens"]
    for key in ("prompt", "text", "generated_tokens", "num_generated", "batch_index", "batch_size", "latency_ms",
                "total_time_ms", "tokens_per_sec", "throughput_tokens_per_sec", "acceptance_rate", "proposed",
                "accepted", "draft_avg_ms", "verify_avg_ms", "batch_metrics", "kv_append_enabled", "kv_append_backend"):
        assert key in got[0], key
    single = pipe.generate(prompts[1], max_tokens=9, do_sample=False)
    for key in ("text", "generated_tokens", "latency_ms", "proposed", "accepted", "acceptance_rate", "tokens_per_sec",
                "steps", "verification_time_ms", "generation_time_ms", "kv_appended_tokens_total", "kv_append_time_ms",
                "kv_append_enabled", "kv_append_backend", "mem_rss_mb", "policy", "controller", "impl", "device",
                "dtype", "base_model", "draft_model", "draft_mode"):
        assert key in single, key
    assert single["generated_tokens"] == oracle.generate(prompts[1], 9)["generated_tokens"]
    assert len(single["generated_tokens"]) <= 9 and single["accepted"] <= single["proposed"]


def test_adaptive_controller_changes_k_mid_run():
    drf, tgt = tiny_pair(flip_fraction=0.0)  # draft == target successor: everything accepted -> K grows
    pipe = _pipe(drf, tgt, 2, controller="adaptive",
                 controller_params={"initial_k": 2, "min_k": 1, "max_k": 4, "target_acceptance_rate": 0.5})
    prompts = synthetic_prompts(2, 8, 1000).tolist()
    got = pipe.generate_batch(prompts, max_tokens=40, do_sample=False)
    lm = OracleLM(tgt, "bf16")
    for b in range(2):
        want, _ = lm.generate_tokens(torch.tensor([prompts[b]]), len(got[b]["generated_tokens"]))
        assert got[b]["generated_tokens"] == want[0].tolist()
    assert got[0]["batch_metrics"]["k"] > 2


@pytest.mark.parametrize("early", ["1", "0"])
def test_per_row_adaptive_k_inside_the_captured_step(early, monkeypatch):
    """sd_specdec_set_adaptive: every row's K is moved by the reference's controller rule ON THE DEVICE (no host round trip,
    steps launched ahead), the step keeps the shape max_k. Tokens, counters and the per-step k of every row equal the oracle
    (one AdaptiveKController per row); the host mirror raises if the device's k ever differs from its own replay."""
    monkeypatch.setenv("SPECDEC_EARLY_LAUNCH", early)
    drf, tgt = tiny_pair(flip_fraction=0.35)
    params = {"initial_k": 2, "min_k": 1, "max_k": 4, "step_size": 1, "target_acceptance_rate": 0.6}
    pipe = _pipe(drf, tgt, 4, controller="adaptive", controller_params=dict(params, per_row=True))
    prompts = synthetic_prompts(3, 9, 1000).tolist()
    got = pipe.generate_batch(prompts, max_tokens=40, do_sample=False)
    o = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4, eos_token_id=tgt.config.eos_token_id)
    want = o.generate_batch(prompts, 40, per_row_k=params)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g["generated_tokens"] == w["generated_tokens"]
        assert (g["proposed"], g["accepted"]) == (w["proposed"], w["accepted"])
        assert g["k_trace"] == o.k_trace[i]
    assert len({tuple(g["k_trace"]) for g in got}) > 1
    # a second run on the same (cached) loop restarts every row's controller
    again = pipe.generate_batch(prompts, max_tokens=40, do_sample=False)
    assert [g["k_trace"] for g in again] == [g["k_trace"] for g in got]
    # and generate() (draft-token emit mode, one row)
    single = pipe.generate(prompts[1], max_tokens=24, do_sample=False)
    assert single["generated_tokens"] == OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4, eos_token_id=tgt.config.eos_token_id).generate(prompts[1], 24)["generated_tokens"]


def test_per_row_adaptive_k_with_resyncs_and_admission():
    """The 'repeating' pair drives the host's de-duplication rules (rows rewound, steps launched ahead voided): the host hands
    the device its in-order controller state on every repair, so the per-step k still equals the oracle's."""
    drf, tgt = cases.g8_pairs(torch.bfloat16)["repeating"]
    params = {"initial_k": 3, "min_k": 1, "max_k": 4, "step_size": 1, "target_acceptance_rate": 0.5}
    pipe = _pipe(drf, tgt, 4, controller="adaptive", controller_params=dict(params, per_row=True))
    with open(os.path.join(GOLD, "pipeline_golden.json")) as f:
        runs = json.load(f)["repeating"]["runs"]
    prompts = [r["prompt_ids"] for r in runs[:3]]
    got = pipe.generate_batch(prompts, max_tokens=24, do_sample=False)
    o = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4, eos_token_id=2)
    want = o.generate_batch(prompts, 24, per_row_k=params)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g["generated_tokens"] == w["generated_tokens"]
        assert (g["proposed"], g["accepted"]) == (w["proposed"], w["accepted"])
        assert g["k_trace"] == o.k_trace[i]
    assert got[0]["batch_metrics"]["resyncs"] > 0, "the case no longer exercises the repair path"


def test_per_row_adaptive_k_with_continuous_batching():
    """generate_many over 2 slots with per-row adaptive K: a slot handed to a new prompt restarts that row's controller (host
    mirror and device state) while the other row keeps its own; every result equals the prompt's own per-row-K run."""
    drf, tgt = tiny_pair(flip_fraction=0.35)
    params = {"initial_k": 3, "min_k": 1, "max_k": 4, "step_size": 1, "target_acceptance_rate": 0.6}
    pipe = _pipe(drf, tgt, 4, controller="adaptive", controller_params=dict(params, per_row=True))
    g = torch.Generator().manual_seed(5)
    prompts = [torch.randint(4, tgt.config.vocab, (int(n),), generator=g).tolist() for n in (6, 15, 4, 9, 12)]
    got = pipe.generate_many(prompts, max_tokens=28, batch_size=2, do_sample=False)
    eos = pipe.base_lm.get_tokenizer_info()["eos_token_id"]
    for p, r in zip(prompts, got):
        o = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4, eos_token_id=eos)
        w = o.generate_batch([p], 28, per_row_k=params)[0]
        assert r["generated_tokens"] == w["generated_tokens"]
        assert (r["proposed"], r["accepted"]) == (w["proposed"], w["accepted"])
        assert r["k_trace"] == o.k_trace[0]


def test_loud_refusals():
    drf, tgt = tiny_pair()
    pipe = _pipe(drf, tgt, 2)
    medusa = _pipe(drf, tgt, 2)
    medusa.config["draft_mode"] = "medusa"
    with pytest.raises(NotImplementedError, match="do_sample"):
        medusa.generate([5, 6, 7], max_tokens=4, do_sample=True)   # sampling inside the self-draft modes is not restated
    from src.specdec import SpeculativePipeline

    with pytest.raises(ValueError, match="implementation"):
        SpeculativePipeline(implementation="mps")     # ("fake" is the reference's test double: tests/test_fake_lm.py)
    import specdec
    import src.specdec

    assert specdec is src.specdec and specdec.SpecDecRunner is specdec.SpeculativePipeline


def test_config1_gpt2_distilgpt2_shapes_k2():
    """BASELINE config 1: GPT-2 (12 L) target + DistilGPT2 (6 L) draft, K=2, batch 1 — full shapes
    (V = 50257 is odd, LayerNorm + biases + learned positions + gelu_new), synthetic weights. `generate`
    and `generate_batch` on the GPU equal the oracle's reference-faithful loop token for token."""
    from specdec_hip import weights as W

    tgt = W.synthetic_gpt2(W.GPT2_SMALL, seed=0, device="cpu")
    drf = W.synthetic_gpt2(W.DISTILGPT2, seed=1, device="cpu", embed_from=tgt, flip_fraction=0.3)
    prompt = synthetic_prompts(1, 12, tgt.config.vocab)[0].tolist()
    pipe = _pipe(drf, tgt, 2)
    oracle = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=2, eos_token_id=tgt.config.eos_token_id)
    got, want = pipe.generate(prompt, max_tokens=12, do_sample=False), oracle.generate(prompt, 12)
    assert got["generated_tokens"] == want["generated_tokens"]
    assert (got["proposed"], got["accepted"], got["steps"]) == (want["proposed"], want["accepted"], want["steps"])
    assert 0 < got["accepted"] < got["proposed"] + got["steps"]
    gb, wb = pipe.generate_batch([prompt], max_tokens=12, do_sample=False)[0], oracle.generate_batch([prompt], 12)[0]
    assert gb["generated_tokens"] == wb["generated_tokens"]
    assert (gb["proposed"], gb["accepted"]) == (wb["proposed"], wb["accepted"])


def test_run_specdec_cli_prints_the_reference_json(capsys, monkeypatch):
    """CLI counterpart (reference run_specdec.py): one JSON line with the reference's keys."""
    import json as _json

    from src.specdec import run_specdec
    from src.specdec.models import hip_lm

    drf, tgt = tiny_pair()
    made = {"synthetic:tiny-target": tgt, "synthetic:tiny-draft": drf}
    real = hip_lm.create_hip_lm
    monkeypatch.setattr("src.specdec.core.pipeline.create_hip_lm", lambda spec, **kw: real(made[spec].to("cuda"), **kw))
    rc = run_specdec.main(["--prompt", "5 6 7 8", "--max-tokens", "12", "--K", "2", "--base-model", "synthetic:tiny-target",
                           "--draft-model", "synthetic:tiny-draft"])
    assert rc == 0
    out = _json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert set(out) == {"latency_ms", "proposed", "accepted", "acceptance_rate", "tokens_per_sec", "text", "impl", "device",
                        "base_model", "draft_model", "draft_mode", "dtype"}
    assert out["impl"] == "hip" and out["proposed"] > 0 and len(out["text"].split()) == 12
    assert run_specdec.main(["--prompt", "5 6", "--K", "2", "--adaptive-K"]) == 1


def _oracle_pair(drf, tgt, k, eos):
    return OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=k, eos_token_id=eos)


@pytest.mark.parametrize("eos_at", [3, 4, 7, 8])
def test_eos_inside_accepted_tokens_and_as_bonus(eos_at):
    """The EOS rules of the reference (accepted EOS is cut and stops the row, a bonus EOS is kept,
    pipeline.py:3120-3131, :3274-3280): make the eos_at-th token of the greedy continuation the EOS id, so it
    lands at different offsets inside a step (accepted prefix, bonus position, first token)."""
    from src.specdec import HipLM, SpeculativePipeline
    from src.specdec.models.hip_lm import IdTokenizer

    drf, tgt = tiny_pair(flip_fraction=0.1)
    V = tgt.config.vocab
    prompts = synthetic_prompts(2, 9, V).tolist()
    free = _oracle_pair(drf, tgt, 4, None).generate_batch(prompts, 16)
    eos = free[0]["generated_tokens"][eos_at]
    want = _oracle_pair(drf, tgt, 4, eos).generate_batch(prompts, 16)
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda"), tokenizer=IdTokenizer(V, eos_token_id=eos)), draft_lm=HipLM(drf.to("cuda")),
                               controller="fixed", controller_params={"k": 4}, seed=1234)
    got = pipe.generate_batch(prompts, max_tokens=16, do_sample=False)
    for b in range(2):
        assert got[b]["generated_tokens"] == want[b]["generated_tokens"], (eos_at, b)
        assert (got[b]["proposed"], got[b]["accepted"], got[b]["sequence"]) == (want[b]["proposed"], want[b]["accepted"], want[b]["sequence"])
    assert len(want[0]["generated_tokens"]) < 16     # the row did stop at the EOS
    single = pipe.generate(prompts[0], max_tokens=16, do_sample=False)
    ws = _oracle_pair(drf, tgt, 4, eos).generate(prompts[0], 16)
    assert single["generated_tokens"] == ws["generated_tokens"] and single["steps"] == ws["steps"]


@pytest.mark.parametrize("plen,max_tokens,k", [(1, 6, 4), (2, 5, 2), (5, 1, 4), (5, 2, 8), (3, 3, 1)])
def test_tiny_prompts_and_budgets(plen, max_tokens, k):
    """One- and two-token prompts (no `prev` token for the first draft pass), budgets smaller than K."""
    drf, tgt = tiny_pair(flip_fraction=0.25)
    V = tgt.config.vocab
    prompts = synthetic_prompts(3, plen, V).tolist()
    pipe = _pipe(drf, tgt, k)
    oracle = _oracle_pair(drf, tgt, k, tgt.config.eos_token_id)
    got = pipe.generate_batch(prompts, max_tokens=max_tokens, do_sample=False)
    want = oracle.generate_batch(prompts, max_tokens)
    for b in range(3):
        assert got[b]["generated_tokens"] == want[b]["generated_tokens"], (plen, max_tokens, k, b)
        assert (got[b]["proposed"], got[b]["accepted"]) == (want[b]["proposed"], want[b]["accepted"])
    gs, ws = pipe.generate(prompts[0], max_tokens=max_tokens, do_sample=False), oracle.generate(prompts[0], max_tokens)
    assert gs["generated_tokens"] == ws["generated_tokens"] and (gs["proposed"], gs["accepted"]) == (ws["proposed"], ws["accepted"])


def test_position_limit_stops_rows_without_faults():
    """A row that reaches the model's position limit / cache capacity is stopped by the host before the device
    could index past them; the tokens emitted until then are the oracle's."""
    import dataclasses

    from helpers import TINY_DRAFT, TINY_TARGET
    from specdec_hip import weights as W

    tcfg, dcfg = dataclasses.replace(TINY_TARGET, max_pos=96), dataclasses.replace(TINY_DRAFT, max_pos=96)
    tgt = W.synthetic_llama(tcfg, seed=0, device="cpu")
    drf = W.synthetic_llama(dcfg, seed=1, device="cpu", embed_from=tgt, flip_fraction=0.25)
    prompts = synthetic_prompts(2, 40, tgt.config.vocab).tolist()
    pipe = _pipe(drf, tgt, 4)
    got = pipe.generate_batch(prompts, max_tokens=200, do_sample=False)     # would need 240 positions; the models have 96
    lm = OracleLM(tgt, "bf16")
    for b in range(2):
        g = got[b]["generated_tokens"]
        assert 30 <= len(g) and len(prompts[b]) + len(g) <= 96
        want, _ = lm.generate_tokens(torch.tensor([prompts[b]]), len(g))
        assert g == want[0].tolist()
    with pytest.raises(ValueError, match="no room"):
        pipe.generate_batch([list(range(4, 94))], max_tokens=4, do_sample=False)


@pytest.mark.parametrize("k", [2, 4])
def test_medusa_lite_tied_heads_generate(k):
    """draft_mode='medusa' (generate() only, as in the reference): heads tied to the lm_head, head 0 on the same
    hidden state for all K proposals = K copies of the target's next token (modes/medusa.py). The first proposal
    is always accepted; tokens, counters and steps equal the oracle's restatement, and the text equals plain
    greedy decoding of the target."""
    from src.specdec import HipLM, SpeculativePipeline

    drf, tgt = tiny_pair()
    prompt = synthetic_prompts(1, 10, tgt.config.vocab)[0].tolist()
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda")), draft_model="none", draft_mode="medusa", controller="fixed",
                               controller_params={"k": k}, seed=1234)
    got = pipe.generate(prompt, max_tokens=14, do_sample=False)
    lm = OracleLM(tgt, "bf16")
    want = OraclePipeline(lm, None, k=k, eos_token_id=tgt.config.eos_token_id, draft_mode="medusa_tied").generate(prompt, 14)
    assert got["generated_tokens"] == want["generated_tokens"]
    assert (got["proposed"], got["accepted"], got["steps"]) == (want["proposed"], want["accepted"], want["steps"])
    assert got["accepted"] >= got["steps"] and got["draft_mode"] == "medusa"
    greedy, _ = lm.generate_tokens(torch.tensor([prompt]), 14)
    assert got["generated_tokens"] == greedy[0].tolist()
    with pytest.raises(ValueError, match="draft model"):
        pipe.generate_batch([prompt], max_tokens=4, do_sample=False)


@pytest.mark.parametrize("k,max_draft", [(4, 2), (4, 4), (1, 2), (8, 8)])
def test_eagle_lite_generate(k, max_draft):
    """draft_mode='eagle' (generate() only, as in the reference): min(k, eagle.max_draft) draft tokens per step from the
    lm_head over hidden states extrapolated on the device (sd_specdec_set_eagle; the reference's _run_eagle_hf,
    pipeline.py:765-889). Tokens, counters and steps equal the oracle's restatement — which reproduces the reference's
    own eagle runs (tests/test_oracle_pipeline.py) — the text equals plain greedy decoding of the target, and a second
    run on the same pipeline starts from a clean extrapolation state."""
    from src.specdec import HipLM, SpeculativePipeline

    drf, tgt = tiny_pair()
    prompt = synthetic_prompts(1, 10, tgt.config.vocab)[0].tolist()
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda")), draft_model="none", draft_mode="eagle", controller="fixed",
                               controller_params={"k": k}, seed=1234)
    pipe.config["eagle"] = {"enabled": True, "alpha": 0.7, "max_draft": max_draft}
    got = pipe.generate(prompt, max_tokens=14, do_sample=False)
    lm = OracleLM(tgt, "bf16")
    ke = min(k, max_draft)
    want = OraclePipeline(lm, None, k=ke, eos_token_id=tgt.config.eos_token_id, draft_mode="eagle", eagle_alpha=0.7).generate(prompt, 14)
    assert got["generated_tokens"] == want["generated_tokens"]
    assert (got["proposed"], got["accepted"], got["steps"]) == (want["proposed"], want["accepted"], want["steps"])
    assert got["proposed"] == ke * got["steps"] and got["draft_mode"] == "eagle"
    greedy, _ = lm.generate_tokens(torch.tensor([prompt]), 14)
    assert got["generated_tokens"] == greedy[0].tolist()
    again = pipe.generate(prompt, max_tokens=14, do_sample=False)
    assert (again["generated_tokens"], again["proposed"], again["accepted"]) == (got["generated_tokens"], got["proposed"], got["accepted"])
    with pytest.raises(ValueError, match="draft model"):
        pipe.generate_batch([prompt], max_tokens=4, do_sample=False)


def test_generate_many_continuous_batching():
    """11 prompts of different lengths and budgets through 3 slots: a finished row's slot is re-used at once.
    Every result equals the prompt's own run (oracle), in prompt order; the run takes fewer device steps than
    the fixed batches of the reference harness would."""
    drf, tgt = tiny_pair(flip_fraction=0.25)
    V = tgt.config.vocab
    g = torch.Generator().manual_seed(11)
    prompts = [torch.randint(4, V, (int(n),), generator=g).tolist() for n in (5, 19, 3, 11, 7, 30, 2, 13, 9, 4, 17)]
    pipe = _pipe(drf, tgt, 4)
    got = pipe.generate_many(prompts, max_tokens=18, batch_size=3, do_sample=False)
    oracle = _oracle_pair(drf, tgt, 4, tgt.config.eos_token_id)
    want = oracle.generate_batch(prompts, 18)
    assert len(got) == len(prompts)
    for i in range(len(prompts)):
        assert got[i]["generated_tokens"] == want[i]["generated_tokens"], i
        assert (got[i]["proposed"], got[i]["accepted"], got[i]["steps"]) == (want[i]["proposed"], want[i]["accepted"], want[i]["steps"])
        assert got[i]["sequence"] == want[i]["sequence"]
    fixed = 0
    for b0 in range(0, len(prompts), 3):
        fixed += max(w["steps"] for w in want[b0:b0 + 3])
    assert got[0]["batch_metrics"]["device_steps"] <= fixed + 4   # (+ the void steps at hand-overs)
    # sampled mode: draw counters restart with every admitted row, streams are the slot indices
    s1 = pipe.generate_many(prompts[:5], max_tokens=10, batch_size=2, do_sample=True, temperature=20.0, top_k=50, top_p=0.95, seed=7)
    s2 = pipe.generate_many(prompts[:5], max_tokens=10, batch_size=2, do_sample=True, temperature=20.0, top_k=50, top_p=0.95, seed=7)
    assert [r["generated_tokens"] for r in s1] == [r["generated_tokens"] for r in s2]


@pytest.mark.parametrize("k,batch,wd,per_head", [(4, 1, "bf16", False), (2, 3, "bf16", False), (4, 2, "fp8", False),
                                                 (4, 2, "bf16", True), (3, 9, "fp8", False), (4, 11, "bf16", False)])
def test_persistent_medusa_heads(k, batch, wd, per_head, monkeypatch):
    """K persistent heads over the target's last hidden state replace the draft forwards (not in the reference,
    SURVEY §8 f4): tokens, proposed/accepted counters and steps equal the oracle restatement; the output is the
    target's greedy continuation; with 20 % wrong head rows the acceptance is high but not total."""
    from oracle import fp8_ref
    from specdec_hip import weights as W
    from src.specdec import HipLM, SpeculativePipeline

    # the K heads are ONE launch (grid row = head) up to 9 rows; per_head forces the one-launch-per-head path, 11 rows take
    # the gathered multi-token path
    if per_head:
        monkeypatch.setenv("SPECDEC_MEDUSA_PER_HEAD", "1")
    drf, tgt = tiny_pair()
    heads = W.synthetic_medusa_heads(tgt, k, flip_fraction=0.2)
    V = tgt.config.vocab
    prompts = synthetic_prompts(batch, 9, V).tolist()
    gh = W.MedusaHeads(heads.weights.cuda())
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda"), weight_dtype=wd), draft_model="none", draft_mode="medusa", medusa_heads=gh,
                               controller="fixed", controller_params={"k": k}, seed=1234)
    got = pipe.generate_batch(prompts, max_tokens=24, do_sample=False)
    if wd == "fp8":
        tq = fp8_ref.dequantized(tgt)
        hq = torch.stack([fp8_ref.quantize_rows(h)[0].float() * fp8_ref.quantize_rows(h)[1][:, None] for h in heads.weights])
        lm, oh = OracleLM(tq, "bf16"), hq
    else:
        lm, oh = OracleLM(tgt, "bf16"), heads.weights
    oracle = OraclePipeline(lm, None, k=k, eos_token_id=tgt.config.eos_token_id, draft_mode="medusa_heads", medusa_heads=oh)
    want = oracle.generate_batch(prompts, 24)
    for b in range(batch):
        assert got[b]["generated_tokens"] == want[b]["generated_tokens"], (k, b)
        assert (got[b]["proposed"], got[b]["accepted"]) == (want[b]["proposed"], want[b]["accepted"])
        greedy, _ = lm.generate_tokens(torch.tensor([prompts[b]]), len(got[b]["generated_tokens"]))
        assert got[b]["generated_tokens"] == greedy[0].tolist()
    acc = sum(r["accepted"] for r in got) / sum(r["proposed"] for r in got)
    assert 0.5 < acc <= (k + 1) / k
    single = pipe.generate(prompts[0], max_tokens=12, do_sample=False)
    ws = oracle.generate(prompts[0], 12)
    assert single["generated_tokens"] == ws["generated_tokens"]
    assert (single["proposed"], single["accepted"], single["steps"]) == (ws["proposed"], ws["accepted"], ws["steps"])


def test_logit_threshold_policies_match_oracle_and_reference():
    """policy = typical / topk_agree / conf_threshold (the reference accepts all four, policies.py:399-425, used at
    pipeline.py:1092 and :3018): the pipeline takes the reference's own verification for them — the base model's K greedy
    tokens and logits from the same prefix — with every forward on the HIP engine. Tokens, counters and steps equal the
    oracle loop on the same bf16 weights, and the reference's own fp32 traces (tests/golden/pipeline_policies_golden.json)."""
    from src.specdec import HipLM, SpeculativePipeline

    with open(os.path.join(GOLD, "pipeline_policies_golden.json")) as f:
        g = json.load(f)
    drf, tgt = cases.policy_pair(torch.bfloat16)
    base, draft = OracleLM(tgt, "bf16"), OracleLM(drf, "bf16")
    blm, dlm = HipLM(tgt.to("cuda")), HipLM(drf.to("cuda"))
    same_as_reference = 0
    for run in g["runs"]:
        k, mt, prompt = run["k"], run["max_tokens"], run["prompt_ids"]
        pipe = SpeculativePipeline(base_lm=blm, draft_lm=dlm, policy=run["policy"], policy_params=run["params"],
                                   controller="fixed", controller_params={"k": k}, seed=1234)
        oracle = OraclePipeline(base, draft, k=k, eos_token_id=2, policy=run["policy"], policy_params=run["params"])
        got = pipe.generate_batch([prompt], max_tokens=mt, do_sample=False)[0]
        want = oracle.generate_batch([prompt], mt)[0]
        assert got["generated_tokens"] == want["generated_tokens"], (run["policy"], run["params"], k)
        assert (got["proposed"], got["accepted"], got["batch_metrics"]["total_steps"]) == (want["proposed"], want["accepted"], want["steps"])
        gs = pipe.generate(prompt, max_tokens=mt, do_sample=False)
        ws = oracle.generate(prompt, mt)
        assert gs["generated_tokens"] == ws["generated_tokens"], (run["policy"], run["params"], k)
        assert (gs["proposed"], gs["accepted"], gs["steps"]) == (ws["proposed"], ws["accepted"], ws["steps"])
        assert gs["policy"]["policy"] == run["policy"]
        ref_b, ref_s = run["batch"], run["single"]
        same_as_reference += (got["generated_tokens"] == ref_b["generated_tokens"] and got["accepted"] == ref_b["accepted"]
                              and gs["generated_tokens"] == ref_s["generated_tokens"] and gs["accepted"] == ref_s["accepted"])
    # a probability threshold can fall between the bf16 and the fp32 value of a borderline token; everything else is the
    # reference's trace exactly
    assert same_as_reference >= len(g["runs"]) - 2, same_as_reference
    # two rows of different lengths keep their own cached prefixes
    pipe = SpeculativePipeline(base_lm=blm, draft_lm=dlm, policy="topk_agree", policy_params={"k": 20}, controller="fixed",
                               controller_params={"k": 4}, seed=1234)
    oracle = OraclePipeline(base, draft, k=4, eos_token_id=2, policy="topk_agree", policy_params={"k": 20})
    prompts = [g["runs"][0]["prompt_ids"], g["runs"][1]["prompt_ids"][:4]]
    got, want = pipe.generate_batch(prompts, max_tokens=10, do_sample=False), oracle.generate_batch(prompts, 10)
    for a, b in zip(got, want):
        assert a["generated_tokens"] == b["generated_tokens"] and (a["proposed"], a["accepted"]) == (b["proposed"], b["accepted"])


@pytest.mark.parametrize("k", [1, 2, 4])
def test_medusa_random_heads_as_the_reference_pipeline_runs_them(k):
    """draft_mode='medusa' with head_init 'random' = the reference PIPELINE's Medusa mode (_run_medusa_hf, pipeline.py:655-763):
    fresh random heads and multinomial draws from the global torch generator every step, over the target's last hidden
    state from the HIP engine. Under torch.manual_seed the run equals the oracle's replay of the generator (which is pinned
    to the reference's own runs, tests/test_oracle_pipeline.py) token for token."""
    from src.specdec import HipLM, SpeculativePipeline

    drf, tgt = tiny_pair()
    prompt = synthetic_prompts(1, 9, tgt.config.vocab)[0].tolist()
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda")), draft_model="none", draft_mode="medusa", controller="fixed",
                               controller_params={"k": k}, seed=1234)
    pipe.config["medusa"] = {"enabled": True, "num_heads": 2, "head_init": "random", "temperature": 0.7, "top_p": 1.0}
    torch.manual_seed(777 + k)
    got = pipe.generate(prompt, max_tokens=10, temperature=0.7, do_sample=False)
    oracle = OraclePipeline(OracleLM(tgt, "bf16"), None, k=k, eos_token_id=tgt.config.eos_token_id, draft_mode="medusa_random",
                            medusa_num_heads=2, medusa_temperature=0.7)
    torch.manual_seed(777 + k)
    want = oracle.generate(prompt, 10)
    assert got["generated_tokens"] == want["generated_tokens"]
    assert (got["proposed"], got["accepted"], got["steps"]) == (want["proposed"], want["accepted"], want["steps"])
    assert got["proposed"] == k * got["steps"]


def test_rejection_sampling_policy_in_the_pipeline():
    """policy='rejection' (opt-in speculative sampling, not one of the reference's policies): the run equals a CPU replay —
    OracleLM logits, oracle/hostlogic_ref.rejection_accept, the same uniform stream — and with draft == target every
    draft token is accepted (p / q = 1)."""
    from oracle.hostlogic_ref import rejection_accept
    from src.specdec import HipLM, SpeculativePipeline
    from src.specdec.policies.policies import RejectionSamplingPolicy

    drf, tgt = cases.policy_pair(torch.bfloat16)
    blm, dlm = HipLM(tgt.to("cuda")), HipLM(drf.to("cuda"))
    k, mt, seed, temp = 3, 12, 21, 0.8
    prompt = [9, 40, 77, 101, 5, 66]
    pipe = SpeculativePipeline(base_lm=blm, draft_lm=dlm, policy="rejection", policy_params={"temperature": temp, "seed": seed},
                               controller="fixed", controller_params={"k": k}, seed=1234)
    got = pipe.generate_batch([prompt], max_tokens=mt, do_sample=True)[0]
    # CPU replay
    base, draft = OracleLM(tgt, "bf16"), OracleLM(drf, "bf16")
    ref = RejectionSamplingPolicy(temp, seed)
    seq, gen, proposed, accepted = list(prompt), [], 0, 0
    while len(gen) < mt:
        drafted, dl = [], []
        for _ in range(k):
            lg, _ = draft.forward(torch.tensor([seq + drafted]))
            dl.append(lg[0, -1])
            drafted.append(ref.draw(ref.distributions(lg[0, -1]), float(ref.uniforms(1)[0])))
        lg, _ = base.forward(torch.tensor([seq + drafted]))
        bl = lg[0, len(seq) - 1:]
        a, nxt = rejection_accept(drafted, torch.stack(dl).numpy(), bl.numpy(), ref.uniforms(k).numpy(), temp)
        tok = ref.draw(torch.from_numpy(nxt), float(ref.uniforms(1)[0]))
        emitted = drafted[:a] + [tok]
        if 2 in emitted:
            emitted = emitted[: emitted.index(2) + 1][: mt - len(gen)]
            seq, gen = seq + emitted, gen + emitted
            proposed, accepted = proposed + k, accepted + a + 1
            break
        emitted = emitted[: mt - len(gen)]          # trimmed to the budget
        seq, gen, proposed, accepted = seq + emitted, gen + emitted, proposed + k, accepted + a + 1
    assert got["generated_tokens"] == gen
    assert (got["proposed"], got["accepted"]) == (proposed, accepted)
    same = SpeculativePipeline(base_lm=blm, draft_lm=blm, policy="rejection", policy_params={"temperature": temp, "seed": 3},
                               controller="fixed", controller_params={"k": k}, seed=1234).generate_batch([prompt], max_tokens=mt, do_sample=True)[0]
    steps = same["batch_metrics"]["total_steps"]
    assert same["accepted"] == (k + 1) * steps or 2 in same["generated_tokens"]
