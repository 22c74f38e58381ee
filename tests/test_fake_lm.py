"""`implementation: fake` — the reference's weight-less test double (src/specdec/models/fake_lm.py:19-146) and its own
configs/specdec.yaml, pinned by tests/golden/fake_pipeline_golden.json (the reference pipeline run from that YAML: SURVEY
section 8c, G7 — generate() of "Hello world" gives [189, 862, 115, 24, 416, 757, 752, 286], proposed 32, accepted 0).

CPU: the double's token function, tokenizer info, decode. GPU: the pipeline from the reference's YAML values (the noise logits live
on the device and the exact-match policy runs the registry's HIP verify_prefix on them)."""

import json
import os

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# the reference's configs/specdec.yaml:5-23, value for value (the file itself does not travel to the GPU box)
REFERENCE_SPECDEC_YAML = """\
base_model: gpt2
draft_model: distilgpt2
max_draft: 4
implementation: fake  # fake for testing, hf for real models
temperature: 0.7
do_sample: true
max_new_tokens: 64
top_p: 0.9
top_k: 50
repetition_penalty: 1.0
device: auto  # prioritize MPS > CPU
max_memory_mb: 500  # Memory limit for HF models
seed: 1234
log_level: INFO
verbose: false
"""


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "fake_pipeline_golden.json")) as f:
        return json.load(f)


def test_reference_yaml_values_are_the_references():
    """The YAML text above is what the reference ships (checked where the reference exists: the build container)."""
    import yaml

    ref = os.path.join(os.environ.get("SPECDEC_REFERENCE", "/root/reference"), "configs", "specdec.yaml")
    if not os.path.exists(ref):
        pytest.skip("needs the reference checkout (build container only)")
    with open(ref) as f:
        assert yaml.safe_load(f) == yaml.safe_load(REFERENCE_SPECDEC_YAML)


def test_token_function_and_tokenizer_info_match_the_reference_double(golden):
    from src.specdec.models.fake_lm import FakeLM

    lm = FakeLM(model_name="fake-base-gpt2", vocab_size=1000, device="cpu", seed=1234)
    assert lm.get_tokenizer_info() == golden["tokenizer_info"]
    for case in golden["token_function"]:
        toks, logits = lm.generate_tokens(torch.tensor([case["input_ids"]]), case["k"])
        assert toks[0].tolist() == case["tokens"], case["input_ids"]
        assert list(logits.shape) == case["logits_shape"] and toks.dtype == torch.long
        assert all(t not in (0, 1, 2, 3) for t in case["tokens"])     # special ids are stepped over
    assert lm.decode(torch.tensor([[189, 862, 115, 24]])) == "fake_text_189_862_115"
    assert lm.decode(torch.empty(1, 0, dtype=torch.long)) == ""
    assert lm.encode("Hello world").shape == (1, 5) and lm.encode("a").shape == (1, 3)


@pytest.mark.gpu
def test_pipeline_runs_the_references_yaml_unchanged(golden, tmp_path):
    from src.specdec import SpeculativePipeline

    cfg = tmp_path / "specdec.yaml"
    cfg.write_text(REFERENCE_SPECDEC_YAML)
    for run in golden["runs"]:
        pipe = SpeculativePipeline(config_path=str(cfg))
        assert pipe.config["implementation"] == "fake" and pipe.base_lm.model_name == "fake-base-gpt2" and pipe.draft_lm.model_name == "fake-draft-distilgpt2"
        # (FakeLM.encode hashes the TEXT — process-dependent — so the reference's encoded ids are fed; do_sample stays the file's
        #  `true`: the double ignores it, as in the reference)
        rs = pipe.generate(run["prompt_ids"], max_tokens=run["max_tokens"], do_sample=False)
        want = run["single"]
        assert rs["generated_tokens"] == want["generated_tokens"], run["prompt"]
        assert (rs["proposed"], rs["accepted"], rs["steps"]) == (want["proposed"], want["accepted"], want["steps"])
        rb = SpeculativePipeline(config_path=str(cfg)).generate_batch([run["prompt_ids"]], max_tokens=run["max_tokens"])[0]
        wb = run["batch"]
        assert rb["generated_tokens"] == wb["generated_tokens"] and (rb["proposed"], rb["accepted"], rb["steps"]) == (wb["proposed"], wb["accepted"], wb["steps"])
        assert rs["text"].startswith("fake_text_")


@pytest.mark.gpu
def test_hub_model_names_of_the_reference_configs_resolve_to_synthetic_presets(caplog):
    """`base_model: gpt2`, `draft_model: distilgpt2` with `implementation: hf`: nothing is downloaded by name; without
    $SPECDEC_MODEL_DIR/<name> the architecture's synthetic preset is used, built as a PAIR, with a logged notice."""
    import logging

    from src.specdec import SpeculativePipeline

    with caplog.at_level(logging.WARNING):
        pipe = SpeculativePipeline(base_model="gpt2", draft_model="distilgpt2", implementation="hf", max_draft=2,
                                   controller="fixed", controller_params={"k": 2}, seed=1234)
    assert any("nothing is downloaded by name" in r.message for r in caplog.records)
    assert pipe.base_lm.config.n_layers == 12 and pipe.draft_lm.config.n_layers == 6 and pipe.base_lm.vocab_size == 50257
    out = pipe.generate_batch([[11, 22, 33, 44, 55, 66]], max_tokens=12, do_sample=False)[0]
    assert len(out["generated_tokens"]) >= 12 and out["accepted"] > out["proposed"] // 2 // 2   # the pair shares its successor structure
