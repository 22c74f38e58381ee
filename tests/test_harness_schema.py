"""Result-file schema of the K-sweep harness and the `specdec` CLI surface against the reference's
(tests/golden/harness_schema_golden.json: keys and value types captured by RUNNING scripts/comprehensive_k_sweep.py of the
reference on local tiny models, make_golden.py harness; CLI options from src/specdec_cli/main.py:79-102)."""

import csv
import importlib.util
import json
import os
import sys
from types import SimpleNamespace

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "harness_schema_golden.json")


def _harness():
    spec = importlib.util.spec_from_file_location("k_sweep", os.path.join(ROOT, "llm-inference-lab_amd", "scripts", "k_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_prompt_suite_is_the_reference_suite():
    with open(GOLD) as f:
        g = json.load(f)
    assert _harness().PROMPT_SUITE == g["prompt_suite"]


def test_cli_surface():
    from src.specdec_cli.main import build_parser

    p = build_parser()
    a = p.parse_args(["run", "--k", "3", "--max-tokens", "5", "1 2 3"])
    assert (a.cmd, a.k, a.max_tokens, a.prompt, a.do_sample, a.temperature) == ("run", 3, 5, "1 2 3", False, 0.7)
    b = p.parse_args(["bench", "--max-tokens", "8", "--iterations", "2", "--deterministic", "--output-dir", "/tmp/x"])
    assert (b.cmd, b.max_tokens, b.iterations, b.deterministic, str(b.output_dir)) == ("bench", 8, 2, True, "/tmp/x")
    with pytest.raises(SystemExit):
        p.parse_args([])
    import specdec_cli
    import src.specdec_cli

    assert specdec_cli is src.specdec_cli


@pytest.mark.gpu
def test_k_sweep_files_have_the_reference_schema(tmp_path, capsys):
    """Runs the harness (tiny pair, K = 1..2, 6 tokens, batch 2) and the CLI's `run`; the CSV header equals the reference's,
    summary / detailed rows carry every reference key with the same kind of value, the JSON has the same top level."""
    from helpers import tiny_pair
    from src.specdec import HipLM

    with open(GOLD) as f:
        g = json.load(f)
    H = _harness()
    drf, tgt = tiny_pair()
    args = SimpleNamespace(base_model="tiny", draft_model="tiny", share_draft_embeddings=False, flip=0.2, max_k=2, max_tokens=6,
                           iterations=1, batch_size=2, continuous=False, do_sample=False)
    results, detailed = H.run(args, base=HipLM(tgt.to("cuda")), draft=HipLM(drf.to("cuda")))
    csv_file, json_file = H.save(results, detailed, tmp_path, 2)
    assert os.path.basename(csv_file).startswith("specdec_cuda_") and str(csv_file).endswith(".csv")
    with open(csv_file) as f:
        assert next(csv.reader(f)) == g["csv_header"]
    with open(json_file) as f:
        written = json.load(f)
    assert sorted(written.keys()) == g["json_top_level"]
    assert set(g["system_info_keys"]) <= set(written["system_info"].keys())
    assert len(written["summary_results"]) == 2 and len(written["detailed_results"]) == 2 * len(g["prompt_suite"])

    def kind(v):
        return {"float64": "float", "int64": "int"}.get(type(v).__name__, type(v).__name__)

    for ours, ref in ((results[0], g["summary_row"]), (detailed[0], g["detailed_row"])):
        assert set(ref) <= set(ours), set(ref) - set(ours)
        for k, t in ref.items():
            assert kind(ours[k]) == {"float64": "float"}.get(t, t), (k, type(ours[k]).__name__, t)
    assert list(results[0].keys()) == list(g["summary_row"].keys())          # same column order
    assert written["system_info"]["kernel_backends"]["verify_backend"] == "hip"


def test_manifest_has_the_reference_export_keys(tmp_path):
    """MANIFEST.json of a results directory: the keys of the reference's exported runs (fixture data below is the key set and
    nesting of docs/results/2025-10-30-T4-Phase3D-Run1-32tok-100iter-fp16/MANIFEST.json)."""
    h = _harness()
    args = SimpleNamespace(max_tokens=32, iterations=3, base_model="synthetic:llama-3.2-3b", draft_model="synthetic:llama-3.2-1b")
    (tmp_path / "a.csv").write_text("x\n")
    (tmp_path / "a.json").write_text("{}")
    path = h.write_manifest(tmp_path, tmp_path / "a.csv", tmp_path / "a.json", args)
    with open(path) as f:
        man = json.load(f)
    assert set(man) == {"export_created_at", "device", "dtype", "max_tokens", "iterations_per_k", "models", "artifacts", "source_dir"}
    assert set(man["models"]) == {"base", "draft"} and set(man["artifacts"]) == {"summary_json", "summary_csv"}
    assert man["artifacts"] == {"summary_json": "a.json", "summary_csv": "a.csv"} and man["iterations_per_k"] == 3
