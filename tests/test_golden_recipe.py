"""The fixtures under tests/golden/ are what tests/golden/make_golden.py produces from the REFERENCE.

Runs in the build container only (the reference does not exist on the GPU box): the generator is executed from /tmp
with the fixture directory redirected, and every file it writes must equal the committed one byte for byte. It also
guards the import order of the generator — the product ships packages named `specdec` / `kernels` too, and a fixture
generated from those would certify the product against itself (make_golden._reference_pipeline_class asserts the
origin of what it imported)."""

import filecmp
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF = os.environ.get("SPECDEC_REFERENCE", "/root/reference")
SETS = ["kernels", "hf", "pipeline", "pipeline_eagle", "hostlogic", "medusa", "policies", "harness", "fake", "pipeline_sampled"]
FILES = ["kernels_golden.json", "kernels_golden.npz", "hf_llama_tiny.json", "hf_llama_tiny.npz", "hf_gpt2_tiny.json",
         "hf_gpt2_tiny.npz", "pipeline_golden.json", "pipeline_eagle_golden.json", "hostlogic_golden.json",
         "medusa_golden.json", "pipeline_policies_golden.json", "harness_schema_golden.json", "fake_pipeline_golden.json",
         "pipeline_sampled_golden.json"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "specdec")), reason="needs the reference checkout (build container only)")
def test_generator_reproduces_every_committed_fixture(tmp_path):
    env = dict(os.environ, SPECDEC_GOLDEN_OUT=str(tmp_path), PYTHONHASHSEED="0")
    env.pop("PYTHONPATH", None)
    res = subprocess.run([sys.executable, os.path.join(GOLD, "make_golden.py"), *SETS], cwd="/tmp", env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    for name in FILES:
        assert os.path.exists(tmp_path / name), f"{name} was not generated"
        assert filecmp.cmp(tmp_path / name, os.path.join(GOLD, name), shallow=False), f"{name} differs from the committed fixture"
