"""The oracle's registry-op restatements against goldens captured from the reference."""

import json
import os

import numpy as np
import pytest
import torch

import cases
from oracle.kernels_ref import kv_append_oracle, kv_append_with_mask_oracle, verify_prefix_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "kernels_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def gold_arrays():
    return np.load(os.path.join(GOLD, "kernels_golden.npz"))


@pytest.mark.parametrize("case", cases.VERIFY_CASES, ids=[c[0] for c in cases.VERIFY_CASES])
def test_verify_prefix_oracle_matches_reference(case, gold):
    g = gold["verify"][case[0]]
    logits, ids = cases.build_verify_case(*case, seed=g["seed"])
    assert cases.checksum(logits) == pytest.approx(g["logits_checksum"], rel=0, abs=0)
    assert cases.checksum(ids) == g["ids_checksum"]
    alen, mask = verify_prefix_oracle(logits, ids)
    assert alen.dtype == torch.int32 and mask.dtype == torch.uint8
    assert alen.tolist() == g["accept_len"]
    assert mask.tolist() == g["mask"]


@pytest.mark.parametrize("case", cases.KV_CASES, ids=[c[0] for c in cases.KV_CASES])
def test_kv_oracles_match_reference(case, gold, gold_arrays):
    name = case[0]
    g = gold["kv"][name]
    bk, bv, nk, nv, mask, alen = cases.build_kv_case(*case, seed=g["seed"])
    assert [cases.checksum(t) for t in (bk, bv, nk, nv, mask, alen)] == g["inputs_checksum"]
    ok, ov = kv_append_oracle(bk, bv, nk, nv)
    mk, mv = kv_append_with_mask_oracle(bk, bv, nk, nv, mask, alen)
    assert [cases.checksum(ok), cases.checksum(ov)] == g["concat_checksum"]
    assert [cases.checksum(mk), cases.checksum(mv)] == g["masked_checksum"]
    if f"{name}/concat_k" in gold_arrays:
        assert np.array_equal(ok.float().numpy(), gold_arrays[f"{name}/concat_k"])
        assert np.array_equal(ov.float().numpy(), gold_arrays[f"{name}/concat_v"])
        assert np.array_equal(mk.float().numpy(), gold_arrays[f"{name}/masked_k"])
        assert np.array_equal(mv.float().numpy(), gold_arrays[f"{name}/masked_v"])


def test_reference_known_answers():
    """The planted-match known answers of reference tests/test_kernels_verify.py:16-65."""
    torch.manual_seed(0)
    B, K, V = 2, 3, 1000
    logits = torch.randn(B, K, V)
    ids = torch.randint(0, V - 1, (B, K))
    logits[0, 0, ids[0, 0]] = 10.0
    logits[0, 1, ids[0, 1]] = 10.0
    logits[0, 2, ids[0, 2] + 1] = 10.0
    logits[1, 0, ids[1, 0] + 1] = 10.0
    alen, mask = verify_prefix_oracle(logits, ids)
    assert alen.tolist() == [2, 0]
    assert mask.tolist() == [[1, 1, 0], [0, 0, 0]]


def test_masked_zero_accept_keeps_base():
    """reference tests/test_kv_cache.py:164-186."""
    B, H, L, D, K = 1, 2, 3, 4, 2
    bk, bv = torch.randn(B, H, L, D), torch.randn(B, H, L, D)
    dk, dv = torch.randn(B, H, K, D), torch.randn(B, H, K, D)
    mk, _ = kv_append_with_mask_oracle(bk, bv, dk, dv, torch.zeros(B, K, dtype=torch.uint8), torch.tensor([0], dtype=torch.int32))
    assert mk.shape == (B, H, L + K, D)
    assert torch.equal(mk[:, :, :L], bk) and torch.equal(mk[:, :, L:], torch.zeros(B, H, K, D))
