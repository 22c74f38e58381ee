"""HIP registry ops vs the oracle and the reference goldens (bit-exact), on the GPU."""

import json
import os

import numpy as np
import pytest
import torch

import cases
from oracle.kernels_ref import (
    argmax_oracle,
    kv_append_oracle,
    kv_append_with_mask_oracle,
    verify_prefix_oracle,
)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "kernels_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def K():
    import src.kernels as k

    assert k.get_kernel_info()["library_loaded"], "HIP library not loaded"
    return k


@pytest.mark.parametrize("case", cases.VERIFY_CASES, ids=[c[0] for c in cases.VERIFY_CASES])
def test_verify_prefix_matches_golden_and_oracle(case, gold, K):
    from specdec_hip.ops import verify_prefix_hip

    g = gold["verify"][case[0]]
    logits, ids = cases.build_verify_case(*case, seed=g["seed"])
    alen, mask, pred = verify_prefix_hip(logits.cuda(), ids.cuda(), return_pred=True)
    assert alen.device.type == "cuda" and alen.dtype == torch.int32 and mask.dtype == torch.uint8
    assert alen.cpu().tolist() == g["accept_len"]
    assert mask.cpu().tolist() == g["mask"]
    assert pred.cpu().tolist() == g["argmax"]
    o_alen, o_mask = verify_prefix_oracle(logits, ids)
    assert torch.equal(alen.cpu(), o_alen) and torch.equal(mask.cpu(), o_mask)
    # registry spelling gives the same
    r_alen, r_mask = K.verify_prefix(logits.cuda(), ids.cuda())
    assert torch.equal(r_alen, alen) and torch.equal(r_mask, mask)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1, 1, 100), (1, 2, 1000), (2, 3, 5000), (4, 4, 10000), (3, 2, 50257), (2, 5, 128256)])
def test_verify_prefix_random_vs_oracle(dtype, shape, K):
    """reference tests/test_kernels_verify.py:67-94 (kernel == reference on random data),
    extended over dtypes and the Llama vocabulary; low-precision dtypes make ties common."""
    B, Kk, V = shape
    g = torch.Generator().manual_seed(B * 1000 + Kk * 10 + V)
    logits = torch.randn(B, Kk, V, generator=g).to(dtype)
    pred = argmax_oracle(logits)
    ids = pred.clone()
    flip = torch.rand(B, Kk, generator=g) < 0.4
    ids[flip] = (ids[flip] + 1) % V
    alen, mask = K.verify_prefix(logits.cuda(), ids.cuda())
    o_alen, o_mask = verify_prefix_oracle(logits, ids)
    assert torch.equal(alen.cpu(), o_alen)
    assert torch.equal(mask.cpu(), o_mask)


def test_verify_prefix_nan_and_strided(K):
    from specdec_hip.ops import verify_prefix_hip

    g = torch.Generator().manual_seed(5)
    full = torch.randn(3, 6, 2000, generator=g)
    full[1, 2, 77] = float("nan")
    full[1, 2, 1500] = float("nan")  # first NaN wins
    view = full[:, ::2, :]  # strided K, contiguous V
    ids = argmax_oracle(view)
    alen, mask, pred = verify_prefix_hip(view.cuda(), ids.cuda(), return_pred=True)
    assert pred.cpu().tolist() == ids.tolist()
    assert int(pred[1, 1]) == 77
    assert alen.cpu().tolist() == [3, 3, 3]
    # non-contiguous vocabulary axis is made contiguous by the wrapper
    tr = torch.randn(2, 300, 3, generator=g).transpose(1, 2)
    ids = argmax_oracle(tr)
    alen, _ = K.verify_prefix(tr.cuda(), ids.cuda())
    assert alen.cpu().tolist() == [3, 3]


def test_verify_prefix_empty_and_errors(K):
    alen, mask = K.verify_prefix(torch.zeros(0, 3, 10).cuda(), torch.zeros(0, 3, dtype=torch.long).cuda())
    assert alen.shape == (0,) and mask.shape == (0, 3)
    alen, mask = K.verify_prefix(torch.zeros(2, 0, 10).cuda(), torch.zeros(2, 0, dtype=torch.long).cuda())
    assert alen.cpu().tolist() == [0, 0] and mask.shape == (2, 0)
    with pytest.raises(AssertionError):
        K.verify_prefix(torch.zeros(2, 3, 10).cuda(), torch.zeros(2, 4, dtype=torch.long).cuda())
    with pytest.raises(TypeError):
        K.verify_prefix(torch.zeros(2, 3, 10, dtype=torch.float64).cuda(), torch.zeros(2, 3, dtype=torch.long).cuda())


def test_verify_prefix_reference_planted_cases(K):
    """reference tests/test_kernels_verify.py:16-41, 96-129 on the device."""
    torch.manual_seed(0)
    B, Kk, V = 2, 3, 1000
    logits = torch.randn(B, Kk, V, device="cuda")
    ids = torch.randint(0, V - 1, (B, Kk), device="cuda")
    logits[0, 0, ids[0, 0]] = 10.0
    logits[0, 1, ids[0, 1]] = 10.0
    logits[0, 2, ids[0, 2] + 1] = 10.0
    logits[1, 0, ids[1, 0] + 1] = 10.0
    alen, mask = K.verify_prefix(logits, ids)
    assert alen.device == logits.device and mask.device == logits.device
    assert alen.tolist() == [2, 0]
    assert mask.tolist() == [[1, 1, 0], [0, 0, 0]]
    logits = torch.randn(1, 2, 50257, device="cuda")
    ids = torch.randint(0, 50257, (1, 2), device="cuda")
    logits[0, 0, ids[0, 0]] = 10.0
    logits[0, 1, ids[0, 1]] = 10.0
    alen, mask = K.verify_prefix(logits, ids)
    assert alen.tolist() == [2] and mask.tolist() == [[1, 1]]


@pytest.mark.parametrize("case", cases.KV_CASES, ids=[c[0] for c in cases.KV_CASES])
def test_kv_ops_match_golden_and_oracle(case, gold, K):
    from specdec_hip.ops import kv_append_with_mask_hip, kv_concat_hip

    name = case[0]
    g = gold["kv"][name]
    bk, bv, nk, nv, mask, alen = cases.build_kv_case(*case, seed=g["seed"])
    dev = [t.cuda() for t in (bk, bv, nk, nv)]
    ok, ov = K.kv_append(*dev)
    ek, ev = kv_append_oracle(bk, bv, nk, nv)
    assert ok.shape == ek.shape and ok.dtype == ek.dtype
    assert torch.equal(ok.cpu(), ek) and torch.equal(ov.cpu(), ev)
    assert [cases.checksum(ok), cases.checksum(ov)] == g["concat_checksum"]
    ck, cv = kv_concat_hip(*dev)
    assert ck.is_contiguous() and torch.equal(ck.cpu(), ek) and torch.equal(cv.cpu(), ev)
    mk, mv = kv_append_with_mask_hip(*dev, mask.cuda(), alen.cuda())
    emk, emv = kv_append_with_mask_oracle(bk, bv, nk, nv, mask, alen)
    assert torch.equal(mk.cpu(), emk) and torch.equal(mv.cpu(), emv)
    assert [cases.checksum(mk), cases.checksum(mv)] == g["masked_checksum"]
    # inputs are borrowed, never mutated
    assert torch.equal(dev[0].cpu(), bk) and torch.equal(dev[2].cpu(), nk)


def test_kv_append_grows_in_place_and_never_touches_old_rows(K):
    """Chain of appends: the second append onto the previous result is in place
    (same storage), every intermediate view keeps its contents, and a fork from an
    old view copies instead of clobbering."""
    g = torch.Generator().manual_seed(3)
    B, H, D = 2, 8, 64
    base_k = torch.randn(B, H, 5, D, generator=g).bfloat16()
    base_v = torch.randn(B, H, 5, D, generator=g).bfloat16()
    cur_k, cur_v = base_k.cuda(), base_v.cuda()
    ref_k, ref_v = base_k, base_v
    views = []
    for step in range(40):
        kk = 1 + step % 5
        nk = torch.randn(B, H, kk, D, generator=g).bfloat16()
        nv = torch.randn(B, H, kk, D, generator=g).bfloat16()
        new_k, new_v = K.kv_append(cur_k, cur_v, nk.cuda(), nv.cuda())
        ref_k, ref_v = kv_append_oracle(ref_k, ref_v, nk, nv)
        views.append((cur_k, ref_k[:, :, : cur_k.shape[2]].clone()))
        cur_k, cur_v = new_k, new_v
        assert torch.equal(cur_k.cpu(), ref_k) and torch.equal(cur_v.cpu(), ref_v)
    for v, want in views:
        assert torch.equal(v.cpu(), want)
    in_place = sum(1 for (a, _), (b, _) in zip(views[1:], views[2:]) if a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr())
    assert in_place >= 30, "appends onto the previous result must not re-copy the cache"
    # fork from an old view: must not overwrite rows the newest view already owns
    old_k, old_v = views[10][0], cur_v[:, :, : views[10][0].shape[2]]
    fk = torch.randn(B, H, 2, D, generator=g).bfloat16()
    before = cur_k.cpu().clone()
    f_k, _ = K.kv_append(old_k, old_v.contiguous(), fk.cuda(), fk.cuda())
    assert torch.equal(cur_k.cpu(), before)
    assert torch.equal(f_k[:, :, -2:].cpu(), fk)


def test_kv_append_inplace_per_row_offsets(K):
    from specdec_hip.ops import kv_append_inplace_hip

    g = torch.Generator().manual_seed(11)
    B, H, Lmax, D, Kk = 3, 8, 64, 128, 5
    cache_k = torch.randn(B, H, Lmax, D, generator=g).bfloat16()
    cache_v = torch.randn(B, H, Lmax, D, generator=g).bfloat16()
    nk = torch.randn(B, H, Kk, D, generator=g).bfloat16()
    nv = torch.randn(B, H, Kk, D, generator=g).bfloat16()
    row_len = torch.tensor([0, 17, Lmax - Kk], dtype=torch.int32)
    ck, cv = cache_k.cuda(), cache_v.cuda()
    kv_append_inplace_hip(ck, cv, nk.cuda(), nv.cuda(), row_len=row_len.cuda())
    ek, ev = cache_k.clone(), cache_v.clone()
    for b in range(B):
        o = int(row_len[b])
        ek[b, :, o : o + Kk] = nk[b]
        ev[b, :, o : o + Kk] = nv[b]
    assert torch.equal(ck.cpu(), ek) and torch.equal(cv.cpu(), ev)
    # rows that would run past Lmax are dropped, never written out of bounds
    row_len2 = torch.tensor([Lmax - 2, 0, 0], dtype=torch.int32)
    ck2 = cache_k.cuda()
    kv_append_inplace_hip(ck2, cache_v.cuda(), nk.cuda(), nv.cuda(), row_len=row_len2.cuda())
    assert torch.equal(ck2[0, :, Lmax - 2 :].cpu(), nk[0, :, :2])


def test_kv_assertions(K):
    with pytest.raises(AssertionError):
        K.kv_append(torch.zeros(2, 2, 3, 4).cuda(), torch.zeros(2, 2, 3, 4).cuda(), torch.zeros(1, 2, 2, 4).cuda(), torch.zeros(1, 2, 2, 4).cuda())
    with pytest.raises(AssertionError):
        K.kv_append(torch.zeros(1, 4, 3, 4).cuda(), torch.zeros(1, 4, 3, 4).cuda(), torch.zeros(1, 2, 2, 4).cuda(), torch.zeros(1, 2, 2, 4).cuda())
    with pytest.raises(AssertionError):
        K.kv_append(torch.zeros(1, 2, 3, 8).cuda(), torch.zeros(1, 2, 3, 8).cuda(), torch.zeros(1, 2, 2, 4).cuda(), torch.zeros(1, 2, 2, 4).cuda())
