"""HIP decoder forward (C-ABI sd_model_forward) vs the CPU oracle, on the GPU."""

import pytest
import torch

from helpers import load_hf_golden, synthetic_prompts, tiny_pair
from oracle.model_ref import OracleLM

pytestmark = pytest.mark.gpu


def _hip_model(mw, batch, l_max):
    from specdec_hip.engine import HipModel

    return HipModel(mw.to("cuda"), batch=batch, l_max=l_max)


def _rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)


@pytest.mark.parametrize("name", ["llama", "gpt2"])
def test_forward_logits_match_oracle_on_hf_weights(name):
    """Random HF-initialised weights (small argmax margins): compare logits numerically.
    Tolerance: bf16 activations between ops, fp32 accumulation in both; the two differ by
    accumulation order only, which moves a bf16-rounded logit by at most a few ulps."""
    mw, toks, _ = load_hf_golden(name, dtype=torch.bfloat16)
    B, L = toks.shape
    lm = OracleLM(mw, precision="bf16")
    want, _ = lm.forward(toks)
    hm = _hip_model(mw, batch=B, l_max=128)
    pos0 = torch.zeros(B, dtype=torch.int32, device="cuda")
    ids, logits = hm.forward(toks.to(torch.int32).cuda(), pos0, 0, want_logits=True)
    got = logits.float().cpu()
    assert _rel_err(got, want) < 0.03, _rel_err(got, want)
    # the fused argmax equals argmax of the logits the kernel itself stored
    assert torch.equal(ids.cpu().long(), got.argmax(-1))
    agree = (ids.cpu().long() == want.argmax(-1)).float().mean().item()
    assert agree > 0.9, agree


def test_incremental_decode_matches_oracle_tokens():
    """Prefill (tiled over >9 tokens, head skipped) then M=1 decode steps and an M=5
    verify-shaped forward, rows at different lengths: token ids identical to the oracle."""
    drf, tgt = tiny_pair()
    lm = OracleLM(tgt, precision="bf16")
    B, P, V = 3, 21, tgt.config.vocab
    prompts = synthetic_prompts(B, P, V)
    hm = _hip_model(tgt, batch=B, l_max=96)
    dev = lambda t: t.to(torch.int32).cuda()
    zero = torch.zeros(B, dtype=torch.int32, device="cuda")
    hm.forward(dev(prompts[:, :-1]), zero, 0, skip_head=True)           # cache positions 0..P-2
    want_ids, _ = lm.generate_tokens(prompts, 12)
    cur = prompts[:, -1:].clone()
    pos = torch.full((B,), P - 1, dtype=torch.int32, device="cuda")
    got = []
    for j in range(7):
        ids, _ = hm.forward(dev(cur), pos, 0)
        cur = ids.cpu().long()
        got.append(cur)
        pos = pos + 1
    got = torch.cat(got, 1)
    assert torch.equal(got, want_ids[:, :7])
    # verify-shaped: feed (last, next 4 true tokens) at once; argmax at each position must
    # reproduce the continuation (same tokens the M=1 path produces)
    ver_in = torch.cat([got[:, -1:], want_ids[:, 7:11]], 1)
    ids, _ = hm.forward(dev(ver_in), pos, 0)
    assert torch.equal(ids.cpu().long(), want_ids[:, 7:12])


def test_verify_forward_logits_with_ragged_rows():
    """Rows at different cache lengths in one verify forward (pos_base per row)."""
    drf, tgt = tiny_pair(layer_gain=0.3)
    lm = OracleLM(tgt, precision="bf16")
    V = tgt.config.vocab
    lens = [5, 17, 30]
    B, M = len(lens), 5
    hm = _hip_model(tgt, batch=B, l_max=64)
    g = torch.Generator().manual_seed(8)
    seqs = [torch.randint(4, V, (n + M,), generator=g) for n in lens]
    # prefill each row separately (positions 0..n-1) into its own row of the cache (row0)
    for b, (n, s) in enumerate(zip(lens, seqs)):
        hm.forward(s[:n].to(torch.int32).view(1, -1).cuda(), torch.zeros(1, dtype=torch.int32, device="cuda"), 0,
                   skip_head=True, row0=b)
    new = torch.stack([s[n:] for n, s in zip(lens, seqs)], 0)
    pos = torch.tensor(lens, dtype=torch.int32, device="cuda")
    ids, logits = hm.forward(new.to(torch.int32).cuda(), pos, 0, want_logits=True)
    for b, (n, s) in enumerate(zip(lens, seqs)):
        want, _ = lm.forward(s.view(1, -1))
        want = want[0, n:]
        assert _rel_err(logits[b].float().cpu(), want) < 0.03
        assert torch.equal(ids[b].cpu().long(), logits[b].float().cpu().argmax(-1))


def test_forward_errors_are_loud():
    from specdec_hip import _abi
    from specdec_hip.engine import HipModel

    drf, tgt = tiny_pair()
    with pytest.raises(RuntimeError, match="no CPU path"):
        HipModel(tgt, batch=1, l_max=16)
    hm = _hip_model(tgt, batch=1, l_max=16)
    with pytest.raises(_abi.HipLibraryError, match="exceeds bound batch"):
        hm.forward(torch.zeros((2, 1), dtype=torch.int32, device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda"))


def _forward_all(mw, toks, env_cap, monkeypatch, l_max=160):
    """One forward over toks [B][L] from empty caches; env_cap = tokens per pass."""
    if env_cap is None:
        monkeypatch.delenv("SPECDEC_MAX_PASS_TOKENS", raising=False)
    else:
        monkeypatch.setenv("SPECDEC_MAX_PASS_TOKENS", str(env_cap))
    hm = _hip_model(mw, batch=toks.shape[0], l_max=l_max)
    assert hm.pass_tokens == (env_cap if env_cap is not None else hm.pass_tokens) and hm.pass_tokens in (9, 64, 128)   # the multi-token kernel covers these shapes
    pos0 = torch.zeros(toks.shape[0], dtype=torch.int32, device="cuda")
    ids, logits = hm.forward(toks.to(torch.int32).cuda(), pos0, 0, want_logits=True)
    torch.cuda.synchronize()
    return ids.cpu().long(), logits.float().cpu()


@pytest.mark.parametrize("name,B,L", [("llama", 1, 64), ("llama", 1, 100), ("llama", 2, 20), ("gpt2", 1, 50), ("gpt2", 3, 17)])
def test_multi_token_pass_matches_oracle_and_small_pass(name, B, L, monkeypatch):
    """10..64 tokens per pass (gemm_skinny.hip: chunked prefill, batched verify) on HF-initialised
    weights: logits within the bf16 forward tolerance of the oracle, fused argmax == argmax of the
    stored logits, and the same forward done in 9-token passes (gemv.hip) gives the same logits up
    to accumulation order (a few bf16 ulps: 1 % of the logit range)."""
    mw, toks, _ = load_hf_golden(name, dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(L * 7 + B)
    toks = torch.randint(0, mw.config.vocab, (B, L), generator=g)
    want, _ = OracleLM(mw, precision="bf16").forward(toks)
    ids64, lg64 = _forward_all(mw, toks, None, monkeypatch)
    ids9, lg9 = _forward_all(mw, toks, 9, monkeypatch)
    assert _rel_err(lg64, want) < 0.03, _rel_err(lg64, want)
    assert torch.equal(ids64, lg64.argmax(-1))
    assert _rel_err(lg64, lg9) < 0.01, _rel_err(lg64, lg9)
    assert (ids64 == ids9).float().mean().item() > 0.97


@pytest.mark.parametrize("B,M", [(8, 5), (4, 9), (7, 9), (2, 5), (3, 16), (1, 33), (1, 64)])
def test_batched_verify_shapes_on_synthetic_pair(B, M, monkeypatch):
    """BASELINE config 3/4 shapes (B rows x K+1 positions in ONE pass) on the synthetic tiny pair, rows
    at different cache lengths: token ids identical to the oracle (large margins by construction) and
    to the 9-token-pass path."""
    drf, tgt = tiny_pair()
    V = tgt.config.vocab
    lm = OracleLM(tgt, precision="bf16")
    lens = [3 + 5 * b for b in range(B)]
    g = torch.Generator().manual_seed(B * 100 + M)
    seqs = [torch.randint(4, V, (n + M,), generator=g) for n in lens]
    outs = {}
    for cap in (None, 9):
        if cap is None:
            monkeypatch.delenv("SPECDEC_MAX_PASS_TOKENS", raising=False)
        else:
            monkeypatch.setenv("SPECDEC_MAX_PASS_TOKENS", str(cap))
        hm = _hip_model(tgt, batch=B, l_max=128)
        assert hm.pass_tokens in ((64, 128) if cap is None else (9,))
        for b, (n, s) in enumerate(zip(lens, seqs)):
            hm.forward(s[:n].to(torch.int32).view(1, -1).cuda(), torch.zeros(1, dtype=torch.int32, device="cuda"), 0,
                       skip_head=True, row0=b)
        new = torch.stack([s[n:] for n, s in zip(lens, seqs)], 0)
        ids, _ = hm.forward(new.to(torch.int32).cuda(), torch.tensor(lens, dtype=torch.int32, device="cuda"), 0)
        outs[cap] = ids.cpu().long()
    for b, s in enumerate(seqs):
        want, _ = lm.forward(s.view(1, -1))
        assert torch.equal(outs[None][b], want[0, lens[b]:].argmax(-1)), (B, M, b)
    assert torch.equal(outs[None], outs[9])


@pytest.mark.parametrize("lens,M", [([300, 1500, 40], 5), ([2100], 1), ([900, 901], 9), ([700], 33)])
def test_long_context_split_kv_attention(lens, M, monkeypatch):
    """Contexts of hundreds to thousands of keys: the keys of a (row, kv head, query tile) are shared by
    several workgroups (one per 256 keys of the row's CURRENT length, merged by the last arrival). Logits
    against the oracle, and against the one-workgroup-per-tile path (SPECDEC_NO_ATTN_SPLIT)."""
    import dataclasses

    from helpers import TINY_TARGET
    from specdec_hip import weights as W

    tgt = W.synthetic_llama(dataclasses.replace(TINY_TARGET, max_pos=4096), seed=0, device="cpu", layer_gain=0.3)
    lm = OracleLM(tgt, precision="bf16")
    V, B = tgt.config.vocab, len(lens)
    g = torch.Generator().manual_seed(sum(lens) + M)
    seqs = [torch.randint(4, V, (n + M,), generator=g) for n in lens]
    outs = {}
    for split in (True, False):
        if split:
            monkeypatch.delenv("SPECDEC_NO_ATTN_SPLIT", raising=False)
        else:
            monkeypatch.setenv("SPECDEC_NO_ATTN_SPLIT", "1")
        hm = _hip_model(tgt, batch=B, l_max=max(lens) + M + 64)
        for b, (n, s) in enumerate(zip(lens, seqs)):
            hm.forward(s[:n].to(torch.int32).view(1, -1).cuda(), torch.zeros(1, dtype=torch.int32, device="cuda"), 0,
                       skip_head=True, row0=b)
        new = torch.stack([s[n:] for n, s in zip(lens, seqs)], 0)
        ids, logits = hm.forward(new.to(torch.int32).cuda(), torch.tensor(lens, dtype=torch.int32, device="cuda"), 0, want_logits=True)
        # a second forward at the same positions re-uses the arrival counters: they must be back at zero
        ids2, logits2 = hm.forward(new.to(torch.int32).cuda(), torch.tensor(lens, dtype=torch.int32, device="cuda"), 0, want_logits=True)
        assert torch.equal(ids, ids2) and torch.equal(logits, logits2)
        outs[split] = (ids.cpu().long(), logits.float().cpu())
    for b, (n, s) in enumerate(zip(lens, seqs)):
        want, _ = lm.forward(s.view(1, -1))
        want = want[0, n:]
        assert _rel_err(outs[True][1][b], want) < 0.03, (lens, b)
        assert torch.equal(outs[True][0][b], outs[True][1][b].argmax(-1))
    assert _rel_err(outs[True][1], outs[False][1]) < 0.01
