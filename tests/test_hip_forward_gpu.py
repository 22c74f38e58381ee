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
