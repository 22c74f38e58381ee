"""Chained GEMV launches (csrc/gemv_chain.hip: [out-projection -> gate/up] and [down projection -> next QKV] as one
kernel each, with an in-kernel fence-free grid barrier) against the same forward with separate launches: the phase
bodies are the gemv.hip kernel, so logits must be BIT-identical, for every token count the small kernel takes, and
the barrier must never time out. The chained path is opt-in (SPECDEC_CHAIN_PAIRS; measured slower than separate
launches) — this test keeps it correct."""

import os

import pytest
import torch

from helpers import synthetic_prompts
from specdec_hip import weights as W

pytestmark = pytest.mark.gpu

# smallest Llama shape whose four matrices split over exactly 256 workgroups in whole 32-k slices
MID = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=3, d_model=512, n_heads=8, n_kv_heads=4, head_dim=64, d_ff=1024,
                    vocab=1000, max_pos=512, rope_theta=500000.0, rope_scaling=None, tie_embeddings=False, name="mid")


def _model(mw, chain, batch, l_max):
    from specdec_hip.engine import HipModel

    old = os.environ.get("SPECDEC_CHAIN_PAIRS")
    os.environ["SPECDEC_CHAIN_PAIRS"] = "3" if chain else "0"   # read when the model is bound
    try:
        return HipModel(mw, batch=batch, l_max=l_max)
    finally:
        if old is None:
            os.environ.pop("SPECDEC_CHAIN_PAIRS", None)
        else:
            os.environ["SPECDEC_CHAIN_PAIRS"] = old


@pytest.mark.parametrize("B,M", [(1, 1), (1, 2), (1, 3), (1, 5), (1, 9), (2, 1), (2, 4), (3, 3)])
def test_chained_forward_is_bit_identical(B, M):
    mw = W.synthetic_llama(MID, seed=3, device="cuda", layer_gain=0.3)
    on, off = _model(mw, True, B, 128), _model(mw, False, B, 128)
    assert on.chain_status() == (True, False) and off.chain_status() == (False, False)
    P = 19
    prompts = synthetic_prompts(B, P + 3 * M, MID.vocab).to(torch.int32).cuda()
    zero = torch.zeros(B, dtype=torch.int32, device="cuda")
    for hm in (on, off):
        hm.forward(prompts[:, :P].contiguous(), zero, 0, skip_head=True)   # prefill (multi-token kernel, unchained)
    pos = torch.full((B,), P, dtype=torch.int32, device="cuda")
    for j in range(3):   # three passes over the same caches: epochs advance, the KV written by a chained QKV is read back
        toks = prompts[:, P + j * M:P + (j + 1) * M].contiguous()
        ids_on, lg_on = on.forward(toks, pos, 0, want_logits=True)
        ids_off, lg_off = off.forward(toks, pos, 0, want_logits=True)
        assert torch.equal(lg_on, lg_off), (j, (lg_on.float() - lg_off.float()).abs().max().item())
        assert torch.equal(ids_on, ids_off)
        pos = pos + M
    assert on.chain_status() == (True, False)


def test_chained_full_size_layers_many_epochs():
    """Llama-3.2-1B shapes (4 layers): 64 chained forwards in a row — every barrier completes, results stay equal."""
    import dataclasses

    cfg = dataclasses.replace(W.LLAMA_3_2_1B, n_layers=4, vocab=4096, name="1b-4l")
    mw = W.synthetic_llama(cfg, seed=1, device="cuda")
    on, off = _model(mw, True, 1, 128), _model(mw, False, 1, 128)
    assert on.chain_status()[0]
    tok = torch.tensor([[17]], dtype=torch.int32, device="cuda")
    pos = torch.zeros(1, dtype=torch.int32, device="cuda")
    for j in range(64):
        a, la = on.forward(tok, pos, 0, want_logits=True)
        b, lb = off.forward(tok, pos, 0, want_logits=True)
        assert torch.equal(la, lb), j
        tok, pos = a.to(torch.int32), pos + 1
    assert on.chain_status() == (True, False)
