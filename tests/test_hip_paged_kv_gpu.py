"""Paged KV (sd_model_bind_paged): rows own pages of a shared pool through a block table instead of
l_max positions each. Same arithmetic as the dense cache, so logits must be BIT-identical to the dense engine's, and the
step loop over paged engines must equal the oracle like the dense one."""

import json
import os

import pytest
import torch

import cases
from helpers import load_hf_golden, synthetic_prompts, tiny_pair
from oracle.model_ref import OracleLM
from oracle.pipeline_ref import OraclePipeline

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _engines(mw, batch, l_max, page_len, n_pages=None):
    from specdec_hip.engine import HipModel

    mw = mw.to("cuda")
    return HipModel(mw, batch=batch, l_max=l_max), HipModel(mw, batch=batch, l_max=l_max, page_len=page_len, n_pages=n_pages)


@pytest.mark.parametrize("page_len", [32, 64])
def test_paged_forward_is_bit_identical_to_dense(page_len):
    """Prefill (tiled, > 9 tokens), ragged rows, M = 1 decode steps across page boundaries, an M = 5 verify-shaped pass:
    ids and logits equal the dense engine bit for bit. Pages are handed out interleaved across rows (scrambled pool order)."""
    _, tgt = tiny_pair()
    B, V = 3, tgt.config.vocab
    dense, paged = _engines(tgt, B, 192, page_len)
    # scramble: rows reserve in an order that interleaves their pages in the pool
    for n in (1, 40, 2 * page_len + 3):
        for b in (2, 0, 1):
            paged.reserve(b, n)
    lens = [21, 70, 33]
    prompts = [synthetic_prompts(1, L, V, seed=50 + i)[0] for i, L in enumerate(lens)]
    dev = lambda t: t.to(torch.int32).cuda()
    for b, p in enumerate(prompts):
        z = torch.zeros(1, dtype=torch.int32, device="cuda")
        for m in (dense, paged):
            m.forward(dev(p[None, :-1]), z, 0, skip_head=True, row0=b)
    cur = torch.stack([p[-1:] for p in prompts])
    pos = torch.tensor([L - 1 for L in lens], dtype=torch.int32, device="cuda")
    for j in range(2 * page_len + 5):           # every row crosses at least two page boundaries
        i_d, l_d = dense.forward(dev(cur), pos, 0, want_logits=True)
        i_p, l_p = paged.forward(dev(cur), pos, 0, want_logits=True)
        assert torch.equal(i_d, i_p) and torch.equal(l_d, l_p), j
        cur = i_d.cpu().long()
        pos = pos + 1
    toks = torch.cat([cur, synthetic_prompts(B, 4, V, seed=7)], dim=1)
    i_d, l_d = dense.forward(dev(toks), pos, 0, want_logits=True)
    i_p, l_p = paged.forward(dev(toks), pos, 0, want_logits=True)
    assert torch.equal(i_d, i_p) and torch.equal(l_d, l_p)
    # the oracle agrees on the tokens of row 0's greedy continuation
    lm = OracleLM(tgt, "bf16")
    want, _ = lm.generate_tokens(prompts[0][None], 6)
    paged2 = _engines(tgt, 1, 192, page_len)[1]
    z = torch.zeros(1, dtype=torch.int32, device="cuda")
    paged2.forward(dev(prompts[0][None, :-1]), z, 0, skip_head=True)
    c, p0, got = prompts[0][None, -1:], torch.tensor([lens[0] - 1], dtype=torch.int32, device="cuda"), []
    for _ in range(6):
        ids, _ = paged2.forward(dev(c), p0, 0)
        c = ids.cpu().long()
        got.append(int(c))
        p0 = p0 + 1
    assert got == want[0].tolist()


def test_paged_long_context_split_kv(monkeypatch):
    """> 512 keys: the keys of a tile are shared by several workgroups (split-KV); a 32-key block never straddles a page."""
    # (both engines absorb the prompt through the same kernels: the GEMM prefill path is for dense caches only, and this test is
    #  about bit-identical attention over paged and dense caches)
    monkeypatch.setenv("SPECDEC_NO_GEMM_PREFILL", "1")
    _, tgt = tiny_pair()
    V = tgt.config.vocab
    dense, paged = _engines(tgt, 1, 2048, 128)
    L = 1300
    p = synthetic_prompts(1, L, V, seed=99) % 500 + 4
    dev = lambda t: t.to(torch.int32).cuda()
    z = torch.zeros(1, dtype=torch.int32, device="cuda")
    for m in (dense, paged):
        m.forward(dev(p[:, :-1]), z, 0, skip_head=True)
    pos = torch.tensor([L - 1], dtype=torch.int32, device="cuda")
    toks = torch.cat([p[:, -1:], synthetic_prompts(1, 4, V, seed=3)], dim=1)
    i_d, l_d = dense.forward(dev(toks), pos, 0, want_logits=True)
    i_p, l_p = paged.forward(dev(toks), pos, 0, want_logits=True)
    assert torch.equal(i_d, i_p) and torch.equal(l_d, l_p)


def test_pool_accounting_and_exhaustion():
    _, tgt = tiny_pair()
    _, paged = _engines(tgt, 2, 256, 32, n_pages=5)
    paged.reserve(0, 65)            # 3 pages
    assert paged.pages_in_use() == 3
    paged.reserve(1, 64)            # 2 pages
    assert paged.pages_in_use() == 5
    with pytest.raises(RuntimeError, match="pool exhausted"):
        paged.reserve(1, 65)
    paged.release(0)
    assert paged.pages_in_use() == 2
    paged.reserve(1, 96)
    assert paged.pages_in_use() == 3
    t = paged.block_table.cpu()
    assert len(set(t[1, :3].tolist())) == 3


@pytest.mark.parametrize("pname", ["structured", "repeating"])
def test_pipeline_over_paged_engines_matches_oracle(pname):
    """The captured step loop with steps launched ahead: pages are reserved before each launch for every step in flight.
    Tokens / counters equal the oracle; 'repeating' drives the host's rewinds (cache rebuild into the row's pages)."""
    from src.specdec import HipLM, SpeculativePipeline

    drf, tgt = cases.g8_pairs(torch.bfloat16)[pname]
    with open(os.path.join(GOLD, "pipeline_golden.json")) as f:
        runs = json.load(f)[pname]["runs"]
    prompts = [r["prompt_ids"] for r in runs[:3]]
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda"), kv_page_len=32), draft_lm=HipLM(drf.to("cuda"), kv_page_len=32),
                               controller="fixed", controller_params={"k": 4}, seed=1234)
    got = pipe.generate_batch(prompts, max_tokens=70, do_sample=False)
    want = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4, eos_token_id=2).generate_batch(prompts, 70)
    for g, w in zip(got, want):
        assert g["generated_tokens"] == w["generated_tokens"]
        assert (g["proposed"], g["accepted"]) == (w["proposed"], w["accepted"])
    rt = next(iter(pipe._runtimes.values()))
    assert rt["target"].page_len == 32 and rt["target"].pages_in_use() >= 3


def test_continuous_batching_reuses_pages():
    """generate_many over paged engines with a pool far smaller than rows x l_max: finished rows hand their pages back when
    the slot is admitted to the next prompt; every result equals the prompt's own run."""
    from src.specdec import HipLM, SpeculativePipeline

    drf, tgt = tiny_pair()
    V = tgt.config.vocab
    prompts = synthetic_prompts(7, 11, V).tolist()
    pages = 2 * 6          # 2 rows x (11 + 20 + margins) / 32 -> a few pages each; 12 is well under 2 x (l_max / 32)
    pipe = SpeculativePipeline(base_lm=HipLM(tgt.to("cuda"), kv_page_len=32, kv_pages=pages), draft_lm=HipLM(drf.to("cuda"), kv_page_len=32, kv_pages=pages),
                               controller="fixed", controller_params={"k": 3}, seed=1234)
    got = pipe.generate_many(prompts, max_tokens=20, batch_size=2, do_sample=False)
    oracle = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=3, eos_token_id=pipe.base_lm.get_tokenizer_info()["eos_token_id"])
    for p, g in zip(prompts, got):
        w = oracle.generate_batch([p], 20)[0]
        assert g["generated_tokens"] == w["generated_tokens"]
        assert (g["proposed"], g["accepted"]) == (w["proposed"], w["accepted"])
    rt = next(iter(pipe._runtimes.values()))
    assert rt["target"].pages_in_use() <= pages
