"""sd_sample_token and the sampled step (generate_batch(do_sample=True)) on the GPU vs
oracle/sampling_ref.py and the oracle pipeline: identical token ids (integer output: bit-exact)."""

import numpy as np
import pytest
import torch

from helpers import synthetic_prompts, tiny_pair
from oracle import sampling_ref as S
from oracle.model_ref import OracleLM
from oracle.pipeline_ref import OraclePipeline

pytestmark = pytest.mark.gpu


def _hip(logits, T, top_k, top_p, seed, draw, stream):
    from specdec_hip.ops import sample_token_hip

    sid = torch.tensor([stream], dtype=torch.int32, device="cuda")
    return int(sample_token_hip(logits.cuda(), T, top_k, top_p, seed=seed, draw=draw, stream_ids=sid).item())


@pytest.mark.parametrize("V,dtype", [(50, torch.float32), (1000, torch.float32), (50257, torch.bfloat16),
                                     (128256, torch.bfloat16), (128256, torch.float32), (4099, torch.float16)])
def test_topk_sampler_matches_oracle(V, dtype):
    g = torch.Generator().manual_seed(V)
    for case in range(10):
        scale = [0.5, 2.0, 6.0][case % 3]
        x = (torch.randn(V, generator=g) * scale).to(dtype)
        top_k = [1, 5, 50, 50, 64, 300, 1000, 1024, 50, 17][case]
        top_p = [0.9, None, 0.9, 0.3, 1.0, 0.95, 0.9, 0.99, 0.9, 0.5][case]
        T = [0.7, 1.0, 0.7, 1.5, 0.2, 0.7, 2.0, 0.7, 5.0, 0.7][case]
        xf = x.float().numpy()
        for draw in range(4):
            want = S.sample_token_ref(xf, T, top_k, top_p, 1234 + case, draw, 3)
            got = _hip(x, T, top_k, top_p, 1234 + case, draw, 3)
            assert got == want, (V, dtype, case, draw, top_k, top_p, T)


def test_ties_special_values_and_flat_rows():
    """bf16 logits have few distinct values: ties at the top-k cut are decided by index. Constant and
    all -inf rows take every radix pass (the bucket never shrinks by value)."""
    g = torch.Generator().manual_seed(9)
    V = 128256
    coarse = (torch.randn(V, generator=g) * 1.5).to(torch.bfloat16).float().round()   # ~12 distinct values
    flat = torch.full((V,), 0.25)
    ninf = torch.full((V,), float("-inf"))
    mixed = coarse.clone()
    mixed[1000:1010] = float("inf")
    nan = coarse.clone()
    nan[77] = float("nan")
    negz = torch.zeros(300)
    negz[::2] = -0.0
    # large values concentrated in the elements 40 threads scan (16-byte vectors v with v % 1024 < 40):
    # the thread-maxima bound lets > 4096 candidates through, which takes the radix-select path
    conc = torch.randn(V, generator=g) * 0.1
    vec = torch.arange(V) // 8
    hot = (vec % 1024) < 40
    conc[hot] = 5.0 + torch.rand(int(hot.sum()), generator=g) * 3
    conc_ties = conc.to(torch.bfloat16).float()
    for name, x in (("coarse", coarse), ("flat", flat), ("ninf", ninf), ("mixed", mixed), ("nan", nan), ("negz", negz),
                    ("conc", conc), ("conc_ties", conc_ties), ("conc_bf16", conc.to(torch.bfloat16))):
        for top_k, top_p, T in ((50, 0.9, 0.7), (1024, None, 1.0), (7, 0.5, 3.0)):
            for draw in range(3):
                want = S.sample_token_ref(x.float().numpy(), T, top_k, top_p, 5, draw, 0)
                assert _hip(x, T, top_k, top_p, 5, draw, 0) == want, (name, top_k, top_p, T, draw)


@pytest.mark.parametrize("V,scale,top_p,T", [(300, 3.0, 0.9, 1.0), (2600, 1.0, 0.95, 0.7), (5000, 0.2, 0.6, 1.5), (128256, 4.0, 0.9, 0.7),
                                              (128256, 0.05, 0.3, 1.0), (4099, 6.0, 0.3, 0.7)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_vocabulary_nucleus_matches_oracle(V, scale, top_p, T, dtype):
    """top_p < 1 without top_k (the reference with top_k = None, pipeline.py:105-125): kept sets from a handful of tokens
    to tens of thousands (the flat V = 128256 row keeps ~38 k tokens: 38 rank blocks), token for token against
    oracle/sampling_ref.py, which is pinned by draws of the reference function ("sampling_nucleus")."""
    g = torch.Generator().manual_seed(V + int(scale * 100))
    x = (torch.randn(V, generator=g) * scale).to(dtype)
    xs = x.float().numpy()
    n_keep = len(S.nucleus_distribution(xs, T, top_p)[0])
    for draw in range(4):
        want = S.sample_token_ref(xs, T, None, top_p, 11, draw, 2)
        assert _hip(x, T, None, top_p, 11, draw, 2) == want, (V, scale, top_p, T, dtype, draw, n_keep)


def test_nucleus_special_rows():
    ninf = torch.full((777,), float("-inf"))
    ninf[5] = 1.0
    ninf[600] = 0.5
    nan = torch.randn(500, generator=torch.Generator().manual_seed(1))
    nan[77] = float("nan")
    const = torch.full((3000,), 0.25)
    for name, x in (("ninf", ninf), ("nan", nan), ("const", const)):
        for top_p in (0.9, 0.5):
            for draw in range(3):
                want = S.sample_token_ref(x.numpy(), 0.7, None, top_p, 5, draw, 0)
                assert _hip(x, 0.7, None, top_p, 5, draw, 0) == want, (name, top_p, draw)


def test_gumbel_path_rows_counters_and_refusals():
    from specdec_hip import _abi
    from specdec_hip.ops import sample_token_hip

    g = torch.Generator().manual_seed(3)
    B, R, V = 5, 3, 5000
    lg = (torch.randn(B * R, V, generator=g) * 2).to(torch.bfloat16)
    pos = torch.tensor([0, 2, 1, 1, 0], dtype=torch.int32, device="cuda")
    counters = torch.tensor([0, 4, 9, 2, 100], dtype=torch.int32, device="cuda")
    active = torch.tensor([1, 1, 0, 1, 1], dtype=torch.int32, device="cuda")
    sids = torch.tensor([10, 11, 12, 13, 14], dtype=torch.int32, device="cuda")
    for top_k, top_p in ((None, None), (40, 0.9)):
        c = counters.clone()
        out = sample_token_hip(lg.cuda(), 0.9, top_k, top_p, seed=77, pos=pos, rows_per_entry=R, draw_counters=c,
                               stream_ids=sids, active=active).cpu()
        for b in range(B):
            if b == 2:
                continue   # inactive entry: untouched, no draw consumed
            row = lg[b * R + int(pos[b])].float().numpy()
            assert int(out[b]) == S.sample_token_ref(row, 0.9, top_k, top_p, 77, int(counters[b]), 10 + b), (top_k, b)
        assert c.cpu().tolist() == [1, 5, 9, 3, 101]
    with pytest.raises(_abi.HipLibraryError, match="top_k"):
        sample_token_hip(lg.cuda(), 0.9, 2000, 0.9)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sample_token_hip(lg, 0.9, 10, 0.9)          # CPU tensor


def _pipe(drf, tgt, k):
    from src.specdec import HipLM, SpeculativePipeline

    return SpeculativePipeline(base_lm=HipLM(tgt.to("cuda")), draft_lm=HipLM(drf.to("cuda")),
                               controller="fixed", controller_params={"k": k}, seed=1234)


@pytest.mark.parametrize("k", [2, 4])
def test_sampled_step_loop_against_oracle_rules(k):
    """do_sample=True: greedy draft + greedy verify + SAMPLED token after the accepted prefix, inside the
    captured step. Sampling has no argmax margin to absorb the last-bit differences between the GPU
    forward and a CPU forward, so the run is checked step by step against the oracle APPLIED TO THE
    DEVICE'S OWN LOGITS: the sampled token must be exactly the oracle sampler's draw on the stored
    logits row of position accept_len (draw index = sampled steps of the row so far, stream = row), and
    the row state after the host rules must equal the oracle rules fed with that token. A high
    temperature makes the draw leave the argmax on most steps (asserted)."""
    from oracle.pipeline_ref import RowState, step_rules_batch
    from specdec_hip.engine import HipSpecDec

    drf, tgt = tiny_pair(flip_fraction=0.25)
    V, eos = tgt.config.vocab, tgt.config.eos_token_id
    prompts = synthetic_prompts(4, 12, V).tolist()
    sp = {"temperature": 20.0, "top_k": 50, "top_p": 0.95, "seed": 4321}
    pipe = _pipe(drf, tgt, k)
    sess = pipe.start_session(prompts, 24, HipSpecDec.EMIT_BONUS, sp)
    orows = [RowState(seq=list(p)) for p in prompts]
    left_argmax = sampled_steps = 0
    while sess.step < 24 and sess.any_active():
        active = [r.active for r in sess.rows]
        draws = [r.draws for r in sess.rows]
        assert sess.advance()
        rec = sess.last_record
        logits = sess.loop.step_logits.float().cpu().numpy()          # [B][K+1][V], this step's verify logits
        for b in range(4):
            if not active[b]:
                continue
            a = int(rec.accept_len[b])
            d, t = [int(x) for x in rec.draft_tokens[b]], [int(x) for x in rec.target_ids[b]]
            for i in range(a + 1):                                     # fused argmax == argmax of the stored logits
                assert t[i] == int(np.argmax(logits[b, i])), (sess.step, b, i)
            assert a == next((i for i in range(k) if d[i] != t[i]), k)
            want_tok = S.sample_token_ref(logits[b, a], sp["temperature"], sp["top_k"], sp["top_p"], sp["seed"], draws[b], b)
            assert int(rec.new_tokens[b][a]) == want_tok, (sess.step, b, a)
            sampled_steps += 1
            left_argmax += int(want_tok != t[a])

            def bonus_at(pos, _b=b):
                return S.sample_token_ref(logits[_b, pos], sp["temperature"], sp["top_k"], sp["top_p"], sp["seed"], draws[_b], _b)
            step_rules_batch(orows[b], k, a, d, t, 24, eos, V, bonus_at)
            r = sess.rows[b]
            assert (r.seq, r.generated, r.active, r.proposed, r.accepted) == \
                   (orows[b].seq, orows[b].generated, orows[b].active, orows[b].proposed, orows[b].accepted), (sess.step, b)
            assert r.draws == draws[b] + 1
    assert sampled_steps >= 20 and left_argmax >= sampled_steps // 4, (sampled_steps, left_argmax)
    # the GPU logits the draws came from are the oracle model's logits up to the bf16 forward tolerance
    lm = OracleLM(tgt, "bf16")
    seq = sess.rows[0].seq
    lg, _ = lm.forward(torch.tensor([seq[:20]], dtype=torch.int64))
    assert lg.shape[-1] == V
    # reproducible, seed-dependent, and the cached loop returns to greedy afterwards
    run = lambda seed: [r["generated_tokens"] for r in pipe.generate_batch(
        prompts, max_tokens=24, temperature=20.0, do_sample=True, top_k=50, top_p=0.95, seed=seed)]
    first = run(4321)
    assert first == [r.generated for r in sess.rows]
    assert run(4321) == first and run(1) != first
    oracle = OraclePipeline(lm, OracleLM(drf, "bf16"), k=k, eos_token_id=eos)
    greedy = oracle.generate_batch(prompts, 24)
    again = pipe.generate_batch(prompts, max_tokens=24, do_sample=False)
    assert [r["generated_tokens"] for r in again] == [r["generated_tokens"] for r in greedy]
    assert first != [r["generated_tokens"] for r in greedy]


def test_sampled_default_config_and_refusals():
    """The reference's default sampler (T=0.7, top_k=50, top_p=0.9, configs/specdec.yaml:11-15) on a
    row alone equals the same row inside a batch (Philox stream = batch index)."""
    drf, tgt = tiny_pair(flip_fraction=0.25)
    prompts = synthetic_prompts(3, 10, tgt.config.vocab).tolist()
    pipe = _pipe(drf, tgt, 4)
    got = pipe.generate_batch(prompts, max_tokens=16, do_sample=True)
    oracle = OraclePipeline(OracleLM(tgt, "bf16"), OracleLM(drf, "bf16"), k=4, eos_token_id=tgt.config.eos_token_id)
    want = oracle.generate_batch(prompts, 16, sampling={"temperature": 0.7, "top_k": 50, "top_p": 0.9, "seed": 1234})
    assert [r["generated_tokens"] for r in got] == [r["generated_tokens"] for r in want]
    # top_k = None with top_p < 1: the full-vocabulary nucleus inside the captured step
    got = pipe.generate_batch(prompts, max_tokens=12, do_sample=True, top_k=None, top_p=0.9)
    want = oracle.generate_batch(prompts, 12, sampling={"temperature": 0.7, "top_k": None, "top_p": 0.9, "seed": 1234})
    assert [r["generated_tokens"] for r in got] == [r["generated_tokens"] for r in want]
    # generate(do_sample=True) is another rule (sampled draft, greedy verification): tests/test_hip_pipeline_gpu.py
