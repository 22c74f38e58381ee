"""Persistent forward (csrc/persist.hip: one launch per pass of <= 2 tokens) against the launch-per-operator
forward it replaces, stage by stage, and against the CPU oracle.

Reference being replaced: the k-step draft loop, /root/reference/src/specdec/models/hf_wrappers.py:417-539.
Tolerances: both paths round to bf16 at the same points (normalised rows, q/k/v, attention rows, residual stream,
activations, logits) and accumulate in fp32; they differ in the ORDER of the fp32 sums (3 interleaved K slices vs
16/n_tiles contiguous ones; the RMSNorm statistic likewise), which moves a bf16 value by one ulp (here: 2^-8 of the
value, half the spacing at the bottom of a binade) and, through a flipped rounding of a normalised input, a GEMV output by
a few. Stage checks therefore bound the error against the tensor's scale: every element within 2^-6 of max|value|, the
mean error below 2^-9 of it — a wrong stage is O(1) of the scale off."""

import pytest
import torch

from helpers import synthetic_prompts
from oracle.model_ref import OracleLM
from specdec_hip import weights as W

pytestmark = pytest.mark.gpu

TOY = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=3, d_model=256, n_heads=4, n_kv_heads=2, head_dim=64, d_ff=512,
                    vocab=2048, max_pos=2048, rope_theta=500000.0, tie_embeddings=False, name="persist-toy")
TOY128 = W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=2, d_model=384, n_heads=3, n_kv_heads=1, head_dim=128, d_ff=1024,
                       vocab=3001, max_pos=2048, rope_theta=500000.0, tie_embeddings=False, name="persist-toy128")


def _shape_1b(n_layers):
    return W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=n_layers, d_model=2048, n_heads=32, n_kv_heads=8, head_dim=64, d_ff=8192,
                         vocab=128256, max_pos=4096, rope_theta=500000.0, tie_embeddings=True, name=f"1b-shape-{n_layers}l")


def _shape_3b(n_layers):
    return W.ModelConfig(arch=W.ARCH_LLAMA, n_layers=n_layers, d_model=3072, n_heads=24, n_kv_heads=8, head_dim=128, d_ff=8192,
                         vocab=128256, max_pos=4096, rope_theta=500000.0, tie_embeddings=True, name=f"3b-shape-{n_layers}l")


def _engines(mw, batch, l_max, monkeypatch, max_t=2):
    """(persistent engine, launch-per-operator engine) over the same weights"""
    from specdec_hip.engine import HipModel

    mw = mw.to("cuda")
    monkeypatch.setenv("SPECDEC_PERSIST_MAX_T", str(max_t))
    a = HipModel(mw, batch=batch, l_max=l_max)
    monkeypatch.setenv("SPECDEC_PERSIST_MAX_T", "0")
    b = HipModel(mw, batch=batch, l_max=l_max)
    monkeypatch.delenv("SPECDEC_PERSIST_MAX_T")
    assert a.persist_tokens == max_t and b.persist_tokens == 0
    return a, b


def _close(got, want, what, floor=None):
    """bf16 tensors equal up to summation-order noise. Per element: |got - want| <= 12 ulp of max(|want_i|, floor), and never
    more than 2^-6 of the tensor's scale (round 3's whole-tensor bound = 4 ulp of the largest element). So an element well
    below the scale is held to ITS OWN bf16 spacing down to `floor` (an absolute magnitude; default 1/8 of the scale): below
    the floor the error is the noise of the SUMMANDS (inputs that differ by an ulp between the two paths, fp32 sums in another
    order), which does not shrink with the result. Call sites say how far down a tensor can be trusted element by element: a
    plain GEMV output far (scale / 64), an attention row not at all (floor = scale: averages of V rows under bf16-rounded
    probabilities whose running maxima depend on how the keys are split over waves). Measured over every case of this file
    (round 4, in ulp of max(|x|, floor)): <= 4 for q / K / V / attention / activation rows, 5.2 for residual rows, 8.0 for
    logits (both on the random_init toy models, whose rows are not damped) — the bound leaves 1.5 x.
    And on average: mean error below 2^-9 of the scale. A wrong stage is O(1) of the scale off."""
    g, w = got.float(), want.float()
    scale = w.abs().max().item()
    fl = scale / 8 if floor is None else min(float(floor), scale)
    err = (g - w).abs()
    ulp = torch.exp2(torch.floor(torch.log2(w.abs().clamp_min(max(fl, 1e-30)))) - 7.0)     # spacing of bf16 values at max(|w|, floor)
    ratio = (err / (12.0 * ulp)).max().item()
    worst, mean = err.max().item() / scale, err.mean().item() / scale
    assert ratio <= 1.0 and worst <= 2.0 ** -6 and mean <= 2.0 ** -9, (
        f"{what}: worst element at {12 * ratio:.1f} ulp of max(|x|, floor) (bound 12), max error {worst:.2e} of scale, mean {mean:.2e} "
        f"(scale {scale:.3g}, floor {fl:.3g})")


def _dev(t):
    return t.to(torch.int32).cuda()


@pytest.mark.parametrize("init", ["synthetic", "random_init"])
@pytest.mark.parametrize("cfg", [TOY, TOY128, _shape_1b(1), _shape_3b(1)], ids=lambda c: c.name)
def test_stages_match_launch_path(cfg, init, monkeypatch):
    """One-layer view of every fused stage (the taps hold the LAST layer's values): q after RoPE, attention rows, MLP
    activation, residual stream, logits, K/V appended to the cache — 1-token and 2-token passes behind a 21-token prefix."""
    one = W.ModelConfig(**{**cfg.__dict__, "n_layers": 1}) if cfg.n_layers != 1 else cfg
    if init == "random_init":   # non-trivial norm weights, small activations
        if one.vocab > 4096:
            pytest.skip("one initialisation is enough at the full vocabulary")
        mw = W.random_init(one, seed=5, std=0.05, device="cpu")
    else:
        mw = W.synthetic_llama(one, seed=5, device="cpu", layer_gain=1.0)
    B, P = 1, 21
    pa, la = _engines(mw, B, 128, monkeypatch)
    prompts = synthetic_prompts(B, P + 3, one.vocab)
    zero = torch.zeros(B, dtype=torch.int32, device="cuda")
    for hm in (pa, la):
        hm.forward(_dev(prompts[:, :P]), zero, 0, skip_head=True)     # > 2 tokens: launch path in both engines
    pos = torch.full((B,), P, dtype=torch.int32, device="cuda")
    for M, off in ((1, 0), (2, 1)):
        toks = _dev(prompts[:, P + off:P + off + M])
        ids_p, lg_p = pa.forward(toks, pos, off, want_logits=True, logits_dtype=torch.bfloat16)
        ids_l, lg_l = la.forward(toks, pos, off, want_logits=True, logits_dtype=torch.bfloat16)
        assert pa.engine_status() == 0
        # q: a plain GEMV output. Attention rows are averages of V rows weighted by bf16-rounded probabilities whose
        # reference maxima depend on how the keys are split over waves (3 here, 4 there): the error scales with the
        # ROW, not the element (floor = the tensor's scale). Everything downstream inherits a fraction of that.
        for which, name, fl in ((pa.DEBUG_Q, "q", 1 / 64), (pa.DEBUG_ATTN, "attention", 1.0), (pa.DEBUG_ACT, "activation", 1 / 8),
                                (pa.DEBUG_X, "residual", 1 / 8)):
            want = la.debug_rows(which, M)
            _close(pa.debug_rows(which, M), want, f"{one.name} M={M} {name}", floor=fl * want.float().abs().max().item())
        _close(lg_p, lg_l, f"{one.name} M={M} logits", floor=lg_l.float().abs().max().item() / 8)
        assert torch.equal(ids_p.cpu().long(), lg_p.float().cpu().argmax(-1)), "fused argmax vs stored logits"
        kp, vp = pa.kv_view()
        kl, vl = la.kv_view()
        lo, hi = P + off, P + off + M
        _close(kp[:, :, :, lo:hi], kl[:, :, :, lo:hi], f"{one.name} M={M} K rows")
        _close(vp[:, :, :, :, lo:hi], vl[:, :, :, :, lo:hi], f"{one.name} M={M} V rows")


@pytest.mark.parametrize("cfg", [TOY, TOY128], ids=lambda c: c.name)
def test_decode_tokens_match_oracle(cfg, monkeypatch):
    """Greedy decode through the persistent launch (1-token passes, then 2-token passes) reproduces the CPU oracle's
    tokens; logits within the tolerance of tests/test_hip_forward_gpu.py (3 % of the largest logit)."""
    mw = W.synthetic_llama(cfg, seed=11, device="cpu", layer_gain=0.05)
    lm = OracleLM(mw, precision="bf16")
    B, P = 1, 14
    prompts = synthetic_prompts(B, P, cfg.vocab)
    want_ids, want_logits = lm.generate_tokens(prompts, 10)
    pa, _ = _engines(mw, B, 96, monkeypatch)
    zero = torch.zeros(B, dtype=torch.int32, device="cuda")
    pa.forward(_dev(prompts[:, :-1]), zero, 0, skip_head=True)
    cur = prompts[:, -1:].clone()
    pos = torch.full((B,), P - 1, dtype=torch.int32, device="cuda")
    got = []
    for j in range(6):
        ids, logits = pa.forward(_dev(cur), pos, 0, want_logits=True)
        rel = (logits.float().cpu()[:, 0] - want_logits[:, j]).abs().max().item() / want_logits[:, j].abs().max().item()
        assert rel < 0.03, (j, rel)
        cur = ids.cpu().long()
        got.append(cur)
        pos = pos + 1
    got = torch.cat(got, 1)
    assert torch.equal(got, want_ids[:, :6])
    # 2-token passes: (last, next true token) -> the two following tokens
    ver_in = torch.cat([got[:, -1:], want_ids[:, 6:7]], 1)
    ids, _ = pa.forward(_dev(ver_in), pos, 0)
    assert torch.equal(ids.cpu().long(), want_ids[:, 6:8])
    assert pa.engine_status() == 0


def test_two_rows_one_pass_and_row_offsets(monkeypatch):
    """T = 2 as two ROWS of one token (ragged lengths, attention units of both rows), and a pass on row 1 of the bound batch."""
    mw = W.synthetic_llama(TOY, seed=2, device="cpu", layer_gain=0.3)
    B = 2
    pa, la = _engines(mw, B, 160, monkeypatch)
    lens = [37, 64]
    V = TOY.vocab
    g = torch.Generator().manual_seed(3)
    seqs = [torch.randint(4, V, (n + 2,), generator=g) for n in lens]
    for hm in (pa, la):
        for b, n in enumerate(lens):
            hm.forward(_dev(seqs[b][:n].view(1, -1)), torch.zeros(1, dtype=torch.int32, device="cuda"), 0, skip_head=True, row0=b)
    pos = torch.tensor(lens, dtype=torch.int32, device="cuda")
    toks = torch.stack([seqs[b][lens[b]:lens[b] + 1] for b in range(B)], 0)
    ids_p, lg_p = pa.forward(_dev(toks), pos, 0, want_logits=True, logits_dtype=torch.bfloat16)
    ids_l, lg_l = la.forward(_dev(toks), pos, 0, want_logits=True, logits_dtype=torch.bfloat16)
    _close(lg_p, lg_l, "two rows logits", floor=lg_l.float().abs().max().item() / 8)
    assert torch.equal(ids_p, ids_l)
    # row 1 alone (row0 = 1), next position
    t1 = seqs[1][lens[1] + 1:lens[1] + 2].view(1, 1)
    p1 = torch.tensor([lens[1] + 1], dtype=torch.int32, device="cuda")
    ids_p, lg_p = pa.forward(_dev(t1), p1, 0, want_logits=True, logits_dtype=torch.bfloat16, row0=1)
    ids_l, lg_l = la.forward(_dev(t1), p1, 0, want_logits=True, logits_dtype=torch.bfloat16, row0=1)
    _close(lg_p, lg_l, "row offset logits", floor=lg_l.float().abs().max().item() / 8)
    assert torch.equal(ids_p, ids_l)
    assert pa.engine_status() == 0


@pytest.mark.parametrize("p0", [0, 1, 31, 32, 33, 95, 96, 700])
def test_attention_block_edges(p0, monkeypatch):
    """Cached lengths around the 32-key block edges and the 3-wave split (0 keys: only the new position; 700: 22 blocks)."""
    mw = W.synthetic_llama(TOY, seed=7, device="cpu", layer_gain=0.5)
    pa, la = _engines(mw, 1, 768, monkeypatch)
    seq = synthetic_prompts(1, p0 + 3, TOY.vocab, seed=99)
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    pos = torch.tensor([p0], dtype=torch.int32, device="cuda")
    for hm in (pa, la):
        if p0 > 0:
            hm.forward(_dev(seq[:, :p0]), zero, 0, skip_head=True)
    for M, off in ((1, 0), (2, 1)):
        toks = _dev(seq[:, p0 + off:p0 + off + M])
        _, lg_p = pa.forward(toks, pos, off, want_logits=True, logits_dtype=torch.bfloat16)
        _, lg_l = la.forward(toks, pos, off, want_logits=True, logits_dtype=torch.bfloat16)
        want = la.debug_rows(la.DEBUG_ATTN, M)
        _close(pa.debug_rows(pa.DEBUG_ATTN, M), want, f"p0={p0} M={M} attention (last layer)", floor=want.float().abs().max().item())
        _close(lg_p, lg_l, f"p0={p0} M={M} logits", floor=lg_l.float().abs().max().item() / 8)
    assert pa.engine_status() == 0


def test_many_launches_keep_state(monkeypatch):
    """300 consecutive persistent launches (tags advance per launch, buffers alternate per layer): tokens stay equal to the
    launch path's, status stays 0."""
    mw = W.synthetic_llama(TOY, seed=4, device="cpu", layer_gain=0.05)
    pa, la = _engines(mw, 1, 384, monkeypatch)
    seq = synthetic_prompts(1, 8, TOY.vocab, seed=5)
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    outs = []
    for hm in (pa, la):
        hm.forward(_dev(seq[:, :-1]), zero, 0, skip_head=True)
        cur = _dev(seq[:, -1:])
        pos = torch.tensor([7], dtype=torch.int32, device="cuda")
        toks = []
        for _ in range(300):
            cur, _ = hm.forward(cur, pos, 0)
            toks.append(cur)
            pos = pos + 1
        outs.append(torch.cat(toks, 1).cpu())
    assert torch.equal(outs[0], outs[1])
    assert pa.engine_status() == 0


def test_full_depth_1b_shape_matches_launch_path(monkeypatch):
    """16 layers at Llama-3.2-1B dimensions, random weights without margin engineering: logits of a 1-token and a 2-token
    pass agree with the launch path within the summation-order ulp compounded over depth (looser: 2.5 % of the largest logit),
    the argmax agrees wherever the launch path's own top-2 margin exceeds that band."""
    cfg = _shape_1b(16)
    mw = W.random_init(cfg, seed=0, device="cpu")
    pa, la = _engines(mw, 1, 256, monkeypatch)
    seq = synthetic_prompts(1, 40, cfg.vocab, seed=21)
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    for hm in (pa, la):
        hm.forward(_dev(seq[:, :37]), zero, 0, skip_head=True)
    pos = torch.tensor([37], dtype=torch.int32, device="cuda")
    for M, off in ((1, 0), (2, 1)):
        toks = _dev(seq[:, 37 + off:37 + off + M])
        ids_p, lg_p = pa.forward(toks, pos, off, want_logits=True)
        ids_l, lg_l = la.forward(toks, pos, off, want_logits=True)
        band = 0.025 * lg_l.abs().max().item()
        assert (lg_p - lg_l).abs().max().item() <= band
        top2 = lg_l.topk(2, dim=-1).values
        sure = (top2[..., 0] - top2[..., 1]) > 2 * band
        assert torch.equal(ids_p[sure], ids_l[sure])
    assert pa.engine_status() == 0


def test_a_launch_that_cannot_complete_gives_up_and_reports(monkeypatch):
    """Every wait of the persistent launch is bounded (50 ms on the 100 MHz clock): launched ONE WORKGROUP SHORT (test hook
    SPECDEC_PERSIST_TEST_DROP_WG: the granules of workgroup 255 and the attention unit it hosts never appear) the launch must
    come back on its own, leave a non-zero status word for the host, and the engine must serve the next pass again once the
    hook is gone (tags advance per launch, so the incomplete buffers are simply stale)."""
    import time

    mw = W.synthetic_llama(TOY, seed=4, device="cpu", layer_gain=0.05)
    pa, la = _engines(mw, 1, 128, monkeypatch)
    seq = synthetic_prompts(1, 8, TOY.vocab, seed=5)
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    for hm in (pa, la):
        hm.forward(_dev(seq[:, :-1]), zero, 0, skip_head=True)
    pos = torch.tensor([7], dtype=torch.int32, device="cuda")
    assert pa.engine_status() == 0
    monkeypatch.setenv("SPECDEC_PERSIST_TEST_DROP_WG", "1")
    t0 = time.time()
    pa.forward(_dev(seq[:, -1:]), pos, 0)
    torch.cuda.synchronize()
    took = time.time() - t0
    monkeypatch.delenv("SPECDEC_PERSIST_TEST_DROP_WG")
    status = pa.engine_status()
    assert status != 0, "an incomplete launch must leave its give-up code"
    assert took < 5.0, f"the launch took {took:.2f} s to give up"
    # the host's pinned copy of the word says the same without a copy (the pass has been synchronised with) ...
    assert pa.health() != 0
    # ... and a further pass is refused rather than piled onto invalid rows
    from specdec_hip.engine import EngineGaveUp
    with pytest.raises(EngineGaveUp):
        pa.forward(_dev(seq[:, -1:]), pos, 0)
    # recovery: clear the word (the launch counter moves past the failed launch's tags: the granules it left are stale for good)
    pa.clear_engine_status()
    assert pa.engine_status() == 0 and pa.health() == 0
    # the same pass again, complete this time: the launch path's token
    got, _ = pa.forward(_dev(seq[:, -1:]), pos, 0)
    want, _ = la.forward(_dev(seq[:, -1:]), pos, 0)
    assert torch.equal(got.cpu(), want.cpu())


def test_default_follows_model_width_and_cache_length(monkeypatch):
    """sd_model_bind's default (engine.hip carve_workspace, by measurement): two tokens per persistent pass for d_model <= 2048, off for
    wider models; whether a pass takes the persistent launch then follows the caller's bound on the rows' CURRENT length (1280
    positions: one CU walks a head's whole cache, past that the launch path's split-KV attention wins); SPECDEC_PERSIST_MAX_T
    overrides both."""
    from specdec_hip.engine import HipModel

    monkeypatch.delenv("SPECDEC_PERSIST_MAX_T", raising=False)
    mw = W.synthetic_llama(TOY, seed=4, device="cuda", layer_gain=0.05)
    short = HipModel(mw, batch=1, l_max=1280)
    assert short.persist_tokens == 2 and short.persist_active(1)      # default bound = the cache size
    long = HipModel(mw, batch=1, l_max=1600)
    # (round 4) a longer cache keeps the capability; what decides a pass is the caller's bound on the rows' CURRENT length
    # (sd_model_set_length_hint; default: the cache size), so a session sized for a long context starts on the persistent launch
    assert long.persist_tokens == 2 and not long.persist_active(1)
    long.set_length_hint(700)
    assert long.persist_active(1) and long.persist_active(2) and not long.persist_active(3)
    long.set_length_hint(1280)
    assert long.persist_active(1)
    long.set_length_hint(1281)
    assert not long.persist_active(1)
    long.set_length_hint(None)
    assert not long.persist_active(1)
    long.set_length_hint(64)
    long.set_persist_tokens(0)
    assert not long.persist_active(1)
    wide = W.random_init(_shape_3b(1), seed=0, device="cuda")
    assert HipModel(wide, batch=1, l_max=256).persist_tokens == 0
    monkeypatch.setenv("SPECDEC_PERSIST_MAX_T", "1")
    forced = HipModel(mw, batch=1, l_max=4096)     # the measurement override lifts the context bound as well
    assert forced.persist_tokens == 1 and forced.persist_active(1)


def test_two_concurrent_persistent_launches_complete_or_recover(monkeypatch):
    """The persistent launch needs every CU to itself (one workgroup per CU, each declaring the CU's whole LDS, spinning on the
    others' hand-offs; include/specdec_hip.h, sd_model_engine_status_clear). Two of them from two streams of one process may be
    dealt CUs alternately: then neither can complete, both leave through their bounded waits, and the health word says so. The
    contract tested here: whatever the dispatcher does, nothing hangs, and every model either produced the launch path's token
    or reports a non-zero status from which recover() (clear + launch path) yields that token."""
    mw = W.synthetic_llama(TOY, seed=4, device="cpu", layer_gain=0.05)
    pa1, la = _engines(mw, 1, 128, monkeypatch)
    pa2, _ = _engines(mw, 1, 128, monkeypatch)
    seq = synthetic_prompts(1, 8, TOY.vocab, seed=5)
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    for hm in (pa1, pa2, la):
        hm.forward(_dev(seq[:, :-1]), zero, 0, skip_head=True)
    pos = torch.tensor([7], dtype=torch.int32, device="cuda")
    want, _ = la.forward(_dev(seq[:, -1:]), pos, 0)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    tok = _dev(seq[:, -1:])
    outs = {}
    import time
    t0 = time.time()
    for rep in range(6):       # the same pass (same position: idempotent), interleaved on the two streams
        for name, hm, st in (("a", pa1, s1), ("b", pa2, s2)):
            if hm.health():
                continue       # (HipModel.forward refuses to pile work onto an invalid state)
            outs[name], _ = hm.forward(tok, pos, 0, stream=st)
    torch.cuda.synchronize()
    assert time.time() - t0 < 10.0, "bounded waits: two colliding launches cost ~50 ms each, never a hang"
    for name, hm in (("a", pa1), ("b", pa2)):
        st = hm.engine_status()
        assert (st != 0) == (hm.health() != 0)
        if st == 0:
            assert torch.equal(outs[name].cpu(), want.cpu()), name
        else:
            assert hm.recover() == st and hm.engine_status() == 0 and not hm.persist_active(1)
            got, _ = hm.forward(tok, pos, 0)
            assert torch.equal(got.cpu(), want.cpu()), name


def test_c_abi_refuses_cache_rows_the_persistent_attention_cannot_walk(monkeypatch):
    """sd_model_bind takes any Lmax >= 1 (the Python wrapper rounds to 32); the persistent launch's attention reads V^T in 16-byte
    vectors of 8 keys, so a cache whose rows are not a multiple of 8 positions must stay on the launch path rather than issue
    misaligned loads."""
    import ctypes

    from specdec_hip import _abi
    from specdec_hip.engine import HipModel

    monkeypatch.delenv("SPECDEC_PERSIST_MAX_T", raising=False)
    mw = W.synthetic_llama(TOY, seed=4, device="cuda", layer_gain=0.05)
    hm = HipModel(mw, batch=1, l_max=128)
    assert hm.persist_tokens == 2
    lib = hm.lib
    for lmax, want in ((100, 0), (4, 0), (104, 2)):
        nbytes = lib.sd_model_kv_bytes(hm.handle, 1, lmax)
        k = torch.zeros(nbytes // 2 + 8, dtype=torch.bfloat16, device="cuda")
        v = torch.zeros(nbytes // 2 + 8, dtype=torch.bfloat16, device="cuda")
        _abi.check(lib.sd_model_bind(hm.handle, k.data_ptr(), v.data_ptr(), 1, lmax, hm.workspace.data_ptr(), hm.workspace.numel()), "sd_model_bind")
        assert lib.sd_model_persist_tokens(hm.handle) == want, lmax
        hm.k_cache, hm.v_cache, hm.l_max = k, v, lmax    # keep the bound buffers alive with the model
