"""Host-side pieces of the pipeline — product modules AND their oracle restatement — against
outputs of the reference's own functions (tests/golden/hostlogic_golden.json), plus the
known-answer cases of the reference's unit tests."""

import json

import numpy as np
import os

import pytest
import torch

import cases
from oracle import hostlogic_ref as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hostlogic_golden.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_policies_product_and_oracle_match_reference(gold):
    from src.specdec.policies.policies import create_policy

    kw = {"conf_threshold": {"tau": 0.3}, "topk_agree": {"k": 3}, "typical": {"p": 0.2}}
    for row in gold["policies"]:
        dl, bl, d_ids, b_ids = cases.build_policy_case(row["K"], row["V"], row["seed"])
        # oracle
        assert O.longest_prefix_accept(d_ids, b_ids, bl) == row["longest_prefix"][0]
        assert O.longest_prefix_accept(d_ids, b_ids) == row["longest_prefix"][1]
        assert O.conf_threshold_accept(d_ids, dl, 0.3) == row["conf_threshold"][0]
        assert O.topk_agree_accept(d_ids, bl, 3) == row["topk_agree"][0]
        assert O.typical_accept(d_ids, bl, 0.2) == row["typical"][0]
        # product (host tensors: the three threshold policies are plain tensor ops; longest-prefix
        # on logits is the HIP op and is checked in the gpu suite)
        for name in ("conf_threshold", "topk_agree", "typical"):
            a, info = create_policy(name, **kw[name]).accept_tokens(d_ids, b_ids, dl, bl)
            assert a == row[name][0], (name, row)
            assert info["policy"] == name and info["proposed_len"] == row["K"]
            a2, _ = create_policy(name, **kw[name]).accept_tokens(d_ids, b_ids)  # no logits -> id prefix
            assert a2 == row[name][1]
        a, info = create_policy("longest_prefix").accept_tokens(d_ids, b_ids)
        assert a == row["longest_prefix"][1] and info["verify_backend"] == "host-ids"


def test_longest_prefix_known_answers_and_cpu_logits_refused():
    """reference tests/specdec/test_policies.py:20-65."""
    from src.specdec.policies.policies import LongestPrefixPolicy, create_policy

    pol = LongestPrefixPolicy()
    t = lambda *x: torch.tensor([list(x)])
    assert pol.accept_tokens(t(1, 2, 3, 4), t(1, 2, 3, 4))[0] == 4
    assert pol.accept_tokens(t(1, 2, 3, 4), t(1, 2, 5, 6))[0] == 2
    assert pol.accept_tokens(t(1, 2, 3, 4), t(5, 6, 7, 8))[0] == 0
    assert pol.accept_tokens(t(1, 2), t(1, 2, 3, 4))[0] == 2
    with pytest.raises(RuntimeError, match="GPU only"):
        pol.accept_tokens(t(1, 2), t(1, 2), None, torch.zeros(1, 2, 10))
    with pytest.raises(ValueError):
        create_policy("nope")


def test_controllers_product_and_oracle_match_reference(gold):
    from src.specdec.policies.controllers import AdaptiveKController, FixedKController, create_controller

    for row in gold["controllers"]:
        prod = create_controller("adaptive", **row["params"])
        orc = O.AdaptiveKOracle(**row["params"])
        ks_p, ks_o = [], []
        for i, r in enumerate(row["rates"]):
            ctx = {"acceptance_rate": r} if i % 7 else {}
            ks_p.append(prod.get_k(i, dict(ctx)))
            ks_o.append(orc.get_k(ctx))
        assert ks_p == row["ks"] and ks_o == row["ks"]
        assert prod.get_info()["recent_acceptance_rate"] == pytest.approx(row["info_recent"])
        assert orc.recent() == pytest.approx(row["info_recent"])
    # reference tests/specdec/test_controllers.py:25-60
    c = FixedKController(k=5)
    assert all(c.get_k(s, {"acceptance_rate": 0.5}) == 5 for s in range(10))
    assert c.get_info() == {"controller": "fixed_k", "k": 5}
    assert isinstance(create_controller("adaptive"), AdaptiveKController) and create_controller("fixed").k == 4
    a = AdaptiveKController(initial_k=4, min_k=1, max_k=8, target_acceptance_rate=0.7)
    for s in range(8):
        k = a.get_k(s, {"acceptance_rate": 0.95})
    assert k > 4
    with pytest.raises(ValueError):
        create_controller("bogus")


def test_bonus_token_oracle_matches_reference(gold):
    import numpy as np

    for row in gold["bonus"]:
        logits = torch.from_numpy(np.random.default_rng(row["seed"]).standard_normal(row["V"]).astype(np.float32)) * 3
        got = O.bonus_token_greedy(logits, row["temperature"], row["top_p"], row["top_k"] or None, row["V"])
        assert got == row["greedy_token"]


def test_sequence_utils_and_clamp_match_reference(gold):
    from src.specdec.core.sequence_utils import create_position_ids, pad_sequences, unpad_append_repad, unpad_sequences
    from src.specdec.utils.token_validation import get_vocab_size, validate_and_clamp_tokens

    dev = torch.device("cpu")
    for row in gold["sequences"]:
        seqs = [torch.tensor(s, dtype=torch.long) for s in row["seqs"]]
        batch, mask, lens = pad_sequences(seqs, 0, dev)
        assert batch.tolist() == row["batch"] and mask.tolist() == row["mask"] and lens == row["lengths"]
        assert create_position_ids(lens, batch.shape[1], dev).tolist() == row["position_ids"]
        back = unpad_sequences(batch, mask)
        assert [b.tolist() for b in back] == row["seqs"]
        ob, om, ol, op = O.pad_right(row["seqs"], 0)
        assert (ob, om, ol, op) == (row["batch"], row["mask"], row["lengths"], row["position_ids"])
        grown, gmask, glens = unpad_append_repad(seqs, [torch.tensor([7, 8])] * len(seqs), 0, dev)
        assert glens == [n + 2 for n in row["lengths"]] and int(gmask.sum()) == sum(glens)
    e, m, l = pad_sequences([], 0, dev)
    assert e.shape == (0, 0) and l == []
    with pytest.raises(ValueError):
        unpad_append_repad([torch.tensor([1])], [], 0, dev)
    for row in gold["clamp"]:
        ids = torch.tensor(row["ids"])
        assert validate_and_clamp_tokens(ids, row["V"], "t").tolist() == row["out"]
        assert O.clamp_tokens(row["ids"], row["V"]) == row["out"]
    assert validate_and_clamp_tokens(None, 5) is None

    class _M:
        vocab_size = 321

    assert get_vocab_size(_M()) == 321


def test_kv_types_match_reference_behaviour():
    """reference tests/test_kv_cache.py:189-353."""
    from src.specdec.cache.kv_types import KVCache, validate_kv_compatibility

    kv = tuple((torch.randn(1, 2, 5, 4), torch.randn(1, 2, 5, 4)) for _ in range(3))
    c = KVCache.from_hf_output(kv)
    assert (c.seq_len, c.get_num_layers(), c.dtype) == (5, 3, torch.float32)
    s = c.slice_prefix(2)
    assert s.seq_len == 2 and torch.equal(s.past_key_values[1][0], kv[1][0][:, :, :2])
    assert s.get_shapes() == ((1, 2, 2, 4), (1, 2, 2, 4))
    with pytest.raises(ValueError):
        c.slice_prefix(6)
    with pytest.raises(ValueError):
        KVCache.from_hf_output(())
    assert c.to(torch.device("cpu")) is c
    validate_kv_compatibility(c, s)
    with pytest.raises(ValueError, match="Layer count"):
        validate_kv_compatibility(c, KVCache.from_hf_output(kv[:2]))
    with pytest.raises(ValueError, match="Dtype"):
        validate_kv_compatibility(c, KVCache.from_hf_output(tuple((k.half(), v.half()) for k, v in kv)))
    with pytest.raises(ValueError, match="Shape"):
        validate_kv_compatibility(c, KVCache.from_hf_output(tuple((torch.randn(1, 3, 5, 4),) * 2 for _ in range(3))))


def test_deterministic_mode(monkeypatch):
    """reference tests/test_deterministic_mode.py:25-80."""
    from src.specdec.utils.deterministic import ensure_deterministic, set_deterministic_mode

    set_deterministic_mode()
    a = torch.rand(3)
    set_deterministic_mode(1234)
    assert torch.equal(a, torch.rand(3))
    monkeypatch.setenv("SPECDEC_DETERMINISTIC", "1")
    assert ensure_deterministic(7) is True
    b = torch.rand(2)
    ensure_deterministic(7)
    assert torch.equal(b, torch.rand(2))
    monkeypatch.setenv("SPECDEC_DETERMINISTIC", "0")
    assert ensure_deterministic() is False


def test_rejection_sampling_policy_matches_restatement_and_preserves_the_target_distribution():
    """The opt-in `rejection` policy (speculative sampling; not one of the reference's policies): (a) product == the numpy
    restatement on seeded logits and uniforms — accepted length and the distribution of the next token; (b) the defining
    property: one draft token drawn from q, accepted / corrected by the policy, is distributed as the target p."""
    from scipy.stats import chisquare

    from oracle.hostlogic_ref import rejection_accept
    from src.specdec.policies.policies import RejectionSamplingPolicy, create_policy

    rng = np.random.default_rng(3)
    for case in range(20):
        K, V = int(rng.integers(1, 6)), int(rng.integers(5, 60))
        dl = torch.from_numpy(rng.standard_normal((1, K, V)).astype(np.float32)) * 2
        bl = torch.from_numpy(rng.standard_normal((1, K + 1, V)).astype(np.float32)) * 2
        bl[0, :K] += dl[0] * float(rng.uniform(0, 2))          # correlated target: some accepts, some rejects
        temp = float(rng.choice([1.0, 0.7, 1.5]))
        pol = create_policy("rejection", temperature=temp, seed=case)
        q = pol.distributions(dl[0])
        d_ids = torch.tensor([[RejectionSamplingPolicy.draw(q[i], rng.random()) for i in range(K)]])
        u = rng.random(K)
        a, info = pol.accept_tokens(d_ids, d_ids, dl, bl, uniforms=u)
        wa, wnext = rejection_accept(d_ids[0].numpy(), dl[0].numpy(), bl[0].numpy(), u, temp)
        assert a == wa and info["policy"] == "rejection"
        assert np.allclose(info["next_distribution"].numpy(), wnext, rtol=1e-12, atol=1e-15)
    # (b) V = 6, K = 1: emitted token ~ p
    V, n = 6, 20000
    dl = torch.tensor([[[1.0, 0.2, -0.5, 2.0, 0.0, -1.0]]])
    bl = torch.tensor([[[0.0, 1.5, 0.3, 0.5, -2.0, 1.0], [0.0] * 6]])
    pol = create_policy("rejection", temperature=1.0, seed=9)
    q, p = pol.distributions(dl[0, 0]), pol.distributions(bl[0, 0])
    counts = np.zeros(V)
    uu = np.random.default_rng(10).random((n, 3))
    for i in range(n):
        d = RejectionSamplingPolicy.draw(q, uu[i, 0])
        a, info = pol.accept_tokens(torch.tensor([[d]]), torch.tensor([[d]]), dl, bl, uniforms=[uu[i, 1]])
        tok = d if a == 1 else RejectionSamplingPolicy.draw(info["next_distribution"], uu[i, 2])
        counts[tok] += 1
    assert chisquare(counts, p.numpy() * n).pvalue > 1e-4
