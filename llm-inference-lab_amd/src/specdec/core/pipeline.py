"""`SpeculativePipeline` — the draft-then-verify loop on MI355X.

Public surface of the reference (src/specdec/core/pipeline.py): the constructor
arguments (:198-218), `generate(prompt, max_tokens, temperature, do_sample)` (:893) and
`generate_batch(prompts, ...)` (:1605), and the result-dict keys (:1350-1380, :3876-3905).

What runs underneath is different. The reference makes 2K HF forwards per step over the
whole prefix from Python, synchronising the host several times per row. Here one call
per step (`HipSpecDec.step`) replays a hipGraph that holds the K draft forwards, ONE
K+1-token verify forward of the target over its appended KV cache, the accept scan and
the in-place state advance; the host reads one small pinned record per step and applies
the reference's host-side rules to it (EOS cut, bonus, de-duplication, budget — restated
in `_rules_batch` / `_rules_single`). When those rules leave a row where the device
assumed (the common case) nothing else happens; when they do not (a de-duplication
dropped or rewound tokens), the row's caches are rebuilt from its sequence.

`generate_batch(do_sample=True)` samples what the reference samples there — only the token after
the accepted prefix (`sample_bonus_token_from_logits`, :3156/:3231/:3355; drafting and verification
stay greedy, :2400/:2645) — inside the same captured step (csrc/sample.hip). `generate(do_sample=True)`
is another rule (:1019-1027, :1217-1224): the draft's proposals and the base token of a zero-accept step are
drawn by transformers' sampling on torch's global CPU generator, verification is greedy; it runs on the
host-policy loop and, seeded alike, makes the reference's draws (models/hip_lm.py hf_sampling_probs).
"""

from __future__ import annotations

import logging
import os
import time
from typing import Any, Dict, List, Optional, Sequence, Union

import psutil
import torch
import yaml

from specdec_hip.engine import EngineGaveUp, HipModel, HipSpecDec

from ..models.hip_lm import HipLM, create_hip_lm, create_hip_pair, is_named_pair
from ..policies.controllers import create_controller
from ..policies.policies import create_policy
from ..utils.deterministic import ensure_deterministic, set_deterministic_mode
from ..utils.interfaces import LanguageModel

PromptLike = Union[str, Sequence[int], torch.Tensor]

_DEFAULTS: Dict[str, Any] = {
    "base_model": "synthetic:llama-3.2-3b",
    "draft_model": "synthetic:llama-3.2-1b",
    "max_draft": 4,
    "implementation": "hip",
    "temperature": 0.7,
    "do_sample": False,
    "max_new_tokens": 64,
    "top_p": 0.9,
    "top_k": 50,
    "repetition_penalty": 1.0,
    "device": "cuda",
    "seed": 1234,
    "log_level": "INFO",
    "verbose": False,
    "draft_mode": "vanilla",
    "medusa": {"enabled": False, "num_heads": 2, "head_init": "tie", "temperature": 0.7, "top_p": 1.0},
    "eagle": {"enabled": False, "alpha": 0.7, "max_draft": 2},
}


def _clamp(tok: int, vocab: int) -> int:
    return 0 if tok < 0 else (vocab - 1 if tok >= vocab else int(tok))


class _Row:
    __slots__ = ("seq", "generated", "active", "proposed", "accepted", "draws", "steps", "strict_acc", "strict_prop", "k_trace", "counters")

    def __init__(self, seq: List[int]):
        self.seq, self.generated, self.active = seq, [], True
        self.proposed = self.accepted = self.draws = self.steps = 0
        self.counters: List[tuple] = []            # after each of the row's own steps: (proposed, accepted, tokens generated, accept length)
        self.strict_acc = self.strict_prop = 0     # per-row adaptive K: what the row's controller is fed with
        self.k_trace: List[int] = []               # ... and the k that counted in each of its steps


class SpeculativePipeline:
    def __init__(self, base_lm: Optional[LanguageModel] = None, draft_lm: Optional[LanguageModel] = None,
                 config_path: Optional[str] = None, base_model: Optional[Any] = None,
                 draft_model: Optional[Any] = None, max_draft: Optional[int] = None, device: str = "auto",
                 seed: Optional[int] = None, implementation: Optional[str] = None,
                 force_device: Optional[str] = None, policy: str = "longest_prefix",
                 policy_params: Optional[Dict[str, Any]] = None, controller: str = "fixed",
                 controller_params: Optional[Dict[str, Any]] = None, draft_mode: str = "vanilla",
                 enable_optimization: bool = True, enable_profiling: bool = False,
                 profile_dir: Optional[str] = None, medusa_heads: Optional[Any] = None):
        self.logger = logging.getLogger(__name__)
        self.config = self._load_config(config_path)
        for key, val in (("base_model", base_model), ("draft_model", draft_model), ("max_draft", max_draft),
                         ("seed", seed), ("implementation", implementation), ("draft_mode", draft_mode)):
            if val is not None:
                self.config[key] = val
        impl = self.config["implementation"]
        if impl not in ("hip", "hf", "fake"):
            # CPU / MPS implementations of the reference have no counterpart here: the product is the GPU path
            raise ValueError(f"implementation={impl!r} is not available in this build (use 'hip')")
        # "fake": the reference's weight-less test double (models/fake_lm.py; what the reference's own configs/specdec.yaml
        # selects): token ids are a function of the input ids, logits are noise on the device; drives the step loop, the
        # policies and the registry ops without a model (generate / generate_batch through the host-policy loop)
        self._fake = impl == "fake"
        mode = self.config.get("draft_mode", "vanilla")
        # persistent heads (specdec_hip.weights.MedusaHeads): not in the reference — K trained/tied heads evaluated
        # in K GEMVs over the target's last hidden state replace the draft forwards
        self.medusa_heads = medusa_heads
        if medusa_heads is not None and mode != "medusa":
            raise ValueError("medusa_heads needs draft_mode='medusa'")
        if mode == "medusa" and medusa_heads is not None:
            pass
        elif mode == "medusa":
            # `head_init: tie` / `copy`: Medusa-lite as the reference's draftor defines it (modes/medusa.py): heads tied to /
            # copied from the lm_head, head 0 evaluated on the same last hidden state for all K proposals -> K copies of the
            # target's own next token.
            # `head_init: random`: what the reference PIPELINE runs for every Medusa configuration (_run_medusa_hf,
            # pipeline.py:655-763: fresh nn.Linear heads with normal_(0, 0.02) weights and multinomial draws from the global
            # torch generator on every step): _draft_medusa_random below, generate() only.
            if self.config.get("medusa", {}).get("head_init", "tie") not in ("tie", "copy", "random"):
                raise ValueError("medusa.head_init must be 'tie', 'copy' or 'random'")
        elif mode == "eagle":
            # EAGLE-lite as the reference's HF path defines it (_run_eagle_hf, pipeline.py:765-889): draft tokens are the
            # argmax of the lm_head over hidden states extrapolated from the target's last two states, min(k, max_draft)
            # of them per step; generate() only, no draft model (sd_specdec_set_eagle).
            pass
        elif mode != "vanilla":
            raise NotImplementedError(f"draft_mode={mode!r}: 'vanilla', 'medusa' and 'eagle' drafting are on the HIP path")
        if not torch.cuda.is_available():
            raise RuntimeError("SpeculativePipeline needs a GPU (PyTorch-ROCm device 'cuda'); there is no CPU path")
        self.device = "cuda"
        self.dtype = torch.bfloat16
        self.amp_enabled = False
        self.enable_cuda_graph = True  # hipGraph replay of the step (the reference pins this to False)
        if self.config.get("seed") is not None:
            set_deterministic_mode(int(self.config["seed"]))
        ensure_deterministic(int(self.config.get("seed") or 1234))

        if self._fake:
            from ..models.fake_lm import create_fake_lm

            if mode != "vanilla" or medusa_heads is not None:
                raise NotImplementedError("implementation='fake' drafts with the fake draft model (draft_mode='vanilla')")
            sd = self.config.get("seed")
            self.base_lm = base_lm if base_lm is not None else create_fake_lm(f"fake-base-{self.config['base_model']}", seed=sd)
            self.draft_lm = draft_lm if draft_lm is not None else create_fake_lm(f"fake-draft-{self.config['draft_model']}", seed=sd)
        else:
            self.base_lm = base_lm if base_lm is not None else None
        no_draft = mode in ("medusa", "eagle") and draft_lm is None and self.config.get("draft_model") in (None, "", "none", "NONE")
        if not self._fake:
            if base_lm is None and draft_lm is None and not no_draft and is_named_pair(self.config["base_model"], self.config["draft_model"]):
                self.base_lm, self.draft_lm = create_hip_pair(self.config["base_model"], self.config["draft_model"])
            else:
                self.base_lm = base_lm if base_lm is not None else create_hip_lm(self.config["base_model"])
                self.draft_lm = draft_lm if (draft_lm is not None or no_draft) else create_hip_lm(self.config["draft_model"])
        for who, lm in (("base", self.base_lm), ("draft", self.draft_lm)):
            if self._fake:
                break
            if lm is None and who == "draft":
                continue                         # Medusa-lite drafts from the target itself (generate() only)
            if not isinstance(lm, HipLM):
                raise TypeError(f"{who}_lm must be a HipLM (got {type(lm).__name__}): the step loop drives the "
                                "models through the C-ABI engine, not through generate_tokens")
        if self.draft_lm is not None and self.base_lm.vocab_size != self.draft_lm.vocab_size:
            raise ValueError(f"draft/base vocabularies differ: {self.draft_lm.vocab_size} vs {self.base_lm.vocab_size}")
        self.speculative_enabled = True
        self.policy = create_policy(policy, **(policy_params or {}))
        # longest_prefix (exact match) is the policy inside the captured device step. The logit-threshold policies
        # (typical, topk_agree, conf_threshold) take the reference's own route (_decode_host_policy): they compare the draft with
        # what the base model generates by itself from the same prefix, which a parallel pass over the DRAFT tokens does
        # not give once a non-argmax draft token is accepted.
        self.policy_name = policy
        if policy != "longest_prefix" and (mode != "vanilla" or medusa_heads is not None):
            raise NotImplementedError(f"policy={policy!r} drafts with the draft model (draft_mode='vanilla')")
        self.controller = create_controller(controller, **(controller_params or {}))
        from src.scheduler import create_speculative_scheduler

        self.scheduler = create_speculative_scheduler(device="cuda")
        self.metrics: Dict[str, Any] = {}
        self._runtimes: Dict[Any, Any] = {}

    # ------------------------------------------------------------------ configuration
    def _load_config(self, config_path: Optional[str]) -> Dict[str, Any]:
        """Flat YAML merged over the defaults (pipeline.py:398-438)."""
        cfg = dict(_DEFAULTS)
        if config_path:
            with open(config_path) as f:
                loaded = yaml.safe_load(f) or {}
            if not isinstance(loaded, dict):
                raise ValueError(f"{config_path}: expected a flat mapping")
            cfg.update(loaded)
        return cfg

    # ------------------------------------------------------------------ runtime pieces
    def _runtime(self, batch: int, need_len: int, k: int, emit_mode: int, self_draft: bool = False, adaptive: bool = False):
        """Engine instances (own KV caches over the shared weights) + the step loop object (`adaptive`: the loop whose
        rows carry their own K, a different captured step from the fixed-K loop of the same shape)."""
        key = (batch, emit_mode, self_draft)
        rt = self._runtimes.get(key)
        if rt is None or rt["l_max"] < need_len:
            l_max = (max(need_len, 256) + 63) // 64 * 64
            rt = {"l_max": l_max, "target": self.base_lm.new_engine(batch, l_max),
                  "draft": None if self_draft else self.draft_lm.new_engine(batch, l_max), "loops": {}}
            self._runtimes[key] = rt
        lkey = ("adaptive", k) if adaptive else k
        loop = rt["loops"].get(lkey)
        if loop is None:
            loop = rt["loops"][lkey] = HipSpecDec(rt["draft"], rt["target"], batch, k, emit_mode)
            if self_draft and self.medusa_heads is not None:
                if self.medusa_heads.n_heads != k:
                    raise ValueError(f"{self.medusa_heads.n_heads} medusa heads but K={k}")
                loop.set_medusa(self.medusa_heads.weights, self.base_lm.weight_dtype)
            elif self_draft and self._eagle():
                # one state workspace per runtime: loops of different K continue the same extrapolation state
                rt["eagle_ws"] = loop.set_eagle(float(self.config.get("eagle", {}).get("alpha", 0.7)), rt.get("eagle_ws"))
        return rt, loop

    def _eagle(self) -> bool:
        return self.config.get("draft_mode") == "eagle"

    def _effective_k(self, k: int, self_draft: bool) -> int:
        """EAGLE-lite proposes min(k, eagle.max_draft) tokens per step (pipeline.py:818 of the reference)."""
        if self_draft and self._eagle():
            return max(1, min(int(k), int(self.config.get("eagle", {}).get("max_draft", 2))))
        return int(k)

    def _encode(self, prompt: PromptLike) -> List[int]:
        if isinstance(prompt, str):
            return [int(x) for x in self.base_lm.encode(prompt).flatten().tolist()]
        if isinstance(prompt, torch.Tensor):
            return [int(x) for x in prompt.flatten().tolist()]
        return [int(x) for x in prompt]

    def _prefill_row(self, rt, b: int, seq: List[int]) -> None:
        """KV of seq[:-1] into row b of both caches (the step recomputes the last two)."""
        if len(seq) < 2:
            return
        toks = torch.tensor([seq[:-1]], dtype=torch.int32, device="cuda")
        zero = torch.zeros(1, dtype=torch.int32, device="cuda")
        rt["target"].forward(toks, zero, 0, skip_head=True, row0=b)
        if rt["draft"] is not None:
            rt["draft"].forward(toks, zero, 0, skip_head=True, row0=b)

    def _prefill(self, rt, rows: List[_Row]) -> None:
        lens = {len(r.seq) for r in rows}
        if len(lens) == 1 and len(rows[0].seq) >= 2:
            toks = torch.tensor([r.seq[:-1] for r in rows], dtype=torch.int32, device="cuda")
            zero = torch.zeros(len(rows), dtype=torch.int32, device="cuda")
            rt["target"].forward(toks, zero, 0, skip_head=True)
            if rt["draft"] is not None:
                rt["draft"].forward(toks, zero, 0, skip_head=True)
        else:
            for b, r in enumerate(rows):
                self._prefill_row(rt, b, r.seq)

    @staticmethod
    def _set_row(loop: HipSpecDec, b: int, row: _Row) -> None:
        seq = row.seq
        loop.set_row(b, len(seq), seq[-2] if len(seq) >= 2 else 0, seq[-1], row.active)

    # ------------------------------------------------------------------ host-side rules
    def _rules_batch(self, row: _Row, k: int, a: int, t: List[int], max_tokens: int, eos: Optional[int],
                     resample=None) -> None:
        """One row of one generate_batch step (pipeline.py:3018-3590), from the step record:
        a = accepted length, t[i] = target argmax after the row's sequence + d_1..d_i (sampling mode:
        t[a] is the token the device sampled at position a; `resample(pos)` draws at another position
        with the same Philox draw — only an EOS-cut full acceptance needs it)."""
        V = self.base_lm.vocab_size
        gen = row.generated
        if a > 0:
            acc = [_clamp(x, V) for x in t[:a]]
            if eos is not None and eos in acc:           # accepted EOS: cut there, row stops (:3120-3131)
                acc = acc[: acc.index(eos)]
                row.active = False
            pos = a if a < k else len(acc)                # (:3140-3231) the extra forward runs over seq + cut acc
            bonus = _clamp(t[pos] if (resample is None or pos == a) else resample(pos), V)
            if eos is not None and bonus == eos:         # bonus EOS stays in the output (:3274-3280)
                row.active = False
            acc.append(bonus)
            if gen:                                      # overlap with the generated tail (:3295-3318)
                for c in range(min(5, len(gen), len(acc)), 0, -1):
                    if gen[-c:] == acc[:c]:
                        acc = acc[c:]
                        break
            gen.extend(acc)
            emitted = acc
        else:
            emitted = [_clamp(t[0], V)]
            if eos is not None and emitted[0] == eos:    # (:3360-3365)
                row.active = False
                emitted = []
            if emitted and gen and gen[-1] == emitted[0]:  # (:3367-3376)
                emitted = []
            gen.extend(emitted)
        row.proposed += k
        row.accepted += len(emitted)
        if emitted:                                      # sequence update with its own overlap rule (:3470-3572)
            acc, cur = list(emitted), row.seq
            c = min(5, len(cur), len(acc))
            if c > 0 and cur[-c:] == acc[:c]:
                if len(acc) > c:
                    acc = acc[c:]
                    if len(gen) >= c:
                        del gen[-c:]
                else:
                    acc = []
            row.seq = cur + acc
        if len(gen) >= max_tokens:                       # (:3589-3590)
            row.active = False

    @staticmethod
    def _rules_single(row: _Row, k: int, a: int, d: List[int], t: List[int], max_tokens: int,
                      eos: Optional[int]) -> None:
        """One step of generate() (pipeline.py:1190-1272): draft ids cut to the budget, or one
        base token when nothing was accepted."""
        remaining = max_tokens - len(row.generated)
        new = [int(x) for x in d[: min(a, max(remaining, 0))]] if a > 0 else ([int(t[0])] if remaining > 0 else [])
        row.proposed += k
        row.accepted += a
        row.generated.extend(new)
        row.seq = row.seq + new
        if len(row.generated) >= max_tokens:
            row.active = False
        elif eos is not None and row.generated and row.generated[-1] == eos:
            row.active = False

    # ------------------------------------------------------------------ the loop
    def start_session(self, prompts: List[List[int]], max_tokens: int, emit_mode: int,
                      sampling: Optional[Dict[str, Any]] = None, step_limit: Optional[int] = None,
                      self_draft: bool = False) -> "DecodeSession":
        """Prefill + device state for a batch of rows; `advance()` then runs one step at a time
        (generate / generate_batch drive it to completion, bench.py times exact step counts)."""
        return DecodeSession(self, prompts, max_tokens, emit_mode, sampling, step_limit, self_draft)

    def _decode(self, prompts: List[List[int]], max_tokens: int, emit_mode: int, step_limit: int,
                sampling: Optional[Dict[str, Any]] = None, self_draft: bool = False):
        t_start = time.time()
        sess = self.start_session(prompts, max_tokens, emit_mode, sampling, step_limit, self_draft)
        while sess.any_active():    # every row stops after `step_limit` steps of its own
            if not sess.advance():
                break
        sess.finish()
        torch.cuda.synchronize()
        st = sess.stats
        st["total_ms"] = (time.time() - t_start) * 1e3
        st["k"] = sess.k
        return sess.rows, st

    def _medusa_random(self) -> bool:
        return (self.config.get("draft_mode") == "medusa" and self.medusa_heads is None
                and self.config.get("medusa", {}).get("head_init", "tie") == "random")

    def _draft_medusa_random(self, seq: List[int], k: int, temperature: float, row: int, rows: int):
        """_run_medusa_hf (pipeline.py:655-763) on the HIP engine: the target's last hidden state (after the final norm) of
        the last token; `num_heads` FRESH heads per call — torch.nn.Linear's default init, then normal_(0, 0.02), both from
        the global torch generator on the host, exactly as the reference creates them; every head still in range draws one
        token per proposal with torch.multinomial from softmax(logits / T) (global generator, host); the proposal of a step
        is head 0's draw. The hidden state does not move between proposals, so every head's logits are computed once
        (a [num_heads x V x d] product on the device, bf16 as the engine holds weights). Returns (ids [1,k], logits [1,k,V])."""
        h = self.base_lm.last_hidden_state(torch.tensor([seq], dtype=torch.long), row=row, rows=rows)   # [1,1,d] fp32, bf16 values
        num_heads = int(self.config.get("medusa", {}).get("num_heads", 2))
        d, V = h.shape[-1], self.base_lm.vocab_size
        heads = []
        for _ in range(num_heads):
            head = torch.nn.Linear(d, V, bias=False)
            torch.nn.init.normal_(head.weight, 0, 0.02)
            heads.append(head.weight.detach().to(torch.bfloat16))
        hw = torch.stack(heads).to("cuda").float()                                    # [H,V,d], bf16 values
        logits = torch.matmul(hw, h.view(-1)).to(torch.bfloat16).float().cpu()         # [H,V] bf16-rounded, on the host for the draws
        ids = []
        for step in range(k):
            toks = []
            for hi in range(min(num_heads, k - step)):
                lg = logits[hi].view(1, 1, -1)
                if temperature > 0:
                    lg = lg / temperature
                toks.append(torch.multinomial(torch.softmax(lg, dim=-1).squeeze(1), 1))
            ids.append(toks[0])
        d_ids = torch.cat(ids, dim=1).to("cuda")
        d_logits = logits[0].view(1, 1, -1).expand(1, k, -1).contiguous().to("cuda")
        return d_ids, d_logits

    def _decode_host_policy(self, prompts: List[List[int]], max_tokens: int, emit_mode: int, step_limit: int,
                            temperature: float = 0.7, sample_t: Optional[float] = None, sample_kwargs: Optional[Dict[str, Any]] = None):
        """The reference's verification for the logit-threshold policies (speculative_scheduler.py:294-368, policies.py:213-396;
        pipeline.py:1019-1100 and :2397-3030): per row and step, K greedy draft tokens with their logits, the base model's
        OWN K greedy tokens with their logits from the same prefix (K one-token HIP forwards each, over the rows' cached
        prefixes), the policy on the device logits, then the same host rules as the device step. A full acceptance takes
        the extra base forward the reference takes (:3199-3206).
        sample_t (generate(do_sample=True), pipeline.py:1019-1027): the draft's tokens are DRAWN at that temperature (a draft that
        draws EOS is shorter, and `proposed` counts what came back, :1122); verification stays greedy; the one base token of a
        zero-accept step is drawn too (:1217-1224) — the step's last draw."""
        skw = dict(sample_kwargs or {})
        t_start = time.time()
        rows = [_Row(list(p)) for p in prompts]
        for r in rows:
            if not r.seq:
                raise ValueError("empty prompt")
        eos = self.base_lm.get_tokenizer_info().get("eos_token_id")
        n = len(rows)
        medusa = self._medusa_random()
        for lm in (self.base_lm, self.draft_lm):
            if lm is not None:
                lm.clear_kv_cache()
        stats = {"steps": 0, "resyncs": 0, "proposed": 0, "accepted": 0, "device_ms": 0.0, "void_row_steps": 0}
        step = 0
        while any(r.active for r in rows):
            step += 1
            ctx = {"step": step, "generated_tokens": max(len(r.generated) for r in rows),
                   "acceptance_rate": stats["accepted"] / max(stats["proposed"], 1)}
            k = int(self.controller.get_k(step, ctx))
            if k <= 0:
                break
            t0 = time.time()
            for b, r in enumerate(rows):
                if not r.active:
                    continue
                ids = torch.tensor([r.seq], dtype=torch.long)
                kb = k   # proposals of this row in this step
                if medusa:
                    d_ids, d_logits = self._draft_medusa_random(r.seq, kb, temperature, b, n)
                elif sample_t is not None:
                    d_ids, d_logits = self.draft_lm.generate_tokens(ids, kb, temperature=sample_t, do_sample=True, row=b, rows=n, **skw)
                    kb = int(d_ids.shape[1])
                else:
                    d_ids, d_logits = self.draft_lm.generate_tokens(ids, kb, do_sample=False, row=b, rows=n)
                b_ids, b_logits = self.base_lm.generate_tokens(ids, kb, do_sample=False, row=b, rows=n)
                a, _info = self.policy.accept_tokens(d_ids, b_ids, d_logits, b_logits)
                d, t = d_ids[0].tolist(), b_ids[0].tolist()
                if sample_t is not None and a == 0:
                    fb, _ = self.base_lm.generate_tokens(ids, 1, temperature=sample_t, do_sample=True, row=b, rows=n, **skw)
                    t[0] = int(fb[0, 0])
                if a == kb and emit_mode == HipSpecDec.EMIT_BONUS:
                    more, _ = self.base_lm.generate_tokens(torch.tensor([r.seq + t], dtype=torch.long), 1, do_sample=False, row=b, rows=n)
                    t.append(int(more[0, 0]))
                else:
                    t.append(-1)
                acc0 = r.accepted
                if emit_mode == HipSpecDec.EMIT_BONUS:
                    self._rules_batch(r, kb, a, t, max_tokens, eos)
                else:
                    self._rules_single(r, kb, a, d, t, max_tokens, eos)
                r.steps += 1
                stats["proposed"] += kb
                stats["accepted"] += r.accepted - acc0
                if r.active and r.steps >= step_limit:
                    r.active = False
            stats["device_ms"] += (time.time() - t0) * 1e3
        torch.cuda.synchronize()
        stats["steps"] = max((r.steps for r in rows), default=0)
        stats["total_ms"] = (time.time() - t_start) * 1e3
        stats["k"] = int(getattr(self.controller, "k", 0) or 0)
        return rows, stats

    def _decode_rejection(self, prompts: List[List[int]], max_tokens: int, step_limit: int):
        """policy="rejection" (opt-in speculative sampling, policies.RejectionSamplingPolicy; generate_batch): per row and
        step K draft tokens DRAWN from the draft's distribution (one HIP forward each over the row's cached prefix), ONE
        K+1-token parallel verify pass of the target over them, the accept test on the device logits, the correction /
        bonus token drawn from the distribution the policy returns. Every uniform comes from the policy's generator."""
        t_start = time.time()
        pol = self.policy
        pol.reseed()
        rows = [_Row(list(p)) for p in prompts]
        eos = self.base_lm.get_tokenizer_info().get("eos_token_id")
        n = len(rows)
        for lm in (self.base_lm, self.draft_lm):
            lm.clear_kv_cache()
        stats = {"steps": 0, "resyncs": 0, "proposed": 0, "accepted": 0, "accepted_draft": 0, "device_ms": 0.0, "void_row_steps": 0}
        step = 0
        while any(r.active for r in rows):
            step += 1
            # the controller sees the STRICT rate (accepted draft tokens / proposed, <= 1): the reported `accepted` counts the
            # bonus / correction token as generate_batch does and would read (k+1)/k on a fully accepted step
            ctx = {"step": step, "generated_tokens": max(len(r.generated) for r in rows),
                   "acceptance_rate": stats["accepted_draft"] / max(stats["proposed"], 1)}
            k = int(self.controller.get_k(step, ctx))
            if k <= 0:
                break
            for b, r in enumerate(rows):
                if not r.active:
                    continue
                drafted, d_logits = [], []
                for _ in range(k):
                    _, lg = self.draft_lm.generate_tokens(torch.tensor([r.seq + drafted], dtype=torch.long), 1, do_sample=False, row=b, rows=n)
                    d_logits.append(lg[:, 0])
                    drafted.append(pol.draw(pol.distributions(lg[0, 0]), float(pol.uniforms(1)[0])))
                d_ids = torch.tensor([drafted], dtype=torch.long, device="cuda")
                _, b_logits = self.base_lm.verify_tokens(torch.tensor([r.seq], dtype=torch.long), d_ids, row=b, rows=n)
                a, info = pol.accept_tokens(d_ids, d_ids, torch.stack(d_logits, dim=1), b_logits)
                nxt = pol.draw(info["next_distribution"], float(pol.uniforms(1)[0]))
                emitted = drafted[:a] + [nxt]
                if eos is not None and eos in emitted:
                    emitted = emitted[: emitted.index(eos) + 1]
                    r.active = False
                emitted = emitted[: max(max_tokens - len(r.generated), 0)]   # this path is the product's own: no overshoot of the budget
                r.seq = r.seq + emitted
                r.generated.extend(emitted)
                r.proposed += k
                r.accepted += a + 1                        # the bonus / correction token counted, as generate_batch counts it
                r.steps += 1
                stats["proposed"] += k
                stats["accepted"] += a + 1
                stats["accepted_draft"] += a
                if len(r.generated) >= max_tokens or r.steps >= step_limit:
                    r.active = False
        torch.cuda.synchronize()
        stats["steps"] = max((r.steps for r in rows), default=0)
        stats["total_ms"] = (time.time() - t_start) * 1e3
        stats["k"] = int(getattr(self.controller, "k", 0) or 0)
        return rows, stats

    # ------------------------------------------------------------------ public API
    def _sampling_config(self, do_sample: bool, temperature: float, kwargs: Dict[str, Any]) -> Optional[Dict[str, Any]]:
        """Sampler parameters of a do_sample=True run (kwargs over config, pipeline.py:3148-3153)."""
        if not do_sample:
            return None
        top_p = kwargs.get("top_p", self.config.get("top_p", None))
        top_k = kwargs.get("top_k", self.config.get("top_k", None))
        if top_k and min(int(top_k), self.base_lm.vocab_size) > 1024:
            raise NotImplementedError(f"do_sample=True: top_k={top_k} > 1024 is not on the HIP path")
        return {"temperature": float(temperature), "top_k": int(top_k) if top_k else None,
                "top_p": None if top_p is None else float(top_p), "seed": int(kwargs.get("seed", self.config.get("seed") or 0))}

    def generate(self, prompt: PromptLike, max_tokens: Optional[int] = None, temperature: Optional[float] = None,
                 do_sample: Optional[bool] = None, **kwargs) -> Dict[str, Any]:
        """Single-prompt speculative decoding (reference :893-1413): accepted tokens are the
        draft's, no bonus token, a zero-accept step emits one base token."""
        t_begin = time.time()
        max_tokens = max_tokens or self.config["max_new_tokens"]
        temperature = temperature or self.config["temperature"]
        do_sample = do_sample if do_sample is not None else self.config["do_sample"]
        sample_t: Optional[float] = None
        if do_sample and not self._fake:   # (the fake double ignores sampling parameters, in the reference as here)
            # pipeline.py:1019-1027 / :1217-1224: the DRAFT's proposals, and the one base token of a zero-accept step, are drawn at the
            # call's temperature (transformers' sampling on torch's global generator, models/hip_lm.py hf_sampling_probs);
            # every verification path of the scheduler is greedy (speculative_scheduler.py:192-199, :304-310, :339-345)
            if self.config.get("draft_mode", "vanilla") in ("medusa", "eagle") or self.policy_name != "longest_prefix":
                raise NotImplementedError("generate(do_sample=True) is restated for the draft-model mode with the longest_prefix policy")
            sample_t = float(temperature)
        ids = self._encode(prompt)
        # draft modes are a generate() feature in the reference (pipeline.py:1016-1041); generate_batch always
        # drafts with the draft model
        if self.policy_name == "rejection":
            raise NotImplementedError("policy='rejection' is a generate_batch policy (it emits a correction / bonus token every step)")
        if self.policy_name != "longest_prefix" or self._medusa_random() or self._fake or sample_t is not None:
            rows, st = self._decode_host_policy([ids], max_tokens, HipSpecDec.EMIT_DRAFT, step_limit=2 * max_tokens,
                                                temperature=float(temperature), sample_t=sample_t,
                                                sample_kwargs={k: kwargs[k] for k in ("top_k", "top_p") if k in kwargs})
        else:
            rows, st = self._decode([ids], max_tokens, HipSpecDec.EMIT_DRAFT, step_limit=2 * max_tokens,
                                    self_draft=self.config.get("draft_mode") in ("medusa", "eagle"))
        r = rows[0]
        total_ms = (time.time() - t_begin) * 1e3
        self.metrics = {"total_proposed": r.proposed, "total_accepted": r.accepted, "total_steps": st["steps"],
                        "total_verification_time_ms": st["device_ms"], "total_generation_time_ms": total_ms,
                        "kv_appended_tokens_total": len(r.generated), "kv_append_time_ms": 0.0}
        text = self.base_lm.decode(r.generated) if r.generated else ""
        n = len(r.generated)
        return {
            "text": text, "generated_tokens": list(r.generated), "latency_ms": total_ms,
            "proposed": r.proposed, "accepted": r.accepted,
            "acceptance_rate": r.accepted / r.proposed if r.proposed > 0 else 0.0,
            "tokens_per_sec": n / (total_ms / 1e3) if total_ms > 0 else 0.0, "steps": st["steps"],
            "verification_time_ms": st["device_ms"], "generation_time_ms": total_ms,
            "kv_appended_tokens_total": n, "kv_append_time_ms": 0.0, "kv_append_enabled": True,
            "kv_append_backend": "hip", **self._sysinfo(),
        }

    def generate_batch(self, prompts: Sequence[PromptLike], max_tokens: Optional[int] = None,
                       temperature: Optional[float] = None, do_sample: Optional[bool] = None,
                       **kwargs) -> List[Dict[str, Any]]:
        """Batched speculative decoding (reference :1605-3931): base tokens + bonus token per
        step, step-count bound, no truncation to max_tokens. Rows are independent sequences."""
        if not prompts:
            return []
        if self._fake:
            # the double has no tokenizer to pad a batch with: the reference falls back to generate() per prompt
            # ("No tokenizer access, falling back to sequential processing", pipeline.py:1680-1700)
            return [self.generate(p, max_tokens=max_tokens, temperature=temperature, do_sample=do_sample, **kwargs) for p in prompts]
        max_tokens = max_tokens or self.config["max_new_tokens"]
        temperature = temperature or self.config["temperature"]
        do_sample = do_sample if do_sample is not None else self.config["do_sample"]
        sampling = self._sampling_config(do_sample, temperature, kwargs)
        heads = self.medusa_heads is not None    # persistent heads also serve generate_batch (opt-in, not in the reference)
        if self.draft_lm is None and not heads:
            raise ValueError("generate_batch drafts with the draft model (the reference ignores draft_mode there): pass draft_lm / draft_model")
        ids = [self._encode(p) for p in prompts]
        if self.policy_name == "rejection":
            rows, st = self._decode_rejection(ids, max_tokens, step_limit=max_tokens)   # sampling IS the policy
        elif self.policy_name != "longest_prefix":
            if sampling is not None:
                raise NotImplementedError(f"policy={self.policy_name!r} with do_sample=True is not on the HIP path")
            rows, st = self._decode_host_policy(ids, max_tokens, HipSpecDec.EMIT_BONUS, step_limit=max_tokens)
        else:
            rows, st = self._decode(ids, max_tokens, HipSpecDec.EMIT_BONUS, step_limit=max_tokens, sampling=sampling, self_draft=heads)
        total_ms = st["total_ms"]
        tot_prop = sum(r.proposed for r in rows)
        tot_acc = sum(r.accepted for r in rows)
        tot_gen = sum(len(r.generated) for r in rows)
        batch_metrics = {
            "total_proposed": tot_prop, "total_accepted": tot_acc, "total_generated_tokens": tot_acc,
            "total_steps": st["steps"], "total_draft_time_ms": 0.0, "total_verification_time_ms": st["device_ms"],
            "total_generation_time_ms": total_ms, "total_time_ms": total_ms,
            "tokens_per_sec": tot_acc / (total_ms / 1e3) if total_ms > 0 else 0.0,
            "emitted_tokens": tot_gen, "resyncs": st["resyncs"], "k": st["k"],
        }
        out = []
        for i, (p, r) in enumerate(zip(prompts, rows)):
            n = len(r.generated)
            text = self.base_lm.decode(r.generated) if r.generated else ""
            tps = n / (total_ms / 1e3) if total_ms > 0 else 0.0
            out.append({
                "prompt": p, "text": text, "generated_text": text, "generated_tokens": list(r.generated),
                "num_generated": n, "batch_index": i, "batch_size": len(rows),
                "latency_ms": total_ms / n if n else 0.0, "total_time_ms": total_ms,
                "tokens_per_sec": tps, "throughput_tokens_per_sec": tps,
                "acceptance_rate": tot_acc / max(tot_prop, 1),   # batch-wide, as the reference (:3834)
                "proposed": r.proposed, "accepted": r.accepted,
                "draft_avg_ms": 0.0, "verify_avg_ms": st["device_ms"] / max(st["steps"], 1),
                "batch_metrics": batch_metrics, "kv_append_enabled": True, "kv_append_backend": "hip",
                "kv_appended_tokens": n, "kv_append_time_ms": 0.0, "sequence": list(r.seq),
                **({"k_trace": list(r.k_trace)} if r.k_trace else {}),     # per-row adaptive K: the k of each of the row's steps
            })
        return out

    def generate_many(self, prompts: Sequence[PromptLike], max_tokens: Optional[int] = None, batch_size: int = 8,
                      temperature: Optional[float] = None, do_sample: Optional[bool] = None, **kwargs) -> List[Dict[str, Any]]:
        """Continuous batching over a list of prompts: `batch_size` rows decode together (generate_batch
        semantics per row) and a finished row's slot is handed to the next waiting prompt at once, instead of
        the reference harness' fixed batches that idle until their slowest row ends
        (comprehensive_k_sweep.py:444-535). Rows are independent, so every result equals the prompt's own
        generate_batch([prompt]) run. Results come back in prompt order."""
        if not prompts:
            return []
        max_tokens = max_tokens or self.config["max_new_tokens"]
        temperature = temperature or self.config["temperature"]
        do_sample = do_sample if do_sample is not None else self.config["do_sample"]
        sampling = self._sampling_config(do_sample, temperature, kwargs)
        if self.draft_lm is None:
            raise ValueError("generate_many drafts with the draft model: pass draft_lm / draft_model")
        ids = [self._encode(p) for p in prompts]
        n_slots = max(1, min(int(batch_size), len(ids)))
        order = sorted(range(len(ids)), key=lambda i: -len(ids[i]))   # the longest prompts size the session
        first = order[:n_slots]
        waiting = [i for i in range(len(ids)) if i not in set(first)]
        t_start = time.time()
        sess = DecodeSession(self, [ids[i] for i in first], max_tokens, HipSpecDec.EMIT_BONUS, sampling, max_tokens)
        owner: List[Optional[int]] = list(first)
        done: Dict[int, Any] = {}
        while len(done) < len(ids):
            if sess.any_active():
                if not sess.advance():
                    break
            for b, r in enumerate(sess.rows):
                if owner[b] is not None and not r.active:
                    done[owner[b]] = (r, (time.time() - t_start) * 1e3)
                    owner[b] = None
                    if waiting:
                        nxt = waiting.pop(0)
                        sess.admit(b, ids[nxt])
                        owner[b] = nxt
            if not sess.any_active() and not waiting and all(o is None for o in owner):
                break
        sess.finish()
        torch.cuda.synchronize()
        total_ms = (time.time() - t_start) * 1e3
        tot_tok = sum(len(r.generated) for r, _ in done.values())
        out = []
        for i, p in enumerate(prompts):
            r, t_ms = done[i]
            n = len(r.generated)
            text = self.base_lm.decode(r.generated) if r.generated else ""
            out.append({"prompt": p, "text": text, "generated_text": text, "generated_tokens": list(r.generated), "num_generated": n,
                        "batch_index": i, "batch_size": n_slots, "latency_ms": t_ms, "total_time_ms": total_ms,
                        "tokens_per_sec": n / (t_ms / 1e3) if t_ms > 0 else 0.0,
                        "throughput_tokens_per_sec": tot_tok / (total_ms / 1e3) if total_ms > 0 else 0.0,
                        "acceptance_rate": r.accepted / max(r.proposed, 1), "proposed": r.proposed, "accepted": r.accepted,
                        "steps": r.steps, "sequence": list(r.seq), "kv_append_enabled": True, "kv_append_backend": "hip",
                        **({"k_trace": list(r.k_trace)} if r.k_trace else {}),
                        "batch_metrics": {"total_steps": sess.stats["steps"], "device_steps": sess.step, "resyncs": sess.stats["resyncs"],
                                          "void_row_steps": sess.stats["void_row_steps"], "k": sess.k}})
        return out

    def _sysinfo(self) -> Dict[str, Any]:
        return {
            "mem_rss_mb": psutil.Process().memory_info().rss / 1024 / 1024,
            "cuda_mem_allocated_mb": float(torch.cuda.memory_allocated() / 1024 / 1024),
            "cuda_mem_peak_mb": float(torch.cuda.max_memory_allocated() / 1024 / 1024),
            "policy": self.policy.get_info(), "controller": self.controller.get_info(),
            "impl": "hip", "device": self.device, "dtype": "bfloat16", "amp_enabled": False,
            "base_model": self.base_lm.model_name, "draft_model": self.draft_lm.model_name if self.draft_lm is not None else "none (self-drafting from the base model: medusa heads tied to its lm_head / eagle extrapolation)",
            "draft_mode": self.config.get("draft_mode", "vanilla"),
        }


class DecodeSession:
    """One batch of rows being decoded: host mirror of the sequences + the device loop.

    The device advances its own state at the end of a step, so the NEXT steps do not need anything from
    the host. With a fixed K and greedy decoding `advance()` therefore keeps up to two steps launched: while
    the host applies the reference's rules to the record of step s, step s+1 is running and step s+2 is
    queued behind it (records land in two alternating pinned slots, each launch has its own completion
    event) — the GPU never waits for the host's turn. When the rules change a row (it finishes, or a
    de-duplication rewrites it), the steps already launched are void FOR THAT ROW: their records are
    ignored for it, no further step is launched until the queue has drained, the row's device state is
    repaired there (idle stream) and it rejoins. Steps are counted per row, so the reference's step bound
    applies to each row's own valid steps. Adaptive K, the sampled mode and the persistent Medusa heads
    keep the launch -> wait -> rules order (their next launch depends on the host, or on counters that a void
    step would disturb)."""

    def __init__(self, pipe: SpeculativePipeline, prompts: List[List[int]], max_tokens: int, emit_mode: int,
                 sampling: Optional[Dict[str, Any]] = None, step_limit: Optional[int] = None, self_draft: bool = False):
        self.pipe, self.max_tokens, self.emit_mode = pipe, max_tokens, emit_mode
        self.self_draft = self_draft
        self.sampling = sampling
        self.step_limit = step_limit
        if sampling is not None and emit_mode != HipSpecDec.EMIT_BONUS:
            raise ValueError("sampling is a generate_batch (bonus-token) feature")
        self.rows = [_Row(list(p)) for p in prompts]
        for r in self.rows:
            if not r.seq:
                raise ValueError("empty prompt")
        self.eos = pipe.base_lm.get_tokenizer_info().get("eos_token_id")
        ctl = pipe.controller
        k_max = getattr(ctl, "max_k", None) or getattr(ctl, "k", 4)
        self.need = max(len(r.seq) for r in self.rows) + max_tokens + 2 * int(k_max) + 8
        # per-row adaptive K: the step keeps the shape max_k, the device moves every row's own k (sd_specdec_set_adaptive)
        self.per_row = bool(getattr(ctl, "per_row", False))
        if self.per_row and self_draft:
            raise NotImplementedError("per-row adaptive K needs a draft model (the self-draft modes keep a fixed K)")
        if self.per_row:
            self.k = int(ctl.max_k)
            self.row_ctl = [self._new_row_controller() for _ in self.rows]
        else:
            self.k = pipe._effective_k(ctl.get_k(1, {"step": 1, "generated_tokens": 0, "acceptance_rate": 0.0}), self_draft)
        self.rt, self.loop = pipe._runtime(len(self.rows), self.need, self.k, emit_mode, self.self_draft, adaptive=self.per_row)
        # positions a row may use: the cache rows and both models' position tables
        self.pos_limit = min(self.rt["l_max"], pipe.base_lm.config.max_pos,
                             pipe.draft_lm.config.max_pos if not self_draft else pipe.base_lm.config.max_pos)
        for r in self.rows:
            if len(r.seq) + 2 * self.k + 4 > self.pos_limit:
                raise ValueError(f"prompt of {len(r.seq)} tokens leaves no room for a step within {self.pos_limit} positions")
        self._engines = [e for e in (self.rt["target"], self.rt["draft"]) if e is not None]
        for e in self._engines:                 # paged KV: the previous session's pages go back to the pool
            for b in range(len(self.rows)):
                e.release(b)
        self._fresh: set = set()                # rows whose slot was handed to a new sequence (admit)
        pipe._prefill(self.rt, self.rows)
        self.loop.join_current_stream()
        for b, r in enumerate(self.rows):
            pipe._set_row(self.loop, b, r)
        if self.per_row:
            params = (ctl.initial_k, ctl.min_k, ctl.max_k, ctl.step_size, ctl.target_acceptance_rate)
            if getattr(self.loop, "_adaptive_params", None) != params:
                self.loop.set_adaptive(True, ctl.initial_k, ctl.min_k, ctl.max_k, ctl.step_size, ctl.target_acceptance_rate)
                self.loop._adaptive_params = params       # (enabling restarts every row; drops the captured step once)
            else:
                for b in range(len(self.rows)):
                    self.loop.set_adaptive_row(b, ctl.initial_k)
        self._apply_sampling()
        self._stateful_draft = self_draft and (pipe.medusa_heads is not None or pipe._eagle())
        if self_draft and pipe._eagle():
            self.loop.reset_eagle()   # the reference keeps the state on the pipeline object, across generate() calls even; here a run starts clean
        self.stats = {"steps": 0, "resyncs": 0, "proposed": 0, "accepted": 0, "device_ms": 0.0, "void_row_steps": 0}
        self.step = 0
        from ..policies.controllers import FixedKController

        # (persistent Medusa heads / EAGLE-lite: a void step would replace the next proposals with ones derived from stale
        # state — harmless for the tokens, but the counters would no longer be those of the in-order loop)
        self._early = ((isinstance(ctl, FixedKController) or self.per_row) and sampling is None
                       and not self._stateful_draft
                       and os.environ.get("SPECDEC_EARLY_LAUNCH", "1") != "0")
        self._depth = 2 if self._early else 1
        if os.environ.get("SPECDEC_LAUNCH_DEPTH"):
            self._depth = max(1, min(2, int(os.environ["SPECDEC_LAUNCH_DEPTH"])))
        from collections import deque

        self._queue = deque()                # launched, not yet consumed: (launch index, set of rows it is void for)
        self._flagged: Dict[int, str] = {}   # rows whose device state must be repaired before the next launch
        # The persistent draft forward serves rows of up to 1280 positions (one CU walks a head's whole cache); a session whose
        # cache is sized for more starts on it all the same: every launch point tells the engines how far the rows can have
        # got by the end of the step (sd_model_set_length_hint), and when that moves a model to the other path the queue is
        # drained once and the step captured again (the captured kernels are correct at any length, only slower past the bound).
        self._hint()

    def _reach(self) -> int:
        """Positions the steps in flight plus one more can have reached on the longest row."""
        return max(len(r.seq) for r in self.rows) + (len(self._queue) + 2) * (self.k + 1) + 2

    def _hint(self) -> tuple:
        reach = self._reach()
        for e in self._engines:
            e.set_length_hint(reach)
        return tuple(e.persist_active(1) for e in self._engines)

    def any_active(self) -> bool:
        return any(r.active for r in self.rows)

    def _new_row_controller(self):
        """The host's mirror of a row's device-side controller: the same rule, already past the reference's first get_k
        (which reports acceptance_rate 0.0 before any step)."""
        m = self.pipe.controller.fork()
        m.get_k(1, {"step": 1, "generated_tokens": 0, "acceptance_rate": 0.0})
        return m

    def _apply_sampling(self) -> None:
        """Loops are cached per (batch, K): (re)configure the one in use for this run."""
        sp = self.sampling
        if sp is None:
            if getattr(self.loop, "sampling", False):
                self.loop.sync()
                self.loop.set_sampling(False)
            return
        self.loop.sync()
        self.loop.set_sampling(True, sp["temperature"], sp["top_k"], sp["top_p"], sp["seed"],
                               stream_ids=list(range(len(self.rows))), draw_counts=[r.draws for r in self.rows])

    def _resampler(self, b: int, row: _Row):
        """Draw at another position of the step's logits with the row's current Philox draw."""
        from specdec_hip.ops import sample_token_hip

        sp, loop = self.sampling, self.loop

        def resample(pos: int) -> int:
            with torch.cuda.stream(loop.stream_t):
                tok = sample_token_hip(loop.step_logits[b, pos], sp["temperature"], sp["top_k"], sp["top_p"], seed=sp["seed"],
                                       draw=row.draws, stream_ids=torch.tensor([b], dtype=torch.int32, device="cuda"))
                return int(tok.item())
        return resample

    # ---- device-side repairs, applied only while the loop's stream is idle
    def _repair_rows(self) -> None:
        pipe, loop, rt = self.pipe, self.loop, self.rt
        for b, what in self._flagged.items():
            r = self.rows[b]
            if b in self._fresh:                   # a new sequence in the slot: the old one's pages go back first (idle stream)
                for e in self._engines:
                    e.release(b)
                self._fresh.discard(b)
            if what == "resync" and r.active:      # the host rules rewrote the row: rebuild its caches
                pipe._prefill_row(rt, b, r.seq)
                loop.join_current_stream()
            pipe._set_row(loop, b, r)              # (re)position, or freeze a finished row
            if self.per_row:                       # the device counted steps the host voided: hand it the in-order view
                m = self.row_ctl[b]
                loop.set_adaptive_row(b, m.current_k, r.strict_acc, r.strict_prop, m.acceptance_history[-4:])
        self._flagged.clear()

    def _recover(self, err: Exception) -> None:
        """A persistent launch of one of the models gave up (its bounded waits expired: the CUs were shared with another
        kernel). Every step since is void: drain the loop, clear the engines' health words and move them to the launch path
        (HipModel.recover), drop the captured step, rebuild every active row's caches from the host's sequences. Twice in one
        session: give up for real."""
        n = self.stats["engine_recoveries"] = self.stats.get("engine_recoveries", 0) + 1
        if n > 2:
            raise err
        self.pipe.logger.warning("decode session: %s — continuing on the launch path", err)
        self.loop.drain()
        self._queue.clear()
        for e in self._engines:
            e.recover(self.loop.stream_t)
        self.loop.invalidate()
        self.loop._captured_paths = None
        for b, r in enumerate(self.rows):
            self._flagged[b] = "resync" if r.active else "freeze"

    @property
    def _inflight(self) -> bool:
        return bool(self._queue)

    def _launch(self) -> None:
        """Only with an empty queue are repairs possible (idle stream); callers guarantee it when rows are flagged."""
        if self._flagged or getattr(self, "_resample_state", False):
            assert not self._queue
            self._repair_rows()
            if getattr(self, "_resample_state", False):
                self._resample_state = False
                self._apply_sampling()
        if not self._queue:
            paths = self._hint()
            was = getattr(self.loop, "_captured_paths", None)
            if was is not None and was != paths:      # a model changed path (rows passed the persistent launch's context bound)
                self.loop.drain()
                self.loop.invalidate()
                self.stats["path_switches"] = self.stats.get("path_switches", 0) + 1
            self.loop._captured_paths = paths
        idx = self.loop.launches
        if self._engines[0].page_len is not None:
            # paged KV: the device advances by itself, so every row gets the pages of all the steps in flight plus this one
            # before it is launched (table writes go to the loop's stream, ahead of the step). Frozen rows keep writing their
            # K+1 positions in place and keep their pages until the slot is handed on.
            reach = (len(self._queue) + 2) * (self.k + 1) + 2
            for b, r in enumerate(self.rows):
                for e in self._engines:
                    e.reserve(b, len(r.seq) + reach, stream=self.loop.stream_t)
        self.loop.step(use_graph=True)
        self._queue.append((idx, set()))

    def _can_launch_ahead(self) -> bool:
        """Conservative: some row cannot finish within the steps already launched, and nothing waits for a repair."""
        if self._flagged or getattr(self, "_resample_state", False):
            return False
        if self._hint() != getattr(self.loop, "_captured_paths", None):
            return False                   # a path switch is due: the queue drains first (_launch re-captures)
        ahead = len(self._queue) + 1       # steps whose records are still to come, counting the one being considered
        per_step = self.k + 1 if self.emit_mode == HipSpecDec.EMIT_BONUS else self.k
        for b, r in enumerate(self.rows):
            if not r.active:
                continue
            if len(r.generated) + ahead * per_step >= self.max_tokens:
                continue
            if self.step_limit is not None and r.steps + ahead + 1 > self.step_limit:
                continue
            if len(r.seq) + (ahead + 2) * self.k + 6 > self.pos_limit:
                continue
            return True
        return False

    def advance(self) -> bool:
        """One draft-then-verify step for every active row. Returns False when the controller
        stops the run."""
        pipe, rows, stats = self.pipe, self.rows, self.stats
        self.step += 1
        step = self.step
        if step > 1 and not self._early and not self.per_row:
            ctx = {"step": step, "generated_tokens": max(len(r.generated) for r in rows),
                   "acceptance_rate": stats["accepted"] / max(stats["proposed"], 1)}
            k_new = int(pipe.controller.get_k(step, ctx))
            if k_new <= 0:
                return False
            k_new = pipe._effective_k(k_new, self.self_draft)
            if k_new != self.k:  # adaptive K: another captured step over the same caches
                self.loop.sync()
                self._repair_rows()
                self.k = k_new
                self.rt, self.loop = pipe._runtime(len(rows), self.need, self.k, self.emit_mode, self.self_draft)
                self.loop.join_current_stream()
                for b, r in enumerate(rows):
                    pipe._set_row(self.loop, b, r)
                self._apply_sampling()
        k, loop, rt = self.k, self.loop, self.rt
        t0 = time.time()
        if not self._queue:
            self._launch()              # (repairs flagged rows first: the stream is idle)
        while self._early and len(self._queue) < self._depth and self._can_launch_ahead():
            self._launch()              # keep the GPU's queue ahead of the host
        idx, void = self._queue.popleft()
        try:
            rec = loop.wait(idx) if self._queue else loop.sync()
        except EngineGaveUp as e:
            self._recover(e)
            return True                 # nothing was taken from the void steps; the rows rejoin at the next launch
        self.last_record = rec
        while self._early and len(self._queue) < self._depth and self._can_launch_ahead():
            self._launch()              # step s+1 (s+2) runs while the rules of step s are applied below
        stats["device_ms"] += (time.time() - t0) * 1e3
        for b, r in enumerate(rows):
            if not r.active:
                continue
            if b in void:           # launched before the host repaired this row: nothing to take from it
                stats["void_row_steps"] += 1
                continue
            a = int(rec.accept_len[b])
            t = [int(x) for x in rec.target_ids[b]]
            d = [int(x) for x in rec.draft_tokens[b]]
            before, acc0 = r.seq, r.accepted
            if self.per_row:                    # the k that counted for this row in this step, from the device's controller
                k = int(rec.k_row[b])
                m = self.row_ctl[b]
                if k != m.current_k:
                    raise RuntimeError(f"row {b}: the device ran step {r.steps + 1} at k={k}, the host's controller says {m.current_k}")
                r.k_trace.append(k)
                r.strict_acc += a
                r.strict_prop += k
                m.get_k(r.steps + 2, {"step": r.steps + 2, "acceptance_rate": r.strict_acc / max(r.strict_prop, 1)})
            if self.sampling is not None:
                t[a] = int(rec.new_tokens[b][a])       # the token the device sampled at position a
            assumed = before + t[: int(rec.n_new[b])]  # what the device advanced to
            if self.emit_mode == HipSpecDec.EMIT_BONUS:
                pipe._rules_batch(r, k, a, t, self.max_tokens, self.eos,
                                  self._resampler(b, r) if self.sampling is not None else None)
                if self.sampling is not None:
                    r.draws += 1
            else:
                pipe._rules_single(r, k, a, d, t, self.max_tokens, self.eos)
            r.steps += 1
            r.counters.append((r.proposed, r.accepted, len(r.generated), a))
            if a >= k:
                stats["full_accepts"] = stats.get("full_accepts", 0) + 1
            stats["proposed"] += k
            stats["accepted"] += r.accepted - acc0
            if r.active and self.step_limit is not None and r.steps >= self.step_limit:
                r.active = False                # the reference's loop bound counts steps (pipeline.py:1984, :984)
            if r.active and len(r.seq) + 2 * self.k + 4 > self.pos_limit:
                r.active = False                # out of cache rows / model positions
            if not r.active:
                self._flagged[b] = "freeze"     # stop the row on the device
            elif r.seq != assumed:
                stats["resyncs"] += 1
                self._flagged[b] = "resync"
            if b in self._flagged:
                for _, vs in self._queue:      # every step already launched is void for this row
                    vs.add(b)
        stats["steps"] = max(r.steps for r in rows)
        return True

    def admit(self, b: int, prompt: List[int]) -> None:
        """Continuous batching: put a new sequence into the slot of a finished row. Its caches are prefilled
        and its device state set at the next launch point; the other rows keep stepping meanwhile."""
        if self.rows[b].active:
            raise ValueError(f"row {b} is still decoding")
        if not prompt:
            raise ValueError("empty prompt")
        if len(prompt) + self.max_tokens + 2 * self.k + 8 > self.rt["l_max"] or len(prompt) + 2 * self.k + 4 > self.pos_limit:
            raise ValueError(f"prompt of {len(prompt)} tokens does not fit this session (l_max {self.rt['l_max']}, positions {self.pos_limit})")
        self.rows[b] = _Row(list(prompt))
        self._fresh.add(b)
        if self.per_row:
            self.row_ctl[b] = self._new_row_controller()
        self._flagged[b] = "resync"
        for _, vs in self._queue:
            vs.add(b)
        if self.sampling is not None:
            self._resample_state = True   # draw counters restart for the new row at the next launch point

    def finish(self) -> None:
        """Drain a step launched ahead of a run that ended, and leave the device rows consistent."""
        if self._queue:
            self.loop.sync()
            self._queue.clear()
        self._repair_rows()
