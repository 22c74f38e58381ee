"""CPU restatement of the reference draft-then-verify loop.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, for greedy decoding (`do_sample=False`, the SPECDEC_DETERMINISTIC setting of
BASELINE.json):
  * `SpeculativePipeline.generate_batch` — src/specdec/core/pipeline.py:1984-3733
  * `SpeculativePipeline.generate`       — src/specdec/core/pipeline.py:984-1275
  * `LongestPrefixPolicy.accept_tokens`  — src/specdec/policies/policies.py:156-180
  * greedy `sample_bonus_token_from_logits` — pipeline.py:48-147 (argmax + clamp)
on top of `oracle.model_ref.OracleLM` (the HF forward restated).

Rows are treated independently: the reference right-pads mixed-length batches every
step (sequence_utils.py:52-57) and its stream path takes the logits of the last —
possibly padded — position (hf_wrappers.py:272-627), so rows shorter than the longest
one are generated from a pad token. That is a defect of the reference's batching, not
semantics; the per-row rules below are exactly what it applies to a batch of one and to
the longest row of any batch (SURVEY §7 viii). Everything else, including the
de-duplication heuristics that can drop or rewind tokens, is reproduced as written.

`step_rules_batch` / `step_rules_single` are pure functions of (a, draft ids, target
argmax ids, host state) so that the same restatement can check both a CPU run (ids
from OracleLM) and a GPU run (ids from the HIP step record).
"""

from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch

from .model_ref import OracleLM, hf_sampling_probs


def longest_prefix(draft: Sequence[int], target_argmax: Sequence[int]) -> int:
    """policies.py:156-180: first index where argmax(base_logits[i]) != proposed[i]."""
    a = 0
    for d, t in zip(draft, target_argmax):
        if int(d) != int(t):
            break
        a += 1
    return a


@dataclass
class RowState:
    seq: List[int]                      # current_input_ids[i] (prompt + appended tokens)
    generated: List[int] = field(default_factory=list)   # batch_generated_tokens[i]
    active: bool = True
    proposed: int = 0
    accepted: int = 0
    steps: int = 0
    draws: int = 0                      # sampled steps so far (index of the next Philox draw)


def step_rules_batch(row: RowState, k: int, a: int, draft: Sequence[int], t: Sequence[int],
                     max_tokens: int, eos: Optional[int], vocab: int, bonus_at=None) -> List[int]:
    """One row of one generate_batch step. `t[i]` is the target's greedy token after the
    row's sequence + draft[:i] (i = 0..k), which for i <= a is what the reference's
    autoregressive base pass yields (base_tokens[i], and the extra forward for i == k).

    Returns the tokens appended to the row's SEQUENCE (which the de-duplication can make
    differ from what was appended to its generated list).

    `bonus_at(pos)` (do_sample=True): the token SAMPLED from the target logits of position pos
    (sample_bonus_token_from_logits, :3156 / :3231 / :3355); greedy uses t[pos].
    """
    clamp = lambda x: max(0, min(int(x), vocab - 1))  # validate_and_clamp_tokens
    pick = bonus_at if bonus_at is not None else (lambda pos: t[pos])
    gen = row.generated
    if a > 0:
        acc = [clamp(x) for x in t[:a]]                       # :3059-3075 accepted = base_tokens[:a]
        if eos is not None and eos in acc:                    # :3120-3131 accepted EOS is cut
            acc = acc[: acc.index(eos)]
            row.active = False
        # bonus: logits at position a (a < k, :3140-3163) or the extra forward over
        # seq + acc (a == k, :3164-3231; acc possibly EOS-cut, so position len(acc))
        bonus = clamp(pick(a if a < k else len(acc)))
        if eos is not None and bonus == eos:                  # :3274-3280 bonus EOS is kept
            row.active = False
        acc = acc + [bonus]
        if gen and acc:                                       # :3295-3318 overlap with generated tail
            for c in range(min(5, len(gen), len(acc)), 0, -1):
                if gen[-c:] == acc[:c]:
                    acc = acc[c:]
                    break
        tokens_to_add = acc
        gen.extend(tokens_to_add)                             # :3321-3322
        accepted_len = len(tokens_to_add)
        accepted_tokens = list(tokens_to_add)
    else:
        first = clamp(pick(0))                                # :3328-3345 first base token
        accepted_tokens = [first]
        if eos is not None and first == eos:                  # :3360-3365 EOS dropped, row stops
            row.active = False
            accepted_tokens = []
        if accepted_tokens and gen and gen[-1] == accepted_tokens[0]:   # :3367-3376
            accepted_tokens = []
        if accepted_tokens:
            gen.extend(accepted_tokens)
        accepted_len = len(accepted_tokens)
    row.proposed += k                                         # :3421-3424
    row.accepted += accepted_len
    appended: List[int] = []
    if accepted_tokens:
        acc = [clamp(x) for x in accepted_tokens]
        cur = row.seq
        if cur and acc:                                       # :3470-3530 overlap with sequence tail
            c = min(5, len(cur), len(acc))
            if c > 0 and cur[-c:] == acc[:c]:
                if len(acc) > c:
                    acc = acc[c:]
                    if len(gen) >= c:
                        del gen[-c:]
                else:
                    acc = []
        appended = acc
        row.seq = cur + acc
    if len(gen) >= max_tokens:                                # :3589-3590
        row.active = False
    row.steps += 1
    return appended


def step_rules_single(row: RowState, k: int, a: int, draft: Sequence[int], t: Sequence[int],
                      max_tokens: int, eos: Optional[int]) -> List[int]:
    """One step of generate(): accepted tokens are the DRAFT ids cut to the remaining
    budget (:1190-1206), a zero-accept step takes one base token (:1209-1225), no bonus."""
    remaining = max_tokens - len(row.generated)
    if a > 0:
        new = [int(x) for x in draft[: min(a, max(remaining, 0))]]
    else:
        new = [int(t[0])] if remaining > 0 else []
    row.proposed += k
    row.accepted += a
    row.generated.extend(new)
    row.seq = row.seq + new
    row.steps += 1
    if len(row.generated) >= max_tokens:                      # :1258-1263
        row.active = False
    elif eos is not None and row.generated and row.generated[-1] == eos:   # :1266-1272
        row.active = False
    return new


def medusa_draftor_ref(base: OracleLM, ids: torch.Tensor, k: int, num_heads: int, temperature: float) -> List[int]:
    """MedusaDraftor.generate_tokens (src/specdec/modes/medusa.py:104-186) with head_init "tie" / "copy" and top_p = 1:
    every head is the base lm_head, evaluated on the same last hidden state (hidden_states[-1], i.e. after the final
    norm) for each of the k proposals; each head in range draws once per proposal from softmax(logits / T) with
    torch.multinomial (global generator); head 0's draw is the proposal. T -> 0 is the greedy limit."""
    base.forward(ids)
    h = base._norm(base.last_hidden[:, -1:, :], "nf", base.w.final_norm_w, base.w.final_norm_b)
    lg = base._r(torch.nn.functional.linear(h, base._m("lm_head", base.w.lm_head)))
    if temperature > 0:
        lg = lg / temperature
    probs = torch.softmax(lg, dim=-1).squeeze(1)
    out = []
    for step in range(k):
        toks = [torch.multinomial(probs, 1) for _ in range(min(num_heads, k - step))]
        out.append(int(toks[0][0, 0]))
    return out


class OraclePipeline:
    """The reference loop on CPU. `reprefill=True` re-feeds the whole prefix for every
    one of the 2K forwards of a step, as the reference does with KV append off
    (pipeline.py:1838; the configuration of every published Llama run) — that is the
    `cpu_baseline` cost model; `reprefill=False` keeps per-model KV caches and gives the
    same tokens faster (used by the parity tests)."""

    def __init__(self, base: OracleLM, draft: Optional[OracleLM], k: int = 4, eos_token_id: Optional[int] = None,
                 reprefill: bool = False, draft_mode: str = "vanilla", medusa_heads=None, eagle_alpha: float = 0.7,
                 medusa_num_heads: int = 2, medusa_temperature: float = 0.7, policy: str = "longest_prefix",
                 policy_params: Optional[Dict] = None):
        """draft_mode "medusa_tied": the reference's MedusaDraftor (src/specdec/modes/medusa.py:71-186) with
        head_init tie/copy under greedy decoding — every head is the base lm_head and head 0 is evaluated on the
        same last hidden state for each of the K proposals, so the draft is K copies of the base model's own
        next token (no draft model)."""
        self.base, self.draft, self.k = base, draft, int(k)
        self.draft_mode = draft_mode
        # "medusa_heads" (not in the reference): persistent heads [K][V][d]; the proposals of a step
        # are argmax head_i(final_norm(h)) with h = the base model's residual row at the position that produced the
        # last emitted token in the PREVIOUS step's verify pass; a row's first step proposes zeros
        self.medusa_heads = medusa_heads
        # "eagle": the reference's _run_eagle_hf (pipeline.py:765-889) under greedy decoding; k = min(k, eagle.max_draft)
        # is the caller's. State per row = the LAST extrapolated hidden row of the previous step (what
        # `_eagle_last_hidden_states[:, -1:]` holds when the next call concatenates the new hidden state to it).
        self.eagle_alpha = float(eagle_alpha)
        # "medusa_random": the reference PIPELINE's Medusa mode (_run_medusa_hf): fresh random heads per call, sampled
        self.medusa_num_heads, self.medusa_temperature = int(medusa_num_heads), float(medusa_temperature)
        self._eagle_state: Dict[int, torch.Tensor] = {}
        self._next_draft: Dict[int, List[int]] = {}
        # acceptance policy (policies.py:76-425). "longest_prefix" compares ids (exact match: one cached K+1 pass gives the
        # same answer as the reference's autoregressive base pass); the logit-threshold policies need what the reference
        # computes — the base model's OWN K greedy tokens and their logits from the same prefix
        # (speculative_scheduler.py:294-368), and the draft's logits
        self.policy, self.policy_params = policy, dict(policy_params or {})
        self.eos = eos_token_id
        self.reprefill = reprefill
        self.vocab = base.cfg.vocab
        self.trace: List[Dict] = []
        self.sample_draft: Optional[float] = None   # temperature of a generate(do_sample=True) call
        self.sample_kw: Dict = {"top_k": 50, "top_p": None}

    def _propose_and_verify(self, seq: List[int], row: int = 0):
        """draft K greedy tokens from seq (pipeline.py:2397-2462) and the target's greedy
        tokens t_0..t_K conditioned on seq + draft[:i]."""
        k = self.k
        ids = torch.tensor([seq], dtype=torch.int64)
        if self.policy != "longest_prefix":
            return self._propose_and_verify_policy(ids, seq, k)
        if self.draft_mode == "medusa_heads":
            draft = self._next_draft.get(row, [0] * k)
            lg, _ = self.base.forward(torch.tensor([seq + draft], dtype=torch.int64))
            self.last_logits = lg[0, len(seq) - 1:]
            t = self.last_logits.argmax(-1).tolist()
            a = longest_prefix(draft, t)
            h = self.base.last_hidden[0, len(seq) - 1 + a]
            self._next_draft[row] = self.base.head_tokens(h, self.medusa_heads)
            return draft, t, a
        if self.draft_mode == "eagle":
            draft = self._eagle_draft(ids, row, k)
        elif self.draft_mode == "medusa_random":
            draft = self._medusa_random_draft(ids, k)
        elif self.draft_mode == "medusa_tied":
            t0, _ = self.base.generate_tokens(ids, 1, reprefill=self.reprefill)
            draft = [int(t0[0, 0])] * k
        else:
            # (generate(do_sample=True), pipeline.py:1019-1027: the DRAFT's k tokens are drawn at the call's temperature; every
            #  verification path of the scheduler is greedy, speculative_scheduler.py:192-199 / :304-310 / :339-345)
            d_ids, _ = self.draft.generate_tokens(ids, k, reprefill=self.reprefill, do_sample=self.sample_draft is not None,
                                                  temperature=self.sample_draft or 1.0,
                                                  eos_token_id=self.eos if self.sample_draft is not None else None, **self.sample_kw)
            draft = d_ids[0].tolist()
            k = len(draft)   # (a sampled draft that drew EOS is shorter: `proposed` counts draft_tokens.shape[1], pipeline.py:1122)
        if self.reprefill:
            # reference-faithful cost: the base model generates K tokens autoregressively
            # from the same prefix (speculative_scheduler.py:192-199), + the extra forward
            # when everything was accepted (pipeline.py:3199-3206)
            b_ids, b_lg = self.base.generate_tokens(ids, k, reprefill=True)
            self.logits0 = b_lg[0, 0]
            base = b_ids[0].tolist()
            a = longest_prefix(draft, base)
            t = list(base)
            if a == k:
                lg, _ = self.base.forward(torch.tensor([seq + base], dtype=torch.int64))
                t.append(int(lg[0, -1].argmax()))
            else:
                t.append(-1)
            # t[i] for i > a is not the target's token after draft[:i]; the rules only
            # read t[:a+1]
            return draft, t, a
        # one cached pass over seq + draft: logits at the last K+1 positions
        lg, _ = self.base.forward(torch.tensor([seq + draft], dtype=torch.int64))
        self.last_logits = lg[0, len(seq) - 1 :]     # [K+1][V]: what the sampled bonus token is drawn from
        self.logits0 = self.last_logits[0]
        t = self.last_logits.argmax(-1).tolist()
        a = longest_prefix(draft, t)
        return draft, t, a

    def _propose_and_verify_policy(self, ids: torch.Tensor, seq: List[int], k: int):
        """pipeline.py:1019-1100 / :2397-3030 with a logit-threshold policy: K greedy draft tokens with their logits, the
        base model's own K greedy tokens with their logits from the SAME prefix, the policy on (draft ids, draft
        logits, base logits); t = base tokens (+ the extra forward's token after a full acceptance, :3199-3206)."""
        from .hostlogic_ref import conf_threshold_accept, topk_agree_accept, typical_accept

        d_ids, d_logits = self.draft.generate_tokens(ids, k, reprefill=self.reprefill)
        b_ids, b_logits = self.base.generate_tokens(ids, k, reprefill=self.reprefill)
        draft, base = d_ids[0].tolist(), b_ids[0].tolist()
        pp = self.policy_params
        if self.policy == "conf_threshold":
            a = conf_threshold_accept(draft, d_logits[0], float(pp.get("tau", 0.5)))
        elif self.policy == "topk_agree":
            a = topk_agree_accept(draft, b_logits[0], int(pp.get("k", 5)))
        elif self.policy == "typical":
            a = typical_accept(draft, b_logits[0], float(pp.get("p", 0.9)))
        else:
            raise ValueError(f"unknown policy {self.policy!r}")
        t = list(base)
        if a == k:
            lg, _ = self.base.forward(torch.tensor([seq + base], dtype=torch.int64))
            t.append(int(lg[0, -1].argmax()))
        else:
            t.append(-1)
        return draft, t, a

    def _medusa_random_draft(self, ids: torch.Tensor, k: int) -> List[int]:
        """pipeline.py:655-763 (_run_medusa_hf), what draft_mode="medusa" runs for HF models: on EVERY call `num_heads`
        fresh `nn.Linear(hidden, vocab, bias=False)` heads are created (their default init draws from the global
        generator) and re-initialised with normal_(0, 0.02); all heads read the SAME last hidden state (after the final
        norm: hidden_states[-1]) for every one of the k proposals; every head still in range draws one token with
        torch.multinomial from softmax(logits / T) (global generator), and the proposal of a step is head 0's draw.
        The restatement consumes the global torch generator in the same order, so under torch.manual_seed it replays
        the reference's draws. (bf16 mode: heads and logits rounded to bf16, as a bf16 engine holds them.)"""
        b = self.base
        b.forward(ids)
        h = b._norm(b.last_hidden[:, -1:, :], "nf", b.w.final_norm_w, b.w.final_norm_b)     # [1][1][d]
        heads = []
        for _ in range(self.medusa_num_heads):
            head = torch.nn.Linear(b.cfg.d_model, self.vocab, bias=False)
            torch.nn.init.normal_(head.weight, 0, 0.02)
            heads.append(b._r(head.weight.detach()))
        self.last_medusa_heads = heads
        draft = []
        for step in range(k):
            toks = []
            for hi in range(min(self.medusa_num_heads, k - step)):
                lg = b._r(torch.nn.functional.linear(h, heads[hi]))                        # [1][1][V]
                if self.medusa_temperature > 0:
                    lg = lg / self.medusa_temperature
                toks.append(torch.multinomial(torch.softmax(lg, dim=-1).squeeze(1), 1))
            draft.append(int(toks[0][0, 0]))
        return draft

    def _eagle_draft(self, ids: torch.Tensor, row: int, k: int) -> List[int]:
        """pipeline.py:786-858: h_t = last hidden state (after the final norm) of the last position; states = [E, h_t]
        (first call: [h_t], where the fallback `h_next = current_hidden` and the then-equal pair make every proposal
        lm_head(h_t)); h_next = h_t + alpha * (h_t - h_{t-1}) with the window sliding over the extrapolated rows;
        token = argmax lm_head(h_next). In bf16 mode every tensor operation rounds, as the reference's bf16 tensors do
        (alpha enters as a float32 scalar)."""
        b = self.base
        b.forward(ids)
        x_last = b.last_hidden[0, -1]
        h_t = b._norm(x_last.view(1, 1, -1), "nf", b.w.final_norm_w, b.w.final_norm_b).view(-1)
        prv = self._eagle_state.get(row, h_t)
        cur = h_t
        head = b._m("lm_head", b.w.lm_head)
        alpha = torch.tensor(self.eagle_alpha, dtype=torch.float32)
        draft = []
        for _ in range(k):
            diff = b._r(cur - prv)
            nxt = b._r(cur + b._r(alpha * diff))
            draft.append(int(b._r(nxt.view(1, -1) @ head.t()).view(-1).argmax()))
            prv, cur = cur, nxt
        self._eagle_state[row] = cur
        return draft

    def generate_batch(self, prompts: Sequence[Sequence[int]], max_tokens: int,
                       max_steps: Optional[int] = None, sampling: Optional[Dict] = None,
                       step_log: Optional[List] = None, per_row_k: Optional[Dict] = None) -> List[Dict]:
        """`max_steps` (not in the reference) bounds a timing sample to a number of steps; `step_log`, when given,
        receives (wall clock, tokens generated so far over all rows) after every step (bench.py times segments with it).
        `sampling` = {temperature, top_k, top_p, seed} turns on do_sample=True: drafting and
        verification stay greedy (pipeline.py:2400, :2645); only the token after the accepted
        prefix is sampled (oracle/sampling_ref.py), one Philox draw per row per step, stream = row."""
        if sampling is not None and self.reprefill:
            raise ValueError("sampling needs the cached verify pass (reprefill=False)")
        rows = [RowState(seq=[int(x) for x in p]) for p in prompts]
        # per_row_k (not in the reference; the semantics of sd_specdec_set_adaptive): the step keeps
        # the shape self.k = max_k, but every row has its own AdaptiveKController (controllers.py:63-141) fed with the
        # row's own strict acceptance rate; only the first k_row proposals of a row count
        ctls, strict = None, None
        if per_row_k is not None:
            from .hostlogic_ref import AdaptiveKOracle

            if self.reprefill or self.policy != "longest_prefix":
                raise ValueError("per-row K: cached verify pass and exact-match acceptance only")
            ctls = [AdaptiveKOracle(**per_row_k) for _ in rows]
            for c in ctls:
                c.get_k({"acceptance_rate": 0.0})            # the reference's first call, before any step
            strict = [[0, 0] for _ in rows]
            self.k_trace = [[] for _ in rows]
        self.trace = []
        self._next_draft = {}
        self._eagle_state = {}
        t0 = time.time()
        step = 0
        while step < max_tokens and (max_steps is None or step < max_steps):   # :1984 bound is STEPS, not tokens
            step += 1
            if not any(r.active for r in rows):
                break
            for i, r in enumerate(rows):
                if not r.active:
                    continue
                draft, t, a = self._propose_and_verify(r.seq, i)
                k_row = self.k
                if ctls is not None:
                    k_row = ctls[i].k
                    a = min(a, k_row)
                    self.k_trace[i].append(k_row)
                    strict[i][0] += a
                    strict[i][1] += k_row
                    ctls[i].get_k({"acceptance_rate": strict[i][0] / max(strict[i][1], 1)})
                before = len(r.seq)
                bonus_at = None
                if sampling is not None:
                    from .sampling_ref import sample_token_ref

                    lg_rows = self.last_logits.float().numpy()

                    def bonus_at(pos, _lg=lg_rows, _r=r, _i=i):
                        return sample_token_ref(_lg[pos], sampling["temperature"], sampling.get("top_k"), sampling.get("top_p"),
                                                int(sampling.get("seed", 0)), _r.draws, _i)
                appended = step_rules_batch(r, k_row, a, draft, t, max_tokens, self.eos, self.vocab, bonus_at)
                if sampling is not None:
                    r.draws += 1
                self.trace.append({"step": step, "row": i, "a": a, "draft": draft, "t": t[: a + 1],
                                   "appended": appended, "seq_len": before})
            if step_log is not None:
                step_log.append((time.time(), sum(len(r.generated) for r in rows)))
        dt = time.time() - t0
        return [self._result(r, dt, len(rows), i) for i, r in enumerate(rows)]

    def generate(self, prompt: Sequence[int], max_tokens: int, do_sample: bool = False, temperature: float = 0.7,
                 top_k: Optional[int] = 50, top_p: Optional[float] = None) -> Dict:
        """pipeline.py:893-1413. do_sample=True: the draft model's proposals are sampled (torch's global generator, seeded by
        the caller), verification stays greedy — accepted draft tokens are the base model's greedy tokens; the draws decide
        accept lengths, step count and the proposed / accepted counters. One exception, restated in _generate: the token of a
        zero-accept step is DRAWN from the base model too (pipeline.py:1217-1224)."""
        self.sample_draft = float(temperature) if do_sample else None
        self.sample_kw = {"top_k": top_k, "top_p": top_p}   # (the call's kwargs reach transformers' generate: hf_wrappers.py:230)
        try:
            return self._generate(prompt, max_tokens)
        finally:
            self.sample_draft = None

    def _generate(self, prompt: Sequence[int], max_tokens: int) -> Dict:
        r = RowState(seq=[int(x) for x in prompt])
        self.trace = []
        self._next_draft = {}
        self._eagle_state = {}
        t0 = time.time()
        step = 0
        while len(r.generated) < max_tokens and step < 2 * max_tokens:   # :984-986
            step += 1
            draft, t, a = self._propose_and_verify(r.seq)
            if self.sample_draft is not None and a == 0:
                # zero-accept fallback of a sampling call: ONE base token drawn at the call's temperature from the prefix's
                # logits (pipeline.py:1217-1224 -> hf_wrappers.py:699-716), the step's last draw
                probs = hf_sampling_probs(self.logits0.unsqueeze(0), self.sample_draft, **self.sample_kw)
                t = [int(torch.multinomial(probs, num_samples=1)[0, 0])] + list(t[1:])
            new = step_rules_single(r, len(draft), a, draft, t, max_tokens, self.eos)
            self.trace.append({"step": step, "a": a, "draft": draft, "t": t[: a + 1], "appended": new})
            if not r.active:
                break
        return self._result(r, time.time() - t0, 1, 0)

    @staticmethod
    def _result(r: RowState, dt: float, batch: int, i: int) -> Dict:
        n = len(r.generated)
        return {
            "generated_tokens": list(r.generated), "num_generated": n, "sequence": list(r.seq),
            "proposed": r.proposed, "accepted": r.accepted, "steps": r.steps,
            "acceptance_rate": r.accepted / max(r.proposed, 1),
            "total_time_ms": dt * 1e3, "tokens_per_sec": n / dt if dt > 0 else 0.0,
            "batch_index": i, "batch_size": batch,
        }
