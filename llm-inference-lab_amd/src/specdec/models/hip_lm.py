"""`HipLM` — the `LanguageModel` wrapper over the gfx950 decoder forward.

It stands where the reference has `HFWrapper` (src/specdec/models/hf_wrappers.py:20-1079):
same interface (`generate_tokens`, `encode`/`decode`, `get_tokenizer_info`, the KV hooks),
but the forward is the C-ABI library (csrc/), the KV cache is a preallocated device buffer
appended to in place by the forward itself, and weights come from a local checkpoint
directory or a synthetic initialiser — nothing is fetched by name.

`generate_tokens` keeps the reference contract (k greedy tokens + their logits, one
forward per token after the prefix has been cached). The pipeline's fast path does not
go through it: it drives both models with `specdec_hip.engine.HipSpecDec`, one graph
launch per step."""

from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Tuple

import torch

from specdec_hip import weights as W
from specdec_hip.engine import EngineGaveUp, HipModel

from ..cache.kv_types import KVCache
from ..utils.interfaces import LanguageModel
from ..utils.token_validation import validate_and_clamp_tokens

logger = logging.getLogger(__name__)


class IdTokenizer:
    """Whitespace-separated decimal token ids: the tokenizer of synthetic runs."""

    def __init__(self, vocab_size: int, eos_token_id: Optional[int] = None, pad_token_id: int = 0):
        self.vocab_size = vocab_size
        self.eos_token_id = eos_token_id if eos_token_id is not None else vocab_size - 1
        self.pad_token_id = pad_token_id
        self.bos_token_id = None

    def encode(self, text: str) -> List[int]:
        return [int(t) for t in text.split()]

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        return " ".join(str(int(i)) for i in ids)

    def __call__(self, texts, padding=True, return_tensors="pt", return_attention_mask=True):
        rows = [self.encode(t) for t in ([texts] if isinstance(texts, str) else texts)]
        n = max(len(r) for r in rows)
        # left padding, as generation tokenizers do (pipeline.py:1767 pads to a rectangle)
        ids = torch.tensor([[self.pad_token_id] * (n - len(r)) + r for r in rows], dtype=torch.long)
        mask = torch.tensor([[0] * (n - len(r)) + [1] * len(r) for r in rows], dtype=torch.long)
        return {"input_ids": ids, "attention_mask": mask}


def hf_sampling_probs(logits: torch.Tensor, temperature: float, top_k: Optional[int] = 50, top_p: Optional[float] = None) -> torch.Tensor:
    """The distribution a `do_sample=True` call of the reference's wrapper draws from when it hands the call to transformers'
    `generate` (hf_wrappers.py:208-232, the path with KV append off — the configuration of its published runs): scores / T,
    everything below the k-th largest score to -inf (top_k: the library's sampling default 50 unless the caller passes one; ties
    with the k-th kept), for top_p < 1 the tokens whose ascending cumulative probability is <= 1 - top_p dropped (at least one
    kept), softmax. logits [B][V] fp32, computed where the logits live (the callers move them to the host first, so that a seeded
    run makes the draws of the reference's CPU run)."""
    s = logits.float() / temperature
    if top_k:
        kth = torch.topk(s, min(int(top_k), s.shape[-1]))[0][..., -1, None]
        s = s.masked_fill(s < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_idx = torch.sort(s, descending=False)
        remove = sorted_logits.softmax(dim=-1).cumsum(dim=-1) <= (1.0 - top_p)
        remove[..., -1:] = False
        s = s.masked_fill(remove.scatter(1, sorted_idx, remove), float("-inf"))
    return torch.softmax(s, dim=-1)


def _draw(last: torch.Tensor, temperature: float, kwargs) -> torch.Tensor:
    """One torch.multinomial draw per row on torch's GLOBAL CPU generator (what the reference's CPU run consumes)."""
    probs = hf_sampling_probs(last.detach().float().cpu(), float(temperature), kwargs.get("top_k", 50), kwargs.get("top_p"))
    return torch.multinomial(probs, num_samples=1)


class HipLM(LanguageModel):
    def __init__(self, weights: W.ModelWeights, tokenizer: Any = None, name: Optional[str] = None,
                 max_len: int = 1024, batch: int = 1, device: str = "cuda", weight_dtype: str = "bf16",
                 kv_page_len: Optional[int] = None, kv_pages: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("HipLM needs a GPU: this build has no CPU compute path")
        self._device = torch.device(device if device != "auto" else "cuda")
        self.weights = weights if weights.tok_emb.device.type == "cuda" else weights.to(self._device)
        self.config = self.weights.config
        self.vocab_size = self.config.vocab
        self._tokenizer = tokenizer or IdTokenizer(self.config.vocab, self.config.eos_token_id)
        self._name = name or self.config.name
        self._eos_id = getattr(self._tokenizer, "eos_token_id", self.config.eos_token_id)   # what a sampling call ends at
        self.weight_dtype = weight_dtype   # "fp8": the engines stream an e4m3 copy of the Linear weights
        self._max_len, self._batch = max_len, batch
        # paged KV (sd_model_bind_paged): engines share a pool of kv_pages pages of kv_page_len positions instead of
        # owning l_max positions per row (kv_pages None: enough for every row to reach l_max)
        self.kv_page_len, self.kv_pages = kv_page_len, kv_pages
        self._model: Optional[HipModel] = None
        self._cached: List[List[int]] = []
        self._last_generated_kv: Optional[KVCache] = None

    # ---- engine instances ----------------------------------------------------------
    def new_engine(self, batch: int, l_max: int) -> HipModel:
        """A forward instance with its own KV cache over the shared weights."""
        return HipModel(self.weights, batch=batch, l_max=l_max, device=self._device, weight_dtype=self.weight_dtype,
                        page_len=self.kv_page_len, n_pages=self.kv_pages)

    def _engine(self, batch: int, need_len: int) -> HipModel:
        m = self._model
        if m is None or m.batch < batch or m.l_max < need_len:
            self._model = m = self.new_engine(max(batch, self._batch), max(need_len + 64, self._max_len))
            self._cached = [[] for _ in range(m.batch)]
        return m

    # ---- health of the persistent launches behind the plain-forward path -----------------------------------------------
    def _guarded(self, what: str, body):
        """Run `body()` (forwards of self._model, returning device tensors), then check the engine's health word once the stream
        has drained. A persistent launch that gave up (EngineGaveUp: its bounded waits expired — typically another kernel held
        CUs it needs) invalidates what it wrote: the model moves to the launch path (HipModel.recover), the cached prefixes are
        forgotten (their K/V rows may hold garbage) and the call is repeated once. Nothing is returned from an invalid pass."""
        for attempt in (0, 1):
            try:
                out = body()
                if self._model is not None:
                    self._model.check_health(what, sync=True)
                return out
            except EngineGaveUp as e:
                if attempt or self._model is None:
                    raise
                logger.warning("%s: %s — repeating on the launch path", what, e)
                self._model.recover()
                self._cached = [[] for _ in self._cached]

    # ---- LanguageModel -----------------------------------------------------------------
    def generate_tokens(self, input_ids: torch.Tensor, max_new_tokens: int, temperature: float = 0.7,
                        do_sample: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._guarded("generate_tokens", lambda: self._generate_tokens(input_ids, max_new_tokens, temperature, do_sample, **kwargs))

    def _generate_tokens(self, input_ids: torch.Tensor, max_new_tokens: int, temperature: float = 0.7,
                         do_sample: bool = True, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
        """k tokens + their logits [B, k, V] (fp32), one forward per token
        (hf_wrappers.py:272-627 semantics: argmax of the last position; with do_sample the
        token is drawn from softmax(logits / T)). `row=b, rows=n` (one sequence): use cache row b of an n-row engine, so
        that several sequences of different lengths each keep their own cached prefix."""
        if input_ids.dim() == 1:
            input_ids = input_ids.unsqueeze(0)
        ids = validate_and_clamp_tokens(input_ids.long(), self.vocab_size, "generate_tokens")
        B, L = ids.shape
        row0 = int(kwargs.get("row", 0))
        if row0 or kwargs.get("rows"):
            if B != 1:
                raise ValueError("generate_tokens(row=...) takes one sequence")
            return self._generate_row(ids, max_new_tokens, row0, int(kwargs.get("rows") or row0 + 1),
                                      temperature if (do_sample and temperature and temperature > 0) else None, kwargs)
        m = self._engine(B, L + max_new_tokens + 1)
        m.set_length_hint(L + max_new_tokens + 1)    # rows of this call stay below it: persistent 1-token passes up to 1280 positions
        host = ids.cpu().tolist()
        dev = self._device
        out_ids, out_logits = [], []
        # cache the prefix (all but the last token); reuse what is already there
        for b in range(B):
            have = self._cached[b]
            common = 0
            for x, y in zip(have, host[b][:-1]):
                if x != y:
                    break
                common += 1
            todo = host[b][common : L - 1]
            if todo:
                m.forward(torch.tensor([todo], dtype=torch.int32, device=dev),
                          torch.tensor([common], dtype=torch.int32, device=dev), 0, skip_head=True, row0=b)
            self._cached[b] = list(host[b][: L - 1])
        cur = ids[:, -1:].to(dev, torch.int32).contiguous()
        pos = torch.full((B,), L - 1, dtype=torch.int32, device=dev)
        for _ in range(max_new_tokens):
            nxt, logits = m.forward(cur, pos, 0, want_logits=True, logits_dtype=torch.float32)
            last = logits[:, 0, :]
            if do_sample and temperature and temperature > 0:
                nxt = _draw(last, temperature, kwargs).to(dev, torch.int32)
            for b in range(B):
                self._cached[b].append(int(cur[b, 0]))
            out_ids.append(nxt.long())
            out_logits.append(last)
            cur = nxt.to(torch.int32).contiguous()
            pos = pos + 1
            if do_sample and self._eos_id is not None and bool((nxt == self._eos_id).all()):
                break   # transformers' generate ends a sampling call when every row has produced EOS: fewer ids come back
        if not out_ids:
            return (torch.empty((B, 0), dtype=torch.long, device=dev),
                    torch.empty((B, 0, self.vocab_size), dtype=torch.float32, device=dev))
        return torch.cat(out_ids, dim=1), torch.stack(out_logits, dim=1)

    def _generate_row(self, ids: torch.Tensor, k: int, b: int, rows: int, sample_t: Optional[float] = None, kwargs=None):
        """generate_tokens of ONE sequence in cache row b (prefix reuse per row); greedy, or drawn at temperature `sample_t`."""
        L = ids.shape[1]
        m = self._engine(rows, L + k + 1)
        m.set_length_hint(L + k + 1)
        dev = self._device
        host = ids[0].cpu().tolist()
        have = self._cached[b]
        common = 0
        for x, y in zip(have, host[:-1]):
            if x != y:
                break
            common += 1
        todo = host[common : L - 1]
        if todo:
            m.forward(torch.tensor([todo], dtype=torch.int32, device=dev), torch.tensor([common], dtype=torch.int32, device=dev), 0,
                      skip_head=True, row0=b)
        self._cached[b] = list(host[: L - 1])
        cur = ids[:, -1:].to(dev, torch.int32).contiguous()
        pos = torch.full((1,), L - 1, dtype=torch.int32, device=dev)
        out_ids, out_logits = [], []
        for _ in range(k):
            nxt, logits = m.forward(cur, pos, 0, want_logits=True, logits_dtype=torch.float32, row0=b)
            if sample_t is not None:
                nxt = _draw(logits[:, 0, :], sample_t, kwargs or {}).to(dev, torch.int32)
            self._cached[b].append(int(cur[0, 0]))
            out_ids.append(nxt.long())
            out_logits.append(logits[:, 0, :])
            cur = nxt.to(torch.int32).contiguous()
            pos = pos + 1
            if sample_t is not None and self._eos_id is not None and int(nxt[0, 0]) == self._eos_id:
                break   # (as transformers' generate: a sampling call ends at EOS)
        if not out_ids:
            return (torch.empty((1, 0), dtype=torch.long, device=dev), torch.empty((1, 0, self.vocab_size), dtype=torch.float32, device=dev))
        return torch.cat(out_ids, dim=1), torch.stack(out_logits, dim=1)

    def last_hidden_state(self, input_ids: torch.Tensor, row: int = 0, rows: int = 1) -> torch.Tensor:
        return self._guarded("last_hidden_state", lambda: self._last_hidden_state(input_ids, row, rows))

    def _last_hidden_state(self, input_ids: torch.Tensor, row: int = 0, rows: int = 1) -> torch.Tensor:
        """`outputs.hidden_states[-1][:, -1:]` of the reference's draft modes (pipeline.py:674-686): the hidden state of
        the LAST token after the final norm, fp32 [1][1][d] with bf16-representable values. The prefix is cached as in
        generate_tokens; one 1-token forward with the head skipped leaves the residual row (sd_model_hidden_rows), and
        the final norm is applied here with the roundings of the engine's own fused norm (csrc/gemv.hip prologue)."""
        ids = input_ids if input_ids.dim() == 2 else input_ids.unsqueeze(0)
        ids = validate_and_clamp_tokens(ids.long(), self.vocab_size, "last_hidden_state")
        self._generate_row(ids, 0, row, max(rows, row + 1))          # caches ids[:-1] in cache row `row`
        m, dev, L = self._model, self._device, ids.shape[1]
        m.forward(ids[:, -1:].to(dev, torch.int32).contiguous(), torch.full((1,), L - 1, dtype=torch.int32, device=dev), 0,
                  skip_head=True, row0=row)
        x = m.hidden_rows(1).float()                                  # [1][d]
        w = self.weights
        bf = lambda t: t.to(torch.bfloat16).float()
        if self.config.arch == W.ARCH_LLAMA:
            xn = bf(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.config.norm_eps))
            h = bf(xn * w.final_norm_w.float())
        else:
            mean = x.mean(-1, keepdim=True)
            var = ((x * x).mean(-1, keepdim=True) - mean * mean).clamp_min(0.0)
            h = bf((x - mean) * torch.rsqrt(var + self.config.norm_eps) * w.final_norm_w.float() + w.final_norm_b.float())
        return h.view(1, 1, -1)

    def verify_tokens(self, input_ids: torch.Tensor, draft_tokens: torch.Tensor, row: Optional[int] = None, rows: int = 1):
        return self._guarded("verify_tokens", lambda: self._verify_tokens(input_ids, draft_tokens, row, rows))

    def _verify_tokens(self, input_ids: torch.Tensor, draft_tokens: torch.Tensor, row: Optional[int] = None, rows: int = 1):
        """The K-token parallel verify as a wrapper call: one forward over
        (last, d_1..d_K) -> (argmax ids [B, K+1], logits [B, K+1, V]). `row` (one sequence): cache row to use."""
        ids = input_ids if input_ids.dim() == 2 else input_ids.unsqueeze(0)
        B, L = ids.shape
        dev = self._device
        if row is None:
            self._generate_tokens(ids, 0)  # caches the prefix
            row0 = 0
        else:
            self._generate_row(validate_and_clamp_tokens(ids.long(), self.vocab_size, "verify_tokens"), 0, row, max(rows, row + 1))
            row0 = row
        m = self._model
        toks = torch.cat([ids[:, -1:].to(dev), draft_tokens.to(dev)], dim=1).to(torch.int32).contiguous()
        pos = torch.full((B,), L - 1, dtype=torch.int32, device=dev)
        t_ids, logits = m.forward(toks, pos, 0, want_logits=True, logits_dtype=torch.float32, row0=row0)
        return t_ids.long(), logits

    def get_tokenizer_info(self) -> Dict[str, Any]:
        t = self._tokenizer
        return {
            "vocab_size": self.vocab_size,
            "pad_token_id": getattr(t, "pad_token_id", None),
            "eos_token_id": getattr(t, "eos_token_id", self.config.eos_token_id),
            "bos_token_id": getattr(t, "bos_token_id", None),
            "model_name": self._name,
            "tokenizer_type": type(t).__name__,
        }

    def encode(self, text: str) -> torch.Tensor:
        t = self._tokenizer
        ids = t.encode(text)
        return torch.tensor([list(ids)], dtype=torch.long)

    def decode(self, token_ids: Any) -> str:
        if isinstance(token_ids, torch.Tensor):
            token_ids = token_ids.flatten().tolist()
        elif token_ids and isinstance(token_ids[0], (list, tuple)):
            token_ids = list(token_ids[0])
        return self._tokenizer.decode(list(token_ids), skip_special_tokens=True)

    @property
    def device(self) -> str:
        return "cuda"

    @property
    def model_name(self) -> str:
        return self._name

    # ---- KV hooks ------------------------------------------------------------------------
    def supports_kv_append(self) -> bool:
        return True

    def get_kv_cache(self) -> Optional[KVCache]:
        m = self._model
        if m is None or not self._cached or not self._cached[0]:
            return None
        n = len(self._cached[0])
        k, v = m.kv_view()
        if m.page_len is None:   # dense rows: k [L,B,H,Lmax,D], v [L,B,H,D,Lmax] -> views
            kv = tuple((k[l, :, :, :n, :], v[l, :, :, :, :n].transpose(-1, -2)) for l in range(self.config.n_layers))
            return KVCache(past_key_values=kv, seq_len=n, dtype=torch.bfloat16, device=self._device)
        # paged engine: k [L,pages,H,P,D], v [L,pages,H,D,P]; a row's positions are spread over the pages it owns
        # (HipModel._owned, in position order) -> gather them into dense [B,H,n,D] copies
        P = m.page_len
        need = (n + P - 1) // P
        rows = []
        for b in range(len(self._cached)):
            own = m._owned[b][:need]
            if len(own) < need:
                raise RuntimeError(f"get_kv_cache: row {b} owns {len(own)} pages, {need} needed for {n} positions")
            rows.append(torch.tensor(own, dtype=torch.long, device=self._device))
        idx = torch.stack(rows, 0)                                   # [B, need]
        kv = []
        for l in range(self.config.n_layers):
            kl = k[l].index_select(0, idx.reshape(-1)).view(idx.shape[0], need, *k.shape[2:])      # [B,need,H,P,D]
            vl = v[l].index_select(0, idx.reshape(-1)).view(idx.shape[0], need, *v.shape[2:])      # [B,need,H,D,P]
            kd = kl.permute(0, 2, 1, 3, 4).reshape(idx.shape[0], k.shape[2], need * P, k.shape[4])[:, :, :n, :]
            vd = vl.permute(0, 2, 1, 4, 3).reshape(idx.shape[0], v.shape[2], need * P, v.shape[3])[:, :, :n, :]
            kv.append((kd, vd))
        return KVCache(past_key_values=tuple(kv), seq_len=n, dtype=torch.bfloat16, device=self._device)

    def get_last_generated_kv(self) -> Optional[KVCache]:
        return self._last_generated_kv

    def append_kv_cache(self, kv_chunk: Any) -> None:
        """The forward appends in place; there is nothing to concatenate afterwards
        (the reference re-copies the cache here, hf_wrappers.py:985-1029)."""
        return None

    def clear_kv_cache(self) -> None:
        self._cached = [[] for _ in self._cached]
        self._last_generated_kv = None
        m = self._model
        if m is not None and m.page_len is not None:   # paged engine: the rows' pages go back to the pool
            for b in range(m.batch):
                m.release(b)

    def optimize(self, *a, **k):
        return self


_PRESETS = {"llama-3.2-1b": W.LLAMA_3_2_1B, "llama-3.2-3b": W.LLAMA_3_2_3B, "llama-3-8b": W.LLAMA_3_8B, "gpt2": W.GPT2_SMALL,
            "distilgpt2": W.DISTILGPT2}
# hub names the reference's configs use (configs/specdec.yaml:5-6, specdec_hf.yaml) -> the architecture preset of that model
_HUB_NAMES = {"gpt2": "gpt2", "distilgpt2": "distilgpt2", "meta-llama/Llama-3.2-1B": "llama-3.2-1b", "meta-llama/Llama-3.2-3B": "llama-3.2-3b",
              "meta-llama/Meta-Llama-3-8B": "llama-3-8b", "meta-llama/Llama-3-8B": "llama-3-8b"}


def _synthetic(preset: str, device: str, embed_from=None, seed: int = 0, flip_fraction: float = 0.2) -> W.ModelWeights:
    cfg = _PRESETS[preset]
    make = W.synthetic_llama if cfg.arch == W.ARCH_LLAMA else W.synthetic_gpt2
    if embed_from is not None:
        return make(cfg, seed=seed, device=device, embed_from=embed_from, flip_fraction=flip_fraction)
    return make(cfg, seed=seed, device=device)


def _named_checkpoint(name: str) -> Optional[str]:
    """$SPECDEC_MODEL_DIR/<name> (also with the hub organisation stripped) when that directory exists."""
    import os

    root = os.environ.get("SPECDEC_MODEL_DIR")
    if not root:
        return None
    for cand in (name, name.split("/")[-1]):
        p = os.path.join(root, cand)
        if os.path.isdir(p):
            return p
    return None


def is_named_pair(base_spec: Any, draft_spec: Any) -> bool:
    """Both entries name a known architecture (a hub name of the reference's configs, or "synthetic:<preset>") and neither has a
    local checkpoint: the case in which the two synthetic models are built together."""
    def known(s):
        return isinstance(s, str) and (s in _HUB_NAMES or (s.startswith("synthetic:") and s.split(":", 1)[1] in _PRESETS)) and _named_checkpoint(s) is None
    return known(base_spec) and known(draft_spec)


def create_hip_pair(base_spec: Any, draft_spec: Any, device: str = "cuda", **kw) -> Tuple["HipLM", "HipLM"]:
    """Target and draft of a pipeline from its two config entries. When both are model NAMES without a local checkpoint (the
    reference's YAMLs name hub models: `gpt2` / `distilgpt2`), the synthetic pair is built TOGETHER — the draft shares the
    target's token tables and most of its successor structure (specdec_hip.weights), as bench.py's pair does — instead of two
    unrelated random models whose acceptance would be zero."""
    if not is_named_pair(base_spec, draft_spec):
        return create_hip_lm(base_spec, device=device, **kw), create_hip_lm(draft_spec, device=device, **kw)
    names = [s.split(":", 1)[1] if s.startswith("synthetic:") else _HUB_NAMES[s] for s in (base_spec, draft_spec)]
    for s in (base_spec, draft_spec):
        if not s.startswith("synthetic:"):
            logger.warning("model %r: nothing is downloaded by name and $SPECDEC_MODEL_DIR/%s does not exist — using architecture-exact "
                           "SYNTHETIC weights of that model's shape (specdec_hip.weights)", s, s)
    tgt = _synthetic(names[0], device, seed=0)
    same_family = _PRESETS[names[0]].arch == _PRESETS[names[1]].arch and _PRESETS[names[0]].vocab == _PRESETS[names[1]].vocab
    drf = _synthetic(names[1], device, embed_from=tgt if same_family else None, seed=1)
    return HipLM(tgt, device=device, name=str(base_spec), **kw), HipLM(drf, device=device, name=str(draft_spec), **kw)


def create_hip_lm(spec: Any, device: str = "cuda", **kw) -> HipLM:
    """`spec`: a ModelWeights, a local checkpoint directory, "synthetic:<preset>", or a hub model name the reference's configs
    use (`gpt2`, `distilgpt2`, `meta-llama/Llama-3.2-1B`, ...): $SPECDEC_MODEL_DIR/<name> when it exists, else synthetic weights
    of that architecture with a logged notice. Nothing is downloaded."""
    if isinstance(spec, W.ModelWeights):
        return HipLM(spec, device=device, **kw)
    if isinstance(spec, str) and spec.startswith("synthetic:"):
        return HipLM(_synthetic(spec.split(":", 1)[1], device), device=device, **kw)
    import os

    if isinstance(spec, str) and not os.path.isdir(spec) and spec in _HUB_NAMES:
        local = _named_checkpoint(spec)
        if local is None:
            logger.warning("model %r: nothing is downloaded by name and $SPECDEC_MODEL_DIR/%s does not exist — using architecture-exact "
                           "SYNTHETIC weights of that model's shape (specdec_hip.weights)", spec, spec)
            return HipLM(_synthetic(_HUB_NAMES[spec], device), device=device, name=spec, **kw)
        spec = local

    if isinstance(spec, str) and os.path.isdir(spec):
        mw = W.load_checkpoint_dir(spec, device=device)
        tok = None
        try:
            import transformers

            tok = transformers.AutoTokenizer.from_pretrained(spec, local_files_only=True)
        except Exception as e:  # tokenizer is optional plumbing; ids can be passed directly
            logger.warning("no tokenizer loaded from %s (%s); using IdTokenizer", spec, e)
        return HipLM(mw, tokenizer=tok, name=os.path.basename(spec.rstrip("/")), device=device, **kw)
    raise ValueError(f"cannot build a HIP model from {spec!r}: pass ModelWeights, a local checkpoint directory, "
                     f"or 'synthetic:<llama-3.2-1b|llama-3.2-3b|llama-3-8b>' (nothing is downloaded by name)")
