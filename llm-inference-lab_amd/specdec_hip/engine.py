"""Python face of the decoder-forward / step-loop entry points of the C-ABI.

Plumbing only: ctypes structs mirroring include/specdec_hip.h, torch tensors as the
owners of device memory (weights, KV caches, workspaces), torch streams as the HIP
streams. All arithmetic happens in csrc/.
"""

from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _abi
from .weights import ARCH_LLAMA, ModelConfig, ModelWeights

_vp = ctypes.c_void_p


class _LayerWeights(ctypes.Structure):
    _fields_ = [(n, _vp) for n in (
        "attn_norm_w", "attn_norm_b", "wqkv", "bqkv", "wo", "bo",
        "mlp_norm_w", "mlp_norm_b", "w_up", "b_up", "w_down", "b_down")]


class _ModelConfig(ctypes.Structure):
    _fields_ = [
        ("arch", ctypes.c_int),
        ("n_layers", ctypes.c_int), ("d_model", ctypes.c_int), ("n_heads", ctypes.c_int),
        ("n_kv_heads", ctypes.c_int), ("head_dim", ctypes.c_int), ("d_ff", ctypes.c_int),
        ("vocab", ctypes.c_int), ("max_pos", ctypes.c_int),
        ("norm_eps", ctypes.c_float),
        ("weight_dtype", ctypes.c_int),
        ("tok_emb", _vp), ("pos_emb", _vp), ("final_norm_w", _vp), ("final_norm_b", _vp),
        ("lm_head", _vp), ("rope_cos", _vp), ("rope_sin", _vp),
        ("layers", ctypes.POINTER(_LayerWeights)),
        ("packed", _vp),
    ]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(stream: Optional[torch.cuda.Stream], device) -> int:
    return (stream or torch.cuda.current_stream(device)).cuda_stream


class EngineGaveUp(_abi.HipLibraryError):
    """A persistent forward hit one of its bounded waits (another kernel held CUs it needs, or a workgroup never ran) and left.
    Nothing hangs and nothing is silently wrong: the health word says so, the caller repeats the work on the launch path."""

    def __init__(self, msg: str, status: int = 0):
        super().__init__(msg)
        self.status = status


class HipModel:
    """A decoder (Llama or GPT-2 shaped) bound to its KV cache on one GPU."""

    def __init__(self, weights: ModelWeights, batch: int, l_max: int, device: Optional[torch.device] = None,
                 weight_dtype: str = "bf16", page_len: Optional[int] = None, n_pages: Optional[int] = None):
        """weight_dtype "fp8": the engine streams an OCP e4m3 copy of the Linear weights (per-output-row
        scales, quantised on the device at load); activations and the KV cache stay bf16.
        page_len (a power of two >= 32): paged KV — the rows share a pool of `n_pages` pages (default: enough for every
        row to reach l_max) through a block table; `reserve(row, length)` / `release(row)` manage a row's pages."""
        if weight_dtype not in ("bf16", "fp8"):
            raise ValueError(f"weight_dtype={weight_dtype!r} (bf16 or fp8)")
        self.weight_dtype = weight_dtype
        self.lib = _abi.load()
        self.cfg: ModelConfig = weights.config
        dev = torch.device(device) if device is not None else weights.tok_emb.device
        if dev.type == "cuda" and dev.index is None:
            dev = weights.tok_emb.device if weights.tok_emb.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        if dev.type != "cuda":
            raise RuntimeError("HipModel needs weights on a GPU (PyTorch-ROCm device 'cuda'); there is no CPU path")
        self.device = dev
        self.weights = weights  # keeps the tensors (borrowed pointers) alive
        for name, t in weights.tensors():
            if t.device != dev:
                raise RuntimeError(f"weight {name} is on {t.device}, expected {dev}")
            want = torch.float32 if name.startswith("rope_") else torch.bfloat16
            if t.dtype != want or not t.is_contiguous():
                raise TypeError(f"weight {name}: need contiguous {want}, got {t.dtype} contiguous={t.is_contiguous()}")
        c = self.cfg
        if not (1 <= c.n_kv_heads <= c.n_heads <= 255 and c.head_dim in (32, 64, 128)):
            # (the kernels' argument blocks carry these in 8-bit fields)
            raise ValueError(f"unsupported attention geometry: {c.n_heads} heads / {c.n_kv_heads} kv heads / head_dim {c.head_dim}")
        self._layers = (_LayerWeights * c.n_layers)()
        for i, l in enumerate(weights.layers):
            for f, _ in _LayerWeights._fields_:
                setattr(self._layers[i], f, _ptr(getattr(l, f)))
        mc = _ModelConfig(
            arch=c.arch, n_layers=c.n_layers, d_model=c.d_model, n_heads=c.n_heads, n_kv_heads=c.n_kv_heads,
            head_dim=c.head_dim, d_ff=c.d_ff, vocab=c.vocab, max_pos=c.max_pos, norm_eps=c.norm_eps,
            weight_dtype=_abi.SD_FP8_E4M3 if weight_dtype == "fp8" else _abi.SD_BF16,
            tok_emb=_ptr(weights.tok_emb), pos_emb=_ptr(weights.pos_emb),
            final_norm_w=_ptr(weights.final_norm_w), final_norm_b=_ptr(weights.final_norm_b),
            lm_head=_ptr(weights.lm_head), rope_cos=_ptr(weights.rope_cos), rope_sin=_ptr(weights.rope_sin),
            layers=self._layers,
        )
        # the engine's packed copy of the Linear weights (one per ModelWeights, shared by every
        # engine instance built over it)
        key = "_packed" if weight_dtype == "bf16" else "_packed_fp8"
        packed = weights.meta.get(key)
        if packed is None and (weight_dtype == "fp8" or not os.environ.get("SPECDEC_NO_PACK")):
            nbytes = self.lib.sd_packed_bytes(ctypes.byref(mc))
            with torch.cuda.device(dev):
                packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                _abi.check(self.lib.sd_pack_weights(ctypes.byref(mc), packed.data_ptr(), nbytes,
                                                    torch.cuda.current_stream(dev).cuda_stream), "sd_pack_weights")
            weights.meta[key] = packed
        self._packed = packed
        mc.packed = _ptr(packed)
        handle = _vp()
        _abi.check(self.lib.sd_model_create(ctypes.byref(mc), ctypes.byref(handle)), "sd_model_create")
        self.handle = handle
        self.batch, self.l_max = int(batch), (int(l_max) + 31) // 32 * 32  # whole 32-key blocks
        self.page_len = None
        if page_len is not None or os.environ.get("SPECDEC_PAGED_KV"):
            self._bind_paged(int(page_len or os.environ["SPECDEC_PAGED_KV"]), n_pages)
            return
        kv_bytes = self.lib.sd_model_kv_bytes(self.handle, self.batch, self.l_max)
        with torch.cuda.device(dev):
            # [n_layers][B][Hkv][Lmax][D] bf16 — sized for 288 GB of HBM: no paging, no realign copies
            self.k_cache = torch.zeros(kv_bytes // 2, dtype=torch.bfloat16, device=dev)
            self.v_cache = torch.zeros(kv_bytes // 2, dtype=torch.bfloat16, device=dev)
            self.workspace = torch.empty(self.lib.sd_model_workspace_bytes(self.handle), dtype=torch.uint8, device=dev)
            _abi.check(self.lib.sd_model_bind(self.handle, self.k_cache.data_ptr(), self.v_cache.data_ptr(),
                                              self.batch, self.l_max, self.workspace.data_ptr(),
                                              self.workspace.numel()), "sd_model_bind")

    # ---- paged KV -----------------------------------------------------------------------------------------------
    def _bind_paged(self, page_len: int, n_pages: Optional[int]) -> None:
        P = int(page_len)
        if P < 32 or P & (P - 1):
            raise ValueError(f"page_len={P}: a power of two >= 32")
        self.page_len = P
        self.max_pages = (self.l_max + P - 1) // P
        self.l_max = self.max_pages * P
        self.n_pages = int(n_pages) if n_pages is not None else self.batch * self.max_pages
        dev = self.device
        pool = self.lib.sd_model_kv_pool_bytes(self.handle, self.n_pages, P)
        with torch.cuda.device(dev):
            self.k_cache = torch.zeros(pool // 2, dtype=torch.bfloat16, device=dev)
            self.v_cache = torch.zeros(pool // 2, dtype=torch.bfloat16, device=dev)
            # every entry names a valid page at all times (unreached entries are never read, but a clamped load may touch them)
            self.block_table = torch.zeros(self.batch, self.max_pages, dtype=torch.int32, device=dev)
            self.workspace = torch.empty(self.lib.sd_model_workspace_bytes(self.handle), dtype=torch.uint8, device=dev)
            _abi.check(self.lib.sd_model_bind_paged(self.handle, self.k_cache.data_ptr(), self.v_cache.data_ptr(), self.n_pages, P,
                                                    self.block_table.data_ptr(), self.max_pages, self.batch,
                                                    self.workspace.data_ptr(), self.workspace.numel()), "sd_model_bind_paged")
        self._free = list(range(self.n_pages - 1, -1, -1))       # stack of free page indices
        self._owned: List[List[int]] = [[] for _ in range(self.batch)]

    def reserve(self, row: int, length: int, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Make positions [0, length) of `row` addressable (paged engines; a no-op on dense ones). New table entries are
        written on `stream` (default: the current stream) — the forward that uses them must be ordered after it."""
        if self.page_len is None:
            return
        need = (min(int(length), self.l_max) + self.page_len - 1) // self.page_len
        own = self._owned[row]
        if need <= len(own):
            return
        if need - len(own) > len(self._free):
            raise RuntimeError(f"KV page pool exhausted: row {row} needs {need - len(own)} more pages, {len(self._free)} free of {self.n_pages}")
        first = len(own)
        while len(own) < need:
            own.append(self._free.pop())
        new = torch.tensor(own[first:], dtype=torch.int32)
        with torch.cuda.device(self.device), torch.cuda.stream(stream or torch.cuda.current_stream(self.device)):
            self.block_table[row, first:need].copy_(new.to(self.device, non_blocking=False))

    def release(self, row: int) -> None:
        """Return the row's pages to the pool (its sequence is finished; the next reserve() starts from nothing)."""
        if self.page_len is None:
            return
        self._free.extend(reversed(self._owned[row]))
        self._owned[row] = []

    def pages_in_use(self) -> int:
        return 0 if self.page_len is None else self.n_pages - len(self._free)

    def kv_view(self):
        c = self.cfg
        if self.page_len is not None:
            return (self.k_cache.view(c.n_layers, self.n_pages, c.n_kv_heads, self.page_len, c.head_dim),
                    self.v_cache.view(c.n_layers, self.n_pages, c.n_kv_heads, c.head_dim, self.page_len))
        k_shape = (c.n_layers, self.batch, c.n_kv_heads, self.l_max, c.head_dim)
        v_shape = (c.n_layers, self.batch, c.n_kv_heads, c.head_dim, self.l_max)  # V is kept transposed
        return self.k_cache.view(k_shape), self.v_cache.view(v_shape)

    def forward(self, tokens: torch.Tensor, pos_base: torch.Tensor, pos_off: int = 0,
                want_ids: bool = True, want_logits: bool = False, logits_dtype=torch.float32,
                skip_head: bool = False, stream: Optional[torch.cuda.Stream] = None, row0: int = 0):
        """tokens int32 [B][M] on the device, pos_base int32 [B] for rows [row0, row0+B) of the
        bound batch. Appends to the cache in place."""
        assert tokens.dtype == torch.int32 and tokens.dim() == 2 and tokens.device == self.device
        assert pos_base.dtype == torch.int32 and pos_base.shape == (tokens.shape[0],) and pos_base.device == self.device
        B, M = tokens.shape
        tokens = tokens.contiguous()
        if self.health():                 # an earlier persistent pass gave up: do not pile more work on invalid rows
            self.check_health("sd_model_forward")
        if self.page_len is not None:     # paged KV: the positions this pass writes must have pages (host-known here at the cost
            for i, p0 in enumerate(pos_base.tolist()):   # of one read-back; the captured step loop reserves ahead instead)
                self.reserve(row0 + i, p0 + int(pos_off) + M, stream=stream)
        ids = torch.empty((B, M), dtype=torch.int32, device=self.device) if (want_ids and not skip_head) else None
        logits = None
        if want_logits and not skip_head:
            logits = torch.empty((B, M, self.cfg.vocab), dtype=logits_dtype, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.sd_model_forward(
                self.handle, tokens.data_ptr(), M, pos_base.data_ptr(), int(pos_off), int(row0), B, M,
                _ptr(ids), M, _ptr(logits),
                _abi.SD_F32 if logits_dtype == torch.float32 else _abi.SD_BF16,
                1 if skip_head else 0, _stream(stream, self.device))
        _abi.check(rc, "sd_model_forward")
        return ids, logits

    def hidden_rows(self, n: int, row0: int = 0, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """bf16 [n][d_model]: the residual-stream rows (before the final norm) of the last forward pass (sd_model_hidden_rows)."""
        out = torch.empty((n, self.cfg.d_model), dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_model_hidden_rows(self.handle, int(row0), int(n), out.data_ptr(), _stream(stream, self.device)),
                       "sd_model_hidden_rows")
        return out

    @property
    def pass_tokens(self) -> int:
        """Tokens per forward pass (64 with the multi-token kernel, else 9)."""
        return int(self.lib.sd_model_pass_tokens(self.handle))

    PROBE_O, PROBE_GATE_UP, PROBE_DOWN, PROBE_LM_HEAD = 1, 2, 3, 4

    def probe_gemv(self, which: int, T: int, iters: int = 200, stream: Optional[torch.cuda.Stream] = None):
        """(average launch duration in microseconds, algorithmic bytes per launch) of one of the
        forward's GEMVs, timed with HIP events on the launch stream (sd_model_probe_gemv)."""
        usec, nbytes = ctypes.c_float(0), ctypes.c_double(0)
        with torch.cuda.device(self.device):
            rc = self.lib.sd_model_probe_gemv(self.handle, int(which), int(T), int(iters), _stream(stream, self.device),
                                              ctypes.byref(usec), ctypes.byref(nbytes))
        _abi.check(rc, "sd_model_probe_gemv")
        return float(usec.value), float(nbytes.value)

    @property
    def persist_tokens(self) -> int:
        """Tokens per pass that run as ONE persistent launch (sd_model_persist_tokens); 0 = launch-per-operator only."""
        return int(self.lib.sd_model_persist_tokens(self.handle))

    def persist_active(self, tokens: int = 1) -> bool:
        """Would a pass of `tokens` tokens of one row run as the persistent launch right now (token limit, length hint, paging)?"""
        return bool(self.lib.sd_model_persist_active(self.handle, int(tokens)))

    def set_persist_tokens(self, max_tokens: int) -> None:
        """Tokens per pass the persistent launch takes from now on (0: launch path only; sd_model_set_persist_tokens). Loops that
        captured a step over this model must be invalidated (HipSpecDec.invalidate)."""
        _abi.check(self.lib.sd_model_set_persist_tokens(self.handle, int(max_tokens)), "sd_model_set_persist_tokens")

    def set_length_hint(self, max_len: Optional[int]) -> None:
        """The caller's bound on the current length of the rows the coming passes touch (None: the cache size). The persistent
        launch serves rows of up to 1280 positions (sd_model_set_length_hint)."""
        _abi.check(self.lib.sd_model_set_length_hint(self.handle, int(max_len or 0)), "sd_model_set_length_hint")

    def health(self) -> int:
        """The persistent launches' health word as the pinned host copy holds it — no copy, no synchronisation: valid for every
        pass the caller has already synchronised with (it has read the pass's ids / logits). 0 = all completed."""
        w = getattr(self, "_status_word", None)
        if w is None:
            ptr = self.lib.sd_model_status_word(self.handle)
            if not ptr:
                return 0
            w = self._status_word = np.ctypeslib.as_array(ptr, shape=(1,))
        return int(w[0])

    def check_health(self, what: str = "forward", sync: bool = False, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Raise EngineGaveUp when a persistent launch of this model gave up (its outputs, and those of every pass after it, are
        invalid). sync: drain `stream` first (for a caller that has not yet read anything of its last pass)."""
        if sync:
            (stream or torch.cuda.current_stream(self.device)).synchronize()
        st = self.health()
        if st:
            raise EngineGaveUp(f"{what}: a persistent forward gave up (status {st:#x}: see sd_model_engine_status); the outputs since "
                               "then are invalid — HipModel.recover() clears the word and continues on the launch path", st)

    def recover(self, stream: Optional[torch.cuda.Stream] = None) -> int:
        """After a give-up: drain the stream, clear the health word (device and host), move the launch counter past the failed
        launch, and switch this model to the launch path (persistent passes stay off until the model is bound again or
        set_persist_tokens re-enables them). The passes since the failure must be repeated. Returns the status that was cleared."""
        st = self.engine_status(stream)
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_model_engine_status_clear(self.handle, _stream(stream, self.device)), "sd_model_engine_status_clear")
        self.set_persist_tokens(0)
        return st

    def clear_engine_status(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        """sd_model_engine_status_clear alone: the persistent path stays on (tests of the give-up path)."""
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_model_engine_status_clear(self.handle, _stream(stream, self.device)), "sd_model_engine_status_clear")

    def engine_status(self, stream: Optional[torch.cuda.Stream] = None) -> int:
        """0 = every persistent launch of this model completed (sd_model_engine_status; synchronises the stream)."""
        st = ctypes.c_uint32(0)
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_model_engine_status(self.handle, ctypes.byref(st), _stream(stream, self.device)), "sd_model_engine_status")
        return int(st.value)

    DEBUG_X, DEBUG_Q, DEBUG_ATTN, DEBUG_ACT = 0, 1, 2, 3

    def debug_rows(self, which: int, n: int, row0: int = 0, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """bf16 [n][width]: rows of a workspace buffer of the last pass, last layer (sd_model_debug_rows)."""
        c = self.cfg
        width = {0: c.d_model, 1: c.n_heads * c.head_dim, 2: c.n_heads * c.head_dim, 3: c.d_ff}[int(which)]
        out = torch.empty((n, width), dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_model_debug_rows(self.handle, int(which), int(row0), int(n), out.data_ptr(), _stream(stream, self.device)),
                       "sd_model_debug_rows")
        return out

    def probe_forward(self, M: int = 1, iters: int = 50, skip_head: bool = False, timeline: bool = False,
                      stream: Optional[torch.cuda.Stream] = None, pos0: int = 0):
        """(average microseconds per forward of M tokens, bytes of weights per forward, timeline or None): sd_model_probe_forward.
        timeline: uint64 [256][12 * n_ops + 4] stamps (100 MHz) of one persistent forward — per op: gather start, input staged,
        op done, attention done, third consumer start, its MFMA end, leader MFMA end, loader done issuing; then (realtime,
        shader clock) at start and end."""
        usec, nbytes = ctypes.c_float(0), ctypes.c_double(0)
        n_ops = 4 * self.cfg.n_layers + (0 if skip_head else 1)
        tl = np.zeros(256 * (12 * n_ops + 4), dtype=np.uint64) if timeline else None
        with torch.cuda.device(self.device):
            rc = self.lib.sd_model_probe_forward(self.handle, int(M), int(pos0), int(iters), 1 if skip_head else 0, _stream(stream, self.device),
                                                 ctypes.byref(usec), ctypes.byref(nbytes),
                                                 tl.ctypes.data if tl is not None else None, tl.size if tl is not None else 0)
        _abi.check(rc, "sd_model_probe_forward")
        return float(usec.value), float(nbytes.value), tl

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sd_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StepRecord:
    """Host view of one completed step (pinned memory written by the device)."""

    __slots__ = ("accept_len", "n_new", "cur_len", "new_tokens", "draft_tokens", "target_ids", "k_row", "engine_status")

    def __init__(self, arr: np.ndarray, K: int):
        self.accept_len = arr[:, 0].copy()
        self.n_new = arr[:, 1].copy()
        self.cur_len = arr[:, 2].copy()
        self.new_tokens = arr[:, 3:4 + K].copy()
        self.draft_tokens = arr[:, 4 + K:4 + 2 * K].copy()
        self.target_ids = arr[:, 4 + 2 * K:5 + 3 * K].copy()
        self.k_row = arr[:, 5 + 3 * K].copy()      # proposals that counted per row (per-row adaptive K), else K
        # health word of the models' persistent launches (sd_model_engine_status): a launch that gave up leaves garbage
        self.engine_status = int(arr[0, 6 + 3 * K]) if arr.shape[1] > 6 + 3 * K else 0
        if self.engine_status:
            raise EngineGaveUp(f"a persistent forward gave up (status {self.engine_status:#x}: see sd_model_engine_status); "
                               "the step's outputs are invalid", self.engine_status)


class HipSpecDec:
    """The draft-then-verify step loop on the device (sd_specdec_*)."""

    EMIT_BONUS, EMIT_DRAFT = 0, 1

    def __init__(self, draft: Optional[HipModel], target: HipModel, batch: int, k: int, emit_mode: int = 0):
        """draft=None: self-draft mode (Medusa-lite with heads tied to the lm_head, see sd_specdec_create)."""
        assert draft is None or draft.device == target.device
        self.lib = _abi.load()
        self.draft, self.target = draft, target
        self.device = target.device
        self.B, self.K = int(batch), int(k)
        handle = _vp()
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_specdec_create(draft.handle if draft is not None else None, target.handle, self.B, self.K, int(emit_mode),
                                                  ctypes.byref(handle)), "sd_specdec_create")
        self.handle = handle
        # hipGraph capture is not allowed on the legacy default stream: the loop owns two
        # non-default streams (verify/target and draft), ordered by events inside the step
        with torch.cuda.device(self.device):
            self.stream_t = torch.cuda.Stream(self.device)
            self.stream_d = torch.cuda.Stream(self.device)
        self.rec_ints = self.lib.sd_specdec_record_ints(self.handle)
        ptr = self.lib.sd_specdec_record(self.handle)
        self._record = np.ctypeslib.as_array(ptr, shape=(2, self.B, self.rec_ints))   # slot = launch index & 1

    def join_current_stream(self):
        """Order the loop's streams after work already enqueued on torch's current
        stream (prefill, weight uploads)."""
        self.stream_t.wait_stream(torch.cuda.current_stream(self.device))

    def set_row(self, b: int, seq_len: int, prev_tok: int, last_tok: int, active: bool = True):
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_specdec_set_row(self.handle, b, int(seq_len), int(prev_tok), int(last_tok),
                                                   1 if active else 0, self.stream_t.cuda_stream), "sd_specdec_set_row")

    def set_adaptive(self, enable: bool, initial_k: int = 4, min_k: int = 1, max_k: Optional[int] = None, step_size: int = 1,
                     target_acceptance_rate: float = 0.7):
        """Per-row adaptive K inside the captured step (sd_specdec_set_adaptive): the loop's K is the shape (max_k),
        the device moves every row's own k by the reference's controller rule after each accept scan."""
        max_k = self.K if max_k is None else int(max_k)
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_specdec_set_adaptive(self.handle, 1 if enable else 0, int(initial_k), int(min_k), max_k, int(step_size),
                                                        float(target_acceptance_rate), self.stream_t.cuda_stream), "sd_specdec_set_adaptive")
        self.adaptive = bool(enable)

    def set_adaptive_row(self, b: int, k: int, accepted: int = 0, proposed: int = 0, history: Optional[Sequence[float]] = None):
        """Row b's controller state (new sequence: defaults; repair: the host mirror's counters and last <= 4 rates)."""
        hist = None
        n = 1
        if history is not None:
            h = [float(x) for x in history][-4:]
            n = len(h)
            hist = (ctypes.c_double * 4)(*(h + [0.0] * (4 - n)))
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_specdec_set_adaptive_row(self.handle, int(b), int(k), int(accepted), int(proposed), n, hist,
                                                            self.stream_t.cuda_stream), "sd_specdec_set_adaptive_row")

    def set_medusa(self, heads: torch.Tensor, weight_dtype: str = "bf16"):
        """Persistent Medusa heads for a loop created with draft=None (sd_specdec_set_medusa). heads: bf16 [K][V][d];
        each is packed (and quantised for fp8) once, like the lm_head."""
        K, V, d = heads.shape
        if K != self.K or heads.dtype != torch.bfloat16 or heads.device != self.device:
            raise ValueError(f"medusa heads must be bf16 [{self.K}][V][d] on {self.device}")
        wd = _abi.SD_FP8_E4M3 if weight_dtype == "fp8" else _abi.SD_BF16
        nbytes = self.lib.sd_packed_head_bytes(V, d, wd)
        stride = (nbytes + 255) // 256 * 256      # heads at a constant stride in ONE buffer: the engine evaluates them in one launch
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device)
            self._heads_packed = torch.empty(K * stride, dtype=torch.uint8, device=self.device)
            base = self._heads_packed.data_ptr()
            for i in range(K):
                _abi.check(self.lib.sd_pack_head(heads[i].contiguous().data_ptr(), V, d, wd, base + i * stride, nbytes, st.cuda_stream), "sd_pack_head")
            st.synchronize()
            arr = (ctypes.c_void_p * K)(*[base + i * stride for i in range(K)])
            _abi.check(self.lib.sd_specdec_set_medusa(self.handle, K, arr, wd), "sd_specdec_set_medusa")

    def set_eagle(self, alpha: float = 0.7, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """EAGLE-lite drafting for a loop created with draft=None (sd_specdec_set_eagle): K extrapolated hidden rows
        per step, scored by one lm_head launch. `workspace` (uint8, shared between loops of different K so that the
        per-row state survives a change of K) is allocated for K = 8 when not given; returns it."""
        d = self.target.cfg.d_model
        if workspace is None:
            workspace = torch.zeros(int(self.lib.sd_specdec_eagle_bytes(self.B, 8, d)), dtype=torch.uint8, device=self.device)
        self._eagle_ws = workspace
        _abi.check(self.lib.sd_specdec_set_eagle(self.handle, ctypes.c_float(alpha), workspace.data_ptr(), workspace.numel()),
                   "sd_specdec_set_eagle")
        return workspace

    def reset_eagle(self):
        """Forget the extrapolation state of every row (start of new sequences); ordered on the loop's target stream."""
        _abi.check(self.lib.sd_specdec_reset_eagle(self.handle, self.stream_t.cuda_stream), "sd_specdec_reset_eagle")

    def set_sampling(self, enable: bool, temperature: float = 1.0, top_k: Optional[int] = None,
                     top_p: Optional[float] = None, seed: int = 0, stream_ids: Optional[Sequence[int]] = None,
                     draw_counts: Optional[Sequence[int]] = None):
        """Sampled bonus token inside the step (sd_specdec_set_sampling). Draw counters start at
        `draw_counts` (default 0: a new run)."""
        with torch.cuda.device(self.device):
            if not enable:
                _abi.check(self.lib.sd_specdec_set_sampling(self.handle, 0, 1.0, 0, 1.0, 0, None, 0, None, None),
                           "sd_specdec_set_sampling")
                self.sampling = False
                return
            V = self.target.weights.config.vocab
            if getattr(self, "_logits", None) is None:
                self._logits = torch.empty((self.B, self.K + 1, V), dtype=torch.bfloat16, device=self.device)
            self._draw = torch.tensor(list(draw_counts) if draw_counts is not None else [0] * self.B, dtype=torch.int32,
                                      device=self.device)
            sid = list(stream_ids) if stream_ids is not None else list(range(self.B))
            self._stream_ids = torch.tensor(sid, dtype=torch.int32, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()
            _abi.check(self.lib.sd_specdec_set_sampling(
                self.handle, 1, float(temperature), int(top_k) if top_k else 0, 1.0 if top_p is None else float(top_p),
                int(seed) & (2 ** 64 - 1), self._logits.data_ptr(), self._logits.numel() * 2, self._draw.data_ptr(),
                self._stream_ids.data_ptr()), "sd_specdec_set_sampling")
            self.sampling = True

    @property
    def step_logits(self) -> torch.Tensor:
        """[B][K+1][V] bf16 logits of the last verify forward (sampling mode only)."""
        return self._logits

    def step(self, use_graph: bool = True, two_streams: bool = True):
        if os.environ.get("SPECDEC_ONE_STREAM"):   # experiment knob: draft and verify on one stream (no fork/join events)
            two_streams = False
        with torch.cuda.device(self.device):
            sd = self.stream_d.cuda_stream if two_streams else None
            _abi.check(self.lib.sd_specdec_step(self.handle, self.stream_t.cuda_stream, sd,
                                                1 if use_graph else 0), "sd_specdec_step")

    def sync(self) -> StepRecord:
        """Drain the loop's stream; the record of the LAST launched step."""
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_specdec_sync(self.handle, self.stream_t.cuda_stream), "sd_specdec_sync")
        last = max(int(self.lib.sd_specdec_launches(self.handle)) - 1, 0)
        return StepRecord(self._record[last & 1], self.K)

    @property
    def launches(self) -> int:
        return int(self.lib.sd_specdec_launches(self.handle))

    def invalidate(self) -> None:
        """Drop the captured step (a model's path changed: set_persist_tokens / set_length_hint / a re-bind). The caller has
        drained the loop; the next step runs eagerly, the one after it captures again (sd_specdec_invalidate)."""
        _abi.check(self.lib.sd_specdec_invalidate(self.handle), "sd_specdec_invalidate")

    def drain(self) -> None:
        """Wait for everything launched, without reading a record (recovery path: the records may be invalid)."""
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_specdec_sync(self.handle, self.stream_t.cuda_stream), "sd_specdec_sync")
            _abi.check(self.lib.sd_specdec_sync(self.handle, self.stream_d.cuda_stream), "sd_specdec_sync")

    def wait(self, launch_index: int) -> StepRecord:
        """Wait for one of the last two launched steps (the other may still be running) and return its record."""
        with torch.cuda.device(self.device):
            _abi.check(self.lib.sd_specdec_wait(self.handle, int(launch_index)), "sd_specdec_wait")
        return StepRecord(self._record[launch_index & 1], self.K)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sd_specdec_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
