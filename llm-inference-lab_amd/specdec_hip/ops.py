"""Tensor-level wrappers over the C-ABI ops (device memory + stream plumbing only).

These are the functions the kernel registry registers for device "cuda" (the
native device string of PyTorch-ROCm). Argument meaning and assertion behaviour
follow the reference contract in src/kernels/reference.py:13-159; the work is done
by the HIP kernels in csrc/. CPU tensors are rejected: there is no CPU fallback.
"""

from __future__ import annotations

import weakref
from typing import Dict, Optional, Tuple

import torch

from . import _abi

_TORCH_TO_SD = {
    torch.float32: _abi.SD_F32,
    torch.float16: _abi.SD_F16,
    torch.bfloat16: _abi.SD_BF16,
    torch.int32: _abi.SD_I32,
    torch.int64: _abi.SD_I64,
    torch.uint8: _abi.SD_U8,
}


def sd_dtype(t: torch.dtype) -> int:
    try:
        return _TORCH_TO_SD[t]
    except KeyError:
        raise TypeError(f"dtype {t} is not supported by the HIP path") from None


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_device(name: str, *tensors: torch.Tensor) -> torch.device:
    dev = tensors[0].device
    for t in tensors:
        if t.device.type != "cuda":
            raise RuntimeError(
                f"{name}: the HIP path needs device tensors (got {t.device}); "
                "there is no CPU fallback in this build"
            )
        if t.device != dev:
            raise RuntimeError(f"{name}: tensors on different devices ({dev} vs {t.device})")
    return dev


# --------------------------------------------------------------------------- verify
def verify_prefix_workspace(B: int, K: int, V: int, device) -> torch.Tensor:
    """Scratch for verify_prefix_hip(..., workspace=): allocate once, reuse for every call of that shape or smaller."""
    return torch.empty(max(_abi.load().sd_verify_prefix_workspace(B, K, V), 1), dtype=torch.uint8, device=device)


def verify_prefix_hip(
    logits: torch.Tensor, draft_ids: torch.Tensor, return_pred: bool = False,
    out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, workspace: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, ...]:
    """accept_len[B] int32, accepted_mask[B,K] uint8 (reference.py:13-56) on gfx950.

    With return_pred=True also returns the argmax ids [B,K] int32. `out` = (accept_len, mask) and `workspace`
    (verify_prefix_workspace) make the call allocation-free, which is what a captured graph needs.
    """
    assert logits.dim() == 3, "logits must be 3D tensor [B][K][V]"
    assert draft_ids.dim() == 2, "draft_ids must be 2D tensor [B][K]"
    assert logits.size(0) == draft_ids.size(0), "Batch size mismatch"
    assert logits.size(1) == draft_ids.size(1), "K dimension mismatch"
    dev = _require_device("verify_prefix", logits, draft_ids)
    if logits.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise TypeError(f"verify_prefix: logits dtype {logits.dtype} not supported (f32/f16/bf16)")
    if draft_ids.dtype not in (torch.int32, torch.int64):
        if draft_ids.dtype.is_floating_point or draft_ids.dtype == torch.bool:
            raise TypeError(f"verify_prefix: draft_ids dtype {draft_ids.dtype} is not an integer type")
        draft_ids = draft_ids.to(torch.int64)
    B, K, V = logits.shape
    if V > 0 and logits.stride(2) != 1:
        logits = logits.contiguous()
    ids = draft_ids.contiguous()
    if out is not None:
        accept_len, mask = out
        assert accept_len.dtype == torch.int32 and accept_len.shape == (B,) and accept_len.is_contiguous() and accept_len.device == dev
        assert mask.dtype == torch.uint8 and mask.shape == (B, K) and mask.is_contiguous() and mask.device == dev
    else:
        accept_len = torch.empty(B, dtype=torch.int32, device=dev)
        mask = torch.empty((B, K), dtype=torch.uint8, device=dev)
    pred = torch.empty((B, K), dtype=torch.int32, device=dev) if return_pred else None
    if B == 0:
        return (accept_len, mask, pred) if return_pred else (accept_len, mask)
    lib = _abi.load()
    ws_bytes = lib.sd_verify_prefix_workspace(B, K, V)
    ws = workspace if workspace is not None else torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    assert ws.device == dev and ws.numel() >= ws_bytes, "verify_prefix: workspace too small"
    with torch.cuda.device(dev):
        rc = lib.sd_verify_prefix(
            logits.data_ptr(), sd_dtype(logits.dtype), ids.data_ptr(), sd_dtype(ids.dtype),
            accept_len.data_ptr(), mask.data_ptr(), pred.data_ptr() if pred is not None else None,
            B, K, V, logits.stride(0), logits.stride(1),
            ws.data_ptr(), ws.numel(), _stream_ptr(dev),
        )
    _abi.check(rc, "sd_verify_prefix")
    return (accept_len, mask, pred) if return_pred else (accept_len, mask)


# ------------------------------------------------------------------------ kv append
class _Arena:
    """A [B,H,cap,D] pair of buffers that kv_append_hip grows in place.

    It lives exactly as long as the newest view handed out of it: each hand-out
    bumps `gen` and arms a finalizer on the returned K view that retires the arena
    if no newer view has replaced it.
    """

    __slots__ = ("k", "v", "frontier", "gen")

    def __init__(self, k: torch.Tensor, v: torch.Tensor, frontier: int):
        self.k, self.v, self.frontier, self.gen = k, v, frontier, 0


# storage pointer of an arena's K buffer -> arena
_arenas: Dict[int, _Arena] = {}


def _retire(key: int, gen: int) -> None:
    ar = _arenas.get(key)
    if ar is not None and ar.gen == gen:
        del _arenas[key]


def _hand_out(key: int, ar: _Arena, rows: int) -> Tuple[torch.Tensor, torch.Tensor]:
    ar.frontier = rows
    ar.gen += 1
    vk, vv = ar.k[:, :, :rows], ar.v[:, :, :rows]
    weakref.finalize(vk, _retire, key, ar.gen)
    return vk, vv


def _find_arena(base_k: torch.Tensor, base_v: torch.Tensor) -> Optional[_Arena]:
    ar = _arenas.get(base_k.untyped_storage().data_ptr())
    if ar is None:
        return None
    B, H, L, D = base_k.shape
    cap = ar.k.shape[2]
    ok = (
        ar.k.dtype == base_k.dtype
        and ar.k.shape[0] == B and ar.k.shape[1] == H and ar.k.shape[3] == D
        and base_k.data_ptr() == ar.k.data_ptr() and base_v.data_ptr() == ar.v.data_ptr()
        and base_k.stride() == ar.k.stride() and base_v.stride() == ar.v.stride()
        and L == ar.frontier and L <= cap
    )
    return ar if ok else None


def _check_kv_shapes(base_k, base_v, new_k, new_v):
    assert base_k.dim() == 4, "base_k must be 4D tensor [B][H][L][D]"
    assert base_v.dim() == 4, "base_v must be 4D tensor [B][H][L][D]"
    assert new_k.dim() == 4, "new_k must be 4D tensor [B][H][K][D]"
    assert new_v.dim() == 4, "new_v must be 4D tensor [B][H][K][D]"
    assert base_k.shape[0] == new_k.shape[0], "Batch size mismatch"
    assert base_k.shape[1] == new_k.shape[1], "Num heads mismatch"
    assert base_k.shape[3] == new_k.shape[3], "Head dim mismatch"


def _row_major(t: torch.Tensor) -> torch.Tensor:
    """Rows of D contiguous elements at stride D (strided b/h allowed)."""
    if t.shape[3] == 0 or t.shape[2] == 0:
        return t
    if t.stride(3) == 1 and (t.stride(2) == t.shape[3] or t.shape[2] == 1):
        return t
    return t.contiguous()


def kv_append_hip(
    base_k: torch.Tensor, base_v: torch.Tensor, new_k: torch.Tensor, new_v: torch.Tensor
) -> Tuple[torch.Tensor, torch.Tensor]:
    """[B,H,L,D] + [B,H,K,D] -> [B,H,L+K,D] (reference.py:59-93) on gfx950.

    The result is a view of a runtime-owned [B,H,cap,D] arena. When `base_*` is
    the result of the previous call and has not been appended to since, the K new
    rows are written in place behind it (sd_kv_append: 2*B*H*K*D elements moved,
    not the reference's re-copy of all L rows) and a longer view is returned; the
    rows [0,L) that `base_*` views are never modified. Any other base is copied
    once into a fresh arena (sd_kv_concat).
    """
    _check_kv_shapes(base_k, base_v, new_k, new_v)
    dev = _require_device("kv_append", base_k, base_v, new_k, new_v)
    if not (base_k.dtype == base_v.dtype == new_k.dtype == new_v.dtype):
        raise TypeError("kv_append: base/new dtypes differ")
    esize = base_k.element_size()
    if esize not in (2, 4):
        raise TypeError(f"kv_append: element size {esize} not supported")
    assert base_v.shape == base_k.shape and new_v.shape == new_k.shape, "K/V shape mismatch"
    B, H, L, D = base_k.shape
    K = new_k.shape[2]
    lib = _abi.load()
    new_k = new_k.contiguous()
    new_v = new_v.contiguous()

    ar = _find_arena(base_k, base_v)
    if ar is not None and L + K <= ar.k.shape[2]:
        with torch.cuda.device(dev):
            rc = lib.sd_kv_append(
                ar.k.data_ptr(), ar.v.data_ptr(), new_k.data_ptr(), new_v.data_ptr(),
                None, L, esize, B, H, ar.k.shape[2], K, D, _stream_ptr(dev),
            )
        _abi.check(rc, "sd_kv_append")
        return _hand_out(ar.k.untyped_storage().data_ptr(), ar, L + K)

    # fresh arena with head-room: amortised O(K) per later append
    cap = max(L + K, 1)
    cap = max(cap + max(64, cap // 2), 2 * K)
    buf_k = torch.empty((B, H, cap, D), dtype=base_k.dtype, device=dev)
    buf_v = torch.empty((B, H, cap, D), dtype=base_k.dtype, device=dev)
    bk, bv = _row_major(base_k), _row_major(base_v)
    if bv.stride()[:2] != bk.stride()[:2]:
        bk, bv = bk.contiguous(), bv.contiguous()
    if B * H * D > 0 and L + K > 0:
        with torch.cuda.device(dev):
            rc = lib.sd_kv_concat(
                buf_k.data_ptr(), buf_v.data_ptr(), bk.data_ptr(), bv.data_ptr(),
                new_k.data_ptr(), new_v.data_ptr(), esize, B, H, L, K, D, cap,
                bk.stride(0) if L else 0, bk.stride(1) if L else 0, _stream_ptr(dev),
            )
        _abi.check(rc, "sd_kv_concat")
    ar = _Arena(buf_k, buf_v, L + K)
    key = buf_k.untyped_storage().data_ptr()
    _arenas[key] = ar
    return _hand_out(key, ar, L + K)


def kv_concat_hip(base_k, base_v, new_k, new_v):
    """Plain out-of-place form: fresh contiguous [B,H,L+K,D] tensors (sd_kv_concat)."""
    _check_kv_shapes(base_k, base_v, new_k, new_v)
    dev = _require_device("kv_concat", base_k, base_v, new_k, new_v)
    esize = base_k.element_size()
    B, H, L, D = base_k.shape
    K = new_k.shape[2]
    bk, bv = _row_major(base_k), _row_major(base_v)
    if bv.stride()[:2] != bk.stride()[:2]:
        bk, bv = bk.contiguous(), bv.contiguous()
    new_k, new_v = new_k.contiguous(), new_v.contiguous()
    out_k = torch.empty((B, H, L + K, D), dtype=base_k.dtype, device=dev)
    out_v = torch.empty((B, H, L + K, D), dtype=base_k.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = _abi.load().sd_kv_concat(
            out_k.data_ptr(), out_v.data_ptr(), bk.data_ptr(), bv.data_ptr(),
            new_k.data_ptr(), new_v.data_ptr(), esize, B, H, L, K, D, L + K,
            bk.stride(0) if L else 0, bk.stride(1) if L else 0, _stream_ptr(dev),
        )
    _abi.check(rc, "sd_kv_concat")
    return out_k, out_v


def kv_append_inplace_hip(
    cache_k: torch.Tensor, cache_v: torch.Tensor, new_k: torch.Tensor, new_v: torch.Tensor,
    row_len: Optional[torch.Tensor] = None, L: int = 0,
) -> None:
    """Write new[b,h,:,:] into cache[b,h,row_len[b]:row_len[b]+K,:] (sd_kv_append)."""
    dev = _require_device("kv_append_inplace", cache_k, cache_v, new_k, new_v)
    assert cache_k.is_contiguous() and cache_v.is_contiguous(), "cache must be contiguous"
    B, H, Lmax, D = cache_k.shape
    K = new_k.shape[2]
    assert new_k.shape == (B, H, K, D) and new_v.shape == (B, H, K, D), "new_k/new_v shape mismatch"
    if row_len is not None:
        assert row_len.dtype == torch.int32 and row_len.shape == (B,) and row_len.device == dev
    new_k, new_v = new_k.contiguous(), new_v.contiguous()
    with torch.cuda.device(dev):
        rc = _abi.load().sd_kv_append(
            cache_k.data_ptr(), cache_v.data_ptr(), new_k.data_ptr(), new_v.data_ptr(),
            row_len.data_ptr() if row_len is not None else None, L,
            cache_k.element_size(), B, H, Lmax, K, D, _stream_ptr(dev),
        )
    _abi.check(rc, "sd_kv_append")


def kv_append_with_mask_hip(
    base_k: torch.Tensor, base_v: torch.Tensor, draft_k: torch.Tensor, draft_v: torch.Tensor,
    accepted_mask: torch.Tensor, accept_len: torch.Tensor, offset: int = 0,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Compacting append (reference.py:96-159) on gfx950; `offset` is unused there too."""
    assert base_k.dim() == 4, "base_k must be 4D tensor [B][H][L][D]"
    assert base_v.dim() == 4, "base_v must be 4D tensor [B][H][L][D]"
    assert draft_k.dim() == 4, "draft_k must be 4D tensor [B][H][K][D]"
    assert draft_v.dim() == 4, "draft_v must be 4D tensor [B][H][K][D]"
    assert accepted_mask.dim() == 2, "accepted_mask must be 2D tensor [B][K]"
    assert accept_len.dim() == 1, "accept_len must be 1D tensor [B]"
    dev = _require_device("kv_append_with_mask", base_k, base_v, draft_k, draft_v, accepted_mask, accept_len)
    B, H, L, D = base_k.shape
    K = draft_k.shape[2]
    esize = base_k.element_size()
    bk, bv = _row_major(base_k), _row_major(base_v)
    if bv.stride()[:2] != bk.stride()[:2]:
        bk, bv = bk.contiguous(), bv.contiguous()
    draft_k, draft_v = draft_k.contiguous(), draft_v.contiguous()
    mask = (accepted_mask != 0).to(torch.uint8).contiguous()
    alen = accept_len.to(torch.int32).contiguous()
    out_k = torch.empty((B, H, L + K, D), dtype=base_k.dtype, device=dev)
    out_v = torch.empty((B, H, L + K, D), dtype=base_v.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = _abi.load().sd_kv_append_masked(
            out_k.data_ptr(), out_v.data_ptr(), bk.data_ptr(), bv.data_ptr(),
            draft_k.data_ptr(), draft_v.data_ptr(), mask.data_ptr(), alen.data_ptr(),
            esize, B, H, L, K, D, bk.stride(0) if L else 0, bk.stride(1) if L else 0,
            _stream_ptr(dev),
        )
    _abi.check(rc, "sd_kv_append_masked")
    return out_k, out_v


def sample_token_hip(logits: torch.Tensor, temperature: float, top_k: Optional[int] = None, top_p: Optional[float] = None,
                     seed: int = 0, draw: int = 0, pos: Optional[torch.Tensor] = None, rows_per_entry: int = 1,
                     draw_counters: Optional[torch.Tensor] = None, stream_ids: Optional[torch.Tensor] = None,
                     active: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sampled token per entry (sd_sample_token): the device form of the reference's
    `sample_bonus_token_from_logits(..., do_sample=True)` (src/specdec/core/pipeline.py:48-147).

    logits: [rows, V] (or [V]) f32/bf16/f16 on the GPU, unit stride along V. Entry b reads row
    b*rows_per_entry + pos[b]. Returns int32 [entries]. `draw` (or the int32 `draw_counters`, read
    and incremented on the device) and `stream_ids` select the Philox value; see include/specdec_hip.h."""
    lib = _abi.load()
    if logits.dim() == 1:
        logits = logits.unsqueeze(0)
    if logits.dim() != 2:
        raise ValueError(f"sample_token: logits must be [rows, V], got {tuple(logits.shape)}")
    dev = _require_device("sample_token", logits)
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    rows, V = logits.shape
    if rows % rows_per_entry:
        raise ValueError(f"sample_token: {rows} rows is not a multiple of rows_per_entry={rows_per_entry}")
    B = rows // rows_per_entry
    for name, t in (("pos", pos), ("draw_counters", draw_counters), ("stream_ids", stream_ids), ("active", active)):
        if t is not None and (t.device != dev or t.dtype != torch.int32 or t.numel() != B or not t.is_contiguous()):
            raise ValueError(f"sample_token: {name} must be a contiguous int32 [{B}] tensor on {dev}")
    k = int(top_k) if top_k else 0
    p = 1.0 if top_p is None else float(top_p)
    out = torch.empty(B, dtype=torch.int32, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    with torch.cuda.device(dev):
        _abi.check(lib.sd_sample_token(logits.data_ptr(), sd_dtype(logits.dtype), logits.stride(0), B, V, ptr(pos),
                                       int(rows_per_entry), ptr(active), float(temperature), k, p, int(seed) & (2 ** 64 - 1),
                                       ptr(draw_counters), int(draw) & 0xFFFFFFFF, ptr(stream_ids), out.data_ptr(),
                                       _stream_ptr(dev)), "sd_sample_token")
    return out


def quantize_fp8_rows_hip(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(q float8_e4m3fn [N][K], scales float32 [N]) of a bf16 matrix: the per-output-row quantiser of the
    engine's fp8 weight storage (sd_quantize_fp8_rows)."""
    lib = _abi.load()
    dev = _require_device("quantize_fp8_rows", w)
    if w.dtype != torch.bfloat16 or w.dim() != 2:
        raise TypeError(f"quantize_fp8_rows: need a bf16 matrix, got {w.dtype} {tuple(w.shape)}")
    w = w.contiguous()
    N, K = w.shape
    q = torch.empty((N, K), dtype=torch.uint8, device=dev)
    sc = torch.empty(N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _abi.check(lib.sd_quantize_fp8_rows(w.data_ptr(), N, K, q.data_ptr(), sc.data_ptr(), _stream_ptr(dev)), "sd_quantize_fp8_rows")
    return q.view(torch.float8_e4m3fn), sc
