"""CPU restatement of the bonus-token sampler (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates `sample_bonus_token_from_logits` (src/specdec/core/pipeline.py:48-147) for
`do_sample=True`: temperature scaling (:90-92), top-k mask (:95-102), nucleus mask on the
softmax of the top-k-filtered logits — a token is dropped when the INCLUSIVE cumulative
probability in descending order exceeds top_p, the first is always kept (:105-121) — softmax
over what is left and one multinomial draw (:124-136).

Two things the reference leaves to torch are fixed here so that a device kernel can be
bit-identical to this file:
  * order among equal logits (torch.topk / torch.sort do not specify it): value descending,
    then index ascending; -0.0 == +0.0, NaN sorts first (as torch's argmax/topk treat it);
  * the random draw. torch.multinomial consumes the process-global generator; here the draw is a
    counter-based Philox4x32-10 value keyed by (seed; draw index, row stream, element, tag), one
    uniform per draw, inverted through the cumulative sums in sorted order. Same distribution,
    reproducible per row regardless of batch composition.
Arithmetic: probabilities are formed in float64 from the float32 logits (scaled = x / T,
e = exp(scaled - max), sums sequential in sorted order), so the only operation that is not
IEEE-exact on both sides is `exp`. The kept set and its probabilities are pinned against draws of
the reference function itself (tests/golden/hostlogic_golden.json: "sampling").

Full-vocabulary nucleus sampling (top_p < 1 without top_k) is `nucleus_distribution`: the same rule over the whole sorted
row, with the normaliser summed in an order a 1024-thread workgroup reproduces (pinned by "sampling_nucleus" draws).

With neither top-k nor top-p the draw is a Gumbel-max over the whole vocabulary (one Philox value
per element): the same categorical distribution without any ordered prefix sum.
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
TAG_CDF, TAG_GUMBEL = 0x5EED0001, 0x5EED0002
MAX_TOP_K = 1024
MASK32 = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """Philox4x32 with 10 rounds (Salmon et al., SC'11). `counter` is 4 arrays / ints of uint32,
    `key` two ints. Vectorised over the counter arrays; returns 4 uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & MASK32 for x in counter]
    c = np.broadcast_arrays(*c)
    c0, c1, c2, c3 = (x.copy() for x in c)
    k0, k1 = int(key[0]) & MASK32, int(key[1]) & MASK32
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c0
        p1 = np.uint64(PHILOX_M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(MASK32)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(MASK32)
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0, k1 = (k0 + PHILOX_W0) & MASK32, (k1 + PHILOX_W1) & MASK32
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def draw_uniform(seed: int, draw: int, stream: int) -> float:
    """The one uniform in [0, 1) of draw number `draw` of row stream `stream`."""
    x0 = philox4x32_10((draw, stream, 0, TAG_CDF), (seed & MASK32, (seed >> 32) & MASK32))[0]
    return float(x0) * 2.0 ** -32


def order_keys(x: np.ndarray) -> np.ndarray:
    """Order-preserving uint32 key of float32 values: larger key = larger value; NaN largest,
    -0.0 == +0.0."""
    x = np.asarray(x, dtype=np.float32).copy()
    x[x == 0] = 0.0
    u = x.view(np.uint32).astype(np.uint64)
    key = np.where(u & 0x80000000, (~u) & MASK32, u | 0x80000000)
    key = np.where(np.isnan(x), MASK32, key)
    return key.astype(np.uint64)


def sorted_top_k(x: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k largest entries: value descending, index ascending among equals."""
    V = x.shape[0]
    comp = (order_keys(x) << np.uint64(20)) | (np.uint64((1 << 20) - 1) - np.arange(V, dtype=np.uint64))
    order = np.argsort(comp, kind="stable")[::-1]
    return order[:k].astype(np.int64)


def filtered_distribution(logits: np.ndarray, temperature: float, top_k: Optional[int],
                          top_p: Optional[float]) -> Tuple[np.ndarray, np.ndarray]:
    """-> (token ids in sorted order, float64 weights e_i of the kept tokens; p_i = e_i / sum).
    Needs a top-k (<= MAX_TOP_K): the nucleus cut is taken inside it."""
    x = np.asarray(logits, dtype=np.float32).reshape(-1)
    V = x.shape[0]
    if not top_k or top_k <= 0:
        raise NotImplementedError("filtered_distribution needs top_k (full-vocabulary nucleus is not restated)")
    k = min(int(top_k), V)
    if k > MAX_TOP_K:
        raise ValueError(f"top_k={k} > {MAX_TOP_K}")
    idx = sorted_top_k(x, k)
    vals = x[idx].astype(np.float64)
    T = float(np.float32(temperature))
    if T > 0 and T != 1.0:
        vals = vals / T
    m = vals[0]
    if not np.isfinite(m):       # -inf / NaN on top: the reference falls back to argmax (:129-131)
        return idx[:1], np.ones(1)
    with np.errstate(invalid="ignore"):
        e = np.exp(vals - m)
    e = np.where(np.isnan(e), 0.0, e)
    n_keep = k
    if top_p is not None and float(np.float32(top_p)) < 1.0:   # the device holds top_p as float32
        tp = float(np.float32(top_p))
        z = 0.0
        for v in e:              # sequential, sorted order
            z += v
        cum, n_keep = 0.0, 0
        for i in range(k):
            cum += e[i] / z
            if i == 0 or not (cum > tp):
                n_keep = i + 1
            else:
                break
    return idx[:n_keep], e[:n_keep]


NUCLEUS_SLOTS = 1024


def nucleus_distribution(logits: np.ndarray, temperature: float, top_p: float) -> Tuple[np.ndarray, np.ndarray]:
    """Full-vocabulary nucleus (top_k = None, top_p < 1; pipeline.py:105-125): -> (token ids of the kept tokens in sorted
    order, their float64 weights e_i). The cumulative probability runs over the softmax of the WHOLE row. The normaliser is
    summed in the order the device kernel can reproduce: slot s = i mod 1024 collects its elements in index order, the 1024
    slot sums are added in slot order."""
    x = np.asarray(logits, dtype=np.float32).reshape(-1)
    V = x.shape[0]
    order = sorted_top_k(x, V)
    T = float(np.float32(temperature))
    vals = x.astype(np.float64)
    if T > 0 and T != 1.0:
        vals = vals / T
    m = vals[order[0]]
    if not np.isfinite(m):
        return order[:1], np.ones(1)
    with np.errstate(invalid="ignore"):
        e_all = np.exp(vals - m)
    e_all = np.where(np.isnan(e_all), 0.0, e_all)
    z = 0.0
    for s in range(min(NUCLEUS_SLOTS, V)):
        p = 0.0
        for v in e_all[s::NUCLEUS_SLOTS]:
            p += v
        z += p
    tp = float(np.float32(top_p))
    e = e_all[order]
    cum, n_keep = 0.0, 0
    for i in range(V):
        cum += e[i] / z
        if i == 0 or not (cum > tp):
            n_keep = i + 1
        else:
            break
    return order[:n_keep], e[:n_keep]


def sample_token_ref(logits: np.ndarray, temperature: float, top_k: Optional[int], top_p: Optional[float],
                     seed: int, draw: int, stream: int) -> int:
    """One sampled token id (do_sample=True)."""
    x = np.asarray(logits, dtype=np.float32).reshape(-1)
    if (not top_k or top_k <= 0):
        if top_p is not None and float(np.float32(top_p)) < 1.0:
            ids, e = nucleus_distribution(x, temperature, top_p)
        else:
            return gumbel_argmax_ref(x, temperature, seed, draw, stream)
    else:
        ids, e = filtered_distribution(x, temperature, top_k, top_p)
    z = 0.0
    for v in e:
        z += v
    target = draw_uniform(seed, draw, stream) * z
    c = 0.0
    for i in range(len(ids)):
        c += e[i]
        if target < c:
            return int(ids[i])
    return int(ids[-1])


def gumbel_argmax_ref(x: np.ndarray, temperature: float, seed: int, draw: int, stream: int) -> int:
    """argmax_i (x_i / T + G_i), G_i = -log(-log(u_i)), u_i = (philox_i + 0.5) * 2^-32; ties and
    NaN as the device argmax (first index wins, NaN is largest)."""
    V = x.shape[0]
    T = float(np.float32(temperature))
    vals = x.astype(np.float64)
    if T > 0 and T != 1.0:
        vals = vals / T
    r = philox4x32_10((np.full(V, draw, dtype=np.uint64), np.full(V, stream, dtype=np.uint64), np.arange(V, dtype=np.uint64),
                       np.full(V, TAG_GUMBEL, dtype=np.uint64)), (seed & MASK32, (seed >> 32) & MASK32))[0]
    u = (r.astype(np.float64) + 0.5) * 2.0 ** -32
    score = vals + (-np.log(-np.log(u)))
    if np.isnan(score).any():
        return int(np.argmax(np.isnan(score)))
    return int(np.argmax(score))
