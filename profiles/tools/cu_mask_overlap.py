"""R1 (draft / verify overlap), measured on the chip rather than argued: two independent, latency-bound launch chains on two
streams with DISJOINT CU masks (hipExtStreamCreateWithCUMask) — chain D = Llama-3.2-1B 1-token forwards (the draft's pass), chain V =
Llama-3.2-3B 5-token forwards (the verify pass), both on the launch path (one 256-workgroup kernel per operator; on a masked stream the
workgroups run in 256 / n_cus rounds). Each chain alone on the whole chip, alone on its share, and both together:

    python profiles/tools/cu_mask_overlap.py [--iters 30]

Prints per-forward times and the aggregate HBM rate (weights streamed by both chains / wall time)."""
import argparse
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-lab_amd"))
sys.path.insert(0, ROOT)
os.environ["SPECDEC_NO_PERSIST"] = "1"      # both chains on the launch path: the persistent launch needs all 256 CUs

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipModel  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: 256 booleans, CU i enabled; returns a torch ExternalStream over a stream created with that CU mask."""
    words = (ctypes.c_uint32 * 8)()
    for i, on in enumerate(bits):
        if on:
            words[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    return torch.cuda.ExternalStream(st.value)


def pattern(k_of_8):
    """CU i enabled when (i % 8) < k: the same share of every group of 8 consecutive CU ids (and so of every XCD / shader engine)."""
    return [(i % 8) < k_of_8 for i in range(256)]


def chain(model, toks, pos, n, stream):
    for _ in range(n):
        model.forward(toks, pos, 0, want_ids=True, stream=stream)


def timed(fn_list, streams):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in streams]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (e0, _), s in zip(ev, streams):
        e0.record(s)
    for f in fn_list:
        f()
    for (_, e1), s in zip(ev, streams):
        e1.record(s)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return [e0.elapsed_time(e1) for e0, e1 in ev], wall * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    n = a.iters
    drf = W.random_init(W.LLAMA_3_2_1B, seed=1, device="cuda")
    tgt = W.random_init(W.LLAMA_3_2_3B, seed=0, device="cuda")
    D, V = HipModel(drf, batch=1, l_max=256), HipModel(tgt, batch=1, l_max=256)
    bytes_d, bytes_v = drf.matmul_bytes(), tgt.matmul_bytes()
    ctx = torch.randint(4, 1000, (1, 64), dtype=torch.int32, device="cuda")
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    for m in (D, V):
        m.forward(ctx, zero, 0, skip_head=True)
    pos = torch.tensor([64], dtype=torch.int32, device="cuda")
    td, tv = ctx[:, :1].contiguous(), ctx[:, :5].contiguous()
    torch.cuda.synchronize()
    full_d, full_v = torch.cuda.Stream(), torch.cuda.Stream()
    # warm-up (one-time attribute setup of every kernel variant)
    chain(D, td, pos, 2, full_d)
    chain(V, tv, pos, 2, full_v)
    torch.cuda.synchronize()
    (ms_d,), _ = timed([lambda: chain(D, td, pos, n, full_d)], [full_d])
    (ms_v,), _ = timed([lambda: chain(V, tv, pos, n, full_v)], [full_v])
    print(f"alone, whole chip: draft 1B 1-token forward {ms_d / n * 1e3:7.1f} us ({bytes_d / (ms_d / n * 1e-3) / 1e12:.2f} TB/s) | "
          f"verify 3B 5-token forward {ms_v / n * 1e3:7.1f} us ({bytes_v / (ms_v / n * 1e-3) / 1e12:.2f} TB/s)")
    # the draft chain runs more forwards, so that both chains take about as long (as in a step: 4 draft forwards per verify pass)
    nd_it = max(1, round(n * ms_v / ms_d))
    # both chains on unmasked streams: the dispatcher interleaves whole 256-workgroup kernels
    (a_d, a_v), wall = timed([lambda: chain(D, td, pos, nd_it, full_d), lambda: chain(V, tv, pos, n, full_v)], [full_d, full_v])
    agg = (bytes_d * nd_it + bytes_v * n) / (max(a_d, a_v) * 1e-3) / 1e12
    print(f"together, no masks ({nd_it} draft / {n} verify forwards): draft {a_d / nd_it * 1e3:7.1f} us, verify {a_v / n * 1e3:7.1f} us per forward; "
          f"aggregate {agg:.2f} TB/s (one after the other: {(bytes_d * nd_it + bytes_v * n) / ((ms_d / n * nd_it + ms_v) * 1e-3) / 1e12:.2f} TB/s)")
    for k in (2, 3, 4):
        sd, sv = masked_stream(pattern(k)), masked_stream([not b for b in pattern(k)])
        nd, nv = 32 * k, 256 - 32 * k
        chain(D, td, pos, 2, sd)
        chain(V, tv, pos, 2, sv)
        torch.cuda.synchronize()
        (m_d,), _ = timed([lambda: chain(D, td, pos, n, sd)], [sd])
        (m_v,), _ = timed([lambda: chain(V, tv, pos, n, sv)], [sv])
        k_d = max(1, round(n * m_v / m_d))     # forwards of the draft chain that take about as long as n verify forwards on these shares
        (b_d, b_v), wall = timed([lambda: chain(D, td, pos, k_d, sd), lambda: chain(V, tv, pos, n, sv)], [sd, sv])
        # rate while BOTH run = bytes of both over the shorter span (the longer chain's share of it pro rata)
        span = min(b_d, b_v)
        both = (bytes_d * k_d * min(1.0, span / b_d) + bytes_v * n * min(1.0, span / b_v)) / (span * 1e-3) / 1e12
        print(f"masks {nd:3d} / {nv:3d} CUs: alone on its share: draft {m_d / n * 1e3:7.1f} us, verify {m_v / n * 1e3:7.1f} us | together "
              f"({k_d} / {n} forwards): draft {b_d / k_d * 1e3:7.1f} us, verify {b_v / n * 1e3:7.1f} us -> {both:.2f} TB/s while both run")


if __name__ == "__main__":
    main()
