#!/usr/bin/env python3
"""verify_prefix microbenchmark — counterpart of the reference's scripts/microbench_verify.py
(:35-101): the same 12 shapes (B in {1,8}, K in {1,2,4}, V in {4096,8192,32768}), 30 % planted
matches, 10 warm-up + 100 timed calls each synchronised, HIP op vs a PyTorch expression of the
same contract (argmax + cumprod on the device). Adds the Llama vocabulary (V = 128256) in bf16."""

from __future__ import annotations

import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from kernels import get_kernel_info, verify_prefix  # noqa: E402

CONFIGS = [(1, 1, 4096), (1, 2, 4096), (1, 4, 4096), (8, 1, 4096), (8, 2, 4096), (8, 4, 4096),
           (1, 1, 8192), (1, 2, 8192), (1, 4, 8192), (1, 1, 32768), (1, 2, 32768), (1, 4, 32768),
           (1, 4, 128256), (8, 4, 128256), (8, 8, 128256)]


def torch_same_contract(logits, ids):
    m = (logits.argmax(-1) == ids).to(torch.int32).cumprod(1)
    return m.sum(1).to(torch.int32), m.to(torch.uint8)


def timed(fn, *a, n=100):
    for _ in range(10):
        fn(*a)
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.time()
        fn(*a)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    return float(np.mean(ts)) * 1e3, float(np.std(ts)) * 1e3


if __name__ == "__main__":
    print("kernel info:", get_kernel_info())
    rng = np.random.default_rng(0)
    rows = []
    for B, K, V in CONFIGS:
        dt = torch.bfloat16 if V == 128256 else torch.float32
        logits = torch.randn(B, K, V).to(dt)
        ids = torch.randint(0, V, (B, K))
        for b in range(B):
            for k in range(K):
                if rng.random() < 0.3:
                    logits[b, k, ids[b, k]] = 10.0
        logits, ids = logits.cuda(), ids.cuda()
        a, m = verify_prefix(logits, ids)
        ra, rm = torch_same_contract(logits, ids)
        assert torch.equal(a, ra) and torch.equal(m, rm)
        hip_ms, hip_sd = timed(verify_prefix, logits, ids)
        ref_ms, ref_sd = timed(torch_same_contract, logits, ids)
        gbps = logits.numel() * logits.element_size() / (hip_ms * 1e-3) / 1e9
        rows.append((B, K, V, str(dt).split(".")[1], hip_ms, ref_ms, ref_ms / hip_ms, gbps))
        print(f"B={B} K={K} V={V:6d} {rows[-1][3]:8s} hip {hip_ms:.4f} ms  torch {ref_ms:.4f} ms  x{ref_ms / hip_ms:.2f}  {gbps:.0f} GB/s (host-timed)")
