"""fp8 weight storage restated on the CPU (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The reference has no fp8 path (BASELINE config 5 names "fp8 weights" as a target, nothing in
/root/reference implements it), so the quantiser is specified here and in include/specdec_hip.h —
parity for it is between this file and the device, unpinned against the reference:
  scale[r] = max|w[r][:]| / 448      (float32 division; 1 for an all-zero row)
  q[r][k]  = e4m3fn( w[r][k] / scale[r] )   (OCP e4m3, round to nearest even; |.| <= 448 by construction)
  y[r]     = scale[r] * sum_k q[r][k] x[k]
`dequantized` returns a ModelWeights whose Linear matrices are the float32 values q*scale, which the
fp32-matrix oracle forward (oracle/model_ref.py) consumes unchanged; everything else (embeddings, norms,
biases, KV, activations) stays as it is."""

from __future__ import annotations

import dataclasses
from typing import Tuple

import torch


def quantize_rows(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(q float8_e4m3fn [N][K], scale float32 [N])"""
    wf = w.detach().to("cpu", torch.float32)
    amax = wf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / torch.tensor(448.0, dtype=torch.float32), torch.ones_like(amax))
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    return q, scale


def dequantized(weights):
    def dq(w):
        q, s = quantize_rows(w)
        return q.to(torch.float32) * s[:, None]

    layers = [dataclasses.replace(l, wqkv=dq(l.wqkv), wo=dq(l.wo), w_up=dq(l.w_up), w_down=dq(l.w_down)) for l in weights.layers]
    cpu = weights.to("cpu")
    return dataclasses.replace(cpu, layers=[dataclasses.replace(l0, wqkv=l1.wqkv, wo=l1.wo, w_up=l1.w_up, w_down=l1.w_down)
                                            for l0, l1 in zip(cpu.layers, layers)],
                               lm_head=dq(weights.lm_head), meta={})
