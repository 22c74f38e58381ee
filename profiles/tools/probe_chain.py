"""Chained pairs of gemv_chain.hip against the two separate launches they replace (sd_model_probe_gemv, HIP events,
round-robin over the layers).  python profiles/tools/probe_chain.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch  # noqa: E402

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipModel  # noqa: E402

for name, preset, Ts in (("1b", W.LLAMA_3_2_1B, (1, 2)), ("3b", W.LLAMA_3_2_3B, (5,))):
    mw = W.synthetic_llama(preset, seed=0, device="cuda")
    hm = HipModel(mw, batch=1, l_max=64)
    st = torch.cuda.Stream()
    for T in Ts:
        us = {w: hm.probe_gemv(w, T=T, iters=200, stream=st)[0] for w in (0, 1, 2, 3, 5, 6)}
        print(f"{name} T={T}: o_proj {us[1]:.2f} + gate_up {us[2]:.2f} = {us[1] + us[2]:.2f} us   chained {us[5]:.2f} us | "
              f"down {us[3]:.2f} + qkv {us[0]:.2f} = {us[3] + us[0]:.2f} us   chained {us[6]:.2f} us", flush=True)
    print(name, "chain status", hm.chain_status(), flush=True)
