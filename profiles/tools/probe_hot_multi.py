"""Are the multi-token launches (17..64 tokens: gemm_pipe / gemm_slice / gemm_direct) limited by the HBM latency of their weight
stream or by what a CU pulls through its L1?  sd_model_probe_gemv round-robin over all layers (weights from HBM) against
SPECDEC_PROBE_HOT=1 (the same matrix every launch: L2 / Infinity Cache resident).  python profiles/tools/probe_hot_multi.py [preset]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch  # noqa: E402

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipModel  # noqa: E402

preset = {"3b": W.LLAMA_3_2_3B, "1b": W.LLAMA_3_2_1B, "8b": W.LLAMA_3_8B}[sys.argv[1] if len(sys.argv) > 1 else "3b"]
mw = W.synthetic_llama(preset, seed=0, device="cuda")
hm = HipModel(mw, batch=8, l_max=64)
st = torch.cuda.Stream()
names = {0: "qkv", 1: "o_proj", 2: "gate_up", 3: "down"}
for hot in ("0", "1"):
    os.environ["SPECDEC_PROBE_HOT"] = hot
    for T in (5, 16, 40):
        row = []
        for which in (0, 1, 2, 3):
            us, nb = hm.probe_gemv(which, T=T, iters=200, stream=st)
            row.append(f"{names[which]} {nb / 1e6:6.1f} MB {us:6.2f} us")
        print(f"hot={hot} T={T:2d} | " + " | ".join(row), flush=True)
