"""Does a GEMV run faster when its weights are cache-resident?  sd_model_probe_gemv round-robin over all layers
(weights from HBM) against SPECDEC_PROBE_HOT=1 (same matrix every launch: L2 / Infinity Cache resident).
python profiles/tools/probe_hot.py [preset]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch  # noqa: E402

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipModel  # noqa: E402

preset = {"3b": W.LLAMA_3_2_3B, "1b": W.LLAMA_3_2_1B, "8b": W.LLAMA_3_8B}[sys.argv[1] if len(sys.argv) > 1 else "3b"]
mw = W.synthetic_llama(preset, seed=0, device="cuda")
hm = HipModel(mw, batch=1, l_max=64)
st = torch.cuda.Stream()
names = {0: "qkv", 1: "o_proj", 2: "gate_up", 3: "down"}
for hot in ("0", "1", "2"):
    os.environ["SPECDEC_PROBE_HOT"] = hot
    for T in (1, 5):
        row = []
        for which in (0, 1, 2, 3):
            us, nb = hm.probe_gemv(which, T=T, iters=200, stream=st)
            row.append(f"{names[which]} {nb / 1e6:6.1f} MB {us:6.2f} us {nb / us / 1e6:5.2f} TB/s")
        print(f"hot={hot} T={T} | " + " | ".join(row), flush=True)
