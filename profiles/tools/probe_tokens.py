"""Launch duration of the forward's GEMVs/GEMMs of a model as a function of the tokens per pass
(sd_model_probe_gemv, HIP events, round-robin over the layers).  python profiles/tools/probe_tokens.py [preset]"""
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
names = {0: "qkv", 1: "o_proj", 2: "gate_up", 3: "down", 4: "lm_head"}
print("tokens | " + " | ".join(f"{names[w]:>16s}" for w in (0, 1, 2, 3, 4)) + "   (us, TB/s)")
for T in [int(t) for t in os.environ.get("PROBE_TOKENS", "1,5,9,10,16,17,24,32,40,48,64,80,96,128").split(",")]:
    if T > hm.pass_tokens:
        break
    row = []
    for which in (0, 1, 2, 3, 4):
        us, nb = hm.probe_gemv(which, T=T, iters=120 if which != 4 else 40, stream=st)
        row.append(f"{us:7.1f} {nb / us / 1e6:6.2f}")
    print(f"{T:6d} | " + " | ".join(f"{r:>16s}" for r in row), flush=True)
