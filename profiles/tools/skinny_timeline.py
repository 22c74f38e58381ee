"""In-kernel timeline (SPECDEC_GEMV_TIMELINE=1: s_memrealtime stamps of wave 0 of every workgroup, last launch) of the
multi-token GEMMs.  SPECDEC_GEMV_TIMELINE=1 python profiles/tools/skinny_timeline.py [preset] [tokens...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch  # noqa: E402

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipModel  # noqa: E402

preset = {"3b": W.LLAMA_3_2_3B, "1b": W.LLAMA_3_2_1B, "8b": W.LLAMA_3_8B}[sys.argv[1] if len(sys.argv) > 1 else "3b"]
tokens = [int(t) for t in sys.argv[2:]] or [20, 40]
mw = W.synthetic_llama(preset, seed=0, device="cuda")
hm = HipModel(mw, batch=1, l_max=64)
st = torch.cuda.Stream()
for which in [int(w) for w in os.environ.get("PROBE_WHICH", "0,2,4").split(",")]:
    for T in tokens:
        us, nb = hm.probe_gemv(which, T=T, iters=40, stream=st)
        print(f"which={which} T={T}: {us:.1f} us", flush=True)
