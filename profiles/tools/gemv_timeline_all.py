"""In-kernel timelines (SPECDEC_GEMV_TIMELINE=1) of the five GEMV kinds at 1 and 5 tokens, 1B and 3B shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch
from specdec_hip import weights as W
from specdec_hip.engine import HipModel
for name, preset, T in (("1b", W.LLAMA_3_2_1B, 1), ("3b", W.LLAMA_3_2_3B, 5)):
    mw = W.synthetic_llama(preset, seed=0, device="cuda")
    hm = HipModel(mw, batch=1, l_max=64)
    st = torch.cuda.Stream()
    for which in (0, 1, 2, 3, 4):
        us, nb = hm.probe_gemv(which, T=T, iters=64, stream=st)
        print(f"{name} which={which} T={T}: {us:.2f} us for {nb/1e6:.1f} MB = {nb/us/1e6:.2f} TB/s", flush=True)
