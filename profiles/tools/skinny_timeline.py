import os, sys
sys.path.insert(0, "/root/repo/llm-inference-lab_amd")
import torch
from specdec_hip import weights as W
from specdec_hip.engine import HipModel
mw = W.synthetic_llama(W.LLAMA_3_2_3B, seed=0, device="cuda")
hm = HipModel(mw, batch=1, l_max=64)
st = torch.cuda.Stream()
for which in (2, 3, 1):
    for T in (5, 16, 40, 64):
        us, nb = hm.probe_gemv(which, T=T, iters=40, stream=st)
        print(f"which={which} T={T}: {us:.1f} us", flush=True)
