"""In-kernel timeline of the QKV GEMV (norm + projection + RoPE + KV append) next to the out-projection, 1B and 3B shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch
from specdec_hip import weights as W
from specdec_hip.engine import HipModel
for name, preset in (("1b", W.LLAMA_3_2_1B), ("3b", W.LLAMA_3_2_3B)):
    mw = W.synthetic_llama(preset, seed=0, device="cuda")
    hm = HipModel(mw, batch=1, l_max=64)
    st = torch.cuda.Stream()
    for which, T in ((0, 1), (1, 1), (0, 5), (1, 5)):
        us, nb = hm.probe_gemv(which, T=T, iters=64, stream=st)
        print(f"{name} which={which} T={T}: {us:.2f} us for {nb/1e6:.1f} MB", flush=True)
