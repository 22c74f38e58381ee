import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch
from specdec_hip import weights as W
from specdec_hip.engine import HipModel
cfg = W.LLAMA_3_2_3B if sys.argv[1] == "3b" else W.LLAMA_3_2_1B
tgt = W.synthetic_llama(cfg, seed=0, device="cuda")
m = HipModel(tgt, 1, 256)
st = torch.cuda.Stream()
for which in (1, 2, 3, 4):
    for T in (1, 5):
        u, nb = m.probe_gemv(which, T, 60, st)
        print("which=%d T=%d  %.2f us  %.2f TB/s" % (which, T, u, nb / u / 1e6), flush=True)
