"""Which workgroups of a weight-streaming launch finish late, and is it the same ones every time?
SPECDEC_GEMV_TIMELINE=2 python profiles/tools/wg_spread.py [3b|1b] [tokens] 2> log; the per-workgroup end stamps of several probes of
the same launch are correlated with each other and with the workgroup's index (XCD = index % 8)."""
import os
import re
import subprocess
import sys

import numpy as np

if os.environ.get("WG_SPREAD_CHILD"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
    import torch
    from specdec_hip import weights as W
    from specdec_hip.engine import HipModel
    preset = {"3b": W.LLAMA_3_2_3B, "1b": W.LLAMA_3_2_1B}[sys.argv[1]]
    T = int(sys.argv[2])
    hm = HipModel(W.synthetic_llama(preset, seed=0, device="cuda"), batch=1, l_max=64)
    st = torch.cuda.Stream()
    for rep in range(4):
        for which in (2, 3):
            hm.probe_gemv(which, T=T, iters=57 + rep, stream=st)   # (different counts: the stamped launch lands on different layers)
    sys.exit(0)

model, T = (sys.argv[1] if len(sys.argv) > 1 else "3b"), (sys.argv[2] if len(sys.argv) > 2 else "5")
env = dict(os.environ, WG_SPREAD_CHILD="1", SPECDEC_GEMV_TIMELINE="2")
res = subprocess.run([sys.executable, os.path.abspath(__file__), model, T], env=env, capture_output=True, text=True)
runs = {}
for m in re.finditer(r"\[timeline-wg which=(\d) T=\d+ (\w+)\]([^\n]*)", res.stderr):
    runs.setdefault((int(m.group(1)), m.group(2)), []).append(np.array([float(x) for x in m.group(3).split()]))
for (which, stamp), v in sorted(runs.items()):
    a = np.stack(v)                       # [runs][256]
    mean_wg = a.mean(0)
    print(f"which={which} {stamp}: {len(v)} probes; per probe min/mean/max = " + ", ".join(f"{x.min():.2f}/{x.mean():.2f}/{x.max():.2f}" for x in a))
    c = np.corrcoef(a)
    print(f"  correlation of the per-workgroup stamps between probes: {c[np.triu_indices(len(v), 1)].round(2).tolist()}")
    print("  mean by XCD (index % 8): " + " ".join(f"{mean_wg[x::8].mean():.2f}" for x in range(8)))
    print("  mean by index // 32:     " + " ".join(f"{mean_wg[32 * x:32 * x + 32].mean():.2f}" for x in range(8)))
    order = np.argsort(mean_wg)
    print(f"  earliest workgroups {order[:8].tolist()} ({mean_wg[order[:8]].round(2).tolist()}), latest {order[-8:].tolist()} ({mean_wg[order[-8:]].round(2).tolist()})")
