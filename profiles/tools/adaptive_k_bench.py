"""Per-row adaptive K inside the captured step vs fixed K, at two acceptance levels (3B target + 1B draft, greedy).
python profiles/tools/adaptive_k_bench.py [rows] [max_tokens]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch  # noqa: E402

from specdec_hip import weights as W  # noqa: E402
from src.specdec import HipLM, SpeculativePipeline  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
max_tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 192
tgt = W.synthetic_llama(W.LLAMA_3_2_3B, seed=0, device="cuda")
base = HipLM(tgt)
prompts = [torch.randint(4, 32000, (32,), generator=torch.Generator().manual_seed(100 + i)).tolist() for i in range(rows)]
for flip in (0.2, 0.6, 0.9):
    drf = W.synthetic_llama(W.LLAMA_3_2_1B, seed=1, device="cuda", embed_from=tgt, flip_fraction=flip)
    draft = HipLM(drf)
    ref = None
    for name, ctl, params in (("fixed K=4", "fixed", {"k": 4}),
                              ("host adaptive (graph swap)", "adaptive", {"initial_k": 4, "min_k": 1, "max_k": 4, "target_acceptance_rate": 0.7}),
                              ("per-row adaptive, on device", "adaptive", {"initial_k": 4, "min_k": 1, "max_k": 4, "target_acceptance_rate": 0.7, "per_row": True})):
        pipe = SpeculativePipeline(base_lm=base, draft_lm=draft, controller=ctl, controller_params=params, seed=0)
        pipe.generate_batch(prompts, max_tokens=24, do_sample=False)       # warm-up: capture
        torch.cuda.synchronize()
        t0 = time.time()
        out = pipe.generate_batch(prompts, max_tokens=max_tokens, do_sample=False)
        torch.cuda.synchronize()
        dt = time.time() - t0
        toks = sum(len(o["generated_tokens"]) for o in out)
        steps = out[0]["batch_metrics"]["total_steps"]
        if ref is None:
            ref = [o["generated_tokens"] for o in out]
        same = all(o["generated_tokens"][: min(len(o["generated_tokens"]), len(r))] == r[: min(len(o["generated_tokens"]), len(r))] for o, r in zip(out, ref))
        ks = out[0].get("k_trace", [])
        print(f"flip {flip:.1f} {name:32s} {toks / dt:8.1f} tok/s  {dt / max(steps, 1) * 1e3:6.3f} ms/step  steps {steps:4d}  proposed {sum(o['proposed'] for o in out):5d}"
              f"  tokens identical to fixed K: {same}  k (row 0, last 8 steps): {ks[-8:]}", flush=True)
    del draft, drf
