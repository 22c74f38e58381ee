"""Where a session's start-up time goes: engine creation (KV caches, workspaces, pinned records), the prompt's prefill passes of both
models, and the first (eager) step — `python profiles/tools/prefill_probe.py [L ...]`. The prefill figure is the second of two prefills of
the same prompt into the same session (idempotent: in-place K/V at the same positions), i.e. without one-time allocation / attribute set-up."""
import dataclasses
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch  # noqa: E402

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipSpecDec  # noqa: E402
from src.specdec import HipLM, SpeculativePipeline  # noqa: E402

lens = [int(x) for x in sys.argv[1:]] or [32, 128, 512, 2048]
tgt = W.synthetic_llama(dataclasses.replace(W.LLAMA_3_2_3B, max_pos=8192), seed=0, device="cuda")
drf = W.synthetic_llama(dataclasses.replace(W.LLAMA_3_2_1B, max_pos=8192), seed=1, device="cuda", embed_from=tgt, flip_fraction=0.2)
pipe = SpeculativePipeline(base_lm=HipLM(tgt), draft_lm=HipLM(drf), controller="fixed", controller_params={"k": 4}, seed=1234)
nbytes = tgt.matmul_bytes() + drf.matmul_bytes()
for L in lens:
    g = torch.Generator().manual_seed(L)
    prompt = torch.randint(4, tgt.config.vocab, (L,), generator=g).tolist()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sess = pipe.start_session([prompt], max_tokens=400, emit_mode=HipSpecDec.EMIT_BONUS)
    torch.cuda.synchronize()
    t_start = time.perf_counter() - t0
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe._prefill(sess.rt, sess.rows)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t_pre = min(ts)
    flop = 2.0 * (tgt.config.n_params_matmul + drf.config.n_params_matmul) * (L - 1)
    print(f"prompt {L:5d}: start_session {t_start * 1e3:8.1f} ms (first call at this cache size: allocation + prefill) | prefill alone {t_pre * 1e3:8.2f} ms "
          f"= {(L - 1) / t_pre:9.0f} tok/s, {flop / t_pre / 1e12:6.1f} TFLOP/s (weights streamed once would be {nbytes / 6.9e12 * 1e3:.2f} ms)", flush=True)
    sess.finish()
