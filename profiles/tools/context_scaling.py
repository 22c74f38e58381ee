"""Step time of the 3B + 1B pair (K=4, 1 row) as a function of the context length already in the KV cache.
python profiles/tools/context_scaling.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch  # noqa: E402

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipSpecDec  # noqa: E402
from src.specdec import HipLM, SpeculativePipeline  # noqa: E402

import dataclasses  # noqa: E402

# the presets carry 4096 positions (rope tables are [max_pos][D/2] fp32); the long-context rows need more
tgt = W.synthetic_llama(dataclasses.replace(W.LLAMA_3_2_3B, max_pos=36864), seed=0, device="cuda")
drf = W.synthetic_llama(dataclasses.replace(W.LLAMA_3_2_1B, max_pos=36864), seed=1, device="cuda", embed_from=tgt, flip_fraction=0.2)
pipe = SpeculativePipeline(base_lm=HipLM(tgt), draft_lm=HipLM(drf), controller="fixed", controller_params={"k": 4}, seed=1234)
for L in (32, 512, 1400, 2048, 8192, 32768):
    g = torch.Generator().manual_seed(L)
    prompt = torch.randint(4, tgt.config.vocab, (L,), generator=g).tolist()
    sess = pipe.start_session([prompt], max_tokens=400, emit_mode=HipSpecDec.EMIT_BONUS)
    torch.cuda.synchronize()
    # prefill alone (the second of two prefills of the same prompt into the session: without the session's one-time allocations)
    t0 = time.perf_counter()
    pipe._prefill(sess.rt, sess.rows)
    torch.cuda.synchronize()
    t_prefill = time.perf_counter() - t0
    for _ in range(5):
        sess.advance()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n0 = len(sess.rows[0].generated)
    for _ in range(30):
        sess.advance()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sess.finish()
    print(f"context {L:6d}: prefill {t_prefill * 1e3:8.1f} ms ({L / t_prefill:8.0f} tok/s) | {dt / 30 * 1e3:6.2f} ms/step | "
          f"{(len(sess.rows[0].generated) - n0) / dt:7.1f} tok/s | draft persistent: {bool(sess.rt['draft'].persist_active(1))}, path switches "
          f"{sess.stats.get('path_switches', 0)}", flush=True)
