"""Where the host's time goes inside one decode step at batch 1 (3B + 1B, K=4): graph launch call,
wait for the step record, host rules. python profiles/tools/host_timeline.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
import torch  # noqa: E402

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipSpecDec  # noqa: E402
from src.specdec import HipLM, SpeculativePipeline  # noqa: E402

tgt = W.synthetic_llama(W.LLAMA_3_2_3B, seed=0, device="cuda")
drf = W.synthetic_llama(W.LLAMA_3_2_1B, seed=1, device="cuda", embed_from=tgt, flip_fraction=0.2)
pipe = SpeculativePipeline(base_lm=HipLM(tgt), draft_lm=HipLM(drf), controller="fixed", controller_params={"k": 4}, seed=1234)
g = torch.Generator().manual_seed(1234)
prompt = torch.randint(4, tgt.config.vocab, (32,), generator=g).tolist()
sess = pipe.start_session([prompt], max_tokens=2000, emit_mode=HipSpecDec.EMIT_BONUS)
for _ in range(10):
    sess.advance()
loop = sess.loop
N = 200
t_launch = t_wait = t_rest = 0.0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
orig_step, orig_sync = loop.step, loop.sync
acc = {"launch": 0.0, "wait": 0.0}


def step(*a, **k):
    t = time.perf_counter()
    orig_step(*a, **k)
    acc["launch"] += time.perf_counter() - t


def sync():
    t = time.perf_counter()
    r = orig_sync()
    acc["wait"] += time.perf_counter() - t
    return r


orig_wait = loop.wait


def wait(i):
    t = time.perf_counter()
    r = orig_wait(i)
    acc["wait"] += time.perf_counter() - t
    return r


loop.step, loop.sync, loop.wait = step, sync, wait
t0 = time.perf_counter()
for _ in range(N):
    sess.advance()
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"per step: total {total / N * 1e6:.1f} us | graph launch call {acc['launch'] / N * 1e6:.1f} us | wait for record "
      f"{acc['wait'] / N * 1e6:.1f} us | rest of advance() (rules, bookkeeping) {(total - acc['launch'] - acc['wait']) / N * 1e6:.1f} us")
