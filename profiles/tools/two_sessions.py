"""Row-group concurrency, measured: G independent decode sessions of B/G rows each, every one with its own captured step
graph on its own pair of HIP streams, driven from G host threads — against ONE session of B rows. This is the upper
bound of what row-group ping-pong inside one step (draft of group A next to verify of group B) could give: the groups
run with no ordering between them at all.   python profiles/tools/two_sessions.py [rows] [groups] [steps]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm-inference-lab_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipSpecDec  # noqa: E402
from src.specdec import HipLM, SpeculativePipeline  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
K = 4
tgt = W.synthetic_llama(W.LLAMA_3_2_3B, seed=0, device="cuda")
drf = W.synthetic_llama(W.LLAMA_3_2_1B, seed=1, device="cuda", embed_from=tgt, flip_fraction=0.2)
blm, dlm = HipLM(tgt), HipLM(drf)


def prompts(n, off):
    return [torch.randint(4, tgt.config.vocab, (32,), generator=torch.Generator().manual_seed(1234 + off + i)).tolist() for i in range(n)]


def session(n, off):
    pipe = SpeculativePipeline(base_lm=blm, draft_lm=dlm, controller="fixed", controller_params={"k": K}, seed=1234)
    return pipe.start_session(prompts(n, off), max_tokens=(steps + 8) * (K + 1) + 1, emit_mode=HipSpecDec.EMIT_BONUS)


def run(sessions):
    for s in sessions:
        for _ in range(5):
            s.advance()
    torch.cuda.synchronize()
    n0 = [sum(len(r.generated) for r in s.rows) for s in sessions]
    t0 = time.perf_counter()
    ths = [threading.Thread(target=lambda s=s: [s.advance() for _ in range(steps)]) for s in sessions]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for s in sessions:
        s.finish()
    tok = sum(sum(len(r.generated) for r in s.rows) - n for s, n in zip(sessions, n0))
    return dt / steps * 1e3, tok / dt


ms1, tps1 = run([session(rows, 0)])
print(f"1 session x {rows} rows: {ms1:.3f} ms/step, {tps1:.0f} tok/s", flush=True)
per = rows // groups
msg, tpsg = run([session(per, g * per) for g in range(groups)])
print(f"{groups} concurrent sessions x {per} rows: {msg:.3f} ms per step of each, {tpsg:.0f} tok/s aggregate ({tpsg / tps1:.2f}x)", flush=True)
msh, tpsh = run([session(per, 0)])
print(f"1 session x {per} rows alone: {msh:.3f} ms/step, {tpsh:.0f} tok/s", flush=True)
