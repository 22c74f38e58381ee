"""Scratch probe: first timing of the 3B+1B step loop (not the contract bench)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "llm-inference-lab_amd"))
import torch
from specdec_hip import weights as W
from specdec_hip.engine import HipModel, HipSpecDec

K = int(os.environ.get("K", 4)); B = int(os.environ.get("B", 1)); P = 32; steps = 30
t0 = time.time()
tgt = W.synthetic_llama(W.LLAMA_3_2_3B, seed=0, device="cuda")
drf = W.synthetic_llama(W.LLAMA_3_2_1B, seed=1, device="cuda", embed_from=tgt, flip_fraction=0.2)
torch.cuda.synchronize(); print("weights built in %.1fs" % (time.time() - t0), flush=True)
Lmax = 512
tm = HipModel(tgt, B, Lmax); dm = HipModel(drf, B, Lmax)
sd = HipSpecDec(dm, tm, B, K, 0)
prompts = torch.stack([torch.randint(4, 128256, (P,), generator=torch.Generator().manual_seed(1234 + i)) for i in range(B)]).to(torch.int32).cuda()
zero = torch.zeros(B, dtype=torch.int32, device="cuda")
t0 = time.time()
tm.forward(prompts[:, :-1], zero, 0, skip_head=True); dm.forward(prompts[:, :-1], zero, 0, skip_head=True)
torch.cuda.synchronize(); print("prefill %d tokens: %.2f ms" % (P - 1, (time.time() - t0) * 1e3), flush=True)
sd.join_current_stream()
for use_graph in (False, True):
    for b in range(B):
        sd.set_row(b, P, int(prompts[b, -2]), int(prompts[b, -1]), True)
    sd.step(use_graph=use_graph); sd.step(use_graph=use_graph); r = sd.sync()
    torch.cuda.synchronize(); t0 = time.time(); acc = 0; new = 0
    for i in range(steps):
        sd.step(use_graph=use_graph); r = sd.sync(); acc += int(r.accept_len.sum()); new += int(r.n_new.sum())
    dt = time.time() - t0
    bytes_step = K * drf.matmul_bytes() + tgt.matmul_bytes()
    print("graph=%s  %.3f ms/step  accept/step=%.2f  tok/s=%.1f  roofline=%.3f of 8TB/s (%.2f GB/step)" % (
        use_graph, dt / steps * 1e3, acc / steps / B, new / dt, bytes_step / (dt / steps) / 8e12, bytes_step / 1e9), flush=True)
print("cur_len", r.cur_len, "last new", r.new_tokens[0])
# per-forward device time via torch graphs (C-ABI launches are captured like any other)
def time_fwd(model, M, name, nbytes):
    toks = torch.randint(4, 1000, (B, M), dtype=torch.int32, device="cuda")
    pos = torch.full((B,), 100, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        model.forward(toks, pos, 0); model.forward(toks, pos, 0)
        st.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            model.forward(toks, pos, 0)
        gr.replay(); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(50): gr.replay()
        e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("%s M=%d: %.3f ms/forward  %.2f TB/s" % (name, M, ms, nbytes / ms / 1e9), flush=True)
time_fwd(dm, 1, "draft 1B", drf.matmul_bytes()); time_fwd(dm, 2, "draft 1B", drf.matmul_bytes())
time_fwd(tm, K + 1, "target 3B", tgt.matmul_bytes()); time_fwd(tm, 1, "target 3B", tgt.matmul_bytes())
