"""Whole-forward timing of the persistent launch (csrc/persist.hip) against the launch-per-operator forward, plus the
in-kernel timeline of one persistent forward (sd_model_probe_forward).

  python profiles/tools/persist_probe.py [--model 1b|3b] [--layers N] [--tokens M] [--ctx L] [--iters N] [--max-t T]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-lab_amd"))
sys.path.insert(0, ROOT)

from specdec_hip import weights as W  # noqa: E402
from specdec_hip.engine import HipModel  # noqa: E402

SHAPES = {
    "1b": dict(d_model=2048, n_heads=32, n_kv_heads=8, head_dim=64, d_ff=8192, n_layers=16),
    "3b": dict(d_model=3072, n_heads=24, n_kv_heads=8, head_dim=128, d_ff=8192, n_layers=28),
}
KINDS = ["qkv", "out", "gateup", "down"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="1b")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--tokens", type=int, default=1)
    ap.add_argument("--ctx", type=int, default=128)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--max-t", type=int, default=2)
    ap.add_argument("--no-timeline", action="store_true")
    ap.add_argument("--persist-only", action="store_true", help="skip the launch-path reference run")
    ap.add_argument("--diag", action="store_true", help="library built with -DSD_P_DIAG=1: slots 8-11 are the third consumer's cycle counts")
    a = ap.parse_args()
    sh = dict(SHAPES[a.model])
    if a.layers:
        sh["n_layers"] = a.layers
    cfg = W.ModelConfig(arch=W.ARCH_LLAMA, vocab=128256, max_pos=4096, rope_theta=500000.0, tie_embeddings=True, name=a.model, **sh)
    mw = W.random_init(cfg, seed=0, device="cuda")
    res = {}
    for label, max_t in (("persistent", a.max_t),) + (() if a.persist_only else (("launches", 0),)):
        os.environ["SPECDEC_PERSIST_MAX_T"] = str(max_t)
        hm = HipModel(mw, batch=1, l_max=max(512, a.ctx + 64))
        if a.ctx:
            toks = torch.randint(4, cfg.vocab, (1, a.ctx), dtype=torch.int32, device="cuda")
            hm.forward(toks, torch.zeros(1, dtype=torch.int32, device="cuda"), 0, skip_head=True)
        # the probe's tokens sit at positions ctx .. ctx+M-1: the attention reads the ctx cached positions
        us, nbytes, tl = hm.probe_forward(M=a.tokens, iters=a.iters, timeline=(label == "persistent" and not a.no_timeline), pos0=a.ctx)
        st = hm.engine_status()
        res[label] = us
        print(f"{label:11s} persist_tokens={hm.persist_tokens} M={a.tokens}: {us:8.1f} us / forward, {nbytes / 1e6:8.1f} MB -> "
              f"{nbytes / us / 1e6:6.2f} TB/s ({nbytes / us / 1e6 / 8.0:5.3f} of 8 TB/s), status {st}", flush=True)
        if tl is not None and hm.persist_tokens >= a.tokens:
            n_ops = 4 * cfg.n_layers + 1
            t = tl.reshape(256, 12 * n_ops + 4).astype(np.int64)
            ev = t[:, :12 * n_ops].reshape(256, n_ops, 12)
            ck = t[:, 12 * n_ops:]
            mhz = (ck[:, 3] - ck[:, 1]) / np.maximum(ck[:, 2] - ck[:, 0], 1) * 100.0
            print(f"shader clock during the launch: mean {mhz.mean():.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})")
            t0 = ev[:, 0, 0].min()
            us_ = lambda x: (x - t0) / 100.0
            print("timeline, us from the first gather start; mean over 256 CUs. gather = staged - gather_start; "
                  "w3wait = third consumer's start - staged; mfma = its MFMA end - start; lead_mfma = leader MFMA end - staged; "
                  "epi = op_done - leader MFMA end; ld_done = loader done issuing the op")
            print(f"{'op':>10s} {'g_start':>9s} {'staged':>9s} {'op_done':>9s} {'attn':>8s} {'ld_done':>9s} | {'gather':>6s} {'w3wait':>6s} {'mfma':>6s} {'lead_mfma':>9s} {'epi':>6s}   staged min..max")
            tot = {}
            for i in range(n_ops):
                kind = KINDS[i % 4] if i < 4 * cfg.n_layers else "head"
                e = us_(ev[:, i, :])
                at = ev[:, i, 3]
                at = us_(at[at > 0]).mean() if (at > 0).any() else float("nan")
                row = dict(gather=(e[:, 1] - e[:, 0]).mean(), w3wait=(e[:, 4] - e[:, 1]).mean(), mfma=(e[:, 5] - e[:, 4]).mean(),
                           lead_mfma=(e[:, 6] - e[:, 1]).mean(), epi=(e[:, 2] - e[:, 6]).mean(), total=(e[:, 2] - e[:, 0]).mean())
                def d(x, y):   # mean over CUs that have both stamps
                    m = (ev[:, i, x] > 0) & (ev[:, i, y] > 0)
                    return float((ev[m, i, x] - ev[m, i, y]).mean() / 100.0) if m.any() else float("nan")
                if a.diag:
                    row.update(c_pre=ev[:, i, 8].mean(), c_body=ev[:, i, 9].mean(), c_loop=ev[:, i, 10].mean(), c_tail=ev[:, i, 11].mean())
                else:
                    row.update(sweep=d(8, 0), parts=d(9, 1), fin=d(2, 9), a_stage=d(10, 2), a_comp=d(11, 10), a_merge=d(3, 11))
                tot.setdefault(kind, []).append(row)
                if 4 <= i < 12 or i >= n_ops - 1:
                    print(f"{kind + str(i // 4):>10s} {e[:, 0].mean():9.2f} {e[:, 1].mean():9.2f} {e[:, 2].mean():9.2f} {at:8.2f} {e[:, 7].mean():9.2f} | "
                          f"{row['gather']:6.2f} {row['w3wait']:6.2f} {row['mfma']:6.2f} {row['lead_mfma']:9.2f} {row['epi']:6.2f}   {e[:, 1].min():.2f}..{e[:, 1].max():.2f}")
            print("per op kind, mean over layers (us):")
            for k, v in tot.items():
                print(f"  {k:7s} " + "  ".join(f"{name} {np.mean([r[name] for r in v]):6.2f}" for name in ("gather", "w3wait", "mfma", "lead_mfma", "epi", "total")) + f"   (x{len(v)})")
                if a.diag:
                    print(f"          third consumer, shader cycles per op: prologue {np.mean([r['c_pre'] for r in v]):7.0f} | LDS reads + MFMAs "
                          f"{np.mean([r['c_body'] for r in v]):7.0f} | waiting for weights {np.mean([r['c_loop'] for r in v]):7.0f} | rest of the chunk loop + partial + done flag {np.mean([r['c_tail'] for r in v]):7.0f}")
                else:
                    print(f"          sweep returned {np.nanmean([r['sweep'] for r in v]):5.2f} after gather start | partials in {np.nanmean([r['parts'] for r in v]):5.2f} after staged | "
                          f"epilogue+publish {np.nanmean([r['fin'] for r in v]):5.2f} | attention: q staged {np.nanmean([r['a_stage'] for r in v]):5.2f} after op_done, "
                          f"blocks {np.nanmean([r['a_comp'] for r in v]):5.2f}, merge+publish {np.nanmean([r['a_merge'] for r in v]):5.2f}")
            if cfg.n_layers > 2:
                starts = us_(ev[:, 0:4 * cfg.n_layers:4, 0]).mean(0)
                per = np.diff(starts)
                print(f"layer period: mean {per[1:].mean():.2f} us (min {per.min():.2f}, max {per.max():.2f})")
        del hm
    if "persistent" in res and "launches" in res:
        print(f"persistent / launches = {res['persistent'] / res['launches']:.3f}")


if __name__ == "__main__":
    main()
