#!/usr/bin/env python3
"""K-sweep harness — counterpart of the reference's scripts/comprehensive_k_sweep.py.

Same protocol (comprehensive_k_sweep.py:113-124, 346-355, 444-535, 1016-1060): a 10-prompt suite,
K = 1..max_k, `--iterations` passes, batches of SPECDEC_BATCH_SIZE prompts through
`SpeculativePipeline.generate_batch`, one CSV row per K (same columns) and one JSON with
system info + detailed per-prompt rows. Two deliberate differences:
  * K is actually applied: the pipeline is built with controller_params={"k": K} (the reference
    passes only max_draft=K, which its controller ignores — every published "K-sweep" ran K=4,
    SURVEY §0.5);
  * greedy decoding (do_sample=False, the SPECDEC_DETERMINISTIC configuration) unless --do-sample is given; the
    reference script samples (T=0.7), which on the device is the sampled bonus token of the captured step.
Prompts are tokenised by the model's tokenizer when a checkpoint directory provides one; with
synthetic weights the suite is mapped to seeded token ids of the same lengths.

    python llm-inference-lab_amd/scripts/k_sweep.py --base-model synthetic:llama-3.2-3b \
        --draft-model synthetic:llama-3.2-1b --max-k 4 --max-tokens 64 --iterations 1
"""

from __future__ import annotations

import argparse
import csv
import json
import os
import platform
import sys
import time
from datetime import datetime
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PROMPT_SUITE = [
    "Explain KV cache simply.",
    "What is the capital of France?",
    "Write a short poem about coding.",
    "How does machine learning work?",
    "Describe the process of photosynthesis.",
    "What are the benefits of exercise?",
    "Explain quantum computing basics.",
    "How do neural networks learn?",
    "What is the meaning of life?",
    "Describe a typical day in the life of a programmer.",
]


def system_info(batch_size: int = 1, deterministic: bool = True):
    """Keys of the reference's get_system_info + run metadata (comprehensive_k_sweep.py:130-206, :1006-1013), plus this
    build's own (`hip`, `env`)."""
    from kernels import get_kernel_info

    kinfo = get_kernel_info()
    return {
        "timestamp": datetime.now().isoformat(), "python_version": platform.python_version(), "pytorch_version": torch.__version__,
        "platform": platform.platform(), "device": "cuda", "cuda_available": torch.cuda.is_available(), "mps_available": False,
        "device_name": torch.cuda.get_device_name(0) if torch.cuda.is_available() else None,
        "dtype": "bfloat16", "kernel_backends": kinfo, "kernel_info": kinfo, "deterministic": bool(deterministic),
        "batch_size": int(batch_size), "kv_append_enabled": True, "cuda_graph": True, "parallel_streams": True,
        "hip": torch.version.hip, "env": {k: v for k, v in os.environ.items() if k.startswith("SPECDEC_")},
    }


def prompt_ids(lm, text, idx):
    """Real tokenizer when there is one; else seeded ids with the prompt's word count + 4."""
    from specdec.models.hip_lm import IdTokenizer

    tok = lm.tokenizer
    if not isinstance(tok, IdTokenizer):
        return [int(x) for x in lm.encode(text).flatten().tolist()]
    n = len(text.split()) + 4
    g = torch.Generator().manual_seed(1234 + idx)
    return torch.randint(4, lm.vocab_size, (n,), generator=g).tolist()


def run(args, base=None, draft=None):
    """`base` / `draft`: ready HipLM objects (tests); else built from args.base_model / args.draft_model."""
    from specdec import SpeculativePipeline, create_hip_lm

    base = base or create_hip_lm(args.base_model)
    if draft is None and not args.share_draft_embeddings:
        draft = create_hip_lm(args.draft_model)
    if draft is None:
        from specdec_hip import weights as W
        from specdec.models.hip_lm import HipLM

        presets = {"llama-3.2-1b": W.LLAMA_3_2_1B, "llama-3.2-3b": W.LLAMA_3_2_3B, "llama-3-8b": W.LLAMA_3_8B}
        name = args.draft_model.split(":", 1)[1]
        draft = HipLM(W.synthetic_llama(presets[name], seed=1, device="cuda", embed_from=base.weights,
                                        flip_fraction=args.flip))
    batch = int(os.getenv("SPECDEC_BATCH_SIZE", str(args.batch_size)))
    results, detailed = [], []
    for k in range(1, args.max_k + 1):
        pipe = SpeculativePipeline(base_lm=base, draft_lm=draft, max_draft=k, controller="fixed",
                                   controller_params={"k": k}, seed=1234)
        pipe.generate(prompt_ids(base, "Hello", 99), max_tokens=4, do_sample=False)  # warm-up (:365)
        sample = bool(getattr(args, "do_sample", False))
        k_rows = []
        for it in range(args.iterations):
            for b0 in range(0, len(PROMPT_SUITE), batch):
                idxs = list(range(b0, min(b0 + batch, len(PROMPT_SUITE))))
                prompts = [prompt_ids(base, PROMPT_SUITE[i], i) for i in idxs]
                if args.continuous:   # all prompts at once through `batch` slots (a finished row's slot is re-used)
                    if b0 != 0:
                        continue
                    idxs = list(range(len(PROMPT_SUITE)))
                    prompts = [prompt_ids(base, PROMPT_SUITE[i], i) for i in idxs]
                    outs = pipe.generate_many(prompts, max_tokens=args.max_tokens, batch_size=batch, do_sample=sample)
                    for o in outs:
                        o.setdefault("kv_appended_tokens", o["num_generated"])
                        o.setdefault("kv_append_time_ms", 0.0)
                else:
                    outs = pipe.generate_batch(prompts, max_tokens=args.max_tokens, temperature=0.7, do_sample=sample)
                for i, r in zip(idxs, outs):
                    row = {
                        "k": k, "iteration": it + 1, "prompt_idx": i + 1, "prompt_name": PROMPT_SUITE[i],
                        "prompt": PROMPT_SUITE[i], "prompt_text": PROMPT_SUITE[i], "completion_text": r["text"],
                        "full_text": PROMPT_SUITE[i] + " " + r["text"],
                        "completion_token_count": len(r["generated_tokens"]), "latency_ms": r["latency_ms"],
                        "tokens_per_sec": r["tokens_per_sec"], "acceptance_rate": r["acceptance_rate"],
                        "proposed": r["proposed"], "accepted": r["accepted"],
                        "kv_appended_tokens": r.get("kv_appended_tokens", 0), "kv_append_time_ms": r["kv_append_time_ms"],
                        "kv_append_enabled": r["kv_append_enabled"], "kv_append_backend": r["kv_append_backend"],
                        "text": r["text"][:100], "success": True, "device": "cuda", "dtype": "bfloat16",
                        "batch_size": batch, "generated_tokens": r["generated_tokens"],   # (generated_tokens: extra to the reference's keys)
                    }
                    k_rows.append(row)
                    detailed.append(row)
        lat = [r["latency_ms"] for r in k_rows]
        tps = [r["tokens_per_sec"] for r in k_rows]
        acc = [r["acceptance_rate"] for r in k_rows]
        kva = [r["kv_appended_tokens"] for r in k_rows]
        kvt = [r["kv_append_time_ms"] for r in k_rows]
        prop = [r["proposed"] for r in k_rows]
        accd = [r["accepted"] for r in k_rows]
        results.append({
            "k": k, "n_samples": len(k_rows), "n_failures": 0, "success_rate": 1.0,
            "latency_ms_mean": float(np.mean(lat)), "latency_ms_std": float(np.std(lat)),
            "tokens_per_sec_mean": float(np.mean(tps)), "tokens_per_sec_std": float(np.std(tps)),
            "acceptance_rate_mean": max(0.0, min(1.0, float(np.mean(acc)))), "acceptance_rate_std": float(np.std(acc)),
            "kv_appended_tokens_mean": float(np.mean(kva)), "kv_appended_tokens_std": float(np.std(kva)),
            "kv_append_time_ms_mean": float(np.mean(kvt)), "kv_append_time_ms_std": float(np.std(kvt)),
            "proposed_mean": float(np.mean(prop)), "proposed_std": float(np.std(prop)),
            "accepted_mean": float(np.mean(accd)), "accepted_std": float(np.std(accd)),
            "device": "cuda", "dtype": "bfloat16",
        })
        print(f"[K={k}] {len(k_rows)} samples | {results[-1]['tokens_per_sec_mean']:.1f} tok/s per prompt | "
              f"acceptance {results[-1]['acceptance_rate_mean']:.3f}", flush=True)
    return results, detailed


def write_manifest(out_dir, csv_file, json_file, args) -> Path:
    """MANIFEST.json of a results directory, the keys of the reference's exported runs
    (docs/results/2025-10-30-T4-Phase3D-Run1-32tok-100iter-fp16/MANIFEST.json; layout: docs/progress.md:966-984)."""
    import torch

    dev = torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu"
    man = {
        "export_created_at": datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"),
        "device": dev,
        "dtype": "bfloat16",
        "max_tokens": int(args.max_tokens),
        "iterations_per_k": int(args.iterations),
        "models": {"base": args.base_model, "draft": args.draft_model},
        "artifacts": {"summary_json": Path(json_file).name, "summary_csv": Path(csv_file).name},
        "source_dir": str(Path(out_dir).resolve()),
    }
    path = Path(out_dir) / "MANIFEST.json"
    with open(path, "w") as f:
        json.dump(man, f, indent=2)
    return path


def save(results, detailed, out_dir, batch_size: int = 1):
    ts = datetime.now().strftime("%Y%m%d_%H%M%S")
    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    csv_file, json_file = out / f"specdec_cuda_{ts}.csv", out / f"specdec_cuda_{ts}.json"
    with open(csv_file, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=results[0].keys())
        w.writeheader()
        w.writerows(results)
    with open(json_file, "w") as f:
        json.dump({"system_info": system_info(batch_size), "summary_results": results, "detailed_results": detailed,
                   "detailed_metrics": {}}, f, indent=2)
    return csv_file, json_file


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--base-model", default="synthetic:llama-3.2-3b")
    ap.add_argument("--draft-model", default="synthetic:llama-3.2-1b")
    ap.add_argument("--share-draft-embeddings", action="store_true", default=True,
                    help="synthetic runs: derive the draft's tables from the target's (non-trivial acceptance)")
    ap.add_argument("--flip", type=float, default=0.2)
    ap.add_argument("--max-k", type=int, default=4)
    ap.add_argument("--max-tokens", type=int, default=64)
    ap.add_argument("--iterations", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--continuous", action="store_true",
                    help="continuous batching (generate_many) instead of the reference harness' fixed batches")
    ap.add_argument("--do-sample", action="store_true", help="sampled bonus token (T = 0.7, the reference script's setting) instead of greedy")
    ap.add_argument("--output-dir", default="gpurun_out/k_sweep")
    a = ap.parse_args()
    if not a.draft_model.startswith("synthetic:"):
        a.share_draft_embeddings = False
    t0 = time.time()
    res, det = run(a)
    c, j = save(res, det, a.output_dir, int(os.getenv("SPECDEC_BATCH_SIZE", str(a.batch_size))))
    m = write_manifest(a.output_dir, c, j, a)
    print(f"saved {c}, {j} and {m} in {time.time() - t0:.1f}s")
