"""`python -m src.specdec_cli.main {bench,run}` — counterpart of the reference's `specdec` console script
(src/specdec_cli/main.py:1-102): `bench` runs the K-sweep harness and writes its CSV + JSON, `run` decodes one prompt
through `SpeculativePipeline.generate` and prints the device / dtype / kernel backends and the text.

Same sub-commands and options; what differs is forced by this build: the device is always `cuda` (PyTorch-ROCm's name for
the MI355X), models are local checkpoint directories or `synthetic:<preset>` (nothing is fetched by name, so the defaults
are the synthetic Llama-3.2 pair instead of gpt2 / distilgpt2), `--k` is actually applied (the reference passes it as
max_draft, which its controller ignores), and `run --do-sample` is refused by generate() as documented there."""

from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

from kernels import get_kernel_info
from specdec import SpeculativePipeline


def _harness():
    scripts = Path(__file__).resolve().parent.parent.parent / "scripts"
    if str(scripts) not in sys.path:
        sys.path.insert(0, str(scripts))
    import k_sweep  # noqa: E402  (llm-inference-lab_amd/scripts/k_sweep.py)

    return k_sweep


def cmd_bench(args: argparse.Namespace) -> int:
    H = _harness()
    ns = argparse.Namespace(base_model=args.base_model, draft_model=args.draft_model, max_tokens=args.max_tokens,
                            iterations=args.iterations, max_k=args.max_k, batch_size=int(os.getenv("SPECDEC_BATCH_SIZE", "1")),
                            share_draft_embeddings=args.draft_model.startswith("synthetic:"), flip=args.flip, continuous=False,
                            do_sample=False)
    if args.deterministic:
        os.environ["SPECDEC_DETERMINISTIC"] = "1"
    results, detailed = H.run(ns)
    csv_file, json_file = H.save(results, detailed, args.output_dir, ns.batch_size)
    print(f"Results saved to {csv_file} and {json_file}")
    return 0


def cmd_run(args: argparse.Namespace) -> int:
    pipe = SpeculativePipeline(base_model=args.base_model, draft_model=args.draft_model, max_draft=args.k, implementation="hip",
                               device=args.device, controller="fixed", controller_params={"k": args.k}, enable_optimization=True,
                               draft_mode="vanilla")
    res = pipe.generate(prompt=args.prompt, max_tokens=args.max_tokens, temperature=args.temperature, do_sample=args.do_sample)
    kinfo = get_kernel_info()
    print(f"Device: {res.get('device')} | Dtype: {res.get('dtype')} | "
          f"Backends: verify={kinfo.get('verify_backend')}, kv_append={kinfo.get('kv_append_backend')}")
    print(f"Text: {res.get('text', '')}")
    return 0


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="specdec", description="MI355X speculative-decoding CLI")
    sub = p.add_subparsers(dest="cmd", required=True)
    pb = sub.add_parser("bench", help="Run a K-sweep benchmark")
    pb.add_argument("--base-model", default="synthetic:llama-3.2-3b")
    pb.add_argument("--draft-model", default="synthetic:llama-3.2-1b")
    pb.add_argument("--max-tokens", type=int, default=32)
    pb.add_argument("--iterations", type=int, default=10)
    pb.add_argument("--max-k", type=int, default=4)
    pb.add_argument("--flip", type=float, default=0.2, help="synthetic pairs: fraction of tokens whose draft successor differs")
    pb.add_argument("--device", choices=["auto", "cuda"], default="auto")
    pb.add_argument("--output-dir", type=Path, default=Path("results"))
    pb.add_argument("--deterministic", action="store_true")
    pb.set_defaults(func=cmd_bench)
    pr = sub.add_parser("run", help="Run a single prompt via SpeculativePipeline")
    pr.add_argument("--base-model", default="synthetic:llama-3.2-3b")
    pr.add_argument("--draft-model", default="synthetic:llama-3.2-1b")
    pr.add_argument("--k", type=int, default=2)
    pr.add_argument("--max-tokens", type=int, default=32)
    pr.add_argument("--device", choices=["auto", "cuda"], default="auto")
    pr.add_argument("--temperature", type=float, default=0.7)
    pr.add_argument("--do-sample", action="store_true")
    pr.add_argument("prompt", type=str)
    pr.set_defaults(func=cmd_run)
    return p


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    raise SystemExit(args.func(args))


if __name__ == "__main__":
    main()
