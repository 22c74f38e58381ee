"""`python -m src.specdec.run_specdec` — command line over SpeculativePipeline.generate.

Counterpart of the reference CLI (src/specdec/run_specdec.py:40-283): same options and the same one-line JSON
on stdout (latency_ms, proposed, accepted, acceptance_rate, tokens_per_sec, text, impl, device, base_model,
draft_model, draft_mode, dtype). Differences forced by this build: `--impl` is `hip` (the reference's
`fake`/`hf` have no counterpart: there is no CPU path), models are local checkpoint directories or
`synthetic:<preset>` (nothing is fetched by name), and with synthetic weights the prompt is a list of token
ids ("12 7 99"). Decoding is greedy (`generate(do_sample=True)` is refused, see pipeline.py)."""

from __future__ import annotations

import argparse
import json
import logging
import sys

from .core.pipeline import SpeculativePipeline


def parse_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(description="Speculative decoding on MI355X (HIP path)")
    ap.add_argument("--prompt", type=str, required=True, help="prompt text (token ids for synthetic models)")
    ap.add_argument("--max-tokens", type=int)
    ap.add_argument("--config", type=str, help="YAML configuration (configs/specdec.yaml keys)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--base-model", type=str, help="checkpoint directory or synthetic:<preset>")
    ap.add_argument("--draft-model", type=str)
    ap.add_argument("--max-draft", type=int)
    ap.add_argument("--temperature", type=float)
    ap.add_argument("--seed", type=int)
    ap.add_argument("--device", type=str, choices=["auto", "cuda"], default="auto")
    ap.add_argument("--impl", type=str, choices=["hip"], default="hip")
    ap.add_argument("--draft-mode", type=str, choices=["vanilla", "medusa", "eagle"], default="vanilla")
    ap.add_argument("--policy", type=str, choices=["longest_prefix", "conf_threshold", "topk_agree", "typical"], default="longest_prefix")
    ap.add_argument("--policy-tau", type=float, help="conf_threshold: tau")
    ap.add_argument("--policy-k", type=int, help="topk_agree: k")
    ap.add_argument("--policy-p", type=float, help="typical: p")
    ap.add_argument("--controller", type=str, choices=["fixed", "adaptive"], default="fixed")
    ap.add_argument("--K", type=int, default=4, help="K of the fixed controller")
    ap.add_argument("--adaptive-K", action="store_true")
    ap.add_argument("--min-k", type=int)
    ap.add_argument("--max-k", type=int)
    ap.add_argument("--target-acceptance", type=float)
    ap.add_argument("--per-row-K", action="store_true",
                    help="adaptive controller: every batch row its own K, moved on the device inside the captured step (not in the reference)")
    return ap.parse_args(argv)


def main(argv=None) -> int:
    args = parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.WARNING, stream=sys.stderr)
    raw = list(argv) if argv is not None else sys.argv[1:]
    if any(a == "--K" or a.startswith("--K=") for a in raw) and args.adaptive_K:
        logging.error("Cannot specify both --K and --adaptive-K")
        return 1
    if args.adaptive_K or args.controller == "adaptive":
        controller, cp = "adaptive", {k: v for k, v in (("min_k", args.min_k), ("max_k", args.max_k),
                                                       ("target_acceptance_rate", args.target_acceptance)) if v is not None}
        if args.per_row_K:
            cp["per_row"] = True
            # the device controller requires min_k <= initial_k <= max_k (sd_specdec_set_adaptive): clamp the default of 4
            lo = args.min_k if args.min_k is not None else 1
            hi = args.max_k if args.max_k is not None else max(4, lo)
            cp["initial_k"] = max(lo, min(4, hi))
    else:
        controller, cp = "fixed", {"k": args.K}
    try:
        pipe = SpeculativePipeline(config_path=args.config, base_model=args.base_model, draft_model=args.draft_model,
                                   max_draft=args.max_draft, device=args.device, seed=args.seed, implementation=args.impl,
                                   policy=args.policy, controller=controller, controller_params=cp, draft_mode=args.draft_mode,
                                   policy_params={k: v for k, v in (("tau", args.policy_tau), ("k", args.policy_k), ("p", args.policy_p)) if v is not None})
        r = pipe.generate(prompt=args.prompt, max_tokens=args.max_tokens, temperature=args.temperature, do_sample=False)
    except Exception as e:  # the reference CLI reports and exits 1 (run_specdec.py:276-278)
        logging.error("Error: %s", e)
        return 1
    keys = ("latency_ms", "proposed", "accepted", "acceptance_rate", "tokens_per_sec", "text", "impl", "device",
            "base_model", "draft_model", "draft_mode", "dtype")
    print(json.dumps({k: r[k] for k in keys}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
