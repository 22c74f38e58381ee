"""Generate golden fixtures by importing the REFERENCE (build container only).

    cd /tmp && python /root/repo/tests/golden/make_golden.py

Requires /root/reference (read-only). It never runs on the GPU box; the fixtures it
writes (tests/golden/*.json / *.npz) are data: inputs-by-seed and the reference's
outputs. No reference source text is stored.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SPECDEC_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, REF)

import cases  # noqa: E402


def gen_kernels():
    from kernels.reference import kv_append_ref, kv_append_with_mask_ref, verify_prefix_ref

    out = {"verify": {}, "kv": {}}
    for i, case in enumerate(cases.VERIFY_CASES):
        name = case[0]
        seed = 1000 + i
        logits, ids = cases.build_verify_case(*case, seed=seed)
        # the reference computes on whatever dtype it is given (argmax on bf16 works on CPU)
        alen, mask = verify_prefix_ref(logits, ids)
        pred = torch.argmax(logits, dim=-1)
        out["verify"][name] = {
            "seed": seed,
            "accept_len": alen.tolist(),
            "mask": mask.tolist(),
            "argmax": pred.tolist(),
            "logits_checksum": cases.checksum(logits),
            "ids_checksum": cases.checksum(ids),
        }
    arrays = {}
    for i, case in enumerate(cases.KV_CASES):
        name = case[0]
        seed = 2000 + i
        bk, bv, nk, nv, mask, alen = cases.build_kv_case(*case, seed=seed)
        ok, ov = kv_append_ref(bk, bv, nk, nv)
        mk, mv = kv_append_with_mask_ref(bk, bv, nk, nv, mask, alen)
        out["kv"][name] = {
            "seed": seed,
            "inputs_checksum": [cases.checksum(t) for t in (bk, bv, nk, nv, mask, alen)],
            "concat_checksum": [cases.checksum(ok), cases.checksum(ov)],
            "masked_checksum": [cases.checksum(mk), cases.checksum(mv)],
        }
        # full outputs for the small cases, as float32 arrays (exact for bf16/f16 values)
        if ok.numel() <= 40000:
            arrays[f"{name}/concat_k"] = ok.float().numpy()
            arrays[f"{name}/concat_v"] = ov.float().numpy()
            arrays[f"{name}/masked_k"] = mk.float().numpy()
            arrays[f"{name}/masked_v"] = mv.float().numpy()
    with open(os.path.join(HERE, "kernels_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "kernels_golden.npz"), **arrays)
    print("kernels goldens:", len(out["verify"]), "verify cases,", len(out["kv"]), "kv cases")


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["kernels"]
    if "kernels" in which:
        gen_kernels()
