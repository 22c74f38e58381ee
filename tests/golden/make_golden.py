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
OUT = os.environ.get("SPECDEC_GOLDEN_OUT", HERE)   # where the fixtures are written (a test regenerates them elsewhere)
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
    with open(os.path.join(OUT, "kernels_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(OUT, "kernels_golden.npz"), **arrays)
    print("kernels goldens:", len(out["verify"]), "verify cases,", len(out["kv"]), "kv cases")


def _tiny_hf_models():
    """Tiny random HF models (fp32) — the third-party forward the reference calls."""
    import transformers

    torch.manual_seed(1234)
    lcfg = transformers.LlamaConfig(
        vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
        num_attention_heads=2, num_key_value_heads=1, head_dim=32, max_position_embeddings=256,
        rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=False,
        rope_scaling={"factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                      "original_max_position_embeddings": 64, "rope_type": "llama3"},
        attention_bias=False, mlp_bias=False,
    )
    llama = transformers.LlamaForCausalLM(lcfg).eval()
    gcfg = transformers.GPT2Config(vocab_size=96, n_positions=128, n_embd=64, n_layer=2, n_head=2,
                                   activation_function="gelu_new", resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    gpt2 = transformers.GPT2LMHeadModel(gcfg).eval()
    # default init leaves biases at zero: randomise them so the bias paths are pinned too
    with torch.no_grad():
        for n, p_ in gpt2.named_parameters():
            if n.endswith("bias"):
                p_.normal_(0.0, 0.1)
    return (lcfg, llama), (gcfg, gpt2)


def gen_hf():
    """Pin for oracle/model_ref.py: logits of transformers' own Llama / GPT-2 forward."""
    import transformers

    (lcfg, llama), (gcfg, gpt2) = _tiny_hf_models()
    rng = np.random.default_rng(77)
    for name, cfg, model in (("llama", lcfg, llama), ("gpt2", gcfg, gpt2)):
        toks = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=(2, 70), dtype=np.int64))
        with torch.no_grad():
            logits = model(input_ids=toks).logits.float()
        arrays = {"__tokens": toks.numpy(), "__logits": logits.numpy()}
        for k, v in model.state_dict().items():
            arrays[k] = v.detach().float().numpy()
        np.savez_compressed(os.path.join(OUT, f"hf_{name}_tiny.npz"), **arrays)
        meta = {"transformers": transformers.__version__, "torch": torch.__version__,
                "config": {k: v for k, v in cfg.to_dict().items()
                           if isinstance(v, (int, float, str, bool, dict, type(None)))}}
        with open(os.path.join(OUT, f"hf_{name}_tiny.json"), "w") as f:
            json.dump(meta, f, indent=1, default=str)
        print("hf golden:", name, tuple(logits.shape))


def _save_local_hf_llama(mw, path):
    """Write a ModelWeights (Llama layout) as a local HF checkpoint dir + a 1:1 tokenizer
    ("t<i>" <-> id i), so the reference's HFWrapper can load it by PATH (hf_wrappers.py:87,115)."""
    import transformers
    from tokenizers import Tokenizer, models, pre_tokenizers

    c = mw.config
    hcfg = transformers.LlamaConfig(
        vocab_size=c.vocab, hidden_size=c.d_model, intermediate_size=c.d_ff, num_hidden_layers=c.n_layers,
        num_attention_heads=c.n_heads, num_key_value_heads=c.n_kv_heads, head_dim=c.head_dim,
        max_position_embeddings=c.max_pos, rms_norm_eps=c.norm_eps, rope_theta=c.rope_theta,
        rope_scaling=c.rope_scaling, tie_word_embeddings=False, attention_bias=False, mlp_bias=False,
        bos_token_id=1, eos_token_id=c.eos_token_id if c.eos_token_id is not None else 2, pad_token_id=0,
    )
    model = transformers.LlamaForCausalLM(hcfg)
    sd = {"model.embed_tokens.weight": mw.tok_emb, "lm_head.weight": mw.lm_head, "model.norm.weight": mw.final_norm_w}
    Hq, Hkv, D, ff = c.n_heads, c.n_kv_heads, c.head_dim, c.d_ff
    for i, l in enumerate(mw.layers):
        b = f"model.layers.{i}."
        sd[b + "input_layernorm.weight"] = l.attn_norm_w
        sd[b + "self_attn.q_proj.weight"] = l.wqkv[: Hq * D]
        sd[b + "self_attn.k_proj.weight"] = l.wqkv[Hq * D : (Hq + Hkv) * D]
        sd[b + "self_attn.v_proj.weight"] = l.wqkv[(Hq + Hkv) * D :]
        sd[b + "self_attn.o_proj.weight"] = l.wo
        sd[b + "post_attention_layernorm.weight"] = l.mlp_norm_w
        sd[b + "mlp.gate_proj.weight"] = l.w_up[:ff]
        sd[b + "mlp.up_proj.weight"] = l.w_up[ff:]
        sd[b + "mlp.down_proj.weight"] = l.w_down
    missing = model.load_state_dict({k: v.float().clone() for k, v in sd.items()}, strict=False)
    assert not [m for m in missing.missing_keys if "rotary" not in m], missing
    model.float().save_pretrained(path, safe_serialization=True)
    vocab = {f"t{i:03d}": i for i in range(c.vocab)}  # fixed width: no special token is a prefix of another
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="t000"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = transformers.PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="t000", pad_token="t000",
                                                bos_token="t001", eos_token=f"t{hcfg.eos_token_id:03d}")
    fast.save_pretrained(path)


def _reference_pipeline_class():
    """`SpeculativePipeline` of the REFERENCE, never this repo's: the product ships packages of the same names
    (`specdec`, `kernels`, `src`) under llm-inference-lab_amd/, so that directory goes to the END of sys.path (it is only
    needed for the weight builders in `specdec_hip`, a name the reference does not have), every already-imported
    `specdec*` / `kernels*` / `src*` module is dropped, REF/src and REF lead the path, and the origin of what was
    imported is asserted before anything is generated from it."""
    pkg = os.path.join(os.path.dirname(os.path.dirname(HERE)), "llm-inference-lab_amd")
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != pkg]
    sys.path.append(pkg)
    for m in [m for m in sys.modules if m.split(".")[0] in ("specdec", "kernels", "src", "scheduler")]:
        del sys.modules[m]
    for p in (REF, os.path.join(REF, "src")):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    from specdec import SpeculativePipeline

    for name in ("specdec", "kernels"):
        origin = os.path.abspath(sys.modules[name].__file__)
        assert origin.startswith(os.path.abspath(REF) + os.sep), f"{name} was imported from {origin}, not from the reference"
    return SpeculativePipeline


def gen_pipeline():
    """G8: traces of the REFERENCE SpeculativePipeline on local tiny Llama pairs (CPU, fp32,
    greedy, KV append off — the configuration of its published runs)."""
    import shutil
    import tempfile

    os.environ["SPECDEC_ENABLE_KV_APPEND"] = "0"
    os.environ["SPECDEC_DETERMINISTIC"] = "1"
    SpeculativePipeline = _reference_pipeline_class()

    pairs = cases.g8_pairs(torch.float32)
    tcfg = pairs["structured"][1].config

    out = {}
    arrays = {}
    tmp = tempfile.mkdtemp(prefix="g8_")
    try:
        for pname, (d, t) in pairs.items():
            ddir, tdir = os.path.join(tmp, pname + "_draft"), os.path.join(tmp, pname + "_target")
            _save_local_hf_llama(d, ddir)
            _save_local_hf_llama(t, tdir)
            out[pname] = {"draft_checksum": cases.weights_checksum(d), "target_checksum": cases.weights_checksum(t),
                          "runs": []}
            rng = np.random.default_rng(5)
            for k in (1, 2, 4):
                pipe = SpeculativePipeline(base_model=tdir, draft_model=ddir, implementation="hf", device="cpu",
                                           controller="fixed", controller_params={"k": k}, max_draft=k, seed=1234)
                for max_tokens, plen in ((12, 6), (20, 9)):
                    prompt_ids = rng.integers(4, tcfg.vocab, size=plen).tolist()
                    prompt = " ".join(f"t{i:03d}" for i in prompt_ids)
                    rb = pipe.generate_batch([prompt], max_tokens=max_tokens, temperature=0.7, do_sample=False)[0]
                    rs = pipe.generate(prompt, max_tokens=max_tokens, temperature=0.7, do_sample=False)
                    out[pname]["runs"].append({
                        "k": k, "max_tokens": max_tokens, "prompt_ids": prompt_ids,
                        "batch": {"generated_tokens": [int(x) for x in rb["generated_tokens"]],
                                  "proposed": int(rb["proposed"]), "accepted": int(rb["accepted"]),
                                  "steps": int(rb["batch_metrics"]["total_steps"])},
                        "single": {"generated_tokens": [int(x) for x in rs["generated_tokens"]] if "generated_tokens" in rs else None,
                                   "text": rs.get("text"), "proposed": int(rs["proposed"]), "accepted": int(rs["accepted"]),
                                   "steps": int(rs["steps"])},
                    })
                    print(pname, "k", k, "batch", rb["generated_tokens"], "acc", rb["accepted"], "/", rb["proposed"],
                          "| single", rs.get("text", "")[:60], rs["accepted"], "/", rs["proposed"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "pipeline_golden.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)


def gen_pipeline_eagle():
    """Traces of the REFERENCE pipeline with draft_mode="eagle" (its HF path, _run_eagle_hf, pipeline.py:765-889) on the
    G8 target models: generate(), CPU fp32, greedy, KV append off. The reference keeps the extrapolation state on the
    pipeline object, so every run uses a fresh pipeline."""
    import shutil
    import tempfile

    os.environ["SPECDEC_ENABLE_KV_APPEND"] = "0"
    os.environ["SPECDEC_DETERMINISTIC"] = "1"
    SpeculativePipeline = _reference_pipeline_class()
    pairs = cases.g8_pairs(torch.float32)
    tcfg = pairs["structured"][1].config
    out = {}
    tmp = tempfile.mkdtemp(prefix="g8e_")
    try:
        for pname, (d, t) in pairs.items():
            ddir, tdir = os.path.join(tmp, pname + "_draft"), os.path.join(tmp, pname + "_target")
            _save_local_hf_llama(d, ddir)
            _save_local_hf_llama(t, tdir)
            out[pname] = {"target_checksum": cases.weights_checksum(t), "runs": []}
            rng = np.random.default_rng(17)
            for k in (1, 2, 4):
                for max_tokens, plen in ((10, 6), (16, 9)):
                    prompt_ids = rng.integers(4, tcfg.vocab, size=plen).tolist()
                    prompt = " ".join(f"t{i:03d}" for i in prompt_ids)
                    pipe = SpeculativePipeline(base_model=tdir, draft_model=ddir, implementation="hf", device="cpu",
                                               controller="fixed", controller_params={"k": k}, max_draft=k, seed=1234,
                                               draft_mode="eagle")
                    rs = pipe.generate(prompt, max_tokens=max_tokens, temperature=0.7, do_sample=False)
                    ecfg = pipe.config.get("eagle", {})
                    out[pname]["runs"].append({
                        "k": k, "alpha": float(ecfg.get("alpha", 0.7)), "max_draft": int(ecfg.get("max_draft", 2)),
                        "max_tokens": max_tokens, "prompt_ids": prompt_ids,
                        "single": {"generated_tokens": [int(x) for x in rs["generated_tokens"]] if "generated_tokens" in rs else None,
                                   "text": rs.get("text"), "proposed": int(rs["proposed"]), "accepted": int(rs["accepted"]),
                                   "steps": int(rs["steps"])},
                    })
                    print(pname, "eagle k", k, rs.get("text", "")[:70], rs["accepted"], "/", rs["proposed"], "steps", rs["steps"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "pipeline_eagle_golden.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)


def gen_medusa():
    """Medusa-lite as the reference has it, on the G8 target models (CPU fp32, KV append off):
      draftor  — `MedusaDraftor` (src/specdec/modes/medusa.py:17-186, head_init="tie"): its proposals for a prompt at
                 temperature 1e-6 (softmax(logits / T) is one-hot: the greedy limit, which pins the oracle's
                 draft_mode="medusa_tied") and at temperature 0.7 under torch.manual_seed (pins the draw order);
      pipeline — `SpeculativePipeline(draft_mode="medusa").generate()` = _run_medusa_hf (pipeline.py:655-763): fresh
                 nn.Linear heads with normal_(0, 0.02) weights and multinomial draws from the GLOBAL torch generator on
                 every step; torch.manual_seed(run seed) is called right before generate() so that a restatement can
                 replay the generator."""
    import shutil
    import tempfile

    import transformers

    os.environ["SPECDEC_ENABLE_KV_APPEND"] = "0"
    os.environ["SPECDEC_DETERMINISTIC"] = "1"
    SpeculativePipeline = _reference_pipeline_class()
    from specdec.modes.medusa import MedusaDraftor

    assert os.path.abspath(sys.modules["specdec.modes.medusa"].__file__).startswith(os.path.abspath(REF) + os.sep)
    pairs = cases.g8_pairs(torch.float32)
    tcfg = pairs["structured"][1].config
    out = {}
    tmp = tempfile.mkdtemp(prefix="g8m_")
    try:
        for pname, (d, t) in pairs.items():
            ddir, tdir = os.path.join(tmp, pname + "_draft"), os.path.join(tmp, pname + "_target")
            _save_local_hf_llama(d, ddir)
            _save_local_hf_llama(t, tdir)
            out[pname] = {"target_checksum": cases.weights_checksum(t), "draftor": [], "pipeline": []}
            model = transformers.AutoModelForCausalLM.from_pretrained(tdir).eval()
            tok = transformers.AutoTokenizer.from_pretrained(tdir)
            rng = np.random.default_rng(23)
            for k in (1, 2, 4):
                for plen in (5, 11):
                    prompt_ids = rng.integers(4, tcfg.vocab, size=plen).tolist()
                    ids = torch.tensor([prompt_ids], dtype=torch.int64)
                    greedy = MedusaDraftor(model, tok, num_heads=k, head_init="tie", temperature=1e-6, device="cpu")
                    g_ids, _, _ = greedy.generate_tokens(ids, k)
                    seed = 31000 + 10 * k + plen
                    sampled = MedusaDraftor(model, tok, num_heads=k, head_init="tie", temperature=0.7, device="cpu")
                    torch.manual_seed(seed)
                    s_ids, s_logits, _ = sampled.generate_tokens(ids, k)
                    out[pname]["draftor"].append({
                        "k": k, "prompt_ids": prompt_ids, "vocab_size": int(tok.vocab_size), "greedy_limit": g_ids[0].tolist(),
                        "seed": seed, "temperature": 0.7, "sampled": s_ids[0].tolist(),
                        "head0_logits_checksum": cases.checksum(s_logits)})
            rng = np.random.default_rng(29)
            for k in (1, 2, 4):
                for max_tokens, plen in ((10, 6), (14, 9)):
                    prompt_ids = rng.integers(4, tcfg.vocab, size=plen).tolist()
                    prompt = " ".join(f"t{i:03d}" for i in prompt_ids)
                    pipe = SpeculativePipeline(base_model=tdir, draft_model=ddir, implementation="hf", device="cpu",
                                               controller="fixed", controller_params={"k": k}, max_draft=k, seed=1234,
                                               draft_mode="medusa")
                    seed = 32000 + 100 * k + max_tokens
                    torch.manual_seed(seed)
                    rs = pipe.generate(prompt, max_tokens=max_tokens, temperature=0.7, do_sample=False)
                    mcfg = pipe.config.get("medusa", {})
                    out[pname]["pipeline"].append({
                        "k": k, "num_heads": int(mcfg.get("num_heads", 2)), "temperature": 0.7, "seed": seed,
                        "max_tokens": max_tokens, "prompt_ids": prompt_ids,
                        "single": {"generated_tokens": [int(x) for x in rs["generated_tokens"]] if "generated_tokens" in rs else None,
                                   "text": rs.get("text"), "proposed": int(rs["proposed"]), "accepted": int(rs["accepted"]),
                                   "steps": int(rs["steps"])}})
                    print(pname, "medusa k", k, rs.get("text", "")[:70], rs["accepted"], "/", rs["proposed"], "steps", rs["steps"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "medusa_golden.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)


def gen_pipeline_policies():
    """Traces of the REFERENCE pipeline with the logit-threshold acceptance policies (policies.py:213-396; called at
    pipeline.py:1092 / :3018) on a pair with soft output distributions (cases.policy_pair): generate_batch and generate,
    CPU fp32, greedy, KV append off."""
    import shutil
    import tempfile

    os.environ["SPECDEC_ENABLE_KV_APPEND"] = "0"
    os.environ["SPECDEC_DETERMINISTIC"] = "1"
    SpeculativePipeline = _reference_pipeline_class()
    d, t = cases.policy_pair(torch.float32)
    out = {"draft_checksum": cases.weights_checksum(d), "target_checksum": cases.weights_checksum(t), "runs": []}
    tmp = tempfile.mkdtemp(prefix="g8p_")
    try:
        ddir, tdir = os.path.join(tmp, "draft"), os.path.join(tmp, "target")
        _save_local_hf_llama(d, ddir)
        _save_local_hf_llama(t, tdir)
        rng = np.random.default_rng(41)
        for name, params in cases.POLICY_RUNS:
            for k in (2, 4):
                pipe = SpeculativePipeline(base_model=tdir, draft_model=ddir, implementation="hf", device="cpu", policy=name,
                                           policy_params=params, controller="fixed", controller_params={"k": k}, max_draft=k, seed=1234)
                prompt_ids = rng.integers(4, t.config.vocab, size=7).tolist()
                prompt = " ".join(f"t{i:03d}" for i in prompt_ids)
                rb = pipe.generate_batch([prompt], max_tokens=14, temperature=0.7, do_sample=False)[0]
                rs = pipe.generate(prompt, max_tokens=14, temperature=0.7, do_sample=False)
                out["runs"].append({
                    "policy": name, "params": params, "k": k, "max_tokens": 14, "prompt_ids": prompt_ids,
                    "batch": {"generated_tokens": [int(x) for x in rb["generated_tokens"]], "proposed": int(rb["proposed"]),
                              "accepted": int(rb["accepted"]), "steps": int(rb["batch_metrics"]["total_steps"])},
                    "single": {"generated_tokens": [int(x) for x in rs["generated_tokens"]] if "generated_tokens" in rs else None,
                               "text": rs.get("text"), "proposed": int(rs["proposed"]), "accepted": int(rs["accepted"]), "steps": int(rs["steps"])}})
                print(name, params, "k", k, "batch acc", rb["accepted"], "/", rb["proposed"], "| single", rs["accepted"], "/", rs["proposed"], rs.get("text", "")[:50])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "pipeline_policies_golden.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)


def gen_harness():
    """Result-file schema of the reference's K-sweep harness (scripts/comprehensive_k_sweep.py:209-1060): the harness itself
    is run on the local tiny G8 pair (CPU, 6 new tokens, 1 iteration, K = 1..2) and the KEYS and value types of what it
    returns and writes are stored — summary rows (one per K: the CSV header), detailed rows (one per prompt), the JSON
    top level and `system_info`. Values are not stored (timings)."""
    import shutil
    import tempfile

    os.environ["SPECDEC_ENABLE_KV_APPEND"] = "0"
    os.environ["SPECDEC_DETERMINISTIC"] = "1"
    os.environ.pop("SPECDEC_DRY_RUN", None)
    _reference_pipeline_class()
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import comprehensive_k_sweep as H

    assert os.path.abspath(H.__file__).startswith(os.path.abspath(REF) + os.sep)
    d, t = cases.g8_pairs(torch.float32)["structured"]
    tmp = tempfile.mkdtemp(prefix="g8h_")
    cwd = os.getcwd()
    try:
        ddir, tdir = os.path.join(tmp, "draft"), os.path.join(tmp, "target")
        _save_local_hf_llama(d, ddir)
        _save_local_hf_llama(t, tdir)
        os.chdir(tmp)
        results, detailed, meta = H.run_comprehensive_k_sweep(base_model=tdir, draft_model=ddir, max_tokens=6, iterations=1, device="cpu",
                                                               deterministic=True, max_k=2)
        sysinfo = H.get_system_info("cpu")
        sysinfo.update(meta)
        csv_file, json_file = H.save_results(results, detailed, sysinfo, os.path.join(tmp, "out"), "cpu")
        with open(json_file) as f:
            written = json.load(f)
        with open(csv_file) as f:
            header = f.readline().strip().split(",")

        def kinds(row):
            return {k: type(v).__name__ for k, v in row.items()}

        out = {"summary_row": kinds(results[0]), "detailed_row": kinds(detailed[0]), "n_summary_rows": len(results),
               "n_detailed_rows": len(detailed), "csv_header": header, "json_top_level": sorted(written.keys()),
               "system_info_keys": sorted(written["system_info"].keys()),
               "file_name_patterns": [os.path.basename(str(csv_file))[:12] + "<timestamp>.csv", os.path.basename(str(json_file))[:12] + "<timestamp>.json"],
               "prompt_suite": list(H.PROMPT_SUITE) if hasattr(H, "PROMPT_SUITE") else sorted({r.get("prompt", "") for r in detailed})}
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "harness_schema_golden.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)
    print("harness schema:", len(out["summary_row"]), "summary keys,", len(out["detailed_row"]), "detailed keys")


def gen_hostlogic():
    """G3-G6: the reference's host-side pieces on seeded inputs (policies, bonus-token
    filtering, controllers, sequence utils, token validation)."""
    from specdec.core.pipeline import sample_bonus_token_from_logits
    from specdec.core.sequence_utils import create_position_ids, pad_sequences
    from specdec.policies.controllers import create_controller
    from specdec.policies.policies import create_policy
    from specdec.utils.token_validation import validate_and_clamp_tokens

    rng = np.random.default_rng(2024)
    out = {"policies": [], "controllers": [], "bonus": [], "sequences": [], "clamp": []}
    for case in range(24):
        K, V = int(rng.integers(1, 7)), int(rng.integers(20, 400))
        seed = 9000 + case
        dl, bl, d_ids, b_ids = cases.build_policy_case(K, V, seed)
        row = {"seed": seed, "K": K, "V": V}
        for name, kw in (("longest_prefix", {}), ("conf_threshold", {"tau": 0.3}), ("topk_agree", {"k": 3}),
                         ("typical", {"p": 0.2})):
            pol = create_policy(name, **kw)
            a_logits, _ = pol.accept_tokens(d_ids, b_ids, dl, bl)
            a_ids, _ = pol.accept_tokens(d_ids, b_ids)  # no logits: id comparison / fallback
            row[name] = [int(a_logits), int(a_ids)]
        out["policies"].append(row)
    for case in range(6):
        params = {"initial_k": int(rng.integers(1, 6)), "min_k": 1, "max_k": int(rng.integers(4, 9)),
                  "step_size": int(rng.integers(1, 3)), "window_size": int(rng.integers(4, 12)),
                  "target_acceptance_rate": float(rng.uniform(0.3, 0.8))}
        ctl = create_controller("adaptive", **params)
        rates = rng.uniform(0, 1, size=40).round(3).tolist()
        ks = [ctl.get_k(i, {"acceptance_rate": r} if i % 7 else {}) for i, r in enumerate(rates)]
        out["controllers"].append({"params": params, "rates": rates, "ks": [int(k) for k in ks],
                                   "info_recent": ctl.get_info()["recent_acceptance_rate"]})
    for case in range(12):
        V = int(rng.integers(30, 300))
        seed = 9500 + case
        logits = torch.from_numpy(np.random.default_rng(seed).standard_normal(V).astype(np.float32)) * 3
        top_k = int(rng.integers(0, 20))
        top_p = float(rng.choice([1.0, 0.9, 0.5, 0.2]))
        temp = float(rng.choice([1.0, 0.7, 0.1]))
        tok = sample_bonus_token_from_logits(logits, temp, False, top_p=top_p, top_k=top_k if top_k else None, vocab_size=V)
        out["bonus"].append({"seed": seed, "V": V, "top_k": top_k, "top_p": top_p, "temperature": temp,
                             "greedy_token": int(tok[0])})
    for case in range(8):
        n = int(rng.integers(1, 6))
        lens = [int(x) for x in rng.integers(1, 9, size=n)]
        seqs = [torch.from_numpy(rng.integers(1, 50, size=L, dtype=np.int64)) for L in lens]
        batch, mask, ol = pad_sequences(seqs, 0, torch.device("cpu"))
        pos = create_position_ids(ol, batch.shape[1], torch.device("cpu"))
        out["sequences"].append({"seqs": [s_.tolist() for s_ in seqs], "batch": batch.tolist(), "mask": mask.tolist(),
                                 "lengths": ol, "position_ids": pos.tolist()})
    for case in range(6):
        V = int(rng.integers(10, 100))
        ids = torch.from_numpy(rng.integers(-5, V + 5, size=(2, 7), dtype=np.int64))
        out["clamp"].append({"V": V, "ids": ids.tolist(), "out": validate_and_clamp_tokens(ids, V, "g").tolist()})
    # sampled bonus token (do_sample=True): empirical histogram of the reference function's own draws,
    # which pins the kept set exactly (tokens never drawn outside it) and the probabilities statistically
    out["sampling"] = []
    srng = np.random.default_rng(77)
    for case in range(14):
        V = int(srng.integers(40, 600))
        seed = 9700 + case
        scale = float(srng.choice([1.0, 3.0, 6.0]))
        logits = torch.from_numpy(np.random.default_rng(seed).standard_normal(V).astype(np.float32)) * scale
        if case % 5 == 4:   # bf16-like ties around the cut
            logits = logits.to(torch.bfloat16).float()
        top_k = int(srng.choice([1, 3, 8, 20, 50, 50, 64]))
        top_p = float(srng.choice([1.0, 0.95, 0.9, 0.9, 0.6, 0.3]))
        temp = float(srng.choice([1.0, 0.7, 0.7, 1.5, 0.2]))
        n_draws = 4000
        torch.manual_seed(4242 + case)
        counts = {}
        for _ in range(n_draws):
            t = int(sample_bonus_token_from_logits(logits, temp, True, top_p=top_p, top_k=top_k, vocab_size=V)[0])
            counts[t] = counts.get(t, 0) + 1
        out["sampling"].append({"seed": seed, "V": V, "scale": scale, "bf16": case % 5 == 4, "top_k": top_k, "top_p": top_p,
                                "temperature": temp, "n_draws": n_draws, "counts": {str(k): v for k, v in sorted(counts.items())}})
    # full-vocabulary nucleus (top_k = None, top_p < 1: pipeline.py:105-125 over the whole sorted row)
    out["sampling_nucleus"] = []
    nrng = np.random.default_rng(91)
    for case in range(8):
        V = int(nrng.choice([300, 700, 1500, 2600]))
        seed = 9900 + case
        scale = float(nrng.choice([1.0, 3.0, 6.0]))
        logits = torch.from_numpy(np.random.default_rng(seed).standard_normal(V).astype(np.float32)) * scale
        if case % 4 == 3:
            logits = logits.to(torch.bfloat16).float()
        top_p = float(nrng.choice([0.95, 0.9, 0.6, 0.3]))
        temp = float(nrng.choice([1.0, 0.7, 1.5]))
        n_draws = 4000
        torch.manual_seed(5151 + case)
        counts = {}
        for _ in range(n_draws):
            t = int(sample_bonus_token_from_logits(logits, temp, True, top_p=top_p, top_k=None, vocab_size=V)[0])
            counts[t] = counts.get(t, 0) + 1
        out["sampling_nucleus"].append({"seed": seed, "V": V, "scale": scale, "bf16": case % 4 == 3, "top_p": top_p, "temperature": temp,
                                        "n_draws": n_draws, "counts": {str(k): v for k, v in sorted(counts.items())}})
    with open(os.path.join(OUT, "hostlogic_golden.json"), "w") as f:
        json.dump(out, f)
    print("hostlogic goldens:", {k: len(v) for k, v in out.items()})


def gen_fake():
    """G7: the REFERENCE pipeline exactly as its own configs/specdec.yaml configures it (implementation: fake — the FakeLM test
    double, src/specdec/models/fake_lm.py:56-106 — base gpt2 / draft distilgpt2 by name, max_draft 4, seed 1234), greedy runs of
    generate() and generate_batch(), plus the double's own token function on fixed id rows. FakeLM.encode hashes the TEXT
    (PYTHONHASHSEED-dependent; the generator is run with PYTHONHASHSEED=0), so the prompts' ids are stored and the product is
    fed ids; the token function hashes a tuple of ints, which CPython does not randomise."""
    SpeculativePipeline = _reference_pipeline_class()
    cfg = os.path.join(REF, "configs", "specdec.yaml")
    out = {"config": "configs/specdec.yaml of the reference, unchanged", "runs": [], "token_function": []}
    for prompt, n in (("Hello world", 8), ("The quick brown fox", 12), ("a", 5)):
        pipe = SpeculativePipeline(config_path=cfg)
        ids = [int(x) for x in pipe.base_lm.encode(prompt)[0].tolist()]
        rs = pipe.generate(prompt, max_tokens=n, do_sample=False)
        pipe = SpeculativePipeline(config_path=cfg)
        rb = pipe.generate_batch([prompt], max_tokens=n, do_sample=False)[0]
        out["runs"].append({"prompt": prompt, "prompt_ids": ids, "max_tokens": n,
                            "single": {"generated_tokens": [int(x) for x in rs["generated_tokens"]], "proposed": int(rs["proposed"]),
                                       "accepted": int(rs["accepted"]), "steps": int(rs["steps"])},
                            "batch": {"generated_tokens": [int(x) for x in rb["generated_tokens"]], "proposed": int(rb["proposed"]),
                                      "accepted": int(rb["accepted"]), "steps": int(rb["steps"])}})   # (FakeLM has no tokenizer: generate_batch falls back to generate() per prompt)
        print("fake", repr(prompt), ids, "->", rs["generated_tokens"], rs["accepted"], "/", rs["proposed"], "| batch", rb["generated_tokens"],
              rb["accepted"], "/", rb["proposed"])
    from specdec.models.fake_lm import FakeLM

    lm = FakeLM(model_name="fake-base-gpt2", vocab_size=1000, seed=1234)
    info = lm.get_tokenizer_info()
    out["tokenizer_info"] = {k: (int(v) if isinstance(v, int) else v) for k, v in info.items()}
    for row, k in (([5, 17, 900], 6), ([1, 2, 3, 4], 4), ([999], 3), (list(range(40, 72)), 8), ([0, 0, 0], 12)):
        toks, logits = lm.generate_tokens(torch.tensor([row]), k)
        out["token_function"].append({"input_ids": row, "k": k, "tokens": [int(x) for x in toks[0].tolist()], "logits_shape": list(logits.shape)})
    with open(os.path.join(OUT, "fake_pipeline_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("fake goldens:", len(out["runs"]), "runs,", len(out["token_function"]), "token-function rows")


def gen_pipeline_sampled():
    """generate(do_sample=True) of the REFERENCE (pipeline.py:893-1413) on the G8 tiny Llama pairs, CPU, fp32: the draft's
    proposals are drawn by torch.multinomial on the global CPU generator (hf_wrappers.py:699-716 / :779-785, seeded here right
    before each call), verification is greedy (speculative_scheduler.py). Stored: the seed, the emitted tokens and the
    counters the draws decide."""
    import shutil
    import tempfile

    os.environ["SPECDEC_ENABLE_KV_APPEND"] = "0"
    os.environ["SPECDEC_DETERMINISTIC"] = "1"
    SpeculativePipeline = _reference_pipeline_class()
    pairs = cases.g8_pairs(torch.float32)
    tcfg = pairs["structured"][1].config
    out = {}
    tmp = tempfile.mkdtemp(prefix="g8s_")
    try:
        for pname, (d, t) in pairs.items():
            ddir, tdir = os.path.join(tmp, pname + "_draft"), os.path.join(tmp, pname + "_target")
            _save_local_hf_llama(d, ddir)
            _save_local_hf_llama(t, tdir)
            out[pname] = {"draft_checksum": cases.weights_checksum(d), "target_checksum": cases.weights_checksum(t), "runs": []}
            rng = np.random.default_rng(7)
            for k in (2, 4):
                pipe = SpeculativePipeline(base_model=tdir, draft_model=ddir, implementation="hf", device="cpu",
                                           controller="fixed", controller_params={"k": k}, max_draft=k, seed=1234)
                for seed, temperature, max_tokens, plen in ((11, 0.7, 12, 6), (12, 12.0, 16, 9), (13, 30.0, 12, 5)):
                    prompt_ids = rng.integers(4, tcfg.vocab, size=plen).tolist()
                    prompt = " ".join(f"t{i:03d}" for i in prompt_ids)
                    torch.manual_seed(seed)
                    rs = pipe.generate(prompt, max_tokens=max_tokens, temperature=temperature, do_sample=True)
                    rg = pipe.generate(prompt, max_tokens=max_tokens, temperature=temperature, do_sample=False)
                    out[pname]["runs"].append({
                        "k": k, "max_tokens": max_tokens, "prompt_ids": prompt_ids, "seed": seed, "temperature": temperature,
                        "sampled": {"generated_tokens": [int(x) for x in rs["generated_tokens"]], "proposed": int(rs["proposed"]),
                                    "accepted": int(rs["accepted"]), "steps": int(rs["steps"])},
                        "greedy": {"generated_tokens": [int(x) for x in rg["generated_tokens"]], "proposed": int(rg["proposed"]),
                                   "accepted": int(rg["accepted"]), "steps": int(rg["steps"])},
                    })
                    print(pname, "k", k, "T", temperature, "sampled", rs["generated_tokens"], rs["accepted"], "/", rs["proposed"], "steps", rs["steps"],
                          "| greedy", rg["accepted"], "/", rg["proposed"], "steps", rg["steps"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "pipeline_sampled_golden.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["kernels"]
    if "kernels" in which:
        gen_kernels()
    if "hf" in which:
        gen_hf()
    if "pipeline" in which:
        gen_pipeline()
    if "pipeline_eagle" in which:
        gen_pipeline_eagle()
    if "hostlogic" in which:
        gen_hostlogic()
    if "medusa" in which:
        gen_medusa()
    if "policies" in which:
        gen_pipeline_policies()
    if "harness" in which:
        gen_harness()
    if "fake" in which:
        gen_fake()
    if "pipeline_sampled" in which:
        gen_pipeline_sampled()
