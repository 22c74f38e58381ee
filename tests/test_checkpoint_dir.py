"""Local checkpoint directories (config.json + *.safetensors + tokenizer files): the path the reference reaches with
`from_pretrained(name_or_path)` (src/specdec/models/hf_wrappers.py:80-141). A tiny HF Llama / GPT-2 is saved with
`save_pretrained` (transformers' own writer) and read back by `load_checkpoint_dir` — every tensor must equal what
`from_hf_state_dict` makes of the live model, tied and untied heads; on the GPU a text prompt goes through the
directory's tokenizer and the pipeline decodes text."""

import pytest
import torch

from specdec_hip import weights as W


def _tiny_llama(tie):
    import transformers

    torch.manual_seed(5)
    cfg = transformers.LlamaConfig(vocab_size=96, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                   num_attention_heads=2, num_key_value_heads=1, head_dim=32, max_position_embeddings=128,
                                   rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=tie,
                                   bos_token_id=1, eos_token_id=2, pad_token_id=0)
    return transformers.LlamaForCausalLM(cfg).eval()


def _tiny_gpt2():
    import transformers

    torch.manual_seed(6)
    cfg = transformers.GPT2Config(vocab_size=80, n_positions=64, n_embd=64, n_layer=2, n_head=2, bos_token_id=1, eos_token_id=2)
    return transformers.GPT2LMHeadModel(cfg).eval()


def _save_tokenizer(path, vocab_size, eos):
    import transformers
    from tokenizers import Tokenizer, models, pre_tokenizers

    vocab = {f"t{i:03d}": i for i in range(vocab_size)}
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="t000"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    transformers.PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="t000", pad_token="t000", bos_token="t001",
                                         eos_token=f"t{eos:03d}").save_pretrained(path)


@pytest.mark.parametrize("kind", ["llama-untied", "llama-tied", "gpt2"])
def test_load_checkpoint_dir_round_trip(kind, tmp_path):
    model = _tiny_gpt2() if kind == "gpt2" else _tiny_llama(kind == "llama-tied")
    model.save_pretrained(tmp_path, safe_serialization=True)
    got = W.load_checkpoint_dir(str(tmp_path), device="cpu")
    cfg = W.config_from_hf(model.config)
    want = W.from_hf_state_dict(cfg, model.state_dict(), dtype=torch.bfloat16, device="cpu")
    assert got.config == cfg
    a, b = dict(got.tensors()), dict(want.tensors())
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
    assert (got.lm_head.data_ptr() == got.tok_emb.data_ptr()) == (kind != "llama-untied")   # tied heads alias the table


@pytest.mark.gpu
def test_pipeline_from_checkpoint_dirs_and_text_prompt(tmp_path):
    """`SpeculativePipeline(base_model=<dir>, draft_model=<dir>)` as the reference is called by path: weights from
    safetensors, the directory's tokenizer encodes the prompt and decodes the text; same logits as the weights handed
    over directly."""
    from oracle.model_ref import OracleLM
    from specdec_hip.engine import HipModel
    from src.specdec import SpeculativePipeline

    tdir, ddir = tmp_path / "target", tmp_path / "draft"
    target = _tiny_llama(False)
    target.save_pretrained(tdir, safe_serialization=True)
    _save_tokenizer(tdir, 96, 2)
    target.save_pretrained(ddir, safe_serialization=True)   # draft == target: everything is accepted
    _save_tokenizer(ddir, 96, 2)
    pipe = SpeculativePipeline(base_model=str(tdir), draft_model=str(ddir), controller="fixed", controller_params={"k": 2}, seed=1234)
    prompt = "t010 t020 t030 t040 t050"
    out = pipe.generate_batch([prompt], max_tokens=8, do_sample=False)[0]
    # (decode skips the special tokens: pad / unk t000, bos t001, eos t002)
    assert out["text"].split() == [f"t{i:03d}" for i in out["generated_tokens"] if i > 2]
    mw = W.from_hf_state_dict(W.config_from_hf(target.config), target.state_dict(), dtype=torch.bfloat16, device="cpu")
    toks = torch.tensor([[10, 20, 30, 40, 50]])
    want, _ = OracleLM(mw, "bf16").forward(toks)
    hm = HipModel(W.load_checkpoint_dir(str(tdir), device="cuda"), batch=1, l_max=64)
    _, logits = hm.forward(toks.to(torch.int32).cuda(), torch.zeros(1, dtype=torch.int32, device="cuda"), 0, want_logits=True)
    assert (logits.float().cpu() - want).abs().max().item() / want.abs().max().item() < 0.03
    # tokens and counters equal the oracle loop over the same weights (draft == target)
    from oracle.pipeline_ref import OraclePipeline

    lm = OracleLM(mw, "bf16")
    want_run = OraclePipeline(lm, lm, k=2, eos_token_id=2).generate_batch([[10, 20, 30, 40, 50]], 8)[0]
    assert out["generated_tokens"] == want_run["generated_tokens"]
    assert (out["proposed"], out["accepted"]) == (want_run["proposed"], want_run["accepted"])
