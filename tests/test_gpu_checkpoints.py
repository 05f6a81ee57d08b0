"""Loading real checkpoint files (safetensors + config.json), the path SmolLmSession::load / Rwkv7Session::load take
(src/models.rs:48-61, 132-146): the CPU oracle's seeded tensors are written out under the checkpoint's own tensor names — HF LLaMA
names for SmolLM2, the names candle_rwkv7's loader reads (rwkv7.rs:104-177, produced by convert_pth_direct.py) for RWKV7 — and the
session opened from those files must be bit-identical to the oracle."""
import json
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    m.lib()
    return m


def _save(path, tensors, dtype="bf16"):
    """tensors: name -> (uint16 bf16 bits, shape)"""
    import torch
    from safetensors.torch import save_file
    out = {}
    for k, (bits, shape) in tensors.items():
        t = torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16).reshape(shape)
        out[k] = t.to(torch.float32) if dtype == "f32" else t
    save_file(out, path)


def _eq(a, b, what):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32); b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    assert np.array_equal(a, b), what


def test_smollm_safetensors_and_config_json(cz, czo, tmp_path):
    cfg = czo.tiny_llama_config()
    o = czo.SmolLmSession(cfg, seed=31, embed_std=0.05)
    d, kvd, f, V = cfg.d_model, cfg.n_kv_heads * 64, cfg.d_ffn, cfg.vocab
    shape = {"embed_tokens.weight": (V, d), "norm.weight": (d,), "input_layernorm.weight": (d,), "post_attention_layernorm.weight": (d,),
             "q_proj.weight": (d, d), "k_proj.weight": (kvd, d), "v_proj.weight": (kvd, d), "o_proj.weight": (d, d),
             "gate_proj.weight": (f, d), "up_proj.weight": (f, d), "down_proj.weight": (d, f)}
    tensors = {}
    for n in o.tensor_names():
        key = next(k for k in shape if n.endswith(k))
        tensors[n] = (o.tensor_bf16(n), shape[key])
    names = sorted(tensors)
    # a sharded bf16 checkpoint (two files) and a single F32 file holding the same bf16-representable values
    _save(str(tmp_path / "model-00001-of-00002.safetensors"), {k: tensors[k] for k in names[: len(names) // 2]})
    _save(str(tmp_path / "model-00002-of-00002.safetensors"), {k: tensors[k] for k in names[len(names) // 2:]})
    _save(str(tmp_path / "f32.safetensors"), tensors, "f32")
    (tmp_path / "config.json").write_text(json.dumps({
        "architectures": ["LlamaForCausalLM"], "vocab_size": V, "hidden_size": d, "num_hidden_layers": cfg.n_layers, "num_attention_heads": cfg.n_heads,
        "num_key_value_heads": cfg.n_kv_heads, "intermediate_size": f, "max_position_embeddings": cfg.max_pos, "rope_theta": cfg.rope_theta,
        "rms_norm_eps": cfg.rms_eps, "tie_word_embeddings": True, "torch_dtype": "bfloat16"}))
    rng = np.random.default_rng(2)
    toks = rng.integers(0, V, 300).astype(np.uint32)
    ids = np.concatenate([[0], toks[:200]]).astype(np.uint32)
    ref_pay, ref_lo, ref_hi, _ = o.encode_tokens(ids)
    for paths in ([str(tmp_path / "model-00001-of-00002.safetensors"), str(tmp_path / "model-00002-of-00002.safetensors")], [str(tmp_path / "f32.safetensors")]):
        g = cz.Session("smollm", None, weight_paths=paths, config_path=str(tmp_path / "config.json"), max_streams=2)
        assert g.vocab_size() == V
        _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:70]), o.reprime_logits(toks[:70]), "reprime from checkpoint files")
        _eq(g.step_logits_tensor(toks[70]), o.step_logits(toks[70]), "step from checkpoint files")
        lo, hi = g.encode_tokens(ids)
        assert np.array_equal(lo, ref_lo) and np.array_equal(hi, ref_hi) and cz.ac_encode_pairs(lo, hi)[0] == ref_pay
        assert np.array_equal(g.decode_tokens(ref_pay, 200), ids)
        g.close()
    # error behaviour: a missing tensor, a missing file, a value that bf16 cannot hold
    broken = dict(tensors); del broken["model.norm.weight"]
    _save(str(tmp_path / "broken.safetensors"), broken)
    with pytest.raises(cz.CandlezipError) as e:
        cz.Session("smollm", None, weight_paths=[str(tmp_path / "broken.safetensors")], config_path=str(tmp_path / "config.json"))
    assert "model.norm.weight" in str(e.value)
    with pytest.raises(cz.CandlezipError):
        cz.Session("smollm", None, weight_paths=[str(tmp_path / "nope.safetensors")], config_path=str(tmp_path / "config.json"))
    import torch
    from safetensors.torch import load_file, save_file
    t = load_file(str(tmp_path / "f32.safetensors")); t["model.norm.weight"] = t["model.norm.weight"] + 1e-4
    save_file(t, str(tmp_path / "f32_full.safetensors"))
    with pytest.raises(cz.CandlezipError) as e:
        cz.Session("smollm", None, weight_paths=[str(tmp_path / "f32_full.safetensors")], config_path=str(tmp_path / "config.json"))
    assert "not bf16-representable" in str(e.value)           # refused loudly, never rounded silently
    # ... unless the caller opts in: round-to-nearest-even bf16 (here the perturbed norm weights round back to the original bits or one ulp)
    g = cz.Session("smollm", None, weight_paths=[str(tmp_path / "f32_full.safetensors")], config_path=str(tmp_path / "config.json"), max_streams=1,
                   round_f32_weights_to_bf16=True)
    lo2, hi2 = g.encode_tokens(ids)
    assert np.array_equal(g.decode_tokens(cz.ac_encode_pairs(lo2, hi2)[0], 200), ids)     # self-consistent whatever the rounding did
    g.close()
    o.close()


def test_rwkv7_safetensors_and_config_json(cz, czo, tmp_path):
    cfg = czo.tiny_rwkv7_config()
    o = czo.Rwkv7Session(cfg, seed=41, embed_std=0.05)
    C_, F, V, L = cfg.hidden, cfg.d_ffn, cfg.vocab, cfg.n_layers
    ranks = {"w_lora": cfg.decay_rank, "a_lora": cfg.a_rank, "v_lora": cfg.v_rank, "g_lora": cfg.gate_rank}
    tensors = {"model.embeddings.weight": (o.tensor_bf16("model.embeddings.weight"), (V, C_)), "lm_head.weight": (o.tensor_bf16("lm_head.weight"), (V, C_)),
               "model.norm.weight": (o.tensor_bf16("model.norm.weight"), (C_,)), "model.norm.bias": (o.tensor_bf16("model.norm.bias"), (C_,))}
    for i in range(L):
        p = f"model.layers.{i}."
        vecs = ["attn_norm.weight", "attn_norm.bias", "ffn_norm.weight", "ffn_norm.bias", "attn.g_norm.weight", "attn.g_norm.bias", "attn.x_r", "attn.x_w",
                "attn.x_k", "attn.x_v", "attn.x_a", "attn.x_g", "attn.k_k", "attn.k_a", "attn.r_k", "ffn.x_k",
                "attn.w_lora.lora.2.bias", "attn.a_lora.lora.2.bias"] + (["pre_norm.weight", "pre_norm.bias"] if i == 0 else ["attn.v_lora.lora.2.bias"])
        for v in vecs:
            tensors[p + v] = (o.tensor_bf16(p + v), (C_,))
        for m_ in ("r_proj", "k_proj", "v_proj", "o_proj"):
            tensors[p + f"attn.{m_}.weight"] = (o.tensor_bf16(p + f"attn.{m_}.weight"), (C_, C_))
        for lo_, r in ranks.items():
            if lo_ == "v_lora" and i == 0:
                continue                                          # layer 0 has no value-residual LoRA (rwkv7.rs:146-152)
            tensors[p + f"attn.{lo_}.lora.0.weight"] = (o.tensor_bf16(p + f"attn.{lo_}.lora.0.weight"), (r, C_))     # Linear [out, in]
            tensors[p + f"attn.{lo_}.lora.2.weight"] = (o.tensor_bf16(p + f"attn.{lo_}.lora.2.weight"), (C_, r))
        tensors[p + "ffn.key.weight"] = (o.tensor_bf16(p + "ffn.key.weight"), (F, C_))
        tensors[p + "ffn.value.weight"] = (o.tensor_bf16(p + "ffn.value.weight"), (C_, F))
    assert all(v[0] is not None for v in tensors.values())
    _save(str(tmp_path / "model.safetensors"), tensors)
    (tmp_path / "config.json").write_text(json.dumps({"vocab_size": V, "hidden_size": C_, "num_hidden_layers": L, "head_size": 64, "intermediate_size": F,
                                                      "a_low_rank_dim": cfg.a_rank, "decay_low_rank_dim": cfg.decay_rank, "v_low_rank_dim": cfg.v_rank,
                                                      "gate_low_rank_dim": cfg.gate_rank, "norm_eps": 1e-5}))
    g = cz.Session("rwkv7", None, weight_paths=[str(tmp_path / "model.safetensors")], config_path=str(tmp_path / "config.json"), max_streams=2)
    assert g.vocab_size() == V
    rng = np.random.default_rng(3)
    toks = rng.integers(0, V, 60).astype(np.uint32)
    o.reset()
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:1]), o.step_logits(toks[0]), "rwkv step 0 from checkpoint files")
    for i, t in enumerate(toks[1:12]):
        _eq(g.step_logits_tensor(t), o.step_logits(t), f"rwkv step {i + 1} from checkpoint files")
    ids = np.concatenate([[0], toks]).astype(np.uint32); ids[7] = V + 65
    ref_pay, ref_lo, ref_hi, _ = o.encode_tokens(ids)
    lo, hi = g.encode_tokens(ids)
    assert np.array_equal(lo, ref_lo) and np.array_equal(hi, ref_hi)
    assert np.array_equal(g.decode_tokens(ref_pay, ids.size - 1), ids)
    g.close(); o.close()
