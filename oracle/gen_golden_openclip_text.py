"""TEST INFRASTRUCTURE -- tests/golden/openclip_text.npz from the REAL `transformers.CLIPTextModel` (random init, reduced config, head_dim 64), its weights stored under
open_clip's parameter names (what the reference's FrozenOpenCLIPEmbedder holds).

    python -m oracle.gen_golden_openclip_text"""
import os

import numpy as np
import torch


def hf_to_openclip(sd: dict, layers: int) -> dict:
    pre = "text_model." if any(k.startswith("text_model.") for k in sd) else ""          # transformers 4.x nests the tower under `text_model`, 5.x does not
    out = {"token_embedding.weight": sd[pre + "embeddings.token_embedding.weight"], "positional_embedding": sd[pre + "embeddings.position_embedding.weight"],
           "ln_final.weight": sd[pre + "final_layer_norm.weight"], "ln_final.bias": sd[pre + "final_layer_norm.bias"]}
    for i in range(layers):
        h, o = f"{pre}encoder.layers.{i}.", f"transformer.resblocks.{i}."
        out[o + "attn.in_proj_weight"] = torch.cat([sd[h + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        out[o + "attn.in_proj_bias"] = torch.cat([sd[h + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        out[o + "attn.out_proj.weight"], out[o + "attn.out_proj.bias"] = sd[h + "self_attn.out_proj.weight"], sd[h + "self_attn.out_proj.bias"]
        out[o + "ln_1.weight"], out[o + "ln_1.bias"] = sd[h + "layer_norm1.weight"], sd[h + "layer_norm1.bias"]
        out[o + "ln_2.weight"], out[o + "ln_2.bias"] = sd[h + "layer_norm2.weight"], sd[h + "layer_norm2.bias"]
        out[o + "mlp.c_fc.weight"], out[o + "mlp.c_fc.bias"] = sd[h + "mlp.fc1.weight"], sd[h + "mlp.fc1.bias"]
        out[o + "mlp.c_proj.weight"], out[o + "mlp.c_proj.bias"] = sd[h + "mlp.fc2.weight"], sd[h + "mlp.fc2.bias"]
    return out


def main():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel
    torch.manual_seed(777)
    layers = 3
    cfg = CLIPTextConfig(vocab_size=300, hidden_size=128, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=2, max_position_embeddings=77,
                         hidden_act="gelu", layer_norm_eps=1e-5, attn_implementation="eager", bos_token_id=298, eos_token_id=299, pad_token_id=0)
    m = CLIPTextModel(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            p.copy_(p.to(torch.bfloat16).float())
    tokens = torch.randint(1, 299, (3, 77))
    tokens[:, 0] = 298; tokens[0, 20:] = 0; tokens[0, 19] = 299; tokens[1, 50:] = 0; tokens[1, 49] = 299      # sot ... eot, zero padded like open_clip.tokenize
    with torch.no_grad():
        o = m(input_ids=tokens, output_hidden_states=True)
        fl = m.text_model.final_layer_norm if hasattr(m, "text_model") else m.final_layer_norm
        last, penult = fl(o.hidden_states[-1]), fl(o.hidden_states[-2])
    assert torch.allclose(last, o.last_hidden_state)
    out = {"transformers_version": np.array(transformers.__version__), "tokens": tokens.numpy(), "last": last.numpy(), "penultimate": penult.numpy(),
           "cfg": np.array([128, 2, layers, 300], dtype=np.int64)}
    for k, v in hf_to_openclip(m.state_dict(), layers).items():
        out["sd." + k] = v.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "openclip_text.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
