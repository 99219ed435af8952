"""TEST INFRASTRUCTURE -- tests/golden/clip_vision.npz from the REAL `transformers.CLIPVisionModelWithProjection` (random init; reduced config with head_dim 80 and
hidden_act "gelu" like the CLIP-ViT-H image encoder SVD ships).

    python -m oracle.gen_golden_clip_vision"""
import os

import numpy as np
import torch


def main():
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(2024)
    cfg = CLIPVisionConfig(hidden_size=160, intermediate_size=320, num_hidden_layers=2, num_attention_heads=2, image_size=56, patch_size=14, projection_dim=64,
                           hidden_act="gelu", layer_norm_eps=1e-5, attn_implementation="eager")
    m = CLIPVisionModelWithProjection(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            p.copy_(p.to(torch.bfloat16).float())
    pix = torch.randn(3, 3, 56, 56)
    with torch.no_grad():
        o = m(pixel_values=pix)
    out = {"transformers_version": np.array(transformers.__version__), "pixel_values": pix.numpy(), "image_embeds": o.image_embeds.numpy(),
           "last_hidden_state": o.last_hidden_state.numpy(), "cfg": np.array([160, 2, 2, 320, 56, 14, 64], dtype=np.int64)}
    for k, v in m.state_dict().items():
        out["sd." + k] = v.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "clip_vision.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
