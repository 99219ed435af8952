"""TEST INFRASTRUCTURE -- generates tests/golden/encoders.npz from the REAL third-party `transformers` classes (VideoMAEModel, Dinov2Model;
random init, reduced configs with head_dim 64) and from ATen's antialiased `interpolate` (what torchvision's Resize calls).

    python -m oracle.gen_golden_encoders [out_dir]

The fixture holds data only: inputs, the models' state dicts (as float32 arrays) and their `last_hidden_state` outputs.  The reference's wrappers
(src/projects/condition/encoders/condition.py:360-400, 561-604) are thin: frame sampling + transforms + `model(...).last_hidden_state`; the
generator applies exactly those steps around the real models (see `oracle/encoders_ref.py` for the torchvision semantics that are restated,
torchvision itself is not installed)."""
import os
import sys

import numpy as np
import torch


def main(out_dir):
    import transformers
    from transformers import Dinov2Config, Dinov2Model, VideoMAEConfig, VideoMAEModel
    from oracle import encoders_ref as E

    torch.manual_seed(1234)
    out = {"transformers_version": np.array(transformers.__version__)}

    def bf16_params_(m):                  # weights exactly representable in bf16: the HIP path (bf16 weights) and the fp32 classes then share them bit for bit
        with torch.no_grad():
            for p in list(m.parameters()) + list(m.buffers()):
                if p.is_floating_point():
                    p.copy_(p.to(torch.bfloat16).float())

    def put_sd(tag, sd):                  # stored as the bf16 bit patterns (uint16)
        for k, v in sd.items():
            out[f"{tag}.sd.{k}"] = v.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)

    # ---- VideoMAE, reduced: 128-d, 2 heads of 64, 2 layers, 4 frames of 32 x 32, tubelets 2 x 16 x 16 -> 8 tokens; then the same weights + a final LayerNorm
    kw = dict(hidden_size=128, num_attention_heads=2, num_hidden_layers=2, intermediate_size=256, image_size=32, num_frames=4,
              tubelet_size=2, patch_size=16, attn_implementation="eager")
    m = VideoMAEModel(VideoMAEConfig(use_mean_pooling=True, **kw)).eval()
    with torch.no_grad():
        for p in m.parameters():          # non-trivial biases / norm weights (the default init zeroes / ones them)
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    bf16_params_(m)
    m_ln = VideoMAEModel(VideoMAEConfig(use_mean_pooling=False, **kw)).eval()
    m_ln.load_state_dict(m.state_dict(), strict=False)
    with torch.no_grad():
        m_ln.layernorm.weight.add_(0.1 * torch.randn(128)); m_ln.layernorm.bias.add_(0.1 * torch.randn(128))
    bf16_params_(m_ln)
    video = torch.rand(2, 7, 3, 40, 52) * 2 - 1                       # b t c h w in [-1, 1]: 7 source frames -> 4 sampled, non-square
    pix = E.preprocess(video[:, E.uniform_frame_indices(7, 4)], 32, 32, "bilinear")
    with torch.no_grad():
        out["vmae.last_hidden_state"] = m(pixel_values=pix).last_hidden_state.numpy()
        out["vmae_ln.last_hidden_state"] = m_ln(pixel_values=pix).last_hidden_state.numpy()
    out["vmae.video"] = video.numpy()
    out["vmae.pixel_values"] = pix.numpy()
    out["vmae.cfg"] = np.array([128, 2, 2, 2, 16, 4], dtype=np.int64)          # hidden, heads, layers, tubelet, patch, frames
    out["vmae.eps"] = np.array(m.config.layer_norm_eps)
    put_sd("vmae", m_ln.state_dict())                                           # = m's state dict + layernorm.{weight,bias}

    # ---- DINOv2, reduced: 128-d, 2 heads, 2 layers, patch 14; position table of a 5 x 5 grid (image 70) evaluated at 56 x 56 (4 x 4: interpolated)
    cfg = Dinov2Config(hidden_size=128, num_attention_heads=2, num_hidden_layers=2, mlp_ratio=2, image_size=70, patch_size=14, attn_implementation="eager")
    m = Dinov2Model(cfg).eval()
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        m.embeddings.position_embeddings.add_(0.5 * torch.randn_like(m.embeddings.position_embeddings))
        m.embeddings.cls_token.add_(0.5 * torch.randn_like(m.embeddings.cls_token))
    bf16_params_(m)
    images = torch.rand(3, 3, 90, 75) * 2 - 1
    for tag, (rs, cr) in (("dino", (64, 56)), ("dino_native", (80, 70))):
        pix = E.preprocess(images, rs, cr, "bicubic")
        with torch.no_grad():
            y = m(pix).last_hidden_state
        out[f"{tag}.pixel_values"] = pix.numpy()
        out[f"{tag}.last_hidden_state"] = y.numpy()
        out[f"{tag}.resize_crop"] = np.array([rs, cr], dtype=np.int64)
    out["dino.images"] = images.numpy()
    out["dino.cfg"] = np.array([128, 2, 2, 14], dtype=np.int64)
    out["dino.eps"] = np.array(cfg.layer_norm_eps)
    put_sd("dino", m.state_dict())

    # ---- ATen antialiased resize on its own (down- and up-scaling, both filters): input -> output planes
    x = torch.rand(2, 3, 45, 61) * 2 - 1
    for mode, size in (("bilinear", (20, 27)), ("bicubic", (20, 27)), ("bilinear", (64, 90)), ("bicubic", (50, 61))):
        out[f"resize.{mode}.{size[0]}x{size[1]}"] = torch.nn.functional.interpolate(x, size=size, mode=mode, align_corners=False, antialias=True).numpy()
    out["resize.input"] = x.numpy()

    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "encoders.npz"), **out)
    print("wrote", os.path.join(out_dir, "encoders.npz"), sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
