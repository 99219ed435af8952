"""Positional rows of the CogVideoX patch embedding at a clip length other than the model's own (cogvideox.py: diffusers 0.32.2
`CogVideoXPatchEmbed.forward` regenerates a sin-cos table with zero text rows instead of slicing the learned one).  Parity unpinned (diffusers is
absent): the product's closed-form table is checked against the oracle's step-by-step restatement of the package's recipe, and against the recipe's
own structure."""
import numpy as np
import pytest
import torch

from motionrag_amd.cogvideox import CogVideoXTransformer3DModel, get_3d_sincos_pos_embed
from oracle import cogvideox_ref


@pytest.mark.parametrize("D,W,H,T,ss,ts", [(64, 6, 4, 3, 1.875, 1.0), (128, 5, 7, 2, 1.0, 2.0), (3072, 45, 30, 5, 1.875, 1.0)])
def test_sincos_table_equals_the_step_by_step_restatement(D, W, H, T, ss, ts):
    got = get_3d_sincos_pos_embed(D, (W, H), T, ss, ts)
    want = torch.from_numpy(cogvideox_ref.sincos_3d(D, (W, H), T, ss, ts)).flatten(0, 1).float()
    assert got.shape == (T * H * W, D) and torch.equal(got, want)
    # structure of a row (t, y, x): [ temporal D/4 | x-slot 3D/8 | y-slot 3D/8 ], each [sin | cos]; position 0 gives sin = 0, cos = 1
    r0 = got[0]
    for lo, n in ((0, D // 4), (D // 4, 3 * D // 8), (D // 4 + 3 * D // 8, 3 * D // 8)):
        assert torch.all(r0[lo:lo + n // 2] == 0) and torch.all(r0[lo + n // 2:lo + n] == 1)
    g = got.view(T, H, W, D)
    assert torch.equal(g[0, 0, :, :D // 4], g[0, 0, :1, :D // 4].expand(W, -1))                       # temporal part constant over a frame
    assert torch.equal(g[:, :, 2, D // 4:D // 4 + 3 * D // 8], g[:1, :1, 2, D // 4:D // 4 + 3 * D // 8].expand(T, H, -1))   # x slot depends on x only
    assert float(g[0, 0, 1, D // 4]) == pytest.approx(np.sin(1.0 / ss), abs=1e-6)                    # first x frequency is 1: sin(x / scale)
    assert float(g[0, 1, 0, D // 4 + 3 * D // 8]) == pytest.approx(np.sin(1.0 / ss), abs=1e-6)       # ... and the y slot follows it


def test_joint_table_choice_follows_patch_embed_forward():
    """model's own clip length -> the learned rows; another length -> zero text rows + sin-cos video rows (NOT a prefix of the learned table);
    another resolution -> refused"""
    m = CogVideoXTransformer3DModel(num_layers=1, num_attention_heads=2, in_channels=16, out_channels=8, time_embed_dim=64, text_embed_dim=64,
                                    max_text_seq_length=10, sample_frames=3, sample_height=8, sample_width=12)
    with torch.no_grad():
        m.patch_embed.pos_embedding.normal_()
    own = m.joint_pos_embedding(3, 4, 6)
    assert own.data_ptr() == m.patch_embed.pos_embedding.data_ptr()
    short = m.joint_pos_embedding(2, 4, 6)
    assert short.shape == (10 + 2 * 24, 128) and short.dtype == torch.bfloat16
    assert torch.all(short[:10] == 0)
    assert torch.equal(short[10:], get_3d_sincos_pos_embed(128, (6, 4), 2, 1.875, 1.0).to(torch.bfloat16))
    assert m.joint_pos_embedding(2, 4, 6) is short                                                  # cached per geometry
    long = m.joint_pos_embedding(5, 4, 6)                                                           # longer than the table: also regenerated
    assert long.shape == (10 + 5 * 24, 128)
    with pytest.raises(ValueError):
        m.joint_pos_embedding(3, 4, 5)
    cfg = cogvideox_ref.DiTConfig(num_layers=1, heads=2, in_channels=16, out_channels=8, time_embed_dim=64, text_embed_dim=64, max_text_len=10,
                                  frames=3, height=8, width=12)
    sd = {"patch_embed.pos_embedding": m.patch_embed.pos_embedding.float()}
    assert torch.equal(cogvideox_ref.patch_embed_positions(sd, cfg, 3, 4, 6), sd["patch_embed.pos_embedding"])
    o = cogvideox_ref.patch_embed_positions(sd, cfg, 2, 4, 6)
    assert torch.equal(o[0].to(torch.bfloat16), short)
    with pytest.raises(ValueError):
        cogvideox_ref.patch_embed_positions(sd, cfg, 3, 4, 5)
