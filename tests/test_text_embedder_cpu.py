"""CPU side of the retrieval text embedder (gte-base-en-v1.5's NewModel, third-party remote code: parity UNPINNED): the product's module tree carries the published
parameter names / shapes the oracle lists (total = the published 136.8 M), and the row permutation that turns the checkpoint's `rotate_half` rotary pairing into the
kernel's interleaved pairing leaves the attention scores unchanged."""
import torch

from oracle import gte_ref as R


def test_module_tree_matches_published_names():
    from motionrag_amd.text_embedder import NewModel
    with torch.device("meta"):
        m = NewModel()
    have = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert have == R.state_shapes(R.CONFIG_BASE)
    assert sum(torch.Size(s).numel() for s in have.values()) == 136_776_192


def test_interleaved_rope_on_permuted_rows_equals_rotate_half():
    g = torch.Generator().manual_seed(0)
    S, H = 9, 2
    q, k = torch.randn(S, H, 64, generator=g), torch.randn(S, H, 64, generator=g)
    cos, sin = R.rope_tables(R.CONFIG_BASE, S)
    want = torch.einsum("shd,thd->hst", q * cos[:, None] + R.rotate_half(q) * sin[:, None], k * cos[:, None] + R.rotate_half(k) * sin[:, None])
    perm = torch.stack([torch.arange(32), torch.arange(32) + 32], dim=1).reshape(-1)
    inv = R.rope_inv_freq(R.CONFIG_BASE)
    ang = (torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]).repeat_interleave(2, dim=1)

    def interleaved(x):                                            # diffusers-style pairs (2i, 2i + 1): what mrag's RoPE epilogue applies
        xr = torch.stack([-x[..., 1::2], x[..., 0::2]], dim=-1).flatten(-2)
        return x * ang.cos()[:, None] + xr * ang.sin()[:, None]
    got = torch.einsum("shd,thd->hst", interleaved(q[..., perm]), interleaved(k[..., perm]))
    assert torch.allclose(got, want, atol=1e-4)
