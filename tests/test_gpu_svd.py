"""GPU parity tests of the SVD UNet denoise step (BASELINE config 'SVD-UNet single denoise step'): the product path
(motionrag_amd.svd_unet on libmrag_hip.so) against the CPU restatement oracle/svd_ref.py (PARITY UNPINNED: diffusers is a
third-party package absent from the reference tree; see the oracle header) on the same seeded weights and inputs."""
import pytest
import torch

from test_gpu_kernels import close
from test_gpu_models import close as close_model
from test_oracle_golden import svd_tiny

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_add_bcast_axpby_kernels(hip):
    from motionrag_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 10, 64, generator=g).to(torch.bfloat16)
    t = torch.randn(3, 64, generator=g).to(torch.bfloat16)
    want = x.float() + t.float()[(torch.arange(60) // 10) % 3].view(6, 10, 64)                 # one vector per 10-row frame, period 3
    close(ops.add_bcast(x.to(DEV), t.to(DEV), 10), want, scale=1.0)
    want = x.float() + t.float()[torch.arange(60) % 3].view(6, 10, 64)
    close(ops.add_bcast(x.to(DEV), t.to(DEV), 1), want, scale=1.0)
    y = torch.randn(6, 10, 64, generator=g).to(torch.bfloat16)
    close(ops.axpby(x.to(DEV), y.to(DEV), 0.3, 0.7), 0.3 * x.float() + 0.7 * y.float(), scale=1.0)
    with pytest.raises(ValueError):
        ops.axpby(x.to(DEV), y.to(DEV)[:3], 0.5, 0.5)


def test_cfg_euler_step_against_oracle(hip):
    from motionrag_amd import ops
    from motionrag_amd.svd_unet import EulerDiscreteScheduler
    from oracle import svd_ref
    g = torch.Generator().manual_seed(4)
    sch = EulerDiscreteScheduler()
    sch.set_timesteps(25)
    sig = svd_ref.karras_sigmas(25)
    assert abs(sch.sigmas[3] - float(sig[3])) < 1e-9 and abs(sch.init_noise_sigma - (700.0 ** 2 + 1) ** 0.5) < 1e-6
    B, Fr = 1, 14
    x = (torch.randn(B, Fr, 4, 8, 16, generator=g) * 5).to(torch.bfloat16)
    v = torch.randn(2, B, Fr, 4, 8, 16, generator=g).to(torch.bfloat16)
    gs = torch.linspace(1.0, 3.0, Fr)
    for i in (0, 10, 24):
        want = svd_ref.euler_cfg_step(v[0].double(), v[1].double(), x.double(), float(sig[i]), float(sig[i + 1]), gs.double())
        got = sch.step_(v.to(DEV), x.clone().to(DEV), i, gs.to(DEV))
        close(got, want.float(), scale=want.abs().mean().item())


def test_svd_unet_against_oracle(hip):
    """reduced-width SVD UNet, adapters on every spatial attn2, CFG batch 2 x 4 frames: cross-attn down/up blocks, mid block,
    spatio-temporal res blocks, temporal attention, AlphaBlender, frame-index and added-time-id embeddings, TupleTensor"""
    from motionrag_amd.svd import TupleTensor
    from oracle import svd_ref
    unet, cfg, inp = svd_tiny()
    want = svd_ref.unet_forward(unet.state_dict(), cfg, inp["sample"], inp["timestep"], inp["image"], inp["added_time_ids"], inp["action"])
    unet = unet.to(DEV)
    ehs = TupleTensor([inp["image"].to(DEV, torch.bfloat16), inp["action"].to(DEV, torch.bfloat16)])
    got = unet(inp["sample"].to(DEV, torch.bfloat16), inp["timestep"], ehs, inp["added_time_ids"]).sample
    assert got.shape == want.shape
    close_model(got, want, rel_l2=3e-2)
    # without motion tokens (plain tensor context + adapter sites given none) the processor must refuse, as the reference asserts
    with pytest.raises(AssertionError):
        unet(inp["sample"].to(DEV, torch.bfloat16), inp["timestep"], inp["image"].to(DEV, torch.bfloat16), inp["added_time_ids"])
