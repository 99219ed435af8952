"""DynamiCrafter generation glue of the motion-injection path, mirroring

    image_guided_synthesis       src/projects/dynamicrafter/DynamiCrafter/scripts/evaluation/inference.py:174-305
    DynamiCrafterPipelineRef     src/projects/dynamicrafter/pipelines/pipeline.py:64-115

with the same names, arguments and return layout.  `model` is the reference's LatentVisualDiffusion-shaped object: this module reads
`embedder`, `image_proj_model`, `condition_transformer` (or `action_embedder` + `action_proj_model`), `get_learned_conditioning`,
`encode_first_stage`, `decode_first_stage`, `uncond_type`, `model.conditioning_key` and `model.diffusion_model` from it.  The
third-party pieces (OpenCLIP embedder, text encoder, first-stage VAE) stay whatever the caller supplies (SURVEY 8f "next" rows);
`model.model.diffusion_model` must be `motionrag_amd.dynamicrafter.UNetModel`, the sampler is `motionrag_amd.dynamicrafter.DDIMSampler`
(tables on the host, every update a gfx950 kernel).

One deliberate difference, forced by SURVEY Appendix D.3: the reference draws x_T and the per-step eta-noise with `torch.randn(..., device)`,
whose stream differs between devices; here they come from a CPU generator (`seed`) or from the caller (`x_T`, `noises`) and are
uploaded, so a fixed seed gives the same video on any device.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops

from .dynamicrafter import DDIMSampler, DynamiCrafterDenoiser


def get_latent_z(model, videos):
    """inference.py:165-170"""
    b, c, t, h, w = videos.shape
    x = videos.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    z = model.encode_first_stage(x)
    return z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)


@torch.no_grad()
def image_guided_synthesis(model, prompts, videos, noise_shape, n_samples=1, ddim_steps=50, ddim_eta=1.0, unconditional_guidance_scale=1.0,
                           cfg_img=None, fs=None, text_input=False, multiple_cond_cfg=False, loop=False, interp=False,
                           timestep_spacing="uniform", guidance_rescale=0.0, ref_videos=None, ref_fusion_type=None, metadata: dict = None,
                           x_T: Optional[torch.Tensor] = None, noises: Optional[List[torch.Tensor]] = None, seed: int = 0, **kwargs):
    if multiple_cond_cfg or loop or interp or guidance_rescale != 0.0 or timestep_spacing != "uniform":
        raise NotImplementedError("the shipped pipeline calls with multiple_cond_cfg=False, loop=False, interp=False, guidance_rescale=0, 'uniform' "
                                  "(pipelines/pipeline.py:96-115, configs/dynamicrafter/MotionRAG_open.yml:165-171)")
    if unconditional_guidance_scale == 1.0:
        raise NotImplementedError("classifier-free guidance is always on in the shipped configs (unconditional_guidance_scale 2.0)")
    device = videos.device
    batch_size = noise_shape[0]
    fs_t = torch.tensor([fs] * batch_size, dtype=torch.long, device=device)                       # :183
    if not text_input:
        prompts = [""] * batch_size
    bf = lambda t: t.to(device=device, dtype=torch.bfloat16).contiguous()

    cond = {"c_crossattn": {}}
    img = videos[:, :, 0]                                                                         # :189  b c h w
    img_emb = model.image_proj_model(model.embedder(img))                                         # :190-191
    cond["c_crossattn"] = {"image": bf(img_emb)}
    has_ae = getattr(model, "action_embedder", None) is not None
    has_ct = getattr(model, "condition_transformer", None) is not None
    if ref_videos is not None:
        if has_ae:                                                                                # :195-217
            b, k = ref_videos.shape[:2]
            action_emb = model.action_embedder(ref_videos.reshape(b * k, *ref_videos.shape[2:]))
            action_emb = action_emb.reshape(b, k, *action_emb.shape[1:])
            # fusion over the k references on the HIP kernel (mrag_weighted_sum_bf16: fp32 weights / accumulation), not torch arithmetic
            action_emb = action_emb.to(device=device, dtype=torch.bfloat16).contiguous()
            if ref_fusion_type == "mean":
                action_emb = ops.weighted_sum(action_emb, None, div=float(k))
            elif ref_fusion_type == "weight":                                                     # :203-206: ONE distance vector [k] for the batch
                distance = torch.as_tensor(metadata["ref_video_distance"], dtype=torch.float32, device=device)
                weight = (1 - distance) / (1 - distance).sum(dim=0, keepdim=True)                 # k numbers (host-sized arithmetic, as :205)
                action_emb = ops.weighted_sum(action_emb, weight.view(1, k).expand(b, k).contiguous())
            elif ref_fusion_type == "concat":
                action_emb = action_emb.reshape(b, -1, action_emb.shape[-1])
            elif ref_fusion_type is None or ref_fusion_type == "top1":
                action_emb = action_emb[:, 0]
            cond["c_crossattn"]["action"] = bf(model.action_proj_model(action_emb))
        elif has_ct:                                                                              # :219-223  (no CFG here: App. D.5)
            batch_ = {"ref_videos": ref_videos, "video": img[:, None].expand(-1, ref_videos.size(2), -1, -1, -1)}
            cond["c_crossattn"]["action"] = bf(model.condition_transformer.predict(batch_))
    cond_emb = model.get_learned_conditioning(prompts)                                            # :225
    cond["c_crossattn"]["prompt"] = bf(cond_emb)
    if model.model.conditioning_key != "hybrid":
        raise NotImplementedError("DynamiCrafter runs conditioning_key='hybrid'")
    z = get_latent_z(model, videos)                                                               # :229  b c t h w
    img_cat_cond = z[:, :, :1].expand(-1, -1, z.shape[2], -1, -1)                                  # :235-236
    cond["c_concat"] = [bf(img_cat_cond)]

    uc = {"c_crossattn": {}}                                                                       # :239-262
    if model.uncond_type == "empty_seq":
        uc_emb = model.get_learned_conditioning(batch_size * [""])
    elif model.uncond_type == "zero_embed":
        uc_emb = torch.zeros_like(cond_emb)
    else:
        raise ValueError(model.uncond_type)
    uc["c_crossattn"]["prompt"] = bf(uc_emb)
    uc["c_crossattn"]["image"] = bf(model.image_proj_model(model.embedder(torch.zeros_like(img))))
    if ref_videos is not None:
        if has_ae:
            uc["c_crossattn"]["action"] = bf(model.action_proj_model(model.action_embedder(torch.zeros_like(ref_videos[:, 0]))))
        elif has_ct:
            uc["c_crossattn"]["action"] = bf(model.condition_transformer.encode_vision(torch.zeros_like(ref_videos[:, 0:1]))[:, 0])
    uc["c_concat"] = [bf(img_cat_cond)]
    cond["fs"], uc["fs"] = fs_t, fs_t                                                              # the sampler hands `fs` to the UNet for both halves (ddim.py:231-233)

    sampler = DDIMSampler(getattr(model, "alphas_cumprod_np", None), use_dynamic_rescale=getattr(model, "use_dynamic_rescale", True))
    denoiser = DynamiCrafterDenoiser(model.model.diffusion_model)
    sampler.make_schedule(ddim_steps, ddim_eta)
    n_steps = len(sampler.ddim_timesteps)
    batch_variants = []
    g = torch.Generator().manual_seed(seed)
    for _ in range(n_samples):                                                                     # :275-302
        xt = (torch.randn(tuple(noise_shape), generator=g) if x_T is None else x_T).to(device=device, dtype=torch.float32)
        ns = noises if noises is not None else [torch.randn(tuple(noise_shape), generator=g) for _ in range(n_steps)]
        ns = [n.to(device=device, dtype=torch.float32).contiguous() for n in ns]
        samples = sampler.sample(denoiser, xt, cond, uc, S=ddim_steps, eta=ddim_eta, unconditional_guidance_scale=unconditional_guidance_scale, noises=ns)
        batch_variants.append(model.decode_first_stage(samples))
    return torch.stack(batch_variants).permute(1, 0, 2, 3, 4, 5)                                    # variants, b, c, t, h, w -> b, variants, ...


class DynamiCrafterPipelineRef:
    """pipelines/pipeline.py:64-115 (and its base :10-61): `eval_pipeline(image=..., positive_prompt=..., ref_videos=..., metadata=...)` -> [b, t, c, h, w] in [-1, 1]"""

    def __init__(self, model):
        self.model = model

    @torch.no_grad()
    def __call__(self, image: torch.Tensor, positive_prompt, negative_prompt=None, dtype: Optional[torch.dtype] = torch.float16, height: int = 512,
                 width: int = 512, num_frames: Optional[int] = 16, num_inference_steps: int = 50, eta: float = 1.0,
                 unconditional_guidance_scale: float = 7.5, cfg_img: Optional[float] = None, frame_stride: int = 20, multiple_cond_cfg: bool = False,
                 timestep_spacing: str = "uniform", guidance_rescale: float = 0.0, ref_videos: torch.Tensor = None, ref_fusion_type=None,
                 metadata: dict = None, *args, **kwargs):
        b = image.shape[0]
        image = image[:, :, None].expand(-1, -1, num_frames, -1, -1)                                # b c h w -> b c t h w  (:95)
        shape = [b, self.model.model.diffusion_model.out_channels, num_frames, height // 8, width // 8]
        videos = image_guided_synthesis(model=self.model, prompts=positive_prompt, videos=image, noise_shape=shape, n_samples=1,
                                        ddim_steps=int(num_inference_steps), ddim_eta=eta, unconditional_guidance_scale=float(unconditional_guidance_scale),
                                        cfg_img=cfg_img, fs=int(frame_stride), text_input=True, multiple_cond_cfg=multiple_cond_cfg, loop=False, interp=False,
                                        timestep_spacing=timestep_spacing, guidance_rescale=guidance_rescale, ref_videos=ref_videos,
                                        ref_fusion_type=ref_fusion_type, metadata=metadata, **kwargs)
        return videos[:, 0].permute(0, 2, 1, 3, 4)                                                 # 'b 1 c t h w -> b t c h w'  (:115)
