"""SVD glue of the motion-injection path (boundary 8b-2).

The reference subclasses diffusers' `StableVideoDiffusionPipeline` (0.32.2, third-party, absent here) and adds in-tree
  * `TupleTensor` (src/projects/svd/pipelines/pipeline.py:25-57): the pair (image_embedding, action_emb) travelling where diffusers expects
    one tensor, so it survives `.to / .repeat_interleave / [idx] / .shape / .dtype` and reaches every spatial `attn2`, where
    `APAdapterAttnProcessor2_0` unpacks it (attn_processor.py:34-41);
  * `SVDActionPipeline` (stage 1: frozen action embedder + `condition_fusion`, pipeline.py:60-119) and `SVDCTPipeline` (stage 2: CAMA
    `condition_transformer.predict`, pipeline.py:122-160) which set `self.action_emb` and override `_encode_image`;
  * `set_attention_processors` (src/projects/svd/module.py:145-165) and `eval_pipeline` (:169-191).
Here the same classes sit on `motionrag_amd.svd_unet` (UNet + Euler scheduler on libmrag_hip.so); the body of the diffusers `__call__`
(image embedding, noise augmentation, VAE image latents, per-frame guidance, Euler loop, decode) is restated around duck-typed third-party
encoders: `image_encoder(pixels[b, 3, 224, 224]) -> .image_embeds / tensor [b, 1024]`, `vae.encode(x) -> .latent_dist.mode() / tensor`,
`vae.decode(z[(b f), 4, h, w], num_frames=) -> .sample / tensor`, `vae.config.scaling_factor` (0.18215)."""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

from . import ops
from .attn_processor import APAdapterAttnProcessor2_0


class TupleTensor(tuple):
    """A tuple of tensors that answers like its FIRST member (the image embedding) where diffusers inspects or indexes it, and like ALL its
    members where diffusers moves or repeats it (pipeline.py:25-57).  Built from one helper instead of one method per tensor op: anything in
    `_BROADCAST` maps over the members and re-wraps; every other attribute is the first member's."""

    _BROADCAST = ("to", "cuda", "cpu", "repeat_interleave", "contiguous", "clone", "detach")

    def _first(self) -> torch.Tensor:
        return tuple.__getitem__(self, 0)

    def __getattr__(self, name):
        if name in TupleTensor._BROADCAST:
            return lambda *a, **k: TupleTensor(getattr(t, name)(*a, **k) for t in tuple.__iter__(self))
        return getattr(self._first(), name)          # shape, dtype, device, size(), dim(), ndim, ...

    def __getitem__(self, item):
        return self._first()[item]

    def to_tuple(self):
        return tuple(tuple.__iter__(self))


class SVDPipelineOutput:
    """diffusers' StableVideoDiffusionPipelineOutput: `.frames`"""

    def __init__(self, frames):
        self.frames = frames


class SVDActionPipeline:
    """pipeline.py:60-119.  Constructor and call surface of the reference; see the module docstring for the third-party pieces."""

    def __init__(self, vae=None, image_encoder=None, unet=None, scheduler=None, feature_extractor=None, action_embedder=None,
                 action_proj_model=None, ref_fusion_type: str = "mean"):
        self.vae, self.image_encoder, self.unet, self.scheduler, self.feature_extractor = vae, image_encoder, unet, scheduler, feature_extractor
        self.action_embedder, self.action_proj_model, self.ref_fusion_type = action_embedder, action_proj_model, ref_fusion_type

    @property
    def _execution_device(self):
        return next(self.unet.parameters()).device

    def set_progress_bar_config(self, **_):
        return None

    # ---- motion tokens ----
    def prepare_action_embeddings(self, ref_videos: torch.Tensor, metadata, image=None) -> torch.Tensor:
        """pipeline.py:99-110: [b, k, f, c, h, w] -> [2b, t, c], uncond first"""
        from .cama import condition_fusion
        b, k = ref_videos.shape[:2]
        emb = self.action_embedder(ref_videos.reshape(b * k, *ref_videos.shape[2:]))
        emb = condition_fusion(emb.view(b, k, *emb.shape[1:]), self.ref_fusion_type,
                               weight=[m["ref_video_distance"] for m in metadata] if self.ref_fusion_type == "weight" else None)
        uncond = self.action_embedder(torch.zeros_like(ref_videos[:, 0]))
        return self.action_proj_model(torch.cat([uncond.to(emb.dtype), emb], dim=0))

    def _image_tensor(self, image) -> torch.Tensor:
        """list of PIL images / uint8 tensors [c, h, w] / one float tensor [b, c, h, w] in [0, 255] -> float [b, c, h, w] in [-1, 1]
        (`pil_to_tensor(img)` + `/ 127.5 - 1.0`, pipeline.py:154-155)"""
        if isinstance(image, torch.Tensor):
            t = image
        else:
            import numpy as np
            t = torch.stack([torch.from_numpy(np.asarray(im)).permute(2, 0, 1) if not isinstance(im, torch.Tensor) else im for im in image])
        return t.to(torch.float32) / 127.5 - 1.0

    # ---- diffusers StableVideoDiffusionPipeline pieces ----
    def _encode_image_base(self, image_pm1: torch.Tensor, device, do_classifier_free_guidance: bool) -> torch.Tensor:
        """diffusers `_encode_image`: CLIP image embedding [b, 1, 1024]; CFG: cat([zeros, emb]).  With the native `clip_vision` tower and no
        `feature_extractor`, the 224 x 224 antialiased resize + `(x + 1) / 2` + CLIP normalisation run in the tower's fused pixel kernel; a
        `feature_extractor(images)` callable (any other encoder) gets the [-1, 1] image as before."""
        if self.feature_extractor is None and hasattr(self.image_encoder, "encode_image"):
            # the native tower does diffusers' antialiased 224 x 224 resize + (x + 1) / 2 + CLIP normalisation itself (clip_vision.encode_image)
            emb = self.image_encoder.encode_image(image_pm1.to(device, torch.bfloat16))
        else:
            x = image_pm1
            if self.feature_extractor is not None:
                x = self.feature_extractor(x)
            emb = self.image_encoder(x.to(device))
        emb = getattr(emb, "image_embeds", emb)
        emb = emb.unsqueeze(1) if emb.dim() == 2 else emb
        if do_classifier_free_guidance:
            emb = torch.cat([torch.zeros_like(emb), emb], dim=0)
        return emb

    def _encode_image(self, *args, **kwargs) -> TupleTensor:
        """pipeline.py:113-119"""
        return TupleTensor([self._encode_image_base(*args, **kwargs), self.action_emb])

    def _vae_scale(self) -> float:
        return float(getattr(getattr(self.vae, "config", None), "scaling_factor", 0.18215))

    @torch.no_grad()
    def _denoise(self, latents, image_latents, ehs: TupleTensor, added_time_ids, num_inference_steps, guidance):
        """Euler loop: latents [b, F, 4, h, w] bf16 (already scaled by init_noise_sigma), image_latents [2b, F, 4, h, w] (uncond = zeros first)"""
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        b, F = latents.shape[:2]
        for i in range(num_inference_steps):
            scaled = ops.axpby(latents, latents, sch.input_scale(i), 0.0)                  # scheduler.scale_model_input
            x = torch.cat([torch.cat([scaled, scaled], dim=0), image_latents], dim=2)       # CFG duplicate + channel concat (memory plumbing)
            v = self.unet(x.contiguous(), float(sch.timesteps[i]), ehs, added_time_ids).sample
            sch.step_(v.view(2, b, F, *latents.shape[2:]), latents, i, guidance)
        return latents

    @torch.no_grad()
    def _run(self, image, height=576, width=1024, num_frames: Optional[int] = None, num_inference_steps: int = 25,
             min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0, fps: int = 7, motion_bucket_id: int = 127,
             noise_aug_strength: float = 0.02, decode_chunk_size: Optional[int] = None, generator=None, latents=None,
             output_type: str = "pt", return_dict: bool = True, **_unused):
        dev = self._execution_device
        num_frames = num_frames if num_frames is not None else getattr(getattr(self.unet, "config", None), "num_frames", 14)
        img = self._image_tensor(image)                                                      # [-1, 1]
        b = img.shape[0]
        ehs = self._encode_image(img, dev, True).to(dev, torch.bfloat16)
        # diffusers draws both noises with randn_tensor(shape, generator, device, dtype): IN the pipeline dtype (bf16: the reference's VideoProcessorDtype casts the
        # preprocessed image to vae.dtype, svd/pipelines/pipeline.py:16-22) on the generator's device (a CPU generator by default), and adds in that dtype
        gdev = generator.device if generator is not None else torch.device("cpu")
        img_bf = img.to(gdev, torch.bfloat16)
        noise = torch.randn(img.shape, generator=generator, dtype=torch.bfloat16, device=gdev)
        img_aug = img_bf + noise_aug_strength * noise
        z = self.vae.encode(img_aug.to(dev))
        z = z.latent_dist.mode() if hasattr(z, "latent_dist") else z                        # [b, 4, h, w] (NOT scaled: diffusers keeps the raw mode)
        image_latents = torch.cat([torch.zeros_like(z), z], dim=0)[:, None].expand(-1, num_frames, -1, -1, -1)
        added = torch.tensor([[float(fps - 1), float(motion_bucket_id), float(noise_aug_strength)]] * (2 * b), device=dev)
        self.scheduler.set_timesteps(num_inference_steps)
        if latents is None:
            latents = torch.randn(b, num_frames, 4, height // 8, width // 8, generator=generator, dtype=torch.bfloat16, device=gdev) * self.scheduler.init_noise_sigma
        latents = latents.to(dev, torch.bfloat16).contiguous()
        guidance = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames, device=dev)
        latents = self._denoise(latents, image_latents.to(dev, torch.bfloat16).contiguous(), ehs, added, num_inference_steps, guidance)
        if output_type == "latent":
            frames = latents
        else:
            zf = (latents.to(torch.float32) / self._vae_scale()).flatten(0, 1)               # [(b f), 4, h, w]
            chunk = decode_chunk_size or num_frames
            outs = []
            for i in range(0, zf.shape[0], chunk):
                o = self.vae.decode(zf[i:i + chunk], num_frames=min(chunk, zf.shape[0] - i))
                outs.append(o.sample if hasattr(o, "sample") and not callable(o.sample) else o)
            video = torch.cat(outs, dim=0).view(b, num_frames, *outs[0].shape[1:])           # [b, f, c, H, W] in [-1, 1]
            frames = (video / 2 + 0.5).clamp(0, 1)
            if output_type != "pt":
                raise NotImplementedError("output_type 'pt' or 'latent'")
        return SVDPipelineOutput(frames) if return_dict else frames

    def __call__(self, ref_videos: torch.Tensor = None, metadata=None, *args, **kwargs):
        """pipeline.py:93-111"""
        self.action_emb = self.prepare_action_embeddings(ref_videos, metadata)
        return self._run(*args, **kwargs)


class StableVideoDiffusionPipeline(SVDActionPipeline):
    """The pipeline WITHOUT motion injection: diffusers' `StableVideoDiffusionPipeline` as the reference's baseline module runs it (`SVDModule`,
    src/projects/svd/module.py, configs/svd/baseline_open.yml): the image embedding alone conditions the UNet (plain attention processors),
    `pipe(image=, height=, width=, num_frames=, decode_chunk_size=, ...)`."""

    def __init__(self, vae=None, image_encoder=None, unet=None, scheduler=None, feature_extractor=None):
        super().__init__(vae=vae, image_encoder=image_encoder, unet=unet, scheduler=scheduler, feature_extractor=feature_extractor)

    def _encode_image(self, *args, **kwargs) -> torch.Tensor:
        return self._encode_image_base(*args, **kwargs)

    def __call__(self, *args, ref_videos=None, metadata=None, **kwargs):
        return self._run(*args, **kwargs)


class SVDCTPipeline(SVDActionPipeline):
    """pipeline.py:122-160"""

    def __init__(self, vae=None, image_encoder=None, unet=None, scheduler=None, feature_extractor=None, condition_transformer=None):
        super().__init__(vae=vae, image_encoder=image_encoder, unet=unet, scheduler=scheduler, feature_extractor=feature_extractor)
        self.condition_transformer = condition_transformer

    def __call__(self, ref_videos: torch.Tensor = None, metadata=None, *args, **kwargs):
        image = self._image_tensor(kwargs.get("image")).to(ref_videos.device, ref_videos.dtype)            # [-1, 1]
        batch_ = {"ref_videos": ref_videos, "video": image[:, None].expand(-1, ref_videos.size(2), -1, -1, -1)}
        self.action_emb = self.condition_transformer.predict(batch_, do_classifier_free_guidance=True)
        return self._run(*args, **kwargs)


def set_attention_processors(unet, adapter_modules: Iterable[str], cross_attention_dim: int, hidden_sizes: Dict[str, int]) -> None:
    """src/projects/svd/module.py:145-165: install `APAdapterAttnProcessor2_0` on the listed `...attn2.processor` sites
    (configs/svd/MotionRAG_open.yml:115-131); `hidden_sizes[name]` is the block's channel count (320 / 640 / 1280)."""
    attn = {}
    for name, orig in unet.attn_processors.items():
        attn[name] = APAdapterAttnProcessor2_0(hidden_sizes[name], cross_attention_dim) if name in adapter_modules else orig
    unet.set_attn_processor(attn)


def eval_pipeline(pipe, image, positive_prompt=None, negative_prompt=None, dtype=None, ref_videos=None, metadata=None, *args, **kwargs):
    """SVDActionModule.eval_pipeline (svd/module.py:169-191): image in [-1, 1] -> `.frames[:, :16] * 2 - 1`.

    The reference hands the pipeline PIL images: `tensor2PIL` (module.py:181) goes through `denormalize` (src/utils/pipeline.py:178-184), i.e.
    `clip((x + 1) / 2, 0, 1) * 255` TRUNCATED to uint8, and the pipeline reads them back with `pil_to_tensor(img) / 127.5 - 1`
    (svd/pipelines/pipeline.py:154-155).  CLIP, the noise-augmented VAE input and CAMA's target frame therefore all see the quantised image; the
    same uint8 hop runs here on the GPU (`mrag_denormalize_u8`, bit-exact with the torch ops) and the pipeline receives the uint8 tensor."""
    from .eval_harness import denormalize
    img_u8 = denormalize(image.to(pipe._execution_device)).to(image.device)
    frames = pipe(image=img_u8, ref_videos=ref_videos, metadata=metadata, output_type="pt", *args, **kwargs).frames[:, :16]
    return frames * 2 - 1
