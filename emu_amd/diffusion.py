"""``EmuVisualGeneration`` -- drop-in for the reference's image-generation pipeline (Emu2/emu/diffusion.py:30-383).

Same call signature and output type: ``pipe(inputs, height=1024, width=1024, num_inference_steps=50, guidance_scale=3.,
crop_info=[0, 0], original_size=[1024, 1024]) -> EmuVisualGenerationPipelineOutput(image, nsfw_content_detected)``.
Stages: prompt -> ``EmuModel.generate_image`` / ``encode_image`` (autoencoding mode) with the cached negative
prompt -> 50-step hipGraph-replayed UNet denoise with cond-first CFG and the Euler scheduler -> VAE decode -> PIL.
The CLIP safety checker is out of scope: ``nsfw_content_detected`` is always ``None``, which is
what the reference returns when ``safety_checker`` is ``None`` (diffusion.py:241-249).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .chat import image_transform
from .conf.emu_conf import CLIPVisionCfg, TextDecoderCfg
from .constants import DEFAULT_IMG_PLACEHOLDER, EVA_IMAGE_SIZE, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
from .unet import UNetCfg, UNetEngine
from .vae import VaeCfg, VaeDecoder

BF16 = torch.bfloat16


@dataclass
class EmuVisualGenerationPipelineOutput:
    image: object                                   # PIL.Image.Image
    nsfw_content_detected: Optional[bool]


class EmuVisualGeneration:
    def __init__(self, multimodal_encoder, unet: UNetEngine, vae: VaeDecoder, eva_size=EVA_IMAGE_SIZE,
                 eva_mean=OPENAI_DATASET_MEAN, eva_std=OPENAI_DATASET_STD, safety_checker=None, **kwargs):
        """``safety_checker``: optional callable ``(images float32 [N, H, W, 3] in [0, 1]) -> (images, [bool] * N)`` run on
        the decoded images, the hook for the reference's StableDiffusionSafetyChecker stage (diffusion.py:154-166,236-249;
        a CLIP classifier outside the hot path, not rebuilt here).  None = no filter: ``nsfw_content_detected`` is None and
        nothing is blacked out -- a checkpoint that carries ``safety_checker.*`` weights triggers a warning at load."""
        self.multimodal_encoder = multimodal_encoder
        self.unet = unet
        self.vae = vae
        self.safety_checker = safety_checker
        self.vae_scale_factor = 2 ** (len(vae.cfg.block_out_channels) - 1)
        self.transform = lambda img: image_transform(img, eva_size, eva_mean, eva_std)
        self.negative_prompt = {}                   # "" / "[NULL_IMAGE]" -> embeds, computed once (diffusion.py:197-210)
        self.use_graph = True
        # classifier-free guidance split over ranks 0 and 1 (``enable_cfg_split``): a single image's denoise loop then
        # runs half the UNet rows per rank and exchanges 131 KB per step, instead of a full replica on every rank
        self.cfg_pair = None

    def enable_cfg_split(self) -> None:
        """Collective over the default process group (torch.distributed must be initialised, world size >= 2)."""
        from .tp import CfgPair
        self.cfg_pair = CfgPair()

    def device(self, module=None):
        return self.multimodal_encoder.ctx.device

    def dtype(self, module=None):
        return BF16

    # ------------------------------------------------------------------ diffusion.py:168-212
    @torch.no_grad()
    def _prepare_and_encode_inputs(self, inputs, do_classifier_free_guidance: bool = False,
                                   placeholder: str = DEFAULT_IMG_PLACEHOLDER) -> torch.Tensor:
        enc = self.multimodal_encoder
        has_image = has_text = False
        text_prompt, images = "", []
        for x in inputs:
            if isinstance(x, str):
                has_text = True
                text_prompt += x
            else:
                has_image = True
                text_prompt += placeholder
                images.append(self.transform(x))
        image_prompt = torch.stack(images).to(self.device()) if images else None
        if has_image and not has_text:                               # autoencoding mode: exactly one image
            prompt = enc.encode_image(image=image_prompt)
            if do_classifier_free_guidance:
                key = "[NULL_IMAGE]"
                if key not in self.negative_prompt:
                    self.negative_prompt[key] = enc.encode_image(image=torch.zeros_like(image_prompt))
                prompt = torch.cat([prompt, self.negative_prompt[key]], dim=0)
        else:                                                        # image generation mode
            key = ""
            if do_classifier_free_guidance and key not in self.negative_prompt:
                # first call: the prompt and the (cached from now on) negative prompt ride one weight stream -- two rows of
                # one batch, each on its own positions, i.e. what the reference's two batch-size-1 calls compute
                both = enc.generate_image(text=[text_prompt, key], image=image_prompt, warn_ragged=False)
                prompt, self.negative_prompt[key] = both[:1], both[1:]
            else:
                prompt = enc.generate_image(text=[text_prompt], image=image_prompt)
            if do_classifier_free_guidance:
                prompt = torch.cat([prompt, self.negative_prompt[key]], dim=0)      # cond FIRST
        return prompt

    # ------------------------------------------------------------------ diffusion.py:77-166
    @torch.no_grad()
    def generate_latents(self, prompt_embeds: torch.Tensor, height: int = 1024, width: int = 1024,
                         num_inference_steps: int = 50, guidance_scale: float = 3.0, crop_info=(0, 0),
                         original_size=(1024, 1024), latents: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Steps 2-4 of the reference forward: timesteps, initial latents (``torch.randn`` on the global generator,
        scaled by init_noise_sigma, diffusion.py:126-127) and the denoising loop.  ``latents`` can be injected
        (un-scaled standard normal noise) for reproducible parity runs."""
        if guidance_scale > 1.0:
            if prompt_embeds.shape[0] != 2:
                raise ValueError("batch size 1 with classifier-free guidance expects prompt_embeds [2, n, C] (cond, uncond)")
        else:
            # no classifier-free guidance (diffusion.py:98,133,144: batch 1, the prediction is used as is).  The engine is
            # built around the CFG pair, so the single prompt rides in both rows: rows of a UNet batch never interact, the
            # two predictions are bit-identical and uncond + g * (cond - uncond) returns them unchanged.
            if prompt_embeds.shape[0] != 1:
                raise ValueError("guidance_scale <= 1 expects prompt_embeds [1, n, C]")
            prompt_embeds = torch.cat([prompt_embeds, prompt_embeds], dim=0)
        dev = self.device()
        sch = self.unet.set_timesteps(num_inference_steps)
        self.unet.set_context(prompt_embeds, height, width, original_size, crop_info)
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        if latents is None:
            latents = torch.randn((1, self.unet.cfg.in_channels, h, w), device=dev, dtype=BF16)
        pair = self.cfg_pair
        if pair is not None and pair.half is not None and guidance_scale > 1.0:
            latents = pair.broadcast(latents.to(dev, BF16).contiguous())         # both ranks denoise rank 0's noise
            latents = (latents * sch.init_noise_sigma).contiguous()
            return self.unet.denoise_cfg_split(latents, guidance_scale, pair.half, pair.all_gather)
        latents = (latents.to(dev, BF16) * sch.init_noise_sigma).contiguous()
        return self.unet.denoise(latents, guidance_scale, use_graph=self.use_graph)

    @torch.no_grad()
    def forward(self, inputs, height: int = 1024, width: int = 1024, num_inference_steps: int = 50,
                guidance_scale: float = 3.0, crop_info: List[int] = [0, 0], original_size: List[int] = [1024, 1024]):
        if not isinstance(inputs, list):
            inputs = [inputs]
        do_cfg = guidance_scale > 1.0
        prompt_embeds = self._prepare_and_encode_inputs(inputs, do_cfg).to(self.device(), BF16)
        latents = self.generate_latents(prompt_embeds, height, width, num_inference_steps, guidance_scale, crop_info,
                                        original_size)
        images = self.decode_latents(latents)
        nsfw = None
        if self.safety_checker is not None:                          # diffusion.py:154-166 (run_safety_checker)
            images, flags = self.safety_checker(images)
            nsfw = bool(flags[0]) if flags is not None else None
        return EmuVisualGenerationPipelineOutput(image=self.numpy_to_pil(images)[0], nsfw_content_detected=nsfw)

    __call__ = forward

    def decode_latents(self, latents: torch.Tensor) -> np.ndarray:
        image = self.vae.decode_latents(latents)
        return image.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def numpy_to_pil(images: np.ndarray):
        from PIL import Image
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        if images.shape[-1] == 1:
            return [Image.fromarray(im.squeeze(), mode="L") for im in images]
        return [Image.fromarray(im) for im in images]

    # ------------------------------------------------------------------ construction (diffusion.py:251-318)
    @classmethod
    def from_config(cls, llama_config_path: Optional[str] = None, device="cuda", unet_cfg: Optional[UNetCfg] = None,
                    vae_cfg: Optional[VaeCfg] = None, vision_cfg: Optional[CLIPVisionCfg] = None, **kwargs):
        """Emu2-Gen: EmuModel with default configs (n_query 64, no instruct tokens; diffusion.py:291), SDXL-shaped UNet,
        VAE.  The CLIP safety checker of the reference is not built."""
        from .emu import EmuModel
        tcfg = TextDecoderCfg() if llama_config_path is None else TextDecoderCfg(llama_config_path=llama_config_path)
        model_kw = {k: kwargs.pop(k) for k in ("llama_cfg", "tp_rank", "tp_size", "ctx") if k in kwargs}
        enc = EmuModel(vision_cfg=vision_cfg or CLIPVisionCfg(), text_decoder_cfg=tcfg, device=device, **model_kw)
        unet = UNetEngine(unet_cfg or UNetCfg(), enc.ctx)
        vae = VaeDecoder(vae_cfg or VaeCfg(), enc.ctx)
        return cls(multimodal_encoder=enc, unet=unet, vae=vae, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True):
        """Pipeline checkpoint keys: ``multimodal_encoder.*``, ``unet.*``, ``vae.*``.  ``safety_checker.*`` weights are
        not consumed (see ``__init__``): their presence without a ``safety_checker`` hook is reported once."""
        if self.safety_checker is None and any(k.startswith("safety_checker.") for k in state_dict):
            import warnings
            warnings.warn("EmuVisualGeneration: the checkpoint's safety_checker.* weights are discarded and NO NSFW filter "
                          "runs (nsfw_content_detected stays None, flagged images are not blacked out as in the "
                          "reference pipeline); pass safety_checker=<callable> to restore the stage", stacklevel=2)
        enc_items = ((k[len("multimodal_encoder."):], v) for k, v in state_dict.items() if k.startswith("multimodal_encoder."))
        self.multimodal_encoder.load_weights(enc_items, strict=strict)
        self.unet.load_state_dict(((k, v) for k, v in state_dict.items()), prefix="unet.", strict=strict)
        self.vae.load_state_dict(state_dict, prefix="vae.", strict=strict)

    @classmethod
    def from_pretrained(cls, model_path: str, config_path: Optional[str] = None, dtype: torch.dtype = torch.bfloat16,
                        use_safetensors: bool = True, **kwargs):
        from .checkpoint import find_index, iter_checkpoint
        ins = cls.from_config(**kwargs)
        sd = dict(iter_checkpoint(model_path, use_safetensors if find_index(model_path) is None else None))
        ins.load_state_dict(sd, strict=True)
        return ins

    def multito(self, device_list):
        if len(device_list) > 1 and self.multimodal_encoder.ctx.tp_size != len(device_list):
            raise NotImplementedError("multi-GPU emu_amd runs one process per GPU (tensor parallel)")
        return self

    multicuda = multito
