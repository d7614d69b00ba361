"""``EmuChatGeneration`` -- drop-in for the reference's chat pipeline (Emu2/emu/chat.py:22-286).

Host-side shell only: image preprocessing (bicubic resize to 448, CLIP normalisation; chat.py:35-39), prompt
templating for plain and multi-turn chat inputs (chat.py:121-195, constants.py:34-43) and checkpoint loading
(chat.py:197-232), all feeding ``emu_amd.EmuModel.generate``.  ``multito`` keeps the reference's signature but maps a
device list onto tensor parallelism instead of layer placement (one process per GPU is required for more than one).
"""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np
import torch

from .conf.emu_conf import CLIPVisionCfg, TextDecoderCfg
from .constants import (ASSISTANT_TOKEN, DEFAULT_EOS_TOKEN, DEFAULT_IMG_PLACEHOLDER, DEFAULT_VID_PLACEHOLDER,
                        DEFAULT_VIDEO_TOKEN, EVA_IMAGE_SIZE, FAKE_VIDEO_END_TOKEN, GRD_SYMBOL, GROUND_SYSTEM_MESSAGE,
                        OPENAI_DATASET_MEAN, OPENAI_DATASET_STD, SYSTEM_MESSAGE, USER_TOKEN)


def image_transform(img, size: int = EVA_IMAGE_SIZE, mean=OPENAI_DATASET_MEAN, std=OPENAI_DATASET_STD) -> torch.Tensor:
    """torchvision ``Resize((size, size), BICUBIC) -> ToTensor -> Normalize(mean, std)`` on a PIL image
    (chat.py:35-39), without torchvision: PIL's own bicubic resize is what torchvision calls for PIL inputs."""
    from PIL import Image
    if not isinstance(img, Image.Image):
        raise TypeError(f"expected a PIL image, got {type(img)}")
    img = img.resize((size, size), Image.BICUBIC)
    a = np.array(img)                     # copy: torch.from_numpy needs a writable buffer
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).to(torch.float32) / 255.0       # ToTensor
    m = torch.tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None, None]
    return (t - m) / s


def prepare_inputs(inputs, transform=image_transform, image_placeholder: str = DEFAULT_IMG_PLACEHOLDER,
                   video_placeholder: str = DEFAULT_VID_PLACEHOLDER):
    """chat.py:121-157: strings are concatenated; images become placeholders; images between "[VIDEO]" and the fake end
    token "[/VIDEO]" (which is NOT emitted) are video frames."""
    is_video = False
    text, images, frames = "", [], []
    for x in inputs:
        if isinstance(x, str) and x == FAKE_VIDEO_END_TOKEN:
            is_video = False
        elif isinstance(x, str):
            if x == DEFAULT_VIDEO_TOKEN:
                is_video = True
            text += x
        elif is_video:
            text += video_placeholder
            frames.append(transform(x))
        else:
            text += image_placeholder
            images.append(transform(x))
    image = torch.stack(images) if images else None
    video = torch.stack(frames) if frames else None
    return [text], image, video


def prepare_chat_inputs(inputs, is_grounding: bool = False, transform=image_transform,
                        image_placeholder: str = DEFAULT_IMG_PLACEHOLDER, video_placeholder: str = DEFAULT_VID_PLACEHOLDER):
    """chat.py:159-195: system message, then alternating " [USER]: " / " [ASSISTANT]: " turns ("</s>[USER]: " after an
    assistant turn), closing with " [ASSISTANT]:" (+ "<grounding>")."""
    text = GROUND_SYSTEM_MESSAGE if is_grounding else SYSTEM_MESSAGE
    image = video = None
    prev = None
    for msg in inputs:
        if prev == ASSISTANT_TOKEN:
            text += f"{DEFAULT_EOS_TOKEN}{USER_TOKEN}: "
            prev = USER_TOKEN
        elif prev is None:
            text += f" {USER_TOKEN}: "
            prev = USER_TOKEN
        else:
            text += f" {ASSISTANT_TOKEN}: "
            prev = ASSISTANT_TOKEN
        t, im, vd = prepare_inputs(msg, transform, image_placeholder, video_placeholder)
        text += t[0]
        if im is not None:
            image = im if image is None else torch.cat([image, im])
        if vd is not None:
            video = vd if video is None else torch.cat([video, vd])
    text += f" {ASSISTANT_TOKEN}:"
    if is_grounding:
        text += GRD_SYMBOL
    return [text], image, video


class EmuChatGeneration:
    def __init__(self, emu_model, eva_size=EVA_IMAGE_SIZE, eva_mean=OPENAI_DATASET_MEAN, eva_std=OPENAI_DATASET_STD, **kwargs):
        self.emu_model = emu_model
        self.transform = lambda img: image_transform(img, eva_size, eva_mean, eva_std)

    @torch.no_grad()
    def forward(self, inputs, is_grounding: bool = False, num_beams: int = 5, max_new_tokens: int = 10, min_len: int = 1,
                do_sample: bool = False, penalty_alpha: Optional[float] = None, top_p: Optional[float] = None,
                top_k: Optional[int] = None, temperature: Optional[float] = None, length_penalty: float = -1,
                repetition_penalty: float = 1.0, synced_gpus: bool = False, skip_special_tokens: bool = True, **kwargs):
        """Chat generation takes List[List[str | Image]] (odd length: last message is the user's); plain generation takes
        List[str | Image]."""
        assert isinstance(inputs, list), "inputs must be a list"
        if isinstance(inputs[0], list):
            assert len(inputs) % 2 == 1, "last message must be user input"
            text, image, video = prepare_chat_inputs(inputs, is_grounding, self.transform)
        else:
            assert all(isinstance(i, str) or hasattr(i, "resize") for i in inputs), \
                "input can't be list of list for normal generation"
            text, image, video = prepare_inputs(inputs, self.transform)
        dev = self.emu_model.device()
        image = None if image is None else image.to(dev)
        video = None if video is None else video.to(dev)
        out = self.emu_model.generate(text=text, image=image, video=video, num_beams=num_beams,
                                      max_new_tokens=max_new_tokens, min_len=min_len, do_sample=do_sample,
                                      penalty_alpha=penalty_alpha, top_p=top_p, top_k=top_k, temperature=temperature,
                                      length_penalty=length_penalty, repetition_penalty=repetition_penalty,
                                      synced_gpus=synced_gpus, skip_special_tokens=skip_special_tokens, **kwargs)
        return out[0]

    __call__ = forward

    @classmethod
    def from_config(cls, instruct: bool = False, llama_config_path: Optional[str] = None, device="cuda", **kwargs):
        """chat.py:215-232: Emu2-Chat uses n_query=256, v_query=64 and the [USER]/[ASSISTANT] tokens."""
        from .emu import EmuModel
        vision_cfg = CLIPVisionCfg(n_query=256, v_query=64) if instruct else CLIPVisionCfg()
        tcfg = TextDecoderCfg(instruct=instruct) if llama_config_path is None else \
            TextDecoderCfg(llama_config_path=llama_config_path, instruct=instruct)
        model_kw = {k: kwargs.pop(k) for k in ("llama_cfg", "tp_rank", "tp_size", "ctx") if k in kwargs}
        return cls(emu_model=EmuModel(vision_cfg=vision_cfg, text_decoder_cfg=tcfg, device=device, **model_kw), **kwargs)

    @classmethod
    def from_pretrained(cls, path: str, instruct: bool = False, dtype: torch.dtype = torch.bfloat16,
                        use_safetensors: bool = False, **kwargs):
        """chat.py:197-213: single-file checkpoint (.pth via torch.load or safetensors), strict key match -- or a Hugging Face
        sharded checkpoint (directory / ``*.index.json``), streamed shard by shard (emu_amd/checkpoint.py).  Weights are
        always stored bf16 on the device (``dtype`` is accepted for signature compatibility)."""
        from .checkpoint import find_index, iter_checkpoint
        ins = cls.from_config(instruct=instruct, **kwargs)
        st = use_safetensors if find_index(path) is None else None
        ins.emu_model.load_weights(iter_checkpoint(path, st), strict=True)
        return ins

    def multito(self, device_list: List[Union[str, torch.device]]):
        """Reference: layer placement over ``device_list`` (chat.py:235-286).  Here multi-GPU means tensor parallelism with
        one process per GPU (construct the model with tp_rank/tp_size); a single device is a no-op."""
        if len(device_list) > 1 and self.emu_model.ctx.tp_size != len(device_list):
            raise NotImplementedError("multi-GPU emu_amd runs one process per GPU (tensor parallel): launch with "
                                      "torch.distributed.run and build the model with tp_rank/tp_size")
        return self

    multicuda = multito
