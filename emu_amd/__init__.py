"""emu_amd -- MI355X-native Emu2 inference path (hand-written gfx950 HIP kernels behind the reference's API).

    from emu_amd import EmuModel, CLIPVisionCfg, TextDecoderCfg

Importing the package does not load the HIP library; the first operator call does, and raises if it is missing
(there is deliberately no CPU fallback).
"""
from .conf.emu_conf import CLIPVisionCfg, LlamaCfg, TextDecoderCfg  # noqa: F401

__all__ = ["EmuModel", "CLIPVisionCfg", "TextDecoderCfg", "LlamaCfg"]


def __getattr__(name):
    if name == "EmuModel":
        from .emu import EmuModel
        return EmuModel
    raise AttributeError(name)
