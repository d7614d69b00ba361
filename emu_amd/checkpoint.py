"""Checkpoint ingestion: one iterator over (reference key name, tensor) pairs for every on-disk form the model ships in.

* a single ``.pth`` / ``.bin`` / ``.pt`` (``torch.load``) or ``.safetensors`` file -- what ``EmuChatGeneration.from_pretrained``
  and ``EmuVisualGeneration.from_pretrained`` of the reference take (Emu2/emu/chat.py:197-213, diffusion.py:251-268);
* a Hugging Face sharded checkpoint: a directory holding ``*.index.json`` (``{"weight_map": {key: shard file}}``, e.g.
  ``Emu2/emu/conf/llama_config/pytorch_model.bin.index.json`` or the hub's ``model.safetensors.index.json``) or the path of
  the index file itself.  Shards are opened one at a time and their tensors handed on one by one, so a 37 B-parameter
  checkpoint (74 GB in bf16) streams through host memory a shard at a time while the engines pack every decoder layer as
  soon as its seven matrices have arrived (``LlamaEngine.load_tensor``).

``prefix`` is put in front of every key (the base LLaMA's index names ``model.layers...``; the Emu state dict calls the same
tensors ``decoder.lm.model.layers...``)."""
from __future__ import annotations

import glob
import json
import os
from typing import Iterator, Optional, Tuple

import torch


def _load_file(path: str, use_safetensors: Optional[bool]):
    st = path.endswith(".safetensors") if use_safetensors is None else use_safetensors
    if st:
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu")


def find_index(path: str) -> Optional[str]:
    """The ``*.index.json`` of a sharded checkpoint, if ``path`` is one (directory or the index file); else None."""
    if os.path.isdir(path):
        cands = sorted(glob.glob(os.path.join(path, "*.index.json")))
        if len(cands) > 1:                           # a hub snapshot may carry both forms: prefer safetensors
            st = [c for c in cands if "safetensors" in os.path.basename(c)]
            cands = st or cands
        return cands[0] if cands else None
    return path if path.endswith(".index.json") else None


def iter_checkpoint(path: str, use_safetensors: Optional[bool] = None, prefix: str = "") -> Iterator[Tuple[str, torch.Tensor]]:
    index = find_index(path)
    if index is None:
        if os.path.isdir(path):
            raise FileNotFoundError(f"{path}: no *.index.json in this directory (pass the checkpoint file itself)")
        for k, v in _load_file(path, use_safetensors).items():
            yield prefix + k, v
        return
    with open(index) as f:
        weight_map = json.load(f)["weight_map"]
    base = os.path.dirname(index)
    by_shard = {}
    for key, shard in weight_map.items():
        by_shard.setdefault(shard, []).append(key)
    for shard in sorted(by_shard):
        sd = _load_file(os.path.join(base, shard), None if use_safetensors is None else shard.endswith(".safetensors"))
        missing = [k for k in by_shard[shard] if k not in sd]
        if missing:
            raise KeyError(f"{shard}: the index lists {missing[:3]}{'...' if len(missing) > 3 else ''} but the shard does not hold them")
        for key in by_shard[shard]:
            yield prefix + key, sd[key]
        del sd
