"""Checkpoint ingestion on CPU: single files and Hugging Face sharded checkpoints stream the same (key, tensor) pairs;
the reference's own index file (Emu2/emu/conf/llama_config/pytorch_model.bin.index.json) names exactly the LLaMA-33B
tensors the engine's inventory expects."""
import json
import os

import pytest
import torch

from emu_amd.checkpoint import find_index, iter_checkpoint


def _sd():
    g = torch.Generator().manual_seed(0)
    return {f"model.layers.{i}.{n}": torch.randn(4, 3, generator=g) for i in range(3) for n in ("a.weight", "b.weight")} | \
           {"lm_head.weight": torch.randn(5, 3, generator=g)}


def test_single_file_pth_and_safetensors(tmp_path):
    sd = _sd()
    torch.save(sd, tmp_path / "m.pth")
    got = dict(iter_checkpoint(str(tmp_path / "m.pth"), prefix="decoder.lm."))
    assert set(got) == {"decoder.lm." + k for k in sd} and all(torch.equal(got["decoder.lm." + k], v) for k, v in sd.items())
    from safetensors.torch import save_file
    save_file(sd, str(tmp_path / "m.safetensors"))
    got = dict(iter_checkpoint(str(tmp_path / "m.safetensors")))
    assert all(torch.equal(got[k], v) for k, v in sd.items())
    assert find_index(str(tmp_path / "m.pth")) is None


@pytest.mark.parametrize("fmt", ["bin", "safetensors"])
def test_sharded_checkpoint_streams_every_tensor_once(tmp_path, fmt):
    sd = _sd()
    keys = sorted(sd)
    shards = {f"model-0000{j + 1}-of-00003.{fmt}": keys[j::3] for j in range(3)}
    for name, ks in shards.items():
        part = {k: sd[k] for k in ks}
        if fmt == "bin":
            torch.save(part, tmp_path / name)
        else:
            from safetensors.torch import save_file
            save_file(part, str(tmp_path / name))
    idx = tmp_path / ("pytorch_model.bin.index.json" if fmt == "bin" else "model.safetensors.index.json")
    idx.write_text(json.dumps({"metadata": {}, "weight_map": {k: n for n, ks in shards.items() for k in ks}}))
    for path in (str(tmp_path), str(idx)):
        items = list(iter_checkpoint(path))
        assert sorted(k for k, _ in items) == keys                     # every tensor exactly once
        assert all(torch.equal(v, sd[k]) for k, v in items)
    os.remove(tmp_path / next(iter(shards)))
    with pytest.raises(FileNotFoundError):
        list(iter_checkpoint(str(tmp_path)))


def test_index_missing_key_is_reported(tmp_path):
    torch.save({"a": torch.zeros(1)}, tmp_path / "s1.bin")
    (tmp_path / "x.index.json").write_text(json.dumps({"weight_map": {"a": "s1.bin", "b": "s1.bin"}}))
    with pytest.raises(KeyError):
        list(iter_checkpoint(str(tmp_path)))


def test_reference_llama_index_matches_engine_inventory():
    """The index file the reference ships lists the base LLaMA-33B tensors; with the ``decoder.lm.`` prefix they are the
    decoder part of the Emu state dict (plus rotary inv_freq buffers of old transformers, which the loader skips)."""
    p = "/root/reference/Emu2/emu/conf/llama_config/pytorch_model.bin.index.json"
    if not os.path.exists(p):
        pytest.skip("reference checkout not present on this box")
    from emu_amd import synth
    from emu_amd.conf.emu_conf import LlamaCfg
    wm = json.load(open(p))["weight_map"]
    ours = {k[len("decoder.lm."):] for k in synth.llama_param_shapes(LlamaCfg(), 32000)}
    theirs = {k for k in wm if not k.endswith("rotary_emb.inv_freq")}
    assert theirs == ours
