"""Worker of tests/test_gpu_cfg_split.py: two rank processes (torch.distributed.run, gloo rendezvous, both on cuda:0) split the
classifier-free-guidance pair of a UNet denoise loop (emu_amd/tp.py::CfgPair, UNetEngine.denoise_cfg_split): each computes one row,
the 2 x [H*W, 4] predictions cross through all_gather every step.  Checks: both ranks end with IDENTICAL latents, equal to the
batched (pair in one batch) loop of the same engine within bf16 noise.  Exit code 0 = pass."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BF16 = torch.bfloat16


def main():
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    from emu_amd import synth
    from emu_amd.llama import EmuHipContext
    from emu_amd.tp import CfgPair
    from emu_amd.unet import UNetCfg, UNetEngine, unet_param_shapes
    cfg = UNetCfg(block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2), num_heads=(1, 2, 4),
                  cross_attention_dim=128, projection_class_embeddings_input_dim=128 + 6 * 256)
    W = synth.synth_state_dict(unet_param_shapes(cfg), seed=5, dtype=torch.float32)
    W = {k: (v * (2.0 if v.dim() > 1 else 1.0)) for k, v in W.items()}
    eng = UNetEngine(cfg, EmuHipContext(torch.device("cuda", 0)))
    eng.load_state_dict(W)
    pair = CfgPair()
    g = torch.Generator().manual_seed(71)
    prompt = torch.randn(2, 8, 128, generator=g).to(BF16).cuda()
    H = 16
    steps = 8
    sch = eng.set_timesteps(steps)
    eng.set_context(prompt, 128, 128)
    # every rank draws its OWN noise (as the pipeline's torch.randn does): the pair denoises rank 0's
    noise = torch.randn(1, 4, H, H, generator=torch.Generator().manual_seed(100 + rank)).to(BF16).cuda()
    noise = pair.broadcast(noise)
    lat0 = (noise.float() * sch.init_noise_sigma).to(BF16).contiguous()
    ref = eng.denoise(lat0.clone(), 3.0, use_graph=False)
    eng.set_timesteps(steps)
    got = eng.denoise_cfg_split(lat0.clone(), 3.0, pair.half, pair.all_gather)
    torch.cuda.synchronize()
    both = pair.all_gather(got)
    same = bool(torch.equal(both[0], both[1]))
    err = float((got.float() - ref.float()).norm() / ref.float().norm())
    print(f"rank {rank}: half {pair.half}, identical latents on both ranks: {same}, rel-L2 vs the batched loop {err:.2e}", flush=True)
    ok = same and err < 2e-2 and bool(torch.isfinite(got.float()).all())
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
