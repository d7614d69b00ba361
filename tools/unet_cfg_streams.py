#!/usr/bin/env python
"""One image's denoise step with the two rows of the classifier-free-guidance pair as TWO CONCURRENT CHAINS on two streams of one GPU
(each chain = emu_unet_forward in cfg_half mode with its own workspace, joined by the guidance + Euler launch), against the batched
step (M = 2 x H x W rows per GEMM, one chain).  tools/unet_2stream.py showed two independent images overlap 1.23x; this asks the same
of the two halves of ONE image, whose GEMMs have half the rows each.  Both forms replayed from a hipGraph.
Usage: python tools/unet_cfg_streams.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import synth, ops
from emu_amd._lib import lib, check
from emu_amd.llama import EmuHipContext
from emu_amd.unet import UNetCfg, UNetEngine, unet_param_shapes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
ctx = EmuHipContext(dev)
cfg = UNetCfg()
e = UNetEngine(cfg, ctx)
e.load_state_dict(synth.iter_synth(unet_param_shapes(cfg), seed=0, device=dev, dtype=BF16))
g = torch.Generator().manual_seed(3)
prompt = torch.randn(2, 64, 1792, generator=g).to(BF16).to(dev)
sch = e.set_timesteps(50)
e.set_context(prompt, 1024, 1024)
H = W = 128
lat0 = (torch.randn(1, 4, H, W, generator=g) * sch.init_noise_sigma).to(BF16).to(dev)
L = lib()

# ---- batched reference
lat = lat0.clone()
with torch.no_grad():
    e.denoise(lat, 3.0, use_graph=True, steps=3)
    e.set_timesteps(50); lat.copy_(lat0)
    torch.cuda.synchronize(); t = time.perf_counter()
    e.denoise(lat, 3.0, use_graph=True, steps=steps)
    torch.cuda.synchronize(); t_b = (time.perf_counter() - t) / steps * 1e3
ref = lat.clone()

# ---- two chains
need = L.emu_unet_workspace_bytes(e.handle, H, W)
ws = [torch.empty(need, device=dev, dtype=torch.uint8) for _ in range(2)]
eps = torch.empty(2 * H * W, 4, device=dev, dtype=BF16)
side = [torch.cuda.Stream(device=dev) for _ in range(2)]
lat2 = lat0.clone()


DELAY = 0                                                 # spin cycles in front of chain 1 (phase shift between the chains)


def split_step():
    main = torch.cuda.current_stream()
    for h in range(2):
        side[h].wait_stream(main)
        check(L.emu_unet_set_cfg_half(e.handle, h), "emu_unet_set_cfg_half")
        with torch.cuda.stream(side[h]):
            if h == 1 and DELAY:
                torch.cuda._sleep(DELAY)
            check(L.emu_unet_forward(e.handle, lat2.data_ptr(), H, W, e.temb_table.data_ptr(), e.sigmas.data_ptr(), e.step_dev.data_ptr(),
                                     eps[h * H * W:].data_ptr(), ws[h].data_ptr(), ws[h].numel(), side[h].cuda_stream), "emu_unet_forward")
    check(L.emu_unet_set_cfg_half(e.handle, -1), "emu_unet_set_cfg_half")
    for h in range(2):
        main.wait_stream(side[h])
    check(L.emu_unet_cfg_euler_step(e.handle, eps.data_ptr(), lat2.data_ptr(), H, W, e.sigmas.data_ptr(), e.step_dev.data_ptr(), 3.0,
                                    main.cuda_stream), "emu_unet_cfg_euler_step")


for DELAY in [int(x) for x in os.environ.get('EMU_CFG_DELAYS', '0').split(',')]:
  with torch.no_grad():
      e.set_timesteps(50)
      cap = torch.cuda.Stream(device=dev)
      with torch.cuda.stream(cap):
          split_step()                                      # eager warm-up
          torch.cuda.synchronize()
          t = time.perf_counter()
          for _ in range(4):
              split_step()
          torch.cuda.synchronize()
          t_e = (time.perf_counter() - t) / 4 * 1e3
      gr = torch.cuda.CUDAGraph()
      with torch.cuda.graph(gr, stream=cap):
          split_step()
      e.set_timesteps(50); lat2.copy_(lat0)
      torch.cuda.synchronize(); t = time.perf_counter()
      for _ in range(steps):
          gr.replay()
      torch.cuda.synchronize(); t_s = (time.perf_counter() - t) / steps * 1e3
  err = float((lat2.float() - ref.float()).norm() / ref.float().norm())
  print(f"[delay {DELAY} cycles] denoise step at 128 x 128, {steps} steps: batched CFG pair {t_b:.2f} ms | two half-batch chains on two streams {t_s:.2f} ms (eager {t_e:.2f}) | "
        f"speed-up {t_b / t_s:.3f} | final latents rel L2 between the two forms {err:.2e} | finite {bool(torch.isfinite(lat2.float()).all())}", flush=True)
