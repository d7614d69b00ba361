import os, sys, torch
sys.path.insert(0, "/root/repo")
from emu_amd import ops
from emu_amd._lib import lib
L = lib()
dev = torch.device("cuda", 0)
sk = torch.zeros(512 * 288 * 256, dtype=torch.float32, device=dev); L.emu_set_splitk_scratch(sk.data_ptr(), sk.numel() * 4)
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K, epi) in [(2048, 3840, 1280, 0), (1025, 15360, 1792, 4)]:
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16); w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    L.emu_gemm_force_config(ord("P"))
    fn = lambda: ops.linear(x, w, bias=bias, epi=epi, out=out)
    for _ in range(3): fn()
    buf.zero_(); L.emu_gemm_trace(buf.data_ptr()); fn(); torch.cuda.synchronize(); L.emu_gemm_trace(None)
    t = buf.view(-1, 8); t = t[t[:, 0] != 0].cpu().double() * 0.01
    med = lambda v: float(v.median())
    print(f"M{M} N{N} K{K} e{epi}: entry->loop end {med(t[:,2]-t[:,0]):.2f} | drain wait {med(t[:,1]-t[:,2]):.2f} | sync+col loads {med(t[:,5]-t[:,1]):.2f} | row-block loop {med(t[:,6]-t[:,5]):.2f} | sync+store+complete {med(t[:,3]-t[:,6]):.2f} us")
