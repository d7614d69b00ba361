"""Headroom reference only (never on the product path): the vendor library GEMM (torch -> hipBLASLt) on the shapes
of this repo's GEMMs, plain (no fused epilogue).  python tools/mm_library_ref.py"""
import torch, time
def t(M,N,K,it=30):
    a=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); b=torch.randn(N,K,device='cuda',dtype=torch.bfloat16)
    for _ in range(5): torch.nn.functional.linear(a,b)
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(it): torch.nn.functional.linear(a,b)
    e.record(); torch.cuda.synchronize(); ms=s.elapsed_time(e)/it
    print(f"torch linear M{M} N{N} K{K}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:8.1f} TF/s")
for s in [(770,19968,6656),(770,35840,6656),(770,6656,17920),(1025,15360,1792),(1025,1792,15360),(2048,1280,1280),(2048,10240,1280),(2048,1280,5120),(8192,5120,640),(4096,4096,4096),(8192,8192,8192)]:
    t(*s)
