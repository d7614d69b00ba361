#!/usr/bin/env python
"""Run ONE kernel a few times (for rocprofv3 --pmc passes): python tools/one_op.py gemm M N K [epi] | gemv M N K | conv B H Cin Cout
EMU_ONE_OP_CFG=P7 pins a GEMM tile configuration / schedule variant (emu_gemm_force_config, as tools/gemm_ab.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emu_amd import ops  # noqa: E402

from emu_amd._lib import lib  # noqa: E402

kind = sys.argv[1]
cfg = os.environ.get("EMU_ONE_OP_CFG", "")
if cfg:
    lib().emu_gemm_force_config(ord(cfg[0]) | (int(cfg[1:] or 0) << 8))
a = [int(x) for x in sys.argv[2:]]
r = lambda *s, scale=1.0: (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)
if kind in ("gemm", "gemv"):
    M, N, K = a[:3]
    epi = a[3] if len(a) > 3 else 0
    x, w = r(M, K), r(N, K, scale=0.02)
    fn = lambda: ops.linear(x, w, epi=epi)
elif kind == "conv":
    B, H, Cin, Cout = a[:4]
    x, w = r(B, H, H, Cin), r(Cout, 3, 3, Cin, scale=0.02)
    fn = lambda: ops.conv3x3_nhwc(x, w)
for _ in range(5):
    fn()
torch.cuda.synchronize()
