"""Tensor-level wrappers of the primitive C-ABI operators (include/emu_hip.h).

PyTorch is only the allocator/stream provider here: every function hands raw device pointers of bf16 CUDA
(ROCm) tensors to libemu_hip.so on torch's current stream.  No function has a torch fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from ._lib import check, lib

EPI_NONE, EPI_RESID, EPI_SWIGLU, EPI_SILU, EPI_GELU, EPI_GEGLU = range(6)
BF16 = torch.bfloat16


def stream(ref=None) -> int:
    """HIP stream handle for one library call: torch's current stream of the device that holds ``ref`` (a tensor or a
    torch.device; None = the current device).  The library launches on that stream without touching the HIP device
    state, and a launch on a stream of another GPU than the calling thread's current one is invalid -- so a ``ref`` on
    another device is refused here (no hidden ``set_device``: one process drives one GPU in this design, and
    ``EmuHipContext`` is the one place that makes its device current)."""
    if ref is None:
        return torch.cuda.current_stream().cuda_stream
    dev = ref.device if isinstance(ref, torch.Tensor) else torch.device(ref)
    if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
        raise ValueError(f"emu_amd: operand on {dev} but the current device is cuda:{torch.cuda.current_device()}; one process "
                         "drives one GPU -- create the EmuHipContext for that device (it makes it current) or call "
                         "torch.cuda.set_device first")
    return torch.cuda.current_stream(dev).cuda_stream


def _p(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _req(t: torch.Tensor, name: str, dtype=BF16):
    if not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor (emu_amd has no CPU path)")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name} must be contiguous in its last dimension")


def linear(x, w, bias=None, res=None, norm_w=None, eps: float = 0.0, epi: int = EPI_NONE, out=None):
    """out[m, n] = epi(sum_k x[m, k] w[n, k] + bias[n]); see emu_linear_bf16."""
    _req(x, "x"); _req(w, "w")
    assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1], (x.shape, w.shape)
    M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if epi in (EPI_SWIGLU, EPI_GEGLU) else N
    if out is None:
        out = torch.empty(M, n_out, device=x.device, dtype=BF16)
    _req(out, "out")
    if res is not None:
        _req(res, "res")
    check(lib().emu_linear_bf16(_p(x), _p(w), _p(bias), _p(res), _p(norm_w), _p(out), M, N, K,
                                x.stride(0), w.stride(0), res.stride(0) if res is not None else 0,
                                out.stride(0), float(eps), int(epi), stream(x)), "emu_linear_bf16")
    return out


def linear_fused(x, w, bias=None, res=None, epi: int = EPI_NONE, out=None, stats_out=None, ln=None, vt=None, cross=None):
    """emu_linear_fused_bf16: ``linear`` (M > 8) with the fused epilogues of the UNet transformer blocks.
    ``stats_out`` fp32 [N/128, M, 2]: per-row partial (sum, sum of squares) of the output per 128-column slot.
    ``ln`` = (c fp32 [N], d fp32 [N], stats fp32 [K/128, M, 2], eps): x is the un-normalised activation, w = W * gamma.
    ``vt`` = (vt_out bf16 [B, N - col0, S_pad], col0, S): columns >= col0 are stored key-contiguous instead of row-major.
    ``cross`` = (k_rows bf16 [B * n, ldk], vt bf16 [B, N / 64, 64, n_pad], n, rows_per_batch, scale): x @ w^T is attn2.to_q and the
    output is the cross-attention over the n <= 64 cached keys (see emu_linear_fx)."""
    import ctypes as C
    from ._lib import LinearFxC
    _req(x, "x"); _req(w, "w")
    M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if epi in (EPI_SWIGLU, EPI_GEGLU) else N
    if out is None:
        out = torch.empty(M, n_out, device=x.device, dtype=BF16)
    fx = LinearFxC()
    if stats_out is not None:
        assert stats_out.dtype == torch.float32 and stats_out.is_contiguous() and stats_out.numel() == (N // 128) * M * 2
        fx.row_stats_out = stats_out.data_ptr()
    if ln is not None:
        c, d, stats, eps = ln
        assert c.dtype == d.dtype == stats.dtype == torch.float32 and c.numel() == d.numel() == N and stats.is_contiguous()
        fx.ln_c, fx.ln_d, fx.ln_stats, fx.ln_slots, fx.ln_eps = c.data_ptr(), d.data_ptr(), stats.data_ptr(), stats.shape[0], float(eps)
    if vt is not None:
        vt_out, col0, S = vt
        _req(vt_out, "vt_out")
        fx.vt_out, fx.vt_col0, fx.vt_s, fx.vt_spad = vt_out.data_ptr(), int(col0), int(S), vt_out.shape[-1]
    if cross is not None:
        ck, cvt, n_ctx, rows, scale = cross
        _req(ck, "cross_k"); _req(cvt, "cross_vt")
        fx.cross_k, fx.cross_vt, fx.cross_ldk, fx.cross_n = ck.data_ptr(), cvt.data_ptr(), ck.stride(0), int(n_ctx)
        fx.cross_npad, fx.cross_rows, fx.cross_scale = cvt.shape[-1], int(rows), float(scale)
    check(lib().emu_linear_fused_bf16(_p(x), _p(w), _p(bias), _p(res), _p(out), M, N, K, x.stride(0), w.stride(0),
                                      res.stride(0) if res is not None else 0, out.stride(0), int(epi), C.byref(fx), stream(x)),
          "emu_linear_fused_bf16")
    return out


def quantize_fp8_rows(w):
    """Per-row symmetric OCP e4m3fn quantisation of a packed bf16 weight [N, K]: returns (bytes uint8 [N, K], scale fp32
    [N]) with w ~= fp8(bytes) * scale[:, None]; see emu_quantize_fp8_rows."""
    _req(w, "w")
    assert w.dim() == 2 and w.stride(1) == 1
    N, K = w.shape
    q = torch.empty(N, K, device=w.device, dtype=torch.uint8)
    sc = torch.empty(N, device=w.device, dtype=torch.float32)
    check(lib().emu_quantize_fp8_rows(_p(w), w.stride(0), _p(q), q.stride(0), _p(sc), N, K, stream(w)),
          "emu_quantize_fp8_rows")
    return q, sc


def linear_fp8(x8, xscale, w8, wscale, bias=None, res=None, epi: int = EPI_NONE, out=None):
    """fp8 x fp8 GEMM (both operands from ``quantize_fp8_rows``) -> bf16; see emu_linear_fp8_bf16."""
    _req(x8, "x8", torch.uint8); _req(w8, "w8", torch.uint8)
    _req(xscale, "xscale", torch.float32); _req(wscale, "wscale", torch.float32)
    assert x8.dim() == 2 and w8.dim() == 2 and x8.shape[1] == w8.shape[1], (x8.shape, w8.shape)
    M, K = x8.shape
    N = w8.shape[0]
    n_out = N // 2 if epi in (EPI_SWIGLU, EPI_GEGLU) else N
    if out is None:
        out = torch.empty(M, n_out, device=x8.device, dtype=BF16)
    _req(out, "out")
    if res is not None:
        _req(res, "res")
    check(lib().emu_linear_fp8_bf16(_p(x8), _p(xscale), _p(w8), _p(wscale), _p(bias), _p(res), _p(out), M, N, K, x8.stride(0),
                                    w8.stride(0), res.stride(0) if res is not None else 0, out.stride(0), int(epi), stream(x8)),
          "emu_linear_fp8_bf16")
    return out


def linear_fp8w(x, w8, wscale, bias=None, res=None, norm_w=None, eps: float = 0.0, epi: int = EPI_NONE, out=None):
    """``linear`` over an fp8 weight stream (decode rows only, M <= 2); see emu_linear_fp8w_bf16."""
    _req(x, "x"); _req(w8, "w8", torch.uint8); _req(wscale, "wscale", torch.float32)
    assert x.dim() == 2 and w8.dim() == 2 and x.shape[1] == w8.shape[1], (x.shape, w8.shape)
    M, K = x.shape
    N = w8.shape[0]
    n_out = N // 2 if epi == EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out, device=x.device, dtype=BF16)
    _req(out, "out")
    if res is not None:
        _req(res, "res")
    check(lib().emu_linear_fp8w_bf16(_p(x), _p(w8), _p(wscale), _p(bias), _p(res), _p(norm_w), _p(out), M, N, K,
                                     x.stride(0), w8.stride(0), res.stride(0) if res is not None else 0,
                                     out.stride(0), float(eps), int(epi), stream(x)), "emu_linear_fp8w_bf16")
    return out


def rmsnorm(x, w, eps: float, out=None):
    _req(x, "x"); _req(w, "w")
    rows, cols = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib().emu_rmsnorm_bf16(_p(x), _p(w), _p(out), rows, cols, x.stride(0), out.stride(0), float(eps), stream(x)),
          "emu_rmsnorm_bf16")
    return out


def layernorm(x, w, b, eps: float, res=None, out=None):
    _req(x, "x")
    assert x.is_contiguous()
    rows, cols = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib().emu_layernorm_bf16(_p(x), _p(w), _p(b), _p(res), _p(out), rows, cols, float(eps), stream(x)),
          "emu_layernorm_bf16")
    return out


def layernorm_q8(x, w, b, eps: float, res=None, want_y: bool = True):
    """``layernorm`` + per-row e4m3 quantisation of its bf16 result in one launch (see emu_layernorm_q8_bf16):
    returns (y or None, bytes uint8 [rows, cols], scale fp32 [rows])."""
    _req(x, "x")
    assert x.is_contiguous()
    rows, cols = x.shape
    y = torch.empty_like(x) if want_y else None
    q = torch.empty(rows, cols, device=x.device, dtype=torch.uint8)
    sc = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(lib().emu_layernorm_q8_bf16(_p(x), _p(w), _p(b), _p(res), _p(y), _p(q), _p(sc), rows, cols, float(eps), stream(x)),
          "emu_layernorm_q8_bf16")
    return y, q, sc


def embed_gather(ids, table, out=None):
    _req(ids, "ids", torch.int32); _req(table, "table")
    n = ids.numel()
    if out is None:
        out = torch.empty(n, table.shape[1], device=table.device, dtype=BF16)
    check(lib().emu_embed_gather_bf16(_p(ids), _p(table), _p(out), n, table.shape[1], table.shape[0], stream(table)),
          "emu_embed_gather_bf16")
    return out


def scatter_rows(src, dst_rows, out):
    _req(src, "src"); _req(dst_rows, "dst_rows", torch.int32); _req(out, "out")
    assert src.is_contiguous() and out.is_contiguous() and src.shape[0] == dst_rows.numel()
    check(lib().emu_scatter_rows_bf16(_p(src), _p(dst_rows), _p(out), src.shape[0], src.shape[1], stream(src)),
          "emu_scatter_rows_bf16")
    return out


def argmax(logits, vocab: Optional[int] = None, suppress_id: int = -1, out=None):
    _req(logits, "logits")
    rows = logits.shape[0]
    vocab = logits.shape[1] if vocab is None else vocab
    if out is None:
        out = torch.empty(rows, device=logits.device, dtype=torch.int32)
    check(lib().emu_argmax_bf16(_p(logits), logits.stride(0), rows, vocab, suppress_id, _p(out), stream(logits)),
          "emu_argmax_bf16")
    return out


def avgpool_tokens(x, grid: int, stride: int):
    """x [B, 1+grid*grid, C] -> [B, (grid/stride)^2, C] (cls dropped)."""
    _req(x, "x")
    assert x.is_contiguous()
    B, _, Cc = x.shape
    go = grid // stride
    out = torch.empty(B, go * go, Cc, device=x.device, dtype=BF16)
    check(lib().emu_avgpool_tokens_bf16(_p(x), _p(out), B, grid, Cc, stride, stream(x)), "emu_avgpool_tokens_bf16")
    return out


def rope_kv_append(qkv, cos, sin, pos, slot, kcache, vcache, B: int, T: int, H: int, D: int):
    _req(qkv, "qkv"); _req(kcache, "kcache"); _req(vcache, "vcache")
    _req(pos, "pos", torch.int32); _req(slot, "slot", torch.int32)
    assert qkv.is_contiguous() and kcache.is_contiguous() and vcache.is_contiguous()
    S_max = kcache.shape[-2]
    check(lib().emu_rope_kv_append_bf16(_p(qkv), _p(cos), _p(sin), _p(pos), _p(slot), _p(kcache), _p(vcache),
                                        B, T, H, D, S_max, stream(qkv)), "emu_rope_kv_append_bf16")


def transpose_v(v, B: int, H: int, S: int, D: int, sb: int, sh: int, ss: int, S_pad: Optional[int] = None):
    _req(v, "v")
    S_pad = (S + 63) // 64 * 64 if S_pad is None else S_pad
    vt = torch.empty(B, H, D, S_pad, device=v.device, dtype=BF16)
    check(lib().emu_transpose_v_bf16(_p(v), sb, sh, ss, _p(vt), B, H, S, D, S_pad, stream(v)), "emu_transpose_v_bf16")
    return vt


def flash_attn(q, k, v, causal: bool, scale: float, kstart=None):
    """q [B,Sq,H,D], k/v [B,Sk,H,D] (any strides with D contiguous) -> o [B,Sq,H,D]."""
    _req(q, "q"); _req(k, "k"); _req(v, "v")
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    vt = transpose_v(v, B, H, Sk, D, v.stride(0), v.stride(2), v.stride(1))
    o = torch.empty(B, Sq, H, D, device=q.device, dtype=BF16)
    check(lib().emu_flash_attn_bf16(_p(q), q.stride(0), q.stride(2), q.stride(1),
                                    _p(k), k.stride(0), k.stride(2), k.stride(1), _p(vt),
                                    _p(o), o.stride(0), o.stride(2), o.stride(1), _p(kstart),
                                    B, H, Sq, Sk, vt.shape[-1], D, int(causal), float(scale), stream(q)),
          "emu_flash_attn_bf16")
    return o


def decode_attn(q, kcache, vcache, ctx: int, scale: float, kstart=None, ctx_dev=None, ctx_max: int = 0):
    """q [B,H,D]; caches [B,H,S_max,D] -> o [B,H,D]."""
    _req(q, "q"); _req(kcache, "kcache"); _req(vcache, "vcache")
    B, H, D = q.shape
    S_max = kcache.shape[-2]
    ws = torch.empty(lib().emu_decode_attn_ws_bytes(B, H, D, max(ctx, ctx_max)) // 4, device=q.device,
                     dtype=torch.float32)
    o = torch.empty(B, H, D, device=q.device, dtype=BF16)
    check(lib().emu_decode_attn_bf16(_p(q), q.stride(0), q.stride(1), _p(kcache), _p(vcache), _p(o), o.stride(0),
                                     o.stride(1), _p(kstart), _p(ctx_dev), ctx, ctx_max, _p(ws), B, H, D, S_max,
                                     float(scale), stream(q)), "emu_decode_attn_bf16")
    return o


CONV_3X3, CONV_3X3_S2, CONV_3X3_UP2 = 1, 2, 3


def groupnorm_nhwc(x, gamma, beta, groups: int, eps: float, silu: bool = False):
    """x [B, HW, C] (NHWC) -> GroupNorm(groups)(+SiLU), same shape."""
    _req(x, "x")
    assert x.is_contiguous() and x.dim() == 3
    B, HW, Cc = x.shape
    ws = torch.empty(lib().emu_groupnorm_ws_bytes(B, HW, Cc), device=x.device, dtype=torch.uint8)
    y = torch.empty_like(x)
    check(lib().emu_groupnorm_nhwc_bf16(_p(x), _p(gamma), _p(beta), _p(y), _p(ws), B, HW, Cc, groups, float(eps), int(silu),
                                        stream(x)), "emu_groupnorm_nhwc_bf16")
    return y


def conv3x3_nhwc(x, w, bias=None, bias2=None, res=None, mode: int = CONV_3X3):
    """x [B, H, W, Cin] NHWC, w [Cout, 3, 3, Cin] -> [B, Ho, Wo, Cout]."""
    _req(x, "x"); _req(w, "w")
    assert x.is_contiguous() and w.is_contiguous()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H, W) if mode == CONV_3X3 else (((H + 1) // 2, (W + 1) // 2) if mode == CONV_3X3_S2 else (2 * H, 2 * W))
    y = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=BF16)
    check(lib().emu_conv3x3_nhwc_bf16(_p(x), _p(w), _p(bias), _p(bias2), bias2.stride(0) if bias2 is not None else 0, _p(res),
                                      _p(y), B, H, W, Cin, Cout, mode, stream(x)), "emu_conv3x3_nhwc_bf16")
    return y


def softmax_rows_(x, scale: float = 1.0, bias=None):
    """In-place x = bf16(softmax(x * scale [+ bias], dim=-1)) for a 2-D bf16 tensor (bias: bf16 [rows, cols])."""
    _req(x, "x")
    assert x.dim() == 2
    if bias is not None:
        _req(bias, "bias")
        assert bias.shape == x.shape
    check(lib().emu_softmax_rows_bf16(_p(x), _p(bias), x.shape[0], x.shape[1], x.stride(0),
                                      bias.stride(0) if bias is not None else 0, float(scale), stream(x)), "emu_softmax_rows_bf16")
    return x


def prefetch(t: torch.Tensor, workgroups: int = 256, nbytes: Optional[int] = None):
    """Touch the first ``nbytes`` (default: all) of a contiguous tensor into the infinity cache; see emu_prefetch."""
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("prefetch needs a contiguous GPU tensor")
    n = t.numel() * t.element_size() if nbytes is None else int(nbytes)
    check(lib().emu_prefetch(_p(t), n, int(workgroups), stream(t)), "emu_prefetch")


def gemv_chain(ctx_handle, specs, err: Optional[torch.Tensor] = None):
    """A chain of one-row projections in ONE persistent launch (emu_gemv_chain_bf16; csrc/decode_engine.hip).  ``specs``: list of
    dicts with keys w [N, K], x ([1, K] or None = the previous op's output), gain, eps, epi, res, out (None when the next op takes
    the output).  Returns (outputs list with None for handed-over ones, err counter tensor).  Raises EmuHipError(-95) for shapes the
    engine does not cover."""
    import ctypes as C
    from ._lib import ChainOpC
    n = len(specs)
    arr = (ChainOpC * n)()
    keep, outs = [], []
    dev = specs[0]["w"].device
    for i, sp in enumerate(specs):
        w = sp["w"]; _req(w, "w")
        N, K = w.shape
        epi = int(sp.get("epi", EPI_NONE))
        x = sp.get("x")
        nxt_takes = i + 1 < n and specs[i + 1].get("x") is None
        out = None
        if not nxt_takes:
            out = sp.get("out")
            if out is None:
                out = torch.empty(1, N // 2 if epi == EPI_SWIGLU else N, device=dev, dtype=BF16)
        outs.append(out)
        gain, res = sp.get("gain"), sp.get("res")
        arr[i] = ChainOpC(_p(w), N, K, _p(gain), float(sp.get("eps", 0.0)), epi, _p(res), _p(x), 0 if x is not None else 1, _p(out))
        keep += [w, x, gain, res, out]
    nb = lib().emu_gemv_chain_granule_bytes(arr, n)
    gran = torch.empty(max(nb, 16), device=dev, dtype=torch.uint8)
    if err is None:
        err = torch.zeros(1, device=dev, dtype=torch.int32)
    check(lib().emu_gemv_chain_bf16(ctx_handle, arr, n, gran.data_ptr(), gran.numel(), err.data_ptr(), stream(dev)), "emu_gemv_chain_bf16")
    return outs, err, (gran, keep)
