"""ctypes binding of libemu_hip.so (the C ABI declared in include/emu_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or cannot be loaded this module raises,
and every operator raises ``EmuHipError`` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
# Production loads the in-tree library.  EMU_HIP_LIB redirects the load to another build for same-box A/B runs of tools/ ONLY:
# it is honoured when EMU_HIP_TOOLS=1 is set as well (a stale variable in a serving environment must not swap the library),
# announced on stderr, and the loaded library must report this binding's ABI version either way.
_OVERRIDE = os.environ.get("EMU_HIP_LIB") if os.environ.get("EMU_HIP_TOOLS") == "1" else None
LIB_PATH = _OVERRIDE or os.path.join(HERE, "csrc", "libemu_hip.so")
ABI_VERSION = 3            # emu_version() of the library these prototypes and struct layouts belong to
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "emu_hip.h")


class EmuHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None

vp, i32, f32, lng, sz = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_size_t


class LlamaCfgC(C.Structure):
    _fields_ = [("hidden", i32), ("heads_local", i32), ("head_dim", i32), ("ffn_local", i32), ("layers", i32),
                ("vocab", i32), ("max_pos", i32), ("rms_eps", f32)]


class VitCfgC(C.Structure):
    _fields_ = [("image_size", i32), ("patch_size", i32), ("width", i32), ("layers", i32), ("heads", i32),
                ("head_width", i32), ("mlp_hidden", i32), ("kpad", i32), ("ln_eps", f32), ("prenorm", i32)]


class LinearFxC(C.Structure):
    """emu_linear_fx (include/emu_hip.h): fused epilogues of the UNet transformer GEMMs."""
    _fields_ = [("row_stats_out", vp), ("ln_c", vp), ("ln_d", vp), ("ln_stats", vp), ("ln_slots", i32), ("ln_eps", f32),
                ("vt_out", vp), ("vt_col0", i32), ("vt_s", i32), ("vt_spad", i32),
                ("cross_k", vp), ("cross_vt", vp), ("cross_ldk", i32), ("cross_n", i32), ("cross_npad", i32), ("cross_rows", i32),
                ("cross_scale", f32)]


class ChainOpC(C.Structure):
    """emu_chain_op (include/emu_hip.h): one projection of an emu_gemv_chain_bf16 launch."""
    _fields_ = [("W", vp), ("N", i32), ("K", i32), ("gain", vp), ("eps", f32), ("epi", i32), ("res", vp), ("x", vp),
                ("x_from_prev", i32), ("out", vp)]


class ProfRowC(C.Structure):
    """emu_prof_row (include/emu_hip.h)."""
    _fields_ = [("klass", C.c_char * 16), ("M", i32), ("N", i32), ("K", i32), ("tag", i32), ("launches", i32),
                ("ms", C.c_double), ("flops", C.c_double)]


class UNetCfgC(C.Structure):
    _fields_ = [("in_ch", i32), ("out_ch", i32), ("ch", i32 * 3), ("layers_per_block", i32), ("depth", i32 * 3),
                ("heads", i32 * 3), ("attn", i32 * 3), ("cross_dim", i32), ("groups", i32), ("gn_eps", f32),
                ("temb_dim", i32), ("kpad_in", i32)]


_PROTOS = {
    "emu_version": (i32, []),
    "emu_gemm_trace": (None, [vp]),
    "emu_gemm_trace_nth": (None, [lng]),
    "emu_gemm_trace_built": (i32, []),
    "emu_profile_launches": (i32, [i32]),
    "emu_profile_launches_read": (i32, [vp, i32]),
    "emu_profile_gemv": (i32, [i32]),
    "emu_profile_gemv_read": (i32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "emu_ctx_create": (i32, [i32, i32, i32, C.POINTER(vp)]),
    "emu_ctx_destroy": (None, [vp]),
    "emu_last_error": (C.c_char_p, [vp]),
    "emu_tp_unique_id": (i32, [vp]),
    "emu_tp_init": (i32, [vp, vp]),
    "emu_allreduce_bf16": (i32, [vp, vp, sz, vp]),
    "emu_tp_p2p_create": (i32, [vp, vp]),
    "emu_tp_p2p_open": (i32, [vp, vp, i32]),
    "emu_tp_p2p_allreduce_bf16": (i32, [vp, vp, sz, vp]),
    "emu_tp_p2p_enable": (i32, [vp, i32]),
    "emu_tp_p2p_set_fenced": (i32, [vp, i32]),
    "emu_tp_p2p_fenced": (i32, [vp]),
    "emu_tp_p2p_giveups": (C.c_uint, []),
    "emu_linear_bf16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp]),
    "emu_linear_fused_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(LinearFxC), vp]),
    "emu_set_splitk_scratch": (None, [vp, sz]),
    "emu_gemm_force_config": (None, [i32]),
    "emu_gemm_tune": (None, [i32]),
    "emu_quantize_fp8_rows": (i32, [vp, i32, vp, i32, vp, i32, i32, vp]),
    "emu_linear_fp8w_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp]),
    "emu_linear_fp8_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "emu_rmsnorm_bf16": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "emu_layernorm_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "emu_layernorm_q8_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "emu_prefetch": (i32, [vp, sz, i32, vp]),
    "emu_gemv_chain_granule_bytes": (sz, [C.POINTER(ChainOpC), i32]),
    "emu_gemv_chain_bf16": (i32, [vp, C.POINTER(ChainOpC), i32, vp, sz, vp, vp]),
    "emu_softmax_rows_bf16": (i32, [vp, vp, i32, i32, i32, i32, f32, vp]),
    "emu_embed_gather_bf16": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "emu_scatter_rows_bf16": (i32, [vp, vp, vp, i32, i32, vp]),
    "emu_argmax_bf16": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "emu_patchify": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "emu_vit_assemble_bf16": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "emu_avgpool_tokens_bf16": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "emu_rope_kv_append_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "emu_transpose_v_bf16": (i32, [vp, lng, lng, lng, vp, i32, i32, i32, i32, i32, vp]),
    "emu_flash_attn_bf16": (i32, [vp, lng, lng, lng, vp, lng, lng, lng, vp, vp, lng, lng, lng, vp,
                                  i32, i32, i32, i32, i32, i32, i32, f32, vp]),
    "emu_decode_attn_ws_bytes": (sz, [i32, i32, i32, i32]),
    "emu_decode_attn_bf16": (i32, [vp, lng, lng, vp, vp, vp, lng, lng, vp, vp, i32, i32, vp, i32, i32, i32, i32, f32, vp]),
    "emu_llama_create": (i32, [vp, C.POINTER(LlamaCfgC), C.POINTER(vp)]),
    "emu_llama_destroy": (None, [vp]),
    "emu_llama_set_layer": (i32, [vp, i32, vp, vp, vp, vp, vp, vp]),
    "emu_llama_set_layer_fp8": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "emu_llama_set_head_fp8": (i32, [vp, vp, vp]),
    "emu_llama_use_fp8": (i32, [vp, i32]),
    "emu_llama_set_head": (i32, [vp, vp, vp, vp, vp, vp]),
    "emu_llama_set_head_shard": (i32, [vp, i32, i32]),
    "emu_llama_set_kv": (i32, [vp, vp, vp, i32, i32]),
    "emu_llama_set_kv_share": (i32, [vp, i32, i32]),
    "emu_beam_step_bf16": (i32, [vp, lng, lng, i32, i32, i32, i32, i32, vp, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "emu_beam_step_workspace_bytes": (sz, [i32, i32, i32]),
    "emu_llama_workspace_bytes": (sz, [vp, i32, i32]),
    "emu_llama_forward": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, sz, vp]),
    "emu_llama_final_norm": (i32, [vp, vp, vp, i32, vp]),
    "emu_llama_logits": (i32, [vp, vp, i32, i32, vp, i32, vp, sz, vp]),
    "emu_llama_greedy_step": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, sz, vp]),
    "emu_vit_create": (i32, [vp, C.POINTER(VitCfgC), C.POINTER(vp)]),
    "emu_vit_destroy": (None, [vp]),
    "emu_vit_set_stem": (i32, [vp, vp, vp, vp, vp]),
    "emu_vit_set_block": (i32, [vp, i32] + [vp] * 12),
    "emu_vit_set_block_fp8": (i32, [vp, i32] + [vp] * 8),
    "emu_vit_use_fp8": (i32, [vp, i32]),
    "emu_vit_set_fusion": (i32, [vp, i32]),
    "emu_vit_workspace_bytes": (sz, [vp, i32]),
    "emu_vit_forward": (i32, [vp, vp, i32, i32, vp, vp, sz, vp]),
    "emu_groupnorm_ws_bytes": (sz, [i32, i32, i32]),
    "emu_groupnorm_nhwc_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "emu_conv3x3_nhwc_bf16": (i32, [vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "emu_unet_create": (i32, [vp, C.POINTER(UNetCfgC), C.POINTER(vp)]),
    "emu_unet_destroy": (None, [vp]),
    "emu_unet_set_weight": (i32, [vp, C.c_char_p, vp]),
    "emu_unet_finalize": (i32, [vp]),
    "emu_unet_set_fusion": (i32, [vp, i32]),
    "emu_unet_use_fp8": (i32, [vp, i32]),
    "emu_unet_temb_total": (i32, [vp]),
    "emu_llama_set_layer_range": (i32, [vp, i32, i32]),
    "emu_llama_set_prefill_fusion": (i32, [vp, i32]),
    "emu_llama_set_tp_overlap": (i32, [vp, i32]),
    "emu_llama_tp_overlap_count": (lng, [vp]),
    "emu_llama_set_decode_tail": (i32, [vp, i32]),
    "emu_llama_set_decode_fused": (i32, [vp, i32, i32]),
    "emu_llama_decode_fused_stats": (i32, [vp, C.POINTER(C.c_uint), C.POINTER(C.c_long)]),
    "emu_llama_set_decode_trace": (i32, [vp, vp]),
    "emu_regress_advance_bf16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "emu_beam_advance": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "emu_llama_beam_reorder_kv": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "emu_vit_blocks": (i32, [vp, vp, i32, i32, i32, vp, sz, vp]),
    "emu_unet_workspace_bytes": (sz, [vp, i32, i32]),
    "emu_unet_context_bytes": (sz, [vp, i32]),
    "emu_unet_set_context": (i32, [vp, vp, i32, vp, i32, vp, sz, vp, sz, vp]),
    "emu_unet_step": (i32, [vp, vp, i32, i32, vp, vp, vp, f32, vp, sz, vp]),
    "emu_unet_forward": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "emu_unet_set_cfg_half": (i32, [vp, i32]),
    "emu_unet_cfg_euler_step": (i32, [vp, vp, vp, i32, i32, vp, vp, f32, vp]),
}


def lib() -> C.CDLL:
    """Load libemu_hip.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EmuHipError(f"{LIB_PATH} not found: build it with `python -m emu_amd.build` "
                              "(there is no CPU fallback in emu_amd)")
        # Bring torch's HIP runtime up BEFORE the library is mapped: mapped first and initialised second, the library's
        # HIP calls fail with hipErrorNoDevice (measured: build() + smoke() in one process); the other order shares one
        # initialised runtime.  No GPU (the CPU-side build / symbol check): nothing to initialise.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
        l = C.CDLL(LIB_PATH)
        l.emu_version.restype, l.emu_version.argtypes = i32, []
        if l.emu_version() != ABI_VERSION:
            raise EmuHipError(f"{LIB_PATH} reports ABI version {l.emu_version()}, this binding needs {ABI_VERSION}: rebuild it "
                              "(`python -m emu_amd.build --force`)")
        if _OVERRIDE:
            import sys
            print(f"emu_amd: EMU_HIP_LIB override active, loaded {LIB_PATH} (tools A/B mode)", file=sys.stderr)
        if os.environ.get("EMU_HIP_TOOLS") == "1" and os.environ.get("EMU_GEMM_TUNE"):
            # tools A/B mode only: start with the given emu_gemm_tune mask (profiling one variant of a whole leg under rocprofv3)
            l.emu_gemm_tune.restype, l.emu_gemm_tune.argtypes = None, [i32]
            l.emu_gemm_tune(int(os.environ["EMU_GEMM_TUNE"]))
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(status: int, what: str, ctx: Optional[int] = None) -> None:
    if status != 0:
        detail = ""
        if ctx:
            msg = lib().emu_last_error(ctx)
            detail = f": {msg.decode()}" if msg else ""
        raise EmuHipError(f"{what} failed with status {status}{detail}")


def declared_symbols() -> list:
    """Function names declared in include/emu_hip.h (used by the symbol-export test)."""
    import re
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(emu_[a-z0-9_]+)\s*\(", txt)))
