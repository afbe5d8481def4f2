"""ctypes binding of libmadiff_hip.so (the C ABI declared in include/madiff_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing this
module raises, and every op raises RuntimeError on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads torch's bundled libamdhip64 first so this library binds to the SAME HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmadiff_hip.so")
# MADIFF_LIB_PATH: another build of the same library (A/B timing of a kernel change on one box; same ABI required)
LIB_PATH = os.environ.get("MADIFF_LIB_PATH") or LIB_PATH

F32, BF16 = 0, 1

_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float

_SIGS = {
    "madiff_gemm_fwd": [_i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i,
                        _vp, _i, _i, _i, _i, _i, _vp],
    "madiff_event_create": [_vp],
    "madiff_event_destroy": [_vp],
    "madiff_event_create_timed": [_vp],
    "madiff_event_elapsed_ms": [_vp, _vp, _vp],
    "madiff_event_record_external": [_vp, _vp],
    "madiff_stream_wait_event": [_vp, _vp],
    "madiff_capture_join": [_vp, _vp],
    "madiff_stream_capture_status": [_vp],
    "madiff_conv_gn_mish_fwd": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i,
                                _f, _i, _vp, _i, _vp, _i, _i, _vp],
    "madiff_conv_gn_mish_fwd_rv": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i,
                                   _f, _i, _vp, _i, _vp, _i, _i, _vp, _vp],
    "madiff_gemm_tile_count": [],
    "madiff_set_splitk_workspace": [_vp, _l, _vp, _i],
    "madiff_splitk_slots_used": [],
    "madiff_gemm_tile_info": [_i, _vp, _vp, _vp],
    "madiff_gemm_dgrad_wt": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp],
    "madiff_gemm_dgrad_wt_add": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp],
    "madiff_conv_dgrad_gn_bwd": [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp,
                                 _vp, _i, _i, _vp],
    "madiff_conv_dgrad_gn_bwd_chain": [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp,
                                       _vp, _vp, _vp, _i, _vp, _i, _i, _vp],
    "madiff_gemm_wgrad": [_i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i,
                          _vp],
    "madiff_gemm_wgrad_bias": [_i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i,
                               _vp, _vp, _vp],
    "madiff_pack_weight": [_i, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _i, _vp],
    "madiff_pack_weights_multi": [_i, _vp, _i, _i, _vp],
    "madiff_gn_mish_fwd": [_i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _i, _vp],
    "madiff_gn_mish_bwd": [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "madiff_attn_fwd": [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "madiff_attn_bwd": [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "madiff_attn_bwd_rb": [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "madiff_attn_mfma_covers": [_i, _i, _i, _i, _i, _i],
    "madiff_pack_input": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "madiff_loss_ws_doubles": [],
    "madiff_loss_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "madiff_loss_static_init": [],
    "madiff_loss_fwd_static": [_vp, _vp, _vp, _vp, _i, _vp, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "madiff_loss_bwd": [_vp, _vp, _vp, _vp, _i, _vp, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp,
                        _vp],
    "madiff_loss_fwd_seg": [_vp, _vp, _vp, _vp, _i, _vp, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "madiff_loss_bwd_seg": [_vp, _vp, _vp, _vp, _i, _vp, _l, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp,
                            _vp, _vp],
    "madiff_inv_pairs": [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "madiff_unpack_output": [_i, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "madiff_posterior_step": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _l, _vp],
    "madiff_posterior_step_rng": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _l, _vp],
    "madiff_inpaint": [_vp, _vp, _vp, _l, _vp],
    "madiff_encode_first_k": [_vp, _i, _i, _i, _i, _i, _vp],
    "madiff_sampler_fill_time": [_vp, _vp, _vp, _i, _vp],
    "madiff_sampler_advance": [_vp, _vp],
    "madiff_mse_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "madiff_mse_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "madiff_embed_mlp_fwd": [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i,
                             _vp, _i, _i, _i, _i, _vp],
    "madiff_sinusoid": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "madiff_act_fwd": [_i, _i, _vp, _vp, _l, _vp],
    "madiff_act_bwd": [_i, _i, _vp, _vp, _vp, _l, _vp],
    "madiff_colsum": [_i, _vp, _i, _vp, _l, _i, _vp],
    "madiff_colsum_scatter": [_i, _vp, _i, _vp, _l, _i, _vp],
    "madiff_seqsum": [_i, _vp, _i, _vp, _i, _l, _i, _i, _vp],
    "madiff_adam_ema": [_vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _f, _f, _i, _f, _vp, _vp, _i, _vp],
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C madiff_amd/csrc` (or __graft_entry__.build()). "
        "madiff_amd has no CPU or PyTorch fallback.")

_lib = C.CDLL(LIB_PATH)
for _name, _args in _SIGS.items():
    _fn = getattr(_lib, _name)
    _fn.argtypes = _args
    _fn.restype = _i
_lib.madiff_version.restype = _i
_lib.madiff_arch.restype = C.c_char_p


def version() -> int:
    return _lib.madiff_version()


def arch() -> str:
    return _lib.madiff_arch().decode()


def call(name: str, *args) -> None:
    """Call a C-ABI entry point; raise RuntimeError on a non-zero status."""
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}" + (" (invalid argument)" if rc == -1 else " (hipError_t)"))


def call_rc(name: str, *args) -> int:
    """Call an entry point whose positive return values carry information (e.g. 1 = shape not covered)."""
    rc = getattr(_lib, name)(*args)
    if rc < 0 or rc > 1:
        raise RuntimeError(f"{name} failed with status {rc}" + (" (invalid argument)" if rc == -1 else " (hipError_t)"))
    return rc


def tile_count() -> int:
    return _lib.madiff_gemm_tile_count()


def tiles(wt: bool = False):
    """Variant ids of the LDS-DMA GEMM (wt=True: those the transposed-weight data-gradient kernel offers)."""
    n = _lib.madiff_gemm_tile_count()
    return [t for t in range(n) if not wt or _lib.madiff_gemm_tile_info(t, None, None, None) == 1]


def tile_info(tile: int):
    bm, bn, d = C.c_int(), C.c_int(), C.c_int()
    rc = _lib.madiff_gemm_tile_info(tile, C.byref(bm), C.byref(bn), C.byref(d))
    return bm.value, bn.value, d.value, rc == 1


def exported(name: str) -> bool:
    try:
        getattr(_lib, name)
        return True
    except AttributeError:
        return False
