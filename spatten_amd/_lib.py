"""ctypes binding of the C-ABI library ``libspatten_hip.so`` (declared in ``include/spatten.h``).

The library is the product: there is NO CPU fallback.  If it has not been built, or a call is
made with tensors that are not on a ROCm device, the ops raise immediately.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p, POINTER, c_float

ABI_VERSION = 5
DECODE_MAX_SPLITS = 64


class DecodeArgs(ctypes.Structure):
    """``spatten_decode_args_t`` (include/spatten.h): the argument block of ``spatten_attn_decode_args``."""
    _fields_ = [
        ("struct_size", c_uint32), ("dtype", c_int32),
        ("q", c_void_p), ("q_sb", c_int64), ("q_sh", c_int64),
        ("k_cache", c_void_p), ("kr_cache", c_void_p), ("v_cache", c_void_p), ("kv_sb", c_int64), ("kv_sh", c_int64),
        ("k_new", c_void_p), ("v_new", c_void_p), ("new_sb", c_int64), ("new_sh", c_int64),
        ("cos", c_void_p), ("sin", c_void_p), ("table_rows", c_int32), ("pad0_", c_int32),
        ("position_ids", c_void_p), ("pos_sb", c_int64),
        ("mask", c_void_p), ("mask_sb", c_int64),
        ("out", c_void_p), ("out_sb", c_int64),
        ("scores", c_void_p), ("sc_sb", c_int64), ("sc_sh", c_int64),
        ("lse", c_void_p),
        ("workspace", c_void_p), ("workspace_splits", c_int32),
        ("batch", c_int32), ("heads", c_int32), ("kv_heads", c_int32), ("head_dim", c_int32), ("kv_len", c_int32),
        ("pos_q", c_int32), ("n_splits", c_int32),
        ("head_ids", c_void_p), ("n_active_heads", c_int32), ("flags", c_int32),
        ("prev_scores", c_void_p), ("prev_sb", c_int64), ("prev_sh", c_int64), ("prev_lse", c_void_p),
        ("prev_len", c_int32), ("pad1_", c_int32),
        ("importance_acc", c_void_p), ("acc_sh", c_int64),
        ("head_abs_acc", c_void_p),
        ("pq_msb", c_void_p), ("pq_lsb", c_void_p), ("pq_scale", c_void_p),
        ("pq_pl_sb", c_int64), ("pq_pl_sh", c_int64), ("pq_sc_sb", c_int64), ("pq_sc_sh", c_int64),
        ("pq_threshold", c_float), ("pad2_", c_int32),
        ("pq_need_lsb", c_void_p),
        ("step_state", c_void_p), ("kv_len_layout", c_int32), ("pad3_", c_int32),
        ("proj_weight", c_void_p), ("proj_w_sn", c_int64), ("proj_bias", c_void_p), ("proj_out", c_void_p),
        ("proj_out_sb", c_int64), ("proj_n", c_int32), ("pad4_", c_int32),
        ("qkv_x", c_void_p), ("qkv_weight", c_void_p), ("qkv_w_sn", c_int64), ("qkv_bias", c_void_p),
        ("qkv_exchange", c_void_p), ("qkv_hidden", c_int32), ("pad5_", c_int32),
    ]

class ChainLayer(ctypes.Structure):
    """``spatten_chain_layer_t`` (include/spatten.h, ABI 5): one entry of the DEVICE table of a chained decode launch."""
    _fields_ = [
        ("k_cache", c_void_p), ("kr_cache", c_void_p), ("v_cache", c_void_p),
        ("q", c_void_p), ("k_new", c_void_p), ("v_new", c_void_p),
        ("out", c_void_p), ("scores", c_void_p),
        ("head_ids", c_void_p), ("n_active", c_int32), ("pad_", c_int32),
    ]


class ChainArgs(ctypes.Structure):
    """``spatten_chain_args_t`` (include/spatten.h, ABI 5): the argument block of ``spatten_attn_decode_chain``."""
    _fields_ = [
        ("struct_size", c_uint32), ("dtype", c_int32),
        ("layers", c_void_p), ("n_layers", c_int32), ("depth", c_int32),
        ("kv_sb", c_int64), ("kv_sh", c_int64), ("new_sb", c_int64), ("new_sh", c_int64), ("out_sb", c_int64),
        ("sc_sb", c_int64), ("sc_sh", c_int64),
        ("cos", c_void_p), ("sin", c_void_p), ("table_rows", c_int32), ("append", c_int32),
        ("workspace", c_void_p), ("workspace_splits", c_int32),
        ("batch", c_int32), ("heads", c_int32), ("head_dim", c_int32), ("kv_len", c_int32), ("pos_q", c_int32),
        ("n_splits", c_int32), ("max_active", c_int32), ("flags", c_int32), ("kv_len_layout", c_int32),
        ("step_state", c_void_p),
    ]


class PQPlanesDesc(ctypes.Structure):
    """``spatten_pq_planes_t`` (include/spatten.h, ABI 4): profiled key planes + the quantised value plane."""
    _fields_ = [
        ("struct_size", c_uint32), ("key_msb_bits", c_int32), ("value_bits", c_int32), ("pad0_", c_int32),
        ("key_msb", c_void_p), ("key_lsb", c_void_p), ("key_scale", c_void_p),
        ("val_q", c_void_p), ("val_scale", c_void_p), ("msb_logit", c_void_p),
        ("km_sb", c_int64), ("km_sh", c_int64), ("kl_sb", c_int64), ("kl_sh", c_int64),
        ("vq_sb", c_int64), ("vq_sh", c_int64), ("sc_sb", c_int64), ("sc_sh", c_int64),
        ("lg_sb", c_int64), ("lg_sh", c_int64),
    ]


class PQDecodeArgs(ctypes.Structure):
    """``spatten_pq_decode_args_t`` (include/spatten.h, ABI 4)."""
    _fields_ = [
        ("struct_size", c_uint32), ("dtype", c_int32),
        ("q", c_void_p), ("q_sb", c_int64), ("q_sh", c_int64),
        ("planes", POINTER(PQPlanesDesc)),
        ("cos", c_void_p), ("sin", c_void_p), ("table_rows", c_int32), ("pos_q", c_int32),
        ("out", c_void_p), ("out_sb", c_int64),
        ("scores", c_void_p), ("sc_sb", c_int64), ("sc_sh", c_int64),
        ("lse", c_void_p),
        ("need_lsb", c_void_p), ("threshold", c_float), ("flags", c_int32),
        ("workspace", c_void_p), ("workspace_splits", c_int32),
        ("batch", c_int32), ("heads", c_int32), ("kv_heads", c_int32), ("head_dim", c_int32), ("kv_len", c_int32),
        ("n_splits", c_int32), ("kv_len_layout", c_int32),
        ("head_ids", c_void_p), ("n_active_heads", c_int32), ("pad0_", c_int32),
        ("head_abs_acc", c_void_p),
        ("step_state", c_void_p),
        ("k_new", c_void_p), ("v_new", c_void_p), ("new_sb", c_int64), ("new_sh", c_int64),
        ("k_cache", c_void_p), ("kr_cache", c_void_p), ("v_cache", c_void_p), ("kv_sb", c_int64), ("kv_sh", c_int64),
    ]


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspatten_hip.so")
if os.environ.get("SPATTEN_LIB"):          # developer A/B: a variant build of the library (tools/mb/build_variant.sh)
    LIB_PATH = os.environ["SPATTEN_LIB"]

_lib = None


class SpattenLibraryError(RuntimeError):
    pass


class SpattenDeviceTimeout(RuntimeError):
    """A kernel's bounded wait for another workgroup's data expired (SPATTEN_ERR_TIMEOUT)."""


def _declare(lib):
    i, i64, p = c_int, c_int64, c_void_p
    lib.spatten_abi_version.restype = c_int
    lib.spatten_abi_version.argtypes = []
    lib.spatten_status_string.restype = c_char_p
    lib.spatten_status_string.argtypes = [c_int]
    lib.spatten_decode_workspace_bytes.restype = c_size_t
    lib.spatten_decode_workspace_bytes.argtypes = [i, i, i, i]
    lib.spatten_decode_auto_splits.restype = c_int
    lib.spatten_decode_auto_splits.argtypes = [i, i, i, i]
    lib.spatten_attn_decode.restype = c_int
    lib.spatten_attn_decode.argtypes = [
        i, p, i64, i64, p, p, p, i64, i64, p, p, i64, i64, p, p, i, p, i64, p, i64, p, i64, p, i64, i64, p, p,
        i, i, i, i, i, i, i, p]
    lib.spatten_attn_decode_args.restype = c_int
    lib.spatten_attn_decode_args.argtypes = [POINTER(DecodeArgs), p]
    lib.spatten_decode_chain_workspace_bytes.restype = c_size_t
    lib.spatten_decode_chain_workspace_bytes.argtypes = [i, i, i, i, i]
    lib.spatten_attn_decode_chain.restype = c_int
    lib.spatten_attn_decode_chain.argtypes = [POINTER(ChainArgs), p]
    lib.spatten_decode_qkv_exchange_bytes.restype = c_size_t
    lib.spatten_decode_qkv_exchange_bytes.argtypes = [i, i, i]
    lib.spatten_decode_qkv_supported.restype = c_int
    lib.spatten_decode_qkv_supported.argtypes = [i, i, i, i, i, i]
    lib.spatten_decode_workspace_status.restype = c_int
    lib.spatten_decode_workspace_status.argtypes = [p, p]
    lib.spatten_step_state_bytes.restype = c_size_t
    lib.spatten_step_state_bytes.argtypes = [i, i]
    lib.spatten_step_set.restype = c_int
    lib.spatten_step_set.argtypes = [p, i, i, p, p, i, i, i, p]
    lib.spatten_step_advance.restype = c_int
    lib.spatten_step_advance.argtypes = [p, i, i, p, p, i, i, p]
    lib.spatten_gemv.restype = c_int
    lib.spatten_gemv.argtypes = [i, p, i64, p, i64, p, p, i64, i, i, i, p]
    lib.spatten_kv_append.restype = c_int
    lib.spatten_kv_append.argtypes = [i, p, p, i64, i64, i64, p, p, p, i64, i64, p, p, i, i, i, i, i, i, p]
    lib.spatten_kv_append_step.restype = c_int
    lib.spatten_kv_append_step.argtypes = [i, p, p, i64, i64, p, p, p, i64, i64, p, p, p, i64, i64, i64, i64, i, i, i, i, p, p]
    lib.spatten_importance_accumulate.restype = c_int
    lib.spatten_importance_accumulate.argtypes = [i, p, i64, i64, i64, p, p, i64, i64, p, i64, i, i, i, i, i, p]
    lib.spatten_row_lse.restype = c_int
    lib.spatten_row_lse.argtypes = [i, p, i64, i64, i64, p, i64, i64, p, i, i, i, i, i, p]
    lib.spatten_importance_compact.restype = c_int
    lib.spatten_importance_compact.argtypes = [p, i64, p, i64, p, i64, i, i, i, i, i, p]
    lib.spatten_cascade_rank.restype = c_int
    lib.spatten_cascade_rank.argtypes = [i, p, i64, p, i64, p, i64, i, p, i64, i, i, p]
    lib.spatten_head_scores.restype = c_int
    lib.spatten_head_scores.argtypes = [i, p, i64, i64, p, i, i, i, i, p]
    lib.spatten_pv_gather.restype = c_int
    lib.spatten_pv_gather.argtypes = [i, p, i64, i64, p, p, i64, p, i64, i64, p, i64, i, p, i64, i, i, i, i, p, c_size_t, p]
    lib.spatten_pv_gather_workspace_bytes.restype = c_size_t
    lib.spatten_pv_gather_workspace_bytes.argtypes = [i, i, i]
    lib.spatten_pq_pack.restype = c_int
    lib.spatten_pq_pack.argtypes = [i, p, i64, i64, p, p, p, i64, i64, i64, i64, i, i, i, i, i, p]
    lib.spatten_local_v_workspace_bytes.restype = c_size_t
    lib.spatten_local_v_workspace_bytes.argtypes = [i, i]
    lib.spatten_attn_decode_local_v.restype = c_int
    lib.spatten_attn_decode_local_v.argtypes = [i, p, i64, i64, p, p, i64, i64, p, p, i, i, p, i64, p, i64, i64, p, p,
                                                i, i, i, i, i, i, ctypes.c_double, i, p, p]
    lib.spatten_attn_decode_local_v_append.restype = c_int
    lib.spatten_attn_decode_local_v_append.argtypes = [i, p, i64, i64, p, p, i64, i64, p, p, p, i64, i64, p, p, i, i, p, i64, p, i64,
                                                       i64, p, p, i, i, i, i, i, i, ctypes.c_double, i, p, p]
    lib.spatten_pq_plane_row_bytes.restype = c_size_t
    lib.spatten_pq_plane_row_bytes.argtypes = [i, i]
    lib.spatten_pq_pack_planes.restype = c_int
    lib.spatten_pq_pack_planes.argtypes = [i, p, p, i64, i64, POINTER(PQPlanesDesc), i, i, i, i, i, p, p]
    lib.spatten_kv_append_planes.restype = c_int
    lib.spatten_kv_append_planes.argtypes = [i, p, p, i64, i64, p, p, p, i64, i64, POINTER(PQPlanesDesc), p, p, i, i, i, i, i, i, p, p]
    lib.spatten_attn_decode_pq.restype = c_int
    lib.spatten_attn_decode_pq.argtypes = [POINTER(PQDecodeArgs), p]
    lib.spatten_prefill_workspace_bytes.restype = c_size_t
    lib.spatten_prefill_workspace_bytes.argtypes = [i, i, i, i, i, i, i]
    lib.spatten_attn_prefill.restype = c_int
    lib.spatten_attn_prefill.argtypes = [
        i, p, i64, i64, i64, p, p, i64, i64, p, p, i, p, i64, p, i64, i64, p, i64, i64,
        p, i64, i64, i64, p, p, p, i, i, i, i, i, i, i, i, p]
    lib.spatten_importance_prefill_workspace_bytes.restype = c_size_t
    lib.spatten_importance_prefill_workspace_bytes.argtypes = [i, i, i, i]
    lib.spatten_importance_accumulate_prefill.restype = c_int
    lib.spatten_importance_accumulate_prefill.argtypes = [i, p, i64, i64, i64, p, i64, i64, p, p, i, p, i64, p, p, i64, p,
                                                          i, i, i, i, i, i, i, i, p]
    lib.spatten_prefill_pq_workspace_bytes.restype = c_size_t
    lib.spatten_prefill_pq_workspace_bytes.argtypes = [i, i, i, i, i, i, i]
    lib.spatten_attn_prefill_pq.restype = c_int
    lib.spatten_attn_prefill_pq.argtypes = [
        i, p, i64, i64, i64, p, p, p, i64, i64, i64, i64, p, i64, i64, p, p, i, p, i64, p, i64, i64, p, i64, i64,
        p, c_float, p, i, i, i, i, i, i, i, i, p]
    lib.spatten_comm_unique_id.restype = c_int
    lib.spatten_comm_unique_id.argtypes = [p]
    lib.spatten_comm_init.restype = c_int
    lib.spatten_comm_init.argtypes = [POINTER(c_void_p), i, i, p]
    lib.spatten_comm_info.restype = c_int
    lib.spatten_comm_info.argtypes = [p, POINTER(c_int), POINTER(c_int)]
    lib.spatten_comm_destroy.restype = c_int
    lib.spatten_comm_destroy.argtypes = [p]
    lib.spatten_allgather.restype = c_int
    lib.spatten_allgather.argtypes = [p, p, p, c_size_t, p]
    lib.spatten_peer_create.restype = c_int
    lib.spatten_peer_create.argtypes = [POINTER(c_void_p), i, i, c_size_t, p]
    lib.spatten_peer_connect.restype = c_int
    lib.spatten_peer_connect.argtypes = [p, p]
    lib.spatten_peer_allgather.restype = c_int
    lib.spatten_peer_allgather.argtypes = [p, p, p, c_size_t, p]
    lib.spatten_peer_status.restype = c_int
    lib.spatten_peer_status.argtypes = [p, p]
    lib.spatten_peer_destroy.restype = c_int
    lib.spatten_peer_destroy.argtypes = [p]
    lib.spatten_rope_single.restype = c_int
    lib.spatten_rope_single.argtypes = [i, p, i64, i64, i64, p, i64, i64, i64, p, p, i, p, i64, i,
                                        i, i, i, i, p]
    lib.spatten_importance.restype = c_int
    lib.spatten_importance.argtypes = [i, p, i64, i64, i64, p, i64, i, i, i, i, p]
    lib.spatten_topk_select.restype = c_int
    lib.spatten_topk_select.argtypes = [i, p, i64, i, i, i, i, p, i64, p]
    lib.spatten_kv_compact.restype = c_int
    lib.spatten_kv_compact.argtypes = [i, p, p, i64, i64, p, p, p, i64, i64, p, p, i, p, i64, i, i, i, i, i, i, i, p]
    lib.spatten_prune_layers.restype = c_int
    lib.spatten_prune_layers.argtypes = [i, i, p, i64, p, p, i64, i64, p, p, p, i64, i64, p, p, i, p,
                                         i, i, i, i, i, i, i, i, p]


    lib.spatten_prune_layers_scored.restype = c_int
    lib.spatten_prune_layers_scored.argtypes = [i, i, i, p, i64, p, p, i64, i64, p, p, p, i64, i64, p, p, i, p,
                                                p, i64, p, i64, i, i, i, i, i, i, i, i, p]


    lib.spatten_decode_set_team.restype = c_int
    lib.spatten_decode_set_team.argtypes = [c_int]
    lib.spatten_decode_set_gqa.restype = c_int
    lib.spatten_decode_set_gqa.argtypes = [c_int]
    lib.spatten_decode_gqa_selected.restype = c_int
    lib.spatten_decode_gqa_selected.argtypes = [i, i, i, i, i, i]
    lib.spatten_prune_layer_cascade.restype = c_int
    lib.spatten_prune_layer_cascade.argtypes = [i, i, i, p, p, p, p, p, p, p, p, p, p, p, p, i, p, i, p, i64, p, p, i, i, i, i, p]
def load():
    """Load (once) and return the ctypes handle.  Raises SpattenLibraryError if the .so is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpattenLibraryError(
                f"{LIB_PATH} not found: build it with `make lib` (hipcc --offload-arch=gfx950) or "
                f"`python -c 'import __graft_entry__ as g; g.build()'`. spatten_amd has no CPU fallback.")
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # missing libamdhip64 etc.
            raise SpattenLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        _declare(lib)
        if lib.spatten_abi_version() != ABI_VERSION:
            raise SpattenLibraryError("libspatten_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(status: int, what: str):
    if status != 0:
        msg = load().spatten_status_string(status).decode()
        if status == -3:
            raise ValueError(f"{what}: {msg}")
        if status == -5:
            raise SpattenDeviceTimeout(f"{what}: {msg} — the affected outputs were poisoned with NaN")
        raise RuntimeError(f"{what}: {msg} (status {status})")
