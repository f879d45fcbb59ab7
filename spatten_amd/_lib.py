"""ctypes binding of the C-ABI library ``libspatten_hip.so`` (declared in ``include/spatten.h``).

The library is the product: there is NO CPU fallback.  If it has not been built, or a call is
made with tensors that are not on a ROCm device, the ops raise immediately.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_int32, c_int64, c_size_t, c_void_p, POINTER, c_float

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspatten_hip.so")

_lib = None


class SpattenLibraryError(RuntimeError):
    pass


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
    lib.spatten_attn_decode_ex.restype = c_int
    lib.spatten_attn_decode_ex.argtypes = lib.spatten_attn_decode.argtypes[:-1] + [p, i, i, p]
    lib.spatten_importance_accumulate.restype = c_int
    lib.spatten_importance_accumulate.argtypes = [i, p, i64, i64, i64, p, p, i64, i64, p, i64, i, i, i, i, i, p]
    lib.spatten_row_lse.restype = c_int
    lib.spatten_row_lse.argtypes = [i, p, i64, i64, i64, p, i64, i64, p, i, i, i, i, i, p]
    lib.spatten_importance_compact.restype = c_int
    lib.spatten_importance_compact.argtypes = [p, i64, p, i64, p, i64, i, i, i, i, i, p]
    lib.spatten_head_scores.restype = c_int
    lib.spatten_head_scores.argtypes = [i, p, i64, i64, p, i, i, i, i, p]
    lib.spatten_pv_gather.restype = c_int
    lib.spatten_pv_gather.argtypes = [i, p, i64, i64, p, p, i64, p, i64, i64, p, i64, i, p, i64, i, i, i, i, p]
    lib.spatten_pq_pack.restype = c_int
    lib.spatten_pq_pack.argtypes = [i, p, i64, i64, p, p, p, i64, i64, i64, i64, i, i, i, i, i, p]
    lib.spatten_pq_scratch_bytes.restype = c_size_t
    lib.spatten_pq_scratch_bytes.argtypes = [i, i, i, i]
    lib.spatten_attn_decode_pq.restype = c_int
    lib.spatten_attn_decode_pq.argtypes = [i, p, i64, i64, p, p, p, i64, i64, i64, i64, p, i64, i64, p, p, i, i, c_float,
                                           p, i64, p, p, p, i, i, i, i, i, p]
    lib.spatten_prefill_workspace_bytes.restype = c_size_t
    lib.spatten_prefill_workspace_bytes.argtypes = [i, i, i, i, i, i, i]
    lib.spatten_attn_prefill.restype = c_int
    lib.spatten_attn_prefill.argtypes = [
        i, p, i64, i64, i64, p, p, i64, i64, p, p, i, p, i64, p, i64, i64, p, i64, i64,
        p, i64, i64, i64, p, p, i, i, i, i, i, i, i, i, p]
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
        if lib.spatten_abi_version() != 1:
            raise SpattenLibraryError("libspatten_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(status: int, what: str):
    if status != 0:
        msg = load().spatten_status_string(status).decode()
        if status == -3:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what}: {msg} (status {status})")
