"""ctypes front end of oracle/liboracle.so (the plain-C restatement, oracle/oracle.c).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
Arrays cross the boundary in the model dtype's raw storage: float32, or uint16 bit patterns for
float16 / bfloat16.  `to_raw` / `from_raw` convert from/to the float32-valued arrays the numpy oracle uses.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
DT = {"f32": 0, "f16": 1, "bf16": 2}
_lib = None


_FLAGS = ["-O3", "-fopenmp", "-fPIC", "-shared"]


def build():
    """The shipped library targets x86-64-v3 (AVX2/FMA) so that one build runs on any host it travels to."""
    src = os.path.join(_HERE, "oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", *_FLAGS, "-march=x86-64-v3", "-o", LIB, src, "-lm"])


def _build_native():
    """bench.py's cpu_baseline leg: give the host CPU its best code (-march=native, built on the box it runs on,
    outside the tree).  Returns the path or None when there is no compiler."""
    import tempfile
    try:
        out = os.path.join(tempfile.mkdtemp(prefix="spatten_oracle_"), "liboracle_native.so")
        subprocess.check_call(["gcc", *_FLAGS, "-march=native", "-o", out, os.path.join(_HERE, "oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out
    except Exception:
        return None


def load(native=False):
    global _lib
    if _lib is None:
        path = _build_native() if native else None
        if path is None:
            if not os.path.exists(LIB):
                build()
            path = LIB
        _lib = ctypes.CDLL(path)
        _lib.orc_max_threads.restype = ctypes.c_int
        _lib.orc_topk_window.restype = ctypes.c_int
    return _lib


def to_raw(a, dt):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if dt == "f32":
        return a
    if dt == "f16":
        return a.astype(np.float16).view(np.uint16)
    return (a.view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def from_raw(r, dt):
    if dt == "f32":
        return r
    if dt == "f16":
        return r.view(np.float16).astype(np.float32)
    return (r.astype(np.uint32) << np.uint32(16)).view(np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def set_threads(n):
    load().orc_set_threads(ctypes.c_int(n))


def max_threads():
    return load().orc_max_threads()


def attn_decode_raw(dt, q, kc, vc, cos, sin, mask, out, stash, B, H, Hkv, d, N, pos_q):
    """All arrays raw storage, C contiguous: q [B,H,d], kc/vc [B,Hkv,N,d], cos/sin [rows,d/2], out [B,H*d]."""
    load().orc_attn_decode(DT[dt], _p(q), _p(kc), _p(vc), _p(cos), _p(sin), _p(mask), _p(out), _p(stash),
                           B, H, Hkv, d, N, pos_q)


def attn_decode(q, kc, vc, cos_half, sin_half, pos_q, dt, mask=None):
    """float32-valued arrays in / out (like the numpy oracle): returns (out [B,1,H*d], stash [B,H,1,N])."""
    B, H, d = q.shape
    Hkv, N = kc.shape[1], kc.shape[2]
    out = np.empty((B, H * d), dtype=np.float32 if dt == "f32" else np.uint16)
    stash = np.empty((B, H, N), dtype=out.dtype)
    attn_decode_raw(dt, to_raw(q, dt), to_raw(kc, dt), to_raw(vc, dt), to_raw(cos_half, dt), to_raw(sin_half, dt),
                    None if mask is None else to_raw(mask, dt), out, stash, B, H, Hkv, d, N, pos_q)
    return from_raw(out, dt)[:, None, :], from_raw(stash, dt)[:, :, None, :]


def topk_window(score, lo, hi, k, dt="f32"):
    H, L = score.shape
    idx = np.empty((H, k), dtype=np.int32)
    raw = to_raw(score, dt)
    rc = load().orc_topk_window(DT[dt], _p(raw), H, L, lo, hi, k, _p(idx))
    if rc != 0:
        raise ValueError("top-k window holds fewer than k candidates")
    return idx


def kv_compact_raw(dt, src, idx, start, tail_lo):
    B, H, L, d = src.shape
    tail_lo = min(tail_lo, L)
    k = idx.shape[1]
    dst = np.empty((B, H, start + k + (L - tail_lo), d), dtype=src.dtype)
    load().orc_kv_compact(DT[dt], _p(src), _p(dst), _p(np.ascontiguousarray(idx, dtype=np.int32)), B, H, L, d, start, k, tail_lo)
    return dst
