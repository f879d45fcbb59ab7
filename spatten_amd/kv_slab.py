"""KV slabs: the device-side layout behind the HF-style ``(K, V)`` tuples.

The reference hands around ``(K_unrotated [B,Hkv,L,d], V [B,Hkv,L,d])`` per layer and re-``cat``s both
on every token (modify_llama.py:95-100).  Here each layer owns three planes of shape [B, Hkv, cap, d]:

    k   un-rotated keys     (what the reference API exposes; only appended to)
    kr  rotated shadow      (row j = RoPE(k row j, position j); what decode streams, see decode_attn.hip)
    v   values

with ``cap`` > L so a new token is appended in place.  The tuple the caller sees holds VIEWS
``k[:, :, :L]``, ``v[:, :, :L]``; the slab object rides along as an attribute of the K view so the next
forward finds the shadow and the spare capacity.  Views that lost the attribute (sliced / copied by the
caller) simply get a fresh slab — correct, only slower for that one step.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops

# capacity granularity (rows): keeps every head's plane a multiple of 32 KiB (bf16, d=128) apart, which
# measured 10-20% faster to stream than odd strides (tools/mb/stride.hip)
CAP_ROUND = 128
# headroom reserved when a slab has to grow (tokens)
GROW = 256

_rope_cache: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def round_capacity(n: int) -> int:
    return (int(n) + CAP_ROUND - 1) // CAP_ROUND * CAP_ROUND


def rope_tables(n: int, d: int, dtype: torch.dtype, device, base: float = 10000.0):
    """Half rotary tables [>=n, d/2] in the model dtype, cached per (device, dtype, d, base), grown on demand."""
    key = (str(device), dtype, d, float(base))
    t = _rope_cache.get(key)
    if t is None or t[0].shape[0] < n:
        rows = max(4096, 1 << (max(n, 1) - 1).bit_length())
        t = _rope_cache[key] = ops.rope_table(rows, d, dtype, device, base)
    return t


class KVSlab:
    __slots__ = ("k", "kr", "v", "length", "rot_len", "base", "__weakref__")

    def __init__(self, k, kr, v, length, rot_len, base=10000.0):
        self.k, self.kr, self.v = k, kr, v          # full-capacity planes [B,Hkv,cap,d]
        self.length = length                        # rows live in k / v
        self.rot_len = rot_len                      # rows of kr that are valid
        self.base = base

    @property
    def capacity(self) -> int:
        return self.k.shape[2]

    def views(self):
        kv = self.k[:, :, :self.length]
        vv = self.v[:, :, :self.length]
        kv._spatten_slab = self
        return kv, vv

    def ensure_shadow(self, upto: int):
        """Rotate rows [rot_len, upto) of k at their slot index into kr (modify_llama.py:103-104)."""
        if self.rot_len < upto:
            B, H, cap, d = self.k.shape
            cos, sin = rope_tables(cap, d, self.k.dtype, self.k.device, self.base)
            ops.build_shadow(self.k, self.kr, self.rot_len, upto, cos, sin)
            self.rot_len = upto


def attach(k_view: torch.Tensor, v_view: torch.Tensor, kr_view: torch.Tensor, length: int, base: float = 10000.0) -> KVSlab:
    """Register freshly produced planes (views ``x[:, :, :length]`` of capacity slabs, see ops.prune_layers)."""
    k, v, kr = (x._base if x._base is not None else x for x in (k_view, v_view, kr_view))
    slab = KVSlab(k, kr, v, length, length, base)
    k_view._spatten_slab = slab
    return slab


def slab_for(k_view: Optional[torch.Tensor], v_view: Optional[torch.Tensor], need: int, batch: int, kv_heads: int,
             d: int, dtype, device, base: float = 10000.0) -> KVSlab:
    """The slab behind a past ``(K, V)`` pair with room for ``need`` rows; creates / grows it when the
    views are foreign tensors (first call, or a caller that re-wrapped them)."""
    P = 0 if k_view is None else k_view.shape[2]
    slab = getattr(k_view, "_spatten_slab", None) if k_view is not None else None
    ok = (slab is not None and slab.length == P and slab.k.data_ptr() == k_view.data_ptr()
          and slab.v.data_ptr() == v_view.data_ptr() and slab.k.stride() == k_view.stride())
    if ok and slab.capacity >= need:
        return slab
    cap = round_capacity(need + GROW)
    k = torch.empty(batch, kv_heads, cap, d, dtype=dtype, device=device)
    kr = torch.empty_like(k)
    v = torch.empty_like(k)
    rot = 0
    if P:
        k[:, :, :P].copy_(k_view)
        v[:, :, :P].copy_(v_view)
        if ok:                                   # growing our own slab: the shadow moves along
            kr[:, :, :slab.rot_len].copy_(slab.kr[:, :, :slab.rot_len])
            rot = slab.rot_len
    return KVSlab(k, kr, v, P, rot, base)
