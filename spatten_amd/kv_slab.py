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

_rope_cache = ops._LRU(8)      # bounded: (device, dtype, d, table identity) -> half tables

# how often slab_for had to re-copy a whole cache because the caller's (K, V) pair had lost its slab — a silent
# performance cliff (one full copy of the layer's cache per forward) made visible: see slab_for
recopy_events = 0


def round_capacity(n: int) -> int:
    return (int(n) + CAP_ROUND - 1) // CAP_ROUND * CAP_ROUND


def rope_tables(n: int, d: int, dtype: torch.dtype, device, base: float = 10000.0, scaling: Optional[tuple] = None):
    """Half rotary tables [>=n, d/2] in the model dtype, cached per (device, dtype, d, base, scaling), grown on demand.
    ``scaling`` = None | ("linear", factor): transformers 4.33 LlamaLinearScalingRotaryEmbedding (config.rope_scaling)."""
    key = (str(device), dtype, d, float(base), scaling)
    t = _rope_cache.get(key)
    if t is None or t[0].shape[0] < n:
        rows = max(4096, 1 << (max(n, 1) - 1).bit_length())
        t = _rope_cache.put(key, ops.rope_table(rows, d, dtype, device, base, scaling))
    return t


# The DecodeGraph (spatten_amd/graph.py) that is warming up / capturing a decode step right now ON THIS THREAD, or None:
# the patched forward then runs its single-token step in the device-length form (ops.StepState) on persistent buffers —
# but only for the slabs that graph is bound to (``graph_ctx_for``): a single-token forward of another model while a
# graph traces takes the ordinary eager path.
import threading

_tls = threading.local()


def set_graph_ctx(ctx):
    """Install ``ctx`` as this thread's tracing DecodeGraph; returns the previous one."""
    prev = getattr(_tls, "graph_ctx", None)
    _tls.graph_ctx = ctx
    return prev


def graph_ctx_for(slab):
    """This thread's tracing DecodeGraph if ``slab`` is one of the slabs it is bound to, else None."""
    ctx = getattr(_tls, "graph_ctx", None)
    if ctx is not None and id(slab) in ctx._offset_of:
        return ctx
    return None


class KVSlab:
    __slots__ = ("k", "kr", "v", "length", "rot_len", "base", "scaling", "pq", "pq_len", "dec", "_tab", "stash",
                 "__weakref__")

    def __init__(self, k, kr, v, length, rot_len, base=10000.0, scaling=None):
        self.k, self.kr, self.v = k, kr, v          # full-capacity planes [B,Hkv,cap,d]
        self.length = length                        # rows live in k / v
        self.rot_len = rot_len                      # rows of kr that are valid
        self.base, self.scaling = float(base), scaling
        self.pq = None                              # ops.PQPlanes of kr (progressive quantisation), built on demand
        self.pq_len = 0                             # rows of the planes that are valid
        self.dec = None                             # ops.SlabDecodeCall: the prefilled argument block of the decode step
        self._tab = None                            # this slab's rotary tables (rows >= capacity), looked up once
        self.stash = None                           # [B, H, cap] persistent stash row of the captured decode step

    @property
    def capacity(self) -> int:
        return self.k.shape[2]

    def tables(self, rows: Optional[int] = None):
        t = self._tab
        if t is None or t[0].shape[0] < (rows or 0):
            B, H, cap, d = self.k.shape
            t = self._tab = rope_tables(max(cap, rows or 0), d, self.k.dtype, self.k.device, self.base, self.scaling)
        return t

    def decode_step_qkv(self, x, qkv_w, qkv_b, heads: int, kv_len: int, pos_q: int, cos, sin, scores, step=None, proj=None):
        """The plain step with its q / k / v projections fused into the attention launch (ops.SlabDecodeCall._run_qkv), or
        None when this slab's shape does not run it (the caller projects and calls ``decode_step``)."""
        B, Hkv, cap, d = self.k.shape
        key = ((B, heads, d), x.dtype, cos.data_ptr())
        dec = self.dec
        if dec is None or dec.key != key:
            dec = self.dec = ops.SlabDecodeCall(self.k, self.kr, self.v, cos, sin, x.new_empty(B, heads, d))
        if not dec.qkv_supported(cap):
            return None
        if step is not None:
            return dec.run(None, None, None, cap, 0, scores, step=step, proj=proj, qkv=(x, qkv_w, qkv_b))
        return dec.run(None, None, None, kv_len, pos_q, scores, layout=cap, proj=proj, qkv=(x, qkv_w, qkv_b))

    def decode_step(self, q, k_new, v_new, kv_len: int, pos_q: int, cos, sin, scores, position_ids=None, mask=None,
                    step=None, proj=None):
        """The plain fused decode step on this slab through its prefilled argument block (ops.SlabDecodeCall).
        The split-N decomposition is laid out for the slab's CAPACITY, whatever the current length: every step of a turn
        — launched with a host length, or (``step``: ops.StepState) with the device-resident length inside a captured
        graph — adds its partials in the same order, so the two forms agree bit for bit."""
        dec = self.dec
        if dec is None or dec.key != (tuple(q.shape), q.dtype, cos.data_ptr()):
            dec = self.dec = ops.SlabDecodeCall(self.k, self.kr, self.v, cos, sin, q)
        cap = self.k.shape[2]
        if step is not None:
            return dec.run(q, k_new, v_new, cap, 0, scores, step=step, proj=proj)
        return dec.run(q, k_new, v_new, kv_len, pos_q, scores, position_ids, mask, layout=cap, proj=proj)

    def stash_row(self, heads: int) -> torch.Tensor:
        """[B, H, cap] in the model dtype, zero-filled once: where a captured decode step leaves its logits (the
        reference allocates a fresh clone per forward, modify_llama.py:116-119; a replayed graph needs a fixed address)."""
        st = self.stash
        if st is None or st.shape[1] != heads:
            B, _, cap, _ = self.k.shape
            st = self.stash = torch.zeros(B, heads, cap, dtype=self.k.dtype, device=self.k.device)
        return st

    def views(self):
        kv = self.k[:, :, :self.length]
        vv = self.v[:, :, :self.length]
        kv._spatten_slab = self
        return kv, vv

    def ensure_shadow(self, upto: int):
        """Rotate rows [rot_len, upto) of k at their slot index into kr (modify_llama.py:103-104)."""
        if self.rot_len < upto:
            cos, sin = self.tables()
            ops.build_shadow(self.k, self.kr, self.rot_len, upto, cos, sin)
            self.rot_len = upto

    def ensure_pq(self, upto: int, profile=None, heads: int = 0):
        """Quantise rows [pq_len, upto) of the rotated shadow into the MSB / LSB planes.  ``profile`` = None: the r02 planes
        (4-bit MSB + 4-bit LSB, V stays in the model dtype); (key MSB bits, value bits): the profiled planes with the
        quantised value plane (ops.PQProfilePlanes; ``heads`` = query heads, for the MSB-logit scratch)."""
        B, Hkv, cap, d = self.k.shape
        if profile is not None:
            kb, vb = profile
            ok = isinstance(self.pq, ops.PQProfilePlanes) and (self.pq.key_bits, self.pq.value_bits) == (kb, vb) \
                and self.pq.msb_logit.shape[1] == (heads or Hkv)
            if not ok:
                self.pq = ops.PQProfilePlanes(B, Hkv, heads or Hkv, cap, d, self.k.device, key_bits=kb, value_bits=vb)
                self.pq_len = 0
            if self.pq_len < upto:
                ops.pq_pack_planes(self.kr, self.v, self.pq, self.pq_len, upto)
                self.pq_len = upto
            return
        if self.pq is None or not isinstance(self.pq, ops.PQPlanes):
            self.pq = ops.PQPlanes(B, Hkv, cap, d, self.k.device)
            self.pq_len = 0
        if self.pq_len < upto:
            ops.pq_pack(self.kr, self.pq, self.pq_len, upto)
            self.pq_len = upto


def attach(k_view: torch.Tensor, v_view: torch.Tensor, kr_view: torch.Tensor, length: int, base: float = 10000.0,
           scaling=None) -> KVSlab:
    """Register freshly produced planes (views ``x[:, :, :length]`` of capacity slabs, see ops.prune_layers)."""
    k, v, kr = (x._base if x._base is not None else x for x in (k_view, v_view, kr_view))
    slab = KVSlab(k, kr, v, length, length, base, scaling)
    k_view._spatten_slab = slab
    return slab


def slab_of(k_view: Optional[torch.Tensor]) -> Optional[KVSlab]:
    return getattr(k_view, "_spatten_slab", None) if k_view is not None else None


def slab_for(k_view: Optional[torch.Tensor], v_view: Optional[torch.Tensor], need: int, batch: int, kv_heads: int,
             d: int, dtype, device, base: float = 10000.0, scaling=None) -> KVSlab:
    """The slab behind a past ``(K, V)`` pair with room for ``need`` rows; creates / grows it when the
    views are foreign tensors (first call, or a caller that re-wrapped them) or were rotated with other tables."""
    global recopy_events
    P = 0 if k_view is None else k_view.shape[2]
    slab = slab_of(k_view)
    ok = (slab is not None and slab.length == P and slab.k.data_ptr() == k_view.data_ptr()
          and slab.v.data_ptr() == v_view.data_ptr() and slab.k.stride() == k_view.stride())
    same_rope = ok and slab.base == float(base) and slab.scaling == scaling
    if ok and same_rope and slab.capacity >= need:
        return slab
    cap = round_capacity(need + GROW)
    k = torch.empty(batch, kv_heads, cap, d, dtype=dtype, device=device)
    kr = torch.zeros_like(k)          # rows past the length are READ by the device-length decode step (weight 0): finite
    v = torch.zeros_like(k)
    rot = 0
    if P:
        if not ok:
            # a (K, V) pair that is not one of our views: its whole cache is copied and re-rotated on this forward.
            # Correct, but a cliff if it happens every step (a caller that clones / re-wraps past_key_values).
            recopy_events += 1
            if recopy_events in (1, 10, 100, 1000):
                import warnings
                warnings.warn(f"spatten_amd: past_key_values arrived without their KV slab ({recopy_events} time(s)): the "
                              f"cache ({P} rows) was copied and re-rotated — keep the tensors the forward returned",
                              RuntimeWarning, stacklevel=3)
        k[:, :, :P].copy_(k_view)
        v[:, :, :P].copy_(v_view)
        if ok and same_rope:                     # growing our own slab: the shadow moves along
            kr[:, :, :slab.rot_len].copy_(slab.kr[:, :, :slab.rot_len])
            rot = slab.rot_len
    return KVSlab(k, kr, v, P, rot, base, scaling)


def reserve(past_key_values, need):
    """Make every layer's slab hold ``need`` rows (one int, or one per layer) and give the rows past the current length
    finite contents (zeros) — what a captured decode step requires (include/spatten.h, "Device-resident step state").
    Returns the list of ``[K, V]`` views to continue with (the same tensors when nothing had to grow)."""
    out = []
    needs = [int(need)] * len(past_key_values) if not isinstance(need, (list, tuple)) else [int(x) for x in need]
    for (K, V), need in zip(past_key_values, needs):
        slab = slab_of(K)
        B, Hkv, P, d = K.shape
        if slab is None:
            raise ValueError("reserve() needs the (K, V) pairs the patched forward / apply_token_pruning returned")
        slab = slab_for(K, V, need, B, Hkv, d, K.dtype, K.device, slab.base, slab.scaling)
        slab.ensure_shadow(P)
        if slab.capacity > P:
            slab.kr[:, :, P:].zero_()
            slab.v[:, :, P:].zero_()
        k, v = slab.views()
        out.append([k, v])
    return out
