"""torch-tensor front end of the C ABI (``include/spatten.h``).

PyTorch is plumbing here: it owns device memory and the stream; every op below passes raw
``data_ptr()``s, strides and ``torch.cuda.current_stream().cuda_stream`` to libspatten_hip.so.
"""
from __future__ import annotations

import ctypes
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"spatten_amd supports float32/float16/bfloat16, got {t.dtype}") from None


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("spatten_amd ops need ROCm device tensors (there is no CPU path in the product)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current HIP stream of the current device as an integer handle.  torch.cuda.current_stream() costs ~10 us of
    Python per call (device-index normalisation, a Stream object) — a third of a decode launch's host time — so the raw
    accessors are used when this torch build has them."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# rotary table in the model dtype, first d/2 columns (transformers 4.33 LlamaRotaryEmbedding)
# ------------------------------------------------------------------------------------------------
def rope_table(n: int, d: int, dtype: torch.dtype, device, base: float = 10000.0,
               scaling: Optional[tuple] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos, sin [n, d/2] = the distinct half of the 4.33 table ``emb = cat(freqs, freqs)``
    (modify_llama.py:89).  Computed in fp32 on the host CPU exactly like the reference module
    (inv_freq, outer product, cos/sin, ``.to(dtype)``) and uploaded once.
    ``scaling``: None (LlamaRotaryEmbedding) or ("linear", f): positions / f (transformers 4.33
    LlamaLinearScalingRotaryEmbedding, config.rope_scaling = {"type": "linear", "factor": f})."""
    t = torch.arange(n, dtype=torch.float32)
    if scaling is not None:
        if scaling[0] != "linear":
            raise NotImplementedError(f"rope scaling {scaling!r}")
        t = t / float(scaling[1])
    inv_freq = 1.0 / (base ** (torch.arange(0, d, 2).float() / d))
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return freqs.cos().to(dtype).to(device).contiguous(), freqs.sin().to(dtype).to(device).contiguous()


class _LRU:
    """Small bounded cache for per-(shape, stream) scratch buffers: a process that cycles through many streams or
    shapes does not accumulate device memory for its lifetime (oldest entry dropped first)."""

    def __init__(self, capacity: int):
        self.capacity, self.d = capacity, {}

    def get(self, key):
        v = self.d.get(key)
        if v is not None:
            self.d[key] = self.d.pop(key)          # move to the young end
        return v

    def put(self, key, value):
        self.d.pop(key, None)
        self.d[key] = value
        while len(self.d) > self.capacity:
            self.d.pop(next(iter(self.d)))
        return value

    def values(self):
        return list(self.d.values())


class DecodeWorkspace:
    """Split-N partials + arrival counters + error word (zero-filled once, re-armed by the kernel)."""

    def __init__(self, batch: int, heads: int, head_dim: int, device, max_splits: int = _lib.DECODE_MAX_SPLITS):
        lib = _lib.load()
        self.max_splits = max_splits
        self.key = (batch, heads, head_dim)
        nbytes = lib.spatten_decode_workspace_bytes(batch, heads, head_dim, max_splits)
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self._xch = None            # exchange scratch of the fused projection + attention launch, allocated on first use

    @property
    def xch(self) -> torch.Tensor:
        if self._xch is None:
            b, h, d = self.key
            n = _lib.load().spatten_decode_qkv_exchange_bytes(b, h, d)
            self._xch = torch.zeros(n, dtype=torch.uint8, device=self.buf.device)
        return self._xch

    def check(self):
        """Synchronises the current stream; raises SpattenDeviceTimeout if a split-N merge of an earlier launch gave
        up waiting for a partial (its outputs were poisoned with NaN)."""
        _lib.check(_lib.load().spatten_decode_workspace_status(self.buf.data_ptr(), _stream()), "decode workspace")


_ws_cache = _LRU(16)

# While a DecodeGraph traces a step (spatten_amd/graph.py) every workspace a launch uses is also appended here: a captured
# graph bakes the workspace's raw pointers in, so the graph — not only this LRU — has to keep the buffers alive (an entry
# evicted by other streams / shapes would otherwise be replayed into freed memory).
# The list is per THREAD, like the tracing context it belongs to (kv_slab.set_graph_ctx): two threads tracing at once — or one
# tracing while another finishes — must not restore each other's list mid-trace.
_pins_tls = threading.local()


def set_ws_pins(pins: Optional[list]) -> Optional[list]:
    """Install this thread's pin list (None = not tracing); returns the previous one."""
    prev = getattr(_pins_tls, "pins", None)
    _pins_tls.pins = pins
    return prev


def _pin(ws):
    pins = getattr(_pins_tls, "pins", None)
    if pins is not None and not any(w is ws for w in pins):
        pins.append(ws)
    return ws


def _workspace(batch, heads, head_dim, device, stream: Optional[int] = None) -> DecodeWorkspace:
    key = (batch, heads, head_dim, device, _stream() if stream is None else stream)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = _ws_cache.put(key, DecodeWorkspace(batch, heads, head_dim, device))
    return _pin(ws)


def check_workspaces():
    """Error words of every cached decode workspace (one stream sync each) — called at natural sync points."""
    for ws in _ws_cache.values():
        ws.check()


def attn_decode(q: torch.Tensor, k_cache: Optional[torch.Tensor], kr_cache: Optional[torch.Tensor], v_cache: torch.Tensor,
                kv_len: int, cos: torch.Tensor, sin: torch.Tensor, pos_q: int,
                k_new: Optional[torch.Tensor] = None, v_new: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                scores: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None,
                n_splits: int = 0, workspace: Optional[DecodeWorkspace] = None,
                head_ids: Optional[torch.Tensor] = None, scores_only: bool = False,
                cascade: Optional[tuple] = None, pq: Optional[tuple] = None,
                head_abs: Optional[torch.Tensor] = None, step: Optional["StepState"] = None,
                layout: int = 0, proj: Optional[tuple] = None) -> torch.Tensor:
    """Fused decode attention (modify_llama.py:86-147 at q_len=1).

    q [B,H,d]; k_cache (un-rotated, only appended to) / kr_cache (rotated shadow, see build_shadow) /
    v_cache [B,Hkv,cap,d] with rows [0,kv_len) live (row kv_len-1 is written from k_new/v_new [B,Hkv,d]
    when given); cos/sin [>=kv_len, d/2]; mask [B,kv_len];
    position_ids optional int64 [B] device tensor (overrides pos_q without a host sync);
    scores (stash) [B,H,>=kv_len]; lse [B,H,2] fp32 (row max, sum exp); head_ids int32 ascending list of the
    heads to run (head pruning; rows of the others are left untouched); scores_only: stash + lse only, no V
    traffic (pass 1 of local V pruning).
    cascade = (acc [H,>=prev_len] fp32, prev_scores [B,H,>=prev_len], prev_lse [B,H,2], prev_len): the PREVIOUS decode
    step's softmax probabilities are added to acc while this step streams (cumulative importance, no extra launch).
    With ``step``: scores / lse and prev_scores / prev_lse are the two buffers that swap roles every step (prev_len unused).
    pq = (PQPlanes, threshold, need_lsb int32 [B*H]): keys come from the progressive-quantisation planes (MSB pass,
    LSB refetch for heads whose max probability is below threshold) instead of kr_cache.
    head_abs fp32 [B*H]: += sum |out| per (b, h) (cumulative head importance for head pruning).
    layout: lay the split-N decomposition out for this length (>= kv_len) instead of kv_len.
    step (StepState): the device-resident length — ``kv_len`` is then the BOUND of the launch, ``pos_q`` is not read, rows
    [length, bound) of kr_cache / v_cache must hold finite values.
    proj = (weight [N, H*d], bias [N] or None, proj_out [B, N]): the step's output projection (modify_llama.py:163) issued by
    the same C call (a second launch of spatten_gemv's kernel: one host call per layer-step); ``proj_out`` is filled.
    Returns out [B, H*d]."""
    _dev(q, k_cache, kr_cache, v_cache, cos, sin, k_new, v_new, mask, out, scores, lse, position_ids, head_abs)
    if position_ids is not None and position_ids.dtype != torch.int64:
        raise TypeError("position_ids must be int64")
    lib = _lib.load()
    B, H, d = q.shape
    Hkv, cap = v_cache.shape[1], v_cache.shape[2]
    if q.stride(2) != 1 or v_cache.stride(3) != 1 or v_cache.stride(2) != d \
            or (kr_cache is not None and kr_cache.stride() != v_cache.stride()) \
            or (k_cache is not None and k_cache.stride() != v_cache.stride()):
        raise ValueError("q/k_cache/kr_cache/v_cache need contiguous rows (pitch d) and identical strides")
    if kr_cache is None and pq is None:
        raise ValueError("kr_cache (or pq planes) required")
    if k_new is not None and k_cache is None:
        raise ValueError("appending needs the un-rotated k_cache")
    if kv_len > cap or max(kv_len, pos_q + 1) > cos.shape[0] or cos.shape[1] * 2 != d:
        raise ValueError("kv_len exceeds cache capacity or rotary table")
    if out is None:
        out = torch.empty(B, H * d, dtype=q.dtype, device=q.device)
    if (k_new is None) != (v_new is None):
        raise ValueError("k_new and v_new come together")
    if k_new is not None and (k_new.stride(2) != 1 or v_new.stride() != k_new.stride()):
        raise ValueError("k_new/v_new need contiguous rows")
    if mask is not None and mask.stride(-1) != 1:
        raise ValueError("mask rows must be contiguous")
    stream = _stream()
    ws = _pin(workspace) if workspace is not None else _workspace(B, H, d, q.device, stream)
    if n_splits > ws.max_splits:
        raise ValueError("n_splits exceeds workspace")
    if head_ids is not None and (head_ids.dtype != torch.int32 or not head_ids.is_cuda or head_ids.dim() != 1):
        raise TypeError("head_ids must be a 1-D int32 device tensor")
    a = _lib.DecodeArgs()
    a.struct_size = ctypes.sizeof(_lib.DecodeArgs)
    a.dtype = _dt(q)
    a.q, a.q_sb, a.q_sh = q.data_ptr(), (H * d if B == 1 else q.stride(0)), q.stride(1)   # B = 1: the batch stride is never applied
    a.k_cache, a.kr_cache, a.v_cache = _ptr(k_cache), _ptr(kr_cache), v_cache.data_ptr()
    a.kv_sb, a.kv_sh = v_cache.stride(0), v_cache.stride(1)
    if k_new is not None:
        a.k_new, a.v_new, a.new_sb, a.new_sh = k_new.data_ptr(), v_new.data_ptr(), k_new.stride(0), k_new.stride(1)
    a.cos, a.sin, a.table_rows = cos.data_ptr(), sin.data_ptr(), cos.shape[0]
    if position_ids is not None:
        a.position_ids, a.pos_sb = position_ids.data_ptr(), position_ids.stride(0)
    if mask is not None:
        a.mask, a.mask_sb = mask.data_ptr(), mask.stride(0)
    a.out, a.out_sb = out.data_ptr(), out.stride(0)
    if scores is not None:
        a.scores, a.sc_sb, a.sc_sh = scores.data_ptr(), scores.stride(0), scores.stride(1)
    a.lse = _ptr(lse)
    a.workspace, a.workspace_splits = ws.buf.data_ptr(), ws.max_splits
    a.batch, a.heads, a.kv_heads, a.head_dim, a.kv_len, a.pos_q, a.n_splits = B, H, Hkv, d, kv_len, pos_q, n_splits
    if head_ids is not None:
        a.head_ids, a.n_active_heads = head_ids.data_ptr(), head_ids.numel()
    a.flags = 1 if scores_only else 0
    if proj is not None:
        _fill_proj(a, proj, q, B, H * d)
    a.kv_len_layout = int(layout)
    # (a static step only lays its splits out for `layout`: every load is clamped to the rows below kv_len, so the layout may
    #  exceed the capacity — e.g. to give the unit's merging split a short chunk; the device-length form ignores it)
    if step is not None:
        if scores is not None and scores.shape[2] < kv_len:
            raise ValueError("device-length step: the stash row must cover the bound")
        a.step_state = step.data_ptr()
    if cascade is not None:
        acc, prev_scores, prev_lse, prev_len = cascade
        _dev(acc, prev_scores, prev_lse)
        if acc.dtype != torch.float32 or acc.stride(1) != 1 or acc.shape[0] != H or acc.shape[1] < prev_len \
                or prev_scores.stride(-1) != 1 or prev_lse.dtype != torch.float32 or not prev_lse.is_contiguous():
            raise ValueError("cascade: acc fp32 [H, >=prev_len], prev_scores [B,H,>=prev_len] rows contiguous, prev_lse [B,H,2]")
        a.importance_acc, a.acc_sh = acc.data_ptr(), acc.stride(0)
        a.prev_scores, a.prev_sb, a.prev_sh = prev_scores.data_ptr(), prev_scores.stride(0), prev_scores.stride(1)
        a.prev_lse, a.prev_len = prev_lse.data_ptr(), prev_len
    if head_abs is not None:
        if head_abs.dtype != torch.float32 or head_abs.numel() != B * H or not head_abs.is_contiguous():
            raise ValueError("head_abs must be a contiguous fp32 [B*H] tensor")
        a.head_abs_acc = head_abs.data_ptr()
    if pq is not None:
        planes, threshold, need_lsb = pq
        _dev(planes.msb, need_lsb)
        a.pq_msb, a.pq_lsb, a.pq_scale = planes.msb.data_ptr(), planes.lsb.data_ptr(), planes.scale.data_ptr()
        a.pq_pl_sb, a.pq_pl_sh = planes.msb.stride(0), planes.msb.stride(1)
        a.pq_sc_sb, a.pq_sc_sh = planes.scale.stride(0), planes.scale.stride(1)
        a.pq_threshold, a.pq_need_lsb = float(threshold), need_lsb.data_ptr()
    _lib.check(lib.spatten_attn_decode_args(ctypes.byref(a), stream), "spatten_attn_decode")
    return out


class DecodeChain:
    """The attention step of ALL layers of one decode token as ONE launch (``spatten_attn_decode_chain``, include/spatten.h
    ABI 5): what the caller's layer loop issues as one ``attn_decode`` per LlamaAttention module (modify_llama.py:86-147).

    Built once from the per-layer tensors (their pointers go into a DEVICE table, so a captured graph replays without
    patching); every layer shares the plane strides, ``kv_len`` / ``pos_q`` (or the device-resident ``step``) and the dtype.
    ``q`` [B,H,d] dense, ``k_new`` / ``v_new`` [B,H,d] or None (no append), planes [B,H,cap,d], ``out`` [B,H*d], ``scores``
    [B,H,>=kv_len] or None, ``head_ids`` per layer: int32 tensor (possibly empty) or None.  MHA, bf16 / f16, d = 128."""

    def __init__(self, q, k_cache, kr_cache, v_cache, out, k_new=None, v_new=None, scores=None, head_ids=None,
                 max_splits: int = 16, depth: int = 0):
        L = len(q)
        lib = _lib.load()
        v0 = v_cache[0]
        B, H, cap, d = v0.shape
        self.L, self.B, self.H, self.d, self.cap = L, B, H, d, cap
        self.dtype = _dt(v0)
        self.append = k_new is not None
        tab = (_lib.ChainLayer * L)()
        self.max_active = 0
        for l in range(L):
            ts = [q[l], kr_cache[l], v_cache[l], out[l]] + ([k_cache[l]] if k_cache is not None else []) + \
                 ([k_new[l], v_new[l]] if self.append else []) + ([scores[l]] if scores is not None else [])
            _dev(*ts)
            if v_cache[l].shape != v0.shape or v_cache[l].stride() != v0.stride() or kr_cache[l].stride() != v0.stride() \
                    or v0.stride(3) != 1 or v0.stride(2) != d or not q[l].is_contiguous() or q[l].shape != (B, H, d) \
                    or (k_cache is not None and k_cache[l].stride() != v0.stride()) or out[l].stride(-1) != 1 \
                    or out[l].stride(0) != out[0].stride(0):
                raise ValueError("chained decode: every layer's planes share shape and strides (rows contiguous), q dense [B,H,d]")
            if self.append and (k_new[l].shape != (B, H, d) or k_new[l].stride(2) != 1 or k_new[l].stride() != k_new[0].stride()
                                or v_new[l].stride() != k_new[0].stride()):
                raise ValueError("chained decode: k_new / v_new [B,H,d] with common strides")
            if scores is not None and (scores[l].stride(2) != 1 or scores[l].stride() != scores[0].stride()):
                raise ValueError("chained decode: stash rows contiguous, common strides")
            e = tab[l]
            e.k_cache = k_cache[l].data_ptr() if k_cache is not None else None
            e.kr_cache, e.v_cache = kr_cache[l].data_ptr(), v_cache[l].data_ptr()
            e.q = q[l].data_ptr()
            e.k_new = k_new[l].data_ptr() if self.append else None
            e.v_new = v_new[l].data_ptr() if self.append else None
            e.out = out[l].data_ptr()
            e.scores = scores[l].data_ptr() if scores is not None else None
            ids = None if head_ids is None else head_ids[l]
            if ids is not None:
                if ids.dtype != torch.int32 or not ids.is_cuda or ids.dim() != 1:
                    raise TypeError("head_ids: int32 device vectors")
                e.head_ids = ids.data_ptr() if ids.numel() else None
                e.n_active = int(ids.numel())
            else:
                e.head_ids, e.n_active = None, H
            self.max_active = max(self.max_active, e.n_active)
        raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8)
        self.table = raw.to(v0.device)
        self.keep = (q, k_cache, kr_cache, v_cache, out, k_new, v_new, scores, head_ids)      # the table holds raw pointers
        self.kv_sb, self.kv_sh = v0.stride(0), v0.stride(1)
        self.new_sb, self.new_sh = (k_new[0].stride(0), k_new[0].stride(1)) if self.append else (0, 0)
        self.out_sb = out[0].stride(0)
        self.sc_sb, self.sc_sh = (scores[0].stride(0), scores[0].stride(1)) if scores is not None else (0, 0)
        self.max_splits, self.depth = max_splits, depth
        self.ws = torch.zeros(lib.spatten_decode_chain_workspace_bytes(L, B, H, d, max_splits), dtype=torch.uint8, device=v0.device)

    def check(self):
        """Synchronises the current stream; raises SpattenDeviceTimeout if a wait of an earlier launch gave up."""
        _lib.check(_lib.load().spatten_decode_workspace_status(self.ws.data_ptr(), _stream()), "decode chain workspace")

    def __call__(self, kv_len: int, cos: torch.Tensor, sin: torch.Tensor, pos_q: int, n_splits: int = 0,
                 step: Optional["StepState"] = None, layout: int = 0):
        """One token: every layer's step at cache length ``kv_len`` (counting the appended row) / query position ``pos_q``.
        Raises NotImplementedError where the chained launch does not apply (launch the layers one by one)."""
        _dev(cos, sin)
        if kv_len > self.cap or (step is None and max(kv_len, pos_q + 1) > cos.shape[0]) or cos.shape[1] * 2 != self.d or (step is not None and layout > self.cap):
            raise ValueError("kv_len exceeds cache capacity or rotary table")
        a = _lib.ChainArgs()
        a.struct_size = ctypes.sizeof(_lib.ChainArgs)
        a.dtype = self.dtype
        a.layers, a.n_layers, a.depth = self.table.data_ptr(), self.L, self.depth
        a.kv_sb, a.kv_sh, a.new_sb, a.new_sh, a.out_sb = self.kv_sb, self.kv_sh, self.new_sb, self.new_sh, self.out_sb
        a.sc_sb, a.sc_sh = self.sc_sb, self.sc_sh
        a.cos, a.sin, a.table_rows, a.append = cos.data_ptr(), sin.data_ptr(), cos.shape[0], 1 if self.append else 0
        a.workspace, a.workspace_splits = self.ws.data_ptr(), self.max_splits
        a.batch, a.heads, a.head_dim, a.kv_len, a.pos_q = self.B, self.H, self.d, int(kv_len), int(pos_q)
        a.n_splits, a.max_active, a.flags, a.kv_len_layout = int(n_splits), self.max_active, 0, int(layout)
        a.step_state = None if step is None else step.data_ptr()
        _pin(self)
        rc = _lib.load().spatten_attn_decode_chain(ctypes.byref(a), _stream())
        if rc == -2:
            raise NotImplementedError("the chained decode launch does not cover this shape (see include/spatten.h)")
        _lib.check(rc, "spatten_attn_decode_chain")


def attn_decode_qkv(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], heads: int, k_cache: torch.Tensor,
                    kr_cache: torch.Tensor, v_cache: torch.Tensor, kv_len: int, cos: torch.Tensor, sin: torch.Tensor, pos_q: int,
                    scores: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, n_splits: int = 0,
                    step: Optional["StepState"] = None, layout: int = 0,
                    workspace: Optional[DecodeWorkspace] = None, proj=None):
    """The plain decode step with its q / k / v projections INSIDE the launch (spatten_decode_args_t::qkv_*,
    modify_llama.py:72-74 + :86-147): x [hidden] (or [1, 1, hidden]) the layer's input row, weight [3*H*d, hidden] = q_proj /
    k_proj / v_proj stacked, bias or None; the new token's K / V rows are appended at row kv_len - 1.  Raises
    NotImplementedError where the fused launch does not apply (spatten_decode_qkv_supported).  Returns out [1, H*d].
    ``proj`` = (o_proj weight [N, H*d], bias or None[, y [1, N]]): the step's output projection too (modify_llama.py:163) —
    inside the same launch when N = 16 x its workgroups and H*d = 4096 (Llama-2-7B), one spatten_gemv launch behind it
    otherwise, the same bits either way; returns (out, y)."""
    _dev(x, weight, bias, k_cache, kr_cache, v_cache, cos, sin, scores, out)
    Hkv, cap, d = v_cache.shape[1], v_cache.shape[2], v_cache.shape[3]
    K = weight.shape[1]
    if v_cache.shape[0] != 1 or Hkv != heads or x.numel() != K or x.stride(-1) != 1 or weight.stride(1) != 1 \
            or weight.shape[0] != 3 * heads * d or weight.dtype != x.dtype or v_cache.stride(3) != 1 or v_cache.stride(2) != d \
            or kr_cache.stride() != v_cache.stride() or k_cache.stride() != v_cache.stride():
        raise ValueError("fused projection step: batch 1, MHA, x [hidden], weight [3*H*d, hidden], slab planes with contiguous rows")
    if kv_len > cap or (step is None and max(kv_len, pos_q + 1) > cos.shape[0]) or layout > cap:
        raise ValueError("kv_len exceeds cache capacity or rotary table")
    lib = _lib.load()
    if out is None:
        out = torch.empty(1, heads * d, dtype=x.dtype, device=x.device)
    stream = _stream()
    ws = _pin(workspace) if workspace is not None else _workspace(1, heads, d, x.device, stream)
    a = _lib.DecodeArgs()
    a.struct_size = ctypes.sizeof(_lib.DecodeArgs)
    a.dtype = _dt(x)
    a.k_cache, a.kr_cache, a.v_cache = k_cache.data_ptr(), kr_cache.data_ptr(), v_cache.data_ptr()
    a.kv_sb, a.kv_sh = v_cache.stride(0), v_cache.stride(1)
    a.cos, a.sin, a.table_rows = cos.data_ptr(), sin.data_ptr(), cos.shape[0]
    a.out, a.out_sb = out.data_ptr(), out.stride(0)
    if scores is not None:
        a.scores, a.sc_sb, a.sc_sh = scores.data_ptr(), scores.stride(0), scores.stride(1)
    a.workspace, a.workspace_splits = ws.buf.data_ptr(), ws.max_splits
    a.batch, a.heads, a.kv_heads, a.head_dim, a.kv_len, a.pos_q, a.n_splits = 1, heads, Hkv, d, int(kv_len), int(pos_q), int(n_splits)
    a.kv_len_layout = int(layout)
    a.step_state = None if step is None else step.data_ptr()
    a.qkv_x, a.qkv_weight, a.qkv_w_sn, a.qkv_bias = x.data_ptr(), weight.data_ptr(), weight.stride(0), _ptr(bias)
    a.qkv_exchange, a.qkv_hidden = ws.xch.data_ptr(), K
    y = None
    if proj is not None:
        y = proj[2] if len(proj) > 2 and proj[2] is not None else torch.empty(1, proj[0].shape[0], dtype=x.dtype, device=x.device)
        _fill_proj(a, (proj[0], proj[1], y.view(1, -1)), out, 1, heads * d)
    rc = lib.spatten_attn_decode_args(ctypes.byref(a), stream)
    if rc == -2:
        raise NotImplementedError("the fused projection + attention launch does not cover this step (spatten_decode_qkv_supported)")
    _lib.check(rc, "spatten_attn_decode (fused projections)")
    return out if proj is None else (out, y)


def _fill_proj(a, proj, q, B, K):
    """proj = (weight [N, K], bias or None, out [B, N]) -> the proj_* fields of a decode argument block."""
    w, bias, y = proj
    _dev(w, bias, y)
    if w.dtype != q.dtype or w.dim() != 2 or w.shape[1] != K or w.stride(1) != 1 or y.dtype != q.dtype \
            or y.shape[0] != B or y.shape[-1] != w.shape[0] or y.stride(-1) != 1 \
            or (bias is not None and (bias.dtype != q.dtype or bias.numel() != w.shape[0] or not bias.is_contiguous())):
        raise ValueError("proj: weight [N, H*d] rows contiguous, bias [N], out [B, N], all in the model dtype")
    a.proj_weight, a.proj_w_sn, a.proj_bias = w.data_ptr(), w.stride(0), _ptr(bias)
    a.proj_out, a.proj_out_sb, a.proj_n = y.data_ptr(), y.stride(0), w.shape[0]


class StepState:
    """Device-resident step state (include/spatten.h, ABI 3): the cache length after the next append, the query's rotary
    position and the rotary rows of both.  One per token stream (all layers of a model share it); ``advance`` is a stream
    operation without host values, so a token's launch sequence — advance, then every layer's decode launch with
    ``step=state`` — is captured once into a HIP graph and replayed for each token."""

    def __init__(self, cos: torch.Tensor, sin: torch.Tensor):
        _dev(cos, sin)
        self.lib = _lib.load()
        self.cos, self.sin = cos, sin                      # half tables [rows, d/2] in the model dtype (kept alive)
        self.d = cos.shape[1] * 2
        self.dt = _dt(cos)
        nbytes = self.lib.spatten_step_state_bytes(self.dt, self.d)
        if nbytes == 0:
            raise ValueError("unsupported head_dim / dtype for a step state")
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=cos.device)

    def data_ptr(self) -> int:
        return self.buf.data_ptr()

    def set(self, kv_len: int, pos_q: int):
        """Host values -> state (not capturable: the values are launch arguments).  ``kv_len`` = the current cache length;
        after the next ``advance()`` the state describes the step that appends row ``kv_len``."""
        if max(kv_len, pos_q + 1) > self.cos.shape[0]:
            raise ValueError("step state beyond the rotary table")
        _lib.check(self.lib.spatten_step_set(self.buf.data_ptr(), self.dt, self.d, self.cos.data_ptr(), self.sin.data_ptr(),
                                             self.cos.shape[0], int(kv_len), int(pos_q), _stream()), "spatten_step_set")

    def advance(self, delta: int = 1):
        _lib.check(self.lib.spatten_step_advance(self.buf.data_ptr(), self.dt, self.d, self.cos.data_ptr(), self.sin.data_ptr(),
                                                 self.cos.shape[0], int(delta), _stream()), "spatten_step_advance")

    def read(self, all_words: bool = False):
        """(kv_len, pos_q) [, steps since the last set, rows of the previous step's stash] — synchronises; for tests."""
        w = self.buf[:16].view(torch.int32).cpu()
        return tuple(int(x) for x in w) if all_words else (int(w[0]), int(w[1]))


class SlabDecodeCall:
    """A prefilled argument block for the plain decode step on ONE KV slab (host-path fast lane of the patched forward:
    a single-token step is host-bound, and filling ~35 ctypes fields + re-validating the slab costs more than the launch).
    Everything that does not change from token to token — the slab planes and their strides, the rotary tables, the
    workspace, the geometry — is validated and written once; ``run`` sets the per-token fields and launches."""

    def __init__(self, k_cache, kr_cache, v_cache, cos, sin, q: torch.Tensor):
        _dev(q, k_cache, kr_cache, v_cache, cos, sin)
        B, H, d = q.shape
        Hkv, cap = v_cache.shape[1], v_cache.shape[2]
        if v_cache.stride(3) != 1 or v_cache.stride(2) != d or kr_cache.stride() != v_cache.stride() \
                or k_cache.stride() != v_cache.stride() or cos.shape[1] * 2 != d:
            raise ValueError("slab planes need contiguous rows (pitch d) and identical strides")
        self.key = (tuple(q.shape), q.dtype, cos.data_ptr())
        self.cap, self.table_rows, self.B, self.H, self.d = cap, cos.shape[0], B, H, d
        self.keep = (k_cache, kr_cache, v_cache, cos, sin)        # the block holds raw pointers: keep the tensors alive
        self.stream = _stream()
        self.ws = _workspace(B, H, d, q.device, self.stream)
        a = self.a = _lib.DecodeArgs()
        a.struct_size = ctypes.sizeof(_lib.DecodeArgs)
        a.dtype = _dt(q)
        a.k_cache, a.kr_cache, a.v_cache = k_cache.data_ptr(), kr_cache.data_ptr(), v_cache.data_ptr()
        a.kv_sb, a.kv_sh = v_cache.stride(0), v_cache.stride(1)
        a.cos, a.sin, a.table_rows = cos.data_ptr(), sin.data_ptr(), cos.shape[0]
        a.workspace, a.workspace_splits = self.ws.buf.data_ptr(), self.ws.max_splits
        a.batch, a.heads, a.kv_heads, a.head_dim = B, H, Hkv, d
        self.lib = _lib.load()

    def qkv_supported(self, layout: int) -> bool:
        """Does a plain step on this slab, laid out for ``layout`` rows, run the fused projection + attention launch?"""
        a = self.a
        return bool(self.lib.spatten_decode_qkv_supported(a.dtype, a.batch, a.heads, a.kv_heads, a.head_dim, int(layout)))

    def run(self, q, k_new, v_new, kv_len: int, pos_q: int, scores: torch.Tensor,
            position_ids: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
            step: Optional[StepState] = None, layout: int = 0, proj: Optional[tuple] = None,
            qkv: Optional[tuple] = None) -> torch.Tensor:
        """q [B,H,d], k_new / v_new [B,Hkv,d] (rows contiguous), scores [B,H,>=kv_len]; optional position_ids int64 [B]
        (device) and additive mask [B,kv_len] as in attn_decode; returns out [B, H*d].
        ``layout``: lay the split-N decomposition out for this length (>= kv_len) instead of kv_len.  ``step``: the
        device-resident length — ``kv_len`` is then the BOUND of the launch (= the layout) and ``pos_q`` is not read."""
        if qkv is not None:
            return self._run_qkv(qkv, kv_len, pos_q, scores, step, layout, proj)
        if kv_len > self.cap or max(kv_len, pos_q + 1) > self.table_rows or q.stride(2) != 1 or k_new.stride(2) != 1 \
                or v_new.stride() != k_new.stride() or scores.stride(2) != 1 or layout > self.cap:
            raise ValueError("decode step outside the slab / rotary table, or operands without contiguous rows")
        if step is not None and (scores.shape[2] < kv_len or position_ids is not None or mask is not None):
            raise ValueError("device-length step: the stash row must cover the bound; no mask / position tensor")
        if position_ids is not None and (position_ids.dtype != torch.int64 or not position_ids.is_cuda):
            raise TypeError("position_ids must be an int64 device tensor")
        if mask is not None and (mask.stride(-1) != 1 or mask.dtype != q.dtype or not mask.is_cuda):
            raise ValueError("mask must be a device tensor in the model dtype with contiguous rows")
        stream = _stream()
        if stream != self.stream:                       # the workspace belongs to a stream
            self.stream, self.ws = stream, _workspace(self.B, self.H, self.d, q.device, stream)
            self.a.workspace = self.ws.buf.data_ptr()
        _pin(self.ws)
        out = torch.empty(self.B, self.H * self.d, dtype=q.dtype, device=q.device)
        a = self.a
        a.q, a.q_sb, a.q_sh = q.data_ptr(), (self.H * self.d if self.B == 1 else q.stride(0)), q.stride(1)
        a.k_new, a.v_new, a.new_sb, a.new_sh = k_new.data_ptr(), v_new.data_ptr(), k_new.stride(0), k_new.stride(1)
        a.out, a.out_sb = out.data_ptr(), out.stride(0)
        a.scores, a.sc_sb, a.sc_sh = scores.data_ptr(), scores.stride(0), scores.stride(1)
        a.kv_len, a.pos_q = kv_len, pos_q
        a.kv_len_layout = layout
        a.step_state = None if step is None else step.data_ptr()
        if proj is not None:
            y = torch.empty(self.B, proj[0].shape[0], dtype=q.dtype, device=q.device)
            _fill_proj(a, (proj[0], proj[1], y), q, self.B, self.H * self.d)
        else:
            a.proj_weight = None
        if position_ids is not None:
            a.position_ids, a.pos_sb = position_ids.data_ptr(), position_ids.stride(0)
        else:
            a.position_ids = None
        if mask is not None:
            a.mask, a.mask_sb = mask.data_ptr(), mask.stride(0)
        else:
            a.mask = None
        a.qkv_x = None
        _lib.check(self.lib.spatten_attn_decode_args(ctypes.byref(a), stream), "spatten_attn_decode")
        return out if proj is None else (out, y)

    def _run_qkv(self, qkv, kv_len, pos_q, scores, step, layout, proj):
        """The step with its q / k / v projections INSIDE the launch (spatten_decode_args_t::qkv_*): ``qkv`` = (x [1, hidden]
        the layer's input row, weight [3*H*d, hidden] stacked q/k/v, bias or None)."""
        x, w, bias = qkv
        _dev(x, w, bias)
        K = w.shape[1]
        if x.numel() != K or x.stride(-1) != 1 or w.stride(1) != 1 or w.shape[0] != 3 * self.H * self.d or w.dtype != x.dtype \
                or scores.stride(2) != 1 or kv_len > self.cap or max(kv_len, pos_q + 1) > self.table_rows or layout > self.cap \
                or (bias is not None and (not bias.is_contiguous() or bias.numel() != w.shape[0] or bias.dtype != x.dtype)):
            raise ValueError("fused projection step: x [hidden], weight [3*H*d, hidden] of one dtype with contiguous rows")
        stream = _stream()
        if stream != self.stream:
            self.stream, self.ws = stream, _workspace(self.B, self.H, self.d, x.device, stream)
            self.a.workspace = self.ws.buf.data_ptr()
        _pin(self.ws)
        out = torch.empty(self.B, self.H * self.d, dtype=x.dtype, device=x.device)
        a = self.a
        a.q = a.k_new = a.v_new = None
        a.qkv_x, a.qkv_weight, a.qkv_w_sn, a.qkv_bias = x.data_ptr(), w.data_ptr(), w.stride(0), _ptr(bias)
        a.qkv_exchange, a.qkv_hidden = self.ws.xch.data_ptr(), K
        a.out, a.out_sb = out.data_ptr(), out.stride(0)
        a.scores, a.sc_sb, a.sc_sh = scores.data_ptr(), scores.stride(0), scores.stride(1)
        a.kv_len, a.pos_q = kv_len, pos_q
        a.kv_len_layout = layout
        a.step_state = None if step is None else step.data_ptr()
        a.position_ids = a.mask = None
        if proj is not None:
            y = torch.empty(self.B, proj[0].shape[0], dtype=x.dtype, device=x.device)
            _fill_proj(a, (proj[0], proj[1], y), out, self.B, self.H * self.d)
        else:
            a.proj_weight = None
        try:
            _lib.check(self.lib.spatten_attn_decode_args(ctypes.byref(a), stream), "spatten_attn_decode (fused projections)")
        finally:
            a.qkv_x = None
        return out if proj is None else (out, y)


def set_decode_team(threads: int) -> int:
    """Threads of the attention team of a single-row decode step: 512 (two waves per SIMD, the default) or 256 (the form the
    fused projection launch contains) — process-wide (include/spatten.h: spatten_decode_set_team).  Returns the previous value."""
    prev = _lib.load().spatten_decode_set_team(int(threads))
    if prev < 0:
        raise ValueError("decode team: 256 or 512 threads")
    return prev


def set_decode_gqa(mode: int) -> int:
    """Grouped-query decode steps on the matrix cores (include/spatten.h: spatten_decode_set_gqa): -1 = where it measured faster
    (default: long caches / large groups, e.g. 32 / 8 heads from ~3k rows), 0 = never (one workgroup column per query head), 1 = whenever the step is eligible.  Process-wide; returns the previous mode."""
    prev = _lib.load().spatten_decode_set_gqa(int(mode))
    if prev < 0:
        raise ValueError("decode gqa mode: -1, 0 or 1")
    return prev - 1


def decode_gqa_selected(dtype: torch.dtype, batch: int, heads: int, kv_heads: int, head_dim: int, kv_len_layout: int) -> bool:
    """Would a plain single-row step of this geometry run in the matrix-core grouped-query form under the current mode
    (spatten_decode_gqa_selected)?  `kv_len_layout`: the cache length of a static launch / the bound of the device-length form."""
    return bool(_lib.load().spatten_decode_gqa_selected(_DT.get(dtype, -1), int(batch), int(heads), int(kv_heads), int(head_dim), int(kv_len_layout)))


def gemv(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``torch.nn.functional.linear(x, weight, bias)`` for single-token rows: x [..., K] with few rows (a decode step),
    weight [N, K] (rows contiguous), as ONE weight-streaming launch per row (modify_llama.py:72-74, :163).
    Returns [..., N]."""
    _dev(x, weight, bias, out)
    N, K = weight.shape
    if x.shape[-1] != K or weight.stride(1) != 1 or x.stride(-1) != 1 or weight.dtype != x.dtype:
        raise ValueError("gemv: x [..., K] and weight [N, K] of one dtype with contiguous rows")
    x2 = x.reshape(-1, K)
    M = x2.shape[0]
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    y2 = out.view(-1, N)
    if bias is not None and (bias.dtype != x.dtype or not bias.is_contiguous() or bias.numel() != N):
        raise ValueError("gemv: bias must be a contiguous [N] tensor of the same dtype")
    rc = _lib.load().spatten_gemv(_dt(x), x2.data_ptr(), x2.stride(0), weight.data_ptr(), weight.stride(0), _ptr(bias),
                                  y2.data_ptr(), y2.stride(0), M, N, K, _stream())
    _lib.check(rc, "spatten_gemv")
    return out


def kv_append(k_new: torch.Tensor, v_new: torch.Tensor, k_cache: Optional[torch.Tensor], kr_cache: torch.Tensor,
              v_cache: torch.Tensor, row0: int, cos: torch.Tensor, sin: torch.Tensor):
    """Rows [row0, row0+n) of the slab planes from k_new / v_new [B,Hkv,n,d] (any strides, d contiguous): K un-rotated,
    its rotation at the slot positions into the shadow, V (modify_llama.py:95-104 without the cat)."""
    _dev(k_new, v_new, k_cache, kr_cache, v_cache, cos, sin)
    B, Hkv, n, d = k_new.shape
    if k_new.stride(3) != 1 or v_new.stride() != k_new.stride():
        raise ValueError("k_new / v_new need contiguous d and identical strides")
    if kr_cache.stride(3) != 1 or kr_cache.stride(2) != d or v_cache.stride() != kr_cache.stride() \
            or (k_cache is not None and k_cache.stride() != kr_cache.stride()):
        raise ValueError("cache planes need contiguous rows (pitch d) and identical strides")
    if row0 + n > kr_cache.shape[2] or row0 + n > cos.shape[0]:
        raise ValueError("append exceeds cache capacity or rotary table")
    rc = _lib.load().spatten_kv_append(_dt(k_new), k_new.data_ptr(), v_new.data_ptr(), k_new.stride(0), k_new.stride(1),
                                       k_new.stride(2), _ptr(k_cache), kr_cache.data_ptr(), v_cache.data_ptr(),
                                       kr_cache.stride(0), kr_cache.stride(1), cos.data_ptr(), sin.data_ptr(), cos.shape[0],
                                       B, Hkv, n, d, row0, _stream())
    _lib.check(rc, "spatten_kv_append")


def kv_append_step(k_new: torch.Tensor, v_new: torch.Tensor, k_cache: Optional[torch.Tensor], kr_cache: torch.Tensor,
                   v_cache: torch.Tensor, step: "StepState", planes: Optional["PQPlanes"] = None):
    """The append of one decode step in device-length form: row (state length) - 1 of the slab planes from k_new / v_new
    [B,Hkv,d], rotated with the state's staged row — and, with ``planes``, that row's progressive-quant nibbles and scale.
    No host length: capturable (the modes whose attention launch does not append: spatten_amd/extensions.py)."""
    _dev(k_new, v_new, k_cache, kr_cache, v_cache)
    B, Hkv, d = k_new.shape
    if k_new.stride(2) != 1 or v_new.stride() != k_new.stride():
        raise ValueError("k_new / v_new need contiguous d and identical strides")
    if kr_cache.stride(3) != 1 or kr_cache.stride(2) != d or v_cache.stride() != kr_cache.stride() \
            or (k_cache is not None and k_cache.stride() != kr_cache.stride()):
        raise ValueError("cache planes need contiguous rows (pitch d) and identical strides")
    cap = kr_cache.shape[2]
    m = l = sc = None
    psb = psh = ssb = ssh = 0
    if planes is not None:
        if planes.msb.shape[2] < cap or planes.msb.stride(3) != 1 or planes.msb.stride(2) != d // 2 \
                or planes.lsb.stride() != planes.msb.stride() or planes.scale.stride(2) != 1:
            raise ValueError("planes smaller than the cache or not row-contiguous")
        m, l, sc = planes.msb, planes.lsb, planes.scale
        psb, psh, ssb, ssh = m.stride(0), m.stride(1), sc.stride(0), sc.stride(1)
    rc = _lib.load().spatten_kv_append_step(_dt(k_new), k_new.data_ptr(), v_new.data_ptr(), k_new.stride(0), k_new.stride(1),
                                            _ptr(k_cache), kr_cache.data_ptr(), v_cache.data_ptr(), kr_cache.stride(0),
                                            kr_cache.stride(1), _ptr(m), _ptr(l), _ptr(sc), psb, psh, ssb, ssh, B, Hkv, d, cap,
                                            step.data_ptr(), _stream())
    _lib.check(rc, "spatten_kv_append_step")


_pf_ws = _LRU(4)


def _prefill_workspace(nbytes: int, device) -> torch.Tensor:
    key = (str(device), _stream())
    buf = _pf_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _pf_ws.put(key, torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device))
    return buf


def attn_prefill(q: torch.Tensor, kr_cache: torch.Tensor, v_cache: torch.Tensor, kv_len: int,
                 cos: torch.Tensor, sin: torch.Tensor, pos_q0: int, causal: bool = True,
                 position_ids: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, scores: Optional[torch.Tensor] = None,
                 col_importance: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None,
                 numerics: str = "auto") -> torch.Tensor:
    """Flash-style prefill (modify_llama.py:86-147 at q_len>1).  q [B,H,q,d] un-rotated (any strides with d
    contiguous); kr_cache = ROTATED shadow of the keys, v_cache values, both already holding the q new rows
    at [kv_len-q, kv_len); mask additive [B,q,kv_len]; position_ids int64 [B,q]; lse optional fp32 [B,H,q,2] output
    (row reference max, sum exp): the softmax statistics.  ``numerics="fast"``: fp32 logits without the reference's two
    16-bit roundings (include/spatten.h, SPATTEN_PREFILL_FAST_NUMERICS) where no by-product needs them; ``"reference"``: both
    roundings of every logit (modify_llama.py:111-113); ``"auto"`` (default, round 6): fast when NOTHING that is defined on the
    rounded logits is requested (no stash, column importance, row statistics or mask) — the roundings are then observable only
    through the output, which stays inside the stated tolerance (tests/util.py) — reference otherwise.
    Returns out [B, q, H*d]."""
    if numerics not in ("auto", "reference", "fast"):
        raise ValueError("numerics must be 'auto', 'reference' or 'fast'")
    if numerics == "auto":
        numerics = "fast" if (scores is None and col_importance is None and lse is None and mask is None) else "reference"
    _dev(q, kr_cache, v_cache, cos, sin, out, scores, col_importance, position_ids, mask, lse)
    if lse is not None and (lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() != q.shape[0] * q.shape[1] * q.shape[2] * 2):
        raise ValueError("lse must be a contiguous fp32 [B,H,q,2] tensor")
    lib = _lib.load()
    B, H, ql, d = q.shape
    Hkv = kr_cache.shape[1]
    if q.stride(3) != 1 or kr_cache.stride(3) != 1 or kr_cache.stride(2) != d or v_cache.stride() != kr_cache.stride():
        raise ValueError("q needs contiguous d; kr_cache/v_cache need contiguous rows (pitch d)")
    if max(kv_len, pos_q0 + ql) > cos.shape[0] and position_ids is None:
        raise ValueError("rotary table too short")
    if position_ids is not None and (position_ids.dtype != torch.int64 or position_ids.stride(-1) != 1):
        raise TypeError("position_ids must be int64 with contiguous rows")
    if mask is not None and (mask.stride(-1) != 1 or mask.dtype != q.dtype):
        raise ValueError("mask must be in the model dtype with contiguous rows")
    if out is None:
        out = torch.empty(B, ql, H * d, dtype=q.dtype, device=q.device)
    ws = _prefill_workspace(lib.spatten_prefill_workspace_bytes(_dt(q), B, H, Hkv, d, ql, kv_len), q.device)
    rc = lib.spatten_attn_prefill(
        _dt(q), q.data_ptr(), q.stride(0), q.stride(1), q.stride(2),
        kr_cache.data_ptr(), v_cache.data_ptr(), kr_cache.stride(0), kr_cache.stride(1),
        cos.data_ptr(), sin.data_ptr(), cos.shape[0],
        _ptr(position_ids), 0 if position_ids is None else (position_ids.stride(0) if position_ids.shape[0] > 1 else 0),
        _ptr(mask), *((0, 0) if mask is None else (mask.stride(0), mask.stride(1))),
        out.data_ptr(), out.stride(0), out.stride(1),
        _ptr(scores), *((0, 0, 0) if scores is None else (scores.stride(0), scores.stride(1), scores.stride(2))),
        _ptr(col_importance), _ptr(lse), ws.data_ptr(),
        B, H, Hkv, d, ql, kv_len, pos_q0, int(bool(causal)) | (2 if numerics == "fast" else 0), _stream())
    _lib.check(rc, "spatten_attn_prefill")
    return out


def attn_prefill_pq(q: torch.Tensor, planes: "PQPlanes", v_cache: torch.Tensor, kv_len: int, cos: torch.Tensor,
                    sin: torch.Tensor, pos_q0: int, threshold: float, causal: bool = True,
                    position_ids: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None, need_lsb: Optional[torch.Tensor] = None):
    """Prefill over progressively quantised keys (BASELINE.json configs[3]): MSB pass for every query row,
    ``need_lsb[b,h,i] = max_j prob_ij < threshold``, LSB refetch + one recompute of the flagged rows.  q [B,H,q,d]
    un-rotated; planes of the ROTATED keys rows [0, kv_len) (ops.pq_pack); bf16 / f16, d 64 / 128.
    Returns (out [B,q,H*d], need_lsb int32 [B,H,q])."""
    _dev(q, planes.msb, v_cache, cos, sin, out, need_lsb, position_ids, mask)
    lib = _lib.load()
    B, H, ql, d = q.shape
    Hkv = v_cache.shape[1]
    if q.stride(3) != 1 or v_cache.stride(3) != 1 or v_cache.stride(2) != d:
        raise ValueError("q needs contiguous d; v_cache needs contiguous rows (pitch d)")
    if max(kv_len, pos_q0 + ql) > cos.shape[0] and position_ids is None:
        raise ValueError("rotary table too short")
    if kv_len > planes.msb.shape[2]:
        raise ValueError("kv_len exceeds the planes")
    if position_ids is not None and (position_ids.dtype != torch.int64 or position_ids.stride(-1) != 1):
        raise TypeError("position_ids must be int64 with contiguous rows")
    if mask is not None and (mask.stride(-1) != 1 or mask.dtype != q.dtype):
        raise ValueError("mask must be in the model dtype with contiguous rows")
    if out is None:
        out = torch.empty(B, ql, H * d, dtype=q.dtype, device=q.device)
    if need_lsb is None:
        need_lsb = torch.empty(B, H, ql, dtype=torch.int32, device=q.device)
    if need_lsb.dtype != torch.int32 or not need_lsb.is_contiguous() or need_lsb.numel() != B * H * ql:
        raise ValueError("need_lsb must be a contiguous int32 [B,H,q] tensor")
    ws = _prefill_workspace(lib.spatten_prefill_pq_workspace_bytes(_dt(q), B, H, Hkv, d, ql, kv_len), q.device)
    rc = lib.spatten_attn_prefill_pq(
        _dt(q), q.data_ptr(), q.stride(0), q.stride(1), q.stride(2),
        planes.msb.data_ptr(), planes.lsb.data_ptr(), planes.scale.data_ptr(), planes.msb.stride(0), planes.msb.stride(1),
        planes.scale.stride(0), planes.scale.stride(1),
        v_cache.data_ptr(), v_cache.stride(0), v_cache.stride(1),
        cos.data_ptr(), sin.data_ptr(), cos.shape[0],
        _ptr(position_ids), 0 if position_ids is None else (position_ids.stride(0) if position_ids.shape[0] > 1 else 0),
        _ptr(mask), *((0, 0) if mask is None else (mask.stride(0), mask.stride(1))),
        out.data_ptr(), out.stride(0), out.stride(1), need_lsb.data_ptr(), float(threshold), ws.data_ptr(),
        B, H, Hkv, d, ql, kv_len, pos_q0, int(causal), _stream())
    _lib.check(rc, "spatten_attn_prefill_pq")
    return out, need_lsb


def build_shadow(k_cache: torch.Tensor, kr_cache: torch.Tensor, lo: int, hi: int, cos: torch.Tensor, sin: torch.Tensor):
    """Rotate rows [lo,hi) of the un-rotated cache at their slot index into the shadow (modify_llama.py:103-104)."""
    if hi > lo:
        rope_single(k_cache[:, :, lo:hi], cos, sin, pos0=lo, out=kr_cache[:, :, lo:hi])


def rope_single(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                position_ids: Optional[torch.Tensor] = None, pos0: int = 0,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """apply_rotary_pos_emb_single (modify_llama.py:21-28); x [B,H,n,d] (d contiguous, any other
    strides); cos/sin are the [rows, d/2] half tables; position_ids int64 [B,n] or [1,n] or None."""
    _dev(x, cos, sin, position_ids)
    lib = _lib.load()
    B, H, n, d = x.shape
    if x.stride(3) != 1:
        x = x.contiguous()
    y = torch.empty(B, H, n, d, dtype=x.dtype, device=x.device) if out is None else out
    if y.stride(3) != 1 or y.shape != x.shape:
        raise ValueError("rope_single output must match x and have contiguous d")
    pos_sb = 0
    if position_ids is not None:
        if position_ids.dim() == 1:
            position_ids = position_ids[None]
        position_ids = position_ids.to(torch.int64)
        if position_ids.stride(-1) != 1:
            position_ids = position_ids.contiguous()
        pos_sb = position_ids.stride(0) if position_ids.shape[0] > 1 else 0
    rc = lib.spatten_rope_single(_dt(x), x.data_ptr(), x.stride(0), x.stride(1), x.stride(2),
                                 y.data_ptr(), y.stride(0), y.stride(1), y.stride(2),
                                 cos.data_ptr(), sin.data_ptr(), cos.shape[0],
                                 _ptr(position_ids), pos_sb, pos0, B, H, n, d, _stream())
    _lib.check(rc, "spatten_rope_single")
    return y


def importance(stash: torch.Tensor) -> torch.Tensor:
    """stash [B,H,q,L] -> [H,L] = stash.sum(0).sum(1) (kv_cache_token_pruning.py:51)."""
    _dev(stash)
    B, H, Q, L = stash.shape
    if B == 1 and Q == 1:
        return stash[0, :, 0, :]          # the two sums are identities; a view, like the values torch returns
    lib = _lib.load()
    if stash.stride(3) != 1:
        stash = stash.contiguous()
    out = torch.empty(H, L, dtype=stash.dtype, device=stash.device)
    rc = lib.spatten_importance(_dt(stash), stash.data_ptr(), stash.stride(0), stash.stride(1), stash.stride(2),
                                out.data_ptr(), out.stride(0), B, H, Q, L, _stream())
    _lib.check(rc, "spatten_importance")
    return out


def topk_select(score: torch.Tensor, lo: int, hi: int, k: int) -> torch.Tensor:
    """score [H, L] -> int32 [H, k]: ascending absolute positions of the k largest of score[:, lo:hi]
    (kv_cache_token_pruning.py:59-63; ties at the k-th value: lowest index first)."""
    _dev(score)
    lib = _lib.load()
    H, L = score.shape
    hi = min(hi, L)
    if score.stride(1) != 1:
        score = score.contiguous()
    idx = torch.empty(H, max(k, 0), dtype=torch.int32, device=score.device)
    rc = lib.spatten_topk_select(_dt(score), score.data_ptr(), score.stride(0), H, lo, hi, k,
                                 idx.data_ptr(), idx.stride(0), _stream())
    _lib.check(rc, "spatten_topk_select")
    return idx


def kv_compact(K: torch.Tensor, V: Optional[torch.Tensor], idx: torch.Tensor, start: int, tail_lo: int,
               L: Optional[int] = None, capacity: Optional[int] = None,
               rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """Fused gather + concat (kv_cache_token_pruning.py:64-96).  K,V [B,H,>=L,d]; idx int32 [H,k].
    Returns (K', V', Kr') [B,H,L',d] views of freshly allocated [B,H,capacity,d] slabs; Kr' (the rotated
    shadow of K' at its new slot positions) only when ``rope=(cos, sin)`` half tables are given."""
    _dev(K, V, idx)
    lib = _lib.load()
    B, H, Lk, d = K.shape
    L = Lk if L is None else L
    k = idx.shape[1]
    tail_lo = min(max(tail_lo, 0), L)
    tail_len = L - tail_lo
    Lp = start + k + tail_len
    cap = max(capacity or Lp, Lp)
    if K.stride(3) != 1 or K.stride(2) != d or (V is not None and V.stride() != K.stride()):
        raise ValueError("K/V need contiguous rows and identical strides")
    Kd = torch.empty(B, H, cap, d, dtype=K.dtype, device=K.device)
    Vd = torch.empty_like(Kd) if V is not None else None
    Krd = torch.empty_like(Kd) if rope is not None else None
    cos, sin = rope if rope is not None else (None, None)
    rc = lib.spatten_kv_compact(_dt(K), K.data_ptr(), _ptr(V), K.stride(0), K.stride(1),
                                Kd.data_ptr(), _ptr(Vd), _ptr(Krd), Kd.stride(0), Kd.stride(1),
                                _ptr(cos), _ptr(sin), 0 if cos is None else cos.shape[0],
                                idx.data_ptr(), idx.stride(0), B, H, d, start, k, tail_lo, tail_len, _stream())
    _lib.check(rc, "spatten_kv_compact")
    return Kd[:, :, :Lp], (None if Vd is None else Vd[:, :, :Lp]), (None if Krd is None else Krd[:, :, :Lp])


class PrunePlan:
    """Device pointer tables for the batched all-layer prune (spatten_prune_layers)."""

    def __init__(self, scores: Sequence[torch.Tensor], Ks, Vs, Kd, Vd, Krd=None, acc_src=None, acc_dst=None):
        dev = Ks[0].device
        groups = [scores, Ks, Vs, Kd, Vd] + ([Krd] if Krd is not None else []) + ([acc_src, acc_dst] if acc_src is not None else [])
        # ONE host->device copy for all pointer tables
        flat = torch.tensor([t.data_ptr() for g in groups for t in g], dtype=torch.int64).to(dev)
        n = len(Ks)
        tabs = [flat[i * n:(i + 1) * n] for i in range(len(groups))]
        self.score_ptrs, self.ks, self.vs, self.kd, self.vd = tabs[:5]
        self.krd = tabs[5] if Krd is not None else None
        self.acc_src, self.acc_dst = (tabs[-2], tabs[-1]) if acc_src is not None else (None, None)
        self.keep = (flat, list(scores), list(Ks), list(Vs), list(Kd), list(Vd), None if Krd is None else list(Krd), acc_src, acc_dst)


def prune_layers(scores: Sequence[torch.Tensor], Ks: Sequence[torch.Tensor], Vs: Sequence[torch.Tensor],
                 L: int, lo: int, hi: int, k: int, capacity: Optional[int] = None,
                 dst: Optional[Tuple[List[torch.Tensor], List[torch.Tensor], Optional[List[torch.Tensor]]]] = None,
                 plan: Optional[PrunePlan] = None, idx: Optional[torch.Tensor] = None,
                 rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                 acc: Optional[Tuple[Sequence[torch.Tensor], Sequence[torch.Tensor]]] = None):
    """All layers of apply_token_pruning's loop (kv_cache_token_pruning.py:55-96) in two launches.
    scores[l] [H, >=hi] (the model dtype, or fp32 cascade accumulators); Ks[l]/Vs[l] [B,H,>=L,d].  With ``rope=(cos, sin)``
    the rotated shadow of every new cache is produced by the same pass.  ``acc=(src list, dst list)``: fp32 [H, >=L] /
    [H, >=L'] accumulators that follow the rows (a third all-layer launch).
    Returns (K' list, V' list, Kr' list or None, idx [layers,H,k])."""
    _dev(*scores, *Ks, *Vs)
    lib = _lib.load()
    nl = len(Ks)
    B, H, _, d = Ks[0].shape
    hi = min(hi, L)
    tail_lo = hi
    tail_len = L - tail_lo
    Lp = lo + k + tail_len
    cap = max(capacity or Lp, Lp)
    if dst is None:
        Kd = [torch.empty(B, H, cap, d, dtype=Ks[0].dtype, device=Ks[0].device) for _ in range(nl)]
        Vd = [torch.empty_like(x) for x in Kd]
        Krd = [torch.empty_like(x) for x in Kd] if rope is not None else None
    else:
        Kd, Vd, Krd = dst
    for t in list(Ks) + list(Vs) + Kd + Vd + (Krd or []):
        if t.stride(3) != 1 or t.stride(2) != d:
            raise ValueError("K/V need contiguous rows (pitch d)")
    if any(t.stride() != Ks[0].stride() for t in list(Ks) + list(Vs)) \
            or any(t.stride() != Kd[0].stride() for t in Kd + Vd + (Krd or [])):
        raise ValueError("all layers must share strides")
    if any(s.stride(1) != 1 or s.stride(0) != scores[0].stride(0) for s in scores):
        raise ValueError("scores need contiguous rows and a common head stride")
    if Krd is not None and rope is None:
        raise ValueError("a shadow destination needs the rotary tables")
    if acc is not None:
        a_src, a_dst = acc
        for t in list(a_src) + list(a_dst):
            if t.dtype != torch.float32 or t.stride(1) != 1 or t.shape[0] != H:
                raise ValueError("accumulators must be fp32 [H, len] with contiguous rows")
        if any(t.stride(0) != a_src[0].stride(0) or t.shape[1] < L for t in a_src) \
                or any(t.stride(0) != a_dst[0].stride(0) or t.shape[1] < Lp for t in a_dst):
            raise ValueError("accumulators need a common head stride and room for the rows")
    if plan is None:
        plan = PrunePlan(scores, Ks, Vs, Kd, Vd, Krd, *(acc if acc is not None else (None, None)))
    if idx is None:
        idx = torch.empty(nl, H, k, dtype=torch.int32, device=Ks[0].device)
    cos, sin = rope if rope is not None else (None, None)
    rc = lib.spatten_prune_layers_scored(
        _dt(scores[0]), _dt(Ks[0]), nl, plan.score_ptrs.data_ptr(), scores[0].stride(0),
        plan.ks.data_ptr(), plan.vs.data_ptr(), Ks[0].stride(0), Ks[0].stride(1),
        plan.kd.data_ptr(), plan.vd.data_ptr(), _ptr(plan.krd), Kd[0].stride(0), Kd[0].stride(1),
        _ptr(cos), _ptr(sin), 0 if cos is None else cos.shape[0], idx.data_ptr(),
        _ptr(plan.acc_src), 0 if acc is None else acc[0][0].stride(0), _ptr(plan.acc_dst), 0 if acc is None else acc[1][0].stride(0),
        B, H, d, lo, hi, k, tail_lo, tail_len, _stream())
    _lib.check(rc, "spatten_prune_layers")
    return ([x[:, :, :Lp] for x in Kd], [x[:, :, :Lp] for x in Vd],
            None if Krd is None else [x[:, :, :Lp] for x in Krd], idx)


class LayerCascadePlan:
    """Everything ``prune_layer_cascade`` prepares on the host — the per-layer table, the pointer rows, the output buffers — built
    once; ``run()`` issues the event (the C call alone) and returns the same tuple.  For a caller that repeats an event on the
    same tensors (bench.py times it like the plain event's ``PrunePlan``), and for issuing the event inside a stream capture
    (the table copy and the allocations happen here, outside).  Note: ``run()`` does not re-zero the new accumulators' tails."""

    def __init__(self, scores, known_ids, id_base, Ks, Vs, lens, his, keeps, start, capacities, rope, accs=None, dst=None):
        _dev(*scores, *Ks, *Vs)
        lib = _lib.load()
        nl = len(Ks)
        B, H, _, d = Ks[0].shape
        dev, dt = Ks[0].device, Ks[0].dtype
        cos, sin = rope
        new_lens = [start + keeps[l] + (lens[l] - his[l]) for l in range(nl)]
        kmax = max(keeps)
        if dst is not None:
            Kd, Vd, Krd = (list(x) for x in dst)
            for l in range(nl):
                sd = Kd[l].stride()
                if Kd[l].shape[2] < new_lens[l] or sd != Vd[l].stride() or sd != Krd[l].stride() or sd[3] != 1 or sd[2] != d:
                    raise ValueError("layer cascade: destination planes too small or not row-contiguous with equal strides")
        else:
            Kd = [torch.empty(B, H, max(capacities[l], new_lens[l]), d, dtype=dt, device=dev) for l in range(nl)]
            Vd = [torch.empty_like(x) for x in Kd]
            Krd = [torch.empty_like(x) for x in Kd]
        # one allocation per kind (32 layers x torch.empty / torch.zeros — 32 fill launches — were most of the event's host time)
        ids_all = torch.empty(nl, H, max(new_lens), dtype=torch.int32, device=dev)
        new_ids = [ids_all[l].narrow(1, 0, new_lens[l]) for l in range(nl)]
        new_accs = None
        if accs is not None:
            widths = [max(Kd[l].shape[2], accs[l].shape[1]) for l in range(nl)]
            acc_all = torch.zeros(nl, H, max(widths), dtype=torch.float32, device=dev)
            new_accs = [acc_all[l].narrow(1, 0, widths[l]) for l in range(nl)]
        idx = torch.empty(nl, H, kmax, dtype=torch.int32, device=dev)
        wmax = max(his[l] - start for l in range(nl))
        scratch = torch.empty(H, wmax, dtype=torch.int32, device=dev)
        sdt = scores[0].dtype
        rows = []
        for l in range(nl):
            sc = scores[l]
            ks, ss, ds = Ks[l].stride(), sc.stride(), Kd[l].stride()
            if ks[3] != 1 or ks[2] != d or Vs[l].stride() != ks or ss[1] != 1 or sc.dtype != sdt:
                raise ValueError("layer cascade: K / V need contiguous rows and equal strides, scores contiguous rows of one dtype")
            kn = known_ids[l]
            if kn is not None and (kn.dtype != torch.int32 or kn.stride(1) != 1 or kn.shape[0] != H):
                raise ValueError("known ids must be int32 [H, n] with contiguous rows")
            rows += [lens[l], his[l], keeps[l], new_lens[l],
                     ss[0], 0 if kn is None else kn.shape[1], 0 if kn is None else kn.stride(0), ids_all.stride(1),
                     ks[0], ks[1], ds[0], ds[1],
                     0 if accs is None else accs[l].stride(0), 0 if accs is None else acc_all.stride(1), int(id_base), 0]
        groups = [scores, known_ids, new_ids, Ks, Vs, Kd, Vd, Krd] + ([list(accs), new_accs] if accs is not None else [])
        # the per-layer table (16 int64 per layer: LayerPrune of layer_cascade.hip) and the pointer rows behind it: ONE host tensor,
        # one copy to the device
        host = torch.tensor(rows + [0 if t is None else t.data_ptr() for g in groups for t in g], dtype=torch.int64).pin_memory()
        both = host.to(dev, non_blocking=True)      # (pinned: the copy does not block the host behind the stream's earlier work)
        ptr = lambda i: both.data_ptr() + (nl * 16 + i * nl) * 8
        args = (_dt(scores[0]), _dt(Ks[0]), nl, both.data_ptr(), host.data_ptr(), ptr(0), ptr(1), ptr(2), ptr(3), ptr(4), ptr(5), ptr(6),
                ptr(7), cos.data_ptr(), sin.data_ptr(), cos.shape[0], idx.data_ptr(), kmax, scratch.data_ptr(), scratch.stride(0),
                ptr(8) if accs is not None else None, ptr(9) if accs is not None else None, B, H, d, start)
        self._lib, self._args = lib, args
        self._keep = (both, host, scratch, ids_all, idx, list(scores), list(known_ids), list(Ks), list(Vs), Kd, Vd, Krd, cos, sin,
                      None if accs is None else list(accs), new_accs)
        self.result = ([Kd[l].narrow(2, 0, new_lens[l]) for l in range(nl)], [Vd[l].narrow(2, 0, new_lens[l]) for l in range(nl)],
                       [Krd[l].narrow(2, 0, new_lens[l]) for l in range(nl)], [idx[l].narrow(1, 0, keeps[l]) for l in range(nl)],
                       new_ids, new_accs)

    def run(self):
        rc = self._lib.spatten_prune_layer_cascade(*self._args, _stream())
        _lib.check(rc, "spatten_prune_layer_cascade")
        return self.result


def prune_layer_cascade(scores: Sequence[torch.Tensor], known_ids: Sequence[Optional[torch.Tensor]], id_base: int,
                        Ks: Sequence[torch.Tensor], Vs: Sequence[torch.Tensor], lens: Sequence[int], his: Sequence[int],
                        keeps: Sequence[int], start: int, capacities: Sequence[int],
                        rope: Tuple[torch.Tensor, torch.Tensor],
                        accs: Optional[Sequence[torch.Tensor]] = None, dst: Optional[tuple] = None):
    """The layer-to-layer cascade's prune event for all layers in three launches (include/spatten.h:
    spatten_prune_layer_cascade).  ``dst`` = (Kd, Vd, Krd) lists of pre-allocated destination planes [B, H, >= new_len_l, d]
    (default: allocated here).  scores[l] [H, >= len_l] (one dtype; rows contiguous); known_ids[l] int32 [H, n_known_l] or
    None; Ks[l] / Vs[l] [B, H, >= len_l, d] (rows contiguous, K and V of a layer with equal strides); his[l] = window end;
    keeps[l] = tokens kept in the window (non-increasing); accs[l] fp32 [H, >= len_l] cascade accumulators (optional).
    Returns (K' list, V' list, Kr' list, idx list [H, k_l], new_ids list [H, new_len_l], new accs or None)."""
    return LayerCascadePlan(scores, known_ids, id_base, Ks, Vs, lens, his, keeps, start, capacities, rope, accs, dst).run()


# ------------------------------------------------------------------------------------------------
# SpAtten semantics beyond the reference's Python (parity unpinned; oracle/spatten_oracle.py restates them)
# ------------------------------------------------------------------------------------------------
def row_lse(stash: torch.Tensor, mask: Optional[torch.Tensor] = None, causal: bool = False) -> torch.Tensor:
    """stash [B,H,q,L] (+ additive mask [B,q,L]) -> lse [B,H,q,2] fp32 = (row max, sum exp)."""
    _dev(stash, mask)
    B, H, Q, L = stash.shape
    if stash.stride(3) != 1:
        stash = stash.contiguous()
    lse = torch.empty(B, H, Q, 2, dtype=torch.float32, device=stash.device)
    rc = _lib.load().spatten_row_lse(_dt(stash), stash.data_ptr(), stash.stride(0), stash.stride(1), stash.stride(2),
                                     _ptr(mask), *((0, 0) if mask is None else (mask.stride(0), mask.stride(1))),
                                     lse.data_ptr(), B, H, Q, L, int(causal), _stream())
    _lib.check(rc, "spatten_row_lse")
    return lse


def importance_accumulate(acc: torch.Tensor, stash: torch.Tensor, lse: Optional[torch.Tensor] = None,
                          mask: Optional[torch.Tensor] = None, causal: bool = False) -> torch.Tensor:
    """Cascade importance: acc[h, j] += sum_{b,i} softmax(stash[b,h,i,:] + mask[b,i,:])[j]  (in place).
    acc [H, >=L] fp32; stash [B,H,q,L]; lse [B,H,q,2] from attn_decode / row_lse (computed when None)."""
    _dev(acc, stash, lse, mask)
    B, H, Q, L = stash.shape
    if stash.stride(3) != 1:
        stash = stash.contiguous()
    if acc.dtype != torch.float32 or acc.stride(1) != 1 or acc.shape[0] != H or acc.shape[1] < L:
        raise ValueError("acc must be fp32 [H, >=L] with contiguous rows")
    if lse is None:
        lse = row_lse(stash, mask, causal)
    rc = _lib.load().spatten_importance_accumulate(
        _dt(stash), stash.data_ptr(), stash.stride(0), stash.stride(1), stash.stride(2), lse.data_ptr(),
        _ptr(mask), *((0, 0) if mask is None else (mask.stride(0), mask.stride(1))),
        acc.data_ptr(), acc.stride(0), B, H, Q, L, int(causal), _stream())
    _lib.check(rc, "spatten_importance_accumulate")
    return acc


def importance_accumulate_prefill(acc: torch.Tensor, q: torch.Tensor, kr_cache: torch.Tensor, kv_len: int,
                                  cos: torch.Tensor, sin: torch.Tensor, pos_q0: int, lse: torch.Tensor,
                                  causal: bool = True, position_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Cascade importance of a multi-token forward without the stash: acc[h, j] += sum over batch and query rows of the
    softmax probability of key j, recomputed from q [B,H,q,d] (un-rotated), the rotated shadow and the row statistics
    ``lse`` [B,H,q,2] that ``attn_prefill(..., lse=lse)`` wrote for the same inputs.  bf16 / f16, d 64 / 128."""
    _dev(acc, q, kr_cache, cos, sin, lse, position_ids)
    lib = _lib.load()
    B, H, ql, d = q.shape
    Hkv = kr_cache.shape[1]
    if acc.dtype != torch.float32 or acc.stride(1) != 1 or acc.shape[0] != H or acc.shape[1] < kv_len:
        raise ValueError("acc must be fp32 [H, >=kv_len] with contiguous rows")
    if q.stride(3) != 1 or kr_cache.stride(3) != 1 or kr_cache.stride(2) != d:
        raise ValueError("q needs contiguous d; kr_cache needs contiguous rows (pitch d)")
    ws = _prefill_workspace(lib.spatten_importance_prefill_workspace_bytes(B, H, d, ql), q.device)
    rc = lib.spatten_importance_accumulate_prefill(
        _dt(q), q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), kr_cache.data_ptr(), kr_cache.stride(0), kr_cache.stride(1),
        cos.data_ptr(), sin.data_ptr(), cos.shape[0],
        _ptr(position_ids), 0 if position_ids is None else (position_ids.stride(0) if position_ids.shape[0] > 1 else 0),
        lse.data_ptr(), acc.data_ptr(), acc.stride(0), ws.data_ptr(), B, H, Hkv, d, ql, kv_len, pos_q0, int(causal), _stream())
    _lib.check(rc, "spatten_importance_accumulate_prefill")
    return acc


def importance_compact(acc: torch.Tensor, idx: torch.Tensor, start: int, tail_lo: int, L: int,
                       capacity: Optional[int] = None) -> torch.Tensor:
    """The accumulator follows the cache through a prune (same row map as kv_compact)."""
    _dev(acc, idx)
    H = acc.shape[0]
    k = idx.shape[1]
    tail_lo = min(max(tail_lo, 0), L)
    tail_len = L - tail_lo
    Lp = start + k + tail_len
    dst = torch.zeros(H, max(capacity or Lp, Lp), dtype=torch.float32, device=acc.device)
    rc = _lib.load().spatten_importance_compact(acc.data_ptr(), acc.stride(0), dst.data_ptr(), dst.stride(0),
                                                idx.data_ptr(), idx.stride(0), H, start, k, tail_lo, tail_len, _stream())
    _lib.check(rc, "spatten_importance_compact")
    return dst


def cascade_rank(score: torch.Tensor, ids: torch.Tensor, prev_ids: torch.Tensor) -> torch.Tensor:
    """rank [H, L] fp32 = score where the slot's token id is among ``prev_ids`` [H, n_prev] (ascending), else -inf —
    the layer-to-layer cascade's candidate filter (include/spatten.h: spatten_cascade_rank)."""
    _dev(score, ids, prev_ids)
    H, L = score.shape
    if score.stride(1) != 1 or ids.stride(1) != 1 or prev_ids.stride(1) != 1 or ids.dtype != torch.int32 or prev_ids.dtype != torch.int32:
        raise ValueError("score / ids / prev_ids need contiguous rows; ids are int32")
    rank = torch.empty(H, L, dtype=torch.float32, device=score.device)
    rc = _lib.load().spatten_cascade_rank(_dt(score), score.data_ptr(), score.stride(0), ids.data_ptr(), ids.stride(0),
                                          prev_ids.data_ptr(), prev_ids.stride(0), prev_ids.shape[1], rank.data_ptr(),
                                          rank.stride(0), H, L, _stream())
    _lib.check(rc, "spatten_cascade_rank")
    return rank


def gather_rows_i32(src: torch.Tensor, idx: torch.Tensor, start: int, tail_lo: int, L: int) -> torch.Tensor:
    """int32 [H, L] -> [H, L'] with the prune's row map (start | idx | tail): the token ids follow the cache."""
    H = src.shape[0]
    k = idx.shape[1]
    tail_lo = min(max(tail_lo, 0), L)
    Lp = start + k + (L - tail_lo)
    dst = torch.empty(H, Lp, dtype=torch.int32, device=src.device)
    if src.stride(1) != 1:
        src = src.contiguous()
    # a pure 4-byte row gather: the fp32 accumulator's compaction kernel moves the bits unchanged
    rc = _lib.load().spatten_importance_compact(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0),
                                                idx.data_ptr(), idx.stride(0), H, start, k, tail_lo, L - tail_lo, _stream())
    _lib.check(rc, "spatten_importance_compact")
    return dst


def head_scores(out: torch.Tensor, heads: int, scores: Optional[torch.Tensor] = None) -> torch.Tensor:
    """scores[h] += sum |out[b, i, h*d:(h+1)*d]|; out [B,q,H*d] (or [B,H*d]); scores fp32 [H] (zeros when None)."""
    _dev(out, scores)
    if out.dim() == 2:
        out = out[:, None, :]
    B, Q, HD = out.shape
    if out.stride(2) != 1:
        out = out.contiguous()
    if scores is None:
        scores = torch.zeros(heads, dtype=torch.float32, device=out.device)
    rc = _lib.load().spatten_head_scores(_dt(out), out.data_ptr(), out.stride(0), out.stride(1), scores.data_ptr(),
                                         B, Q, heads, HD // heads, _stream())
    _lib.check(rc, "spatten_head_scores")
    return scores


_pv_ws_cache = _LRU(16)


def _pv_workspace(batch, heads, head_dim, device) -> torch.Tensor:
    key = (batch, heads, head_dim, device, _stream())
    ws = _pv_ws_cache.get(key)
    if ws is None:
        n = _lib.load().spatten_pv_gather_workspace_bytes(batch, heads, head_dim)
        ws = _pv_ws_cache.put(key, torch.zeros(n, dtype=torch.uint8, device=device))
    return ws


def pv_gather(stash: torch.Tensor, lse: torch.Tensor, v_cache: torch.Tensor, idx: torch.Tensor,
              mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Local V pruning, pass 2: out[b,h] = sum_{j in idx[b*H+h]} softmax_full(stash[b,h] + mask[b])[j] * V[b,hkv,j].
    stash [B,H,>=L]; lse [B,H,2]; v_cache [B,Hkv,cap,d]; idx int32 [B*H, k].  ``workspace``: zero-filled uint8 tensor
    of spatten_pv_gather_workspace_bytes (default: one cached per (shape, stream))."""
    _dev(stash, lse, v_cache, idx, mask, out)
    B, H = stash.shape[0], stash.shape[1]
    Hkv, d = v_cache.shape[1], v_cache.shape[3]
    if out is None:
        out = torch.empty(B, H * d, dtype=v_cache.dtype, device=v_cache.device)
    ws = workspace if workspace is not None else _pv_workspace(B, H, d, v_cache.device)
    rc = _lib.load().spatten_pv_gather(_dt(v_cache), stash.data_ptr(), stash.stride(0), stash.stride(1), lse.data_ptr(),
                                       _ptr(mask), 0 if mask is None else mask.stride(0), v_cache.data_ptr(),
                                       v_cache.stride(0), v_cache.stride(1), idx.data_ptr(), idx.stride(0), idx.shape[1],
                                       out.data_ptr(), out.stride(0), B, H, Hkv, d, ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "spatten_pv_gather")
    return out


_lv_ws_cache = _LRU(16)


def _local_v_workspace(batch, heads, device) -> torch.Tensor:
    key = (batch, heads, device, _stream())
    ws = _lv_ws_cache.get(key)
    if ws is None:
        n = _lib.load().spatten_local_v_workspace_bytes(batch, heads)
        ws = _lv_ws_cache.put(key, torch.zeros(n, dtype=torch.uint8, device=device))
    return _pin(ws)


def attn_decode_local_v(q: torch.Tensor, kr_cache: torch.Tensor, v_cache: torch.Tensor, kv_len: int, cos: torch.Tensor,
                        sin: torch.Tensor, pos_q: int, keep: int, scores: torch.Tensor, out: Optional[torch.Tensor] = None,
                        lse: Optional[torch.Tensor] = None, keep_fraction: float = 0.0, step: Optional["StepState"] = None,
                        layout: int = 0, workspace: Optional[torch.Tensor] = None, k_new: Optional[torch.Tensor] = None,
                        v_new: Optional[torch.Tensor] = None, k_cache: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The decode step with local V pruning as ONE launch (spatten_attn_decode_local_v): stash + (max, sum), exact per-head
    top-``keep`` of the logits, P.V over the kept V rows with the full denominator.  q [B,H,d]; kr_cache / v_cache
    [B,Hkv,cap,d]; scores [B,H,>=kv_len] (written).  With ``step`` the kept count is ceil(keep_fraction * device length).
    With ``k_new`` / ``v_new`` [B,Hkv,d] the launch also APPENDS the step's row kv_len - 1 (round 5,
    spatten_attn_decode_local_v_append; ``k_cache`` = the optional un-rotated plane): kv_len counts that row."""
    _dev(q, kr_cache, v_cache, cos, sin, scores, out, lse, k_new, v_new, k_cache)
    B, H, d = q.shape
    Hkv, cap = v_cache.shape[1], v_cache.shape[2]
    if q.stride(2) != 1 or v_cache.stride(3) != 1 or v_cache.stride(2) != d or kr_cache.stride() != v_cache.stride() \
            or scores.stride(2) != 1 or scores.shape[2] < kv_len:
        raise ValueError("q / caches / scores need contiguous rows; scores [B,H,>=kv_len]")
    if kv_len > cap or (step is None and max(kv_len, pos_q + 1) > cos.shape[0]) or cos.shape[1] * 2 != d or layout > cap:
        raise ValueError("kv_len exceeds cache capacity or rotary table")
    if out is None:
        out = torch.empty(B, H * d, dtype=q.dtype, device=q.device)
    ws = workspace if workspace is not None else _local_v_workspace(B, H, q.device)
    if (k_new is None) != (v_new is None):
        raise ValueError("k_new and v_new go together")
    if k_new is not None:
        if k_new.shape != (B, Hkv, d) or k_new.stride(2) != 1 or v_new.stride() != k_new.stride() \
                or (k_cache is not None and k_cache.stride() != v_cache.stride()):
            raise ValueError("k_new / v_new [B,Hkv,d] with contiguous d and identical strides; k_cache like v_cache")
        rc = _lib.load().spatten_attn_decode_local_v_append(
            _dt(q), q.data_ptr(), (H * d if B == 1 else q.stride(0)), q.stride(1), k_new.data_ptr(), v_new.data_ptr(),
            k_new.stride(0), k_new.stride(1), _ptr(k_cache), kr_cache.data_ptr(), v_cache.data_ptr(), v_cache.stride(0),
            v_cache.stride(1), cos.data_ptr(), sin.data_ptr(), cos.shape[0], int(pos_q), out.data_ptr(), out.stride(0),
            scores.data_ptr(), scores.stride(0), scores.stride(1), _ptr(lse), ws.data_ptr(), B, H, Hkv, d, int(kv_len), int(keep),
            float(keep_fraction), int(layout), None if step is None else step.data_ptr(), _stream())
        if rc == -2:
            raise NotImplementedError("spatten_attn_decode_local_v_append: split longer than 16384 rows")
        _lib.check(rc, "spatten_attn_decode_local_v_append")
        return out
    rc = _lib.load().spatten_attn_decode_local_v(
        _dt(q), q.data_ptr(), (H * d if B == 1 else q.stride(0)), q.stride(1), kr_cache.data_ptr(), v_cache.data_ptr(),
        v_cache.stride(0), v_cache.stride(1), cos.data_ptr(), sin.data_ptr(), cos.shape[0], int(pos_q), out.data_ptr(),
        out.stride(0), scores.data_ptr(), scores.stride(0), scores.stride(1), _ptr(lse), ws.data_ptr(), B, H, Hkv, d,
        int(kv_len), int(keep), float(keep_fraction), int(layout), None if step is None else step.data_ptr(), _stream())
    if rc == -2:
        raise NotImplementedError("spatten_attn_decode_local_v: split longer than 16384 rows — use cascade.local_v_decode's "
                                  "three-launch form")
    _lib.check(rc, "spatten_attn_decode_local_v")
    return out


class PQPlanes:
    """Progressive-quantisation planes of a rotated key cache: msb / lsb [B,Hkv,cap,d/2] uint8 (two 4-bit fields
    per byte), scale [B,Hkv,cap] fp32;  q8 = msb*16 + lsb,  x ~ q8 * scale."""

    def __init__(self, batch, kv_heads, cap, d, device):
        self.msb = torch.zeros(batch, kv_heads, cap, d // 2, dtype=torch.uint8, device=device)
        self.lsb = torch.zeros_like(self.msb)
        self.scale = torch.ones(batch, kv_heads, cap, dtype=torch.float32, device=device)

    def unpack(self, n: int):
        """(msb int8 [B,H,n,d], lsb uint8 [B,H,n,d], scale [B,H,n,1]) on the host — for tests."""
        m, l = self.msb[:, :, :n].cpu().numpy(), self.lsb[:, :, :n].cpu().numpy()
        import numpy as np
        mm = np.stack([m & 15, m >> 4], axis=-1).reshape(*m.shape[:-1], -1).astype(np.int8)
        mm = np.where(mm > 7, mm - 16, mm).astype(np.int8)
        ll = np.stack([l & 15, l >> 4], axis=-1).reshape(*l.shape[:-1], -1).astype(np.uint8)
        return mm, ll, self.scale[:, :, :n, None].cpu().numpy()


PQ_PROFILES = ((4, 8), (8, 8), (6, 6))      # (key MSB bits, value bits) of the profiled planes (include/spatten.h, ABI 4)


class PQProfilePlanes:
    """Profiled progressive-quantisation planes (include/spatten.h "Bit profiles and the quantised VALUE plane"): key MSB
    plane of ``key_bits`` (4 / 6 / 8) bits + 4-bit LSB plane + per-row scale, value plane of ``value_bits`` (8 / 6) bits +
    per-row scale, and the fp32 MSB-logit scratch [B, H, cap] the refetch pass reads.  Layouts are the library's (piece
    order); ``unpack`` undoes them for tests."""

    def __init__(self, batch, kv_heads, heads, cap, d, device, key_bits: int = 4, value_bits: int = 8):
        if (key_bits, value_bits) not in PQ_PROFILES:
            raise ValueError(f"unsupported bit profile {(key_bits, value_bits)}: one of {PQ_PROFILES}")
        self.key_bits, self.value_bits, self.d = int(key_bits), int(value_bits), d
        u8 = dict(dtype=torch.uint8, device=device)
        self.msb = torch.zeros(batch, kv_heads, cap, d * key_bits // 8, **u8)
        self.lsb = torch.zeros(batch, kv_heads, cap, d // 2, **u8)
        self.scale = torch.ones(batch, kv_heads, cap, dtype=torch.float32, device=device)
        self.vq = torch.zeros(batch, kv_heads, cap, d * value_bits // 8, **u8)
        self.vscale = torch.ones(batch, kv_heads, cap, dtype=torch.float32, device=device)
        self.msb_logit = torch.zeros(batch, heads, cap, dtype=torch.float32, device=device)
        a = self.desc = _lib.PQPlanesDesc()
        a.struct_size = ctypes.sizeof(_lib.PQPlanesDesc)
        a.key_msb_bits, a.value_bits = self.key_bits, self.value_bits
        a.key_msb, a.key_lsb, a.key_scale = self.msb.data_ptr(), self.lsb.data_ptr(), self.scale.data_ptr()
        a.val_q, a.val_scale, a.msb_logit = self.vq.data_ptr(), self.vscale.data_ptr(), self.msb_logit.data_ptr()
        a.km_sb, a.km_sh = self.msb.stride(0), self.msb.stride(1)
        a.kl_sb, a.kl_sh = self.lsb.stride(0), self.lsb.stride(1)
        a.vq_sb, a.vq_sh = self.vq.stride(0), self.vq.stride(1)
        a.sc_sb, a.sc_sh = self.scale.stride(0), self.scale.stride(1)
        a.lg_sb, a.lg_sh = self.msb_logit.stride(0), self.msb_logit.stride(1)

    @property
    def capacity(self) -> int:
        return self.msb.shape[2]

    @staticmethod
    def _fields(plane, bits: int, d: int):
        """[..., d*bits/8] bytes in piece order -> unsigned fields [..., d] in ELEMENT order (numpy)."""
        import numpy as np
        x = plane.cpu().numpy()
        lpr = d // 16
        pb = 2 * bits                                   # bytes per piece
        bitsarr = np.unpackbits(x.reshape(*x.shape[:-1], lpr, pb), axis=-1, bitorder="little")       # [..., lpr, 8 pb]
        f = (bitsarr.reshape(*bitsarr.shape[:-1], 16, bits).astype(np.int64) << np.arange(bits)).sum(-1)   # [..., lpr, 16]
        out = np.zeros((*f.shape[:-2], d), dtype=np.int64)
        for c in range(lpr):
            out[..., 8 * c:8 * c + 8] = f[..., c, :8]
            out[..., d // 2 + 8 * c:d // 2 + 8 * c + 8] = f[..., c, 8:]
        return out

    def unpack(self, n: int):
        """(msb int [B,Hkv,n,d] signed, lsb [B,Hkv,n,d] in [0,15], scale [B,Hkv,n,1], qv int [B,Hkv,n,d] signed,
        vscale [B,Hkv,n,1]) on the host — for tests."""
        import numpy as np
        d, kb, vb = self.d, self.key_bits, self.value_bits
        mm = self._fields(self.msb[:, :, :n], kb, d) - (1 << (kb - 1))
        ll = self._fields(self.lsb[:, :, :n], 4, d)
        qv = self._fields(self.vq[:, :, :n], vb, d) - (1 << (vb - 1))
        return (mm, ll, self.scale[:, :, :n, None].cpu().numpy(), qv, self.vscale[:, :, :n, None].cpu().numpy())


def pq_pack_planes(kr_cache: torch.Tensor, v_cache: torch.Tensor, planes: PQProfilePlanes, lo: int, hi: int,
                   step: Optional["StepState"] = None):
    """Quantise rows [lo, hi) of the rotated shadow into the key planes and of the values into the value plane; with
    ``step`` the one row (state length) - 1 (device-length form: capturable), ``hi`` = the planes' capacity."""
    _dev(kr_cache, v_cache, planes.msb)
    B, Hkv, cap, d = kr_cache.shape
    if v_cache.stride() != kr_cache.stride() or kr_cache.stride(3) != 1 or kr_cache.stride(2) != d:
        raise ValueError("kr_cache / v_cache need contiguous rows (pitch d) and identical strides")
    if hi > planes.capacity or hi > cap:
        raise ValueError("rows beyond the planes' capacity")
    rc = _lib.load().spatten_pq_pack_planes(_dt(kr_cache), kr_cache.data_ptr(), v_cache.data_ptr(), kr_cache.stride(0),
                                            kr_cache.stride(1), ctypes.byref(planes.desc), B, Hkv, d, int(lo), int(hi),
                                            None if step is None else step.data_ptr(), _stream())
    _lib.check(rc, "spatten_pq_pack_planes")


def kv_append_planes(k_new: torch.Tensor, v_new: torch.Tensor, k_cache: Optional[torch.Tensor], kr_cache: torch.Tensor,
                     v_cache: torch.Tensor, planes: PQProfilePlanes, row: int, cos: torch.Tensor, sin: torch.Tensor,
                     step: Optional["StepState"] = None):
    """One decode step's append AND its rows of the profiled planes in one launch (spatten_kv_append_planes): row ``row`` of the
    slab planes from k_new / v_new [B,Hkv,d] — what ``kv_append`` + ``pq_pack_planes(row, row + 1)`` leave, bit for bit; with
    ``step`` the row (state length) - 1, rotated with the state's staged row (capturable; ``row`` / ``cos`` / ``sin`` unused)."""
    _dev(k_new, v_new, k_cache, kr_cache, v_cache, planes.msb)
    B, Hkv, d = k_new.shape
    if k_new.stride(2) != 1 or v_new.stride() != k_new.stride():
        raise ValueError("k_new / v_new need contiguous d and identical strides")
    if kr_cache.stride(3) != 1 or kr_cache.stride(2) != d or v_cache.stride() != kr_cache.stride() \
            or (k_cache is not None and k_cache.stride() != kr_cache.stride()):
        raise ValueError("cache planes need contiguous rows (pitch d) and identical strides")
    cap = min(kr_cache.shape[2], planes.capacity)
    if step is None and not (0 <= row < cap and row < cos.shape[0]):
        raise ValueError("append exceeds cache capacity, the planes or the rotary table")
    rc = _lib.load().spatten_kv_append_planes(_dt(k_new), k_new.data_ptr(), v_new.data_ptr(), k_new.stride(0), k_new.stride(1),
                                              _ptr(k_cache), kr_cache.data_ptr(), v_cache.data_ptr(), kr_cache.stride(0),
                                              kr_cache.stride(1), ctypes.byref(planes.desc), cos.data_ptr(), sin.data_ptr(),
                                              cos.shape[0], B, Hkv, d, int(row), int(cap),
                                              None if step is None else step.data_ptr(), _stream())
    _lib.check(rc, "spatten_kv_append_planes")


def attn_decode_pqv(q: torch.Tensor, planes: PQProfilePlanes, kv_len: int, cos: torch.Tensor, sin: torch.Tensor, pos_q: int,
                    threshold: float, out: Optional[torch.Tensor] = None, need_lsb: Optional[torch.Tensor] = None,
                    scores: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None,
                    head_ids: Optional[torch.Tensor] = None, head_abs: Optional[torch.Tensor] = None,
                    step: Optional["StepState"] = None, layout: int = 0, n_splits: int = 0, msb_only: bool = False,
                    workspace: Optional[DecodeWorkspace] = None, append=None) -> torch.Tensor:
    """Decode over the profiled planes (spatten_attn_decode_pq): MSB pass + LSB-only refetch for the flagged heads, P.V over
    the quantised values.  q [B,H,d]; returns out [B, H*d]; ``need_lsb`` int32 [B*H] is written.
    ``append`` = (k_new, v_new, k_cache | None, kr_cache, v_cache): the step's append INSIDE the MSB pass (round 5) — row kv_len - 1
    of the cache planes and of every quantised plane, for the launched heads (what ``kv_append_planes`` leaves)."""
    _dev(q, planes.msb, cos, sin, out, need_lsb, scores, lse, head_ids, head_abs)
    B, H, d = q.shape
    Hkv = planes.msb.shape[1]
    if d != planes.d or q.stride(2) != 1 or cos.shape[1] * 2 != d:
        raise ValueError("q [B,H,d] with contiguous rows; planes and rotary tables of the same head_dim")
    if kv_len > planes.capacity or (step is None and max(kv_len, pos_q + 1) > cos.shape[0]) or layout > planes.capacity:
        raise ValueError("kv_len exceeds the planes' capacity or the rotary table")
    if planes.msb_logit.shape[1] != H:
        raise ValueError("planes were allocated for another number of query heads")
    if out is None:
        out = torch.empty(B, H * d, dtype=q.dtype, device=q.device)
    if need_lsb is None:
        need_lsb = torch.empty(B * H, dtype=torch.int32, device=q.device)
    stream = _stream()
    ws = _pin(workspace) if workspace is not None else _workspace(B, H, d, q.device, stream)
    a = _lib.PQDecodeArgs()
    a.struct_size = ctypes.sizeof(_lib.PQDecodeArgs)
    a.dtype = _dt(q)
    a.q, a.q_sb, a.q_sh = q.data_ptr(), (H * d if B == 1 else q.stride(0)), q.stride(1)
    a.planes = ctypes.pointer(planes.desc)
    a.cos, a.sin, a.table_rows, a.pos_q = cos.data_ptr(), sin.data_ptr(), cos.shape[0], int(pos_q)
    a.out, a.out_sb = out.data_ptr(), out.stride(0)
    if scores is not None:
        if scores.stride(2) != 1 or scores.shape[2] < kv_len:
            raise ValueError("scores [B,H,>=kv_len] with contiguous rows")
        a.scores, a.sc_sb, a.sc_sh = scores.data_ptr(), scores.stride(0), scores.stride(1)
    a.lse = _ptr(lse)
    a.need_lsb, a.threshold, a.flags = need_lsb.data_ptr(), float(threshold), 1 if msb_only else 0
    a.workspace, a.workspace_splits = ws.buf.data_ptr(), ws.max_splits
    a.batch, a.heads, a.kv_heads, a.head_dim, a.kv_len = B, H, Hkv, d, int(kv_len)
    a.n_splits, a.kv_len_layout = int(n_splits), int(layout)
    if head_ids is not None:
        if head_ids.dtype != torch.int32 or head_ids.dim() != 1:
            raise TypeError("head_ids must be a 1-D int32 device tensor")
        a.head_ids, a.n_active_heads = head_ids.data_ptr(), head_ids.numel()
    a.head_abs_acc = _ptr(head_abs)
    a.step_state = None if step is None else step.data_ptr()
    if append is not None:
        k_new, v_new, k_cache, kr_cache, v_cache = append
        _dev(k_new, v_new, k_cache, kr_cache, v_cache)
        if k_new.shape != (B, Hkv, d) or k_new.stride(2) != 1 or v_new.stride() != k_new.stride():
            raise ValueError("append: k_new / v_new [B,Hkv,d] with contiguous d and identical strides")
        if kr_cache.stride(3) != 1 or kr_cache.stride(2) != d or v_cache.stride() != kr_cache.stride() \
                or (k_cache is not None and k_cache.stride() != kr_cache.stride()) or kr_cache.shape[2] < kv_len:
            raise ValueError("append: cache planes need contiguous rows (pitch d), identical strides and >= kv_len rows")
        a.k_new, a.v_new, a.new_sb, a.new_sh = k_new.data_ptr(), v_new.data_ptr(), k_new.stride(0), k_new.stride(1)
        a.k_cache, a.kr_cache, a.v_cache = _ptr(k_cache), kr_cache.data_ptr(), v_cache.data_ptr()
        a.kv_sb, a.kv_sh = kr_cache.stride(0), kr_cache.stride(1)
    _lib.check(_lib.load().spatten_attn_decode_pq(ctypes.byref(a), stream), "spatten_attn_decode_pq")
    return out


def pq_pack(kr_cache: torch.Tensor, planes: PQPlanes, lo: int, hi: int):
    """Quantise rows [lo, hi) of the rotated shadow into the MSB / LSB planes."""
    _dev(kr_cache, planes.msb)
    B, Hkv, cap, d = kr_cache.shape
    rc = _lib.load().spatten_pq_pack(_dt(kr_cache), kr_cache.data_ptr(), kr_cache.stride(0), kr_cache.stride(1),
                                     planes.msb.data_ptr(), planes.lsb.data_ptr(), planes.scale.data_ptr(),
                                     planes.msb.stride(0), planes.msb.stride(1), planes.scale.stride(0), planes.scale.stride(1),
                                     B, Hkv, d, lo, hi, _stream())
    _lib.check(rc, "spatten_pq_pack")


def attn_decode_pq(q: torch.Tensor, planes: PQPlanes, v_cache: torch.Tensor, kv_len: int, cos: torch.Tensor,
                   sin: torch.Tensor, pos_q: int, threshold: float, out: Optional[torch.Tensor] = None,
                   need_lsb: Optional[torch.Tensor] = None, workspace: Optional[DecodeWorkspace] = None,
                   head_ids: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None,
                   scores: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decode over progressively quantised keys: MSB-plane pass, LSB refetch for heads whose max probability is
    below ``threshold``, softmax + P·V with the un-quantised V.  q [B,H,d]; need_lsb optional int32 [B*H]."""
    B, H, d = q.shape
    if need_lsb is None:
        need_lsb = torch.empty(B * H, dtype=torch.int32, device=q.device)
    return attn_decode(q, None, None, v_cache, kv_len, cos, sin, pos_q, out=out, workspace=workspace,
                       head_ids=head_ids, lse=lse, scores=scores, pq=(planes, threshold, need_lsb))
