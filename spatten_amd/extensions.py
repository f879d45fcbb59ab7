"""SpAtten's cascade semantics wired INTO the plugin surface — opt-in keyword arguments of
``enable_spatten_llm``; with none of them set the plugin is the reference's behaviour and this module is never touched.

PARITY UNPINNED: the reference's Python implements none of this (README.md:11,21 describe it, the RTL under
spatten_hardware/ carries the control flow); the rules are restated in oracle/spatten_oracle.py and the tests compare
against those restatements.

    importance_mode="cascade"   cumulative token importance = running sum of softmax probabilities (README.md:11; trace
                                flag if_accumulate_importance).  Decode steps fold the PREVIOUS step's probabilities into
                                the accumulator inside the attention launch (no extra kernel); the last step before a
                                prune is folded by the prune.
    head_keep=k | [k_0..k_L-1]  cascade head pruning (README.md:21): head importance = cumulative sum |attn_out_h|
                                (accumulated by the attention launch's merge step), summed layer over layer; at every
                                prune event layer l keeps the k_l best heads among those layer l-1 kept (and it kept
                                before) — once pruned, a head stays pruned here and in all later layers.  Pruned heads are
                                not launched; their slice of attn_output is zero.
    pq_threshold=t              progressive quantisation of the keys at decode (RequantDecision.scala:44-72;
                                MatrixFetcher.scala:341-348): MSB plane first, LSB refetch for heads whose max probability
                                is below t.
    pq_profile=(kb, vb)         the bit profile of the planes (MatrixFetcher.scala:48-51; TestSpAtten.scala:64,83-97,173-176):
                                key MSB plane of kb bits (+ 4 LSBs on refetch), VALUE plane of vb bits — (4, 8), (8, 8) (the RTL
                                harness default) or (6, 6) (the per8 trace).  None: 4-bit MSB, V in the model dtype (r02).
                                With a profile the refetch pass reads ONLY the LSB plane (+ the stashed MSB logits).
    local_v_keep=f              local V pruning at decode (SpAttenController.scala:546-558,591-612): only the
                                ceil(f * kv_len) most probable keys of a head fetch their V row.
    layer_keep=[k_0..k_L-1]     layer-to-layer cascade token pruning (README.md:11; the traces' key_fetch_num shrinks layer
                                by layer): at a prune event layer l keeps k_l tokens of its window, chosen among the
                                tokens layer l-1 just kept — the surviving set shrinks through the layers.  Caches of
                                different layers then have different lengths; every layer rotates with its own
                                cache-relative positions.
All modes assume the HF causal mask (a single-token step sees every key), like ``assume_causal=True``.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Union

import torch

from . import kv_slab, ops


class LayerState:
    """Per-layer device state of the extensions: double-buffered stash / (max, sum) (the previous step's pair feeds the
    fused cascade accumulation), the accumulators, the kept-head list."""

    def __init__(self):
        self.stash: List[Optional[torch.Tensor]] = [None, None]     # [B, H, cap] model dtype
        self.lse: List[Optional[torch.Tensor]] = [None, None]       # [B, H, 2] fp32
        self.parity = 0
        self.pending_len = 0            # rows of stash[parity ^ 1] not yet folded into acc (0 = nothing pending)
        self.acc: Optional[torch.Tensor] = None                     # [H, cap] fp32 cumulative importance
        self.head_abs: Optional[torch.Tensor] = None                # [B*H] fp32: sum |attn_out| per (b, h), decode steps
        self.head_abs_prefill: Optional[torch.Tensor] = None        # [H] fp32: the same from multi-token forwards
        self.head_ids: Optional[torch.Tensor] = None                # int32 ascending kept heads of THIS process (None = all)
        self.kept_global: Optional[torch.Tensor] = None             # int32 kept heads of the whole model (head-parallel: all ranks)
        self.pruned_ids: Optional[torch.Tensor] = None              # int64 pruned heads
        self.need_lsb: Optional[torch.Tensor] = None                # int32 [B*H]
        self.out: Optional[torch.Tensor] = None                     # [B, H*d], zero in the pruned heads' slices

    def ensure(self, B, H, d, n, dtype, device, want_acc):
        cap = kv_slab.round_capacity(n + kv_slab.GROW)
        if self.stash[0] is None or self.stash[0].shape[0] != B or self.stash[0].shape[2] < n:
            old = self.stash
            self.stash = [torch.zeros(B, H, cap, dtype=dtype, device=device) for _ in range(2)]
            for i in range(2):          # a pending stash survives the growth
                if old[i] is not None and old[i].shape[0] == B:
                    self.stash[i][:, :, :old[i].shape[2]] = old[i]
            if self.lse[0] is None or self.lse[0].shape[0] != B:
                self.lse = [torch.zeros(B, H, 2, dtype=torch.float32, device=device) for _ in range(2)]
                self.pending_len = 0
        if want_acc and (self.acc is None or self.acc.shape[1] < n):
            grown = torch.zeros(H, cap, dtype=torch.float32, device=device)
            if self.acc is not None:
                grown[:, :self.acc.shape[1]] = self.acc
            self.acc = grown
        if self.head_abs is None or self.head_abs.numel() != B * H:
            self.head_abs = torch.zeros(B * H, dtype=torch.float32, device=device)
            self.head_abs_prefill = torch.zeros(H, dtype=torch.float32, device=device)
        if self.need_lsb is None or self.need_lsb.numel() != B * H:
            self.need_lsb = torch.zeros(B * H, dtype=torch.int32, device=device)
        if self.out is None or self.out.shape != (B, H * d) or self.out.dtype != dtype:
            self.out = torch.zeros(B, H * d, dtype=dtype, device=device)


def _per_layer(x, n_layers: int, name: str):
    if x is None:
        return None
    if isinstance(x, (int, float)):
        return [x] * n_layers
    x = list(x)
    if len(x) != n_layers:
        raise ValueError(f"{name} needs one entry per layer ({n_layers}), got {len(x)}")
    return x


class SpattenExtensions:
    def __init__(self, cache, n_layers: int, cascade: bool = False,
                 head_keep: Union[None, int, Sequence[int]] = None, pq_threshold: Optional[float] = None,
                 local_v_keep: Optional[float] = None, layer_keep: Optional[Sequence[int]] = None, head_parallel=None,
                 pq_profile: Optional[Sequence[int]] = None):
        if pq_profile is not None:
            pq_profile = (int(pq_profile[0]), int(pq_profile[1]))
            if pq_threshold is None:
                raise ValueError("pq_profile needs pq_threshold")
            if pq_profile not in ops.PQ_PROFILES:
                raise ValueError(f"pq_profile {pq_profile}: one of {ops.PQ_PROFILES} (key MSB bits, value bits)")
            if cascade:
                raise ValueError("pq_profile with importance_mode='cascade': the profiled decode launch does not carry the fused "
                                 "accumulation — use the default planes (pq_profile=None)")
        self.pq_profile = pq_profile
        if pq_threshold is not None and local_v_keep is not None:
            raise ValueError("pq_threshold and local_v_keep cannot be combined (the local-V pass scores from the bf16 shadow)")
        if local_v_keep is not None and not (0.0 < float(local_v_keep) <= 1.0):
            raise ValueError("local_v_keep is a fraction in (0, 1]")
        self.cache = cache
        self.hp = head_parallel                     # spatten_amd.parallel.HeadParallel: the layers hold H/G local heads
        self.n_layers = n_layers
        self.cascade = bool(cascade)
        self.head_keep = _per_layer(head_keep, n_layers, "head_keep")
        if self.head_keep is not None and any(b > a for a, b in zip(self.head_keep, self.head_keep[1:])):
            raise ValueError("head_keep must not grow from layer to layer (a pruned head stays pruned in later layers)")
        self.pq_threshold = None if pq_threshold is None else float(pq_threshold)
        self.local_v_keep = None if local_v_keep is None else float(local_v_keep)
        self.layer_keep = None if layer_keep is None else [int(x) for x in _per_layer(list(layer_keep), n_layers, "layer_keep")]
        if self.layer_keep is not None:
            if any(b > a for a, b in zip(self.layer_keep, self.layer_keep[1:])) or min(self.layer_keep) <= 0:
                raise ValueError("layer_keep must be positive and must not grow from layer to layer")
            if self.layer_keep[0] > cache.important_size:
                raise ValueError("layer_keep[0] exceeds important_size")
        self.layers = [LayerState() for _ in range(n_layers)]
        # layer cascade: token id held by every cache slot, per layer and head (int32 [H, len]); ids grow with time
        self.tok_ids: List[Optional[torch.Tensor]] = [None] * n_layers
        self.next_token_id = 0

    # ------------------------------------------------------------------------------------------------
    # attention forward, q_len == 1
    # ------------------------------------------------------------------------------------------------
    def graph_capable(self) -> bool:
        """Modes whose decode step runs in the device-length form (spatten_amd/graph.py): cascade importance and head
        pruning — one fused launch with fixed buffers — and progressive quantisation (the step's append + plane packing as
        one device-length launch, then the two passes over the planes); the layer cascade changes only the prune events (its
        layers decode on caches of different lengths: one step state per length).  Local V pruning is ONE launch whose kept
        count ceil(f * length) is evaluated on the device (round 4) — captured too, except together with cascade importance
        (its accumulation there runs on host lengths)."""
        return self.local_v_keep is None or not self.cascade

    def decode_step_graph(self, layer: int, q, k_new, v_new, slab, kv_len: int, cos, sin, gctx):
        """The decode step under a DecodeGraph: every buffer at capacity and at a fixed address; with cascade importance
        the two stash / (max, sum) buffers swap roles on the DEVICE (state word 2), so one captured graph serves every
        token.  Returns (attn_output [B, H*d], stash view of the buffer this step writes)."""
        if getattr(self, "_graph_bind", None) is not gctx.bind_id:     # first traced step of this binding (eager)
            self.graph_begin()
            self._graph_bind = gctx.bind_id
        st = self.layers[layer]
        B, H, d = q.shape
        cap = slab.capacity
        st.ensure(B, H, d, cap, q.dtype, q.device, self.cascade)
        if st.stash[0].shape[2] < cap or (self.cascade and st.acc.shape[1] < cap):
            raise RuntimeError("extension buffers smaller than the slab capacity")
        casc = (st.acc, st.stash[1], st.lse[1], 0) if self.cascade else None
        step = gctx.state_for(slab, cos, sin)
        if self.local_v_keep is not None:
            # local V pruning: the step's append in device-length form, then the one-launch step (the kept count follows the
            # device length); head pruning's zero-fill and head scores are stream operations on fixed buffers
            ops.attn_decode_local_v(q, slab.kr, slab.v, cap, cos, sin, 0, 1, st.stash[0], out=st.out, lse=st.lse[0],
                                    keep_fraction=self.local_v_keep, step=step, k_new=k_new, v_new=v_new, k_cache=slab.k)   # (r05: appends inside)
            if st.pruned_ids is not None:
                st.out.view(B, H, d).index_fill_(1, st.pruned_ids, 0)
            if self.head_keep is not None:
                ops.head_scores(st.out, H, st.head_abs_prefill)
            return st.out, st.stash[0][:, :, None, :kv_len]
        if self.pq_threshold is not None and self.pq_profile is not None:
            # profiled planes: rows [0, kv_len - 1) packed with host lengths before the capture, the step's row by the
            # device-length append + pack of that row (one launch: spatten_kv_append_planes), then the two passes over the planes
            slab.ensure_pq(kv_len - 1, self.pq_profile, H)
            if slab.pq.capacity < cap:
                raise RuntimeError("progressive-quant planes smaller than the slab capacity")
            if st.head_ids is None:      # (r05) every head is launched: the append + plane rows ride inside the MSB pass
                app = (k_new, v_new, slab.k, slab.kr, slab.v)
            else:                        # a head list: the pruned heads' rows are still appended (a later turn may re-rank the heads)
                ops.kv_append_planes(k_new, v_new, slab.k, slab.kr, slab.v, slab.pq, 0, cos, sin, step=step)
                app = None
            slab.pq_len = kv_len
            ops.attn_decode_pqv(q, slab.pq, cap, cos, sin, 0, self.pq_threshold, out=st.out, need_lsb=st.need_lsb,
                                scores=st.stash[0], lse=st.lse[0], head_ids=st.head_ids,
                                head_abs=st.head_abs if self.head_keep is not None else None, step=step, append=app)
            return st.out, st.stash[0][:, :, None, :kv_len]
        if self.pq_threshold is not None:
            # rows [0, kv_len - 1) packed with host lengths BEFORE the capture (the eager first step of the binding does it:
            # nothing is left for the captured trace), this step's row by the device-length append
            slab.ensure_pq(kv_len - 1)
            if slab.pq.msb.shape[2] < cap:
                raise RuntimeError("progressive-quant planes smaller than the slab capacity")
            ops.kv_append_step(k_new, v_new, slab.k, slab.kr, slab.v, step, slab.pq)
            slab.pq_len = kv_len
            ops.attn_decode(q, None, None, slab.v, cap, cos, sin, 0, out=st.out, scores=st.stash[0], lse=st.lse[0],
                            head_ids=st.head_ids, cascade=casc, pq=(slab.pq, self.pq_threshold, st.need_lsb),
                            head_abs=st.head_abs if self.head_keep is not None else None, step=step)
            cur = (gctx.steps_traced & 1) if self.cascade else 0
            return st.out, st.stash[cur][:, :, None, :kv_len]
        ops.attn_decode(q, slab.k, slab.kr, slab.v, cap, cos, sin, 0, k_new=k_new, v_new=v_new, out=st.out,
                        scores=st.stash[0], lse=st.lse[0], head_ids=st.head_ids, cascade=casc,
                        head_abs=st.head_abs if self.head_keep is not None else None, step=step)
        cur = (gctx.steps_traced & 1) if self.cascade else 0
        return st.out, st.stash[cur][:, :, None, :kv_len]

    def graph_begin(self):
        """A DecodeGraph binds: fold whatever decode step is still pending and start from buffer 0."""
        for layer, st in enumerate(self.layers):
            self.flush(layer)
            st.parity = 0

    def graph_sync(self, steps: int, length):
        """Host view of the per-layer state after ``steps`` device-length steps (replays do not run Python); ``length``:
        the cache length, or one per layer."""
        lengths = list(length) if isinstance(length, (list, tuple)) else [int(length)] * len(self.layers)
        for st, n in zip(self.layers, lengths):
            if st.stash[0] is None:
                continue
            if self.cascade:
                st.parity = steps & 1
                st.pending_len = n if steps > 0 else st.pending_len

    def decode_step(self, layer: int, q, k_new, v_new, slab, kv_len: int, past_len: int, cos, sin):
        """q [B,H,d], k_new / v_new [B,Hkv,d].  Returns (attn_output [B, H*d], stash view [B,H,1,kv_len])."""
        st = self.layers[layer]
        B, H, d = q.shape
        st.ensure(B, H, d, kv_len, q.dtype, q.device, self.cascade)
        cur = st.parity
        st.parity ^= 1
        stash, lse = st.stash[cur], st.lse[cur]
        casc = None
        if self.cascade and st.pending_len > 0:
            casc = (st.acc, st.stash[cur ^ 1], st.lse[cur ^ 1], min(st.pending_len, kv_len))
        head_abs = st.head_abs if self.head_keep is not None else None
        if self.pq_threshold is not None and self.pq_profile is not None:
            slab.ensure_pq(past_len, self.pq_profile, H)                  # (rows a prefill left unpacked, if any)
            if st.head_ids is None:
                app = (k_new, v_new, slab.k, slab.kr, slab.v)
            else:
                ops.kv_append_planes(k_new, v_new, slab.k, slab.kr, slab.v, slab.pq, past_len, cos, sin)
                app = None
            slab.pq_len = kv_len
            ops.attn_decode_pqv(q, slab.pq, kv_len, cos, sin, past_len, self.pq_threshold, out=st.out, need_lsb=st.need_lsb,
                                scores=stash, lse=lse, head_ids=st.head_ids, head_abs=head_abs, layout=slab.capacity, append=app)
        elif self.pq_threshold is not None:
            ops.kv_append(k_new[:, :, None], v_new[:, :, None], slab.k, slab.kr, slab.v, past_len, cos, sin)
            slab.ensure_pq(kv_len)
            # (splits laid out for the slab capacity, as below: the eager and the captured step then agree bit for bit)
            ops.attn_decode(q, None, None, slab.v, kv_len, cos, sin, past_len, out=st.out, scores=stash, lse=lse,
                            head_ids=st.head_ids, cascade=casc, pq=(slab.pq, self.pq_threshold, st.need_lsb),
                            head_abs=head_abs, layout=slab.capacity)
        elif self.local_v_keep is not None:
            from .cascade import local_v_decode
            keep = max(1, min(kv_len, math.ceil(self.local_v_keep * kv_len)))
            local_v_decode(q, slab.kr, slab.v, kv_len, cos, sin, past_len, keep, out=st.out, stash=stash, lse=lse,
                           layout=slab.capacity, k_new=k_new, v_new=v_new, k_cache=slab.k)      # (r05: the launch appends)
            if casc is not None:        # the scores-only launch does not carry the fused accumulation
                ops.importance_accumulate(st.acc, casc[1][:, :, None, :casc[3]], casc[2][:, :, None, :])
            if st.pruned_ids is not None:
                st.out.view(B, H, d).index_fill_(1, st.pruned_ids, 0)
            if head_abs is not None:
                ops.head_scores(st.out, H, st.head_abs_prefill)
        else:
            # (splits laid out for the slab capacity, like the plain plugin step: eager and captured steps then agree
            # bit for bit — kv_slab.KVSlab.decode_step)
            ops.attn_decode(q, slab.k, slab.kr, slab.v, kv_len, cos, sin, past_len, k_new=k_new, v_new=v_new,
                            out=st.out, scores=stash, lse=lse, head_ids=st.head_ids, cascade=casc, head_abs=head_abs,
                            layout=slab.capacity)
        if self.cascade:
            st.pending_len = kv_len
        return st.out, stash[:, :, None, :kv_len]

    def prefill_uses_pq(self, dtype, head_dim: int, q_len: int) -> bool:
        """The PQ-keyed flash kernel covers 16-bit dtypes at head_dim 64 / 128 and blocks of more than 8 rows (shorter
        blocks / fp32 run the exact rows leg on the un-quantised shadow)."""
        return (self.pq_threshold is not None and self.pq_profile is None and not self.cascade
                and dtype in (torch.float16, torch.bfloat16) and head_dim in (64, 128) and q_len > 8)

    # ------------------------------------------------------------------------------------------------
    # attention forward, q_len > 1: by-products of the flash path
    # ------------------------------------------------------------------------------------------------
    def prefill_wants_lse(self, dtype, head_dim: int, q_len: int, explicit_mask: bool) -> bool:
        """Cascade mode: the stash-free accumulation (row statistics + recomputed logits on the matrix cores) covers what
        the flash kernel covers under the causal / no mask; fp32, short blocks and explicit masks accumulate from the stash."""
        return (self.cascade and not explicit_mask and dtype in (torch.float16, torch.bfloat16) and head_dim in (64, 128)
                and q_len > 8)

    def after_prefill(self, layer: int, attn_output, stash, mask, num_heads: int, causal: bool = True, q4=None, slab=None,
                      kv_len: int = 0, cos=None, sin=None, past_len: int = 0, position_ids=None, lse=None):
        """attn_output [B,q,H*d]; stash [B,H,q,N] or None; mask additive [B,q,N] or None (then ``causal``); lse [B,H,q,2]
        when the forward ran the stash-free route (then q4 / slab / tables describe the same launch)."""
        st = self.layers[layer]
        B, q_len = attn_output.shape[0], attn_output.shape[1]
        d = attn_output.shape[2] // num_heads
        if self.cascade:
            n = kv_len if stash is None else stash.shape[-1]
            st.ensure(B, num_heads, d, n, attn_output.dtype, attn_output.device, True)
            self.flush(layer)
            if lse is not None:
                ops.importance_accumulate_prefill(st.acc, q4, slab.kr, kv_len, cos, sin, past_len, lse, causal=causal,
                                                  position_ids=position_ids)
            elif stash is not None:
                ops.importance_accumulate(st.acc, stash, None, mask, causal=causal)
            else:
                raise RuntimeError("importance_mode='cascade': this forward (fp32 / short block / explicit mask) accumulates "
                                   "from the stash — enable it (prefill_stash=True)")
        if self.head_keep is not None:
            st.ensure(B, num_heads, d, 1, attn_output.dtype, attn_output.device, False)
            if st.pruned_ids is not None:
                attn_output.view(B, q_len, num_heads, d).index_fill_(2, st.pruned_ids, 0)
            ops.head_scores(attn_output, num_heads, st.head_abs_prefill)

    def flush(self, layer: int):
        """Fold the last decode step's probabilities (still pending) into the accumulator."""
        st = self.layers[layer]
        if self.cascade and st.pending_len > 0:
            last = st.parity ^ 1
            n = st.pending_len
            ops.importance_accumulate(st.acc, st.stash[last][:, :, None, :n], st.lse[last][:, :, None, :])
            st.pending_len = 0

    # ------------------------------------------------------------------------------------------------
    # prune event
    # ------------------------------------------------------------------------------------------------
    def head_scores(self, layer: int) -> torch.Tensor:
        st = self.layers[layer]
        H = st.head_abs_prefill.numel()
        return st.head_abs.view(-1, H).sum(0) + st.head_abs_prefill

    def select_heads(self):
        """Cascade head pruning at a prune event: cumulative scores layer over layer, layer l picks its k_l best among
        the heads layer l-1 kept and it has not pruned before (ties: lowest head id)."""
        if self.head_keep is None or self.layers[0].head_abs is None:
            return
        dev = self.layers[0].head_abs.device
        Hl = self.layers[0].head_abs_prefill.numel()          # heads this process holds
        hp = self.hp
        H = Hl if hp is None else hp.num_heads
        lo = 0 if hp is None else hp.head_range()[0]
        cum = torch.zeros(H, dtype=torch.float32, device=dev)
        alive = torch.ones(H, dtype=torch.bool, device=dev)
        neg = torch.full((H,), float("-inf"), dtype=torch.float32, device=dev)
        for layer, st in enumerate(self.layers):
            if st.head_abs is None:
                break
            local = self.head_scores(layer)
            # head-parallel: ONE all-gather of the H/G local scores, then the same deterministic top-k on every rank
            # (head_keep counts heads of the whole model; every rank tracks the global kept set and launches its share)
            cum = cum + (local if hp is None else hp.gather_head_scores(local))
            if st.kept_global is not None:
                mine = torch.zeros(H, dtype=torch.bool, device=dev)
                mine[st.kept_global.long()] = True
                alive = alive & mine
            k = min(int(self.head_keep[layer]), H)
            k = min(k, int(alive.sum().item()))
            ids = ops.topk_select(torch.where(alive, cum, neg)[None, :].contiguous(), 0, H, k)[0].contiguous()
            alive = torch.zeros(H, dtype=torch.bool, device=dev)
            alive[ids.long()] = True
            st.kept_global = ids if k < H else None
            mine_alive = alive[lo:lo + Hl]
            if k < H:
                st.head_ids = mine_alive.nonzero().flatten().to(torch.int32).contiguous()     # local ids, ascending
                st.pruned_ids = (~mine_alive).nonzero().flatten()
                if st.head_ids.numel() == 0:
                    raise NotImplementedError("head pruning left this rank without a head in one layer")
            else:
                st.head_ids = st.pruned_ids = None
            if st.out is not None:
                st.out.zero_()

    def before_prune(self):
        for layer in range(self.n_layers):
            self.flush(layer)
        self.select_heads()

    def compact_importance(self, layer: int, idx, start: int, tail_lo: int, seq_len: int):
        st = self.layers[layer]
        if st.acc is not None:
            st.acc = ops.importance_compact(st.acc, idx, start, tail_lo, seq_len, st.acc.shape[1])
        st.pending_len = 0

    # ------------------------------------------------------------------------------------------------
    # layer-to-layer cascade: token ids follow the rows through prunes
    # ------------------------------------------------------------------------------------------------
    def token_ids(self, layer: int, heads: int, length: int, device) -> torch.Tensor:
        """int32 [H, length]: the ids of the slots already known (after the last prune) + fresh ids for the rows
        appended since (the same tokens in every layer, so the same ids)."""
        known = self.tok_ids[layer]
        n_known = 0 if known is None else known.shape[1]
        if layer == 0:
            self._append_base = self.next_token_id          # ids of the rows appended since the last prune start here
            self._append_count = length - n_known
        n_new = length - n_known
        fresh = (torch.arange(n_new, dtype=torch.int32, device=device) + self._append_base)[None, :].expand(heads, n_new)
        return fresh.contiguous() if known is None else torch.cat([known, fresh], dim=1)

    def after_layer_cascade(self, new_ids: List[torch.Tensor]):
        self.tok_ids = list(new_ids)
        self.next_token_id = self._append_base + self._append_count

    def stats(self):
        """Host-readable summary (synchronises): kept heads per layer, LSB refetches of the last step."""
        out = {"kept_heads": [None if st.head_ids is None else st.head_ids.cpu().tolist() for st in self.layers]}
        if self.pq_threshold is not None:
            out["lsb_refetch_last_step"] = [None if st.need_lsb is None else int(st.need_lsb.sum().item()) for st in self.layers]
        return out
