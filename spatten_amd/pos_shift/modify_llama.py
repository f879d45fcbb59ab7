"""Cache-relative-RoPE ("pos_shift") attention forward for HF LlamaAttention on the HIP kernels.

Mirror of the reference's spatten_llm/pos_shift/modify_llama.py: same public names
(``apply_rotary_pos_emb_single``, ``llama_pos_shift_attention_forward``,
``enable_llama_pos_shift_attention``), same forward signature / return triple / side effect
(``self.attn_scores`` = raw scaled logits before mask and softmax, :116-119), same ValueErrors.

What changed underneath (reference lines in parentheses):
  * q_len == 1: ONE fused launch (ops.attn_decode) replaces RoPE(Q) (:92), torch.cat of the cache (:95-98),
    RoPE of the whole K cache (:103-104), repeat_kv (:108-109), QK^T/sqrt(d) (:111-113), the stash clone
    (:116-119), +mask (:132), fp32 softmax (:135-137), PV (:138) and the head merge (:146-147).
  * q_len  > 1: rope kernel for the new rows + MFMA flash prefill (ops.attn_prefill).
  * the KV cache lives in capacity slabs with a rotated shadow (kv_slab.py); the returned
    ``past_key_value`` is still ``(K_unrotated[B,Hkv,N,d], V[B,Hkv,N,d])`` — views, appended in place.
The q/k/v/o projections stay torch (hipBLASLt) GEMMs, as in the reference (:43-74, :149-163).
"""
from __future__ import annotations

import types
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .. import kv_slab, ops

__all__ = ["enable_llama_pos_shift_attention"]


def apply_rotary_pos_emb_single(x, cos, sin, position_ids):
    """modify_llama.py:21-28.  ``cos``/``sin`` are the 4.33-style tables [1,1,S,d] (or [S,d]); only their
    first d/2 columns are read (the table is ``cat(freqs, freqs)``)."""
    cos = cos.reshape(-1, cos.shape[-1])
    sin = sin.reshape(-1, sin.shape[-1])
    h = x.shape[-1] // 2
    return ops.rope_single(x, cos[:, :h].contiguous(), sin[:, :h].contiguous(), position_ids=position_ids)


def _cfg(self, name, *fallbacks, default=None):
    v = getattr(self, name, None)
    if v is not None:
        return v
    cfg = getattr(self, "config", None)
    for fb in fallbacks:
        v = getattr(cfg, fb, None)
        if v is not None:
            return v
    return default


def _rope_params(self):
    """(base, scaling) of the module's rotary embedding.  The reference calls ``self.rotary_emb(value_states, seq_len=)``
    (modify_llama.py:89), so whatever table that module builds is what it rotates with: the plain 4.33
    LlamaRotaryEmbedding (``base``), or the linear-scaling subclass (``scaling_factor``; config.rope_scaling type
    "linear").  Dynamic-NTK tables change with the sequence length — every cached key would have to be re-rotated when
    the table is rebuilt, which the rotated shadow cannot follow: rejected loudly instead of diverging silently."""
    rot = getattr(self, "rotary_emb", None)
    cfg = getattr(self, "config", None)
    base = getattr(rot, "base", None)
    if base is None:
        base = getattr(cfg, "rope_theta", None)
    base = float(base) if base is not None else 10000.0
    rs = getattr(cfg, "rope_scaling", None)
    kind = type(rot).__name__ if rot is not None else ""
    factor = getattr(rot, "scaling_factor", None)
    if isinstance(rs, dict) and rs.get("type", rs.get("rope_type")) not in (None, "default"):
        rtype = rs.get("type", rs.get("rope_type"))
        factor = rs.get("factor", factor)
        if rtype != "linear":
            raise NotImplementedError(f"rope_scaling type {rtype!r}: only the static tables (plain, linear) are supported")
        return base, ("linear", float(factor))
    if "DynamicNTK" in kind:
        raise NotImplementedError("LlamaDynamicNTKScalingRotaryEmbedding: only the static tables (plain, linear) are supported")
    if "LinearScaling" in kind and factor is not None:
        return base, ("linear", float(factor))
    return base, None


def llama_pos_shift_attention_forward(
    self,
    hidden_states: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    position_ids: Optional[torch.LongTensor] = None,
    past_key_value: Optional[Tuple[torch.Tensor]] = None,
    output_attentions: bool = False,
    use_cache: bool = False,
    padding_mask: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[Tuple[torch.Tensor]]]:
    bsz, q_len, _ = hidden_states.size()
    geom = self.__dict__.get("_spatten_geom")          # the module's geometry, read once (a decode step is host-bound)
    if geom is None:
        num_heads = _cfg(self, "num_heads", "num_attention_heads")
        num_kv_heads = _cfg(self, "num_key_value_heads", "num_key_value_heads", default=num_heads)
        head_dim = _cfg(self, "head_dim", "head_dim") or (_cfg(self, "hidden_size", "hidden_size") // num_heads)
        hidden_size = _cfg(self, "hidden_size", "hidden_size", default=num_heads * head_dim)
        tp = getattr(getattr(self, "config", None), "pretraining_tp", 1) or 1
        geom = self.__dict__["_spatten_geom"] = (num_heads, num_kv_heads, head_dim, hidden_size, tp)
    num_heads, num_kv_heads, head_dim, hidden_size, tp = geom

    # head-parallel (enable_spatten_llm(..., head_parallel=hp)): this rank projects, caches, attends and prunes
    # ONLY its H/G heads — column-sharded q/k/v projections — and the one exchange is the all-gather of the attention
    # outputs in front of the full o_proj (:146-163)
    hp = self.__dict__.get("_spatten_hp")
    if hp is not None:
        if tp > 1:
            raise NotImplementedError("head_parallel with config.pretraining_tp > 1")
        num_heads, num_kv_heads = hp[0].local_heads, hp[0].local_kv_heads
    # single-token rows through the library's weight-streaming kernel (opt-in), everything else through torch's GEMMs
    # (only for PLAIN nn.Linear projections of the activation dtype — all four: the kernel reads ``.weight`` directly, which
    #  would skip the forward of a wrapped projection (LoRA, a quantised linear exposing .weight) on single-token steps only)
    native_rows = (tp == 1 and bsz * q_len <= 4 and bool(self.__dict__.get("_spatten_gemv", False))
                   and hidden_states.is_cuda and _plain_linears(self, hidden_states.dtype))
    # opt-in (enable_spatten_llm(..., fused_step=True), with fuse_qkv + native_gemv): the single-token step's q / k / v
    # projections run INSIDE the attention launch (include/spatten.h: spatten_decode_args_t::qkv_*) — decided below, once the
    # slab is known; until then nothing is projected
    fused_qkv = None
    if (native_rows and q_len == 1 and bsz == 1 and hp is None and self.__dict__.get("_spatten_fused_step", False)
            and getattr(self, "_spatten_ext", None) is None and getattr(self, "_spatten_qkv", None) is not None
            and num_heads == num_kv_heads and past_key_value is not None):
        w, bias, nq, nk = self._spatten_qkv
        if self.q_proj.weight.data_ptr() == w.data_ptr() and w.dtype == hidden_states.dtype:
            fused_qkv = (w, bias)
    if fused_qkv is not None:
        query_states = key_states = value_states = None
    elif tp > 1:                                                                    # :43-69
        kv_slicing = (num_kv_heads * head_dim) // tp
        q_slices = self.q_proj.weight.split((num_heads * head_dim) // tp, dim=0)
        k_slices = self.k_proj.weight.split(kv_slicing, dim=0)
        v_slices = self.v_proj.weight.split(kv_slicing, dim=0)
        query_states = torch.cat([F.linear(hidden_states, q_slices[i]) for i in range(tp)], dim=-1)
        key_states = torch.cat([F.linear(hidden_states, k_slices[i]) for i in range(tp)], dim=-1)
        value_states = torch.cat([F.linear(hidden_states, v_slices[i]) for i in range(tp)], dim=-1)
    elif hp is not None:
        lin = ops.gemv if native_rows else F.linear
        if hp[2] is not None:                       # fuse_qkv: the three local slices stacked into one weight
            w, bias, nq, nk = hp[2]
            qkv = lin(hidden_states, w, bias)
            query_states, key_states, value_states = qkv[..., :nq], qkv[..., nq:nq + nk], qkv[..., nq + nk:]
        else:
            (qw, qb), (kw, kb), (vw, vb) = hp[1]
            query_states, key_states, value_states = lin(hidden_states, qw, qb), lin(hidden_states, kw, kb), lin(hidden_states, vw, vb)
    elif getattr(self, "_spatten_qkv", None) is not None:
        # opt-in (enable_spatten_llm(..., fuse_qkv=True)): ONE GEMM over the stacked q/k/v weights — a single-token step
        # is host-bound and each torch linear costs ~20 us of launch path; the three results are slices of one row
        w, bias, nq, nk = self._spatten_qkv
        qw = self.q_proj.weight
        if qw.data_ptr() != w.data_ptr() or qw.dtype != w.dtype or qw.device != w.device:
            # the parameters were moved / cast / reassigned after the fusion (model.to(), .half()): stack them again
            self._spatten_qkv = None
            if not fuse_qkv_projections(self):
                raise RuntimeError("fuse_qkv: the q/k/v projections can no longer be stacked (mixed dtype / device / bias)")
            w, bias, nq, nk = self._spatten_qkv
        qkv = ops.gemv(hidden_states, w, bias) if native_rows else F.linear(hidden_states, w, bias)
        query_states, key_states, value_states = qkv[..., :nq], qkv[..., nq:nq + nk], qkv[..., nq + nk:]
    elif native_rows:
        # opt-in (enable_spatten_llm(..., native_gemv=True)): a single-token projection is a 2-FLOP-per-byte weight
        # stream, HBM-bound like the attention next to it — the library's streaming kernel instead of the GEMM library
        query_states = ops.gemv(hidden_states, self.q_proj.weight, self.q_proj.bias)
        key_states = ops.gemv(hidden_states, self.k_proj.weight, self.k_proj.bias)
        value_states = ops.gemv(hidden_states, self.v_proj.weight, self.v_proj.bias)
    else:                                                                         # :72-74
        query_states = self.q_proj(hidden_states)
        key_states = self.k_proj(hidden_states)
        value_states = self.v_proj(hidden_states)

    dtype, device = hidden_states.dtype if query_states is None else query_states.dtype, hidden_states.device
    past_len = 0 if past_key_value is None else past_key_value[0].shape[-2]       # :86-88
    kv_seq_len = past_len + q_len
    ext = getattr(self, "_spatten_ext", None)         # (SpattenExtensions, layer index) — opt-in SpAtten semantics
    assume_causal = bool(getattr(self, "spatten_assume_causal", False)) or ext is not None
    if ext is not None and ext[0].layer_keep is None and attention_mask is not None and q_len > 1 \
            and not bool(getattr(self, "spatten_assume_causal", False)) and attention_mask.numel() \
            and not _mask_is_causal(attention_mask, past_len):
        # the extension modes run on the causal rule (their kernels take no mask): a padding / custom mask would be dropped
        # silently — refuse it instead (round-2 advisor finding).  One device comparison per forward, cached on the tensor.
        raise ValueError("the SpAtten extension modes (cascade importance, head / local-V / layer pruning, progressive "
                         "quantisation) assume the HF causal mask; this forward was given another attention_mask "
                         "(left-padded batch?) — run it without the extensions, or pass assume_causal=True to take the "
                         "responsibility")
    if ext is not None and ext[0].layer_keep is not None:
        # layer-to-layer cascade: the layers' caches have different lengths, HF sizes its mask / positions for layer 0 —
        # every layer uses the causal rule and its own cache-relative positions instead
        attention_mask = None if q_len == 1 else attention_mask[..., :0]
        position_ids = None
    if attention_mask is not None and attention_mask.numel() and attention_mask.size() != (bsz, 1, q_len, kv_seq_len):   # :127-131
        raise ValueError(
            f"Attention mask should be of size {(bsz, 1, q_len, kv_seq_len)}, but is {attention_mask.size()}")

    rope = getattr(self, "_spatten_rope", None)       # (base, scaling), resolved once per module
    if rope is None:
        rope = self._spatten_rope = _rope_params(self)
    base, scaling = rope
    pk, pv = (None, None) if past_key_value is None else (past_key_value[0], past_key_value[1])
    slab = kv_slab.slab_for(pk, pv, kv_seq_len, bsz, num_kv_heads, head_dim, dtype, device, base, scaling)
    cos, sin = slab.tables(kv_seq_len)
    if position_ids is not None and assume_causal:
        # HF's own position_ids at this call are arange(past_len, past_len + q_len) (4.33 LlamaModel.forward): with
        # assume_causal the tensor is not read — the query positions are a launch constant, which is what lets a
        # single-token step run the lean decode kernel (no position tensor, no mask)
        position_ids = None
    if position_ids is not None and not (position_ids.dtype == torch.int64 and position_ids.device == device
                                         and position_ids.shape == (bsz, q_len) and position_ids.is_contiguous()):
        position_ids = position_ids.to(device=device, dtype=torch.int64)
        if position_ids.dim() == 1:
            position_ids = position_ids[None]
        if position_ids.shape[0] not in (1, bsz) or position_ids.shape[-1] != q_len:
            raise ValueError(f"position_ids should be of size {(bsz, q_len)}, but is {tuple(position_ids.shape)}")
        if position_ids.shape[0] == 1 and bsz > 1:
            position_ids = position_ids.expand(bsz, q_len)
        position_ids = position_ids.contiguous()

    # extension: no [B,H,q,N] stash for multi-token forwards (enable_spatten_llm(..., prefill_stash=False))
    want_stash = q_len == 1 or output_attentions or bool(getattr(self, "spatten_prefill_stash", True))
    if attention_mask is not None and not (assume_causal and not output_attentions):
        attention_mask = attention_mask.to(dtype)
    slab.ensure_shadow(past_len)                     # a foreign past: its rows get their rotation now (:103-104)
    if q_len == 1:
        gctx = kv_slab.graph_ctx_for(slab)     # this thread's tracing DecodeGraph, if THIS cache is one it is bound to
        fused_out = None
        if fused_qkv is not None:
            # the projections inside the attention launch — where the slab's shape runs it (else: project now, as usual)
            fproj = (self.o_proj.weight, self.o_proj.bias)
            # (only for EAGER calls: under a captured graph the separate launches are 2 % faster — DESIGN 3.13 — and the two
            #  forms are bit-identical, so a traced step simply projects first)
            if gctx is not None:
                pass
            elif position_ids is None and (attention_mask is None or assume_causal):
                stash = torch.empty(bsz, num_heads, 1, kv_seq_len, dtype=dtype, device=device)
                fused_out = slab.decode_step_qkv(hidden_states, fused_qkv[0], fused_qkv[1], num_heads, kv_seq_len, past_len, cos,
                                                 sin, stash.view(bsz, num_heads, kv_seq_len), proj=fproj)
            if fused_out is None:
                qkv = ops.gemv(hidden_states, fused_qkv[0], fused_qkv[1])
                nq = num_heads * head_dim
                query_states, key_states, value_states = qkv[..., :nq], qkv[..., nq:2 * nq], qkv[..., 2 * nq:]
        if fused_out is None:
            q3 = query_states.view(bsz, num_heads, head_dim)
            k3, v3 = key_states.view(bsz, num_kv_heads, head_dim), value_states.view(bsz, num_kv_heads, head_dim)
        if gctx is not None and attention_mask is not None and attention_mask.numel() \
                and not torch.cuda.is_current_stream_capturing() and bool((attention_mask != 0).any().item()):
            # the captured step does not read the mask (device-length causal rule); checked on the eager warm-up step
            raise ValueError("DecodeGraph: the single-token step was given a non-zero attention_mask (a padded batch?) — "
                             "the captured step attends to the whole cache")
        # native projections (opt-in) on the plain path: o_proj rides in the attention call (include/spatten.h: proj_* — the
        # library issues both launches: one host call per layer-step)
        fused_proj = (self.o_proj.weight, self.o_proj.bias) if (native_rows and hp is None and ext is None) else None
        projected = None
        if fused_out is not None:
            attn_output = fused_out
        elif ext is not None and gctx is not None:
            if not ext[0].graph_capable():
                raise RuntimeError("DecodeGraph captures every mode but local V pruning COMBINED with cascade importance "
                                   "(its accumulation runs on host lengths)")
            attn_output, stash = ext[0].decode_step_graph(ext[1], q3, k3, v3, slab, kv_seq_len, cos, sin, gctx)
            gctx.touched.append((self, slab, ext))
        elif ext is not None:
            attn_output, stash = ext[0].decode_step(ext[1], q3, k3, v3, slab, kv_seq_len, past_len, cos, sin)
        elif gctx is not None:
            # a DecodeGraph is warming up / capturing this step (spatten_amd/graph.py): the cache length is read from the
            # graph's device-resident step state, the logits go to the slab's persistent stash row — nothing in the
            # launch depends on a host value that changes from token to token.  The HF mask / position_ids of a
            # single-token step (zeros / past_len, transformers 4.33) are not read, as with assume_causal.
            row = slab.stash_row(num_heads)
            attn_output = slab.decode_step(q3, k3, v3, kv_seq_len, past_len, cos, sin, row, step=gctx.state_for(slab, cos, sin),
                                           proj=fused_proj)
            stash = row[:, :, None, :kv_seq_len]
            gctx.touched.append((self, slab, None))
        else:
            # the slab's prefilled argument block (host-path fast lane: per token only pointers and two lengths are
            # written).  With assume_causal the HF mask of a single-token step (all zeros) and its position_ids are not
            # read: the lean decode kernel.
            stash = torch.empty(bsz, num_heads, 1, kv_seq_len, dtype=dtype, device=device)
            attn_output = slab.decode_step(
                q3, k3, v3, kv_seq_len, past_len, cos, sin, stash.view(bsz, num_heads, kv_seq_len),
                None if position_ids is None else position_ids[:, 0],
                None if (attention_mask is None or assume_causal) else attention_mask[:, 0, 0, :], proj=fused_proj)
        slab.length = slab.rot_len = kv_seq_len
        if isinstance(attn_output, tuple):
            attn_output, projected = attn_output
            projected = projected.view(bsz, 1, -1)
        attn_output = attn_output.view(bsz, 1, num_heads * head_dim)
    else:
        projected = None
        stash = torch.empty(bsz, num_heads, q_len, kv_seq_len, dtype=dtype, device=device) if want_stash else None
        ops.kv_append(key_states.view(bsz, q_len, num_kv_heads, head_dim).transpose(1, 2),
                      value_states.view(bsz, q_len, num_kv_heads, head_dim).transpose(1, 2),
                      slab.k, slab.kr, slab.v, past_len, cos, sin)                 # :95-104 without the cat
        slab.length = slab.rot_len = kv_seq_len
        # a mask that IS the HF causal mask lets the kernel skip the tiles above the diagonal and never read the mask
        causal = attention_mask is not None and (assume_causal or _mask_is_causal(attention_mask, past_len))
        q4 = query_states.view(bsz, q_len, num_heads, head_dim).transpose(1, 2)
        pmask = None if (attention_mask is None or causal) else attention_mask[:, 0]
        lse_pf = None
        if ext is not None and ext[0].prefill_uses_pq(dtype, head_dim, q_len):
            # progressive quantisation also at multi-token forwards (BASELINE.json configs[3]): MSB-first keys, per-row
            # max-probability decision, LSB refetch of the flagged rows; the logits are never materialised here
            slab.ensure_pq(kv_seq_len)
            stash = None
            attn_output, _ = ops.attn_prefill_pq(q4, slab.pq, slab.v, kv_seq_len, cos, sin, past_len, ext[0].pq_threshold,
                                                 causal=causal, position_ids=position_ids, mask=pmask)
        else:
            if ext is not None and ext[0].prefill_wants_lse(dtype, head_dim, q_len, pmask is not None):
                lse_pf = torch.empty(bsz, num_heads, q_len, 2, dtype=torch.float32, device=device)
                if not (output_attentions or bool(getattr(self, "spatten_prefill_stash", True))):
                    stash = None
            attn_output = ops.attn_prefill(q4, slab.kr, slab.v, kv_seq_len, cos, sin, past_len, causal=causal,
                                           position_ids=position_ids, mask=pmask, scores=stash, lse=lse_pf,
                                           numerics=self.__dict__.get("_spatten_numerics", "auto"))
        if ext is not None:
            ext[0].after_prefill(ext[1], attn_output, stash, pmask, num_heads, causal, q4=q4, slab=slab, kv_len=kv_seq_len,
                                 cos=cos, sin=sin, past_len=past_len, position_ids=position_ids,
                                 lse=lse_pf)

    # store attention scores for deciding which token to prune (:116-119) — raw scaled logits, pre-mask
    object.__setattr__(self, "attn_scores", stash)      # (nn.Module.__setattr__ costs ~2.5 us of type checks per call)

    if hp is not None:
        # [B, q, H/G*d] of every rank -> [B, q, H*d], rank-major = head-major: the reference's transpose(1, 2).reshape
        # layout (:146-147).  (attn_scores, the KV cache and hence the prune stay local: they are per head.)
        attn_output, _ = hp[0].gather_heads(attn_output.reshape(bsz, q_len, num_heads * head_dim))
    if attn_output.size() != (bsz, q_len, hidden_size):                           # :140-147
        raise ValueError(
            f"`attn_output` should be of size {(bsz, q_len, hidden_size)}, but is {attn_output.size()}")

    if tp > 1:                                                                    # :149-161
        attn_output = attn_output.split(hidden_size // tp, dim=2)
        o_slices = self.o_proj.weight.split(hidden_size // tp, dim=1)
        attn_output = sum(F.linear(attn_output[i], o_slices[i]) for i in range(tp))
    elif projected is not None:
        attn_output = projected                                                   # :163, computed inside the attention call
    elif native_rows:
        attn_output = ops.gemv(attn_output, self.o_proj.weight, self.o_proj.bias)
    else:
        attn_output = self.o_proj(attn_output)                                    # :163

    attn_weights = None
    if output_attentions:                                                         # :135-137 on request only
        logits = stash if attention_mask is None else stash + attention_mask.to(dtype)
        attn_weights = torch.softmax(logits, dim=-1, dtype=torch.float32).to(dtype)

    new_past = slab.views() if use_cache else None                                # :100
    return attn_output, attn_weights, new_past


def _plain_linears(module, dtype) -> bool:
    """All four projections are exactly ``torch.nn.Linear`` with weights of ``dtype`` (answer cached per module and dtype)."""
    mods = module._modules
    ident = (dtype, id(mods.get("q_proj")), id(mods.get("k_proj")), id(mods.get("v_proj")), id(mods.get("o_proj")))
    key = module.__dict__.get("_spatten_plain")
    if key is not None and key[0] == ident:
        return key[1]
    ok = all(type(mods.get(n)) is torch.nn.Linear and mods[n].weight.dtype == dtype
             for n in ("q_proj", "k_proj", "v_proj", "o_proj"))
    module.__dict__["_spatten_plain"] = (ident, ok)
    return ok


def fuse_qkv_projections(module) -> bool:
    """Stack q_proj / k_proj / v_proj of one attention module into ONE weight [nq + 2 nk, hidden] (and bias) for the
    patched forward; the three modules keep working — their ``weight`` / ``bias`` become views of the stacked tensors, so
    no memory is duplicated.  Not applied (returns False) with ``pretraining_tp > 1`` or mixed bias / dtype / device."""
    tp = getattr(getattr(module, "config", None), "pretraining_tp", 1) or 1
    projs = [getattr(module, n, None) for n in ("q_proj", "k_proj", "v_proj")]
    if tp > 1 or any(p is None or not hasattr(p, "weight") for p in projs):
        return False
    ws = [p.weight for p in projs]
    bs = [getattr(p, "bias", None) for p in projs]
    if len({(w.dtype, w.device, w.shape[1]) for w in ws}) != 1 or len({b is None for b in bs}) != 1:
        return False
    with torch.no_grad():
        w = torch.cat([x.detach() for x in ws], dim=0).contiguous()
        bias = None if bs[0] is None else torch.cat([x.detach() for x in bs], dim=0).contiguous()
        off = 0
        for p_, x in zip(projs, ws):
            n = x.shape[0]
            p_.weight.data = w[off:off + n]
            if bias is not None:
                p_.bias.data = bias[off:off + n]
            off += n
    module._spatten_qkv = (w, bias, ws[0].shape[0], ws[1].shape[0])
    return True


def _mask_is_causal(mask: torch.Tensor, past_len: int) -> bool:
    """True when ``mask`` [B,1,q,N] is exactly the additive causal mask transformers 4.33 builds (0 for j <= P + i,
    finfo.min above).  HF hands the SAME tensor object to every decoder layer of a forward, so the answer (one device
    comparison and one host sync per prefill forward) is remembered on the tensor."""
    cached = getattr(mask, "_spatten_is_causal", None)
    if cached is not None and cached[0] == (mask._version, past_len):
        return cached[1]
    q_len, n = mask.shape[-2], mask.shape[-1]
    i = torch.arange(q_len, device=mask.device)[:, None]
    j = torch.arange(n, device=mask.device)[None, :]
    lo = torch.finfo(mask.dtype).min
    want = torch.where(j <= past_len + i, torch.zeros((), dtype=mask.dtype, device=mask.device),
                       torch.full((), lo, dtype=mask.dtype, device=mask.device))
    ok = bool((mask == want).all().item())
    try:
        mask._spatten_is_causal = ((mask._version, past_len), ok)
    except Exception:       # a tensor subclass that refuses attributes: just recompute next time
        pass
    return ok


def attention_modules(model):
    """The patched attention modules in ``model.modules()`` order (= layer order, run_spatten_llama.py:74-77)."""
    return [m for m in model.modules() if _is_llama_attention(m)]


def _is_llama_attention(module) -> bool:
    if getattr(module, "_spatten_llama_attention", False) or type(module).__name__ == "LlamaAttention":
        return True
    try:
        from transformers.models.llama.modeling_llama import LlamaAttention
    except Exception:  # transformers absent: duck typing above is all there is
        return False
    return isinstance(module, LlamaAttention)


def enable_llama_pos_shift_attention(model):
    """modify_llama.py:171-181: depth-first over ``model._modules`` (reversed), rebinding ``forward`` on
    every LlamaAttention instance."""
    for name, module in reversed(model._modules.items()):
        if len(list(module.children())) > 0:
            enable_llama_pos_shift_attention(module)
        if _is_llama_attention(module):
            model._modules[name].forward = types.MethodType(llama_pos_shift_attention_forward, model._modules[name])


def shard_attention_projections(module, hp, fuse: bool = False):
    """Give one patched attention module its head-parallel form: the rows of q_proj / k_proj / v_proj that produce this
    rank's heads (``HeadParallel.shard_projection``: column-sharded projections, copies), optionally stacked into one
    weight; ``o_proj`` stays whole (it consumes the gathered [B, q, H*d]).  The module's own parameters are not touched —
    a caller that wants the memory back can drop q/k/v_proj afterwards."""
    heads = _cfg(module, "num_heads", "num_attention_heads")
    kv_heads = _cfg(module, "num_key_value_heads", "num_key_value_heads", default=heads)
    head_dim = _cfg(module, "head_dim", "head_dim") or (_cfg(module, "hidden_size", "hidden_size") // heads)
    if heads != hp.num_heads or kv_heads != hp.num_kv_heads:
        raise ValueError(f"head_parallel was built for {hp.num_heads}/{hp.num_kv_heads} heads, the module has {heads}/{kv_heads}")
    parts = [hp.shard_projection(module.q_proj.weight, getattr(module.q_proj, "bias", None), head_dim),
             hp.shard_projection(module.k_proj.weight, getattr(module.k_proj, "bias", None), head_dim, kv=True),
             hp.shard_projection(module.v_proj.weight, getattr(module.v_proj, "bias", None), head_dim, kv=True)]
    stacked = None
    if fuse and len({b is None for _, b in parts}) == 1:
        w = torch.cat([w_ for w_, _ in parts], dim=0).contiguous()
        b = None if parts[0][1] is None else torch.cat([b_ for _, b_ in parts], dim=0).contiguous()
        stacked = (w, b, parts[0][0].shape[0], parts[1][0].shape[0])
    module.__dict__["_spatten_hp"] = (hp, parts, stacked)
