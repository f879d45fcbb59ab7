"""Plugin entry point of the drop-in surface.

Contract (reference: spatten_llm/enable_spatten_llm.py:5-23): ``enable_spatten_llm(model, start_size,
important_size, recent_size)`` patches the model's attention modules for cache-relative RoPE + score stashing and
returns the ``SpAttenKVCache`` that prunes its KV cache; model families the path does not cover raise
``ValueError("got <model_type>")``.
"""
from typing import Callable, Dict, Tuple

from .kv_cache_token_pruning import SpAttenKVCache

__all__ = ["enable_spatten_llm"]


def _patch_llama(model) -> Tuple[int, int]:
    from .pos_shift.modify_llama import enable_llama_pos_shift_attention

    enable_llama_pos_shift_attention(model)
    return 2, 2            # K and V are [B, H, L, d]: the sequence axis is dim 2


# model_type substring -> (patcher returning the (k_seq_dim, v_seq_dim) of that family's cache layout)
_FAMILIES: Dict[str, Callable] = {"llama": _patch_llama}


def enable_spatten_llm(model, start_size, important_size, recent_size, importance_mode="reference",
                       prefill_stash=True, assume_causal=False, head_keep=None, pq_threshold=None, local_v_keep=None,
                       layer_keep=None, fuse_qkv=False, native_gemv=False, head_parallel=None, numerics="auto",
                       auto_graph=False, pq_profile=None, fused_step=False, token_scope="head"):
    """The reference's four positional arguments (enable_spatten_llm.py:5) plus opt-in extensions:

    ``prefill_stash=False``: forwards with ``q_len > 1`` do not materialise ``self.attn_scores`` ([B,H,q,N]: 4 GiB per
    layer at q = N = 8192; the reference writes it on every forward, modify_llama.py:116-119) — it is ``None`` there
    (cascade mode then accumulates from the flash kernel's row statistics; fp32 / short blocks still need the stash).  The
    caller protocol (run_spatten_llama.py:71-79) prunes from the LAST DECODE step's stash, which is still written.
    ``assume_causal=True``: the HF mask / position_ids of a forward are taken to be what transformers 4.33 builds
    (causal mask, positions arange(past, past+q)) and are not read: tiles above the diagonal are skipped and a
    single-token step runs the lean decode kernel.
    ``fuse_qkv=True``: the q / k / v projections of every patched module run as ONE torch GEMM over the stacked weights
    (host-bound decode: two launches fewer per layer; a different GEMM shape, so results can differ from three separate
    ``nn.Linear`` calls in the last bit).
    ``native_gemv=True``: the q / k / v / o projections of a SINGLE-TOKEN step run on the library's weight-streaming kernel
    (``spatten_gemv``) instead of torch's GEMM library — at q_len = 1 they are HBM-bound streams that make up four fifths
    of the decode step's bytes (same fp32 accumulation and single rounding as ``nn.Linear``; a different summation order,
    so the last bit can differ).  Multi-token forwards keep torch's GEMMs.

    ``fused_step=True`` (needs ``fuse_qkv`` and ``native_gemv``): the q / k / v projections of a single-token step run INSIDE
    the attention launch (spatten_decode_args_t::qkv_*: every workgroup projects its share of its head's q / k / v while the
    step's K/V stream is already in flight; the values equal ``spatten_gemv``'s bit for bit) — one launch and one
    host call per layer-step — for the EAGER per-call loop, which is host-bound (684 -> 859 tokens/s at Llama-2-7B geometry);
    a step traced by ``DecodeGraph`` / ``auto_graph`` runs the separate launches (2 % faster there, bit-identical).  Applies
    to the plain step on MHA stacks at head_dim 128 in bf16 / f16, batch 1, up to 320 cache rows per split; other steps run
    the separate launches.  The fused launch contains the 256-thread attention team, so this option selects that team for
    the process (``ops.set_decode_team(256)``: the separate steps then equal the fused ones bit for bit; the default
    512-thread team is 4 % faster per attention launch and differs in summation order).

    ``auto_graph=True`` (or a token horizon; True = 64, the reference's max_gen_len — run_spatten_llama.py:61): ``model.forward`` is wrapped so that the reference's per-token loop
    (run_spatten_llama.py:27-35), unchanged, replays one captured HIP graph of the whole patched stack per token
    (spatten_amd/graph.py:auto_graph) — the zero-change form of ``DecodeGraph``.

    ``numerics``: "auto" (default, round 6) — multi-token forwards that materialise NO stash (``prefill_stash=False``) keep
    their logits in fp32 instead of reproducing the reference's two 16-bit roundings per logit (modify_llama.py:111-113): the
    roundings are then unobservable except through the output, which stays inside the stated tolerance (it leaves it only
    where logits are large); forwards that do stash always reproduce them.  "reference": always both roundings.
    "fast": as auto.
    ``head_parallel=HeadParallel(H, Hkv)`` (spatten_amd/parallel.py; one process per GPU): this rank's modules project,
    cache, attend and prune only its H/G heads (column-sharded q/k/v projections; ``past_key_values`` and ``attn_scores``
    hold the local heads) and all-gather the attention outputs [B, q, H/G*d] in front of the full ``o_proj``
    (modify_llama.py:146-163) — the reference's only multi-device story is ``device_map="auto"`` (utils.py:58-63).
    Token pruning needs no communication; ``head_keep`` ranks the heads of ALL ranks (one all-gather of H/G scores).

    SpAtten semantics the reference's Python does not implement (PARITY UNPINNED; spatten_amd/extensions.py):
    ``importance_mode="cascade"`` (cumulative importance = running sum of softmax probabilities, accumulated inside the
    decode launch), ``head_keep`` (cascade head pruning: int or one int per layer), ``pq_threshold`` (progressive
    quantisation of the keys at decode: MSB plane first, LSB refetch below this max-probability), ``pq_profile`` ((key MSB
    bits, value bits) of the planes: (4, 8), (8, 8) or (6, 6) — the quantised VALUE plane and the LSB-only refetch;
    MatrixFetcher.scala:48-51, TestSpAtten.scala:64,83-97,173-176; None = 4-bit MSB plane, V in the model dtype), ``local_v_keep``
    (local V pruning at decode: fraction of the keys whose V row is fetched), ``layer_keep`` (layer-to-layer cascade token
    pruning: one important-token count per layer, non-increasing — the surviving set shrinks through the layers),
    ``token_scope="global"`` (ONE kept token set per layer shared by all heads, ranked by the importance summed over the
    heads — README.md:21, workloads/small.csv:1; head-parallel ranks all-reduce the [layers, L] sums once per prune event;
    default "head" = the reference's per-head top-k)."""
    model_type = model.config.model_type
    patch = next((fn for key, fn in _FAMILIES.items() if key in model_type), None)
    if patch is None:
        raise ValueError(f"got {model_type}")
    k_dim, v_dim = patch(model)
    cache = SpAttenKVCache(start_size=start_size, recent_size=recent_size, important_size=important_size,
                           k_seq_dim=k_dim, v_seq_dim=v_dim, importance_mode=importance_mode, token_scope=token_scope)
    cache.head_parallel = head_parallel
    from .pos_shift.modify_llama import attention_modules

    mods = attention_modules(model)                                # model.modules() order = layer order (:74-77)
    if fused_step and not (fuse_qkv and native_gemv):
        raise ValueError("fused_step needs fuse_qkv=True and native_gemv=True (it reads the stacked weight the way spatten_gemv does)")
    if pq_profile is not None and pq_threshold is None:
        raise ValueError("pq_profile needs pq_threshold")
    prev_team = None
    if fused_step:
        import warnings

        from . import ops
        prev_team = ops.set_decode_team(256)   # the team the fused launch contains: fused and separate steps stay bit-identical
        if prev_team != 256:
            # PROCESS-WIDE: every other model / DecodeGraph of this process changes its summation order with it —
            # graphs captured earlier with the 512-thread team must be re-captured to stay bit-identical with eager steps
            warnings.warn("spatten_amd: fused_step=True selected the 256-thread decode team for the whole process "
                          f"(was {prev_team}); re-capture DecodeGraphs made before this call, or restore it with "
                          "cache.restore_process_options()", RuntimeWarning, stacklevel=2)
    cache._prev_decode_team = prev_team
    extended = (importance_mode == "cascade" or head_keep is not None or pq_threshold is not None or local_v_keep is not None
                or layer_keep is not None)
    for m in mods:
        m.spatten_prefill_stash = bool(prefill_stash)
        m.spatten_assume_causal = bool(assume_causal)
        m._spatten_ext = None
        m._spatten_rope = None
        m.__dict__.pop("_spatten_geom", None)       # geometry cache of the patched forward: re-read on the next call
        m._spatten_qkv = None
        m.__dict__["_spatten_gemv"] = bool(native_gemv)
        m.__dict__["_spatten_fused_step"] = bool(fused_step)
        if numerics not in ("auto", "reference", "fast"):
            raise ValueError("numerics must be 'auto', 'reference' or 'fast'")
        m.__dict__["_spatten_numerics"] = numerics
        m.__dict__.pop("_spatten_hp", None)
        if head_parallel is not None:
            from .pos_shift.modify_llama import shard_attention_projections

            shard_attention_projections(m, head_parallel, fuse=bool(fuse_qkv))
        elif fuse_qkv:
            from .pos_shift.modify_llama import fuse_qkv_projections

            fuse_qkv_projections(m)
    if extended:
        from .extensions import SpattenExtensions

        cache.ext = SpattenExtensions(cache, len(mods), cascade=importance_mode == "cascade", head_keep=head_keep,
                                      pq_threshold=pq_threshold, local_v_keep=local_v_keep, layer_keep=layer_keep,
                                      head_parallel=head_parallel, pq_profile=pq_profile)
        for layer, m in enumerate(mods):
            m._spatten_ext = (cache.ext, layer)
    if auto_graph:
        if cache.ext is not None and not cache.ext.graph_capable():
            raise ValueError("auto_graph: local V pruning together with cascade importance runs its decode step eagerly "
                             "(SpattenExtensions.graph_capable)")
        from .graph import auto_graph as _auto

        _auto(model, horizon=64 if auto_graph is True else int(auto_graph))
    return cache
