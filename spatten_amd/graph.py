"""DecodeGraph — one captured HIP graph of the WHOLE decode step, replayed for every token of a turn.

The reference's caller is a per-token Python loop (run_spatten_llama.py:27-35: ``model(pred_token, past_key_values)``
for up to 63 tokens per turn); through the patched forward that loop is host-bound (~30 launches x 32 layers of Python,
ctypes and torch dispatch per token for ~0.4 ms of GPU work).  What kept the step from being captured once were the
host integers that change every token: the cache length (it sized the stash, the K/V views, the split-N grid) and the
query position.  With the device-resident step state (include/spatten.h, ABI 3; ops.StepState) the attention launch of
every layer reads both from device memory; the stash row lives in the KV slab at capacity; the K/V views HF sees are
rebuilt on demand.  So:

    graph = DecodeGraph(step_fn, past_key_values)          # step_fn(past, *inputs) -> (new_past, outputs)
    for t in range(n):
        logits = graph.step(token)                         # call 1: eager (warm-up), call 2: capture + replay, then replays
        token = logits.argmax(-1)                          # outputs are STATIC tensors, overwritten by the next step
    past_key_values = graph.past_key_values                # fresh [K, V] views at the current length; also refreshes
                                                           # ``module.attn_scores`` of every patched module

``step_fn`` is whatever runs one token through the patched model, e.g.
``lambda past, tok: (lambda o: (o.past_key_values, o.logits))(model(tok, past_key_values=past, use_cache=True))``.
Requirements: batch and shapes fixed; every mode of the plugin is captured (plain, cascade importance, head pruning,
progressive quantisation, the layer cascade, local V pruning); the HF mask / position_ids of the step are the ones
transformers 4.33 builds (zeros / the past length) — they are not read (a non-zero mask is refused while tracing).

Numerics: every step — the warm-up, and a plain eager step through the patched forward on slabs of the same capacity —
lays its split-N decomposition out for the slab capacity, so graph replays and eager steps agree bit for bit
(tests/test_gpu_graph_decode.py).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

from . import kv_slab, ops

__all__ = ["DecodeGraph", "GraphCaptureError", "auto_graph"]


class GraphCaptureError(RuntimeError):
    """The step could not be CAPTURED (``step_fn`` synchronises with the host, runs a non-capturable collective, …).
    Raised by ``DecodeGraph.step`` only for errors inside capture_begin / capture_end: a capture records, it does not
    execute, so nothing of that step ran on the device, and the host-side slab state has been put back — the caller may
    run the step eagerly instead.  Errors of the constructor, of a re-bind and of the eager warm-up step (which DOES run
    on the device) are raised as they are."""


_capture_streams: dict = {}


def _capture_stream(device) -> torch.cuda.Stream:
    """ONE side stream per device for every DecodeGraph of the process: the decode workspaces and the caching allocator's
    pools are per stream, so a stream per graph (a graph per turn) grew device memory turn after turn (a 60-turn run:
    +38 MiB per turn until torch's 32 pooled streams repeated) and left the workspace cache evicting buffers of live graphs."""
    key = torch.device(device)
    key = (key.type, torch.cuda.current_device() if key.index is None else key.index)
    st = _capture_streams.get(key)
    if st is None:
        st = _capture_streams[key] = torch.cuda.Stream(device=torch.device(*key))
    return st


class DecodeGraph:
    def __init__(self, step_fn: Callable, past_key_values: Sequence, horizon: int = 128):
        """``past_key_values``: the per-layer ``(K, V)`` pairs the patched forward (or ``apply_token_pruning``) returned;
        ``horizon``: tokens the slabs are sized for up front (more are possible: the graph is re-captured on new slabs)."""
        if past_key_values is None or len(past_key_values) == 0:
            raise ValueError("DecodeGraph needs the past_key_values of a prefilled / pruned cache")
        self.step_fn = step_fn
        self.horizon = int(horizon)
        # the layers' caches may differ in length (layer-to-layer cascade pruning: the surviving set shrinks through the
        # layers); every step appends one row to each, so layer i stays at length + offsets[i] — one step state per offset
        self.length = int(past_key_values[0][0].shape[2])
        self.offsets = [int(kv[0].shape[2]) - self.length for kv in past_key_values]
        self.stream = _capture_stream(past_key_values[0][0].device)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.states: dict = {}                  # cache-length offset (vs layer 0) -> ops.StepState
        self.touched: List[tuple] = []          # (module, slab) pairs of the last traced step, in call order
        self.workspaces: List = []              # every scratch buffer a traced launch used: the captured graph holds raw
                                                # pointers into them (ops.set_ws_pins), so it keeps them alive
        self.static_in: Optional[List[torch.Tensor]] = None
        self.static_out = None
        self.n_replays = 0
        self.n_binds = 0                        # bindings so far (> 1: the graph outgrew its slabs and was re-captured)
        self.steps_traced = 0                   # device-length steps since the state was set (= state word 2 before a step)
        self._ext = None                        # SpattenExtensions of the patched modules, once a traced step met them
        self._past = None
        self._bind(past_key_values)

    # ------------------------------------------------------------------------------------------------
    def _bind(self, past_key_values):
        """Size the slabs for the next ``horizon`` tokens, zero their tails, (re)create the step states."""
        past = kv_slab.reserve(past_key_values, [self.length + o + self.horizon for o in self.offsets])
        self._past = past
        slabs = [kv_slab.slab_of(kv[0]) for kv in past]
        self._offset_of = {id(sl): o for sl, o in zip(slabs, self.offsets)}
        self.bound = min(sl.capacity - o for sl, o in zip(slabs, self.offsets))     # in units of layer 0's length
        self.graph = None
        self.states = {}
        self.touched = []
        self.steps_traced = 0
        self.bind_id = object()                 # identity of this binding: the extensions restart their buffers with it
        self.n_binds += 1

    def length_of(self, slab) -> int:
        return self.length + self._offset_of.get(id(slab), 0)

    def state_for(self, slab, cos, sin) -> ops.StepState:
        """Called by the patched forward while a step is traced: the step state of the layers with this cache length."""
        off = self._offset_of.get(id(slab))
        if off is None:
            raise RuntimeError("DecodeGraph: a layer ran on a cache this graph was not bound to")
        st = self.states.get(off)
        if st is None:
            st = self.states[off] = ops.StepState(cos, sin)
            st.set(self.length + off, self.length + off - 1)
            st.advance(1)                     # the traced step is already under way: its advance, issued late
        elif st.cos.data_ptr() != cos.data_ptr():
            raise RuntimeError("DecodeGraph: layers of one cache length rotate with different rotary tables")
        if slab.capacity < self.bound + off:
            raise RuntimeError("DecodeGraph: a layer's slab is smaller than the graph's bound")
        return st

    def _trace(self, inputs):
        """Run step_fn once in device-length mode on the current stream (eagerly, or under capture)."""
        self.touched = []
        prev, prev_pins = kv_slab.set_graph_ctx(self), ops.set_ws_pins(self.workspaces)
        try:
            for st in self.states.values():
                st.advance(1)
            new_past, out = self.step_fn(self._past, *inputs)
        finally:
            kv_slab.set_graph_ctx(prev)
            ops.set_ws_pins(prev_pins)
        if not self.touched:
            raise RuntimeError("DecodeGraph: step_fn did not run a single-token step through the patched forward")
        self._past = new_past
        return out

    # ------------------------------------------------------------------------------------------------
    def step(self, *inputs):
        """One token.  Returns step_fn's outputs (static tensors from the second call on)."""
        if self.length + 1 > self.bound:                   # out of room: larger slabs, new graph
            self._bind(self.past_key_values)
        cur = torch.cuda.current_stream()
        if self.graph is None and not self.states:
            # call 1 on these slabs: eager, on the capture stream (creates the per-stream workspaces, warms the allocator
            # and the GEMM heuristics) — the same device-length kernels the graph will replay
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                out = self._trace(inputs)
            cur.wait_stream(self.stream)
            self.length += 1
            self.steps_traced += 1
            return out
        if self.graph is None:
            self.static_in = [x.clone() if isinstance(x, torch.Tensor) else x for x in inputs]
            self.stream.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            # capture_begin / capture_end directly: the torch.cuda.graph context manager also runs gc.collect() and
            # torch.cuda.empty_cache() on entry — after a prune the previous turn's slabs (1.6 GB at Llama-2-7B) sit in the
            # caching allocator, and handing them back to the driver cost 20 ms per turn here, plus the hipMallocs of the
            # next prune.
            # host-side state a traced step moves (the patched forward sets slab.length / rot_len layer by layer): put back
            # if the capture fails, so that the caller's views and an eager re-run of this step see the pre-step cache
            slabs = [kv_slab.slab_of(kv[0]) for kv in self._past]
            saved = [(sl.length, sl.rot_len, sl.pq_len) for sl in slabs]
            past0, touched0 = self._past, self.touched
            try:
                with torch.cuda.stream(self.stream):
                    self.stream.synchronize()
                    g.capture_begin()
                    try:
                        self.static_out = self._trace(self.static_in)
                    finally:
                        g.capture_end()
            except Exception as e:      # noqa: BLE001 - whatever broke the capture: nothing of this step ran
                for sl, (n, r, pq_n) in zip(slabs, saved):
                    sl.length, sl.rot_len, sl.pq_len = n, r, pq_n
                self._past, self.touched, self.static_in, self.static_out = past0, touched0, None, None
                raise GraphCaptureError(f"{type(e).__name__}: {e}") from e
            finally:
                cur.wait_stream(self.stream)
            self.graph = g
        else:
            for dst, src in zip(self.static_in, inputs):
                if isinstance(dst, torch.Tensor) and dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        self.graph.replay()
        self.n_replays += 1
        self.length += 1
        self.steps_traced += 1
        return self.static_out

    def sync_scores(self):
        """Point every patched module's ``attn_scores`` at its stash row of the last step (modify_llama.py:116-119) without
        rebuilding the cache views — what a caller that reads only ``m.attn_scores`` between steps needs."""
        synced = set()
        for module, slab, ext in self.touched:
            if ext is None:
                if slab.stash is not None:
                    object.__setattr__(module, "attn_scores", slab.stash[:, :, None, :self.length_of(slab)])
                continue
            exts, layer = ext
            if id(exts) not in synced:
                exts.graph_sync(self.steps_traced, [self.length + o for o in self.offsets])
                synced.add(id(exts))
            st = exts.layers[layer]
            cur = ((self.steps_traced - 1) & 1) if (exts.cascade and self.steps_traced > 0) else 0
            object.__setattr__(module, "attn_scores", st.stash[cur][:, :, None, :self.length_of(slab)])

    @property
    def past_key_values(self):
        """``[K, V]`` views of every layer at the CURRENT length (the tuples a replayed step cannot update), in layer
        order; also points every patched module's ``attn_scores`` at its stash row of the last step (:116-119)."""
        slabs = [kv_slab.slab_of(kv[0]) for kv in self._past]
        out = []
        for slab, off in zip(slabs, self.offsets):
            slab.length = slab.rot_len = self.length + off
            if slab.pq is not None and slab.pq_len >= slab.length - self.steps_traced:
                slab.pq_len = slab.length          # progressive quantisation: every replayed step packed its own row
            k, v = slab.views()
            out.append([k, v])
        self.sync_scores()
        self._past = out
        return out


class _LazyPast(list):
    """``past_key_values`` of a replayed step: the per-layer ``[K, V]`` views are built when somebody looks (the next
    captured step does not need them; ``apply_token_pruning`` and a multi-token forward do)."""

    def __init__(self, graph: DecodeGraph):
        super().__init__()
        self._graph, self._length, self._done = graph, graph.length, False

    def _fill(self):
        if not self._done:
            if self._graph.length != self._length:
                raise RuntimeError("stale past_key_values: the model has decoded further since this object was returned")
            super().extend(self._graph.past_key_values)
            self._done = True

    def __getitem__(self, i):
        self._fill()
        return super().__getitem__(i)

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        self._fill()
        return super().__len__()


def auto_graph(model, horizon: int = 64):
    """Zero-change mode of the drop-in: wrap ``model.forward`` so that the reference's own per-token loop
    (run_spatten_llama.py:27-35: ``model(input_ids=tok, past_key_values=past, use_cache=True)``) runs every single-token call
    as a replay of ONE captured graph of the whole patched stack.  A call is taken over when it has exactly ``input_ids``
    [B, 1] (or [B, 1, hidden] for a stack fed with embeddings), ``past_key_values`` and ``use_cache=True`` (keywords) under ``torch.no_grad()``; the first such call on a cache
    (after a prefill or a prune) runs eagerly, the second is captured, later ones replay.  Everything else — prefill, calls
    with masks / positions / embeddings — goes to the original forward.  The returned ``logits`` are a static buffer the
    next call overwrites; ``past_key_values`` is materialised when it is indexed; ``m.attn_scores`` is current after every
    call.  Results equal the per-call forward's bit for bit when its slabs have the capacity the graph reserves (``horizon``
    rows beyond the cache, rounded to 128) — the split layout of a decode step follows the slab capacity — and to rounding
    otherwise.  ``horizon``: rows reserved beyond the cache when a graph is bound (the captured step reads its slabs up to their
    capacity: a larger horizon streams more discarded rows per token; a generation that outgrows it re-binds and re-captures,
    a few ms).  Returns the wrapped model (the same object)."""
    orig = model.forward
    state = {"graph": None, "lazy": None, "proto": None}

    def step_fn(past, ids):
        out = orig(input_ids=ids, past_key_values=past, use_cache=True)
        state["proto"] = out
        if isinstance(out, tuple):
            return out[1], out[0]
        return out.past_key_values, out.logits

    def forward(*args, **kw):
        ids, past = kw.get("input_ids"), kw.get("past_key_values")
        others = [k for k, v in kw.items() if k not in ("input_ids", "past_key_values", "use_cache") and v is not None]
        take = (not args and not others and isinstance(ids, torch.Tensor) and ids.dim() >= 2 and ids.shape[1] == 1 and ids.is_cuda
                and past is not None and bool(kw.get("use_cache")) and not torch.is_grad_enabled())
        if not take:
            state["graph"] = state["lazy"] = None
            return orig(*args, **kw)
        if state.get("disabled"):
            return orig(*args, **kw)
        # errors of the constructor / a re-bind / the eager warm-up step (shape errors, OOM, genuine bugs — the warm-up DOES
        # run on the device) propagate; only a failed CAPTURE falls back
        if state["graph"] is None or past is not state["lazy"]:
            state["graph"] = DecodeGraph(step_fn, list(past), horizon=horizon)
        graph = state["graph"]
        try:
            logits = graph.step(ids)
        except GraphCaptureError as e:    # e.g. a forward that synchronises (.item(), .cpu()) cannot be captured
            # nothing of the failed step ran on the device (a capture records, it does not execute) and DecodeGraph.step has
            # put the slabs' host-side lengths back: the cache at the pre-step length goes to the original forward, which
            # takes this call — and every later one
            import warnings

            warnings.warn(f"spatten auto_graph: the single-token step could not be captured ({e}); "
                          "falling back to the per-call forward", RuntimeWarning)
            state["disabled"] = True
            kw = dict(kw, past_key_values=graph.past_key_values)
            state["graph"] = state["lazy"] = None
            return orig(*args, **kw)
        graph.sync_scores()
        lazy = state["lazy"] = _LazyPast(graph)
        proto = state["proto"]
        if isinstance(proto, tuple):
            return (logits, lazy) + tuple(proto[2:])
        try:
            return type(proto)(logits=logits, past_key_values=lazy)
        except TypeError:
            proto.logits, proto.past_key_values = logits, lazy
            return proto

    model.forward = forward
    model._spatten_auto_graph = state
    return model
