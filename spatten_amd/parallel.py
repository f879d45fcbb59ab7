"""Head-parallel partitioning of the hot path across the GPUs of one node (one process per GPU).

Every op on the path is independent per (batch, head): rotation, Q·K, softmax, P·V, the importance row,
the per-head top-k and the KV gather (kv_cache_token_pruning.py:59-69 work row-wise on [H, .]).  Rank r owns
heads [r*H/G, (r+1)*H/G) and the KV planes of those heads; KV never moves.  The only exchange on the path
is the all-gather of the per-rank attention outputs [B, q, H/G*d] -> [B, q, H*d] in front of o_proj (RCCL
``all_gather_into_tensor`` over xGMI; backend "nccl" on ROCm IS RCCL), plus — for head pruning — an
all-gather of H/G fp32 head scores followed by an identical, deterministic top-k on every rank.
Per-head token pruning needs no communication at all; the GLOBAL token scope (one kept set per layer for all heads) needs one
all-reduce of the [layers, L] head-summed importance per prune event (``all_reduce_sum``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


class _Done:
    work = None

    def __init__(self, t):
        self.t = t

    def wait(self):
        return self.t


class _Pending:
    def __init__(self, work, merge):
        self.work, self.merge = work, merge

    def wait(self):
        self.work.wait()
        return self.merge()


class HeadParallel:
    def __init__(self, num_heads: int, num_kv_heads: Optional[int] = None, group=None, rank: Optional[int] = None,
                 world: Optional[int] = None, gather_fn=None, reduce_fn=None):
        """Default: rank / world of the torch.distributed process group.  ``rank`` / ``world`` / ``gather_fn`` given
        explicitly: a partition WITHOUT a process group — ``gather_fn(local [B,q,H/G*d], rank) -> full [B,q,H*d]`` stands
        in for the all-gather (tests run every rank's shard in one process, one after the other)."""
        self.group = group
        self.gather_fn = gather_fn
        self.reduce_fn = reduce_fn          # explicit partitions: reduce_fn(t, rank) -> the sum over ranks (global token scope)
        if rank is not None or world is not None:
            if rank is None or world is None or not (0 <= rank < world) or gather_fn is None:
                raise ValueError("an explicit partition needs rank, world and gather_fn")
            self.world, self.rank = int(world), int(rank)
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        num_kv_heads = num_kv_heads or num_heads
        if num_heads % self.world or num_kv_heads % self.world:
            raise ValueError(f"heads ({num_heads}/{num_kv_heads}) must divide evenly over {self.world} ranks")
        self.num_heads, self.num_kv_heads = num_heads, num_kv_heads
        self.local_heads = num_heads // self.world
        self.local_kv_heads = num_kv_heads // self.world

    def head_range(self, rank: Optional[int] = None) -> Tuple[int, int]:
        r = self.rank if rank is None else rank
        return r * self.local_heads, (r + 1) * self.local_heads

    def kv_head_range(self, rank: Optional[int] = None) -> Tuple[int, int]:
        r = self.rank if rank is None else rank
        return r * self.local_kv_heads, (r + 1) * self.local_kv_heads

    def shard_heads(self, x: torch.Tensor, dim: int = 1, kv: bool = False) -> torch.Tensor:
        lo, hi = self.kv_head_range() if kv else self.head_range()
        return x.narrow(dim, lo, hi - lo)

    def gather_staging(self, batch: int, q_len: int, head_dim: int, dtype, device) -> torch.Tensor:
        """[world, B, q, H/G*d] receive buffer for ``gather_heads``."""
        return torch.empty(self.world, batch, q_len, self.local_heads * head_dim, dtype=dtype, device=device)

    def gather_heads(self, out_local: torch.Tensor, staging: Optional[torch.Tensor] = None,
                     async_op: bool = False):
        """all-gather of [B, q, H/G*d] -> [B, q, H*d] (rank-major = head-major, the reference's
        ``transpose(1,2).reshape`` layout, modify_llama.py:146-147).  Synchronous: returns (full, None).
        ``async_op=True``: returns (None, handle); ``handle.wait()`` yields the merged tensor (the
        head-major merge reads the staging buffer, so it must not run before the collective finished)."""
        B, ql, hd = out_local.shape
        if self.gather_fn is not None:
            full = self.gather_fn(out_local, self.rank)
            return (None, _Done(full)) if async_op else (full, None)
        if getattr(self, "_comm", None) is not None and out_local.is_cuda:
            # the library-owned RCCL communicator (init_native): an ordinary stream operation — the form a captured decode
            # step (spatten_amd/graph.py) can hold; torch's process-group collectives cannot be captured on this stack
            if staging is None:
                staging = torch.empty(self.world, B, ql, hd, dtype=out_local.dtype, device=out_local.device)
            self.allgather_native(out_local.contiguous(), staging)
            full = staging.permute(1, 2, 0, 3).reshape(B, ql, self.world * hd)
            return (None, _Done(full)) if async_op else (full, None)
        if self.world == 1 and not (dist.is_initialized() and staging is not None):
            return (None, _Done(out_local)) if async_op else (out_local, None)
        if staging is None:
            staging = torch.empty(self.world, B, ql, hd, dtype=out_local.dtype, device=out_local.device)
        # concatenation form (output = inputs stacked along dim 0): accepted by RCCL and by gloo alike
        work = dist.all_gather_into_tensor(staging.view(self.world * B, ql, hd), out_local.contiguous(),
                                           group=self.group, async_op=async_op)
        merge = lambda: staging.permute(1, 2, 0, 3).reshape(B, ql, self.world * hd)
        if async_op:
            return None, _Pending(work, merge)
        return merge(), None

    # ---- the plugin's sharded projections (enable_spatten_llm(..., head_parallel=hp)) --------------------------------
    def shard_projection(self, weight: torch.Tensor, bias: Optional[torch.Tensor], head_dim: int, kv: bool = False):
        """Rows of a q / k / v projection that produce this rank's heads: nn.Linear weight [H*d, hidden] (bias [H*d])
        -> ([H/G*d, hidden], [H/G*d] or None) — the column-sharded projection of SURVEY §8e (output features = rows of
        the stored weight).  Contiguous copies: the full matrices can then be dropped by the caller."""
        lo, hi = self.kv_head_range() if kv else self.head_range()
        w = weight.detach()[lo * head_dim:hi * head_dim].contiguous()
        b = None if bias is None else bias.detach()[lo * head_dim:hi * head_dim].contiguous()
        return w, b

    # ---- the library-owned RCCL communicator (include/spatten.h: spatten_comm_*) -----------------------------------
    def init_native(self):
        """Create the C-ABI communicator for this rank's CURRENT device.  The 128-byte id is made on rank 0 and shipped
        through torch.distributed (any backend); with a single rank no process group is needed.  Collective."""
        import ctypes

        from . import _lib
        lib = _lib.load()
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.spatten_comm_unique_id(ident), "spatten_comm_unique_id")
        if self.world > 1:
            box = [ident.raw if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=self.group)
            ident = ctypes.create_string_buffer(box[0], 128)
        comm = ctypes.c_void_p()
        _lib.check(lib.spatten_comm_init(ctypes.byref(comm), self.rank, self.world, ident), "spatten_comm_init")
        self._comm = comm
        return self

    def allgather_native(self, send: torch.Tensor, recv: torch.Tensor):
        """recv[r * n : (r + 1) * n] = rank r's ``send`` (n = send.numel()), on the current stream: an ordinary stream
        operation, so it may be captured into a HIP graph with the attention launches around it."""
        from . import _lib
        if getattr(self, "_comm", None) is None:
            raise RuntimeError("init_native() first")
        if not (send.is_contiguous() and recv.is_contiguous()) or recv.numel() != self.world * send.numel() or recv.dtype != send.dtype:
            raise ValueError("send / recv must be contiguous, recv = world x send")
        rc = _lib.load().spatten_allgather(self._comm, send.data_ptr(), recv.data_ptr(), send.numel() * send.element_size(),
                                           torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "spatten_allgather")
        return recv

    def native_info(self):
        """(ranks, rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        import ctypes

        from . import _lib
        if getattr(self, "_comm", None) is None:
            raise RuntimeError("init_native() first")
        n, r = ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.check(_lib.load().spatten_comm_info(self._comm, ctypes.byref(n), ctypes.byref(r)), "spatten_comm_info")
        return n.value, r.value

    def close_native(self):
        from . import _lib
        if getattr(self, "_comm", None) is not None:
            _lib.load().spatten_comm_destroy(self._comm)
            self._comm = None

    # ---- the peer-store all-gather (include/spatten.h: spatten_peer_*): direct writes into every peer's receive window ----
    def init_peer_store(self, max_bytes_per_rank: int, exchange_handles=None):
        """Create this rank's receive window (on its CURRENT device), exchange the 64-byte hipIpc handles — through
        torch.distributed (any backend), or through ``exchange_handles(my_handle: bytes) -> list[bytes]`` (rank-major) — and
        map every peer's window.  Collective.  With one rank nothing is exchanged (loopback)."""
        import ctypes

        from . import _lib
        lib = _lib.load()
        peer = ctypes.c_void_p()
        mine = ctypes.create_string_buffer(64)
        rc = lib.spatten_peer_create(ctypes.byref(peer), self.rank, self.world, int(max_bytes_per_rank), mine)
        if self.world == 1:
            _lib.check(rc, "spatten_peer_create")
        # A rank whose window could not be created (no fine-grained memory, no IPC handle) still takes part in the exchange — with
        # an all-zero handle — so that EVERY rank sees the failure and falls back together; raising before the collective would
        # leave the others blocked in it.
        raw = mine.raw if rc == 0 else bytes(64)
        handles = list(exchange_handles(raw)) if exchange_handles is not None else self.gather_handles(raw)
        if len(handles) != self.world or any(len(h) != 64 for h in handles):
            if rc == 0:
                lib.spatten_peer_destroy(peer)
            raise ValueError("peer-store: the handle exchange must return one 64-byte handle per rank, rank-major")
        failed = [r for r, h in enumerate(handles) if bytes(h) == bytes(64)] if self.world > 1 else []
        if failed:
            if rc == 0:
                lib.spatten_peer_destroy(peer)
            _lib.check(rc, "spatten_peer_create")            # this rank's own error, if it is one of them
            raise NotImplementedError(f"peer-store: receive window unavailable on rank(s) {failed}; every rank keeps RCCL")
        blob = ctypes.create_string_buffer(b"".join(handles), 64 * self.world)
        try:
            _lib.check(lib.spatten_peer_connect(peer, blob), "spatten_peer_connect")
        except Exception:
            lib.spatten_peer_destroy(peer)
            raise
        self._peer = peer
        return peer

    def gather_handles(self, mine: bytes):
        """Every rank's 64-byte window handle, rank-major, through torch.distributed (any backend; objects, not tensors:
        the handles are host bytes)."""
        if self.world == 1:
            return [mine]
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=self.group)
        return handles

    def allgather_peer(self, send: torch.Tensor, recv: torch.Tensor):
        """recv[r * n : (r + 1) * n] = rank r's ``send`` through the peer windows — one launch on the current stream."""
        from . import _lib
        if getattr(self, "_peer", None) is None:
            raise RuntimeError("init_peer_store() first")
        if not (send.is_contiguous() and recv.is_contiguous()) or recv.numel() != self.world * send.numel() or recv.dtype != send.dtype:
            raise ValueError("send / recv must be contiguous, recv = world x send")
        rc = _lib.load().spatten_peer_allgather(self._peer, send.data_ptr(), recv.data_ptr(), send.numel() * send.element_size(),
                                                torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "spatten_peer_allgather")
        return recv

    def peer_status(self):
        """Synchronises the current stream; raises SpattenDeviceTimeout if a peer's slice did not arrive in an earlier call."""
        from . import _lib
        if getattr(self, "_peer", None) is not None:
            _lib.check(_lib.load().spatten_peer_status(self._peer, torch.cuda.current_stream().cuda_stream), "spatten_peer_allgather")

    def close_peer_store(self):
        from . import _lib
        if getattr(self, "_peer", None) is not None:
            _lib.load().spatten_peer_destroy(self._peer)
            self._peer = None

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """Sum over the ranks, on every rank (global token pruning: the [layers, L] importance summed over the heads of all
        ranks, once per prune event — kv_cache_token_pruning.SpAttenKVCache._global_scores)."""
        if self.reduce_fn is not None:
            return self.reduce_fn(t, self.rank)
        if self.world == 1:
            return t
        if self.gather_fn is not None:
            raise ValueError("an explicit partition needs reduce_fn for the global token scope")
        t = t.contiguous()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def surviving_local_heads(self, layer_scores: torch.Tensor, keep) -> list:
        """Cascade head pruning over the heads of ALL ranks with static ownership ("keep static ownership, pruned
        heads simply skip"; oracle: head_prune_cascade).  ``layer_scores`` fp32 [layers, H/G] = sum |O_h| of this rank's heads
        per layer; ``keep`` = heads that survive (int, or one per layer, non-increasing).  One all-gather of the [layers, H/G]
        block; every rank then runs the same deterministic rule — cumulative score over the layers, a head pruned in a
        layer stays pruned, ties keep the lowest head id — and returns, per layer, the int32 LOCAL ids (ascending) of its
        own surviving heads: what its attention launch is given as ``head_ids``.  The lists are uneven across ranks and may
        be empty (that rank then launches nothing for the layer; its output slice stays zero)."""
        L, Hl = layer_scores.shape
        sc = layer_scores.to(torch.float32).contiguous()
        if self.gather_fn is not None:
            sc = self.gather_fn(sc.reshape(1, L, Hl).transpose(0, 1).contiguous(), self.rank).reshape(L, -1)
        elif self.world > 1:
            allsc = torch.empty(self.world * L, Hl, dtype=torch.float32, device=sc.device)   # concatenation form (RCCL and gloo)
            dist.all_gather_into_tensor(allsc, sc, group=self.group)
            sc = allsc.view(self.world, L, Hl).permute(1, 0, 2).reshape(L, self.world * Hl)
        sc = sc.double().cpu()
        H = sc.shape[1]
        keeps = [int(keep)] * L if isinstance(keep, int) else [int(k) for k in keep]
        cum = torch.zeros(H, dtype=torch.float64)
        alive = torch.ones(H, dtype=torch.bool)
        lo = self.rank * Hl
        out = []
        for l in range(L):
            cum += sc[l]
            ranked = torch.where(alive, cum, torch.full_like(cum, -float("inf")))
            order = torch.sort(ranked, descending=True, stable=True).indices
            ids = order[: min(keeps[l], int(alive.sum()))].sort().values
            alive = torch.zeros_like(alive)
            alive[ids] = True
            mine = ids[(ids >= lo) & (ids < lo + Hl)] - lo
            out.append(mine.to(torch.int32).to(layer_scores.device))
        return out

    def gather_head_scores(self, local_scores: torch.Tensor) -> torch.Tensor:
        """[H/G] fp32 -> [H] on every rank (head pruning: every rank then runs the same top-k)."""
        if self.gather_fn is not None:
            return self.gather_fn(local_scores.reshape(1, 1, -1), self.rank).reshape(-1)
        if self.world == 1:
            return local_scores
        full = torch.empty(self.world * local_scores.numel(), dtype=local_scores.dtype, device=local_scores.device)
        dist.all_gather_into_tensor(full, local_scores.contiguous(), group=self.group)
        return full
