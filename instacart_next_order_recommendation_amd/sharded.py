"""Row-sharded catalog search across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank r owns catalog
rows [bounds[r], bounds[r+1]) as a DeviceIndex with row_offset = bounds[r], encodes its slice of
the query batch, then:

    all_gather(query embeddings)          [Q/W, 384] fp32 per rank  ->  [Q, 384]
    local fused score + top-k             every query against the local shard -> keys [Q, k]
    all_gather(partial keys)              [Q, k] u64 per rank       ->  [W, Q, k]
    k-way merge                           -> global top-k, identical to the unsharded result

The union of per-shard top-k lists contains the global top-k and the (score desc, row asc)
order is total, so the merged result is bit-identical to a single-GPU search.

The reference has no distributed code at all; this is new design.

Product path: `NativeComm` (an RCCL communicator created by libicrec: icrec_comm_init) +
`ShardedSearch(..., comm=NativeComm)`: ONE C call, icrec_search_sharded, runs both all-gathers, the
shard-local search and the merge on the caller's stream — torch.distributed is used only to hand the
128-byte rendezvous id to the other ranks.  Without a `comm` the class falls back to collectives issued
through torch.distributed on the same C kernels; that form is backend-agnostic so that the row arithmetic
can be exercised with gloo on CPU in tests (with the oracle standing in for the kernels).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Protocol, Sequence

import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int) -> list[int]:
    """Even row split; the first n_rows % world shards get one extra row."""
    q, r = divmod(n_rows, world)
    out = [0]
    for i in range(world):
        out.append(out[-1] + q + (1 if i < r else 0))
    return out


def exclusions_to_shard_csr(off_all: torch.Tensor, rows_all: torch.Tensor, row_lo: int, row_hi: int):
    """The device-side step of the exclusion exchange on its own (icrec_exclusions_to_shard_csr): the rank-major
    gathered buffers `off_all` int32 [world, n_local + 1] and `rows_all` int32 [world, excl_cap] (GLOBAL rows) ->
    (csr_off int32 [world * n_local + 1], csr_idx int32 [world * excl_cap]) of LOCAL rows for the shard
    [row_lo, row_hi), as icrec_search / icrec_search_partial take them."""
    from . import _native

    if off_all.dtype != torch.int32 or rows_all.dtype != torch.int32 or off_all.dim() != 2 or rows_all.dim() != 2 \
            or off_all.shape[0] != rows_all.shape[0] or not off_all.is_cuda or rows_all.device != off_all.device:
        raise ValueError("off_all int32 [world, n_local + 1] and rows_all int32 [world, excl_cap] on one HIP device")
    dev = off_all.device
    world, n_local, cap = int(off_all.shape[0]), int(off_all.shape[1]) - 1, int(rows_all.shape[1])
    L = _native.lib()
    ws = torch.empty(int(L.icrec_exclusions_to_shard_csr_workspace_bytes(world, n_local)), dtype=torch.uint8, device=dev)
    csr_off = torch.empty(world * n_local + 1, dtype=torch.int32, device=dev)
    csr_idx = torch.zeros(world * cap, dtype=torch.int32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _native.check(L.icrec_exclusions_to_shard_csr(P(off_all.contiguous()), P(rows_all.contiguous()), world, n_local, cap,
                                                  int(row_lo), int(row_hi), P(csr_off), P(csr_idx), P(ws), ws.numel(),
                                                  dev.index, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                  "icrec_exclusions_to_shard_csr")
    return csr_off, csr_idx


class ShardBackend(Protocol):
    def search_partial(self, q: torch.Tensor, k: int, exclude: Optional[Sequence[Iterable[int]]]) -> torch.Tensor: ...
    def merge(self, keys: torch.Tensor, k: int) -> tuple[torch.Tensor, torch.Tensor]: ...


class HipShardBackend:
    """The product backend: libicrec kernels on this rank's GPU."""

    def __init__(self, shard_rows, row_offset: int, device, storage: str = "f32"):
        from .search import DeviceIndex

        self.index = DeviceIndex(shard_rows, device, row_offset=row_offset, storage=storage)

    def search_partial(self, q, k, exclude):
        return self.index.search_partial(q, k, exclude)

    def merge(self, keys, k):
        from .search import merge_topk

        return merge_topk(keys, k)


class NativeComm:
    """An RCCL communicator owned by libicrec (include/icrec.h: icrec_comm_*): one per process/GPU."""

    def __init__(self, rank: int, world: int, device, unique_id: Optional[bytes] = None):
        from . import _native

        dev = torch.device(device)
        if dev.type != "cuda":
            raise _native.IcrecError("NativeComm needs a CUDA/HIP device")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.rank, self.world = int(rank), int(world)
        if unique_id is not None and len(unique_id) != _native.COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {_native.COMM_ID_BYTES} bytes")
        buf = C.create_string_buffer(unique_id, _native.COMM_ID_BYTES) if unique_id is not None else None
        h = C.c_void_p()
        _native.check(_native.lib().icrec_comm_init(buf, self.rank, self.world, self.device.index, C.byref(h)),
                      "icrec_comm_init")
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        """ncclGetUniqueId (call on ONE rank, ship the bytes to the others)."""
        from . import _native

        buf = C.create_string_buffer(_native.COMM_ID_BYTES)
        _native.check(_native.lib().icrec_comm_unique_id(buf), "icrec_comm_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, device, group=None) -> "NativeComm":
        """Bootstrap over an initialised torch.distributed group: rank 0 draws the id, a broadcast delivers it (the
        only use of torch.distributed on the product path).

        Every rank issues the SAME collective sequence whatever fails locally: (1) each rank loads librccl by
        drawing an id of its own (local, no communication; only rank 0's is used); (2) rank 0 broadcasts
        [status byte | 128-byte id]; (3) an all_reduce(MIN) of "librccl loaded and rank 0's id is valid" — when it
        is 0 EVERY rank raises IcrecError and nobody enters ncclCommInitRank; (4) the collective
        ncclCommInitRank.  A rank that dies inside (4) leaves the others waiting there: that is RCCL's contract and
        cannot be repaired from outside.  The world > 1 RCCL path has run on a one-rank communicator only (no
        multi-GPU box is available to the build); bench.py verifies its first real run against an unsharded search."""
        from . import _native

        if not dist.is_initialized():
            return cls(0, 1, device, None)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        if world == 1:
            return cls(0, 1, device, None)
        on_gpu = dist.get_backend(group) == "nccl"
        where = torch.device(device) if on_gpu else torch.device("cpu")
        my_id, err = None, ""
        try:
            my_id = cls.unique_id()  # dlopen(librccl) + ncclGetUniqueId, on every rank
        except Exception as exc:  # noqa: BLE001 - reported through the collective below, never by skipping one
            err = f"{type(exc).__name__}: {exc}"
        t = torch.zeros(1 + _native.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0 and my_id is not None:
            t[0] = 1
            t[1:] = torch.frombuffer(bytearray(my_id), dtype=torch.uint8)
        t = t.to(where)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        t = t.cpu()
        ok = torch.tensor([1 if (my_id is not None and int(t[0]) == 1) else 0], dtype=torch.int32, device=where)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) != 1:
            raise _native.IcrecError("RCCL communicator not created: " + (err or (
                "rank 0 could not draw the rendezvous id" if int(t[0]) != 1 else "librccl failed to load on another rank")))
        return cls(rank, world, device, bytes(t[1:].numpy().tobytes()))

    def close(self) -> None:
        if getattr(self, "_h", None):
            from . import _native

            _native.lib().icrec_comm_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class ShardedSearch:
    """Collective top-k over a row-sharded catalog.  Every rank calls `search` with ITS slice of
    the query embeddings (equal slice sizes on all ranks); every rank gets the full result."""

    def __init__(self, backend: ShardBackend, row_lo: int, row_hi: int, group=None,
                 comm: Optional[NativeComm] = None):
        self.backend = backend
        self.row_lo, self.row_hi = int(row_lo), int(row_hi)
        self.group = group
        self.comm = comm
        if comm is not None:
            if not isinstance(backend, HipShardBackend):
                raise TypeError("a NativeComm drives the HIP backend only")
            self.world = comm.world
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._ws_by_stream: dict[int, torch.Tensor] = {}

    def _workspace(self, need: int, device) -> torch.Tensor:
        """One scratch block per stream (see DeviceIndex._workspace)."""
        key = int(torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws_by_stream.get(key)
        if ws is None or ws.numel() < need:
            self._ws_by_stream.pop(key, None)
            ws = self._ws_by_stream[key] = torch.empty(need, dtype=torch.uint8, device=device)
        return ws

    def _search_native(self, q_local: torch.Tensor, k: int, exclude_global):
        """icrec_search_sharded: all-gather, shard-local search, all-gather, merge — one C call."""
        from . import _native
        from .search import exclusion_csr

        ix = self.backend.index
        q = q_local.to(device=ix.device, dtype=torch.float32).contiguous()
        n_local = int(q.shape[0])
        Q = n_local * self.world
        ei, eo = exclusion_csr(self._local_exclusions(exclude_global), Q, ix.device) if exclude_global is not None \
            else (None, None)
        L = _native.lib()
        need = int(L.icrec_search_sharded_workspace_bytes(ix._h, self.comm._h, n_local, k))
        if need == 0:
            raise _native.IcrecError(f"bad sharded search shape: n_local={n_local}, k={k}")
        ws = self._workspace(need, ix.device)
        idx = torch.empty((Q, k), dtype=torch.int64, device=ix.device)
        sc = torch.empty((Q, k), dtype=torch.float32, device=ix.device)
        P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
        st = C.c_void_p(torch.cuda.current_stream(ix.device).cuda_stream)
        _native.check(L.icrec_search_sharded(ix._h, self.comm._h, P(q), n_local, k, P(ei), P(eo), P(idx), P(sc),
                                             P(ws), ws.numel(), st), "icrec_search_sharded")
        return idx, sc

    DEFAULT_EXCL_PER_QUERY = 128  # default id capacity per local query of the exclusion exchange (excl_cap = n_local x this)

    def _excl_cap(self, n_local: int, excl_cap: Optional[int]) -> int:
        return int(excl_cap) if excl_cap else n_local * self.DEFAULT_EXCL_PER_QUERY

    @staticmethod
    def _local_csr(exclude_local, n_local: int, cap: int):
        """This rank's per-query GLOBAL rows -> (rows int32[cap] zero-padded, off int32[n_local+1]) numpy arrays."""
        import numpy as np

        if len(exclude_local) != n_local:
            raise ValueError(f"exclude_local has {len(exclude_local)} entries for {n_local} local queries")
        off = np.zeros(n_local + 1, np.int32)
        flat: list[int] = []
        for i, e in enumerate(exclude_local):
            flat.extend(sorted(set(int(v) for v in e)))
            off[i + 1] = len(flat)
        if len(flat) > cap:
            raise ValueError(f"{len(flat)} excluded rows on this rank exceed excl_cap={cap} (the same constant on every rank)")
        rows = np.zeros(cap, np.int32)
        rows[:len(flat)] = flat
        return rows, off

    def _search_native_local_excl(self, q_local: torch.Tensor, k: int, exclude_local, excl_cap: Optional[int]):
        """icrec_search_sharded_excl: every rank hands in the exclusions of ITS queries (global rows); the library
        exchanges them (two more all-gathers) and applies each list on the shard that holds the rows."""
        from . import _native

        ix = self.backend.index
        q = q_local.to(device=ix.device, dtype=torch.float32).contiguous()
        n_local = int(q.shape[0])
        Q = n_local * self.world
        cap = self._excl_cap(n_local, excl_cap)
        rows_h, off_h = self._local_csr(exclude_local, n_local, cap)
        rows = torch.from_numpy(rows_h).to(ix.device)
        off = torch.from_numpy(off_h).to(ix.device)
        L = _native.lib()
        need = int(L.icrec_search_sharded_excl_workspace_bytes(ix._h, self.comm._h, n_local, k, cap))
        if need == 0:
            raise _native.IcrecError(f"bad sharded search shape: n_local={n_local}, k={k}, excl_cap={cap}")
        ws = self._workspace(need, ix.device)
        idx = torch.empty((Q, k), dtype=torch.int64, device=ix.device)
        sc = torch.empty((Q, k), dtype=torch.float32, device=ix.device)
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        st = C.c_void_p(torch.cuda.current_stream(ix.device).cuda_stream)
        _native.check(L.icrec_search_sharded_excl(ix._h, self.comm._h, P(q), n_local, k, P(rows), P(off), cap, P(idx), P(sc),
                                                  P(ws), ws.numel(), st), "icrec_search_sharded_excl")
        return idx, sc

    def _gather_local_exclusions(self, exclude_local, n_local: int, excl_cap: Optional[int]):
        """torch.distributed form of the same exchange: -> per gathered query (rank-major) the global rows."""
        cap = self._excl_cap(n_local, excl_cap)
        rows_h, off_h = self._local_csr(exclude_local, n_local, cap)
        if self.world == 1:
            rows_all, off_all = rows_h[None], off_h[None]
        else:
            rows_all = self._all_gather_rows(torch.from_numpy(rows_h)[None]).numpy()
            off_all = self._all_gather_rows(torch.from_numpy(off_h)[None]).numpy()
        return [rows_all[r, off_all[r, i]:off_all[r, i + 1]].tolist() for r in range(self.world) for i in range(n_local)]

    def _local_exclusions(self, exclude_global):
        if exclude_global is None:
            return None
        return [[r - self.row_lo for r in e if self.row_lo <= r < self.row_hi] for e in exclude_global]

    def _all_gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """Rank-major concatenation of equally shaped [n, c] tensors.  With the gloo backend (CPU
        rehearsals / tests) device tensors are staged through the host; nccl (= RCCL) gathers in HBM."""
        out = torch.empty((self.world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        if local.is_cuda and dist.get_backend(self.group) == "gloo":
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def gather_queries(self, q_local: torch.Tensor) -> torch.Tensor:
        return q_local if self.world == 1 else self._all_gather_rows(q_local)

    def search(self, q_local: torch.Tensor, k: int, exclude_global: Optional[Sequence[Iterable[int]]] = None,
               exclude_local: Optional[Sequence[Iterable[int]]] = None, excl_cap: Optional[int] = None):
        """q_local [Q/W, d] -> (idx int64 [Q, k] global rows, score float32 [Q, k]) on every rank.
        `exclude_local`: the exclusions of THIS rank's queries only (per local query, GLOBAL row numbers) - what a
        data-parallel front-end knows; they are exchanged with the queries (`excl_cap`: ids per rank, the same
        constant on every rank, default n_local x 128).  Every rank must use the same form in a given call.
        `exclude_global`: the replicated form - per query (all Q of them, in gathered order) GLOBAL row numbers,
        identical on every rank; nothing is exchanged."""
        if exclude_local is not None:
            if exclude_global is not None:
                raise ValueError("give exclusions either per local query (exclude_local) or replicated (exclude_global)")
            if self.comm is not None:
                return self._search_native_local_excl(q_local, k, exclude_local, excl_cap)
            exclude_global = self._gather_local_exclusions(exclude_local, int(q_local.shape[0]), excl_cap)
        if self.comm is not None:
            return self._search_native(q_local, k, exclude_global)
        q_all = self.gather_queries(q_local)
        keys = self.backend.search_partial(q_all, k, self._local_exclusions(exclude_global))
        if self.world == 1:
            return self.backend.merge(keys.unsqueeze(0), k)
        gathered = self._all_gather_rows(keys)
        return self.backend.merge(gathered.view(self.world, keys.shape[0], keys.shape[1]), k)
