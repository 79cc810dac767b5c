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

The reference has no distributed code at all; this is new design.  The class
is backend-agnostic so that the collective plumbing can be exercised with gloo on CPU in
tests (with the oracle standing in for the kernels); the product backend is HipShardBackend.
"""
from __future__ import annotations

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


class ShardedSearch:
    """Collective top-k over a row-sharded catalog.  Every rank calls `search` with ITS slice of
    the query embeddings (equal slice sizes on all ranks); every rank gets the full result."""

    def __init__(self, backend: ShardBackend, row_lo: int, row_hi: int, group=None):
        self.backend = backend
        self.row_lo, self.row_hi = int(row_lo), int(row_hi)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

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

    def search(self, q_local: torch.Tensor, k: int, exclude_global: Optional[Sequence[Iterable[int]]] = None):
        """q_local [Q/W, d] -> (idx int64 [Q, k] global rows, score float32 [Q, k]) on every rank.
        `exclude_global`: per query (all Q of them, in gathered order) GLOBAL row numbers."""
        q_all = self.gather_queries(q_local)
        keys = self.backend.search_partial(q_all, k, self._local_exclusions(exclude_global))
        if self.world == 1:
            return self.backend.merge(keys.unsqueeze(0), k)
        gathered = self._all_gather_rows(keys)
        return self.backend.merge(gathered.view(self.world, keys.shape[0], keys.shape[1]), k)
