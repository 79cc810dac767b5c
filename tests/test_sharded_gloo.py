"""The N>1 path on CPU: two gloo ranks, row-sharded catalog, all-gather of query embeddings and
of per-shard partial key lists, k-way merge.  The oracle stands in for the HIP kernels through a
test-only backend that emits the same packed keys, so what is exercised is the collective
plumbing, the global/local row arithmetic and the exclusion routing of sharded.py."""
from __future__ import annotations

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from instacart_next_order_recommendation_amd import synthetic as syn
from instacart_next_order_recommendation_amd.sharded import ShardedSearch, shard_bounds


def _orderable(score: np.ndarray) -> np.ndarray:
    u = score.astype(np.float32).view(np.uint32).astype(np.uint64)
    neg = (u & np.uint64(0x80000000)) != 0
    return np.where(neg, (~u) & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))


class OracleBackend:
    """TEST ONLY: shard-local search by the CPU oracle, packed exactly like icrec_search_partial."""

    def __init__(self, rows: np.ndarray, row_offset: int):
        self.rows, self.off = rows, row_offset

    def search_partial(self, q, k, exclude):
        from oracle import oracle

        idx, sc = oracle.search(q.numpy(), self.rows, k, exclude, row_offset=self.off)
        keys = (_orderable(sc) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - idx.astype(np.uint64) & np.uint64(0xFFFFFFFF))
        keys = np.where(idx < 0, np.uint64(0), keys)
        return torch.from_numpy(keys.view(np.int64))

    def merge(self, keys, k):
        kk = keys.numpy().view(np.uint64)                     # [W, Q, k]
        W, Q, _ = kk.shape
        flat = np.transpose(kk, (1, 0, 2)).reshape(Q, W * k)
        order = np.argsort(flat, axis=1)[:, ::-1][:, :k]     # larger key = better hit
        top = np.take_along_axis(flat, order, axis=1)
        u = (top >> np.uint64(32)).astype(np.uint32)
        bits = np.where(u & np.uint32(0x80000000), u & np.uint32(0x7FFFFFFF), ~u)
        score = np.where(top == 0, np.float32(0), bits.astype(np.uint32).view(np.float32))
        idx = np.where(top == 0, -1, (np.uint64(0xFFFFFFFF) - (top & np.uint64(0xFFFFFFFF))).astype(np.int64))
        return torch.from_numpy(idx), torch.from_numpy(score.astype(np.float32))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_rows: int, q_total: int, k: int, out_dir: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = syn.synthetic_embeddings(n_rows, 384, seed=1)
        q = syn.synthetic_embeddings(q_total, 384, seed=2)
        b = shard_bounds(n_rows, world)
        lo, hi = b[rank], b[rank + 1]
        per = q_total // world
        excl = [[(7 * i) % n_rows, (13 * i + 1) % n_rows, 5] for i in range(q_total)]  # global rows
        ss = ShardedSearch(OracleBackend(P[lo:hi], lo), lo, hi)
        idx, sc = ss.search(torch.from_numpy(q[rank * per:(rank + 1) * per]), k, excl)
        # the same exclusions handed in per rank (each rank knows only ITS queries' lists): exchanged, same result
        idx2, sc2 = ss.search(torch.from_numpy(q[rank * per:(rank + 1) * per]), k,
                              exclude_local=excl[rank * per:(rank + 1) * per], excl_cap=16)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), idx=idx.numpy(), sc=sc.numpy(), idx2=idx2.numpy(), sc2=sc2.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_search_equals_unsharded(tmp_path):
    from oracle import oracle

    n_rows, q_total, k, world = 1001, 6, 20, 2
    mp.start_processes(_worker, args=(world, _free_port(), n_rows, q_total, k, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")
    P = syn.synthetic_embeddings(n_rows, 384, seed=1)
    q = syn.synthetic_embeddings(q_total, 384, seed=2)
    excl = [[(7 * i) % n_rows, (13 * i + 1) % n_rows, 5] for i in range(q_total)]
    want_i, want_s = oracle.search(q, P, k, excl)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        np.testing.assert_array_equal(got["idx"], want_i)   # every rank holds the full, identical result
        np.testing.assert_array_equal(got["sc"], want_s)
        np.testing.assert_array_equal(got["idx2"], want_i)  # per-rank exclusion lists, exchanged
        np.testing.assert_array_equal(got["sc2"], want_s)


def test_local_exclusion_csr_rejects_overflow_and_bad_shapes():
    P = syn.synthetic_embeddings(64, 384, seed=1)
    ss = ShardedSearch(OracleBackend(P, 0), 0, 64)
    q = torch.from_numpy(syn.synthetic_embeddings(2, 384, seed=2))
    with pytest.raises(ValueError, match="excl_cap"):
        ss.search(q, 5, exclude_local=[[1, 2, 3], [4]], excl_cap=3)
    with pytest.raises(ValueError, match="local queries"):
        ss.search(q, 5, exclude_local=[[1]])
    with pytest.raises(ValueError, match="either"):
        ss.search(q, 5, exclude_global=[[1], [2]], exclude_local=[[1], [2]])


def test_single_process_degenerates_to_plain_search():
    from oracle import oracle

    P = syn.synthetic_embeddings(300, 384, seed=1)
    q = syn.synthetic_embeddings(3, 384, seed=2)
    ss = ShardedSearch(OracleBackend(P, 0), 0, 300)
    idx, sc = ss.search(torch.from_numpy(q), 10)
    wi, ws = oracle.search(q, P, 10)
    np.testing.assert_array_equal(idx.numpy(), wi)
    np.testing.assert_array_equal(sc.numpy(), ws)
