"""The whole N>1 product path on the one-GPU box: two ranks (both on cuda:0, gloo rendezvous) run the
real HipShardBackend + ShardedSearch and bench.py's multi-rank flow."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ICREC_ROOT"])
from instacart_next_order_recommendation_amd import synthetic as syn
from instacart_next_order_recommendation_amd.sharded import HipShardBackend, ShardedSearch, shard_bounds
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
P = syn.synthetic_embeddings(3001, 384, seed=1)
q = syn.synthetic_embeddings(8, 384, seed=2)
b = shard_bounds(3001, world)
lo, hi = b[rank], b[rank + 1]
excl = [[(7 * i) % 3001, 5, 2999] for i in range(8)]
ss = ShardedSearch(HipShardBackend(torch.from_numpy(P[lo:hi]).to(dev), lo, dev), lo, hi)
per = 8 // world
idx, sc = ss.search(torch.from_numpy(q[rank * per:(rank + 1) * per]).to(dev), 20, excl)
np.savez(os.path.join(os.environ["ICREC_OUT"], f"r{rank}.npz"), idx=idx.cpu().numpy(), sc=sc.cpu().numpy())
dist.destroy_process_group()
"""


def _torchrun(args, env_extra, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517"] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_real_kernels_equal_unsharded(tmp_path):
    from instacart_next_order_recommendation_amd import synthetic as syn
    from oracle import oracle

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    r = _torchrun([str(script)], {"ICREC_ROOT": str(ROOT), "ICREC_OUT": str(tmp_path)})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    P = syn.synthetic_embeddings(3001, 384, seed=1)
    q = syn.synthetic_embeddings(8, 384, seed=2)
    excl = [[(7 * i) % 3001, 5, 2999] for i in range(8)]
    wi, ws = oracle.search(q, P, 20, excl)
    for rank in range(2):
        got = np.load(tmp_path / f"r{rank}.npz")
        np.testing.assert_array_equal(got["idx"], wi)
        np.testing.assert_array_equal(got["sc"], ws)


def test_bench_multi_rank_flow_rehearsal():
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64"],
                  {"ICREC_BENCH_REHEARSAL": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["rehearsal_not_a_measurement"]
    assert d["config"]["contexts_per_gpu_per_step"] == 64


def test_native_rccl_world1_equals_plain_search():
    """icrec_comm_init (a real ncclCommInitRank on one rank) + icrec_search_sharded: both ncclAllGathers run
    on the stream and the result is bit-equal to icrec_search and to the oracle, exclusions included."""
    import torch

    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.sharded import HipShardBackend, NativeComm, ShardedSearch
    from oracle import oracle

    dev = torch.device("cuda:0")
    P = syn.synthetic_embeddings(6211, 384, seed=1)   # one shard of configs[3] (49,688 / 8 rows)
    q = syn.synthetic_embeddings(512, 384, seed=2)    # 4,096 / 8 queries
    excl = [[(13 * i) % 6211, 5, 6210] if i % 3 == 0 else [] for i in range(512)]
    comm = NativeComm(0, 1, dev, NativeComm.unique_id())
    assert comm.world == 1 and comm.rank == 0
    be = HipShardBackend(torch.from_numpy(P).to(dev), 0, dev)
    ss = ShardedSearch(be, 0, 6211, comm=comm)
    qd = torch.from_numpy(q).to(dev)
    for ex in (None, excl):
        idx, sc = ss.search(qd, 20, ex)
        pi, ps = be.index.search(qd, 20, ex)
        assert torch.equal(idx, pi) and torch.equal(sc, ps)
        wi, ws = oracle.search(q, P, 20, ex)
        np.testing.assert_array_equal(idx.cpu().numpy(), wi)
        np.testing.assert_array_equal(sc.cpu().numpy(), ws)
    # a shard with a row offset: global rows come back
    be2 = HipShardBackend(torch.from_numpy(P[1000:3000]).to(dev), 1000, dev)
    ss2 = ShardedSearch(be2, 1000, 3000, comm=comm)
    idx2, sc2 = ss2.search(qd[:64], 20, [[1500, 2999, 5]] * 64)
    wi2, ws2 = oracle.search(q[:64], P[1000:3000], 20, [[500, 1999]] * 64)
    np.testing.assert_array_equal(idx2.cpu().numpy(), wi2 + 1000)
    np.testing.assert_array_equal(sc2.cpu().numpy(), ws2)
    comm.close()


def test_native_per_rank_exclusions_world1():
    """icrec_search_sharded_excl through a real one-rank RCCL communicator: the exclusions arrive as GLOBAL rows of
    this rank's own queries, are all-gathered (offsets + padded ids), cut to the shard and turned into the local
    CSR on the device - same bits as the replicated form and as the oracle, with a row offset."""
    import torch

    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.sharded import HipShardBackend, NativeComm, ShardedSearch
    from oracle import oracle

    dev = torch.device("cuda:0")
    P = syn.synthetic_embeddings(6211, 384, seed=1)
    q = syn.synthetic_embeddings(512, 384, seed=2)
    comm = NativeComm(0, 1, dev, NativeComm.unique_id())
    lo, hi = 1000, 5000  # the shard holds global rows [1000, 5000): ids outside it belong to other shards
    be = HipShardBackend(torch.from_numpy(P[lo:hi]).to(dev), lo, dev)
    ss = ShardedSearch(be, lo, hi, comm=comm)
    qd = torch.from_numpy(q).to(dev)
    rng = np.random.default_rng(5)
    excl = [sorted(set(rng.integers(0, 6211, size=int(rng.integers(0, 40))).tolist())) if i % 4 else [] for i in range(512)]
    idx, sc = ss.search(qd, 20, exclude_local=excl, excl_cap=512 * 40)
    ridx, rsc = ss.search(qd, 20, exclude_global=excl)
    assert torch.equal(idx, ridx) and torch.equal(sc, rsc)
    wi, ws = oracle.search(q, P[lo:hi], 20, [[r - lo for r in e if lo <= r < hi] for e in excl], row_offset=lo)
    np.testing.assert_array_equal(idx.cpu().numpy(), wi)
    np.testing.assert_array_equal(sc.cpu().numpy(), ws)
    # every query excludes its own best hit: the exchange really reaches the kernel
    pi, ps = ss.search(qd, 20)
    best = pi[:, 0].cpu().tolist()
    idx2, _ = ss.search(qd, 20, exclude_local=[[b] for b in best])
    assert not any(b in row for b, row in zip(best, idx2.cpu().tolist()))
    assert torch.equal(idx2[:, :19], pi[:, 1:])
    # no exclusions at all on this rank (other ranks may still have some): zero offsets
    idx3, sc3 = ss.search(qd, 20, exclude_local=[[] for _ in range(512)])
    assert torch.equal(idx3, pi) and torch.equal(sc3, ps)
    comm.close()


def test_configs3_sizes_4096_queries():
    """BASELINE configs[3] at its own sizes on one GPU: (a) 4,096 queries over the whole 49,688-row catalog (filter
    index, 8-query oracle sample, bit-equal to the exact index); (b) the 4,096 gathered queries over ONE
    6,211-row shard (rows 6,211 .. 12,421) through icrec_search_sharded with a one-rank RCCL communicator."""
    import torch

    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.search import DeviceIndex
    from instacart_next_order_recommendation_amd.sharded import HipShardBackend, NativeComm, ShardedSearch, shard_bounds
    from oracle import oracle

    dev = torch.device("cuda:0")
    P = syn.synthetic_embeddings(49_688, 384, seed=1)
    q = syn.synthetic_embeddings(4096, 384, seed=2)
    qd = torch.from_numpy(q).to(dev)
    Pd = torch.from_numpy(P).to(dev)
    fi = DeviceIndex(Pd, dev, storage="f32+filter")
    ei = DeviceIndex(Pd, dev, storage="f32")
    idx, sc = fi.search(qd, 20)
    xi, xs = ei.search(qd, 20)
    assert torch.equal(idx, xi) and torch.equal(sc, xs)
    sample = [0, 1, 511, 512, 2047, 2048, 4094, 4095]
    wi, ws = oracle.search(q[sample], P, 20)
    np.testing.assert_array_equal(idx[sample].cpu().numpy(), wi)
    np.testing.assert_array_equal(sc[sample].cpu().numpy(), ws)
    b = shard_bounds(49_688, 8)
    assert b[1] - b[0] == 6211
    lo, hi = b[1], b[2]
    comm = NativeComm(0, 1, dev, NativeComm.unique_id())
    be = HipShardBackend(Pd[lo:hi], lo, dev, storage="f32+filter")
    ss = ShardedSearch(be, lo, hi, comm=comm)
    si, s_sc = ss.search(qd, 20)
    wi2, ws2 = oracle.search(q[sample], P[lo:hi], 20, row_offset=lo)
    np.testing.assert_array_equal(si[sample].cpu().numpy(), wi2)
    np.testing.assert_array_equal(s_sc[sample].cpu().numpy(), ws2)
    # shard-level property at full size: each list is sorted (score desc, row asc), rows inside the shard
    s_np, i_np = s_sc.cpu().numpy(), si.cpu().numpy()
    assert ((i_np >= lo) & (i_np < hi)).all()
    d = np.diff(s_np, axis=1)
    assert (d <= 0).all() and ((d < 0) | (np.diff(i_np, axis=1) > 0)).all()
    comm.close()
    fi.close(); ei.close()
