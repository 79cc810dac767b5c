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
    """The BARE form the driver uses, `python bench.py --gpus 2 ...` without a launcher: bench.py starts its own two
    ranks as a fresh child (torch.distributed.run), relays rank 0's line and exits with the child's code.  Rehearsal
    mode: both ranks on cuda:0, gloo.  The line carries the configs[4] leg (here 300,000 rows instead of 10 M) and the
    verdict of the exchange check, exclusion exchange included."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ICREC_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--rows-10m", "300000"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["rehearsal_not_a_measurement"]
    assert d["config"]["contexts_per_gpu_per_step"] == 64
    assert d["exchange_verified"] is True and d["exclusion_exchange_verified"] is True
    leg = d["configs4_10m_rows"]
    assert leg["queries_per_step"] == 4096 and leg["contexts_encoded_per_gpu"] == 2048 and leg["rows_per_gpu"] == 150000
    assert leg["result_properties_ok"] and leg["qps"] > 0


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


def _csr_restatement(off_all, rows_all, cap, row_lo, row_hi):
    """numpy restatement of comm.hip's excl_sanitize / count / scan / fill: offsets clamped to [0, cap] and made
    non-decreasing by a running maximum, then per gathered query (rank-major) the ids inside [row_lo, row_hi), rebased."""
    world, n1 = off_all.shape
    off = np.maximum.accumulate(np.clip(off_all.astype(np.int64), 0, cap), axis=1)
    csr_off, csr_idx = [0], []
    for r in range(world):
        for i in range(n1 - 1):
            seg = rows_all[r, off[r, i]:off[r, i + 1]].astype(np.int64)
            seg = seg[(seg >= row_lo) & (seg < row_hi)] - row_lo
            csr_idx.extend(seg.tolist())
            csr_off.append(len(csr_idx))
    return np.asarray(csr_off, np.int32), np.asarray(csr_idx, np.int32)


def test_exclusion_exchange_layout_world8_on_one_gpu():
    """The W > 1 branch of icrec_search_sharded_excl without a second GPU: the device-side step
    (icrec_exclusions_to_shard_csr = excl_sanitize / count / scan / fill of csrc/comm.hip) on gathered buffers laid out
    exactly as ncclAllGather lays them down — [8][512 + 1] offsets, [8][cap] ids — at BASELINE configs[3]'s sizes (8
    ranks x 512 local queries, shard 1 of the 49,688-row catalog), lists incl. empty and cap-full ranks, against a
    numpy restatement; then MALFORMED offsets (negative, beyond cap, decreasing — ADVICE r3): never out of bounds,
    equal to the restatement's sanitised reading; and the CSR drives icrec_search to the oracle's result."""
    import torch

    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.search import DeviceIndex
    from instacart_next_order_recommendation_amd.sharded import exclusions_to_shard_csr, shard_bounds
    from oracle import oracle

    dev = torch.device("cuda:0")
    W, n_local, cap = 8, 512, 512 * 12
    b = shard_bounds(49_688, W)
    lo, hi = b[1], b[2]
    rng = np.random.default_rng(11)
    off_all = np.zeros((W, n_local + 1), np.int32)
    rows_all = np.zeros((W, cap), np.int32)
    lists = []
    for r in range(W):
        flat = []
        for i in range(n_local):
            if r == 3:          # a rank without any exclusion
                n = 0
            elif r == 5:        # a rank that fills its buffer to the last slot
                n = 12
            else:
                n = int(rng.integers(0, 13)) if i % 7 else 0
            e = sorted(set(rng.integers(0, 49_688, size=n).tolist())) if r != 5 else \
                sorted(rng.choice(49_688, size=12, replace=False).tolist())
            lists.append(e)
            flat.extend(e)
            off_all[r, i + 1] = len(flat)
        assert len(flat) <= cap
        rows_all[r, :len(flat)] = flat
        rows_all[r, len(flat):] = 7          # padding past the last offset must be ignored
    assert off_all[5, -1] == cap
    want_off, want_idx = _csr_restatement(off_all, rows_all, cap, lo, hi)
    got_off, got_idx = exclusions_to_shard_csr(torch.from_numpy(off_all).to(dev), torch.from_numpy(rows_all).to(dev), lo, hi)
    np.testing.assert_array_equal(got_off.cpu().numpy(), want_off)
    np.testing.assert_array_equal(got_idx.cpu().numpy()[:want_off[-1]], want_idx)
    # the CSR means what the search expects: gathered queries against the shard, a sample against the oracle
    P = syn.synthetic_embeddings(49_688, 384, seed=1)
    q = syn.synthetic_embeddings(W * n_local, 384, seed=2)
    ix = DeviceIndex(torch.from_numpy(P[lo:hi]).to(dev), dev, row_offset=lo)
    idx = torch.empty((W * n_local, 20), dtype=torch.int64, device=dev)
    sc = torch.empty((W * n_local, 20), dtype=torch.float32, device=dev)
    ix.search_into(torch.from_numpy(q).to(dev), 20, got_idx, got_off, idx, sc)
    sample = [0, 1, 511, 512, 513, 1535, 1536, 2560, 2561, 3071, 4095]
    wi, ws = oracle.search(q[sample], P[lo:hi], 20, [[v - lo for v in lists[g] if lo <= v < hi] for g in sample], row_offset=lo)
    np.testing.assert_array_equal(idx[sample].cpu().numpy(), wi)
    np.testing.assert_array_equal(sc[sample].cpu().numpy(), ws)
    ix.close()
    # malformed offsets from "another rank": overlapping full segments, negatives, beyond cap, decreasing runs
    bad = off_all.copy()
    bad[0, :5] = [0, cap, 0, cap, 0]
    bad[1, 1:9] = [-5, 2 * cap, 3, 2, 1, -1, cap + 1, 0]
    bad[2] = bad[2][::-1]
    bad[4, -1] = 2**31 - 1
    want_off, want_idx = _csr_restatement(bad, rows_all, cap, lo, hi)
    assert want_off[-1] <= W * cap
    got_off, got_idx = exclusions_to_shard_csr(torch.from_numpy(bad).to(dev), torch.from_numpy(rows_all).to(dev), lo, hi)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got_off.cpu().numpy(), want_off)
    np.testing.assert_array_equal(got_idx.cpu().numpy()[:want_off[-1]], want_idx)


def test_gathered_partial_keys_layout_world8_on_one_gpu():
    """The W > 1 branch of icrec_search_sharded without a second GPU: keys_all [8, 4096, 20] built from eight
    icrec_search_partial calls over the eight row shards of the 49,688-row catalog (rank-major, exactly what
    ncclAllGather of the per-rank [4096, 20] key blocks lays down) -> icrec_merge_topk == the unsharded index, bit
    for bit on all 4,096 queries, and == the oracle on a sample.  The first and last shards have 6,211 rows each
    (49,688 = 8 x 6,211)."""
    import torch

    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.search import DeviceIndex, merge_topk
    from instacart_next_order_recommendation_amd.sharded import shard_bounds
    from oracle import oracle

    dev = torch.device("cuda:0")
    W, Q, k = 8, 4096, 20
    P = syn.synthetic_embeddings(49_688, 384, seed=1)
    q = syn.synthetic_embeddings(Q, 384, seed=2)
    Pd, qd = torch.from_numpy(P).to(dev), torch.from_numpy(q).to(dev)
    b = shard_bounds(49_688, W)
    assert all(b[r + 1] - b[r] == 6211 for r in range(W))
    rng = np.random.default_rng(3)
    excl = [sorted(set(rng.integers(0, 49_688, size=int(rng.integers(0, 30))).tolist())) if g % 3 else [] for g in range(Q)]
    keys_all = torch.empty((W, Q, k), dtype=torch.int64, device=dev)
    for r in range(W):
        lo, hi = b[r], b[r + 1]
        shard = DeviceIndex(Pd[lo:hi], dev, row_offset=lo, storage="f32+filter" if r % 2 else "f32")
        keys_all[r] = shard.search_partial(qd, k, [[v - lo for v in e if lo <= v < hi] for e in excl])
        shard.close()
    idx, sc = merge_topk(keys_all, k)
    full = DeviceIndex(Pd, dev)
    fi, fs = full.search(qd, k, excl)
    assert torch.equal(idx, fi) and torch.equal(sc, fs)
    sample = [0, 1, 511, 512, 2047, 2048, 4094, 4095]
    wi, ws = oracle.search(q[sample], P, k, [excl[g] for g in sample])
    np.testing.assert_array_equal(idx[sample].cpu().numpy(), wi)
    np.testing.assert_array_equal(sc[sample].cpu().numpy(), ws)
    # a rank-major slip would not survive: swapping two ranks' blocks changes nothing (the merge is order-free in the
    # rank axis), but reading the buffer query-major does
    wrong, _ = merge_topk(keys_all.permute(1, 0, 2).contiguous().view(W, Q, k), k)
    assert not torch.equal(wrong, fi)
    full.close()
