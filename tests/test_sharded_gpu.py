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
