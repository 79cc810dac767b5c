#!/usr/bin/env python3
"""bench.py — recommend() throughput on MI355X: SBERT encode -> cosine -> top-20 over the
49,688-product catalog (BASELINE.json metric), with the kernel roofline and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload 49k7|10m]

One "step" = one pass of the hot path over one batch of synthetic user contexts whose token
ids are already resident in HBM: icrec_encode (BERT forward + pool + normalise) then
icrec_search (cosine + top-20, exclusions off), results left in HBM.  At N > 1 (launched by
torch.distributed.run, one rank per GPU, RCCL) the catalog is row-sharded, each rank encodes
`--batch` contexts (weak scaling: global batch = N * batch), query embeddings and per-shard
partial top-k lists are all-gathered and merged (instacart_next_order_recommendation_amd/sharded.py).

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

CATALOG_ROWS = 49_688  # notebooks/serve_recommendations.ipynb:101
TOP_K = 20
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # >= 200 timed iterations after 20 warm-ups
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None,
                    help="user contexts per GPU per step (default: 1024 on one GPU = BASELINE configs[1]/[2]; "
                         "4096 / N at N > 1 = configs[3]'s 4,096-query batch over the sharded catalog)")
    ap.add_argument("--workload", default="49k7", choices=["49k7", "10m"])
    ap.add_argument("--catalog-rows", default=None, choices=["f32", "bf16", "f32+filter", "bf16+filter"],
                    help="how the index keeps its rows in HBM (default: f32+filter for 49k7 — what Recommender uses: fp32 rows "
                         "plus f16 filter planes, results bit-identical to f32; bf16+filter for 10m: bf16 rows as BASELINE configs[4] says, plus filter planes)")
    ap.add_argument("--gemm-mode", default=None, choices=["f32", "f16x3"],
                    help="encoder GEMM arithmetic (default: the package default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-request (Q=1) latency loop")
    ap.add_argument("--cpu-sample", type=int, default=128, help="queries in the CPU baseline sample")
    ap.add_argument("--rows-10m", type=int, default=10_000_000,
                    help="catalog rows of the configs[4] workload / leg (tests pass a small number)")
    ap.add_argument("--no-10m-leg", action="store_true",
                    help="skip the configs[4] leg (4,096 queries over the 10 M-row bf16 catalog) the default line carries")
    ap.add_argument("--no-http", action="store_true", help="skip the /recommend-over-HTTP leg")
    a = ap.parse_args()
    if a.batch is None:
        a.batch = 1024 if a.gpus == 1 else max(4096 // a.gpus, 1)
        a.batch_defaulted = True
    else:
        a.batch_defaulted = False
    return a


def cpu_baseline(weights, shape, ids, cu, catalog, n_sample: int):
    """The reference's CPU path on this box's host cores, on a bounded sample of the same workload.

    kind "port:torch": the reference's own op sequence (serve_recommendations.py:213-225) on torch CPU
    kernels — padded batches of 64 through a BertModel-equivalent forward, mean-pool, Normalize,
    F.normalize x2 + torch.mm, argsort(descending=True), the Python top-k loop (oracle/torch_reference.py;
    the reference module itself is not importable here: sentence_transformers is absent).  Its embeddings
    are checked against the C oracle in the same run.  The OpenMP C oracle's timings stay as `c_port`."""
    from oracle import oracle, torch_reference as tr

    n_thr = oracle.usable_cpus()  # the container's CPU quota, not the host's core count
    r = tr.measure(weights, shape, ids, cu, catalog, n_sample, TOP_K, n_thr)
    emb_t = r.pop("emb")
    oracle.set_threads(n_thr)
    cfg = oracle.make_cfg(vocab_size=shape.vocab_size, n_normalize=shape.n_normalize)
    cu_s = cu[: n_sample + 1]
    ids_s = ids[: cu_s[-1]]
    t0 = time.perf_counter()
    emb = oracle.encode(weights, cfg, ids_s, cu_s)
    t1 = time.perf_counter()
    o_idx, o_sc = oracle.search(emb, catalog, TOP_K, None)
    t2 = time.perf_counter()
    _, o_sc21 = oracle.search(emb, catalog, TOP_K + 1, None)  # for the ambiguity rule of the self-check (untimed)
    one = []
    for i in range(5):
        a = time.perf_counter()
        e1 = oracle.encode(weights, cfg, ids[cu[i]:cu[i + 1]], np.array([0, cu[i + 1] - cu[i]], np.int32))
        oracle.search(e1, catalog, TOP_K, None)
        one.append((time.perf_counter() - a) * 1e3)
    total = r["encode_s"] + r["rank_s"]
    return {"_oracle": {"emb": emb, "idx": o_idx, "score": o_sc, "score21": o_sc21},
            "value": n_sample / total, "unit": "queries/s", "cores": r["torch_threads"], "kind": "port:torch",
            "single_request_p50_ms": r["single_request_p50_ms"],
            "reference_serving_qps_one_request_at_a_time": 1e3 / r["single_request_p50_ms"],
            "configs0_single_query_vs_1k_products_p50_ms": r["configs0_p50_ms"],
            "torch_get_num_threads": r["torch_threads"], "os_cpu_count": r["os_cpu_count"],
            "cgroup_cpu_quota": r["cgroup_cpu_quota"],
            "max_abs_embedding_diff_vs_c_oracle": float(np.abs(emb_t - emb).max()),
            "sample": f"{n_sample} of the step's contexts ({int(cu_s[-1])} tokens) on torch CPU kernels: model.encode in "
                      f"padded batches of 64 {r['encode_s']:.2f}s + per query cos_sim (F.normalize of the whole catalog, "
                      f"as the reference does on every call) / argsort / top-{TOP_K} loop over {catalog.shape[0]} rows "
                      f"{r['rank_s']:.2f}s; the reference serves one request at a time: single_request_p50_ms",
            "c_port": {"value": n_sample / (t2 - t0), "unit": "queries/s", "cores": oracle.threads(),
                       "single_request_p50_ms": float(np.median(one)),
                       "note": f"oracle/icrec_oracle.c (fixed-order OpenMP C restatement): encode {t1 - t0:.2f}s + "
                               f"search {t2 - t1:.2f}s on the same sample"}}


def load_traffic(kernel_key: str):
    """HBM-side bytes per launch of `kernel_key` as RECORDED by the last committed PMC run (profiles/
    r0N_roofline_traffic.json: two rocprofv3 --pmc passes over this bench command, FETCH_SIZE doubled per the gfx950
    rule, WRITE_SIZE as is; written by tools/pmc_traffic_json.py) - counters cannot be read from inside the timed
    process, so this field is a recorded value, not one measured by this run.  (None, note) when the record is missing."""
    paths = sorted((ROOT / "profiles").glob("r0*_roofline_traffic.json"))
    path = paths[-1] if paths else ROOT / "profiles" / "roofline_traffic.json"
    try:
        rec = json.loads(path.read_text())[kernel_key]
        return float(rec["traffic_bytes"]), f"RECORDED, not measured by this run: {rec['note']} ({path.name})"
    except Exception:  # noqa: BLE001
        return None, f"no PMC record for {kernel_key} in {path.name}"


def roofline(mode: str, achieved: float, n: int, ms: float, flops: float, tokens: int) -> dict:
    """Roofline entry for the dominant kernel.  f16x3 mode: the fused post-attention kernel of a layer (attention-output
    projection + residual + LayerNorm, then FFN-up + GELU + FFN-down + residual + LayerNorm; 6 launches per step);
    f32 mode: the FFN up-projection GEMM.  `achieved` counts ALGORITHMIC FLOPs once, whatever the arithmetic (the
    3-term split issues 3 MFMAs per product and is priced against 2500/3 TFLOP/s)."""
    if mode == "f32":
        traffic, note = load_traffic("linear_kernel_gelu")
        return {"kernel": "linear_kernel<128x128, GELU> (FFN up-projection, v_mfma_f32_32x32x2_f32)",
                "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_note": note,
                "launches_timed": n, "avg_launch_ms": ms, "flops_per_launch": flops, "tokens_per_launch": tokens}
    peak = PEAK_F16_MFMA_TFLOPS / 3.0
    traffic, note = load_traffic("ffn_fused2_kernel")
    w_bytes = (2 * 1536 * 384 + 384 * 384 + 5 / 6 * 1152 * 384) * 2 * 2  # W1, W2, Wo (+ 5/6 Wqkv) as packed hi/lo f16 fragments
    l2_bytes = (tokens // 64) * (w_bytes + 64 * 384 * 4 * 2 + 6 * 1024)
    return {"kernel": "ffn_fused2_kernel<0, AO> = the whole post-attention part of a layer in one launch: attention-out + residual "
                      "+ LayerNorm, FFN-up + erf-GELU + FFN-down + residual + LayerNorm, then the NEXT layer's QKV projection "
                      "(5 of the 6 launches of a step), all on chip; 3x v_mfma_f32_16x16x32_f16 per product, weights streamed "
                      "L2 -> registers",
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "peak_note": "fp32-accurate product = 3 f16 MFMAs, so the algorithm's MFMA roof is 2500/3 TFLOP/s of "
                         "algorithmic FLOPs; against the raw f16 dense peak the fraction is frac_of_f16_dense_peak",
            "frac_of_f16_dense_peak": achieved / PEAK_F16_MFMA_TFLOPS,
            "frac_of_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_note": note,
            "algorithmic_bytes": tokens * 384 * (4 + 4 + 4) + tokens * 1152 * 4 * 5 / 6 + w_bytes,
            "algorithmic_bytes_note": "context planes in + x planes in (residual) + x planes out (2 x 2 B/elt each) + fp32 "
                                      "Q/K/V rows out (5 of 6 launches), + the layer's packed Wo/W1/W2 (+ next Wqkv) fragments once",
            "ceiling_note": "a registers-only loop of the same MFMA instruction on RANDOM operands sustains 1,970 TFLOP/s f16 "
                            "dense on this chip at 1.95 GHz (tools/mfma_shape.hip, profiles/r03_mfma_shape_microbench.txt; the "
                            "32x32x16 form the engine used before: 1,710 at 1.69 GHz; both reach 2,450 on all-zero operands): "
                            "the clock falls under matrix load, i.e. 657 TFLOP/s of f16x3 products is the measured ceiling",
            "frac_of_measured_mfma_ceiling": achieved / (1970.0 / 3.0),
            # every 64-token workgroup pulls the layer's Wo/W1/W2 fragments + its planes through its vector L1
            "l2_stream": {"bytes_per_launch": l2_bytes,
                          "TBps_at_this_launch_time": l2_bytes / (ms * 1e-3) / 1e12 if ms > 0 else None,
                          "note": "L2 -> L1 bytes per launch BY CONSTRUCTION (one pass over the layer's fragments + the block's planes "
                                  "per 64-token workgroup), averaged over a step's launches (5 of 6 stream the next layer's Wqkv too). "
                                  "Counter check (TCP_TCC_READ_REQ x 128 B, profiles/r04_pmc_l2_stream.txt): 11.65 GB for the last "
                                  "layer's launch against 11.3 GB by construction for that launch; profiles/README.md has the table. "
                                  "Not a wall: the same access pattern without arithmetic streams 23-32 TB/s (tools/l2_stream.hip, "
                                  "profiles/r03_l2_stream_microbench.txt)"},
            "launches_timed": n, "avg_launch_ms": ms, "flops_per_launch": flops, "tokens_per_launch": tokens}


def self_launch(args) -> None:
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (the bare form the driver uses): start
    the N ranks as a FRESH child `python -m torch.distributed.run ... bench.py <same arguments>`, relay its output (rank 0's
    one JSON line) and exit with its code.  Decided before this process imports torch or touches a GPU: the parent only
    parses arguments, picks a free loopback port and waits."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def http_leg(seconds: float = 3.0) -> dict:
    """/recommend over HTTP (north_star words the target as "/recommend QPS"; reference surface
    src/api/routes/recommend.py:84-199, src/api/schemas.py:15-70): tools/http_load.py as a CHILD process — it starts
    api/serve.py (3 GPU-owner processes on the one GPU + 10 FastAPI front-ends on one port: a GPU owner is one Python thread
    and saturates at ~19 k requests per second) on the synthetic 49,688-product catalog and 4 load-generator processes holding
    256 keep-alive connections that POST user contexts (top_k 20) back to back.  Everything shares this box's CPU quota (16
    CPUs on the GPU box: 13 server processes + 4 generators), so the figure is a host-side number; per-request
    `recommendation_served` log records are off (METRICS_LOG_LEVEL=WARNING).
    Runs before this process touches the GPU."""
    import subprocess

    env = dict(os.environ, METRICS_LOG_LEVEL="WARNING", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "tools" / "http_load.py"), "--gpu-workers", "3", "--frontends", "10", "--client-procs", "4",
           "--clients", "256", "--seconds", str(seconds)]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"http_load exited {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
        d = json.loads(line[-1])
    except Exception as exc:  # noqa: BLE001 - the leg must never take the bench line down with it
        return {"error": f"{type(exc).__name__}: {exc}"}
    return {"http_qps": d["qps"], "http_p50_ms": d["p50_ms"], "http_p95_ms": d["p95_ms"], "http_p99_ms": d["p99_ms"],
            "failed": d["requests_failed"], "requests_ok": d["requests_ok"], "seconds": d["seconds"],
            "gpu_workers": d.get("gpu_workers", 1), "frontends": d["frontends"], "load_generator_processes": d["client_procs"],
            "connections": d["connections"], "cpus_busy_per_process": d.get("cpus_busy_per_process"),
            "cpu_quota": d["cpu_quota"], "server_startup_s": d["server_startup_s"], "top_k": TOP_K,
            "leg_wall_s": round(time.perf_counter() - t0, 1),
            "note": "POST /recommend over loopback TCP, HTTP/1.1 keep-alive, through api/serve.py --gpu-workers 3: 3 GPU-owner "
                    "processes on the one GPU (tokenise + micro-batch + the same encode/search calls as `value`) + 10 FastAPI "
                    "front-ends, driven by 4 generator processes on the same CPU quota; 0 failed is part of the claim; metrics log "
                    "records off"}


def main() -> None:
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    http = None
    if (args.gpus == 1 and args.workload == "49k7" and not args.no_http and not args.no_latency
            and "WORLD_SIZE" not in os.environ):
        http = http_leg()
    import torch
    import torch.distributed as dist

    from instacart_next_order_recommendation_amd import _native, synthetic as syn
    from instacart_next_order_recommendation_amd.encoder import DeviceEncoder
    from instacart_next_order_recommendation_amd.sharded import HipShardBackend, NativeComm, ShardedSearch, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # ICREC_BENCH_REHEARSAL=1: every rank uses cuda:0 and gloo — lets the N>1 code path be exercised on
    # a one-GPU box (numbers from such a run are meaningless and flagged in the output).
    rehearsal = os.environ.get("ICREC_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    shape = syn.BertShape()
    weights = syn.synthetic_bert_weights(shape, seed=0)
    enc = DeviceEncoder(weights, shape, dev, gemm_mode=args.gemm_mode)

    # ---- catalog shard (embedding-level synthetic: clustered unit vectors)
    def make_shard(n_rows: int, generated: bool, storage: str):
        """-> (HipShardBackend over this rank's rows, lo, hi, host catalog or None).  `generated`: rows drawn on the
        device per shard (200 cluster centres + noise, fp32 before the index converts them) instead of the host array."""
        bounds = shard_bounds(n_rows, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        if not generated:
            catalog = syn.synthetic_embeddings(n_rows, shape.hidden, seed=1)
            shard = torch.from_numpy(catalog[lo:hi]).to(dev)
        else:
            catalog = None
            g = torch.Generator(device=dev).manual_seed(1000 + rank)
            centres = torch.randn(200, shape.hidden, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
            shard = torch.empty((hi - lo, shape.hidden), device=dev)
            for s in range(0, hi - lo, 1 << 18):
                m = min(1 << 18, hi - lo - s)
                cid = torch.randint(0, 200, (m,), device=dev, generator=g)
                shard[s:s + m] = centres[cid] + 0.35 * torch.randn(m, shape.hidden, device=dev, generator=g)
        backend = HipShardBackend(shard, lo, dev, storage=storage)
        del shard
        return backend, lo, hi, catalog

    n_rows = CATALOG_ROWS if args.workload == "49k7" else args.rows_10m
    row_storage = args.catalog_rows or ("f32+filter" if args.workload == "49k7" else "bf16+filter")
    backend, lo, hi, catalog = make_shard(n_rows, args.workload != "49k7", row_storage)
    # N > 1: the exchange runs inside libicrec (icrec_search_sharded: two ncclAllGathers on the step's stream);
    # torch.distributed only carries the 128-byte rendezvous id, the barrier and the max-over-ranks of the clock
    comm, comm_note = None, None
    if world > 1 and not rehearsal:
        # from_process_group keeps every rank on ONE collective sequence: each rank loads librccl (a local step),
        # rank 0's id travels behind a status byte, an all_reduce(MIN) of "loaded and have the id" decides for all
        # ranks before anyone enters ncclCommInitRank - a rank that cannot start never leaves the others waiting in
        # a collective it does not join.  (A rank that dies INSIDE ncclCommInitRank stalls the others there: RCCL's
        # semantics, not recoverable here.)
        try:
            comm = NativeComm.from_process_group(dev)
        except Exception as exc:  # noqa: BLE001 - raised on EVERY rank alike: the same kernels, collectives through torch.distributed
            comm_note = f"icrec_comm_init not usable ({type(exc).__name__}: {exc}); exchange through torch.distributed (RCCL) instead"
        ok = torch.tensor([1 if comm is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # ncclCommInitRank itself failed on some rank: all fall back together
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None
            comm_note = "ncclCommInitRank failed on another rank; exchange through torch.distributed (RCCL) instead"
    search = ShardedSearch(backend, lo, hi, comm=comm)

    # ---- this rank's batch of user contexts as packed token ids, resident in HBM
    ids_h, cu_h = syn.synthetic_token_batch(args.batch, seed=1234 + rank)
    max_len = int(np.diff(cu_h).max())
    total_tokens = int(cu_h[-1])
    ids_d = torch.from_numpy(ids_h).to(dev)
    cu_d = torch.from_numpy(cu_h).to(dev)
    emb = torch.empty((args.batch, shape.hidden), device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    verified_excl = [None, None]  # [every rank's exclusion exchange equals the unsharded search, error text of this rank]

    def measure(search, batch=None, steps=None, warmup=None, verify=True):
        """W warm-up steps, then exactly K timed steps between two barriers; N > 1: max over ranks and the check of the
        exchange.  `batch` = (ids, cu_seqlens, max_len, out) on the device (default: the step's own batch).
        -> (elapsed seconds, last idx, last scores, timer readings, exchange_verified or None)"""
        b_ids, b_cu, b_max, b_emb = batch if batch is not None else (ids_d, cu_d, max_len, emb)
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup

        def step():
            enc.encode_packed(b_ids, b_cu, b_max, out=b_emb)
            return search.search(b_emb, TOP_K)

        for _ in range(warmup):
            step()
        barrier()
        _native.timing_reset()
        _native.timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            idx, sc = step()
        barrier()
        elapsed = time.perf_counter() - t0
        _native.timing_enable(False)
        timers = (_native.timing_query(1), _native.timing_query(2), _native.timing_query(3), _native.timing_query(0))
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            if dist.get_backend() == "gloo":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # ---- N > 1: the exchange validates itself.  Every rank searches a 64-query sample of the gathered batch against
        # a replicated, UNSHARDED copy of the catalog and compares with what the sharded step just returned for those
        # queries: a rank-major layout slip in either all-gather would still produce a plausible QPS, not equal bits.
        # The same sample then goes through the exclusion exchange (icrec_search_sharded_excl: two more all-gathers and
        # the device-side CSR rebuild): every LOCAL query excludes its own three best rows (global numbers, mostly on
        # other ranks' shards), and the gathered result must equal the unsharded search with the gathered lists.
        verified = None
        if verify and world > 1 and args.workload == "49k7":
            from instacart_next_order_recommendation_amd.search import DeviceIndex

            q_all = search.gather_queries(b_emb)  # torch.distributed all-gather: independent of the library's own
            n_all = int(q_all.shape[0])
            n_loc = int(b_emb.shape[0])
            sample = torch.linspace(0, n_all - 1, steps=min(64, n_all), device=dev).round().long().unique()
            full = DeviceIndex(torch.from_numpy(catalog).to(dev), dev, storage="f32")
            ref_idx, ref_sc = full.search(q_all[sample], TOP_K)
            same = bool(torch.equal(ref_idx, idx[sample]) and torch.equal(ref_sc, sc[sample])) and idx.shape[0] == n_all
            # the exclusion exchange: its native path (icrec_search_sharded_excl over two more ncclAllGathers) has never run
            # on more than one GPU before a real multi-GPU launch of this file - an exception there must cost the line its
            # `exclusion_exchange_verified`, not the whole scaling record
            same_excl, excl_err = False, None
            try:
                mine = idx[rank * n_loc:(rank + 1) * n_loc, :3].cpu().tolist()  # this rank's own queries, gathered order
                excl_local = [row if (i % 5) else [] for i, row in enumerate(mine)]  # every fifth list empty
                xi, xs = search.search(b_emb, TOP_K, exclude_local=excl_local, excl_cap=n_loc * 4)
                all_top3 = idx[:, :3].cpu().tolist()
                excl_all = [all_top3[g] if ((g % n_loc) % 5) else [] for g in sample.cpu().tolist()]
                rxi, rxs = full.search(q_all[sample], TOP_K, excl_all)
                same_excl = bool(torch.equal(rxi, xi[sample]) and torch.equal(rxs, xs[sample]))
            except Exception as exc:  # noqa: BLE001
                excl_err = f"{type(exc).__name__}: {exc}"
                print(f"rank {rank}: exclusion exchange check raised {excl_err}", file=sys.stderr)
            flag = torch.tensor([1 if same else 0, 1 if same_excl else 0], device=dev)
            if dist.get_backend() == "gloo":
                flag = flag.cpu()
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # the same verdict on every rank
            verified = bool(int(flag[0].item()) == 1)
            verified_excl[0] = bool(int(flag[1].item()) == 1)
            verified_excl[1] = excl_err
            full.close()
        return elapsed, idx, sc, timers, verified

    if comm is not None:
        # one probe step before anything is timed: icrec_search_sharded over more than one rank has never run before a real
        # multi-GPU launch of this file.  If it raises (on every rank alike: an RCCL error is returned to all of them), the
        # ranks agree through torch.distributed and measure with the exchange through torch.distributed instead of dying.
        probe_ok = 1
        try:
            enc.encode_packed(ids_d, cu_d, max_len, out=emb)
            search.search(emb, TOP_K)
            torch.cuda.synchronize(dev)
        except Exception as exc:  # noqa: BLE001
            probe_ok = 0
            comm_note = f"icrec_search_sharded raised on the probe step ({type(exc).__name__}: {exc}); exchange through torch.distributed (RCCL) instead"
            print(f"rank {rank}: {comm_note}", file=sys.stderr)
        t_ok = torch.tensor([probe_ok], device=dev)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        if int(t_ok.item()) == 0:
            comm_note = comm_note or "icrec_search_sharded raised on another rank's probe step; exchange through torch.distributed (RCCL) instead"
            comm.close()
            comm = None
            search = ShardedSearch(backend, lo, hi, comm=None)
    elapsed, idx, sc, timers, exchange_verified = measure(search)
    if exchange_verified is False and comm is not None:
        # the library's own exchange (icrec_search_sharded over ncclAllGather) has never run on more than one GPU before
        # this launch: if its result is wrong, say so, and measure the same kernels with the collectives through
        # torch.distributed (RCCL) instead of reporting nothing.  Every rank takes this branch together (MIN above).
        comm_note = ("icrec_search_sharded FAILED the check against the unsharded search on this run; re-measured with the "
                     "exchange through torch.distributed (RCCL)")
        comm.close()
        comm = None
        search = ShardedSearch(backend, lo, hi, comm=None)
        elapsed, idx, sc, timers, exchange_verified = measure(search)
    if exchange_verified is False:
        raise SystemExit(f"rank {rank}: sharded result differs from the unsharded search on the 64-query sample")
    (ffn_ms, ffn_n), (enc_ms, _), (srch_ms, _), (skern_ms, _) = timers
    n_keep = min(args.cpu_sample, args.batch)  # the timed step's own outputs, kept for the oracle check at the end
    kept = (emb[:n_keep].cpu().numpy(), idx[:n_keep].cpu().numpy(), sc[:n_keep].cpu().numpy())

    # ---- BASELINE configs[4] beside it, in the same line at every N: 4,096 user contexts per step in total (4,096 / N
    # encoded per GPU) against a 10 M x 384 bf16 catalog (+ filter fragments) row-sharded N ways, generated on the device
    # per shard.  north_star's ">= 6x at 8 GPUs on a 10M-row catalog" is the ratio of this leg's `qps` at N = 8 to its
    # value at N = 1 (strong scaling: the total work per step does not change with N).
    leg_10m = None
    if args.workload == "49k7" and not args.no_10m_leg and not (world == 1 and args.no_latency):
        q_total = 4096 - 4096 % world
        per_rank = q_total // world
        be10, lo10, hi10, _ = make_shard(args.rows_10m, True, "bf16+filter")
        search10 = ShardedSearch(be10, lo10, hi10, comm=comm)
        ids10_h, cu10_h = syn.synthetic_token_batch(per_rank, seed=4321 + rank)
        batch10 = (torch.from_numpy(ids10_h).to(dev), torch.from_numpy(cu10_h).to(dev), int(np.diff(cu10_h).max()),
                   torch.empty((per_rank, shape.hidden), device=dev))
        steps10, warm10 = max(1, min(args.steps, 10)), max(1, min(args.warmup, 2))
        el10, idx10, sc10, tm10, _ = measure(search10, batch10, steps10, warm10, verify=False)
        # size-independent properties of the result (the oracle cannot score 10 M rows in bench time): every list sorted
        # (score descending, lower row first on ties), rows inside the catalog, no row twice
        s_h, i_h = sc10[:256].cpu().numpy(), idx10[:256].cpu().numpy()
        d_s = np.diff(s_h, axis=1)
        props_ok = bool(idx10.shape[0] == q_total and (d_s <= 0).all() and ((d_s < 0) | (np.diff(i_h, axis=1) > 0)).all()
                        and (i_h >= 0).all() and (i_h < args.rows_10m).all()
                        and all(len(set(r.tolist())) == TOP_K for r in i_h))
        leg_10m = {"qps": q_total * steps10 / el10, "ms_per_step": el10 / steps10 * 1e3, "queries_per_step": q_total,
                   "contexts_encoded_per_gpu": per_rank, "tokens_per_gpu_per_step": int(cu10_h[-1]),
                   "catalog_rows": args.rows_10m, "rows_per_gpu": hi10 - lo10, "catalog_row_storage": "bf16+filter",
                   "steps": steps10, "warmup": warm10, "scaling": "strong",
                   "encode_ms_per_step": tm10[1][0], "search_ms_per_step": tm10[2][0],
                   "result_properties_ok": props_ok,
                   "note": "BASELINE configs[4]: the step of `value` at 4,096 queries in total over a 10 M-row bf16 catalog "
                           "(+ f16 filter fragments) row-sharded over the GPUs; same kernels, same exchange"}
        if not props_ok:
            raise SystemExit(f"rank {rank}: 10 M-row leg returned unsorted / out-of-range / duplicate rows")
        del search10, be10, batch10, idx10, sc10
        torch.cuda.empty_cache()

    # ---- the same step with the product's two-stream encode (DeviceEncoder splits the batch over two HIP
    # streams when it has the host copy of cu_seqlens, as recommend_batch does).  Reported separately: the
    # timed region above keeps ONE icrec_encode per step so that per-kernel event times and the rocprof
    # averages describe the same launches.
    two_stream = None
    if rank == 0 and world == 1 and not args.no_latency:
        def step2():
            enc.encode_packed(ids_d, cu_d, max_len, out=emb, cu_host=cu_h)
            return search.search(emb, TOP_K)
        for _ in range(3):
            step2()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(10):
            step2()
        torch.cuda.synchronize(dev)
        dt2 = (time.perf_counter() - t2) / 10
        two_stream = {"qps": args.batch / dt2, "ms_per_step": dt2 * 1e3}

    # ---- the same step replayed from a hipGraph: what the ~35 dependent dispatches of a step cost on
    # the host / command-processor side.  Not `value`: a serving batch has a different token count every time, and a graph
    # bakes it; reported so that the gap is a number.
    graph_leg = None
    if rank == 0 and world == 1 and not args.no_latency:
        try:
            g_idx = torch.empty((args.batch, TOP_K), dtype=torch.int64, device=dev)
            g_sc = torch.empty((args.batch, TOP_K), dtype=torch.float32, device=dev)
            g_side = torch.cuda.Stream(dev)

            def gstep():
                enc.encode_packed(ids_d, cu_d, max_len, out=emb)
                backend.index.search_into(emb, TOP_K, None, None, g_idx, g_sc)

            g_side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(g_side):
                for _ in range(2):
                    gstep()
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=g_side):
                    gstep()
                torch.cuda.synchronize(dev)
                t_e = time.perf_counter()
                for _ in range(10):
                    gstep()
                torch.cuda.synchronize(dev)
                t_g = time.perf_counter()
                for _ in range(10):
                    graph.replay()
                torch.cuda.synchronize(dev)
                t_end = time.perf_counter()
            same = bool(torch.equal(g_idx, idx)) if g_idx.shape == idx.shape else None
            graph_leg = {"ms_per_step_replayed": (t_end - t_g) / 10 * 1e3, "ms_per_step_kernel_by_kernel_same_loop": (t_g - t_e) / 10 * 1e3,
                         "qps_replayed": args.batch * 10 / (t_end - t_g), "result_identical_to_timed_step": same,
                         "note": "encode + search of the step's batch captured once and replayed 10 times, beside 10 eager steps "
                                 "in the same loop; a graph bakes the batch's token count, which is why `value` does not use one"}
            del graph
        except Exception as exc:  # noqa: BLE001 - a diagnostic leg must not take the line down
            graph_leg = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.current_stream(dev).wait_stream(g_side)

    # ---- the same step in the exact-f32 GEMM mode (every linear layer bit-identical to the oracle's fmaf
    # chains), measured in the same run on the same inputs, plus how far the default mode's embeddings and
    # top-20 lists are from it: the record carries a plain-fp32 number next to the f16x3 one.
    f32_leg = None
    if rank == 0 and world == 1 and not args.no_latency and enc.gemm_mode != "f32":
        enc32 = DeviceEncoder(weights, shape, dev, gemm_mode="f32")
        emb32 = torch.empty_like(emb)

        def step32():
            enc32.encode_packed(ids_d, cu_d, max_len, out=emb32)
            return search.search(emb32, TOP_K)
        for _ in range(2):
            i32, s32 = step32()
        torch.cuda.synchronize(dev)
        t32 = time.perf_counter()
        for _ in range(5):
            i32, s32 = step32()
        torch.cuda.synchronize(dev)
        dt32 = (time.perf_counter() - t32) / 5
        enc.encode_packed(ids_d, cu_d, max_len, out=emb)  # the default mode's embeddings again
        # queries whose ordered top-20 differs between the two modes: are they near-ties?  The two embeddings differ by
        # ~2e-7, so two catalog rows can swap only when their scores are closer than that; SURVEY 7.2.1 calls a list
        # AMBIGUOUS when the exact-mode scores of its top-21 hold a gap below 4e-6.  `unexplained` must be 0.
        differ = (~(i32 == idx).all(dim=1)).nonzero().flatten()
        n_amb = n_unexpl = 0
        if differ.numel():
            _, s21 = search.search(emb32[differ], TOP_K + 1)
            gaps = (s21[:, :-1] - s21[:, 1:]).min(dim=1).values
            n_amb = int((gaps < 4e-6).sum().item())
            n_unexpl = int(differ.numel()) - n_amb
            # and the stronger statement: the two lists hold the same rows up to swaps across such near-ties
            worst_swap = 0.0
            s32c, i32c, idc = s32[differ].cpu().numpy(), i32[differ].cpu().numpy(), idx[differ].cpu().numpy()
            for r in range(len(idc)):
                for pos in np.nonzero(i32c[r] != idc[r])[0]:
                    j = np.nonzero(i32c[r] == idc[r][pos])[0]
                    worst_swap = max(worst_swap, abs(float(s32c[r][pos] - s32c[r][j[0]])) if len(j) else
                                     abs(float(s32c[r][pos] - s32c[r][-1])))
        f32_leg = {"qps": args.batch / dt32, "ms_per_step": dt32 * 1e3,
                   "max_abs_embedding_diff_vs_default_mode": float((emb32 - emb).abs().max().item()),
                   "top20_lists_identical_to_default_mode": float((i32 == idx).all(dim=1).float().mean().item()),
                   "top20_lists_that_differ": {
                       "count": int(differ.numel()), "ambiguous_top21_gap_below_4e-6": n_amb, "unexplained": n_unexpl,
                       "largest_exact_mode_score_gap_across_a_swapped_pair": worst_swap if differ.numel() else 0.0,
                       "note": "exact-mode (f32) scores of the differing queries' top-21: a list counts as ambiguous when its "
                               "smallest adjacent gap is < 4e-6 (SURVEY 7.2.1); rows that changed place are that close in score"},
                   "note": "gemm_mode=f32: exact-f32 MFMA everywhere (157 TF roof); same inputs, same search"}
        del enc32, emb32

    # ---- single-request latency (Q = 1): what Recommender.recommend() does per request, host-timed
    # from token ids in host memory to k results in host memory.  (a) hipGraph replay (fastpath.py,
    # the product path), (b) the same two library calls launched kernel by kernel.
    p50_ms = p50_plain_ms = None
    if rank == 0 and world == 1 and not args.no_latency:
        from instacart_next_order_recommendation_amd.fastpath import SingleRequestPath

        n_tok = int(cu_h[1])
        one = ids_h[:n_tok].tolist()
        fast = SingleRequestPath(enc, backend.index)
        lat = []
        for i in range(110):
            a = time.perf_counter()
            fast.run(one, TOP_K)
            lat.append((time.perf_counter() - a) * 1e3)
        p50_ms = float(np.median(lat[10:]))
        lat = []
        one_ids, one_cu = ids_d[:n_tok], cu_d[:2]
        one_emb = torch.empty((1, shape.hidden), device=dev)
        for i in range(60):
            torch.cuda.synchronize(dev)
            a = time.perf_counter()
            enc.encode_packed(one_ids, one_cu, n_tok, out=one_emb)
            i1, s1 = search.search(one_emb, TOP_K)
            i1.cpu()
            lat.append((time.perf_counter() - a) * 1e3)
        p50_plain_ms = float(np.median(lat[10:]))

    # ---- catalog index build (start-up path, BASELINE configs[1]): encode 49,688 product texts
    # (token-packed, lengths ~ clipped N(20,5) as SURVEY.md §8d) and build the normalised index
    index_build_ms = None
    if rank == 0 and world == 1 and args.workload == "49k7" and not args.no_latency:
        from instacart_next_order_recommendation_amd.search import DeviceIndex

        cat_ids, cat_cu = syn.synthetic_token_batch(CATALOG_ROWS, seed=42, mean_len=20, std_len=5, lo=8, hi=40)
        ci, cc = torch.from_numpy(cat_ids).to(dev), torch.from_numpy(cat_cu).to(dev)
        cat_emb = torch.empty((CATALOG_ROWS, shape.hidden), device=dev)

        def build_index():
            step_rows = 8192
            for s0 in range(0, CATALOG_ROWS, step_rows):
                s1 = min(CATALOG_ROWS, s0 + step_rows)
                t0_, t1_ = int(cat_cu[s0]), int(cat_cu[s1])
                enc.encode_packed(ci[t0_:t1_], (cc[s0:s1 + 1] - t0_).contiguous(), 40, out=cat_emb[s0:s1])
            return DeviceIndex(cat_emb, dev)

        build_index().close()
        torch.cuda.synchronize(dev)
        a = time.perf_counter()
        ix_tmp = build_index()
        torch.cuda.synchronize(dev)
        index_build_ms = (time.perf_counter() - a) * 1e3
        ix_tmp.close()
        catalog_tokens = int(cat_cu[-1])

    # ---- the same step from TEXT in host memory (not `value`): native tokenizer on the host cores, one
    # H2D of the packed ids, encode + search, D2H of the results — the PCIe- and tokeniser-inclusive rate
    text_path = strings_pipelined = None
    if rank == 0 and world == 1 and args.workload == "49k7" and not args.no_latency:
        import tempfile

        from instacart_next_order_recommendation_amd.model_io import NativeTokenizer

        vdir = Path(tempfile.mkdtemp(prefix="icrec_vocab_"))
        (vdir / "vocab.txt").write_text("\n".join(syn.synthetic_vocab()) + "\n")
        tok = NativeTokenizer(vdir / "vocab.txt", True, 256)
        # heavier users than the reference's 20-item cap so that the text workload matches `value`'s: ~128 tokens
        texts = syn.synthetic_user_contexts(args.batch, seed=1234, max_items=36, min_orders=3, max_orders=8, per_order=8)

        def text_step():
            t_a = time.perf_counter()
            ids_t, cu_t = tok.packed(texts)  # the product's form (SbertModel.encode_to_device): packed ids, no Python lists
            t_b = time.perf_counter()
            e = enc.encode_packed_host(ids_t, cu_t)
            i_t, s_t = search.search(e, TOP_K)
            i_t.cpu(); s_t.cpu()
            return t_b - t_a, int(cu_t[-1])

        text_step()
        reps, tok_s = 5, 0.0
        t_a = time.perf_counter()
        for _ in range(reps):
            dt, n_tok_text = text_step()
            tok_s += dt
        wall = time.perf_counter() - t_a
        # the same steps through the product's two-deep pipeline (pipeline.py, Recommender.recommend_batches):
        # batch i+1 is tokenised on a worker thread while the GPU works on batch i.  Timed over --steps batches
        # (warm-up first), cycling through FOUR different batches of contexts so that no call sees the previous
        # call's strings, ids or sequence boundaries again.
        from instacart_next_order_recommendation_amd.pipeline import pipelined_search

        text_batches = [texts] + [syn.synthetic_user_contexts(args.batch, seed=4321 + b, max_items=36, min_orders=3,
                                                               max_orders=8, per_order=8) for b in range(3)]
        n_pipe = max(args.steps, 8)
        srch = lambda e, k, ex: search.search(e, k)  # noqa: E731
        list(pipelined_search(tok, enc, srch, [text_batches[i % 4] for i in range(max(args.warmup, 2))], TOP_K))
        t_p = time.perf_counter()
        n_out = sum(r[0].shape[0] for r in pipelined_search(tok, enc, srch, (text_batches[i % 4] for i in range(n_pipe)), TOP_K))
        wall_p = time.perf_counter() - t_p
        assert n_out == args.batch * n_pipe
        pipe_tokens = [int(tok.packed(b)[1][-1]) for b in text_batches]
        text_path = {"qps": args.batch * reps / wall, "ms_per_step": wall / reps * 1e3,
                     "tokenize_ms_per_step": tok_s / reps * 1e3, "tokens_per_step": n_tok_text,
                     "host_threads": "min(CPU quota, texts/64) worker threads inside icrec_tokenize", "host_cpus_visible": os.cpu_count(),
                     "mean_tokens_per_context": n_tok_text / args.batch,
                     "note": "synthetic user-context STRINGS (~128 tokens each, like the token-id batches `value` is "
                             "measured on) -> native WordPiece on the host -> H2D -> encode -> "
                             "search -> D2H, strictly serial (no overlap of tokenisation with GPU work); 5 steps"}
        strings_pipelined = {"qps": n_out / wall_p, "ms_per_step": wall_p / n_pipe * 1e3, "steps": n_pipe,
                             "warmup": max(args.warmup, 2), "distinct_batches": 4,
                             "mean_tokens_per_context": float(np.mean(pipe_tokens)) / args.batch,
                             "fraction_of_ids_resident_value": (n_out / wall_p) / (args.batch * args.steps / elapsed),
                             "note": "north_star's workload: synthetic user-context STRINGS in host memory -> top-20 rows "
                                     "and scores in host memory, through Recommender.recommend_batches' pipeline "
                                     "(pipeline.py): native WordPiece of batch i+1 on a worker thread under the GPU work "
                                     "of batch i, results read back from pinned buffers after the next launch"}

    # ---- single request from a STRING through the serving objects themselves (what /recommend constructs:
    # MonitoredRecommender, reference src/api/main.py:73): tokenise -> replayed hipGraph(s) -> k results on the host
    p50_rec_ms = p50_mon_ms = mon_fields = None
    if rank == 0 and world == 1 and args.workload == "49k7" and not args.no_latency:
        import tempfile

        from instacart_next_order_recommendation_amd.model_io import write_synthetic_model_dir
        from instacart_next_order_recommendation_amd.recommender import MonitoredRecommender

        root = Path(tempfile.mkdtemp(prefix="icrec_bench_rec_"))
        mdir = write_synthetic_model_dir(root / "model", seed=0)
        (root / "processed").mkdir()
        cpath = root / "processed" / "eval_corpus.json"
        cpath.write_text(json.dumps(syn.synthetic_catalog(CATALOG_ROWS)))
        logging_off = __import__("logging").getLogger("recommender.metrics")
        logging_off.setLevel(__import__("logging").WARNING)
        mon = MonitoredRecommender(mdir, cpath, use_index=False)
        one_text = syn.synthetic_user_contexts(4, seed=77, max_items=28, min_orders=3, max_orders=6, per_order=7)[1]

        def p50_of(fn, n=110):
            lat = []
            for _ in range(n):
                a = time.perf_counter()
                fn()
                lat.append((time.perf_counter() - a) * 1e3)
            return float(np.median(lat[10:]))

        from instacart_next_order_recommendation_amd.recommender import Recommender
        p50_rec_ms = p50_of(lambda: Recommender.recommend(mon, one_text, TOP_K))
        p50_mon_ms = p50_of(lambda: mon.recommend(one_text, TOP_K, user_id="bench"))
        same = Recommender.recommend(mon, one_text, TOP_K) == mon.recommend(one_text, TOP_K)
        m = mon.last_metrics
        mon_fields = {"tokens": len(mon.model.tokenizer([one_text])[0]), "results_equal_to_plain_recommend": bool(same),
                      "query_embedding_time_ms": m.query_embedding_time_ms,
                      "similarity_compute_time_ms": m.similarity_compute_time_ms, "total_latency_ms": m.total_latency_ms,
                      "note": "MonitoredRecommender.recommend(str): host WordPiece + two replayed hipGraphs (encode | search) "
                              "bracketed by HIP events (the three timing fields) + D2H of k results; catalog of 49,688 "
                              "synthetic product texts encoded at construction"}

    if rank == 0:
        q_per_step = args.batch * world
        ms_per_step = elapsed / args.steps * 1e3
        import ctypes as C

        t_main, t_tail = C.c_int64(total_tokens), C.c_int64(0)
        if enc.gemm_mode != "f32":  # the timed launches (slot 1) are the fused FFN kernel over the batch-kernel share of the tokens
            _native.check(_native.lib().icrec_encode_batch_split(enc._h, total_tokens, C.byref(t_main), C.byref(t_tail)),
                          "icrec_encode_batch_split")
        ffn_tokens = int(t_main.value)
        # algorithmic FLOPs per timed launch: FFN-up only in f32 mode; attention-out + FFN-up + FFN-down in the fused kernel
        # (+ the next layer's QKV projection in all but the last layer's launch: the figure is the average over a step's launches)
        n_l = shape.layers
        ffn_flops = 2.0 * ffn_tokens * shape.hidden * shape.intermediate if enc.gemm_mode == "f32" else \
            ffn_tokens * (4.0 * shape.hidden * shape.intermediate + 2.0 * shape.hidden * shape.hidden
                          + (n_l - 1) / n_l * 2.0 * shape.hidden * 3 * shape.hidden)
        achieved = ffn_flops / (ffn_ms * 1e-3) / 1e12 if ffn_ms > 0 else 0.0
        out = {
            "metric": "recommend_qps_top20_49k7_catalog" if args.workload == "49k7" else "recommend_qps_top20_10m_catalog",
            "value": q_per_step * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if (world > 1 and args.batch_defaulted and args.workload == "49k7") else "weak",
            "vs_baseline": None,
            "dtype": "f32" if enc.gemm_mode == "f32" else "f16x3 (f32 operands split into 2 f16 planes, 3 f16 MFMAs per product, f32 accumulate; fp32-level accuracy)",
            "data": "synthetic",
            "config": {
                "workload": (("BASELINE configs[1]/[2] with the token ids of the step's contexts already resident in HBM (from "
                              "user-context STRINGS in host memory: the key from_strings_pipelined; over HTTP: http_qps): "
                              f"encode + cos_sim + top-20 over the 49,688-row catalog, {args.batch} user contexts per GPU per step") if world == 1 else
                             (f"BASELINE configs[3]: 49,688-row catalog row-sharded {world}-way, {q_per_step}-query batch "
                              f"({args.batch} contexts encoded per GPU), RCCL all-gather of query embeddings and of "
                              "per-shard partial top-20 lists, merged on every rank")) if args.workload == "49k7" else
                            f"BASELINE configs[4]: 10M x 384 synthetic catalog kept as {row_storage} rows, row-sharded",
                "catalog_row_storage": row_storage,
                "catalog_rows": n_rows, "dim": shape.hidden, "top_k": TOP_K,
                "contexts_per_gpu_per_step": args.batch, "tokens_per_gpu_per_step": total_tokens,
                "mean_tokens_per_context": total_tokens / args.batch, "max_tokens": max_len,
                "encoder": "all-MiniLM-L6-v2 shape (6 layers, hidden 384, 12 heads, ffn 1536), seeded random weights",
                "parallelism": (f"catalog row-sharded x{world}, queries data-parallel, 2 ncclAllGather per step issued by "
                                f"libicrec (icrec_search_sharded){' [gloo rehearsal through torch.distributed]' if rehearsal else ''}")
                if world > 1 else "single GPU",
            },
            "rehearsal_not_a_measurement": True if rehearsal else None,
            "configs4_10m_rows": leg_10m,
            "http": http, "http_qps": None if not http else http.get("http_qps"),
            "http_p50_ms": None if not http else http.get("http_p50_ms"),
            "exchange_verified": exchange_verified,
            "exclusion_exchange_verified": verified_excl[0], "exclusion_exchange_error": verified_excl[1],
            "exchange_verified_note": None if exchange_verified is None else
            ("every rank: 64-query sample of the gathered batch searched against a replicated unsharded index, indices and scores "
             "bit-equal (false: the run is re-measured with the exchange through torch.distributed, or fails); "
             "exclusion_exchange_verified: the same sample again with per-rank exclusion lists (each local query's own top-3 rows, "
             "every fifth list empty) through the exclusion exchange against the unsharded search with the gathered lists"),
            "exchange": None if world == 1 else ("icrec_search_sharded (RCCL inside libicrec)" if comm is not None else
                                                 (comm_note or "torch.distributed collectives (gloo rehearsal)")),
            "p50_latency_ms_single_request": p50_ms,
            "p50_latency_ms_single_request_without_hipgraph": p50_plain_ms,
            "single_request_tokens": int(cu_h[1]),
            "with_two_stream_encode": two_stream,
            "with_hipgraph_replay": graph_leg,
            "exact_f32_gemm_mode": f32_leg,
            "from_text_in_host_memory": text_path,
            "from_strings_pipelined": strings_pipelined,
            "p50_latency_ms_recommend_from_string": p50_rec_ms,
            "p50_latency_ms_monitored_recommend_from_string": p50_mon_ms,
            "monitored_recommend": mon_fields,
            "catalog_index_build_ms": index_build_ms,
            "catalog_index_build_note": None if index_build_ms is None else
            f"encode {CATALOG_ROWS} products ({catalog_tokens} tokens, from ids in HBM) + normalise into a DeviceIndex",
            "encode_ms_per_step": enc_ms, "search_ms_per_step": srch_ms, "search_kernel_ms": skern_ms,
            "roofline": roofline(enc.gemm_mode, achieved, ffn_n, ffn_ms, ffn_flops, ffn_tokens),
        }
        # the whole step against the same roof: SURVEY 8d's algorithmic FLOPs (encoder 21,233,664 L + 9,216 L^2 per
        # sequence of L tokens, similarity 2 Q N d; split-precision products count once) / step time / the MFMA roof
        lens = np.diff(cu_h).astype(np.float64)
        per_tok = n_l * 2.0 * (4 * shape.hidden ** 2 + 2 * shape.hidden * shape.intermediate)
        step_flops = world * (float((per_tok * lens + n_l * 4.0 * shape.hidden * lens ** 2).sum())) \
            + 2.0 * q_per_step * n_rows * shape.hidden
        roof = out["roofline"]["peak"]
        out["roofline"]["whole_step_flops"] = step_flops
        out["roofline"]["whole_step_frac"] = step_flops / (ms_per_step * 1e-3) / 1e12 / (roof * world)
        out["roofline"]["whole_step_note"] = ("algorithmic encoder + similarity FLOPs of one step (SURVEY 8d) / ms_per_step / "
                                             "the same MFMA roof (x n_gpus); attention and search run partly on other "
                                             "instructions, the roof is the one of the dominant kernel")
        failed_check = None
        if not args.no_cpu_baseline and world == 1 and args.workload == "49k7":
            n_chk = min(args.cpu_sample, args.batch)
            out["cpu_baseline"] = cpu_baseline(weights, shape, ids_h, cu_h, catalog, n_chk)
            # ---- the line certifies itself: the embeddings and top-20 lists the TIMED step left in HBM for the first
            # n_chk contexts, against the C oracle's for the same contexts (computed for the CPU baseline anyway).
            # Embeddings: the f16x3 bar of the parity tests (5e-6 on unit-norm rows).  Lists: identical rows in
            # identical order, or the oracle's own top-21 scores of that query hold an adjacent gap below 4e-6 (two rows
            # that close may legitimately swap when the two embeddings differ by ~2e-7: AMBIGUOUS);
            # scores within 1e-5 of the oracle's at every position either way.  Anything else fails the run.
            orc = out["cpu_baseline"].pop("_oracle")
            g_emb, g_idx, g_sc = kept
            emb_diff = float(np.abs(g_emb - orc["emb"]).max())
            same_list = (g_idx == orc["idx"]).all(axis=1)
            gaps = (orc["score21"][:, :-1] - orc["score21"][:, 1:]).min(axis=1)
            ambiguous = (~same_list) & (gaps < 4e-6)
            mismatched = (~same_list) & ~(gaps < 4e-6)
            sc_diff = float(np.abs(g_sc - orc["score"]).max())
            ok = bool(emb_diff < 5e-6 and not mismatched.any() and sc_diff < 1e-5)
            out["checked_vs_oracle"] = {
                "n": int(n_chk), "max_abs_emb_diff": emb_diff, "top20_identical": int(same_list.sum()),
                "ambiguous": int(ambiguous.sum()), "mismatched": int(mismatched.sum()), "max_abs_score_diff": sc_diff, "ok": ok,
                "note": "the timed step's own outputs (emb / idx / score of its first n contexts, read back after the timed "
                        "region) against oracle/icrec_oracle.c on the same contexts; ambiguous = lists differ and the oracle's "
                        "top-21 scores of that query hold an adjacent gap < 4e-6; bars: emb 5e-6, scores 1e-5, mismatched 0"}
            if not ok:
                failed_check = f"timed step disagrees with the oracle: {out['checked_vs_oracle']}"
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        if failed_check:
            raise SystemExit(failed_check)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
