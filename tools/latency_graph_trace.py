#!/usr/bin/env python3
"""Single-request latency anatomy on the PRODUCT path: N replays of SingleRequestPath's hipGraph (encode + search of one
99-token request) for `rocprofv3 --kernel-trace`, and (second mode) the summary of such a trace: per kernel of a replay,
in launch order, the median duration and the median gap to the end of the previous kernel.
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/latency_graph_trace.py run [n_tokens]
       python3 tools/latency_graph_trace.py summary OUT"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def run(n_tok: int) -> None:
    import time
    import numpy as np, torch
    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.encoder import DeviceEncoder
    from instacart_next_order_recommendation_amd.fastpath import SingleRequestPath
    from instacart_next_order_recommendation_amd.search import DeviceIndex

    shape = syn.BertShape()
    enc = DeviceEncoder(syn.synthetic_bert_weights(shape, seed=0), shape)
    ix = DeviceIndex(syn.synthetic_embeddings(49688, 384, seed=1), storage="f32+filter")
    ids, _ = syn.synthetic_token_batch(1, seed=5, mean_len=n_tok, std_len=0, lo=n_tok, hi=n_tok)
    fast = SingleRequestPath(enc, ix)
    one = ids.tolist()
    lat = []
    for _ in range(120):
        a = time.perf_counter()
        fast.run(one, 20)
        lat.append((time.perf_counter() - a) * 1e3)
    # the same replay between two events (GPU-side span of the graph alone) and the host cost of the pieces of run()
    cap = next(iter(fast._graphs.values()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    span, t_replay, t_sync = [], [], []
    st = torch.cuda.current_stream()
    for _ in range(60):
        cap.d_in.copy_(cap.h_in, non_blocking=True)
        e0.record(st)
        a = time.perf_counter()
        cap.graph.replay()
        b = time.perf_counter()
        e1.record(st)
        st.synchronize()
        c = time.perf_counter()
        span.append(e0.elapsed_time(e1)); t_replay.append((b - a) * 1e3); t_sync.append((c - b) * 1e3)
    print(f"p50 host-to-host {np.median(lat[20:]):.4f} ms over {len(lat) - 20} replays ({n_tok} tokens); graph between events "
          f"{np.median(span[10:]):.4f} ms; host time inside graph.replay() {np.median(t_replay[10:]):.4f} ms, then waiting "
          f"{np.median(t_sync[10:]):.4f} ms")


def summary(out: str) -> None:
    import csv, glob
    import numpy as np

    f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "embed_ln" in r["Kernel_Name"]]
    its = [rows[a:b] for a, b in zip(starts[-60:-1], starts[-59:])]
    n = min(len(it) for it in its)
    its = [it for it in its if len(it) == n]
    dur = np.array([[(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in it] for it in its])
    gap = np.array([[0.0] + [(int(it[i]["Start_Timestamp"]) - int(it[i - 1]["End_Timestamp"])) / 1e3 for i in range(1, n)] for it in its])
    span = np.array([(int(it[-1]["End_Timestamp"]) - int(it[0]["Start_Timestamp"])) / 1e3 for it in its])
    print(f"{len(its)} replays of {n} kernels; first kernel start -> last kernel end: median {np.median(span):.1f} us; "
          f"sum of kernel durations {np.median(dur.sum(1)):.1f} us, sum of gaps {np.median(gap.sum(1)):.1f} us")
    for i in range(n):
        name = its[0][i]["Kernel_Name"].split("(")[0][-70:]
        print(f"{i:3d} {np.median(dur[:, i]):7.2f} us  gap {np.median(gap[:, i]):6.2f}  grid {its[0][i].get('Grid_Size_X', '?'):>7}  {name}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    else:
        summary(sys.argv[2])
