"""The reference's per-request op sequence on torch CPU kernels — bench.py's `cpu_baseline` (kind "torch").

TEST / MEASUREMENT INFRASTRUCTURE ONLY: imported by bench.py's cpu_baseline leg and by tests/, never by
the product package.

The reference's module cannot be imported here (sentence_transformers / dotenv are not installed, SURVEY.md
§8c), so the calls it makes are restated op for op with `torch.nn.functional`, which is what those libraries
dispatch to on CPU:

    Recommender.recommend()                        src/inference/serve_recommendations.py:206-225
      model.encode([query], normalize_embeddings=True)                                     :213
          tokenise (host; the caller passes token ids), sort by length, pad per batch of 64,
          BertModel forward (transformers modeling_bert.py: BertEmbeddings :68-108,
          BertSelfAttention + sdpa :111-136,:164-203, BertSelfOutput :289-293, BertIntermediate :334-337
          erf-GELU, BertOutput :347-351), ST Pooling(mean, clamp 1e-9), ST Normalize, F.normalize
      cos_sim(query_emb, product_embeddings)       F.normalize(a), F.normalize(b), torch.mm       :214
      scores.argsort(descending=True)                                                       :215
      Python exclusion / top-k loop                                                         :216-225

Weights: the same seeded blob the GPU path and the C oracle use (include/icrec.h order).
"""
from __future__ import annotations

import os
import time
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


class TorchCpuSbert:
    """BertModel(6 x post-LN) + mean-pool + Normalize with torch CPU ops, padded batches as
    SentenceTransformer.encode builds them."""

    def __init__(self, blob: np.ndarray, shape, n_threads: Optional[int] = None):
        from instacart_next_order_recommendation_amd import synthetic as syn  # data generator only (no kernels)

        if n_threads:
            torch.set_num_threads(int(n_threads))
        self.shape = shape
        sd = syn.blob_to_state_dict(np.asarray(blob, np.float32), shape)
        self.p = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
        self.heads = shape.heads
        self.eps = float(shape.ln_eps)

    @torch.no_grad()
    def forward_padded(self, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """ids int64 [B, L], mask [B, L] (1 = token) -> sentence embeddings [B, H] after Pooling + Normalize."""
        p, H = self.p, self.shape.hidden
        B, L = ids.shape
        x = F.embedding(ids, p["embeddings.word_embeddings.weight"])
        x = x + p["embeddings.token_type_embeddings.weight"][0]
        x = x + p["embeddings.position_embeddings.weight"][:L]
        x = F.layer_norm(x, (H,), p["embeddings.LayerNorm.weight"], p["embeddings.LayerNorm.bias"], self.eps)
        # additive key mask, as transformers builds it for sdpa: 0 for tokens, -inf-like for padding
        amask = torch.zeros((B, 1, 1, L), dtype=x.dtype)
        amask.masked_fill_(mask[:, None, None, :] == 0, torch.finfo(x.dtype).min)
        dh = H // self.heads
        for l in range(self.shape.layers):
            q = f"encoder.layer.{l}."
            def lin(t, name):
                return F.linear(t, p[q + name + ".weight"], p[q + name + ".bias"])
            qh = lin(x, "attention.self.query").view(B, L, self.heads, dh).transpose(1, 2)
            kh = lin(x, "attention.self.key").view(B, L, self.heads, dh).transpose(1, 2)
            vh = lin(x, "attention.self.value").view(B, L, self.heads, dh).transpose(1, 2)
            ctx = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=amask)  # scale = dh ** -0.5
            ctx = ctx.transpose(1, 2).reshape(B, L, H)
            a = lin(ctx, "attention.output.dense")
            x = F.layer_norm(a + x, (H,), p[q + "attention.output.LayerNorm.weight"],
                             p[q + "attention.output.LayerNorm.bias"], self.eps)
            h = F.gelu(lin(x, "intermediate.dense"))  # exact erf form (ACT2FN["gelu"])
            o = lin(h, "output.dense")
            x = F.layer_norm(o + x, (H,), p[q + "output.LayerNorm.weight"], p[q + "output.LayerNorm.bias"], self.eps)
        m = mask.to(x.dtype).unsqueeze(-1)
        pooled = (x * m).sum(1) / m.sum(1).clamp(min=1e-9)  # ST Pooling(mean)
        return F.normalize(pooled, p=2, dim=1)               # ST Normalize module

    @torch.no_grad()
    def encode(self, seqs: Sequence[Sequence[int]], batch_size: int = 64, normalize_embeddings: bool = True) -> np.ndarray:
        """SentenceTransformer.encode on token-id lists: length-sorted, batches of `batch_size`, padded to the
        longest of each batch, original order restored."""
        order = np.argsort([-len(s) for s in seqs], kind="stable")
        out = np.empty((len(seqs), self.shape.hidden), np.float32)
        for s0 in range(0, len(seqs), batch_size):
            sel = order[s0:s0 + batch_size]
            L = max(len(seqs[i]) for i in sel)
            ids = torch.zeros((len(sel), L), dtype=torch.int64)
            mask = torch.zeros((len(sel), L), dtype=torch.int64)
            for r, i in enumerate(sel):
                n = len(seqs[i])
                ids[r, :n] = torch.as_tensor(np.asarray(seqs[i], np.int64))
                mask[r, :n] = 1
            e = self.forward_padded(ids, mask)
            if normalize_embeddings:
                e = F.normalize(e, p=2, dim=1)
            out[sel] = e.numpy()
        return out


@torch.no_grad()
def cos_sim(a, b) -> torch.Tensor:
    """sentence_transformers.util.cos_sim: to tensor, unsqueeze 1-D, F.normalize both, mm."""
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    if a.dim() == 1:
        a = a.unsqueeze(0)
    if b.dim() == 1:
        b = b.unsqueeze(0)
    return torch.mm(F.normalize(a, p=2, dim=1), F.normalize(b, p=2, dim=1).transpose(0, 1))


@torch.no_grad()
def recommend(model: TorchCpuSbert, ids: Sequence[int], product_embeddings: np.ndarray, product_ids: Sequence[str],
              top_k: int = 10, exclude_product_ids: Optional[set] = None):
    """Recommender.recommend (serve_recommendations.py:206-225) for one tokenised query."""
    query_emb = model.encode([ids], normalize_embeddings=True)[0]
    scores = cos_sim(query_emb, product_embeddings)[0]
    indices = scores.argsort(descending=True)
    excluded = exclude_product_ids or set()
    results = []
    for idx in indices:
        pid = product_ids[idx]
        if pid in excluded:
            continue
        results.append((pid, float(scores[idx])))
        if len(results) >= top_k:
            break
    return results


def cgroup_cpu_quota() -> Optional[float]:
    """CPUs this container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def measure(blob, shape, ids: np.ndarray, cu: np.ndarray, catalog: np.ndarray, n_sample: int, top_k: int,
            n_threads: int) -> dict:
    """Time the reference's CPU path on a bounded sample: (a) one request at a time over the full catalog
    (what the reference serves), (b) the same against a 1,000-product subset (BASELINE configs[0]),
    (c) batched: model.encode(batch_size=64) over the sample + one cos_sim/argsort/top-k per query."""
    model = TorchCpuSbert(blob, shape, n_threads)
    seqs = [ids[cu[i]:cu[i + 1]].tolist() for i in range(n_sample)]
    pids = [str(i) for i in range(catalog.shape[0])]
    recommend(model, seqs[0], catalog[:1000], pids[:1000], top_k)  # warm the thread pool and kernels
    one, sub = [], []
    for r in range(7):
        a = time.perf_counter()
        recommend(model, seqs[r], catalog, pids, top_k)
        one.append((time.perf_counter() - a) * 1e3)
    for r in range(7):
        a = time.perf_counter()
        recommend(model, seqs[r], catalog[:1000], pids[:1000], top_k)
        sub.append((time.perf_counter() - a) * 1e3)
    t0 = time.perf_counter()
    emb = model.encode(seqs, batch_size=64, normalize_embeddings=True)
    t1 = time.perf_counter()
    for i in range(n_sample):
        scores = cos_sim(emb[i], catalog)[0]
        indices = scores.argsort(descending=True)
        res = []
        for idx in indices:
            res.append((pids[idx], float(scores[idx])))
            if len(res) >= top_k:
                break
    t2 = time.perf_counter()
    return {"emb": emb, "encode_s": t1 - t0, "rank_s": t2 - t1, "single_request_p50_ms": float(np.median(one)),
            "configs0_p50_ms": float(np.median(sub)), "torch_threads": torch.get_num_threads(),
            "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota": cgroup_cpu_quota()}
