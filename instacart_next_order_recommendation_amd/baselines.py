"""Batched ranking consumer: the reference's ContentBasedBaseline.rank_all
(src/baselines/content_based.py:16-64; the same shape is used by
scripts/compare_untrained_vs_trained.py:38-85).

The reference encodes all eval queries, builds the dense [13,120 x 49,688] fp32 score matrix on
the host (2.6 GB) and fully argsorts every row, although its metrics
(src/baselines/metrics.py:122-176: Accuracy@1/3/5/10, Recall@10, MRR@10, NDCG@10, MAP@100) never
look past rank 100.  Here queries stream through the GPU in passes of `queries_per_pass`.
`rank_all()` returns the reference's complete order (icrec_rank_all: exact score rows sorted on the
device per pass); `rank_all(depth=100)` is the fast form for the metrics: one encode + one fused
score/top-`depth` search per pass, no score matrix at all.
"""
from __future__ import annotations

from pathlib import Path

import torch

from . import _native
from .model_io import load_model_dir
from .recommender import SbertModel
from .search import DeviceIndex


class ContentBasedBaseline:
    """Same constructor and `rank_all` contract as the reference class, with `model_name` a LOCAL
    SentenceTransformer directory (no hub access)."""

    def __init__(self, eval_queries: dict[str, str], eval_corpus: dict[str, str], model_name: str | Path,
                 batch_size: int = 64, device: str | torch.device = "cuda:0"):
        self.eval_queries = eval_queries
        self.eval_corpus = eval_corpus
        self.product_ids = list(eval_corpus.keys())
        self.corpus_texts = [eval_corpus[pid] for pid in self.product_ids]
        self.device = torch.device(device)
        self.model = SbertModel(load_model_dir(model_name), self.device)
        self.corpus_embeddings = self.model.encode(self.corpus_texts, batch_size=batch_size,
                                                   show_progress_bar=True, normalize_embeddings=True)
        self._index = DeviceIndex(self.corpus_embeddings, self.device)

    def rank_all(self, depth: int | None = None, queries_per_pass: int | None = None) -> dict[str, list[str]]:
        """query_id -> product ids by cosine similarity, best first (score desc, then lower corpus row first
        on exact ties).  depth=None: every product, like the reference (content_based.py:58-63);
        1 <= depth <= 128: only the best `depth` (what the metrics read), through the fused search."""
        full = depth is None
        if not full and not 1 <= depth <= _native.ICREC_MAX_K:
            raise ValueError(f"depth must be None (full order) or in [1, {_native.ICREC_MAX_K}]")
        if queries_per_pass is None:
            queries_per_pass = 256 if full else 1024
        depth = len(self.product_ids) if full else min(depth, len(self.product_ids))
        query_ids = list(self.eval_queries.keys())
        out: dict[str, list[str]] = {}
        for s in range(0, len(query_ids), queries_per_pass):
            qids = query_ids[s:s + queries_per_pass]
            emb = self.model.encode_to_device([self.eval_queries[q] for q in qids])
            idx = self._index.rank_all(emb) if full else self._index.search(emb, depth)[0]
            idx = idx.cpu().numpy()
            for i, qid in enumerate(qids):
                out[qid] = [self.product_ids[j] for j in idx[i] if j >= 0]
        return out
