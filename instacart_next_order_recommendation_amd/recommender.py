"""Drop-in serving objects: Recommender / MonitoredRecommender / EmbeddingIndex.

Same constructor arguments, attributes, return types and error behaviour as the
reference's src/inference/serve_recommendations.py (class Recommender :133-225,
MonitoredRecommender :228-293, EmbeddingIndex :66-130, RecommendationMetrics :52-63),
with the arithmetic moved onto the GPU behind libicrec's C ABI:

    model.encode([query])        -> DeviceEncoder   (icrec_encode)
    cos_sim + argsort + filter   -> DeviceIndex     (icrec_search)
    model.encode(product_texts)  -> DeviceEncoder over the whole catalog at start-up

`recommend_batch` is the one addition: many contexts in one GPU pass (what a
micro-batching server calls); `recommend` is its single-query form.
"""
from __future__ import annotations

import hashlib
import json
import logging
import os
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native
from .encoder import DeviceEncoder
from .model_io import LoadedModel, load_model_dir
from .search import DeviceIndex

logger = logging.getLogger(__name__)

# on-disk cache names (reference: src/constants.py:88-92)
INDEX_SUBDIR = ".embedding_index"
MANIFEST_FILENAME = "manifest.json"
EMBEDDINGS_FILENAME = "embeddings.npy"
PRODUCT_IDS_FILENAME = "product_ids.json"


@dataclass
class RecommendationMetrics:
    """Per-request metrics (same seven fields as the reference, :52-63)."""

    user_id: str
    query_embedding_time_ms: float
    similarity_compute_time_ms: float
    total_latency_ms: float
    num_recommendations: int
    top_score: float
    avg_score: float
    timestamp: float


class EmbeddingIndex:
    """On-disk cache of the [N, 384] fp32 product matrix, byte-compatible with the reference's
    (:66-130): <corpus dir>/.embedding_index/<sha256("model_dir|corpus_path")[:16]>/
    {manifest.json, embeddings.npy, product_ids.json}; valid only for the same corpus path,
    model dir, corpus mtime and product-id list."""

    def __init__(self, corpus_path: Path, model_dir: Path | str):
        self.corpus_path = Path(corpus_path).resolve()
        self.model_dir = model_dir
        digest = hashlib.sha256(f"{self.model_dir!s}|{self.corpus_path!s}".encode()).hexdigest()
        self._dir = self.corpus_path.parent / INDEX_SUBDIR / digest[:16]

    @property
    def directory(self) -> Path:
        return self._dir

    def _manifest_ok(self) -> bool:
        try:
            meta = json.loads((self._dir / MANIFEST_FILENAME).read_text())
            return (meta.get("corpus_path") == str(self.corpus_path)
                    and meta.get("model_dir") == str(self.model_dir)
                    and meta.get("corpus_mtime") == self.corpus_path.stat().st_mtime)
        except (OSError, ValueError):
            return False

    def load(self, product_ids: list[str]) -> Optional[np.ndarray]:
        """Cached matrix, or None on any mismatch / missing file."""
        if not self._manifest_ok():
            return None
        try:
            emb = np.load(self._dir / EMBEDDINGS_FILENAME)  # allow_pickle stays False
            cached_ids = json.loads((self._dir / PRODUCT_IDS_FILENAME).read_text())
        except (OSError, ValueError):
            return None
        if cached_ids != product_ids or len(emb) != len(product_ids):
            return None
        return emb

    def save(self, product_ids: list[str], embeddings: np.ndarray) -> None:
        self._dir.mkdir(parents=True, exist_ok=True)
        try:
            mtime = self.corpus_path.stat().st_mtime
        except OSError:
            mtime = 0
        (self._dir / MANIFEST_FILENAME).write_text(json.dumps({
            "corpus_path": str(self.corpus_path), "model_dir": str(self.model_dir),
            "corpus_mtime": mtime, "n_products": len(product_ids)}, indent=2))
        np.save(self._dir / EMBEDDINGS_FILENAME, np.asarray(embeddings, dtype=np.float32))
        (self._dir / PRODUCT_IDS_FILENAME).write_text(json.dumps(product_ids))
        logger.info("Saved embedding index to %s (%d products)", self._dir, len(product_ids))


class SbertModel:
    """What `self.model` is in the reference (a SentenceTransformer): tokenizer + device encoder
    with an `encode(texts, batch_size, show_progress_bar, normalize_embeddings)` method."""

    def __init__(self, loaded: LoadedModel, device: torch.device):
        self.tokenizer = loaded.tokenizer
        self.max_seq_length = loaded.max_seq_length
        self.shape = loaded.shape
        self.device = device
        self.encoder = DeviceEncoder(loaded.weights, loaded.shape, device)
        self._weights = loaded.weights
        self._encoder_no_flag: Optional[DeviceEncoder] = None  # normalize_embeddings=False: one normalisation fewer

    def encode_to_device(self, texts: Sequence[str], tokens_per_call: int = 1 << 18) -> torch.Tensor:
        """Embeddings [n, 384] left on the GPU (serving path: no host round trip)."""
        packed = getattr(self.tokenizer, "packed", None)
        if packed is not None:
            return self.encoder.encode_packed_host(*packed(texts), max_tokens_per_call=tokens_per_call)
        return self.encoder.encode_ids(self.tokenizer(texts), max_tokens_per_call=tokens_per_call)

    def encode(self, sentences, batch_size: int = 64, show_progress_bar: bool = False,
               normalize_embeddings: bool = True, **_ignored) -> np.ndarray:
        """SentenceTransformer.encode-compatible: numpy float32 [n, 384] (or [384] for a str).

        The reference pads and runs `batch_size` texts per forward; packed varlen batching makes
        the result independent of batch composition, so `batch_size` only bounds tokens per call."""
        single = isinstance(sentences, str)
        texts = [sentences] if single else list(sentences)
        if not texts:
            return np.zeros((0, self.shape.hidden), np.float32)
        if normalize_embeddings or self.shape.n_normalize < 1:
            enc = self.encoder
        else:
            # the flag's F.normalize is the LAST of the n_normalize passes the device encoder applies (the
            # pipeline's own Normalize module, if any, stays): a second encoder configured with one pass fewer
            if self._encoder_no_flag is None:
                from dataclasses import replace

                self._encoder_no_flag = DeviceEncoder(self._weights, replace(self.shape, n_normalize=self.shape.n_normalize - 1),
                                                      self.device, gemm_mode=self.encoder.gemm_mode)
            enc = self._encoder_no_flag
        packed = getattr(self.tokenizer, "packed", None)
        per_call = max(int(batch_size), 1) * 4096
        emb = (enc.encode_packed_host(*packed(texts), max_tokens_per_call=per_call) if packed is not None
               else enc.encode_ids(self.tokenizer(texts), max_tokens_per_call=per_call)).cpu().numpy()
        return emb[0] if single else emb


class Recommender:
    """Two-tower recommender: same surface as the reference's Recommender (:133-225)."""

    def __init__(self, model_dir: Path | str, corpus_path: Path, batch_size: int = 64, use_index: bool = True):
        self.model_dir = self._resolve_model_dir(model_dir)
        self.corpus_path = Path(corpus_path).resolve()
        self.product_ids, self.product_texts = self._load_corpus()
        self.pid_to_text = dict(zip(self.product_ids, self.product_texts))
        self._pid_to_row = {pid: i for i, pid in enumerate(self.product_ids)}
        self.device = self._inference_device()
        self.model = self._load_model()
        self.product_embeddings = self._load_or_build_embeddings(batch_size, use_index)
        # fp32 rows + f16 filter planes: large batches (>= 256 queries) rank on the f16 matrix cores and are verified
        # exactly — same bits out as plain "f32" (ICREC_INDEX_STORAGE=f32 turns the planes off, =bf16 halves the rows)
        self._index = DeviceIndex(self.product_embeddings, self.device,
                                  storage=os.getenv("ICREC_INDEX_STORAGE", "f32+filter"))
        self._fast = None
        if os.getenv("ICREC_USE_GRAPH", "1") != "0":
            from .fastpath import SingleRequestPath

            self._fast = SingleRequestPath(self.model.encoder, self._index)

    # -- construction helpers (names follow the reference) ---------------------------------
    def _resolve_model_dir(self, model_dir: Path | str) -> Path | str:
        p = Path(model_dir)
        return p.resolve() if p.exists() else model_dir

    def _load_corpus(self) -> tuple[list[str], list[str]]:
        """eval_corpus.json: {product_id: text}; key order = row order of the embedding matrix."""
        with open(self.corpus_path) as f:
            corpus = json.load(f)
        ids = list(corpus.keys())
        return ids, [corpus[pid] for pid in ids]

    def _inference_device(self) -> torch.device:
        """INFERENCE_DEVICE env ("cuda", "cuda:1") or the current HIP device.  This build has no
        CPU/MPS path: anything else raises."""
        override = os.getenv("INFERENCE_DEVICE")
        name = override or "cuda"
        dev = torch.device(name)
        if dev.type != "cuda":
            raise _native.IcrecError(f"INFERENCE_DEVICE={name!r}: this build runs on MI355X (cuda/HIP devices) only")
        if not torch.cuda.is_available():
            raise _native.IcrecError("no HIP device visible; the MI355X kernels have no CPU fallback")
        return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())

    def _load_model(self) -> SbertModel:
        logger.info("Using inference device: %s", self.device)
        return SbertModel(load_model_dir(self.model_dir), self.device)

    def _load_or_build_embeddings(self, batch_size: int, use_index: bool) -> np.ndarray:
        index = EmbeddingIndex(self.corpus_path, self.model_dir)
        if use_index:
            cached = index.load(self.product_ids)
            if cached is not None:
                logger.info("Loaded model from %s, corpus %d products (embeddings from index)", self.model_dir,
                            len(self.product_ids))
                return cached
        embeddings = self.model.encode(self.product_texts, batch_size=batch_size, show_progress_bar=True,
                                       normalize_embeddings=True)
        if use_index:
            index.save(self.product_ids, embeddings)
        logger.info("Loaded model from %s, corpus %d products", self.model_dir, len(self.product_ids))
        return embeddings

    # -- the hot path -------------------------------------------------------------------------
    def _excluded_rows(self, exclude_product_ids) -> list[int]:
        if not exclude_product_ids:
            return []
        return [self._pid_to_row[p] for p in exclude_product_ids if p in self._pid_to_row]

    def _rank(self, query_emb: torch.Tensor, top_k: int, exclude_lists: Optional[list[list[int]]]):
        k = min(int(top_k), _native.ICREC_MAX_K, len(self.product_ids))
        if top_k > _native.ICREC_MAX_K:
            raise ValueError(f"top_k={top_k} exceeds the kernel limit {_native.ICREC_MAX_K} "
                             "(the API schema allows at most 100)")
        idx, sc = self._index.search(query_emb, k, exclude_lists)
        return idx.cpu().numpy(), sc.cpu().numpy()

    def _to_results(self, idx_row: np.ndarray, sc_row: np.ndarray) -> list[tuple[str, float]]:
        return [(self.product_ids[int(i)], float(s)) for i, s in zip(idx_row, sc_row) if i >= 0]

    def recommend_batch(self, queries: Sequence[str], top_k: int = 10,
                        exclude_product_ids: Optional[Sequence[Optional[set[str]]]] = None
                        ) -> list[list[tuple[str, float]]]:
        """Many contexts in one GPU pass; element i equals recommend(queries[i], ...)."""
        if not queries:
            return []
        top_k = max(int(top_k), 1)  # the reference's loop appends before testing len >= top_k (:223-224)
        ex = None
        if exclude_product_ids is not None and any(exclude_product_ids):
            ex = [self._excluded_rows(e) for e in exclude_product_ids]
        emb = self.model.encode_to_device(list(queries))
        idx, sc = self._rank(emb, top_k, ex)
        return [self._to_results(idx[i], sc[i]) for i in range(len(queries))]

    def recommend_batches(self, batches, top_k: int = 10, exclude_product_ids=None):
        """Generator over a stream of query batches: element j equals recommend_batch(batches[j], top_k,
        exclude_product_ids[j]).  Batch j+1 is tokenised on a worker thread while the GPU works on batch j and
        batch j is read back after batch j+1 has been launched (pipeline.py)."""
        from .pipeline import pipelined_search

        top_k = max(int(top_k), 1)
        if top_k > _native.ICREC_MAX_K:
            raise ValueError(f"top_k={top_k} exceeds the kernel limit {_native.ICREC_MAX_K} "
                             "(the API schema allows at most 100)")
        k = min(top_k, len(self.product_ids))

        def exclude(j):
            e = exclude_product_ids[j] if exclude_product_ids is not None else None
            return [self._excluded_rows(x) for x in e] if e is not None and any(e) else None

        for idx, sc in pipelined_search(self.model.tokenizer, self.model.encoder, self._index.search, batches, k, exclude):
            yield [self._to_results(idx[i], sc[i]) for i in range(idx.shape[0])]

    def recommend_batch_timed(self, queries: Sequence[str], top_k: int = 10, exclude_product_ids=None):
        """recommend_batch plus (embedding ms incl. host tokenisation, similarity ms) from HIP events
        on the launch stream — what the micro-batching server reports as per-request stats."""
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        top_k = max(int(top_k), 1)
        ex = None
        if exclude_product_ids is not None and any(exclude_product_ids):
            ex = [self._excluded_rows(e) for e in exclude_product_ids]
        t0 = time.time()
        packed = getattr(self.model.tokenizer, "packed", None)
        ids = packed(list(queries)) if packed is not None else self.model.tokenizer(list(queries))
        tok_ms = (time.time() - t0) * 1000
        stream = torch.cuda.current_stream(self.device)
        e0.record(stream)
        emb = self.model.encoder.encode_packed_host(*ids) if packed is not None else self.model.encoder.encode_ids(ids)
        e1.record(stream)
        if top_k > _native.ICREC_MAX_K:
            raise ValueError(f"top_k={top_k} exceeds the kernel limit {_native.ICREC_MAX_K}")
        idx_d, sc_d = self._index.search(emb, min(top_k, len(self.product_ids)), ex)
        e2.record(stream)
        idx, sc = idx_d.cpu().numpy(), sc_d.cpu().numpy()
        return ([self._to_results(idx[i], sc[i]) for i in range(len(queries))],
                tok_ms + e0.elapsed_time(e1), e1.elapsed_time(e2))

    def recommend(self, query: str, top_k: int = 10,
                  exclude_product_ids: set[str] | None = None) -> list[tuple[str, float]]:
        """Top-k (product_id, score) by cosine similarity, best first (reference :206-225).
        One query = one hipGraph replay (fastpath.py) when ICREC_USE_GRAPH is not "0"."""
        fast = self._fast_path()
        if fast is not None:
            top_k = max(int(top_k), 1)
            ids = self.model.tokenizer([query])[0]
            ex = self._excluded_rows(exclude_product_ids)
            k = min(top_k, len(self.product_ids))
            if top_k <= _native.ICREC_MAX_K and fast.supports(len(ids), k, len(ex)):
                idx, sc = fast.run(ids, k, ex)
                return self._to_results(idx, sc)
        return self.recommend_batch([query], top_k, [exclude_product_ids])[0]

    def _fast_path(self):
        """The hipGraph single-request path, rebuilt when the index or the model was replaced under it (a captured
        graph bakes the index / encoder handles), or None under ICREC_USE_GRAPH=0."""
        if self._fast is not None and (self._fast.index is not self._index or self._fast.encoder is not self.model.encoder):
            self._fast = type(self._fast)(self.model.encoder, self._index)
        return self._fast


class MonitoredRecommender(Recommender):
    """Recommender with timing; sets last_metrics after each recommend() (reference :228-293).
    Device time comes from HIP events on the launch stream, host tokenisation is added to the
    embedding time (it is part of model.encode in the reference)."""

    def __init__(self, *args, metrics_logger: Optional[logging.Logger] = None, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.metrics_logger = metrics_logger or logging.getLogger("recommender.metrics")
        self.last_metrics: Optional[RecommendationMetrics] = None

    def recommend(self, query: str, top_k: int = 10, user_id: Optional[str] = None,
                  exclude_product_ids: set[str] | None = None) -> list[tuple[str, float]]:
        start = time.time()
        top_k = max(int(top_k), 1)  # the reference's loop appends before testing len >= top_k (:259-262)
        t_tok = time.time()
        ids = self.model.tokenizer([query])
        tok_ms = (time.time() - t_tok) * 1000
        fast = self._fast_path()
        if fast is not None and top_k <= _native.ICREC_MAX_K:
            # the same replayed request as Recommender.recommend, cut at the encode / search seam with HIP events
            # around the two replays (fastpath.py): the three timing fields keep their meaning on the graph path
            ex_rows = self._excluded_rows(exclude_product_ids)
            k = min(top_k, len(self.product_ids))
            if fast.supports(len(ids[0]), k, len(ex_rows)):
                idx1, sc1, enc_ms, sim_ms = fast.run(ids[0], k, ex_rows, timed=True)
                results = self._to_results(idx1, sc1)
                self.note_served(results, user_id, tok_ms + enc_ms, sim_ms, (time.time() - start) * 1000)
                return results
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        stream = torch.cuda.current_stream(self.device)
        e0.record(stream)
        emb = self.model.encoder.encode_ids(ids)
        e1.record(stream)
        ex = [self._excluded_rows(exclude_product_ids)] if exclude_product_ids else None
        k = min(top_k, len(self.product_ids))
        if top_k > _native.ICREC_MAX_K:
            raise ValueError(f"top_k={top_k} exceeds the kernel limit {_native.ICREC_MAX_K}")
        idx_d, sc_d = self._index.search(emb, k, ex)
        e2.record(stream)
        idx, sc = idx_d.cpu().numpy(), sc_d.cpu().numpy()  # synchronises the stream
        encode_ms = tok_ms + e0.elapsed_time(e1)
        sim_ms = e1.elapsed_time(e2)
        results = self._to_results(idx[0], sc[0])
        self.note_served(results, user_id, encode_ms, sim_ms, (time.time() - start) * 1000)
        return results

    def note_served(self, results: list[tuple[str, float]], user_id: Optional[str], encode_ms: float, sim_ms: float,
                    total_ms: float) -> RecommendationMetrics:
        """Fill last_metrics and emit the `recommendation_served` record for ONE served request (reference
        :268-278).  recommend() calls it per request; the micro-batching server calls it once per request of a
        batch with the batch's timings (api/batcher.py)."""
        n = len(results)
        m = self.last_metrics = RecommendationMetrics(
            user_id=user_id or "anonymous", query_embedding_time_ms=encode_ms, similarity_compute_time_ms=sim_ms,
            total_latency_ms=total_ms, num_recommendations=n, top_score=results[0][1] if results else 0.0,
            avg_score=sum(s for _, s in results) / n if n else 0.0, timestamp=time.time())
        if self.metrics_logger.isEnabledFor(logging.INFO):
            self._log_metrics(m)
        return m

    def _log_metrics(self, m: RecommendationMetrics) -> None:
        self.metrics_logger.info("recommendation_served", extra={
            "user_id": m.user_id, "latency_ms": m.total_latency_ms, "encode_time_ms": m.query_embedding_time_ms,
            "similarity_time_ms": m.similarity_compute_time_ms, "num_results": m.num_recommendations,
            "top_score": m.top_score, "avg_score": m.avg_score})
