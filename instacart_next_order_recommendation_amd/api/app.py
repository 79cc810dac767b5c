"""FastAPI app: /health /ready /metrics, POST /recommend, POST /admin/corpus.

Follows the reference's src/api/main.py:51-166 (lifespan, request-id middleware, probes,
metrics endpoint), src/api/routes/recommend.py:84-199 (context resolution, response assembly,
Prometheus observations), src/api/routes/corpus.py:47-106 (re-index on upload) and
src/api/auth.py:39-71 (X-API-Key / Bearer).  Differences, all deliberate:
  * requests go through a MicroBatcher instead of a blocking call on the event loop;
  * eval_queries.json is cached (the reference re-parses it on every user_id request, :40-63,:115);
  * no slowapi rate limiter (its 100/min default would throttle any throughput measurement);
  * feedback endpoints are out of scope.
Env: MODEL_DIR, CORPUS_PATH, API_KEY, INFERENCE_DEVICE, MAX_CORPUS_UPLOAD_PRODUCTS,
     BATCH_MAX_SIZE (256), BATCH_MAX_WAIT_MS (2).
     ICREC_GPU_WORKER_SOCKET: this process is an HTTP front-end of the multi-process server (serve.py): it loads
     the corpus JSON only and forwards every request to the GPU-owner process(es) (worker.py; a comma-separated list of
     sockets = several GPU owners on one GPU, requests round-robin).
"""
from __future__ import annotations

import asyncio
import json
import logging
import os
import tempfile
import time
from contextlib import asynccontextmanager
from pathlib import Path
from typing import AsyncIterator, Optional
from uuid import uuid4

from fastapi import Depends, FastAPI, HTTPException, Request, Response, status
from prometheus_client import CONTENT_TYPE_LATEST, generate_latest

from ..recommender import MonitoredRecommender, Recommender
from ..recommender import MonitoredRecommender as _MonitoredType  # isinstance target (tests patch the constructor name)
from .batcher import BatcherStopped, MicroBatcher
from .metrics import (API_REGISTRY, MODEL_LOADED, RECOMMENDATION_BATCH_SIZE, RECOMMENDATION_ENCODE_SECONDS,
                      RECOMMENDATION_LATENCY_SECONDS, RECOMMENDATION_REQUESTS_TOTAL)
from .schemas import (CorpusUploadRequest, CorpusUploadResponse, HealthResponse, InferenceStatistics,
                      RecommendationItem, RecommendationRequest, RecommendationResponse)

logger = logging.getLogger(__name__)

EVAL_QUERIES_FILENAME = "eval_queries.json"  # src/constants.py:55
DEFAULT_MAX_CORPUS_UPLOAD_PRODUCTS = 100_000  # src/constants.py:83


def _env_path(name: str, default: str) -> Path:
    return Path(os.getenv(name) or default)


def _install(app: FastAPI, recommender, corpus_path, batcher=None) -> None:
    """Swap in a recommender + its batcher (plain attribute stores: a request sees either the old pair or the new)."""
    batcher = batcher or MicroBatcher(recommender, max_batch=int(os.getenv("BATCH_MAX_SIZE", "256")),
                                      max_wait_ms=float(os.getenv("BATCH_MAX_WAIT_MS", "2")))
    app.state.batcher = batcher
    app.state.recommender = recommender
    app.state.corpus_path = corpus_path
    app.state.eval_queries_cache = None


def _install_frontend(app: FastAPI, sock_path: str, corpus_path) -> None:
    """Front-end of the multi-process server: corpus texts only + a socket to the GPU worker."""
    from .remote import CorpusView, MultiRemoteBatcher, RemoteBatcher

    def on_corpus(new_path: str) -> None:  # the worker re-indexed (another front-end's /admin/corpus): reload texts
        app.state.recommender = CorpusView(new_path)
        app.state.corpus_path = Path(new_path)
        app.state.eval_queries_cache = None

    socks = [p for p in sock_path.split(",") if p]  # serve.py --gpu-workers N hands every front-end all the sockets
    batcher = RemoteBatcher(socks[0], on_corpus) if len(socks) == 1 else MultiRemoteBatcher(socks, on_corpus)
    _install(app, CorpusView(corpus_path), corpus_path, batcher)


@asynccontextmanager
async def lifespan(app: FastAPI) -> AsyncIterator[None]:
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    model_dir = _env_path("MODEL_DIR", "models/two_tower_sbert/final")
    corpus_path = _env_path("CORPUS_PATH", "processed/p5_mp20_ef0.1/eval_corpus.json")
    sock_path = os.getenv("ICREC_GPU_WORKER_SOCKET")
    if sock_path:
        _install_frontend(app, sock_path, corpus_path)
        await app.state.batcher.start()  # connect now: /ready reflects the worker socket from the first probe on
    else:
        logger.info("Loading recommender model_dir=%s corpus=%s", model_dir, corpus_path)
        _install(app, MonitoredRecommender(model_dir=model_dir, corpus_path=corpus_path), corpus_path)
    app.state.ready = True
    MODEL_LOADED.set(1)
    import gc

    gc.collect()
    gc.freeze()  # the catalog and the modules live as long as the process: keep full collections away from them (worker.py)
    try:
        yield
    finally:
        MODEL_LOADED.set(0)
        b = getattr(app.state, "batcher", None)
        if b is not None:
            await b.stop()


app = FastAPI(title="Instacart Next-Order Recommendation API (MI355X)", lifespan=lifespan)


class RequestIdMiddleware:
    """X-Request-ID propagation + request log line (reference: src/api/main.py request_logging_middleware) as a
    plain ASGI middleware: Starlette's BaseHTTPMiddleware costs a task group and two memory streams per
    request, which at micro-batched rates is a visible share of the per-request budget."""

    def __init__(self, inner):
        self.inner = inner

    async def __call__(self, scope, receive, send):
        if scope["type"] != "http":
            await self.inner(scope, receive, send)
            return
        start = time.time()
        req_id = None
        for k, v in scope.get("headers") or ():
            if k == b"x-request-id":
                req_id = v.decode("latin-1")
                break
        req_id = req_id or str(uuid4())
        scope.setdefault("state", {})["request_id"] = req_id
        status_code = 0

        async def send_with_id(message):
            nonlocal status_code
            if message["type"] == "http.response.start":
                status_code = message["status"]
                message.setdefault("headers", [])
                message["headers"] = list(message["headers"]) + [(b"x-request-id", req_id.encode("latin-1"))]
            await send(message)

        await self.inner(scope, receive, send_with_id)
        if logger.isEnabledFor(logging.DEBUG):
            logger.debug("request path=%s method=%s status=%d request_id=%s latency_ms=%d", scope.get("path"),
                         scope.get("method"), status_code, req_id, int((time.time() - start) * 1000))


app.add_middleware(RequestIdMiddleware)


async def verify_api_key(request: Request) -> None:
    """When API_KEY is set, require it as X-API-Key or `Authorization: Bearer` (src/api/auth.py)."""
    expected = os.getenv("API_KEY")
    if not expected:
        return
    got = request.headers.get("X-API-Key")
    if not got:
        auth = request.headers.get("Authorization", "")
        if auth.lower().startswith("bearer "):
            got = auth[7:].strip()
    if got != expected:
        raise HTTPException(status_code=status.HTTP_401_UNAUTHORIZED, detail="Invalid or missing API key")


_lazy_lock: Optional[asyncio.Lock] = None


async def get_recommender(request: Request) -> Recommender:  # async: a sync dependency costs a thread-pool hop per request
    """The app's recommender; loaded ON DEMAND when the lifespan did not run or failed to install one — the reference's
    fallback (src/api/routes/recommend.py:76-80: `MonitoredRecommender(DEFAULT_MODEL_DIR, DEFAULT_CORPUS_PATH)` inside
    the dependency).  Same outcome, two differences: the construction (model upload + catalog encode) runs in a worker
    thread so the event loop keeps answering probes, and concurrent first requests wait for ONE load instead of each
    starting their own.  A load that fails answers 503 with the reason (the reference lets the exception become a 500)."""
    rec = getattr(request.app.state, "recommender", None)
    if rec is not None:
        return rec
    global _lazy_lock
    if _lazy_lock is None:
        _lazy_lock = asyncio.Lock()
    async with _lazy_lock:
        rec = getattr(request.app.state, "recommender", None)
        if rec is not None:
            return rec
        logger.warning("Recommender not preloaded; loading on-demand")
        model_dir = _env_path("MODEL_DIR", "models/two_tower_sbert/final")
        corpus_path = _env_path("CORPUS_PATH", "processed/p5_mp20_ef0.1/eval_corpus.json")
        try:
            sock_path = os.getenv("ICREC_GPU_WORKER_SOCKET")
            if sock_path:
                _install_frontend(request.app, sock_path, corpus_path)
                await request.app.state.batcher.start()
            else:
                rec = await asyncio.to_thread(MonitoredRecommender, model_dir=model_dir, corpus_path=corpus_path)
                _install(request.app, rec, corpus_path)
        except Exception as exc:  # noqa: BLE001
            logger.exception("on-demand load failed")
            raise HTTPException(status_code=status.HTTP_503_SERVICE_UNAVAILABLE,
                                detail=f"recommender not loaded: {type(exc).__name__}: {exc}") from exc
        request.app.state.ready = True
        MODEL_LOADED.set(1)
        return request.app.state.recommender


def _eval_queries(app: FastAPI, corpus_path: Path) -> dict[str, str]:
    """eval_queries.json next to the corpus, cached by (path, mtime)."""
    path = Path(corpus_path).parent / EVAL_QUERIES_FILENAME
    try:
        mtime = path.stat().st_mtime
    except OSError:
        return {}
    cache = getattr(app.state, "eval_queries_cache", None)
    if cache and cache[0] == (str(path), mtime):
        return cache[1]
    try:
        data = json.loads(path.read_text())
        data = {str(k): str(v) for k, v in data.items()} if isinstance(data, dict) else {}
    except (OSError, ValueError):
        logger.exception("Failed to load %s", path)
        data = {}
    app.state.eval_queries_cache = ((str(path), mtime), data)
    return data


@app.get("/health", response_model=HealthResponse)
async def health() -> HealthResponse:
    return HealthResponse(status="ok")


@app.get("/ready", response_model=HealthResponse)
async def ready(request: Request) -> HealthResponse:
    ok = bool(getattr(request.app.state, "ready", False)) and getattr(request.app.state, "recommender", None)
    # front-end of the multi-process server: ready only while the socket to the GPU-owner process is up
    if ok and getattr(getattr(request.app.state, "batcher", None), "connected", True) is False:
        ok = False
    return HealthResponse(status="ready" if ok else "not_ready")


@app.get("/metrics")
async def metrics() -> Response:
    return Response(content=generate_latest(API_REGISTRY), media_type=CONTENT_TYPE_LATEST)


@app.post("/recommend", response_model=RecommendationResponse, status_code=status.HTTP_200_OK)
async def recommend_endpoint(payload: RecommendationRequest, request: Request,
                             recommender: Recommender = Depends(get_recommender),
                             _: None = Depends(verify_api_key)) -> RecommendationResponse:
    start_time = time.perf_counter()
    try:
        context = payload.user_context
        if context is None and payload.user_id is not None:
            corpus_path = getattr(request.app.state, "corpus_path", None) or recommender.corpus_path
            context = _eval_queries(request.app, Path(corpus_path)).get(str(payload.user_id))
        if payload.query is not None and payload.query.strip():
            retrieval_query = f"{payload.query} {context}" if context else payload.query
        else:
            retrieval_query = context
        if not retrieval_query:
            raise HTTPException(
                status_code=status.HTTP_400_BAD_REQUEST,
                detail="Either query (optional) must be provided, or user_context must be provided / user_id must be resolvable.")

        request_id = str(uuid4())
        exclude_ids = set(payload.exclude_product_ids or [])
        user_id_str = str(payload.user_id) if payload.user_id is not None else None
        stats = None
        batcher: Optional[MicroBatcher] = getattr(request.app.state, "batcher", None)
        if batcher is not None and hasattr(recommender, "recommend_batch") and not _is_mock(recommender):
            t_submit = time.time()
            try:
                results, tm = await batcher.submit(retrieval_query, payload.top_k, exclude_ids, user_id_str)
            except BatcherStopped:  # raced an /admin/corpus swap: the app holds the new pair by now
                batcher, recommender = request.app.state.batcher, request.app.state.recommender
                results, tm = await batcher.submit(retrieval_query, payload.top_k, exclude_ids, user_id_str)
            except Exception as exc:  # noqa: BLE001
                if type(exc).__name__ == "WorkerUnavailable":  # front-end mode: the GPU-owner process is gone
                    raise HTTPException(status_code=status.HTTP_503_SERVICE_UNAVAILABLE, detail=str(exc)) from exc
                raise
            RECOMMENDATION_BATCH_SIZE.observe(tm.batch_size)
            if isinstance(recommender, _MonitoredType) or getattr(recommender, "reports_stats", False) is True:
                n = len(results)
                stats = InferenceStatistics(
                    total_latency_ms=(time.time() - t_submit) * 1000, query_embedding_time_ms=tm.encode_ms,
                    similarity_compute_time_ms=tm.search_ms, num_recommendations=n,
                    top_score=results[0][1] if results else 0.0,
                    avg_score=sum(s for _, s in results) / n if n else 0.0, timestamp=time.time())
                RECOMMENDATION_ENCODE_SECONDS.observe(tm.encode_ms / 1000.0)
        else:  # duck-typed recommender (the reference's tests patch in a MagicMock): direct call
            if isinstance(recommender, _MonitoredType):
                results = recommender.recommend(query=retrieval_query, top_k=payload.top_k, user_id=user_id_str,
                                                exclude_product_ids=exclude_ids)
            else:
                results = recommender.recommend(query=retrieval_query, top_k=payload.top_k,
                                                exclude_product_ids=exclude_ids)
        items = [RecommendationItem(product_id=pid, score=score, product_text=recommender.pid_to_text.get(pid))
                 for pid, score in results]
        RECOMMENDATION_LATENCY_SECONDS.observe(time.perf_counter() - start_time)
        RECOMMENDATION_REQUESTS_TOTAL.labels(status="success").inc()
        return RecommendationResponse(request_id=request_id, recommendations=items, stats=stats,
                                      purchase_history_used=context)
    except Exception:
        RECOMMENDATION_REQUESTS_TOTAL.labels(status="error").inc()
        raise


def _is_mock(obj) -> bool:
    return type(obj).__module__.startswith("unittest.mock")


@app.post("/admin/corpus", response_model=CorpusUploadResponse)
async def corpus_upload_endpoint(payload: CorpusUploadRequest, request: Request,
                                 _: None = Depends(verify_api_key)) -> CorpusUploadResponse:
    """Replace the catalog: write it to a JSON file, build a NEW recommender (full GPU re-encode),
    swap it in (reference: routes/corpus.py:47-106)."""
    limit = int(os.getenv("MAX_CORPUS_UPLOAD_PRODUCTS", str(DEFAULT_MAX_CORPUS_UPLOAD_PRODUCTS)))
    if len(payload.corpus) > limit:
        raise HTTPException(status_code=status.HTTP_413_REQUEST_ENTITY_TOO_LARGE,
                            detail=f"corpus has {len(payload.corpus)} products; limit is {limit}")
    current = getattr(request.app.state, "recommender", None)
    tmp_dir = Path(tempfile.mkdtemp(prefix="icrec_corpus_"))
    corpus_path = tmp_dir / "eval_corpus.json"
    corpus_path.write_text(json.dumps(payload.corpus))
    old = getattr(request.app.state, "batcher", None)
    try:
        if getattr(current, "remote", False) is True:
            # multi-process server: the GPU worker re-encodes (in a thread of its own, still serving the old catalog)
            # and swaps; every front-end then reloads the texts (this one right here, the others on the broadcast)
            await old.reindex(str(corpus_path))
            from .remote import CorpusView

            request.app.state.recommender = CorpusView(corpus_path)
            request.app.state.corpus_path = corpus_path
            request.app.state.eval_queries_cache = None
            return CorpusUploadResponse(status="ok", n_products=len(payload.corpus))
        model_dir = getattr(current, "model_dir", None) or _env_path("MODEL_DIR", "models/two_tower_sbert/final")
        # the full GPU re-encode runs in a worker thread: requests keep being served from the old recommender
        # (the reference builds it inline and stalls its event loop, routes/corpus.py:87-98), then one swap
        new_rec = await asyncio.get_running_loop().run_in_executor(
            None, lambda: MonitoredRecommender(model_dir=model_dir, corpus_path=corpus_path))
    except HTTPException:
        raise
    except Exception as exc:  # noqa: BLE001
        raise HTTPException(status_code=status.HTTP_500_INTERNAL_SERVER_ERROR,
                            detail=f"Failed to load corpus: {exc}") from exc
    _install(request.app, new_rec, corpus_path)
    if old is not None:
        await old.stop()
    return CorpusUploadResponse(status="ok", n_products=len(payload.corpus))
