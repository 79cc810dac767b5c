"""HTTP surface tests, modelled on the reference's tests/test_api.py + tests/conftest.py: the
recommender is a MagicMock patched in where the lifespan constructs it, so no model loads.
Feedback endpoints are out of scope and not tested."""
from __future__ import annotations

import asyncio
import os
from pathlib import Path
from unittest.mock import MagicMock, patch

import pytest
from fastapi.testclient import TestClient

from instacart_next_order_recommendation_amd.api.app import app
from instacart_next_order_recommendation_amd.api.batcher import MicroBatcher
from instacart_next_order_recommendation_amd.recommender import RecommendationMetrics

APP_MOD = "instacart_next_order_recommendation_amd.api.app"


@pytest.fixture
def mock_recommender():
    mock = MagicMock()
    mock.recommend.return_value = [("13517", 0.76), ("34479", 0.71), ("48628", 0.70)]
    mock.pid_to_text = {
        "13517": "Product: Whole Wheat Bread. Aisle: bread. Department: bakery.",
        "34479": "Product: Whole Wheat Walnut Bread. Aisle: bread. Department: bakery.",
        "48628": "Product: Organic Whole Wheat Bread. Aisle: bread. Department: bakery.",
    }
    mock.corpus_path = Path("/tmp/test/corpus.json")
    mock.last_metrics = RecommendationMetrics("test", 10.0, 5.0, 15.0, 3, 0.76, 0.72, 1234567890.0)
    return mock


@pytest.fixture
def client(mock_recommender):
    with patch(f"{APP_MOD}.MonitoredRecommender", return_value=mock_recommender):
        with TestClient(app) as c:
            yield c


def test_health_ready_and_request_id(client):
    assert client.get("/health").json() == {"status": "ok"}
    assert client.get("/ready").json() == {"status": "ready"}
    assert "X-Request-ID" in client.get("/health").headers
    assert client.get("/health", headers={"X-Request-ID": "abc"}).headers["X-Request-ID"] == "abc"


def test_recommend_with_user_context_returns_200(client):
    resp = client.post("/recommend", json={"user_context": "[+7d w4h14] Organic Milk, Whole Wheat Bread.", "top_k": 5})
    assert resp.status_code == 200
    data = resp.json()
    assert "request_id" in data and len(data["recommendations"]) == 3
    assert data["recommendations"][0]["product_id"] == "13517"
    assert data["recommendations"][0]["score"] == 0.76
    assert data["recommendations"][0]["product_text"].startswith("Product: Whole Wheat Bread")
    assert data["purchase_history_used"] == "[+7d w4h14] Organic Milk, Whole Wheat Bread."


def test_recommend_without_context_returns_400(client):
    resp = client.post("/recommend", json={"top_k": 5})
    assert resp.status_code == 400
    assert "user_context" in resp.json()["detail"].lower() or "user_id" in resp.json()["detail"].lower()


def test_recommend_validates_top_k_range(client):
    assert client.post("/recommend", json={"user_context": "[+7d] Milk.", "top_k": 0}).status_code == 422
    assert client.post("/recommend", json={"user_context": "[+7d] Milk.", "top_k": 101}).status_code == 422
    assert client.post("/recommend", json={"user_context": "x" * 10_001}).status_code == 422


def test_recommend_forwards_exclusions_and_query(client, mock_recommender):
    mock_recommender.recommend.return_value = [("999", 0.5)]
    resp = client.post("/recommend", json={"user_context": "[+7d] Milk.", "query": "oat", "top_k": 5,
                                           "exclude_product_ids": ["13517"]})
    assert resp.status_code == 200
    kw = mock_recommender.recommend.call_args[1]
    assert kw["exclude_product_ids"] == {"13517"} and kw["top_k"] == 5
    assert kw["query"] == "oat [+7d] Milk."          # f"{query} {context}" (routes/recommend.py:121-123)
    assert resp.json()["recommendations"][0]["product_text"] is None  # unknown pid -> no text


def test_user_id_resolves_through_eval_queries(client, mock_recommender, tmp_path):
    (tmp_path / "eval_queries.json").write_text('{"42": "[+3d w1h9] Banana, Greek Yogurt."}')
    client.app.state.corpus_path = tmp_path / "eval_corpus.json"
    resp = client.post("/recommend", json={"user_id": "42"})
    assert resp.status_code == 200 and resp.json()["purchase_history_used"] == "[+3d w1h9] Banana, Greek Yogurt."
    assert client.post("/recommend", json={"user_id": "43"}).status_code == 400


def test_api_key(mock_recommender):
    os.environ["API_KEY"] = "secret-test-key"
    try:
        with patch(f"{APP_MOD}.MonitoredRecommender", return_value=mock_recommender):
            with TestClient(app) as c:
                body = {"user_context": "[+7d] Milk.", "top_k": 5}
                assert c.post("/recommend", json=body).status_code == 401
                assert c.post("/recommend", json=body, headers={"X-API-Key": "secret-test-key"}).status_code == 200
                assert c.post("/recommend", json=body, headers={"Authorization": "Bearer secret-test-key"}).status_code == 200
                assert c.post("/admin/corpus", json={"corpus": {"1": "a"}}).status_code == 401
                assert c.get("/health").status_code == 200
    finally:
        os.environ.pop("API_KEY", None)


def test_admin_corpus(client, mock_recommender):
    new = MagicMock()
    new.pid_to_text = {"1": "Product: A."}
    with patch(f"{APP_MOD}.MonitoredRecommender", return_value=new) as ctor:
        resp = client.post("/admin/corpus", json={"corpus": {"1": "Product: A.", "2": "Product: B."}})
        assert resp.status_code == 200 and resp.json() == {"status": "ok", "n_products": 2}
        assert ctor.call_args[1]["corpus_path"].name == "eval_corpus.json"
    assert client.app.state.recommender is new
    assert client.post("/admin/corpus", json={"corpus": {}}).status_code == 422
    with patch(f"{APP_MOD}.MonitoredRecommender", side_effect=RuntimeError("boom")):
        r = client.post("/admin/corpus", json={"corpus": {"1": "x"}})
        assert r.status_code == 500 and "boom" in r.json()["detail"]


def test_metrics_names(client):
    client.post("/recommend", json={"user_context": "[+7d] Milk."})
    client.post("/recommend", json={"top_k": 5})
    text = client.get("/metrics").text
    for name in ("recommendation_requests_total", "recommendation_latency_seconds", "recommendation_encode_seconds",
                 "model_loaded"):
        assert name in text
    assert 'recommendation_requests_total{status="success"}' in text
    assert 'recommendation_requests_total{status="error"}' in text


# ---------------------------------------------------------------- micro-batcher
class FakeRecommender:
    def __init__(self, delay=0.0, fail=False):
        self.calls, self.delay, self.fail = [], delay, fail

    def recommend_batch(self, queries, top_k, excl):
        import time

        self.calls.append((list(queries), top_k, list(excl)))
        time.sleep(self.delay)
        if self.fail:
            raise RuntimeError("gpu fell over")
        return [[(f"{q}-{i}", 1.0 - 0.01 * i) for i in range(top_k)] for q in queries]


def test_micro_batcher_coalesces_and_slices():
    async def go():
        rec = FakeRecommender(delay=0.02)
        b = MicroBatcher(rec, max_batch=64, max_wait_ms=30)
        await b.start()
        ks = [3, 7, 1, 5, 2, 9, 4, 6]
        outs = await asyncio.gather(*[b.submit(f"q{i}", ks[i], {"x"} if i % 2 else None) for i in range(8)])
        await b.stop()
        return rec, ks, outs

    rec, ks, outs = asyncio.run(go())
    assert len(rec.calls) == 1                                # one GPU pass for 8 concurrent requests
    qs, k, excl = rec.calls[0]
    assert qs == [f"q{i}" for i in range(8)] and k == 9 and excl[1] == {"x"} and excl[0] is None
    for i, (res, tm) in enumerate(outs):
        assert [p for p, _ in res] == [f"q{i}-{j}" for j in range(ks[i])]   # own rows, own top_k
        assert tm.batch_size == 8


def test_micro_batcher_respects_max_batch_and_propagates_errors():
    async def go():
        rec = FakeRecommender()
        b = MicroBatcher(rec, max_batch=3, max_wait_ms=50)
        await b.start()
        await asyncio.gather(*[b.submit(f"q{i}", 2, None) for i in range(7)])
        sizes = [len(c[0]) for c in rec.calls]
        bad = MicroBatcher(FakeRecommender(fail=True), max_batch=4, max_wait_ms=5)
        errs = await asyncio.gather(*[bad.submit("q", 1, None) for _ in range(3)], return_exceptions=True)
        await b.stop(); await bad.stop()
        return sizes, errs

    sizes, errs = asyncio.run(go())
    assert sum(sizes) == 7 and max(sizes) <= 3
    assert all(isinstance(e, RuntimeError) for e in errs)


def test_recommender_is_loaded_on_demand_when_not_preloaded(mock_recommender):
    """The reference's fallback (src/api/routes/recommend.py:76-80): a request that finds no recommender on the app
    constructs one.  Here the lifespan's recommender is removed behind the app's back; the next /recommend must load
    one (ONE constructor call even though it is awaited off the event loop) and answer 200."""
    with patch(f"{APP_MOD}.MonitoredRecommender", return_value=mock_recommender) as ctor:
        with TestClient(app) as c:
            assert ctor.call_count == 1
            app.state.recommender = None
            app.state.batcher = None
            assert c.get("/ready").json() == {"status": "not_ready"}
            r = c.post("/recommend", json={"user_context": "[+7d w4h14] Organic Milk.", "top_k": 3})
            assert r.status_code == 200 and len(r.json()["recommendations"]) == 3
            assert ctor.call_count == 2
            assert c.get("/ready").json() == {"status": "ready"}


def test_on_demand_load_failure_is_a_503(mock_recommender):
    with patch(f"{APP_MOD}.MonitoredRecommender", return_value=mock_recommender) as ctor:
        with TestClient(app) as c:
            app.state.recommender = None
            ctor.side_effect = FileNotFoundError("no such model dir")
            r = c.post("/recommend", json={"user_context": "x", "top_k": 3})
            assert r.status_code == 503 and "no such model dir" in r.json()["detail"]
            ctor.side_effect = None
            app.state.recommender = mock_recommender
