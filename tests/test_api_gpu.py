"""The HTTP surface over the real MI355X recommender: concurrent /recommend requests are
micro-batched and return exactly what direct recommend() calls return."""
from __future__ import annotations

import json
from concurrent.futures import ThreadPoolExecutor

import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_requests_equal_direct_calls(tmp_path, monkeypatch):
    import torch

    assert torch.cuda.is_available()
    from fastapi.testclient import TestClient

    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.api.app import app
    from instacart_next_order_recommendation_amd.model_io import write_synthetic_model_dir

    model_dir = write_synthetic_model_dir(tmp_path / "model", seed=2)
    corpus_path = tmp_path / "processed" / "eval_corpus.json"
    corpus_path.parent.mkdir()
    corpus_path.write_text(json.dumps(syn.synthetic_catalog(500)))
    (corpus_path.parent / "eval_queries.json").write_text(json.dumps({"7": syn.synthetic_user_contexts(1, seed=7)[0]}))
    monkeypatch.setenv("MODEL_DIR", str(model_dir))
    monkeypatch.setenv("CORPUS_PATH", str(corpus_path))
    monkeypatch.setenv("BATCH_MAX_WAIT_MS", "20")
    ctxs = syn.synthetic_user_contexts(24, seed=5)
    with TestClient(app) as c:
        rec = c.app.state.recommender
        want = [rec.recommend(q, top_k=5 + i % 4, exclude_product_ids={"3", "9"} if i % 3 == 0 else None)
                for i, q in enumerate(ctxs)]

        def call(i):
            body = {"user_context": ctxs[i], "top_k": 5 + i % 4}
            if i % 3 == 0:
                body["exclude_product_ids"] = ["3", "9"]
            return c.post("/recommend", json=body)

        with ThreadPoolExecutor(12) as pool:
            resps = list(pool.map(call, range(24)))
        for i, r in enumerate(resps):
            assert r.status_code == 200, r.text
            d = r.json()
            assert [(x["product_id"], x["score"]) for x in d["recommendations"]] == want[i]
            assert d["recommendations"][0]["product_text"] == rec.pid_to_text[want[i][0][0]]
            s = d["stats"]
            assert s["num_recommendations"] == len(want[i]) and s["top_score"] == want[i][0][1]
            assert s["query_embedding_time_ms"] > 0 and s["similarity_compute_time_ms"] > 0
        r = c.post("/recommend", json={"user_id": "7", "top_k": 3})
        assert r.status_code == 200 and r.json()["purchase_history_used"].startswith("[+")
        text = c.get("/metrics").text
        assert "recommendation_batch_size_bucket" in text and "model_loaded 1.0" in text
        # corpus upload re-encodes on the GPU and swaps the recommender
        r = c.post("/admin/corpus", json={"corpus": syn.synthetic_catalog(64)})
        assert r.status_code == 200 and r.json()["n_products"] == 64
        r = c.post("/recommend", json={"user_context": ctxs[0], "top_k": 100})
        assert r.status_code == 200 and len(r.json()["recommendations"]) == 64
