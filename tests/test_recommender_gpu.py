"""End to end through the drop-in classes: text in, (product_id, score) out, against the
oracle run on the same token ids; EmbeddingIndex build / reuse; MonitoredRecommender metrics."""
from __future__ import annotations

import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    import torch

    assert torch.cuda.is_available()
    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.model_io import write_synthetic_model_dir

    root = tmp_path_factory.mktemp("rec")
    model_dir = write_synthetic_model_dir(root / "model", seed=1)
    corpus = syn.synthetic_catalog(700)
    corpus_path = root / "processed" / "eval_corpus.json"
    corpus_path.parent.mkdir()
    corpus_path.write_text(json.dumps(corpus))
    return {"model_dir": model_dir, "corpus_path": corpus_path, "corpus": corpus,
            "queries": syn.synthetic_user_contexts(6, seed=9)}


def _oracle_pipeline(rec, queries, k, excl_pids):
    """The reference's recommend() restated with the oracle on the recommender's own token ids."""
    from oracle import oracle

    shape = rec.model.shape
    cfg = oracle.make_cfg(vocab_size=shape.vocab_size, n_normalize=shape.n_normalize)
    from instacart_next_order_recommendation_amd.encoder import pack_token_ids
    from instacart_next_order_recommendation_amd.model_io import load_model_dir

    w = load_model_dir(rec.model_dir).weights
    ids, cu, _ = pack_token_ids(rec.model.tokenizer(queries))
    q_emb = oracle.encode(w, cfg, ids, cu)
    pids, cu_p, _ = pack_token_ids(rec.model.tokenizer(rec.product_texts))
    P = oracle.encode(w, cfg, pids, cu_p)
    row = {p: i for i, p in enumerate(rec.product_ids)}
    excl = [[row[p] for p in e if p in row] for e in excl_pids]
    idx, sc = oracle.search(q_emb, P, k, excl)
    return P, [[(rec.product_ids[i], float(s)) for i, s in zip(idx[j], sc[j]) if i >= 0] for j in range(len(queries))]


def test_recommender_end_to_end(world):
    from instacart_next_order_recommendation_amd.recommender import MonitoredRecommender, Recommender

    rec = Recommender(world["model_dir"], world["corpus_path"])
    assert rec.product_ids == list(world["corpus"].keys())
    assert rec.pid_to_text["1"] == world["corpus"]["1"]
    assert rec.product_embeddings.shape == (700, 384) and rec.product_embeddings.dtype == np.float32
    excl = [set(), {"1", "2"}, None, {"no-such-id"}, set(rec.product_ids[:50]), set()]
    P, want = _oracle_pipeline(rec, world["queries"], 10, [e or set() for e in excl])
    assert np.abs(rec.product_embeddings - P).max() < 5e-6  # catalog encode (index build) parity
    for i, q in enumerate(world["queries"]):
        got = rec.recommend(q, top_k=10, exclude_product_ids=excl[i])
        assert isinstance(got, list) and all(isinstance(p, str) and isinstance(s, float) for p, s in got)
        assert not {p for p, _ in got} & (excl[i] or set())
        # embeddings differ from the oracle's by <= 5e-6, so ids agree except across near-ties
        w = want[i]
        for (gp, gs), (wp, ws) in zip(got, w):
            assert abs(gs - ws) < 1e-4
            if gp != wp:
                j = [p for p, _ in w].index(gp) if gp in [p for p, _ in w] else None
                assert j is not None and abs(w[j][1] - ws) < 2e-5, (i, gp, wp)
    # batch form == single form, bit for bit
    batch = rec.recommend_batch(world["queries"], 10, excl)
    for i, q in enumerate(world["queries"]):
        assert batch[i] == rec.recommend(q, 10, excl[i])
    # fewer than top_k only when the catalog is exhausted
    few = rec.recommend(world["queries"][0], top_k=100, exclude_product_ids=set(rec.product_ids[:650]))
    assert len(few) == 50

    # second construction reuses the on-disk EmbeddingIndex (same bits)
    rec2 = MonitoredRecommender(world["model_dir"], world["corpus_path"])
    np.testing.assert_array_equal(rec2.product_embeddings, rec.product_embeddings)
    assert isinstance(rec2, Recommender)
    out = rec2.recommend(world["queries"][2], top_k=5, user_id="u42", exclude_product_ids={"3"})
    assert out == rec.recommend(world["queries"][2], 5, {"3"})
    m = rec2.last_metrics
    assert m.user_id == "u42" and m.num_recommendations == 5 and m.top_score == out[0][1]
    assert abs(m.avg_score - sum(s for _, s in out) / 5) < 1e-12
    assert 0 < m.query_embedding_time_ms and 0 < m.similarity_compute_time_ms and m.total_latency_ms >= m.similarity_compute_time_ms
    assert rec2.recommend(world["queries"][2], 5)[0][1] >= 0 and rec2.last_metrics.user_id == "anonymous"


def test_use_index_false_and_stale_cache(world, tmp_path):
    from instacart_next_order_recommendation_amd.recommender import EmbeddingIndex, Recommender

    corpus_path = tmp_path / "eval_corpus.json"
    small = dict(list(world["corpus"].items())[:40])
    corpus_path.write_text(json.dumps(small))
    rec = Recommender(world["model_dir"], corpus_path, use_index=False)
    assert not (tmp_path / ".embedding_index").exists()
    # a cache written for a different id list is ignored and overwritten
    EmbeddingIndex(corpus_path, rec.model_dir).save(["x"], np.zeros((1, 384), np.float32))
    rec2 = Recommender(world["model_dir"], corpus_path)
    np.testing.assert_array_equal(rec2.product_embeddings, rec.product_embeddings)
    assert EmbeddingIndex(corpus_path, rec.model_dir).load(rec.product_ids) is not None


def test_inference_device_env(world, monkeypatch):
    from instacart_next_order_recommendation_amd._native import IcrecError
    from instacart_next_order_recommendation_amd.recommender import Recommender

    monkeypatch.setenv("INFERENCE_DEVICE", "cpu")
    with pytest.raises(IcrecError):
        Recommender(world["model_dir"], world["corpus_path"])
    monkeypatch.setenv("INFERENCE_DEVICE", "cuda:0")
    assert Recommender(world["model_dir"], world["corpus_path"]).device.index == 0


def test_graph_fast_path_equals_plain_path(world, monkeypatch):
    """recommend() through the replayed hipGraph == the un-captured batch path, bit for bit, across
    token buckets, k values, exclusions and repeated replays."""
    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.recommender import Recommender

    rec = Recommender(world["model_dir"], world["corpus_path"])
    assert rec._fast is not None
    queries = ["[+1d w0h1] Milk.", world["queries"][0], world["queries"][1],
               "; ".join(syn.synthetic_user_contexts(8, seed=77))]          # 1 .. >128 tokens
    lens = [len(rec.model.tokenizer([q])[0]) for q in queries]
    assert min(lens) <= 32 and max(lens) > 128
    for rep in range(2):
        for q in queries:
            for k, ex in [(5, None), (20, {"1", "2", "3"}), (5, set(rec.product_ids[:300]))]:
                got = rec.recommend(q, k, ex)
                want = rec.recommend_batch([q], k, [ex])[0]
                assert got == want, (q[:20], k)
    assert len(rec._fast._graphs) >= 4
    monkeypatch.setenv("ICREC_USE_GRAPH", "0")
    assert Recommender(world["model_dir"], world["corpus_path"])._fast is None


def test_monitored_recommender_runs_on_the_graph_path(world, monkeypatch, caplog):
    """MonitoredRecommender.recommend (what the API constructs, reference src/api/main.py:73) replays the captured
    request cut at the encode / search seam with HIP events around the two replays: same results as the plain
    class's one-graph path and as the un-captured path, bit for bit; the three timing fields are filled; the
    `recommendation_served` record carries the user id."""
    import logging

    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.recommender import MonitoredRecommender, Recommender

    mon = MonitoredRecommender(world["model_dir"], world["corpus_path"])
    assert mon._fast is not None
    queries = ["[+1d w0h1] Milk.", world["queries"][0], "; ".join(syn.synthetic_user_contexts(8, seed=77))]
    with caplog.at_level(logging.INFO, logger="recommender.metrics"):
        for rep in range(2):
            for q in queries:
                for k, ex in [(5, None), (20, {"1", "2", "3"}), (5, set(mon.product_ids[:300]))]:
                    got = mon.recommend(q, k, user_id="u7", exclude_product_ids=ex)
                    assert got == Recommender.recommend(mon, q, k, ex), (q[:20], k)          # one-graph path
                    assert got == mon.recommend_batch([q], k, [ex])[0], (q[:20], k)          # kernel by kernel
                    m = mon.last_metrics
                    assert m.user_id == "u7" and m.num_recommendations == len(got) and m.top_score == got[0][1]
                    assert 0 < m.query_embedding_time_ms < 50 and 0 < m.similarity_compute_time_ms < 50
                    assert m.total_latency_ms >= m.similarity_compute_time_ms
    assert any(c.timed is not None for c in mon._fast._graphs.values())   # the split graphs were captured and used
    served = [r for r in caplog.records if r.getMessage() == "recommendation_served"]
    assert len(served) == 18 and all(r.user_id == "u7" for r in served)
    # ICREC_USE_GRAPH=0: the event-timed kernel-by-kernel path, same results
    monkeypatch.setenv("ICREC_USE_GRAPH", "0")
    plain = MonitoredRecommender(world["model_dir"], world["corpus_path"])
    assert plain._fast is None
    assert plain.recommend(queries[1], 10, user_id="x") == mon.recommend(queries[1], 10)


def test_two_different_batches_with_equal_split_point(world):
    """encode_packed_host on two DIFFERENT batches of the same size whose first halves hold the same token count:
    the second half's rebased cu_seqlens must come from the call's own sequence boundaries (an address-keyed cache
    once handed the second batch the first batch's boundaries)."""
    import torch

    from instacart_next_order_recommendation_amd.encoder import DeviceEncoder
    from instacart_next_order_recommendation_amd.model_io import load_model_dir

    lm = load_model_dir(world["model_dir"])
    enc = DeviceEncoder(lm.weights, lm.shape, "cuda:0")
    n = 2 * DeviceEncoder.SPLIT_MIN_SEQS
    rng = np.random.default_rng(3)

    def batch(lens_a, lens_b):
        lens = np.concatenate([lens_a, lens_b]).astype(np.int32)
        cu = np.zeros(n + 1, np.int32)
        np.cumsum(lens, out=cu[1:])
        ids = rng.integers(1000, lm.shape.vocab_size, size=int(cu[-1])).astype(np.int32)
        return ids, cu

    half = n // 2
    first = rng.integers(120, 200, size=half)
    assert first.sum() >= DeviceEncoder.SPLIT_MIN_TOKENS
    b1 = batch(first, rng.integers(120, 200, size=half))
    second_lens = rng.permutation(rng.integers(130, 256, size=half))     # different boundaries, longer rows
    b2 = batch(rng.permutation(first), second_lens)                      # same n, same token count in the first half
    assert b1[1][half] == b2[1][half] and not np.array_equal(b1[1], b2[1])
    outs = []
    for ids, cu in (b1, b2, b1):
        got = enc.encode_packed_host(ids, cu)                             # two-stream split inside
        want = enc.encode_packed(torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda(), int(np.diff(cu).max()))
        assert torch.equal(got, want)
        outs.append(got.clone())
    assert torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[1])
    enc.close()


def test_encode_without_flag_normalisation(world):
    """SentenceTransformer.encode(normalize_embeddings=False): one normalisation pass fewer.  The synthetic model
    directory has a Normalize module, so the output is still a unit vector and equals the default up to one
    division by ~1.0; without that module it would be the raw mean pool."""
    from instacart_next_order_recommendation_amd.recommender import Recommender

    rec = Recommender(world["model_dir"], world["corpus_path"])
    texts = ["[+3d w1h10] Organic Milk, Whole Wheat Bread.", "Product: Oat Milk. Aisle: aisle 3. Department: dept tea."]
    a = rec.model.encode(texts, normalize_embeddings=True)
    b = rec.model.encode(texts, normalize_embeddings=False)
    assert a.shape == b.shape == (2, 384)
    assert np.abs(a - b).max() < 1e-6
    assert rec.model._encoder_no_flag is not None and rec.model._encoder_no_flag.shape.n_normalize == rec.model.shape.n_normalize - 1


def test_cli_runs_demo_query(tmp_path, capsys):
    """`python -m instacart_next_order_recommendation_amd --config ...` (reference: python -m src.inference)."""
    import json

    import yaml

    from instacart_next_order_recommendation_amd import cli, synthetic as syn
    from instacart_next_order_recommendation_amd.model_io import write_synthetic_model_dir

    model_dir = write_synthetic_model_dir(tmp_path / "model", seed=1)
    corpus = tmp_path / "eval_corpus.json"
    corpus.write_text(json.dumps(syn.synthetic_catalog(200)))
    (tmp_path / "eval_queries.json").write_text(json.dumps({"42": "[+1d w2h9] Greek Yogurt, Honey."}))
    for extra, argv, origin in (({}, [], "built-in demo context"), ({"query": "[+2d w0h8] Oat Milk."}, [], "(config)"),
                                ({"eval_query_id": "42"}, [], "eval_queries.json[42]"),
                                ({"query": "ignored"}, ["--query", "[+3d w1h7] Rye Bread."], "(command line)")):
        cfg = tmp_path / "inference.yaml"
        cfg.write_text(yaml.safe_dump({"model_dir": str(model_dir), "corpus": str(corpus), "use_index": False,
                                       "top_k": 3, "corpus_hf_repo": "ignored/offline", **extra}))
        assert cli.main(["--config", str(cfg), *argv]) == 0
        out = capsys.readouterr().out.splitlines()
        assert origin in out[0] and len(out) == 4 and all("Product:" in ln for ln in out[1:])
    assert cli.main(["--config", str(cfg), "--json", "--top-k", "2"]) == 0
    rows = [json.loads(ln) for ln in capsys.readouterr().out.splitlines()]
    assert [r["rank"] for r in rows] == [1, 2] and set(rows[0]) == {"rank", "product_id", "score", "product_text"}


def test_graph_path_ignores_rows_outside_the_sequence(world):
    """The hipGraph fast path encodes a fixed row count (the token bucket); rows past the real sequence hold whatever
    the buffers held before.  Poison every buffer the captured graph owns: the result must not change."""
    from instacart_next_order_recommendation_amd.recommender import Recommender

    rec = Recommender(world["model_dir"], world["corpus_path"])
    q = world["queries"][1]
    want = rec.recommend(q, top_k=10)
    assert rec._fast is not None and rec._fast._graphs
    for cap in rec._fast._graphs.values():
        cap.enc_ws.fill_(0xFF)      # NaN patterns in every workspace byte (contexts / planes of the unused rows)
        cap.srch_ws.fill_(0xFF)
    got = rec.recommend(q, top_k=10)
    assert got == want and all(np.isfinite(s) for _, s in got)


def test_recommend_batches_pipeline_equals_recommend_batch(world):
    """recommend_batches() overlaps the host tokenisation of batch j+1 with the GPU work of batch j and reads batch j
    back after batch j+1 has been launched: every batch must equal recommend_batch() on the same inputs, bit for bit
    (same launches, one stream, in order) - with exclusions, ragged batch sizes and an empty batch in the stream."""
    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.recommender import Recommender

    rec = Recommender(world["model_dir"], world["corpus_path"])
    qs = syn.synthetic_user_contexts(37, seed=21)
    batches = [qs[:8], qs[8:9], [], qs[9:30], qs[30:]]
    excl = [[set(rec.product_ids[:20]) if i % 3 == 0 else None for i in range(len(b))] for b in batches]
    got = list(rec.recommend_batches(batches, top_k=7, exclude_product_ids=excl))
    assert len(got) == len(batches)
    for b, e, g in zip(batches, excl, got):
        assert g == rec.recommend_batch(b, 7, e)
    # a generator input and no exclusions
    got2 = list(rec.recommend_batches((b for b in batches), top_k=7))
    assert [len(g) for g in got2] == [len(b) for b in batches]
    assert got2[3] == rec.recommend_batch(batches[3], 7)


def test_index_used_from_the_callers_stream_inside_a_recommend_batches_loop(world):
    """While the generator is suspended at `yield`, batch j+1's search is already in flight on the pipeline's side
    stream.  A consumer that searches the SAME index from its own stream between yields (recommend_batch, scores) must
    not disturb it, nor be disturbed: the index keeps one scratch block per stream (pipeline.py)."""
    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.recommender import Recommender

    rec = Recommender(world["model_dir"], world["corpus_path"])
    qs = syn.synthetic_user_contexts(96, seed=33)
    batches = [qs[i:i + 16] for i in range(0, 96, 16)]
    want = [rec.recommend_batch(b, 9) for b in batches]
    other = syn.synthetic_user_contexts(24, seed=34)
    want_other = rec.recommend_batch(other, 9)
    got = []
    for j, g in enumerate(rec.recommend_batches(batches, top_k=9)):
        assert rec.recommend_batch(other, 9) == want_other  # same index, caller's stream, a different query count
        got.append(g)
    assert got == want
    assert len(rec._index._ws_by_stream) >= 2  # the two streams really had their own scratch blocks
