"""Host-side logic that needs no GPU: EmbeddingIndex cache format, model-dir IO, tokenizer stage,
packing helpers, shard bounds, synthetic generators."""
from __future__ import annotations

import hashlib
import json
import os
import time

import numpy as np
import pytest

from instacart_next_order_recommendation_amd import synthetic as syn


def test_embedding_index_roundtrip_and_layout(tmp_path):
    from instacart_next_order_recommendation_amd.recommender import EmbeddingIndex

    corpus = tmp_path / "eval_corpus.json"
    corpus.write_text(json.dumps({"1": "a", "2": "b", "3": "c"}))
    ids = ["1", "2", "3"]
    emb = np.arange(3 * 384, dtype=np.float32).reshape(3, 384)
    ix = EmbeddingIndex(corpus, "models/two_tower_sbert/final")
    assert ix.load(ids) is None  # miss
    ix.save(ids, emb)
    # layout the reference writes (serve_recommendations.py:69-73,112-130; constants.py:88-92)
    want_dir = corpus.resolve().parent / ".embedding_index" / hashlib.sha256(
        f"models/two_tower_sbert/final|{corpus.resolve()}".encode()).hexdigest()[:16]
    assert ix.directory == want_dir
    assert sorted(p.name for p in want_dir.iterdir()) == ["embeddings.npy", "manifest.json", "product_ids.json"]
    man = json.loads((want_dir / "manifest.json").read_text())
    assert set(man) == {"corpus_path", "model_dir", "corpus_mtime", "n_products"} and man["n_products"] == 3
    assert np.load(want_dir / "embeddings.npy").dtype == np.float32
    np.testing.assert_array_equal(ix.load(ids), emb)
    # invalidation rules (:90-110): id list, model dir, corpus mtime
    assert ix.load(["1", "3", "2"]) is None
    assert EmbeddingIndex(corpus, "other/model").load(ids) is None
    time.sleep(0.01)
    os.utime(corpus, (time.time() + 5, time.time() + 5))
    assert ix.load(ids) is None
    ix.save(ids, emb)
    (want_dir / "manifest.json").write_text("{not json")
    assert ix.load(ids) is None


def test_model_dir_roundtrip(tmp_path):
    from instacart_next_order_recommendation_amd.model_io import load_model_dir, write_synthetic_model_dir

    d = write_synthetic_model_dir(tmp_path / "m", seed=4)
    m = load_model_dir(d)
    assert m.shape.hidden == 384 and m.shape.layers == 6 and m.shape.n_normalize == 2 and m.max_seq_length == 256
    want = syn.synthetic_bert_weights(m.shape, seed=4)
    np.testing.assert_array_equal(m.weights, want)
    with pytest.raises(FileNotFoundError):
        load_model_dir("sentence-transformers/all-MiniLM-L6-v2")  # hub ids cannot be fetched offline


def test_tokenizer_stage(tmp_path):
    from instacart_next_order_recommendation_amd.model_io import load_model_dir, write_synthetic_model_dir

    m = load_model_dir(write_synthetic_model_dir(tmp_path / "m"))
    ids = m.tokenizer(["[+7d w4h14] Organic Milk, Whole Wheat Bread.", "Product: Banana. Aisle: fresh fruits."])
    assert all(s[0] == 101 and s[-1] == 102 for s in ids)       # [CLS] ... [SEP]
    assert 100 not in ids[0]                                     # no [UNK] on the synthetic formats
    long = m.tokenizer(["milk " * 1000])[0]
    assert len(long) == 256 and long[-1] == 102                  # truncation to max_seq_length keeps [SEP]
    # agreement with transformers' BertTokenizer on the same vocab (the reference's tokenizer class)
    from transformers import BertTokenizer

    ref = BertTokenizer(str(tmp_path / "m" / "vocab.txt"), do_lower_case=True)
    for text in syn.synthetic_user_contexts(20, seed=3) + list(syn.synthetic_catalog(20).values()):
        assert m.tokenizer([text])[0] == ref(text, truncation=True, max_length=256)["input_ids"]


def test_pack_token_ids():
    from instacart_next_order_recommendation_amd.encoder import pack_token_ids

    ids, cu, mx = pack_token_ids([[1, 2, 3], [4], [5, 6]])
    assert ids.tolist() == [1, 2, 3, 4, 5, 6] and cu.tolist() == [0, 3, 4, 6] and mx == 3 and ids.dtype == np.int32
    with pytest.raises(ValueError):
        pack_token_ids([[1], []])
    with pytest.raises(ValueError):
        pack_token_ids([[0] * 257])


def test_shard_bounds():
    from instacart_next_order_recommendation_amd.sharded import shard_bounds

    assert shard_bounds(49688, 8) == [6211 * i for i in range(9)]  # 8 x 6,211 exactly
    b = shard_bounds(10, 4)
    assert b == [0, 3, 6, 8, 10]


def test_synthetic_formats():
    cat = syn.synthetic_catalog(50)
    assert list(cat)[:3] == ["1", "2", "3"]
    assert all(t.startswith("Product: ") and ". Aisle: " in t and ". Department: " in t and t.endswith(".") for t in cat.values())
    ctx = syn.synthetic_user_contexts(50)
    assert all(c.startswith("[+") and c.endswith(".") for c in ctx)
    ids, cu = syn.synthetic_token_batch(1024)
    ln = np.diff(cu)
    assert ln.min() >= 16 and ln.max() <= 256 and 120 < ln.mean() < 136
    assert (ids[cu[:-1]] == 101).all() and (ids[cu[1:] - 1] == 102).all()
    e = syn.synthetic_embeddings(100, 384, seed=5)
    assert np.abs(np.linalg.norm(e.astype(np.float64), axis=1) - 1).max() < 1e-6


def test_bench_bare_multi_gpu_form_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 4` without WORLD_SIZE (the form the driver uses) must not die at start-up: it
    re-launches itself under torch.distributed.run as a CHILD process - before importing torch or touching a GPU - with the same
    arguments, a loopback rendezvous, and exits with the child's code."""
    import importlib.util
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    spec = importlib.util.spec_from_file_location("bench_under_test", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    imported_before = "torch" in sys.modules
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7                                   # the child's exit code
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert ("torch" in sys.modules) == imported_before           # decided before torch is imported by bench.main
