"""Thin CLI equivalent of the reference's `python -m src.inference`
(src/inference/serve_recommendations.py:296-373: InferenceConfig + main()).

Same YAML keys (configs/inference.yaml): model_dir, corpus, use_index, query, eval_query_id, top_k.
`corpus_hf_repo*` are accepted and ignored: this build never reaches for the hub — a missing model
directory or corpus file is an error.  Relative paths resolve against the current directory.

    python -m instacart_next_order_recommendation_amd --config configs/inference.yaml
"""
from __future__ import annotations

import argparse
import json
import logging
from pathlib import Path

import yaml

EVAL_QUERIES_FILENAME = "eval_queries.json"  # reference: src/constants.py
DEMO_QUERY = "[+7d w4h14] Organic Milk, Whole Wheat Bread."  # serve_recommendations.py:364


class InferenceConfig:
    """Loads inference config from YAML. Attributes: model_dir, corpus, use_index, query, eval_query_id, top_k."""

    def __init__(self, raw: dict):
        self.model_dir = Path(raw.get("model_dir", "models/two_tower_sbert/final"))
        self.corpus = Path(raw.get("corpus") or "processed/p5_mp20_ef0.1/eval_corpus.json")
        self.use_index = bool(raw.get("use_index", True))
        self.query = raw.get("query")
        self.eval_query_id = raw.get("eval_query_id")
        self.top_k = int(raw.get("top_k", 10))

    @classmethod
    def load(cls, config_path: Path | None = None) -> "InferenceConfig":
        path = Path(config_path) if config_path else Path("configs/inference.yaml")
        with open(path) as f:
            return cls(yaml.safe_load(f) or {})


def main(argv=None) -> None:
    """Load config, create Recommender, run the configured (or demo) query and print top-k."""
    parser = argparse.ArgumentParser(description="Serve product recommendations (MI355X)")
    parser.add_argument("--config", type=Path, default=None, help="Path to YAML config (default: configs/inference.yaml)")
    args = parser.parse_args(argv)
    cfg = InferenceConfig.load(args.config)
    logging.basicConfig(level=logging.INFO, format="%(message)s")

    from .recommender import Recommender

    if not cfg.corpus.exists():
        raise FileNotFoundError(f"corpus {cfg.corpus} not found (no hub fallback in this build)")
    rec = Recommender(model_dir=cfg.model_dir, corpus_path=cfg.corpus, use_index=cfg.use_index)

    if cfg.eval_query_id:
        queries_path = cfg.corpus.parent / EVAL_QUERIES_FILENAME
        with open(queries_path) as f:
            eval_queries = json.load(f)
        if cfg.eval_query_id not in eval_queries:
            raise KeyError(f"eval_query_id {cfg.eval_query_id} not in {queries_path}")
        query = eval_queries[cfg.eval_query_id]
        print(f"Query (eval_id={cfg.eval_query_id}):\n  {query[:200]}...\n")
    elif cfg.query:
        query = cfg.query
        print(f"Query:\n  {query}\n")
    else:
        query = DEMO_QUERY
        print("No query or eval_query_id in config. Using demo query:\n")
        print(f"  {query}\n")

    results = rec.recommend(query=query, top_k=cfg.top_k)
    print(f"Top-{cfg.top_k} recommendations:")
    for i, (pid, score) in enumerate(results, 1):
        print(f"  {i}. product_id={pid} (score={score:.4f}) {rec.pid_to_text[pid]}")


if __name__ == "__main__":
    main()
