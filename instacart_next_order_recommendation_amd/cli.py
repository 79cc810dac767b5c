"""Command-line front of the recommender: one query in, the ranked products out.

Takes the YAML file the reference's `python -m src.inference` takes (configs/inference.yaml; keys model_dir, corpus,
use_index, query, eval_query_id, top_k — the drop-in contract, serve_recommendations.py:296-340) so an existing
deployment's config keeps working, but it is a different tool: the query can also come from the command line, the
output is either a table or one JSON object per result, and nothing is fetched from a hub (`corpus_hf_repo*` keys are
ignored; a missing model directory or corpus file is an error).

    python -m instacart_next_order_recommendation_amd --config configs/inference.yaml [--query "..."] [--json]
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import yaml

_DEFAULTS = {"model_dir": "models/two_tower_sbert/final", "corpus": "processed/p5_mp20_ef0.1/eval_corpus.json",
             "use_index": True, "query": None, "eval_query_id": None, "top_k": 10}
_FALLBACK_QUERY = "[+7d w4h14] Organic Milk, Whole Wheat Bread."  # the reference's demo context


def read_settings(path: Path | None) -> dict:
    """YAML -> settings dict with the reference's defaults filled in; unknown keys are dropped."""
    raw = yaml.safe_load(Path(path or "configs/inference.yaml").read_text()) or {}
    cfg = {key: (raw[key] if raw.get(key) is not None else default) for key, default in _DEFAULTS.items()}
    cfg["model_dir"], cfg["corpus"] = Path(str(cfg["model_dir"])), Path(str(cfg["corpus"]))
    cfg["top_k"], cfg["use_index"] = int(cfg["top_k"]), bool(cfg["use_index"])
    return cfg


def pick_query(cfg: dict, override: str | None) -> tuple[str, str]:
    """(query text, where it came from): --query beats eval_query_id beats `query` beats the demo context."""
    if override:
        return override, "command line"
    if cfg["eval_query_id"]:
        table = json.loads((cfg["corpus"].parent / "eval_queries.json").read_text())
        try:
            return table[cfg["eval_query_id"]], f"eval_queries.json[{cfg['eval_query_id']}]"
        except KeyError:
            raise SystemExit(f"eval_query_id {cfg['eval_query_id']!r} is not in {cfg['corpus'].parent / 'eval_queries.json'}")
    if cfg["query"]:
        return str(cfg["query"]), "config"
    return _FALLBACK_QUERY, "built-in demo context"


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="instacart_next_order_recommendation_amd", description=__doc__.splitlines()[0])
    ap.add_argument("--config", type=Path, default=None, help="YAML settings (default: configs/inference.yaml)")
    ap.add_argument("--query", default=None, help="user context to rank for (overrides the config)")
    ap.add_argument("--top-k", type=int, default=None, help="overrides top_k of the config")
    ap.add_argument("--json", action="store_true", help="one JSON object per result instead of the table")
    args = ap.parse_args(argv)
    cfg = read_settings(args.config)
    if not cfg["corpus"].exists():
        raise SystemExit(f"corpus file {cfg['corpus']} does not exist (this build never downloads one)")
    query, origin = pick_query(cfg, args.query)
    top_k = args.top_k or cfg["top_k"]

    from .recommender import Recommender

    rec = Recommender(model_dir=cfg["model_dir"], corpus_path=cfg["corpus"], use_index=cfg["use_index"])
    hits = rec.recommend(query=query, top_k=top_k)
    if args.json:
        for rank, (pid, score) in enumerate(hits, 1):
            print(json.dumps({"rank": rank, "product_id": pid, "score": score, "product_text": rec.pid_to_text[pid]}))
        return 0
    shown = query if len(query) <= 200 else query[:200] + " ..."
    print(f"context ({origin}): {shown}")
    width = max((len(pid) for pid, _ in hits), default=1)
    for rank, (pid, score) in enumerate(hits, 1):
        print(f"{rank:>3}  {pid:>{width}}  {score:7.4f}  {rec.pid_to_text[pid]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
