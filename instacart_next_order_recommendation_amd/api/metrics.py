"""Prometheus metrics with the reference's names (src/api/metrics.py:13-66).  Buckets get finer
low ends: GPU latencies sit far below the reference's 10 ms / 50 ms floors."""
from __future__ import annotations

from prometheus_client import CollectorRegistry, Counter, Gauge, Histogram

API_REGISTRY = CollectorRegistry()

RECOMMENDATION_REQUESTS_TOTAL = Counter("recommendation_requests_total", "Total /recommend requests", ["status"],
                                        registry=API_REGISTRY)
RECOMMENDATION_LATENCY_SECONDS = Histogram(
    "recommendation_latency_seconds", "End-to-end /recommend latency",
    buckets=(0.0005, 0.001, 0.0025, 0.005, 0.01, 0.025, 0.05, 0.1, 0.5, 1.0, 5.0), registry=API_REGISTRY)
RECOMMENDATION_ENCODE_SECONDS = Histogram(
    "recommendation_encode_seconds", "Query embedding (tokenise + encoder forward) time",
    buckets=(0.00025, 0.0005, 0.001, 0.0025, 0.005, 0.01, 0.05, 0.1, 0.5, 1.0), registry=API_REGISTRY)
RECOMMENDATION_BATCH_SIZE = Histogram(
    "recommendation_batch_size", "Requests coalesced into one GPU pass",
    buckets=(1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024), registry=API_REGISTRY)
MODEL_LOADED = Gauge("model_loaded", "1 when the recommender is loaded", registry=API_REGISTRY)
