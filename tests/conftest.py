"""pytest config: markers, paths, shared fixtures."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_encoder():
    return np.load(GOLDEN / "encoder_minilm_seed0.npz")


@pytest.fixture(scope="session")
def golden_search():
    return np.load(GOLDEN / "search_n1024_q16_k20.npz")


@pytest.fixture(scope="session")
def golden_search_full():
    return np.load(GOLDEN / "search_n49688_q8_k20.npz")


@pytest.fixture(scope="session")
def minilm_weights():
    """The seeded synthetic all-MiniLM-L6-v2-shaped weight blob (regenerated, ~90 MB)."""
    from instacart_next_order_recommendation_amd import synthetic as syn

    return syn.synthetic_bert_weights(syn.BertShape(), seed=0)


def excl_lists(flat, off):
    return [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
