"""Parity at the sizes BASELINE.json's configs are quoted on (the shapes every perf claim rests on).

configs[2]  the bench batch itself: 1,024 contexts / 131,150 tokens, both GEMM modes: >= 160 of its sequences (the
            first 128 — the ones bench.py's own `checked_vs_oracle` covers — plus the extremes and a random draw)
            against the oracle (<= 5e-6); those same sequences re-encoded as a batch of their own must give the same
            bits (batch invariance); every row of the full batch a finite unit vector
configs[1]  the 49,688-product catalog encode, a sample of rows against the oracle
configs[4]  a 2 M-row bf16 / bf16+filter catalog against oracle.search(storage="bf16") on a query sample,
            and the full 10 M-row catalog through size-independent properties
Reference call sites: src/inference/serve_recommendations.py:195-200 (catalog encode), :213-225 (request).
"""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EMB_TOL = 5e-6  # same bar as tests/test_encoder_gpu.py


@pytest.fixture(scope="module")
def cuda():
    import torch

    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("mode", ["f16x3", "f32"])
def test_bench_batch_encode_vs_oracle(cuda, minilm_weights, mode):
    """bench.py's step input (synthetic_token_batch(1024, seed=1234)): >= 160 of its 1,024 sequences vs oracle.encode."""
    torch = cuda
    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.encoder import DeviceEncoder
    from oracle import oracle

    ids, cu = syn.synthetic_token_batch(1024, seed=1234)
    assert int(cu[-1]) == 131150
    enc = DeviceEncoder(minilm_weights, gemm_mode=mode)
    emb = enc.encode_packed(torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda(), int(np.diff(cu).max())).cpu().numpy()
    assert np.isfinite(emb).all() and np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-6
    lens = np.diff(cu)
    pick = sorted({*range(128), 1023, int(lens.argmin()), int(lens.argmax()),
                   *np.random.default_rng(7).integers(128, 1024, 40).tolist()})
    assert len(pick) >= 160
    sub_ids = np.concatenate([ids[cu[s]:cu[s + 1]] for s in pick])
    sub_cu = np.concatenate([[0], np.cumsum([lens[s] for s in pick])]).astype(np.int32)
    oracle.set_threads(oracle.usable_cpus())
    want = oracle.encode(minilm_weights, oracle.make_cfg(), sub_ids, sub_cu)
    err = np.abs(emb[pick] - want).max()
    print(f"[{mode}] 1,024-context batch: max|emb - oracle| over {len(pick)} sampled sequences = {err:.3e}")
    assert err < EMB_TOL
    # the sampled sequences encode to the same bits on their own (batch invariance at this size)
    alone = enc.encode_packed(torch.from_numpy(sub_ids).cuda(), torch.from_numpy(sub_cu).cuda(), int(np.diff(sub_cu).max())).cpu().numpy()
    np.testing.assert_array_equal(alone, emb[pick])
    enc.close()


def test_catalog_encode_49688_products_vs_oracle(cuda, minilm_weights):
    """The start-up path at full size: 49,688 product-length sequences through encode_ids (chunked calls),
    32 sampled rows against the oracle, every row a unit vector."""
    from instacart_next_order_recommendation_amd import synthetic as syn
    from instacart_next_order_recommendation_amd.encoder import DeviceEncoder
    from oracle import oracle

    n = 49_688
    ids, cu = syn.synthetic_token_batch(n, seed=42, mean_len=20, std_len=5, lo=8, hi=40)
    seqs = [ids[cu[i]:cu[i + 1]] for i in range(n)]
    enc = DeviceEncoder(minilm_weights)
    emb = enc.encode_ids(seqs).cpu().numpy()
    assert emb.shape == (n, 384) and np.isfinite(emb).all()
    assert np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-6
    pick = sorted({0, n - 1, *np.random.default_rng(3).integers(0, n, 30).tolist()})
    sub_ids = np.concatenate([seqs[s] for s in pick])
    sub_cu = np.concatenate([[0], np.cumsum([len(seqs[s]) for s in pick])]).astype(np.int32)
    want = oracle.encode(minilm_weights, oracle.make_cfg(), sub_ids, sub_cu)
    err = np.abs(emb[pick] - want).max()
    print(f"49,688-product catalog: max|emb - oracle| over {len(pick)} sampled rows = {err:.3e}")
    assert err < EMB_TOL
    enc.close()


def _device_catalog(torch, n_rows: int, seed: int):
    """Clustered rows generated on the device (bench.py's 10m generator): 200 centres + 0.35 N(0, I)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    centres = torch.randn(200, 384, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    rows = torch.empty((n_rows, 384), device=dev)
    for s in range(0, n_rows, 1 << 18):
        m = min(1 << 18, n_rows - s)
        cid = torch.randint(0, 200, (m,), device=dev, generator=g)
        rows[s:s + m] = centres[cid] + 0.35 * torch.randn(m, 384, device=dev, generator=g)
    return rows


def test_2m_row_bf16_catalog_vs_oracle(cuda, monkeypatch):
    """2,000,000 x 384 bf16 rows (and bf16 + filter: the resident form with the rows as packed fragments, and the
    staged form with row-major planes): Q in {1, 64, 1024}; a sample of queries is checked bit for bit against
    oracle.search(storage="bf16") over the same fp32 input rows."""
    torch = cuda
    from instacart_next_order_recommendation_amd.search import DeviceIndex
    from oracle import oracle

    n = 2_000_000
    rows = _device_catalog(torch, n, seed=11)
    q_all = torch.nn.functional.normalize(rows[torch.randint(0, n, (1024,), device=rows.device)] +
                                          0.2 * torch.randn(1024, 384, device=rows.device), dim=1)
    P_host = rows.cpu().numpy()
    oracle.set_threads(oracle.usable_cpus())
    sample = [0, 63, 500, 1023]
    want_i, want_s = oracle.search(q_all[sample].cpu().numpy(), P_host, 20, None, storage="bf16")
    for storage, resident in (("bf16", "1"), ("bf16+filter", "1"), ("bf16+filter", "0")):
        monkeypatch.setenv("ICREC_FILTER_RESIDENT", resident)
        ix = DeviceIndex(rows, storage=storage)
        for nq in (1, 64, 1024):
            idx, sc = ix.search(q_all[:nq], 20)
            got = [s for s in sample if s < nq]
            sel = [sample.index(s) for s in got]
            np.testing.assert_array_equal(idx[got].cpu().numpy(), want_i[sel])
            np.testing.assert_array_equal(sc[got].cpu().numpy(), want_s[sel])
        ix.close()
    del rows


def test_10m_row_catalog_properties(cuda):
    """BASELINE configs[4] on one GPU: 10,000,000 x 384 as bf16 rows + filter planes.  Size-independent
    properties: every list sorted (score desc, row asc on ties), rows unique and in range, idempotent, the
    returned scores are the library's own exact score matrix at the returned rows, and the k-th score is the
    k-th largest of the full score row; the filter path equals the exact bf16 path."""
    torch = cuda
    from instacart_next_order_recommendation_amd.search import DeviceIndex

    n = 10_000_000
    rows = _device_catalog(torch, n, seed=1000)
    q = torch.nn.functional.normalize(rows[torch.randint(0, n, (256,), device=rows.device)] +
                                      0.2 * torch.randn(256, 384, device=rows.device), dim=1)
    ix = DeviceIndex(rows, storage="bf16+filter", row_offset=0)
    del rows
    torch.cuda.empty_cache()
    idx, sc = ix.search(q, 20)                      # filter + verify path (Q >= 256)
    idx2, sc2 = ix.search(q, 20)
    assert torch.equal(idx, idx2) and torch.equal(sc, sc2)
    i_h, s_h = idx.cpu().numpy(), sc.cpu().numpy()
    assert i_h.min() >= 0 and i_h.max() < n
    for r in range(256):
        assert len(set(i_h[r].tolist())) == 20
        d = np.diff(s_h[r])
        assert (d <= 0).all()
        ties = np.nonzero(d == 0)[0]
        assert all(i_h[r][t] < i_h[r][t + 1] for t in ties)
    # exact streaming / MFMA paths on subsets of the same queries give the same lists
    for nq in (1, 8, 64):
        i_s, s_s = ix.search(q[:nq], 20)
        assert torch.equal(i_s, idx[:nq]) and torch.equal(s_s, sc[:nq])
    # the library's own full score rows (exact fp32 chain, bit-exact vs the oracle at smaller sizes)
    full = ix.scores(q[:4])                         # [4, 10M] = 160 MB
    got = torch.gather(full, 1, idx[:4])
    assert torch.equal(got, sc[:4])
    top = torch.topk(full, 20, dim=1).values
    assert torch.equal(top, sc[:4])
    ix.close()
