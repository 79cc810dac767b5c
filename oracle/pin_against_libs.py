#!/usr/bin/env python3
"""Pin the C oracle against the third-party libraries the reference calls, and
write the golden fixtures under tests/golden/.

Run in the build container (CPU only):   python oracle/pin_against_libs.py

What is compared (the reference module itself cannot be imported
here because sentence_transformers / dotenv / slowapi are not installed):

  encoder   transformers.BertModel (installed 5.15.0; reference pins 5.1.0)
            built from a local BertConfig at the all-MiniLM-L6-v2 shape with the
            repo's seeded synthetic weights, run on a PADDED batch with an
            attention mask exactly as SentenceTransformer.encode would
            (serve_recommendations.py:195-200), followed by mean pooling and two
            L2 normalisations written with torch ops.
  cos_sim   torch.nn.functional.normalize(a), normalize(b), torch.mm(a, b.T)
            (what sentence_transformers.util.cos_sim does; called at
            serve_recommendations.py:214).
  ranking   scores.argsort(descending=True) + the reference's own Python
            exclusion loop (serve_recommendations.py:215-225), restated inline.

The fixtures hold inputs + library outputs + oracle outputs; tests/ check the
oracle against the library outputs (not-gpu) and the HIP path against the
oracle outputs (gpu).  This script is the generator the fixtures came from.
"""
from __future__ import annotations

import hashlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from instacart_next_order_recommendation_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def hf_encode(blob: np.ndarray, shape: syn.BertShape, ids: np.ndarray, cu: np.ndarray):
    """The reference's encode() restated with the installed libraries."""
    from transformers import BertConfig, BertModel

    cfg = BertConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden, num_hidden_layers=shape.layers,
                     num_attention_heads=shape.heads, intermediate_size=shape.intermediate,
                     hidden_act="gelu", max_position_embeddings=shape.max_position,
                     type_vocab_size=shape.type_vocab, layer_norm_eps=shape.ln_eps,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    sd = {k: torch.from_numpy(v.copy()) for k, v in syn.blob_to_state_dict(blob, shape).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m for m in missing), missing
    n = cu.size - 1
    L = int(np.max(np.diff(cu)))
    pad_ids = np.zeros((n, L), np.int64)
    mask = np.zeros((n, L), np.int64)
    for s in range(n):
        ln = cu[s + 1] - cu[s]
        pad_ids[s, :ln] = ids[cu[s]:cu[s + 1]]
        mask[s, :ln] = 1
    with torch.no_grad():
        hs = model(input_ids=torch.from_numpy(pad_ids), attention_mask=torch.from_numpy(mask)).last_hidden_state
        m = torch.from_numpy(mask).unsqueeze(-1).to(hs.dtype)
        pooled = (hs * m).sum(1) / m.sum(1).clamp(min=1e-9)          # ST Pooling(mean)
        emb = pooled
        for _ in range(shape.n_normalize):                            # ST Normalize + normalize_embeddings
            emb = torch.nn.functional.normalize(emb, p=2, dim=1)
    hidden_packed = np.concatenate([hs[s, : cu[s + 1] - cu[s]].numpy() for s in range(n)], 0)
    return emb.numpy(), hidden_packed


def torch_cos_sim_rank(q: np.ndarray, P: np.ndarray, k: int, excl):
    a = torch.nn.functional.normalize(torch.from_numpy(q), p=2, dim=1)
    b = torch.nn.functional.normalize(torch.from_numpy(P), p=2, dim=1)
    scores = torch.mm(a, b.transpose(0, 1))
    idx_out = np.full((q.shape[0], k), -1, np.int64)
    sc_out = np.zeros((q.shape[0], k), np.float32)
    for qi in range(q.shape[0]):
        order = scores[qi].argsort(descending=True)
        ex = set(excl[qi]) if excl is not None else set()
        got = 0
        for i in order.tolist():
            if i in ex:
                continue
            idx_out[qi, got] = i
            sc_out[qi, got] = float(scores[qi, i])
            got += 1
            if got >= k:
                break
    return scores.numpy(), idx_out, sc_out


def main() -> None:
    torch.manual_seed(0)
    torch.set_num_threads(8)
    GOLD.mkdir(parents=True, exist_ok=True)
    shape = syn.BertShape()
    ocfg = oracle.make_cfg(n_normalize=shape.n_normalize)

    # ---------------------------------------------------------------- encoder
    blob = syn.synthetic_bert_weights(shape, seed=0)
    assert blob.size == oracle.weight_count(ocfg) == shape.weight_count()
    ids, cu = syn.synthetic_token_batch(8, seed=11, mean_len=24, std_len=10, lo=4, hi=48)
    hf_emb, hf_hidden = hf_encode(blob, shape, ids, cu)
    or_emb, or_hidden = oracle.encode(blob, ocfg, ids, cu, return_hidden=True)
    d_emb = float(np.abs(hf_emb - or_emb).max())
    d_hid = float(np.abs(hf_hidden - or_hidden).max())
    cos = (hf_emb * or_emb).sum(1)
    print(f"encoder: max|emb_hf - emb_oracle| = {d_emb:.3e}   max|hidden diff| = {d_hid:.3e}   min cos = {cos.min():.8f}")
    assert d_emb < 2e-6 and d_hid < 5e-5, "oracle encoder disagrees with transformers.BertModel"

    # long sequences (query-like, up to 256 tokens)
    ids2, cu2 = syn.synthetic_token_batch(3, seed=12, mean_len=200, std_len=60, lo=100, hi=256)
    hf_emb2, _ = hf_encode(blob, shape, ids2, cu2)
    or_emb2 = oracle.encode(blob, ocfg, ids2, cu2)
    d2 = float(np.abs(hf_emb2 - or_emb2).max())
    print(f"encoder (long): lens={np.diff(cu2).tolist()} max|diff| = {d2:.3e}")
    assert d2 < 2e-6

    np.savez_compressed(
        GOLD / "encoder_minilm_seed0.npz",
        weights_seed=np.int64(0), weights_sha256=np.array(sha(blob)),
        ids=ids, cu_seqlens=cu, hf_embeddings=hf_emb, oracle_embeddings=or_emb,
        hf_hidden_row0=hf_hidden[0], oracle_hidden_row0=or_hidden[0],
        ids_long=ids2, cu_seqlens_long=cu2, hf_embeddings_long=hf_emb2, oracle_embeddings_long=or_emb2,
    )

    # ---------------------------------------------------- similarity + ranking
    k = 20
    P = syn.synthetic_embeddings(1024, 384, seed=1)
    # un-normalise the inputs a little so that cos_sim's re-normalisation matters
    scale = 0.5 + uniform_f32(5, 1024)[:, None]
    P_raw = (P * scale).astype(np.float32)
    q = syn.synthetic_embeddings(16, 384, seed=2)
    q_raw = (q * (0.5 + uniform_f32(6, 16)[:, None])).astype(np.float32)
    excl = [[] for _ in range(16)]
    t_scores, t_idx0, _ = torch_cos_sim_rank(q_raw, P_raw, k, None)
    for qi in range(8, 16):  # second half of the queries exclude some of their own best hits
        excl[qi] = sorted(set(t_idx0[qi, [0, 2, 5]].tolist() + [int(v) for v in (hash_rows(7 + qi, 6) % 1024)]))
    t_scores, t_idx, t_sc = torch_cos_sim_rank(q_raw, P_raw, k, excl)
    o_scores = oracle.scores(oracle.normalize_rows(q_raw), oracle.normalize_rows(P_raw))
    o_idx, o_sc = oracle.search(q_raw, P_raw, k, excl)
    ds = float(np.abs(t_scores - o_scores).max())
    # gap analysis: a row is "ambiguous" when two neighbouring torch scores in its
    # top-(k+1+|excl|) differ by less than 4e-6
    amb = []
    for qi in range(16):
        srt = np.sort(t_scores[qi])[::-1][: k + 1 + len(excl[qi])]
        amb.append(bool((srt[:-1] - srt[1:]).min() < 4e-6))
    amb = np.asarray(amb)
    same = np.array([np.array_equal(t_idx[qi], o_idx[qi]) for qi in range(16)])
    print(f"cos_sim: max|scores_torch - scores_oracle| = {ds:.3e}; top-{k} rows identical: {same.sum()}/16; ambiguous rows: {amb.sum()}")
    assert ds < 1e-4 and (same | amb).all(), "oracle ranking disagrees with torch outside near-ties"
    assert float(np.abs(t_sc - o_sc)[same].max()) < 1e-4

    # tie policy fixture: duplicate rows => exactly equal scores; the build's own policy
    # (score desc, row asc) is asserted, torch's unstable argsort is only set-compared.
    P_tie = P_raw.copy()
    P_tie[[100, 300, 900]] = P_tie[7]
    P_tie[[5, 55]] = P_tie[501]
    tie_idx, tie_sc = oracle.search(q_raw[:4], P_tie, k, None)
    tt_scores, tt_idx, _ = torch_cos_sim_rank(q_raw[:4], P_tie, k, None)

    np.savez_compressed(
        GOLD / "search_n1024_q16_k20.npz",
        q=q_raw, P=P_raw, k=np.int64(k),
        excl_flat=np.concatenate([np.asarray(e, np.int32) for e in excl]).astype(np.int32),
        excl_off=np.cumsum([0] + [len(e) for e in excl]).astype(np.int32),
        torch_scores=t_scores, torch_idx=t_idx, torch_topk_scores=t_sc, torch_ambiguous=amb,
        oracle_scores=o_scores, oracle_idx=o_idx, oracle_topk_scores=o_sc,
        tie_rows=np.asarray([7, 100, 300, 900, 501, 5, 55], np.int64), P_tie=P_tie,
        oracle_tie_idx=tie_idx, oracle_tie_scores=tie_sc, torch_tie_idx=tt_idx,
    )

    # full-size catalog: inputs regenerated from the seeded generator (sha-checked), outputs stored
    Pf = syn.synthetic_embeddings(49688, 384, seed=1)
    qf = syn.synthetic_embeddings(8, 384, seed=2)
    f_scores, f_idx, f_sc = torch_cos_sim_rank(qf, Pf, k, None)
    fo_idx, fo_sc = oracle.search(qf, Pf, k, None)
    ambf = []
    for qi in range(8):
        srt = np.sort(f_scores[qi])[::-1][: k + 1]
        ambf.append(bool((srt[:-1] - srt[1:]).min() < 4e-6))
    ambf = np.asarray(ambf)
    samef = np.array([np.array_equal(f_idx[qi], fo_idx[qi]) for qi in range(8)])
    print(f"full catalog: identical rows {samef.sum()}/8, ambiguous {ambf.sum()}, max|score diff| = {np.abs(f_sc - fo_sc).max():.3e}")
    assert (samef | ambf).all()
    np.savez_compressed(
        GOLD / "search_n49688_q8_k20.npz",
        P_seed=np.int64(1), q_seed=np.int64(2), P_sha256=np.array(sha(Pf)), q_sha256=np.array(sha(qf)),
        k=np.int64(k), torch_idx=f_idx, torch_topk_scores=f_sc, torch_ambiguous=ambf,
        oracle_idx=fo_idx, oracle_topk_scores=fo_sc,
    )
    print("fixtures written to", GOLD)


def uniform_f32(stream: int, n: int) -> np.ndarray:
    return syn.uniform(77, stream, n).astype(np.float32)


def hash_rows(stream: int, n: int) -> np.ndarray:
    return syn.hash_u64(78, stream, n)


if __name__ == "__main__":
    main()
