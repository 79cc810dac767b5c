"""Deterministic synthetic inputs at the reference's shapes.

There is no network on the build or GPU boxes, so the fine-tuned
all-MiniLM-L6-v2 weights, the WordPiece vocab and the Kaggle catalog are
unavailable.  Everything here is generated from a counter-based
integer hash (splitmix64), using only integer ops and exact float64
add/multiply, so every machine regenerates bit-identical arrays and the 90 MB
weight blob never has to be shipped.

String formats follow the reference's data prep:
  product text  src/data/prepare_instacart_sbert.py:185-193
  user context  src/data/prepare_instacart_sbert.py:233-262, src/constants.py:60-66
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_u64(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit hashes of (seed, stream, offset + i)."""
    with np.errstate(over="ignore"):
        base = _splitmix(np.asarray([seed], np.uint64) ^ _splitmix(np.asarray([stream], np.uint64)))
        ctr = np.arange(offset, offset + n, dtype=np.uint64)
        return _splitmix(ctr ^ base)


def uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """float64 uniform [0,1) — exact (53 random bits * 2^-53)."""
    return (hash_u64(seed, stream, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normalish(seed: int, stream: int, n: int, std: float = 1.0) -> np.ndarray:
    """Approximately normal (Irwin-Hall of 4 uniforms, variance-matched), float32.

    Uses only exact float64 adds and one multiply, so it is bit-reproducible on
    any IEEE machine (no libm transcendental in the path)."""
    acc = np.zeros(n, np.float64)
    for j in range(4):
        acc += uniform(seed, stream * 4 + j + 1_000_003, n)
    return ((acc - 2.0) * (np.sqrt(3.0) * std)).astype(np.float32)


# --------------------------------------------------------------------- BERT


@dataclass(frozen=True)
class BertShape:
    """all-MiniLM-L6-v2 (configs/train.yaml:10)."""

    vocab_size: int = 30522
    hidden: int = 384
    layers: int = 6
    heads: int = 12
    intermediate: int = 1536
    max_position: int = 512
    type_vocab: int = 2
    ln_eps: float = 1e-12
    n_normalize: int = 2  # ST Normalize module + encode(normalize_embeddings=True)

    def blob_layout(self) -> list[tuple[str, tuple[int, ...]]]:
        """(name, shape) in the order documented in include/icrec.h."""
        H, I = self.hidden, self.intermediate
        out = [
            ("embeddings.word_embeddings.weight", (self.vocab_size, H)),
            ("embeddings.position_embeddings.weight", (self.max_position, H)),
            ("embeddings.token_type_embeddings.weight", (self.type_vocab, H)),
            ("embeddings.LayerNorm.weight", (H,)),
            ("embeddings.LayerNorm.bias", (H,)),
        ]
        for l in range(self.layers):
            p = f"encoder.layer.{l}."
            out += [
                (p + "attention.self.query.weight", (H, H)), (p + "attention.self.query.bias", (H,)),
                (p + "attention.self.key.weight", (H, H)), (p + "attention.self.key.bias", (H,)),
                (p + "attention.self.value.weight", (H, H)), (p + "attention.self.value.bias", (H,)),
                (p + "attention.output.dense.weight", (H, H)), (p + "attention.output.dense.bias", (H,)),
                (p + "attention.output.LayerNorm.weight", (H,)), (p + "attention.output.LayerNorm.bias", (H,)),
                (p + "intermediate.dense.weight", (I, H)), (p + "intermediate.dense.bias", (I,)),
                (p + "output.dense.weight", (H, I)), (p + "output.dense.bias", (H,)),
                (p + "output.LayerNorm.weight", (H,)), (p + "output.LayerNorm.bias", (H,)),
            ]
        return out

    def weight_count(self) -> int:
        return int(sum(int(np.prod(s)) for _, s in self.blob_layout()))


def synthetic_bert_weights(shape: BertShape = BertShape(), seed: int = 0, std: float = 0.05) -> np.ndarray:
    """Flat fp32 weight blob in include/icrec.h order.

    Matrices ~ N(0, std) (std 0.05 rather than BERT's 0.02 init so that attention
    is not uniform and the FFN is exercised off the linear part of GELU),
    biases ~ N(0, 0.02), LayerNorm gamma = 1 + N(0, 0.05), beta ~ N(0, 0.02)."""
    parts = []
    for t, (name, shp) in enumerate(shape.blob_layout()):
        n = int(np.prod(shp))
        if name.endswith("LayerNorm.weight"):
            a = 1.0 + normalish(seed, t, n, 0.05)
        elif name.endswith(".bias"):
            a = normalish(seed, t, n, 0.02)
        else:
            a = normalish(seed, t, n, std)
        parts.append(a.astype(np.float32))
    return np.concatenate(parts)


def blob_to_state_dict(blob: np.ndarray, shape: BertShape) -> dict[str, np.ndarray]:
    """Split a blob into HF BertModel parameter names (no 'bert.' prefix)."""
    out, o = {}, 0
    for name, shp in shape.blob_layout():
        n = int(np.prod(shp))
        out[name] = blob[o:o + n].reshape(shp)
        o += n
    assert o == blob.size
    return out


def state_dict_to_blob(sd: dict, shape: BertShape) -> np.ndarray:
    """Pack an HF state dict (numpy or torch tensors; optional 'bert.' prefix,
    pooler/position_ids ignored) into the blob order."""
    parts = []
    for name, shp in shape.blob_layout():
        t = sd[name] if name in sd else sd["bert." + name]
        a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
        if tuple(a.shape) != tuple(shp):
            raise ValueError(f"{name}: expected {shp}, got {a.shape}")
        parts.append(a.reshape(-1))
    return np.concatenate(parts)


def synthetic_token_batch(n_seqs: int, seed: int = 1234, mean_len: float = 128.0, std_len: float = 40.0,
                          lo: int = 16, hi: int = 256, vocab_size: int = 30522):
    """Kernel-only query batch: lengths ~ clipped Normal(128, 40) in
    [16, 256]; ids uniform in [1000, vocab) with [CLS]=101 first and [SEP]=102 last.
    Returns (ids int32[T], cu_seqlens int32[n_seqs+1])."""
    ln = np.clip(np.rint(mean_len + std_len * normalish(seed, 7, n_seqs, 1.0).astype(np.float64)), lo, hi).astype(np.int64)
    cu = np.zeros(n_seqs + 1, np.int64)
    np.cumsum(ln, out=cu[1:])
    T = int(cu[-1])
    ids = (1000 + (hash_u64(seed, 8, T) % np.uint64(vocab_size - 1000))).astype(np.int32)
    ids[cu[:-1]] = 101
    ids[cu[1:] - 1] = 102
    return ids, cu.astype(np.int32)


# ------------------------------------------------------------- embeddings


def synthetic_embeddings(n: int, d: int = 384, seed: int = 1, n_clusters: int = 200, noise: float = 0.35,
                         chunk: int = 1 << 16) -> np.ndarray:
    """Clustered unit vectors: 200 Gaussian centres + 0.35*N(0,I),
    L2-normalised in float64 then rounded to fp32 — realistic near-ties."""
    centres = normalish(99, 1, n_clusters * d, 1.0).reshape(n_clusters, d).astype(np.float64)
    out = np.empty((n, d), np.float32)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        cid = (hash_u64(seed, 2, m, s) % np.uint64(n_clusters)).astype(np.int64)
        x = centres[cid] + noise * normalish(seed, 3 + s // chunk * 16, m * d, 1.0).reshape(m, d).astype(np.float64)
        # stream id depends on the chunk so rows are independent of chunking only
        # for a fixed `chunk`; callers that need stable rows keep the default.
        x /= np.sqrt((x * x).sum(1, keepdims=True))
        out[s:s + m] = x.astype(np.float32)
    return out


# ------------------------------------------------------------------- text

_WORDS = (
    "organic whole wheat bread milk almond oat greek yogurt honey banana strawberry blueberry apple "
    "avocado spinach kale broccoli carrot tomato potato onion garlic ginger lemon lime orange chicken "
    "breast beef turkey bacon sausage salmon tuna shrimp egg butter cheese cheddar mozzarella parmesan "
    "cream sour pasta penne spaghetti rice quinoa granola cereal oatmeal coffee tea green black juice "
    "sparkling water soda chips salsa hummus cracker cookie chocolate dark vanilla ice frozen pizza "
    "burrito soup broth bean lentil chickpea tofu tempeh olive oil vinegar mustard ketchup mayo sauce "
    "pepper salt sugar flour baking powder cinnamon peanut almond cashew walnut raisin cranberry bar "
    "protein gluten free low fat unsweetened original classic family size pack mini large small fresh"
).split()
_AISLES = [f"aisle {i} {_WORDS[(i * 7) % len(_WORDS)]} {_WORDS[(i * 13 + 5) % len(_WORDS)]}" for i in range(134)]
_DEPTS = [f"dept {_WORDS[(i * 11 + 3) % len(_WORDS)]}" for i in range(21)]


def synthetic_catalog(n: int = 49688, seed: int = 42) -> dict[str, str]:
    """product_id -> "Product: {name}. Aisle: {aisle}. Department: {department}."
    (prepare_instacart_sbert.py:185-193); ids are "1".."n" like Instacart's."""
    h = hash_u64(seed, 1, n * 9).reshape(n, 9)
    out: dict[str, str] = {}
    for i in range(n):
        nw = 2 + int(h[i, 0] % np.uint64(6))
        name = " ".join(_WORDS[int(h[i, 1 + j] % np.uint64(len(_WORDS)))].capitalize() for j in range(nw))
        aisle = _AISLES[int(h[i, 7] % np.uint64(len(_AISLES)))]
        dept = _DEPTS[int(h[i, 8] % np.uint64(len(_DEPTS)))]
        out[str(i + 1)] = f"Product: {name}. Aisle: {aisle}. Department: {dept}."
    return out


def synthetic_user_contexts(n: int, seed: int = 1234, max_items: int = 20, max_orders: int = 5,
                            min_orders: int = 1, per_order: int = 6) -> list[str]:
    """"[+{d}d w{dow}h{hh}] n1, n2; [+..] ..." with 1-5 orders and <= 20 names
    (prepare_instacart_sbert.py:233-258; configs/data_prep.yaml:10-11) — about 65 WordPiece tokens each.
    bench.py's text leg passes max_items=40, min_orders=5, max_orders=8, per_order=8: heavier users
    whose contexts average ~128 tokens, the same workload size as its token-id batches."""
    W = 64 if max_items <= 20 else 160
    h = hash_u64(seed, 2, n * W).reshape(n, W)
    out = []
    for i in range(n):
        n_orders = min_orders + int(h[i, 0] % np.uint64(max_orders - min_orders + 1))
        left = max_items
        segs = []
        c = 1
        for o in range(n_orders):
            if left <= 0:
                break
            d, dow, hh = int(h[i, c] % np.uint64(31)), int(h[i, c + 1] % np.uint64(7)), int(h[i, c + 2] % np.uint64(24))
            cnt = min(left, 1 + int(h[i, c + 3] % np.uint64(per_order)))
            c += 4
            names = []
            for _ in range(cnt):
                a, b = int(h[i, c % W] % np.uint64(len(_WORDS))), int(h[i, (c + 1) % W] % np.uint64(len(_WORDS)))
                names.append(f"{_WORDS[a].capitalize()} {_WORDS[b].capitalize()}")
                c += 2
            left -= cnt
            segs.append(f"[+{d}d w{dow}h{hh}] " + ", ".join(names))
        out.append("; ".join(segs) + ".")
    return out


def synthetic_vocab() -> list[str]:
    """A WordPiece vocab (BERT layout: [PAD]=0, [UNK]=100, [CLS]=101, [SEP]=102,
    [MASK]=103) covering the synthetic strings; stands in for the absent
    all-MiniLM-L6-v2 vocab.txt."""
    v = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    v += [f"[unused{i}]" for i in range(99, 995)]  # keep real-BERT offset ~999 for first real token
    chars = list("!\"#$%&'()*+,-./0123456789:;<=>?@[\\]^_`abcdefghijklmnopqrstuvwxyz{|}~")
    v += chars
    v += ["##" + ch for ch in "abcdefghijklmnopqrstuvwxyz0123456789"]
    seen = set(v)
    extra = sorted(set(_WORDS) | {"product", "aisle", "department", "dept", "d", "w", "h"})
    for wd in extra:
        if wd not in seen:
            v.append(wd); seen.add(wd)
    for wd in extra:  # some continuation pieces so multi-piece words occur
        if len(wd) > 4:
            pc = "##" + wd[2:]
            if pc not in seen:
                v.append(pc); seen.add(pc)
            hd = wd[:2]
            if hd not in seen:
                v.append(hd); seen.add(hd)
    return v
