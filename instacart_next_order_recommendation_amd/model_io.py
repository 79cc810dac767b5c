"""Read (and, for tests/benchmarks, write) a SentenceTransformer model directory.

The reference loads `SentenceTransformer(str(model_dir))`
(src/inference/serve_recommendations.py:166-170) where model_dir is what
`SentenceTransformer.save` wrote after fine-tuning (src/training/train_sbert.py:139-141):
  modules.json, config.json, model.safetensors, tokenizer files,
  sentence_bert_config.json, 1_Pooling/config.json, 2_Normalize/.
Only local directories are supported (no hub download: there is no network).

Tokenisation is a host stage: the WordPiece tokenizer is built from the directory's
tokenizer.json / vocab.txt with the `tokenizers` package (the same Rust tokenizer the
reference ends up using through transformers, uv.lock:3841).
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path
from typing import Sequence

import numpy as np

from .synthetic import BertShape, blob_to_state_dict, state_dict_to_blob, synthetic_bert_weights, synthetic_vocab

DEFAULT_MAX_SEQ_LENGTH = 256  # configs/train.yaml:11


@dataclass
class LoadedModel:
    shape: BertShape
    weights: np.ndarray  # flat fp32 blob, include/icrec.h order
    max_seq_length: int
    tokenizer: "HostTokenizer"


class NativeTokenizer:
    """libicrec's C++ WordPiece tokenizer (csrc/tokenizer.cpp): batched, multi-threaded, GIL-free."""

    def __init__(self, vocab_path: Path, lowercase: bool, max_seq_length: int, n_threads: int = 0):
        import ctypes as C

        from . import _native

        self._C, self._native = C, _native
        h = C.c_void_p()
        _native.check(_native.lib().icrec_tokenizer_create(str(vocab_path).encode(), 1 if lowercase else 0,
                                                           int(max_seq_length), C.byref(h)), "icrec_tokenizer_create")
        self._h, self.max_seq_length, self.n_threads = h, max_seq_length, n_threads

    def packed(self, texts: Sequence[str]) -> tuple[np.ndarray, np.ndarray]:
        """texts -> (ids int32[T], cu_seqlens int32[n+1]): the packed form icrec_encode takes, straight from
        icrec_tokenize (no per-sequence Python lists: at 1,024 contexts per batch the list round trip costs
        more host time - under the GIL - than the tokenisation itself)."""
        C = self._C
        n = len(texts)
        if n == 0:
            return np.zeros(0, np.int32), np.zeros(1, np.int32)
        arr = (C.c_char_p * n)(*[t.replace("\x00", "").encode("utf-8", "replace") for t in texts])
        cu = np.empty(n + 1, np.int32)
        cap = n * self.max_seq_length
        ids = np.empty(cap, np.int32)
        self._native.check(self._native.lib().icrec_tokenize(self._h, arr, n, ids.ctypes.data_as(C.c_void_p), cap,
                                                             cu.ctypes.data_as(C.c_void_p), self.n_threads),
                           "icrec_tokenize")
        return ids[: int(cu[n])], cu

    def __call__(self, texts: Sequence[str]) -> list[list[int]]:
        ids, cu = self.packed(texts)
        return [ids[cu[i]:cu[i + 1]].tolist() for i in range(len(texts))]

    def __del__(self):  # pragma: no cover
        try:
            self._native.lib().icrec_tokenizer_destroy(self._h)
        except Exception:
            pass


class HostTokenizer:
    """BERT WordPiece tokenisation on the host ([CLS] ... [SEP], truncation to max_seq_length).

    backend "native" (default when the directory has vocab.txt): libicrec's C++ tokenizer, ~6x the
    throughput of the Rust one on 8 cores (68k vs 11k user contexts/s) — at >50k QPS per GPU the
    tokenizer would otherwise be the bottleneck; agrees with the Rust tokenizer on tests/test_tokenizer.py.
    backend "tokenizers" (ICREC_TOKENIZER=tokenizers, or no vocab.txt): the Rust library the reference
    itself ends up in."""

    def __init__(self, model_dir: Path, max_seq_length: int, backend: str | None = None):
        import os

        model_dir = Path(model_dir)
        tj, vt = model_dir / "tokenizer.json", model_dir / "vocab.txt"
        backend = backend or os.getenv("ICREC_TOKENIZER") or ("native" if vt.exists() else "tokenizers")
        lower = True
        tc = model_dir / "tokenizer_config.json"
        if tc.exists():
            lower = bool(json.loads(tc.read_text()).get("do_lower_case", True))
        self.max_seq_length = max_seq_length
        self.backend = backend
        if backend == "native":
            if not vt.exists():
                raise FileNotFoundError(f"{vt} missing: the native tokenizer needs vocab.txt")
            self._native_tok = NativeTokenizer(vt, lower, max_seq_length)
            return
        if backend != "tokenizers":
            raise ValueError(f"unknown tokenizer backend {backend!r}")
        from tokenizers import Tokenizer
        from tokenizers.implementations import BertWordPieceTokenizer

        if tj.exists():
            self._tok = Tokenizer.from_file(str(tj))
        elif vt.exists():
            self._tok = BertWordPieceTokenizer(str(vt), lowercase=lower)._tokenizer
        else:
            raise FileNotFoundError(f"{model_dir} has neither tokenizer.json nor vocab.txt")
        self._tok.no_padding()
        self._tok.enable_truncation(max_length=max_seq_length)

    def __call__(self, texts: Sequence[str]) -> list[list[int]]:
        if self.backend == "native":
            return self._native_tok(texts)
        return [e.ids for e in self._tok.encode_batch(list(texts))]

    def packed(self, texts: Sequence[str]) -> tuple[np.ndarray, np.ndarray]:
        """(ids int32[T], cu_seqlens int32[n+1]) - see NativeTokenizer.packed."""
        if self.backend == "native":
            return self._native_tok.packed(texts)
        seqs = self(texts)
        cu = np.zeros(len(seqs) + 1, np.int32)
        np.cumsum([len(s) for s in seqs], out=cu[1:])
        ids = np.concatenate([np.asarray(s, np.int32) for s in seqs]) if seqs else np.zeros(0, np.int32)
        return ids, cu


def load_model_dir(model_dir: Path | str) -> LoadedModel:
    """Parse config + weights + tokenizer of a local SentenceTransformer directory."""
    from safetensors.numpy import load_file

    d = Path(model_dir)
    if not d.is_dir():
        raise FileNotFoundError(
            f"model_dir {model_dir!r} is not a local directory (hub ids cannot be fetched: no network)")
    cfg = json.loads((d / "config.json").read_text())
    if cfg.get("hidden_act", "gelu") != "gelu":
        raise ValueError(f"unsupported hidden_act {cfg.get('hidden_act')!r} (kernels implement erf-GELU)")
    n_norm = 1  # encode(..., normalize_embeddings=True) at every reference call site
    mj = d / "modules.json"
    if mj.exists():
        for m in json.loads(mj.read_text()):
            t = m.get("type", "")
            if t.endswith("Normalize"):
                n_norm += 1
            if t.endswith("Pooling"):
                pc = d / m.get("path", "1_Pooling") / "config.json"
                if pc.exists():
                    p = json.loads(pc.read_text())
                    others = [k for k, v in p.items() if k.startswith("pooling_mode_") and v and k != "pooling_mode_mean_tokens"]
                    if not p.get("pooling_mode_mean_tokens", True) or others:
                        raise ValueError(f"only mean-token pooling is implemented (got {p})")
    shape = BertShape(vocab_size=int(cfg["vocab_size"]), hidden=int(cfg["hidden_size"]),
                      layers=int(cfg["num_hidden_layers"]), heads=int(cfg["num_attention_heads"]),
                      intermediate=int(cfg["intermediate_size"]), max_position=int(cfg["max_position_embeddings"]),
                      type_vocab=int(cfg.get("type_vocab_size", 2)), ln_eps=float(cfg.get("layer_norm_eps", 1e-12)),
                      n_normalize=n_norm)
    st = d / "model.safetensors"
    if not st.exists():
        raise FileNotFoundError(f"{st} missing (pytorch_model.bin pickles are not loaded: only safetensors)")
    weights = state_dict_to_blob(load_file(str(st)), shape)
    max_len = DEFAULT_MAX_SEQ_LENGTH
    sb = d / "sentence_bert_config.json"
    if sb.exists():
        max_len = int(json.loads(sb.read_text()).get("max_seq_length") or max_len)
    max_len = min(max_len, DEFAULT_MAX_SEQ_LENGTH, shape.max_position)
    return LoadedModel(shape, weights, max_len, HostTokenizer(d, max_len))


def write_synthetic_model_dir(path: Path | str, seed: int = 0, shape: BertShape | None = None) -> Path:
    """Write a SentenceTransformer-layout directory with seeded random weights and the synthetic
    WordPiece vocab (stand-in for the fine-tuned all-MiniLM-L6-v2 that cannot be downloaded here)."""
    from safetensors.numpy import save_file

    d = Path(path)
    d.mkdir(parents=True, exist_ok=True)
    vocab = synthetic_vocab()
    if shape is None:
        shape = BertShape(vocab_size=len(vocab))
    if shape.vocab_size < len(vocab):
        raise ValueError("shape.vocab_size smaller than the synthetic vocab")
    blob = synthetic_bert_weights(shape, seed=seed)
    save_file({k: np.ascontiguousarray(v) for k, v in blob_to_state_dict(blob, shape).items()},
              str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps({
        "architectures": ["BertModel"], "model_type": "bert", "vocab_size": shape.vocab_size,
        "hidden_size": shape.hidden, "num_hidden_layers": shape.layers, "num_attention_heads": shape.heads,
        "intermediate_size": shape.intermediate, "hidden_act": "gelu",
        "max_position_embeddings": shape.max_position, "type_vocab_size": shape.type_vocab,
        "layer_norm_eps": shape.ln_eps}, indent=2))
    (d / "vocab.txt").write_text("\n".join(vocab) + "\n")
    (d / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": True, "tokenizer_class": "BertTokenizer"}))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 256, "do_lower_case": False}))
    (d / "modules.json").write_text(json.dumps([
        {"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
        {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
        {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}], indent=2))
    (d / "1_Pooling").mkdir(exist_ok=True)
    (d / "1_Pooling" / "config.json").write_text(json.dumps({
        "word_embedding_dimension": shape.hidden, "pooling_mode_cls_token": False,
        "pooling_mode_mean_tokens": True, "pooling_mode_max_tokens": False}))
    (d / "2_Normalize").mkdir(exist_ok=True)
    return d
