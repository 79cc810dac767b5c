"""The C-ABI library loads on a CPU-only box and exports every symbol include/icrec.h declares
(no compute calls here: there is no GPU)."""
from __future__ import annotations

import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def native():
    from instacart_next_order_recommendation_amd import _native

    if not _native.LIB_PATH.exists():
        _native.build()
    return _native


def test_header_symbols_match_binding(native):
    header = (ROOT / "include" / "icrec.h").read_text()
    declared = set(re.findall(r"ICREC_API\s+[\w\s\*]+?\b(icrec_\w+)\s*\(", header))
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)


def test_library_exports_every_symbol(native):
    lib = native.lib()
    for name in native.EXPORTS:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.icrec_version()


def test_struct_layout_matches_oracle(native):
    """icrec_bert_cfg is shared verbatim with the oracle's struct."""
    import ctypes as C

    from oracle import oracle

    assert C.sizeof(native.BertCfg) == C.sizeof(oracle.BertCfg) == 40
    assert [f[0] for f in native.BertCfg._fields_] == [f[0] for f in oracle.BertCfg._fields_]


def test_weight_count_agrees(native):
    import ctypes as C

    from instacart_next_order_recommendation_amd.synthetic import BertShape
    from oracle import oracle

    s = BertShape()
    cfg = native.BertCfg(s.vocab_size, s.hidden, s.layers, s.heads, s.intermediate, s.max_position, s.type_vocab,
                         s.ln_eps, s.n_normalize, 0)
    assert native.lib().icrec_encoder_weight_count(C.byref(cfg)) == s.weight_count() == oracle.weight_count(oracle.make_cfg())
    assert s.weight_count() == 22_713_216 - (384 * 384 + 384)  # BertModel's 22.7M minus the unused pooler


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = ROOT / "instacart_next_order_recommendation_amd"
    for py in pkg.rglob("*.py"):
        src = py.read_text()
        assert "from oracle" not in src and "import oracle" not in src, py
    for c in (pkg / "csrc").glob("*"):
        if c.is_file():  # comments may cite the oracle; nothing may include or link it
            for line in c.read_text().splitlines():
                code = line.split("//")[0]
                assert "oracle" not in code, (c, line)


def test_no_gpu_means_loud_failure(native):
    """Without a HIP device the product classes raise instead of falling back."""
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from instacart_next_order_recommendation_amd.search import DeviceIndex

    with pytest.raises(Exception):
        DeviceIndex(np.ones((4, 384), np.float32))
    with pytest.raises(native.IcrecError):
        DeviceIndex(np.ones((4, 384), np.float32), device="cpu")


def test_comm_entry_points_validate_arguments(native):
    """The exchange entry points reject bad arguments without touching a GPU."""
    import ctypes as C

    lib = native.lib()
    h = C.c_void_p()
    assert lib.icrec_comm_init(None, 3, 2, 0, C.byref(h)) == -1  # rank >= world
    assert b"rank" in lib.icrec_last_error()
    assert lib.icrec_comm_init(None, 0, 2, 0, C.byref(h)) == -1  # world > 1 needs the rendezvous id
    assert lib.icrec_comm_world(None) == 0 and lib.icrec_comm_rank(None) == -1
    assert lib.icrec_search_sharded_workspace_bytes(None, None, 4, 20) == 0
    assert lib.icrec_index_dim(None) == 0 and lib.icrec_index_device(None) == -1


def test_lib_shares_torchs_hip_runtime():
    """libicrec.so must bind the HIP runtime torch ships (one runtime per process): _native.lib() imports torch before
    it maps the library, so that a process which builds / loads the library first and touches torch later (the
    driver's build() then smoke() in one interpreter) does not end up with /opt/rocm's runtime beside torch's — the
    second one then reports "no ROCm-capable device is detected"."""
    import subprocess
    import sys

    code = ("import sys; from instacart_next_order_recommendation_amd import _native; "
            "assert 'torch' not in sys.modules; _native.lib(); assert 'torch' in sys.modules; "
            "maps = open('/proc/self/maps').read(); "
            "hip = sorted({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}); "
            "assert len(hip) == 1 and '/torch/lib/' in hip[0], hip; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
