"""Short randomised parity sweeps (tools/fuzz_search.py, tools/fuzz_encoder.py) — every run draws the same
seeded cases; the tools take a case count and a seed for longer soaks."""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("tool,cases,seed", [("fuzz_search.py", "80", "7"), ("fuzz_encoder.py", "12", "7")])
def test_randomised_parity(tool, cases, seed):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / tool), cases, seed], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 mismatches" in r.stdout
