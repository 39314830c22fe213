"""Differential fuzz of the Hermitian fast paths (four-product exponential with hand-over, one-wave-per-batch derivatives)
against their predecessors and the Pade route on random small shapes -- tools/fuzz_paths.py with a fixed seed."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_paths_agree_with_their_predecessors_on_random_shapes():
    env = {k: v for k, v in os.environ.items() if not k.startswith("GRAPE_")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_paths.py"), "120", "31"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "120 cases agree" in out.stdout
