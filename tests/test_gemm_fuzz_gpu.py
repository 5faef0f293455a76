"""GPU: the conv GEMM on random shapes / taps / epilogues -- every tile and a forced split-K against a float64 convolution of the same
data, and the operand image a launch writes against the split of its own fp32 output (scripts/exp/gemm_fuzz.py; 900 cases were run
when the 32-row tile went in, a sample of them runs here)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_conv_gemms(cuda):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "exp", "gemm_fuzz.py"), "40", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
