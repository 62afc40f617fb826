"""a short run of tools/fuzz_parity.py in the suite: random sizes / windows / levels / flags with adversarial sub-pixel
fractions (LK), random homographies over all pixel types incl. tie-heavy translations (warp), random thresholds / cells (FAST);
every comparison is bit-exact against the oracle.  (This sweep, pointed at the 1080p workload, is what found the negative
fourth-weight bug.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_parity_sweep():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "30", "7"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "30 cases, 0 mismatches" in r.stdout
