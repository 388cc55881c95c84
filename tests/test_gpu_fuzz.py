"""Randomised parity sweep (scripts/fuzz_parity.py): random shapes / modes / samplers / optimizers, three steps each on the engine
and on the CPU oracle.  A short fixed-seed run here; longer runs by hand (four seeds x 60-80 cases agreed at the end of round 3)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_match_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "BPRX_ITEM_MODE"}
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_parity.py"), "30", "11"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0 and "30 cases, 0 failed" in r.stdout, tail
