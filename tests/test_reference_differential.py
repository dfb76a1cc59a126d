"""Live differential test against the reference itself — runs only where /root/reference exists (the build container; the GPU box and
any other checkout skip it).  The committed fixtures pin fixed cases; this draws random ones around them on every run:
60 NMS configurations (bit-equal), 10 DMFF blocks, 7 whole models of random family / rectangular shape / iteration count, 24 ap_per_class /
box-helper cases and 60 images through the inline TP-matching block."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_equals_the_live_reference_on_random_cases(seed):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "reference_differential.py"), str(seed)], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0 and "DIFFERENTIAL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
