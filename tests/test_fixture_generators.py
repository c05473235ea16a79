"""CPU, this container only: the committed data fixtures regenerate from /root/reference with the committed generator
(skipped on the GPU box, where the reference does not exist)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/config_train"), reason="the reference tree is only present in the build container")
def test_ref_config_archs_fixture_matches_its_generator():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_ref_config_archs.py"), "--check"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
