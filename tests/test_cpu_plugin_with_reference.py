"""Drop-in check against the REAL reference host code (only where /root/reference is mounted --
the build container; skipped on the GPU box).  Runs in a subprocess because it stubs `zmq` /
`flashinfer` import-time dependencies of `minisgl`."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(not Path("/root/reference/python/minisgl").exists(), reason="reference not mounted")
def test_plugin_drives_reference_radix_cache_to_the_golden_trace():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, str(ROOT / "tests/golden/make_golden.py"), "--check-plugin"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "plugin check ok" in r.stdout, r.stdout + r.stderr
