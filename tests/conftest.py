import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """What the session's parity tests compared (tests/parity_stats.py): bit-identical forward counts and the logit error
    distributions of ours / the GPU-library floor / the CPU fp32-accumulate floor -- readable from the pytest output alone."""
    try:
        import parity_stats
    except Exception:
        return
    lines = parity_stats.summary_lines()
    if lines:
        terminalreporter.write_sep("-", "parity summary")
        for ln in lines:
            terminalreporter.write_line(ln)
        out = ROOT / "gpurun_out"
        if out.is_dir() or os.environ.get("MSGL_PARITY_SUMMARY"):
            parity_stats.write_summary(Path(os.environ.get("MSGL_PARITY_SUMMARY", out / "parity_summary.json")))
