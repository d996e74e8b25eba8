"""pytest configuration: marker registration + shared fixtures."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests"))
sys.dont_write_bytecode = True

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-depth CPU oracle runs")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_finish(session):
    """GPU tier: start drawing the full-depth synthetic state dicts in the background (tests/_cases.py) as soon as the collection shows
    that full-depth GPU tests will run -- the CPU draws (~10 s each, 14 distinct ones) then overlap the GPU-side tests."""
    names = {item.fspath.basename for item in session.items if item.get_closest_marker("gpu")}
    if names & {"test_fp16_gpu.py", "test_fp8_gpu.py", "test_e2e_gpu.py", "test_zz_c2_parity_gpu.py"}:
        import _cases
        _cases.prefetch(GOLDEN)
