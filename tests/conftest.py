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


_PREFETCH = {"wanted": False, "started": False}


def pytest_collection_finish(session):
    """GPU tier: note whether full-depth GPU tests were collected (tests/_cases.py draws their synthetic state dicts in the background)."""
    names = {item.fspath.basename for item in session.items if item.get_closest_marker("gpu")}
    _PREFETCH["wanted"] = bool(names & {"test_fp16_gpu.py", "test_fp8_gpu.py", "test_e2e_gpu.py", "test_zz_c2_parity_gpu.py"})


def pytest_runtest_setup(item):
    """Start the background draws with the first GPU test that is NOT one of the live bench invocations (tests/test_bench_live_gpu.py sorts
    first and runs `bench.py` in subprocesses that draw their own weights on all host cores: with the draws beside it that test took 222 s
    instead of 37); from there on the CPU draws (~10 s each, 14 distinct ones) run UNDER the GPU-side tests instead of between them."""
    if _PREFETCH["wanted"] and not _PREFETCH["started"] and item.get_closest_marker("gpu") and item.fspath.basename != "test_bench_live_gpu.py":
        _PREFETCH["started"] = True
        import _cases
        _cases.prefetch(GOLDEN)


def pytest_sessionfinish(session, exitstatus):
    if _PREFETCH["started"]:
        import _cases
        _cases.stop_prefetch()
