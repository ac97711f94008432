import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes-long GPU cases kept out of the default `-m gpu` run (the driver's step limit is 1200 s; "
                                       "VERDICT r05 weak 9); DESIRE_SLOW_TESTS=1 runs them -- profiles/collect_r06.sh does")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("DESIRE_SLOW_TESTS") == "1":
        return
    skip = pytest.mark.skip(reason="slow case: set DESIRE_SLOW_TESTS=1 (profiles/collect_r06.sh runs it)")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
