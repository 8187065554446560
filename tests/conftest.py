import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU (the long reference-vs-oracle replay); skipped unless RG_RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("RG_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow: set RG_RUN_SLOW=1 (last run recorded in profiles/r04_slow_replay.txt)")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)
