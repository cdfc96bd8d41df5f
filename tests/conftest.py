import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a dead-locked kernel or host thread must end the run, not hold the GPU box until the outer limit: every test gets
    # 10 minutes (the slowest takes 5 s) when pytest-timeout is available.  "thread" method: a hang inside a native call
    # never returns to the interpreter, so a signal handler would not run.
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 600
        config.option.timeout_method = "thread"


@pytest.fixture(scope="session")
def pins():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_pins.json")) as f:
        return json.load(f)
