import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Parity gates are part of the product contract: nothing in the environment may loosen them (rounds 2-3 had a
    # measuring knob of this name; a session that still sets it is refused instead of silently ignoring it).
    if os.environ.get("AGDIFF_PARITY_GATE_SCALE"):
        raise pytest.UsageError("AGDIFF_PARITY_GATE_SCALE is set: the parity gates of tests/helpers.py are not "
                                "adjustable from the environment; unset it")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """Parity figures measured by this session (tests/helpers.py:check_close) -> gpurun_out/parity_errors.json."""
    import json
    try:
        import helpers
    except Exception:
        return
    if not helpers._RECORDS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_errors.json"), "w") as f:
        json.dump(helpers._RECORDS, f, indent=1)
