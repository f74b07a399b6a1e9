import os
import sys

import pytest

# torch bundles its own HIP runtime: import it BEFORE libhqtick.so pulls in /opt/rocm's, so the process ends up with one runtime
# (the sharded path hands torch tensors to the library; bench.py imports torch first for the same reason).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def oracle_backend():
    from oracle.oracle import Oracle

    class _Backend:
        """tick()/batches() through the CPU oracle; a fresh oracle ctx per SchedulerConfig."""

        def __init__(self):
            self._cache = {}

        def _get(self, snap_cfg):
            key = (snap_cfg.proactive_filling_reserve, snap_cfg.proactive_filling_max)
            if key not in self._cache:
                self._cache[key] = Oracle(snap_cfg)
            return self._cache[key]

    return _Backend()


def pytest_terminal_summary(terminalreporter):
    """which side ran into a solver limit, and how often (tests/limits.py): printed, so that such ticks are counted instead of vanishing as skips"""
    try:
        from limits import SIDES
    except Exception:
        return
    if any(SIDES.values()):
        terminalreporter.write_line(f"solver limits hit in this session: product only {SIDES['product']}, canonical oracle only {SIDES['oracle']}, both {SIDES['both']}")
