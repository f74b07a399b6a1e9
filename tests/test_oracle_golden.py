"""Pins the CPU oracle against the reference's own unit-test vectors.  CPU only."""
import pytest

import golden_cases
from hyperqueue_amd import abi
from oracle.oracle import Oracle


class OracleBackend:
    """Adapter: SchedEnv carries its SchedulerConfig; the oracle ctx is created per config."""

    def __init__(self):
        self._ctx = {}

    def _o(self, cfg: abi.Config) -> Oracle:
        key = (cfg.proactive_filling_reserve, cfg.proactive_filling_max)
        if key not in self._ctx:
            self._ctx[key] = Oracle(cfg)
        return self._ctx[key]

    cfg = abi.make_config()

    def tick(self, snap):
        return self._o(getattr(snap, "config", None) or self.cfg).tick(snap)

    def batches(self, snap):
        return self._o(getattr(snap, "config", None) or self.cfg).batches(snap)

    def query(self, snap, *a):
        return self._o(getattr(snap, "config", None) or self.cfg).query(snap, *a)


@pytest.fixture(scope="module")
def backend():
    return OracleBackend()


@pytest.mark.parametrize("case", golden_cases.ALL_CASES, ids=lambda f: f.__name__)
def test_golden(case, backend):
    if case.__name__ == "test_many_cuts":
        pytest.skip("tolerance test, run in test_oracle_slow")
    case(backend)


class CanonicalOracleBackend(OracleBackend):
    """The oracle with the MI355X path's tie-break applied: must still satisfy every reference expectation."""

    def _o(self, cfg):
        key = (cfg.proactive_filling_reserve, cfg.proactive_filling_max)
        if key not in self._ctx:
            self._ctx[key] = Oracle(cfg, canonical=True)
        return self._ctx[key]


@pytest.mark.parametrize("case", golden_cases.ALL_CASES, ids=lambda f: f.__name__)
def test_golden_canonical(case):
    if case.__name__ == "test_many_cuts":
        pytest.skip("tolerance test; canonicalising a 600-column model with one HiGHS call per column is slow")
    if case.__name__ == "test_schedule_many_distinct_shapes_stays_bounded":
        pytest.skip("time-bound test; 1200 HiGHS calls to canonicalise would break the bound by construction")
    case(CanonicalOracleBackend())


@pytest.mark.slow
def test_many_cuts_oracle():
    golden_cases.test_many_cuts(OracleBackend())


@pytest.mark.parametrize("case", golden_cases.E2E_EXTRA_CASES, ids=lambda f: f.__name__)
@pytest.mark.parametrize("canonical", [False, True])
def test_e2e_extra(case, canonical):
    """five more scheduling outcomes of the reference's pytest suite (tests/test_job.py, tests/test_job_mn.py)"""
    case(CanonicalOracleBackend() if canonical else OracleBackend())
