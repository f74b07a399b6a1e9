"""The worker-side allocator oracle against the reference's own allocator tests (26 cases, test_allocator.rs), plus the
second opinion of scipy's HiGHS on the objective of every group model those cases build."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import alloc_oracle as api  # noqa: E402
from tests import alloc_cases  # noqa: E402


@pytest.mark.parametrize("case", alloc_cases.CASES, ids=lambda c: c.__name__)
def test_reference_case(case):
    case(api)


def test_group_models_agree_with_highs(monkeypatch):
    """Every group model the reference cases solve: the enumerated optimum has HiGHS's objective value."""
    seen = []
    real = api.solve_group_model

    def spy(m):
        out = real(m)
        seen.append((m, out))
        return out

    monkeypatch.setattr(api, "solve_group_model", spy)
    for case in alloc_cases.CASES:
        case(api)
    assert len(seen) > 40
    for m, out in seen:
        h = api.highs_objective(m)
        if out is None:
            assert h is None
        else:
            assert h is not None and abs(h - out[1]) <= 1e-6 * max(1.0, abs(h)), (h, out[1])


def test_hbmap_matches_insertion_only_growth():
    """HbMap on the pinned worker-id probe of SURVEY.md §7.3-2 (50, 51 -> buckets 2, 3) and survives churn."""
    m = api.HbMap()
    m.insert(50, 1)
    m.insert(51, 2)
    assert [k for k, _ in m.items_in_order()] == [50, 51]
    import random
    rnd = random.Random(1)
    m, ref = api.HbMap(), {}
    for step in range(4000):
        k = rnd.randrange(64)
        if k in ref and rnd.random() < 0.5:
            m.remove(k)
            del ref[k]
        else:
            m.insert(k, step)
            ref[k] = step
        assert len(m) == len(ref)
    assert dict(m.items_in_order()) == ref
