"""What a parity test does when a solver ran into its time limit ("What's weak" 2: such ticks used to be SKIPPED).

The reference returns `Done` for every tick HiGHS certifies within `mip_rel_gap = 1e-4` inside the scheduler's time limit (solver/highs.rs:65-88) and
`NeedMoreCompute` otherwise.  A tick the product does not certify where plain HiGHS — the reference's options, the same limit — does is a RESULT
difference: the test fails, unless the seed is on the test's explicit allow-list (`allow`), which is short, counted, and asserted never to grow.  Whatever
came out is still checked: a NeedMoreCompute answer must be a feasible point of the reference's model, and everything downstream of the counts (decode,
mapping, prefill: tier T3) must equal the oracle's on the same counts.
"""
from __future__ import annotations

import numpy as np

from hyperqueue_amd import abi

SIDES = {"product": 0, "oracle": 0, "both": 0}  # how often which side hit its limit in this session (printed by the conftest's summary)


def model_point(model, counts):
    """the counts as a point of the oracle's model; flag columns (zero-cost 0/1) switched on where their `>=` row needs them (they are existential)"""
    cd = {(q, v, w): c for (q, v, w, c) in counts}
    n = len(model["obj"])
    x = np.zeros(n)
    for j in range(n):
        if model["ctype"][j] == 0:
            x[j] = cd.get((int(model["crq"][j]), int(model["cvariant"][j]), int(model["cworker"][j])), 0)
    roff, rcol, rcoef = model["roff"], model["rcol"], model["rcoef"]
    for i in range(len(model["rhs"])):
        if model["rtype"][i] != 0:
            continue
        a, b = roff[i], roff[i + 1]
        act = float(np.dot(rcoef[a:b], x[rcol[a:b]]))
        if act < model["rhs"][i] - 1e-6:
            for k in range(a, b):
                j = rcol[k]
                if model["kind"][j] == 1 and model["obj"][j] == 0.0 and rcoef[k] >= model["rhs"][i] - 1e-9:
                    x[j] = 1.0
                    break
    return x


def rows_hold(model, x) -> bool:
    from scipy.sparse import csr_matrix

    A = csr_matrix((model["rcoef"], model["rcol"], model["roff"]), shape=(len(model["rhs"]), len(x)))
    act = A @ x
    rt, rhs = model["rtype"], model["rhs"]
    return bool(np.all(act[rt == 1] <= rhs[rt == 1] + 1e-6) and np.all(act[rt == 0] >= rhs[rt == 0] - 1e-6) and np.all(np.abs(act[rt == 2] - rhs[rt == 2]) <= 1e-6))


def check_given_counts(snap, got, time_limit_s: float = 5.0):
    """tier T3 on whatever the product answered: feasible for every row of the reference's model, and decode + mapping + prefill on these counts
    (Oracle.tick_given, scheduler/mapping.rs:23-234) give the product's records, retracts, redirects and free vectors.  Returns (model, x)."""
    from oracle.oracle import Oracle

    o = Oracle(abi.make_config(time_limit_s=time_limit_s))
    want = o.tick_given(snap, got.counts, is_optimal=bool(got.is_optimal))
    model = o.last_model()
    x = model_point(model, got.counts)
    assert rows_hold(model, x), "the product's counts violate a row of the reference's model"
    assert got.batches == want.batches and got.counts == want.counts
    assert got.records == want.records and got.retracts == want.retracts and sorted(got.redirects) == sorted(want.redirects)
    assert (got.new_free == want.new_free).all()
    return model, x


def plain_highs(snap, cfg_time_limit_s: float):
    """the same snapshot through HiGHS as the reference configures it (time_limit only, default mip_rel_gap): (result, model)"""
    from oracle.oracle import Oracle

    o = Oracle(abi.make_config(time_limit_s=cfg_time_limit_s), reference_solver_options=True)
    r = o.tick(snap)
    return r, o.last_model()


def uncertified(snap, got, seed, allow, time_limit_s: float = 5.0, what: str = ""):
    """The product came back without a certificate (got.is_optimal false).  Checks what it did answer, asks plain HiGHS under the same limit, and FAILS when
    HiGHS certifies what the product did not — unless `seed` is on `allow`."""
    assert got.status in (abi.HQTICK_NEED_MORE_COMPUTE, abi.HQTICK_DONE), got.status
    assert not got.is_optimal
    model, x = check_given_counts(snap, got, time_limit_s)
    ref, _ = plain_highs(snap, time_limit_s)
    if ref.is_optimal:
        SIDES["product"] += 1
        z_got = float(np.dot(model["obj"], x))
        assert seed in allow, (f"{what} seed {seed}: the product returns NeedMoreCompute (objective {z_got:.9f}) on a tick that HiGHS with the reference's options certifies "
                               f"within the same {time_limit_s} s limit; not on the allow-list {sorted(allow)}")
    else:
        SIDES["both"] += 1
