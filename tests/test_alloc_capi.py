"""libhqalloc.so (include/hqalloc.h) through its C ABI: the reference's 26 allocator tests, then randomised allocate / release
sequences compared step by step with the oracle (same indices, same groups, same fractions, same concise state)."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from hyperqueue_amd import allocator as api  # noqa: E402
from oracle import alloc_oracle as ora  # noqa: E402
from tests import alloc_cases  # noqa: E402


@pytest.mark.parametrize("case", alloc_cases.CASES, ids=lambda c: c.__name__)
def test_reference_case(case):
    case(api)


def test_exports_every_declared_symbol():
    import re

    lib = api.load()
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "hqalloc.h")).read()
    declared = set(re.findall(r"\b(hqalloc_[a-z_]+)\s*\(", header))
    assert declared == set(api.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hqalloc_abi_version() == api.HQALLOC_ABI_VERSION


def random_descriptor(rnd):
    pools, group_pools = [], []
    for r in range(rnd.randint(1, 5)):
        t = rnd.random()
        if t < 0.45 and len(group_pools) < 3:
            n, size = rnd.randint(2, 4), rnd.randint(1, 4)
            if rnd.random() < 0.3:  # ragged groups, shuffled labels
                flat = list(range(n * size))
                rnd.shuffle(flat)
                groups, pos = [], 0
                for g in range(n):
                    k = rnd.randint(1, size)
                    groups.append(flat[pos:pos + k])
                    pos += k
                pools.append((api.GROUPS_POOL, groups, 0))
            else:
                pools.append((api.GROUPS_POOL, [list(range(s * size, (s + 1) * size)) for s in range(n)], 0))
            group_pools.append(r)
        elif t < 0.75:
            start = rnd.choice([0, 0, 5])
            pools.append((api.INDICES_POOL, [list(range(start, start + rnd.randint(1, 8)))], 0))
        elif t < 0.95:
            pools.append((api.SUM_POOL, [], rnd.choice([300, 20_000, 1_000_000, 25_000])))
        else:
            pools.append((api.EMPTY_POOL, [], 0))
    coupling = []
    if len(group_pools) >= 2 and rnd.random() < 0.7:
        for _ in range(rnd.randint(1, 6)):
            r1, r2 = sorted(rnd.sample(group_pools, 2))
            coupling.append((r1, rnd.randrange(len(pools[r1][1])), r2, rnd.randrange(len(pools[r2][1])), rnd.choice([0, 64, 128, 256, 256])))
    return pools, coupling


def random_request(rnd, pools):
    entries = []
    for r, (kind, groups, size) in enumerate(pools):
        if rnd.random() < (0.9 if r == 0 else 0.5):
            total = sum(len(g) for g in groups) if kind != api.SUM_POOL else size // 10_000
            k = rnd.choice([api.COMPACT] * 4 + [api.TIGHT, api.SCATTER, api.FORCE_COMPACT, api.FORCE_TIGHT, api.ALL])
            units = rnd.randint(0, max(1, min(total, 5)))
            frac = rnd.choice([0, 0, 0, 2500, 5000, 5000, 7500, 1, 3333])
            if units == 0 and frac == 0:
                units = 1
            entries.append((r, k, units * 10_000 + frac))
    if not entries:
        entries.append((0, api.COMPACT, 10_000))
    return entries


def plain(al):
    return None if al is None else [(ra.resource_id, ra.amount, [(i.index, i.group_idx, i.fractions) for i in ra.indices]) for ra in al.resources]


def make(mod, pools, coupling):
    return mod.ResourceAllocator(mod.Descriptor([mod.PoolDesc(k, [list(g) for g in gs], s) for (k, gs, s) in pools], list(coupling)))


def attempt(fn):
    """(value, None) or (None, 'error'): an assertion of the reference's own invariants counts as the same outcome on both sides."""
    try:
        return fn(), None
    except (AssertionError, api.AllocatorError):
        return None, "error"


@pytest.mark.parametrize("seed", range(60))
def test_random_sequences_match_oracle(seed):
    rnd = random.Random(1000 + seed)
    pools, coupling = random_descriptor(rnd)
    a, o = make(api, pools, coupling), make(ora, pools, coupling)
    live, panicked = [], False
    for step in range(60):
        if live and rnd.random() < 0.4:
            x, y = live.pop(rnd.randrange(len(live)))
            a.release_allocation(x)
            o.release_allocation(y)
        else:
            ent = random_request(rnd, pools)
            ra, ro = [api.Entry(*e) for e in ent], [ora.Entry(*e) for e in ent]
            if rnd.random() < 0.2:
                (ea, erra), (eo, erro) = attempt(lambda: a.is_enabled(ra)), attempt(lambda: o.is_enabled(ro))
                assert (ea, erra) == (eo, erro), (seed, step, ent)
                if erra:
                    panicked = True
                    break
            (x, erra), (y, erro) = attempt(lambda: a.try_allocate(ra)), attempt(lambda: o.try_allocate(ro))
            assert erra == erro, (seed, step, ent, erra, erro)
            if erra:
                panicked = True  # the reference would have panicked here: the run ends for both
                break
            assert plain(x) == plain(y), (seed, step, ent)
            if x is not None:
                live.append((x, y))
        for r in range(len(pools)):
            assert a.get_current_free(r) == o.get_current_free(r)
            assert a.free_groups(r) == o.free_groups(r), (seed, step, r)
            assert a.concise_amount_sum(r) == o.concise_amount_sum(r)
        a.validate()
        o.validate()
    if panicked:
        return
    for x, y in live:
        a.release_allocation(x)
        o.release_allocation(y)
    for r, (kind, groups, size) in enumerate(pools):
        full = size if kind == api.SUM_POOL else sum(len(g) for g in groups) * 10_000
        assert a.get_current_free(r) == full and a.free_amount_sum(r) == full


def test_argument_errors():
    a = api.ResourceAllocator(api.Descriptor([api.regular_sockets(2, 2)]))
    with pytest.raises(api.AllocatorError) as e:
        a.try_allocate([api.Entry(0, api.COMPACT, 10_000), api.Entry(0, api.COMPACT, 10_000)])  # not strictly increasing
    assert e.value.code == api.HQALLOC_E_INVALID
    with pytest.raises(api.AllocatorError):
        a.try_allocate([api.Entry(0, 9, 10_000)])  # unknown kind
    with pytest.raises(api.AllocatorError):
        a.try_allocate([api.Entry(0, api.COMPACT, 0)])  # zero amount (request.rs:24-32)
    assert a.try_allocate([api.Entry(7, api.COMPACT, 10_000)]) is None  # a resource this worker does not have (allocator.rs:124-126)
    al = a.try_allocate([api.Entry(0, api.COMPACT, 10_000)])
    a.release_allocation(al)
    with pytest.raises(api.AllocatorError):
        a.release_allocation(al)  # released twice
    with pytest.raises(api.AllocatorError):
        api.ResourceAllocator(api.Descriptor([api.PoolDesc(api.GROUPS_POOL, [[0, 1]], 0)], [(0, 5, 0, 0, 1)]))  # coupling to a group that does not exist
