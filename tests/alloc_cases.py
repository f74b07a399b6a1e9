"""The reference's allocator tests, transcribed (worker/resources/test_allocator.rs under
/root/reference/crates/tako/src/internal/; every case cites its lines).  Each case takes `api`: a module-like object with
`ResourceAllocator`, `Descriptor`, `Entry`, `request`, `amount`, the pool constructors and the request kinds -- the oracle
(`oracle/alloc_oracle.py`) and the product's host mirror (`hyperqueue_amd/allocator.py`) both provide it, so the same
assertions pin the oracle and then check the C ABI.
"""
from __future__ import annotations


def units(api, n):
    return api.amount(n, 0)


def simple_descriptor(api, n_sockets, socket_size):
    """test_allocator.rs:42-53"""
    return api.Descriptor([api.regular_sockets(n_sockets, socket_size)])


def rq(api, *entries):
    """ResBuilder...finish(); an entry is (id, kind, amount) with amount in fixed point."""
    return api.request([api.Entry(i, k, a) for (i, k, a) in entries])


def cpus_compact(api, n, *more):
    return rq(api, (0, api.COMPACT, units(api, n)), *more)


def simple_allocator(api, free, running):
    """test_allocator.rs:59-90"""
    pools = []
    for i, c in enumerate(free):
        total = c + sum(r[i] for r in running if i < len(r))
        pools.append(api.simple_indices(total))
    ac = api.ResourceAllocator(api.Descriptor(pools))
    for r in running:
        assert len(r) == 1
        assert ac.try_allocate(rq(api, (0, api.COMPACT, units(api, r[0])))) is not None
    return ac


def simple_alloc(api, ac, counts, expect_pass):
    """test_allocator.rs:92-108"""
    al = ac.try_allocate(rq(api, *[(i, api.COMPACT, units(api, c)) for i, c in enumerate(counts) if c > 0]))
    assert (al is not None) == expect_pass


def assert_free_simple(ac, counts):
    """ConciseFreeResources::assert_eq (concise.rs:224-226): one group per resource, no fraction entries at all."""
    for r, c in enumerate(counts):
        assert ac.free_groups(r) == [(c, {})]


def get_indices(al, r):
    """allocation.rs:91-98"""
    return [a.index for a in next(x for x in al.resources if x.resource_id == r).indices]


def get_groups(al, r):
    """allocation.rs:99-110"""
    out = {}
    for a in next(x for x in al.resources if x.resource_id == r).indices:
        out[a.group_idx] = out.get(a.group_idx, 0) + 1
    return out


def get_sockets(al, r):
    """pool.rs:568-590 for a groups pool ([0] for an indices pool)."""
    return sorted(set(a.group_idx for a in next(x for x in al.resources if x.resource_id == r).indices))


# ---------------------------------------------------------------------------------------------------------------
def test_allocator_single_socket(api):
    """test_allocator.rs:110-124"""
    ac = simple_allocator(api, [4], [])
    simple_alloc(api, ac, [3], True)
    assert_free_simple(ac, [1])
    simple_alloc(api, ac, [2], False)
    assert_free_simple(ac, [1])
    simple_alloc(api, ac, [1], True)
    assert_free_simple(ac, [0])
    ac.validate()


def test_pool_single_socket(api):
    """test_allocator.rs:126-181"""
    ac = api.ResourceAllocator(simple_descriptor(api, 1, 4))
    al = ac.try_allocate(cpus_compact(api, 3))
    assert len(al.resources) == 1 and al.resources[0].resource_id == 0
    assert len(al.resources[0].indices) == 3
    assert all(i < 4 for i in get_indices(al, 0))
    assert ac.try_allocate(cpus_compact(api, 2)) is None
    ac.release_allocation(al)
    al = ac.try_allocate(cpus_compact(api, 4))
    assert get_indices(al, 0) == [3, 2, 1, 0]
    ac.release_allocation(al)
    assert ac.free_amount_sum(0) == units(api, 4)
    assert len(ac.free_groups(0)) == 1
    r1, r2 = cpus_compact(api, 1), cpus_compact(api, 2)
    al1, al2, al3, al4 = (ac.try_allocate(r1) for _ in range(4))
    assert None not in (al1, al2, al3, al4)
    assert ac.try_allocate(r1) is None and ac.try_allocate(r2) is None
    ac.release_allocation(al2)
    ac.release_allocation(al4)
    al5 = ac.try_allocate(r2)
    assert al5 is not None
    assert ac.try_allocate(r1) is None and ac.try_allocate(r2) is None
    v = get_indices(al1, 0)
    assert len(v) == 1
    v += get_indices(al5, 0)
    assert len(v) == 3
    v += get_indices(al3, 0)
    assert sorted(v) == [0, 1, 2, 3]
    ac.validate()


def test_pool_compact1(api):
    """test_allocator.rs:183-237"""
    ac = api.ResourceAllocator(simple_descriptor(api, 4, 6))
    rq1 = cpus_compact(api, 4)
    s1 = get_sockets(ac.try_allocate(rq1), 0)
    assert len(s1) == 1
    s2 = get_sockets(ac.try_allocate(rq1), 0)
    assert len(s2) == 1 and s1 != s2
    rq2 = cpus_compact(api, 3)
    s3 = get_sockets(ac.try_allocate(rq2), 0)
    assert len(s3) == 1
    s4 = get_sockets(ac.try_allocate(rq2), 0)
    assert len(s4) == 1
    assert s3 != s1 and s4 != s1 and s3 != s2 and s4 != s2 and s3 == s4
    for n, sockets in ((6, 1), (7, 2), (8, 2), (9, 3)):
        al = ac.try_allocate(cpus_compact(api, n))
        assert len(get_sockets(al, 0)) == sockets
        ac.release_allocation(al)
    ac.validate()


def test_pool_allocate_compact_all(api):
    """test_allocator.rs:239-257"""
    ac = api.ResourceAllocator(simple_descriptor(api, 4, 6))
    al = ac.try_allocate(cpus_compact(api, 24))
    assert get_indices(al, 0) == list(range(24))
    assert ac.get_current_free(0) == 0
    ac.release_allocation(al)
    assert ac.get_current_free(0) == units(api, 24)
    ac.validate()


def test_pool_allocate_all(api):
    """test_allocator.rs:259-281"""
    ac = api.ResourceAllocator(simple_descriptor(api, 4, 6))
    rq_all = rq(api, (0, api.ALL, 0))
    al = ac.try_allocate(rq_all)
    assert get_indices(al, 0) == list(range(24))
    assert ac.get_current_free(0) == 0
    ac.release_allocation(al)
    assert ac.get_current_free(0) == units(api, 24)
    assert ac.try_allocate(cpus_compact(api, 1)) is not None
    assert ac.try_allocate(rq_all) is None
    ac.validate()


def test_pool_force_compact1(api):
    """test_allocator.rs:283-300"""
    ac = api.ResourceAllocator(simple_descriptor(api, 2, 4))
    assert ac.try_allocate(rq(api, (0, api.FORCE_COMPACT, units(api, 9)))) is None
    rq2 = rq(api, (0, api.FORCE_COMPACT, units(api, 2)))
    for _ in range(4):
        al = ac.try_allocate(rq2)
        assert len(get_indices(al, 0)) == 2 and len(get_sockets(al, 0)) == 1
    assert ac.try_allocate(rq2) is None
    ac.validate()


def test_pool_force_compact2(api):
    """test_allocator.rs:302-321"""
    ac = api.ResourceAllocator(simple_descriptor(api, 2, 4))
    rq1 = rq(api, (0, api.FORCE_COMPACT, units(api, 3)))
    for _ in range(2):
        al = ac.try_allocate(rq1)
        assert len(get_indices(al, 0)) == 3 and len(get_sockets(al, 0)) == 1
    assert ac.try_allocate(rq(api, (0, api.FORCE_COMPACT, units(api, 2)))) is None
    assert ac.try_allocate(cpus_compact(api, 2)) is not None
    ac.validate()


def test_pool_force_compact3(api):
    """test_allocator.rs:323-348"""
    ac = api.ResourceAllocator(simple_descriptor(api, 3, 4))
    for n, sockets in ((8, 2), (5, 2), (10, 3)):
        al = ac.try_allocate(rq(api, (0, api.FORCE_COMPACT, units(api, n))))
        assert len(get_indices(al, 0)) == n and len(get_sockets(al, 0)) == sockets
        ac.release_allocation(al)
        ac.validate()


def test_pool_force_scatter1(api):
    """test_allocator.rs:350-371"""
    ac = api.ResourceAllocator(simple_descriptor(api, 3, 4))
    for n, sockets in ((3, 3), (4, 3), (2, 2)):
        al = ac.try_allocate(rq(api, (0, api.SCATTER, units(api, n))))
        assert len(get_indices(al, 0)) == n and len(get_sockets(al, 0)) == sockets
    ac.validate()


def test_pool_force_scatter2(api):
    """test_allocator.rs:373-387"""
    ac = api.ResourceAllocator(simple_descriptor(api, 3, 4))
    assert ac.try_allocate(rq(api, (0, api.FORCE_COMPACT, units(api, 4)))) is not None
    al2 = ac.try_allocate(rq(api, (0, api.SCATTER, units(api, 5))))
    assert len(get_indices(al2, 0)) == 5 and len(get_sockets(al2, 0)) == 2
    ac.validate()


def test_pool_generic_resources(api):
    """test_allocator.rs:389-481"""
    ac = api.ResourceAllocator(api.Descriptor([
        api.regular_sockets(1, 4), api.range_pool(5, 100), api.sum_pool(units(api, 100_000_000)),
        api.simple_indices(2), api.simple_indices(2)]))
    assert_free_simple(ac, [4, 96, 100_000_000, 2, 2])
    r = rq(api, (0, api.COMPACT, units(api, 1)), (4, api.COMPACT, units(api, 1)), (1, api.COMPACT, units(api, 12)),
           (2, api.COMPACT, units(api, 1_000_000)))
    al = ac.try_allocate(r)
    assert [x.resource_id for x in al.resources] == [0, 1, 2, 4]
    assert len(al.resources[1].indices) == 12
    assert al.resources[2].amount == units(api, 1_000_000)
    assert len(al.resources[3].indices) == 1
    assert ac.get_current_free(1) == units(api, 84)
    assert ac.get_current_free(2) == units(api, 99_000_000)
    assert ac.get_current_free(3) == units(api, 2)
    assert ac.get_current_free(4) == units(api, 1)
    r2 = cpus_compact(api, 1, (4, api.COMPACT, units(api, 2)))
    assert ac.try_allocate(r2) is None
    ac.release_allocation(al)
    assert ac.get_current_free(1) == units(api, 96)
    assert ac.get_current_free(2) == units(api, 100_000_000)
    assert ac.get_current_free(3) == units(api, 2)
    assert ac.get_current_free(4) == units(api, 2)
    assert ac.try_allocate(r2) is not None
    ac.validate()


def test_allocator_sum_max_fractions(api):
    """test_allocator.rs:483-507"""
    ac = api.ResourceAllocator(api.Descriptor([api.sum_pool(api.amount(0, 300))]))
    assert ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(1, 0)))) is None
    assert ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(0, 301)))) is None
    assert ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(0, 250)))) is not None


def test_allocator_indices_and_fractions(api):
    """test_allocator.rs:509-565"""
    ac = api.ResourceAllocator(simple_descriptor(api, 1, 4))
    assert ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(4, 1)))) is None
    al1 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(2, 1500))))
    assert [i.fractions for i in al1.resources[0].indices] == [0, 0, 1500]
    assert al1.resources[0].amount == api.amount(2, 1500)
    r = rq(api, (0, api.COMPACT, api.amount(0, 5200)))
    al2 = ac.try_allocate(r)
    assert [i.fractions for i in al2.resources[0].indices] == [5200]
    assert al2.resources[0].indices[0].index == al1.resources[0].indices[2].index
    assert al2.resources[0].amount == api.amount(0, 5200)
    al3 = ac.try_allocate(r)
    assert [i.fractions for i in al3.resources[0].indices] == [5200]
    assert al3.resources[0].indices[0].index != al1.resources[0].indices[2].index
    assert al3.resources[0].amount == api.amount(0, 5200)
    assert ac.try_allocate(r) is None
    ac.release_allocation(al1)
    assert ac.concise_amount_sum(0) == api.amount(2, 9600)
    ac.release_allocation(al3)
    ac.release_allocation(al2)
    assert ac.concise_amount_sum(0) == api.amount(4, 0)


def test_allocator_fractions_compactness(api):
    """test_allocator.rs:567-608"""
    ac = api.ResourceAllocator(simple_descriptor(api, 1, 2))
    rq1 = rq(api, (0, api.COMPACT, api.amount(0, 7500)))
    rq2 = rq(api, (0, api.COMPACT, api.amount(0, 2500)))
    al1, al2 = ac.try_allocate(rq1), ac.try_allocate(rq1)
    al3, al4 = ac.try_allocate(rq2), ac.try_allocate(rq2)
    assert None not in (al1, al2, al3, al4)
    assert ac.concise_amount_sum(0) == 0
    ac.release_allocation(al1)
    ac.release_allocation(al2)
    assert ac.concise_amount_sum(0) == api.amount(1, 5000)
    rq3 = rq(api, (0, api.COMPACT, api.amount(1, 5000)))
    assert ac.try_allocate(rq3) is None
    ac.release_allocation(al4)
    al5 = ac.try_allocate(rq3)
    assert al5 is not None
    ac.release_allocation(al3)
    ac.release_allocation(al5)
    assert ac.concise_amount_sum(0) == api.amount(2, 0)


def test_allocator_groups_and_fractions_scatter(api):
    """test_allocator.rs:610-636"""
    ac = api.ResourceAllocator(simple_descriptor(api, 3, 2))
    assert ac.try_allocate(rq(api, (0, api.SCATTER, api.amount(6, 1)))) is None
    r = rq(api, (0, api.SCATTER, api.amount(2, 5000)))
    al1, al2 = ac.try_allocate(r), ac.try_allocate(r)
    ac.validate()
    r1, r2 = al1.resources[0].indices, al2.resources[0].indices
    assert r1[2].fractions == 5000 and r2[2].fractions == 5000
    assert r1[2].group_idx == r2[2].group_idx
    ac.release_allocation(al1)
    ac.release_allocation(al2)
    assert ac.concise_amount_sum(0) == units(api, 6)


def test_allocator_groups_and_fractions(api):
    """test_allocator.rs:638-714"""
    ac = api.ResourceAllocator(simple_descriptor(api, 3, 2))
    assert ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(6, 1)))) is None
    al1 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(3, 5000))))
    r1 = al1.resources[0].indices
    assert len(r1) == 4
    assert r1[0].group_idx == r1[1].group_idx and r1[2].group_idx == r1[3].group_idx and r1[0].group_idx != r1[2].group_idx
    al2 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(0, 4000))))
    r2 = al2.resources[0].indices
    assert len(r2) == 1 and r2[0].group_idx == r1[2].group_idx
    ac.release_allocation(al1)
    al3 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(2, 8000))))
    r3 = al3.resources[0].indices
    assert len(r3) == 3
    assert r3[0].group_idx == r3[2].group_idx or r3[1].group_idx == r3[2].group_idx
    assert r3[0].group_idx != r3[1].group_idx
    assert all(i.index != j.index for i in r3 for j in r2)
    al4 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(0, 7000))))
    assert al4 is not None
    ac.validate()
    ac.release_allocation(al2)
    al6 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(2, 3000))))
    r5 = al6.resources[0].indices
    assert [i.fractions for i in r5] == [0, 0, 3000]
    assert r5[0].group_idx == r5[2].group_idx or r5[1].group_idx == r5[2].group_idx
    assert r5[1].group_idx != r5[0].group_idx
    ac.validate()
    ac.release_allocation(al3)
    ac.release_allocation(al4)
    ac.release_allocation(al6)
    assert ac.concise_amount_sum(0) == units(api, 6)


def test_allocator_sum_fractions(api):
    """test_allocator.rs:716-785"""
    ac = api.ResourceAllocator(api.Descriptor([api.sum_pool(units(api, 2))]))
    assert ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(2, 3000)))) is None
    al1 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(1, 3000))))
    assert al1.resources[0].indices == [] and al1.resources[0].amount == api.amount(1, 3000)
    ac.validate()
    assert ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(0, 7001)))) is None
    al2 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(0, 7000))))
    assert al2.resources[0].indices == [] and al2.resources[0].amount == api.amount(0, 7000)
    ac.release_allocation(al1)
    assert ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(2, 0)))) is None
    assert ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(1, 3001)))) is None
    al3 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(1, 0))))
    al4 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(0, 2000))))
    assert al3 is not None and al4 is not None
    ac.release_allocation(al4)
    assert ac.concise_amount_sum(0) == api.amount(0, 3000)
    ac.release_allocation(al2)
    ac.release_allocation(al3)
    assert ac.concise_amount_sum(0) == units(api, 2)


def test_coupling1(api):
    """test_allocator.rs:787-827"""
    for i in range(3):
        coupling = [(0, j, 2, j, 256) for j in range(4)]
        ac = api.ResourceAllocator(api.Descriptor(
            [api.regular_sockets(4, 3), api.regular_sockets(4, 1), api.regular_sockets(4, 4)], coupling))
        for _ in range(i):
            assert ac.try_allocate(cpus_compact(api, 2)) is not None
        al3 = ac.try_allocate(cpus_compact(api, 2, (2, api.COMPACT, units(api, 2))))
        s1, s2 = get_sockets(al3, 0), get_sockets(al3, 2)
        assert len(s1) == 1 and s1 == s2
        assert len(get_indices(al3, 0)) == 2 and len(get_indices(al3, 2)) == 2
        ac.validate()


def descriptor_cpus_gpus(api, n_sockets, size1, size2, coupled):
    """test_allocator.rs:829-854"""
    coupling = [(0, j, 1, j, 256) for j in range(n_sockets)] if coupled else []
    return api.Descriptor([api.regular_sockets(n_sockets, size1), api.regular_sockets(n_sockets, size2)], coupling)


def test_coupling2(api):
    """test_allocator.rs:856-878"""
    ac = api.ResourceAllocator(descriptor_cpus_gpus(api, 4, 4, 2, True))
    al1 = ac.try_allocate(cpus_compact(api, 4, (1, api.COMPACT, units(api, 3))))
    ac.validate()
    s1, s2 = get_sockets(al1, 0), get_sockets(al1, 1)
    assert len(s1) == 1 and len(s2) == 2 and s1[0] in s2
    g0 = get_groups(al1, 0)
    assert len(g0) == 1 and all(x == 4 for x in g0.values())
    assert sorted(get_groups(al1, 1).values()) == [1, 2]


def test_coupling3(api):
    """test_allocator.rs:880-898"""
    ac = api.ResourceAllocator(descriptor_cpus_gpus(api, 4, 4, 2, True))
    al1 = ac.try_allocate(rq(api, (0, api.COMPACT, api.amount(0, 1000)), (1, api.COMPACT, api.amount(0, 5000))))
    s1, s2 = get_sockets(al1, 0), get_sockets(al1, 1)
    assert len(s1) == 1 and s1 == s2
    g0, g1 = get_groups(al1, 0), get_groups(al1, 1)
    assert g0 == g1 and list(g1.values()) == [1]


def test_complex_coupling1(api):
    """test_allocator.rs:900-949"""
    coupling = []
    for i in range(6):
        coupling.append((0, i, 1, i // 2, 256))
        coupling.append((1, i // 2, 2, i, 128))
    ac = api.ResourceAllocator(api.Descriptor(
        [api.regular_sockets(6, 2), api.regular_sockets(3, 1), api.regular_sockets(6, 3)], coupling))
    ac.force_claim_from_groups(0, [0], units(api, 1))
    ac.force_claim_from_groups(2, [5], units(api, 2))
    al1 = ac.try_allocate(rq(api, (0, api.FORCE_COMPACT, units(api, 4)), (1, api.FORCE_COMPACT, units(api, 1)),
                             (2, api.FORCE_COMPACT, units(api, 5))))
    g = get_groups(al1, 0)
    assert sorted(g.keys()) == [2, 3] and sorted(g.values()) == [2, 2]
    g = get_groups(al1, 1)
    assert sorted(g.keys()) == [1] and sorted(g.values()) == [1]
    g = get_groups(al1, 2)
    assert sorted(g.keys()) == [2, 3] and sorted(g.values()) == [2, 3]


def test_complex_coupling2(api):
    """test_allocator.rs:951-988"""
    coupling = [(0, 2, 1, 1, 256), (0, 0, 1, 1, 128), (1, 1, 2, 0, 256)]
    ac = api.ResourceAllocator(api.Descriptor(
        [api.regular_sockets(3, 1), api.regular_sockets(3, 1), api.regular_sockets(3, 1)], coupling))
    al1 = ac.try_allocate(rq(api, (0, api.FORCE_COMPACT, units(api, 1)), (1, api.FORCE_COMPACT, units(api, 1)),
                             (2, api.FORCE_COMPACT, units(api, 1))))
    assert get_indices(al1, 0) == [2]
    assert get_indices(al1, 1) == [1]
    assert get_indices(al1, 2) == [0]


def test_coupling_force2(api):
    """test_allocator.rs:990-1009"""
    for coupled in (True, False):
        ac = api.ResourceAllocator(descriptor_cpus_gpus(api, 3, 2, 2, coupled))
        for g in (0, 1):
            ac.force_claim_from_groups(0, [g], units(api, 2))
        for g in (1, 2):
            ac.force_claim_from_groups(1, [g], units(api, 2))
        r = rq(api, (0, api.FORCE_COMPACT, units(api, 1)), (1, api.FORCE_COMPACT, units(api, 1)))
        assert (ac.try_allocate(r) is None) == coupled


def test_coupling_force3(api):
    """test_allocator.rs:1011-1036"""
    ac = api.ResourceAllocator(descriptor_cpus_gpus(api, 4, 2, 2, True))
    for g in (0, 1):
        ac.force_claim_from_groups(0, [g], units(api, 2))
    for g in (1, 3):
        ac.force_claim_from_groups(1, [g], units(api, 1))
    ac.validate()
    al = ac.try_allocate(rq(api, (0, api.FORCE_COMPACT, units(api, 3)), (1, api.FORCE_COMPACT, units(api, 3))))
    g0, g1 = get_groups(al, 0), get_groups(al, 1)
    assert len(g0) == 2 and 2 in g0 and 3 in g0
    assert len(g1) == 2 and 2 in g1 and 3 in g1


def test_compact_scattering(api):
    """test_allocator.rs:1038-1053"""
    ac = api.ResourceAllocator(simple_descriptor(api, 4, 4))
    r1 = ac.try_allocate(rq(api, (0, api.COMPACT, units(api, 6)))).resources[0].indices
    assert len(r1) == 6
    assert r1[0].group_idx == r1[1].group_idx == r1[2].group_idx
    assert r1[3].group_idx == r1[4].group_idx == r1[5].group_idx
    assert r1[0].group_idx != r1[3].group_idx


def test_tight_scattering(api):
    """test_allocator.rs:1055-1070"""
    ac = api.ResourceAllocator(simple_descriptor(api, 4, 4))
    r1 = ac.try_allocate(rq(api, (0, api.TIGHT, units(api, 6)))).resources[0].indices
    assert len(r1) == 6
    assert r1[0].group_idx == r1[1].group_idx == r1[2].group_idx == r1[3].group_idx
    assert r1[4].group_idx == r1[5].group_idx
    assert r1[0].group_idx != r1[4].group_idx


# ---------------------------------------------------------------------------------------------------------------
# End-to-end pins of the reference's pytest suite (paths relative to /root/reference/tests/): a worker with the given
# `--cpus` / `--resource` / `--coupling`, one task at a time on an idle worker, the indices the task saw in
# HQ_RESOURCE_VALUES_* grouped as the test groups them.  Labels there are strings; the allocator works on positions of the
# flattened groups (worker/resources/map.rs:28-37), so a label's group is its position's group.
def label_groups(api, groups):
    """`[[1, 2, 3, 4], [11, ...]]` -> pool over positions + position -> label."""
    flat = [x for g in groups for x in g]
    pos, out = 0, []
    for g in groups:
        out.append(list(range(pos, pos + len(g))))
        pos += len(g)
    return api.PoolDesc(api.GROUPS_POOL, out, 0), flat


def sizes_by_decade(al, r, labels):
    """`groups()` of test_resources.py:565-569 / test_coupling.py:160-168: Counter(label // 10) values, sorted."""
    c = {}
    for i in get_indices(al, r):
        c[labels[i] // 10] = c.get(labels[i] // 10, 0) + 1
    return sorted(c.values())


def test_e2e_tight_vs_compact_policy(api):
    """test_resources.py:572-596"""
    pool, labels = label_groups(api, [[1, 2, 3, 4], [11, 12, 13, 14], [21, 22, 23, 24]])
    for kind, expect in ((api.TIGHT, [2, 4]), (api.FORCE_TIGHT, [2, 4]), (api.COMPACT, [3, 3]), (api.FORCE_COMPACT, [3, 3])):
        ac = api.ResourceAllocator(api.Descriptor([pool]))
        al = ac.try_allocate(rq(api, (0, kind, units(api, 6))))
        assert sizes_by_decade(al, 0, labels) == expect
        ac.release_allocation(al)
        ac.validate()


def test_e2e_fractional_force_compact(api):
    """test_resources.py:599-606: a hundred 0.3 compact! tasks come and go, then 2.5 compact! takes groups of 1 and 2."""
    pool, labels = label_groups(api, [[1, 2], [11, 12], [21, 22]])
    ac = api.ResourceAllocator(api.Descriptor([pool]))
    small = rq(api, (0, api.FORCE_COMPACT, api.amount(0, 3000)))
    running = []
    for _ in range(100):
        al = ac.try_allocate(small)
        if al is None:  # worker full: the oldest task ends
            ac.release_allocation(running.pop(0))
            al = ac.try_allocate(small)
        assert al is not None
        running.append(al)
    for al in running:
        ac.release_allocation(al)
    ac.validate()
    al = ac.try_allocate(rq(api, (0, api.FORCE_COMPACT, api.amount(2, 5000))))
    assert sizes_by_decade(al, 0, labels) == [1, 2]


def test_e2e_job_num_of_cpus(api):
    """test_cpus.py:19-72 (worker `--cpus 3x4`)"""
    ac = api.ResourceAllocator(api.Descriptor([api.regular_sockets(3, 4)]))
    for kind, n, sockets in ((api.COMPACT, 1, 1), (api.SCATTER, 2, 2), (api.FORCE_COMPACT, 4, 1), (api.FORCE_COMPACT, 5, 2)):
        al = ac.try_allocate(rq(api, (0, kind, units(api, n))))
        idx = get_indices(al, 0)
        assert len(idx) == n and len(set(x // 4 for x in idx)) == sockets
        ac.release_allocation(al)
    al = ac.try_allocate(rq(api, (0, api.ALL, 0)))
    assert sorted(get_indices(al, 0)) == list(range(12))


def test_e2e_coupling_alloc1(api):
    """test_coupling.py:54-110: cpus [[1,2,3],[4,5,6]] coupled group-wise with foo [[10,20,30,40],[50,60,70,80]];
    `cpus=1` + `foo=3 compact!` (then `foo=1 compact!`) always lands in one NUMA group."""
    cpus, cl = label_groups(api, [[1, 2, 3], [4, 5, 6]])
    foo, fl = label_groups(api, [[10, 20, 30, 40], [50, 60, 70, 80]])
    ac = api.ResourceAllocator(api.Descriptor([cpus, api.sum_pool(units(api, 123)), foo], [(0, 0, 2, 0, 256), (0, 1, 2, 1, 256)]))
    for n_foo in (3, 1):
        running = []
        for _ in range(50):
            r = rq(api, (0, api.COMPACT, units(api, 1)), (2, api.FORCE_COMPACT, units(api, n_foo)))
            al = ac.try_allocate(r)
            while al is None:
                ac.release_allocation(running.pop(0))
                al = ac.try_allocate(r)
            g = set(cl[i] // 4 for i in get_indices(al, 0)) | set(fl[i] // 50 for i in get_indices(al, 2))
            assert len(g) == 1
            running.append(al)
        for al in running:
            ac.release_allocation(al)
        ac.validate()


def test_e2e_coupling_alloc2(api):
    """test_coupling.py:106-159: `cpus=4, foo=2` then `cpus=2, foo=4` on cpus [[1,2,3],[10,11,12],[21,22,23]] coupled group-wise with
    foo [[1,2,3],[10,11,12],[20,21,23]]; each task's cpus and foos together touch exactly two NUMA groups (label // 10).  Both jobs are
    submitted before the worker connects, so they are started back to back; the property must hold for either start order."""
    cpus, cl = label_groups(api, [[1, 2, 3], [10, 11, 12], [21, 22, 23]])
    foo, fl = label_groups(api, [[1, 2, 3], [10, 11, 12], [20, 21, 23]])
    coupling = [(0, g, 2, g, 256) for g in range(3)]
    jobs = [((0, api.COMPACT, units(api, 4)), (2, api.COMPACT, units(api, 2))), ((0, api.COMPACT, units(api, 2)), (2, api.COMPACT, units(api, 4)))]
    for order in (jobs, jobs[::-1]):
        ac = api.ResourceAllocator(api.Descriptor([cpus, api.sum_pool(units(api, 123)), foo], coupling))
        for entries in order:
            al = ac.try_allocate(rq(api, *entries))
            g = set(cl[i] // 10 for i in get_indices(al, 0)) | set(fl[i] // 10 for i in get_indices(al, 2))
            assert len(g) == 2, (entries, g)
        ac.validate()


def test_e2e_fractions_sharing_small(api):
    """test_resources.py:210-237: worker `foo=[a,b]`, four tasks of foo=0.4 run at once; every index is shared by exactly two of them"""
    ac = api.ResourceAllocator(api.Descriptor([api.simple_indices(4), api.simple_indices(2)]))
    seen = {}
    for _ in range(4):
        al = ac.try_allocate(rq(api, (1, api.COMPACT, api.amount(0, 4000))))
        (i,) = get_indices(al, 1)
        seen[i] = seen.get(i, 0) + 1
    assert seen == {0: 2, 1: 2}
    ac.validate()


def test_e2e_fractions_sharing_larger(api):
    """test_resources.py:240-270: ten indices, four tasks of foo=2.5 at once: eight whole indices used once each, two indices shared by two halves"""
    ac = api.ResourceAllocator(api.Descriptor([api.simple_indices(4), api.simple_indices(10)]))
    whole, frac = {}, {}
    for _ in range(4):
        al = ac.try_allocate(rq(api, (1, api.COMPACT, api.amount(2, 5000))))
        idx = get_indices(al, 1)
        for i in idx[:-1]:
            whole[i] = whole.get(i, 0) + 1
        frac[idx[-1]] = frac.get(idx[-1], 0) + 1
    assert tuple(frac.values()) == (2, 2) and tuple(whole.values()) == (1,) * 8
    ac.validate()


def test_e2e_fractions_scatter(api):
    """test_resources.py:163-182: foo=[[a,b,c],[i,j,k],[x,y],[v,w]], one task of foo=3.5 gets four different indices"""
    foo, _ = label_groups(api, [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]])
    ac = api.ResourceAllocator(api.Descriptor([api.simple_indices(4), foo]))
    al = ac.try_allocate(rq(api, (1, api.COMPACT, api.amount(3, 5000))))
    idx = get_indices(al, 1)
    assert len(idx) == 4 and len(set(idx)) == 4


def test_e2e_range_multiple_allocated_values(api):
    """test_resources.py:107-135 / :68-104 / :273-295: `fairy=range(31-36)`, three tasks of fairy=2 hold six different values of the range; a sum
    resource next to it yields no indices; `foo=[a,b,c]` with foo=2 yields two different labels"""
    ac = api.ResourceAllocator(api.Descriptor([api.simple_indices(4), api.range_pool(31, 36), api.sum_pool(units(api, 2_000_000))]))
    vals = []
    for _ in range(3):
        al = ac.try_allocate(rq(api, (1, api.COMPACT, units(api, 2)), (2, api.COMPACT, units(api, 1000))))
        assert len(get_indices(al, 1)) == 2 and get_indices(al, 2) == []
        vals += get_indices(al, 1)
    assert all(31 <= v <= 36 for v in vals) and len(set(vals)) == 6
    ac = api.ResourceAllocator(api.Descriptor([api.simple_indices(1), api.simple_indices(3)]))
    al = ac.try_allocate(rq(api, (1, api.COMPACT, units(api, 2))))
    assert len(set(get_indices(al, 1))) == 2 and set(get_indices(al, 1)) <= {0, 1, 2}


def test_e2e_fractions_blocked(api):
    """test_resources.py:606-610: `cpus=2`, three tasks of 0.6 cpus: the amounts add up to 1.8 <= 2, but a third 0.6 fits on no single cpu -- the
    worker rejects it (the server then blocks the request, test_reactor.rs:664-705) until one of the first two has finished"""
    ac = api.ResourceAllocator(api.Descriptor([api.simple_indices(2)]))
    r = rq(api, (0, api.COMPACT, api.amount(0, 6000)))
    a1, a2 = ac.try_allocate(r), ac.try_allocate(r)
    assert a1 is not None and a2 is not None and get_indices(a1, 0) != get_indices(a2, 0)
    assert ac.try_allocate(r) is None and not ac.is_enabled(r)
    ac.release_allocation(a1)
    assert ac.is_enabled(r) and ac.try_allocate(r) is not None


def test_e2e_strict_compact_blocked(api):
    """test_resources.py:613-617: cpus [[1,2,3],[11,12,13]], three tasks of `2 compact!`: two run (one per socket), the third would need both
    sockets and is rejected until a socket is whole again"""
    cpus, _ = label_groups(api, [[1, 2, 3], [11, 12, 13]])
    ac = api.ResourceAllocator(api.Descriptor([cpus]))
    r = rq(api, (0, api.FORCE_COMPACT, units(api, 2)))
    a1, a2 = ac.try_allocate(r), ac.try_allocate(r)
    assert len(get_sockets(a1, 0)) == 1 and len(get_sockets(a2, 0)) == 1 and get_sockets(a1, 0) != get_sockets(a2, 0)
    assert ac.try_allocate(r) is None
    ac.release_allocation(a2)
    a3 = ac.try_allocate(r)
    assert a3 is not None and len(get_sockets(a3, 0)) == 1


def test_e2e_integration_force_compact(api):
    """crates/tako/src/internal/tests/integration/test_resources.rs:251-276: `cpus = 4 compact!` runs on a 2 x 2 socket worker (both groups)"""
    ac = api.ResourceAllocator(api.Descriptor([api.regular_sockets(2, 2)]))
    al = ac.try_allocate(rq(api, (0, api.FORCE_COMPACT, units(api, 4))))
    assert al is not None and sorted(get_indices(al, 0)) == [0, 1, 2, 3] and get_sockets(al, 0) == [0, 1]


def test_e2e_coupling_combined(api):
    """test_coupling.py:154-199: the seven (cpus policy, foo policy) rows on an idle worker."""
    cpus, cl = label_groups(api, [[1, 2, 3, 4], [11, 12, 13, 14], [21, 22, 23, 24]])
    foo, fl = label_groups(api, [[1, 2], [10, 11], [22, 21]])
    coupling = [(0, g, 1, g, 256) for g in range(3)]
    T, S, Cp = api.TIGHT, api.SCATTER, api.COMPACT
    for kc, kf, expect in ((T, T, ([2, 4], [2])), (S, S, ([2, 2, 2], [1, 1])), (Cp, Cp, ([3, 3], [2])), (T, Cp, ([2, 4], [2])),
                           (Cp, T, ([3, 3], [2])), (S, T, ([2, 2, 2], [2])), (S, Cp, ([2, 2, 2], [2]))):
        ac = api.ResourceAllocator(api.Descriptor([cpus, foo], coupling))
        al = ac.try_allocate(rq(api, (0, kc, units(api, 6)), (1, kf, units(api, 2))))
        assert (sizes_by_decade(al, 0, cl), sizes_by_decade(al, 1, fl)) == expect, (kc, kf)
        ac.release_allocation(al)
        ac.validate()


CASES = [v for k, v in sorted(globals().items()) if k.startswith("test_") and callable(v)]
