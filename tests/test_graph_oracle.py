"""Pins oracle/graph_oracle.py with the reference's own dependency tests (tests/graph_cases.py)."""
import pytest

from graph_cases import CASES
from oracle.graph_oracle import GraphOracle


def run_oracle(steps):
    g = GraphOracle()
    for st in steps:
        op = st[0]
        if op == "add":
            ready = g.on_new_tasks([(i, 0, 0, deps) for i, deps in st[1]])
            if "ready" in st[2]:
                assert ready == sorted(st[2]["ready"])
            for i, n in st[2].get("unfinished", {}).items():
                assert g.unfinished(i) == n
        elif op == "take":
            g.take_from_ready(st[1])
        elif op == "finish":
            rel, unknown = g.task_finished(st[1])
            assert unknown == 0 and rel == sorted(st[2]["released"])
            for i, n in st[2].get("unfinished", {}).items():
                assert g.unfinished(i) == n
        elif op == "fail":
            removed, _ = g.remove([st[1]], recursive=True)
            assert removed == sorted(st[2]["removed"])
        elif op == "collect":
            s = set()
            g.collect_recursive_consumers(st[1], s)
            assert sorted(s) == sorted(st[2]["consumers"])
        elif op == "exists":
            for i, e in st[1].items():
                assert (i in g.tasks) == e
    return g


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_case(name):
    run_oracle(CASES[name]())


def test_dep_later_in_batch_is_dropped():
    # reactor.rs:193-203: find_task_mut misses a task that the same call has not inserted yet
    g = GraphOracle()
    assert g.on_new_tasks([(1, 0, 0, [2]), (2, 0, 0, [1])]) == [1]
    assert g.unfinished(1) == 0 and g.unfinished(2) == 1


def test_unknown_finished_is_counted():
    g = GraphOracle()
    g.on_new_tasks([(1, 0, 0, [])])
    g.take_from_ready([1])
    assert g.task_finished([7, 1, 1]) == ([], 2)


# ---- the b-level extension's checker (no reference counterpart: a definition, checked on graphs small enough to do by hand) -----------------------------------------
def test_blevels_are_the_longest_paths_to_a_sink():
    from oracle.graph_oracle import GraphOracle

    g = GraphOracle()
    #   1 -> 2 -> 4 -> 6        3 -> 4        5 (alone)        2 -> 7
    g.on_new_tasks([(1, 10 << 32, 0, []), (2, 10 << 32, 0, [1]), (3, 10 << 32, 0, []), (4, 10 << 32, 0, [2, 3]), (5, 10 << 32, 0, []), (6, 10 << 32, 0, [4]), (7, 10 << 32, 0, [2])])
    assert g.blevels() == {1: 3, 2: 2, 3: 2, 4: 1, 5: 0, 6: 0, 7: 0}
    assert g.apply_blevels() == 3
    assert g.tasks[1].priority == (10 << 32) | 3 and g.ready[1][0] == (10 << 32) | 3 and g.ready[5][0] == 10 << 32
    # the path through 4 goes away with it: 2 keeps its consumer 7, 3 becomes a sink
    g.remove([4], recursive=True)
    assert g.blevels() == {1: 2, 2: 1, 3: 0, 5: 0, 7: 0}
    # finished tasks leave the map: what is left is measured
    g.take_from_ready([1, 3, 5])
    g.task_finished([1])
    assert g.blevels() == {2: 1, 3: 0, 5: 0, 7: 0}
