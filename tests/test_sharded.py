"""Worker-sharded tick: partition, sink layout, and the one all-gather (gloo, world_size 2) on CPU with the oracle standing in
for the per-rank tick; the HIP shards themselves are checked in test_gpu_parity.py::test_sharded_*."""
import os
import socket

import numpy as np
import pytest

from hyperqueue_amd import abi, sharded, workloads
from hyperqueue_amd.core import SchedEnv, TaskBuilder as TB, WorkerBuilder as WB


def make_env() -> SchedEnv:
    env = SchedEnv(abi.make_config(reserve=2, fill_max=5, time_limit_s=20.0))
    env.new_named_resource("gpus/amd")
    env.new_workers(7, WB(8).res_sum("gpus/amd", 2))
    env.new_workers(3, WB(4))
    env.new_tasks(120, TB().cpus(1))
    env.new_tasks(20, TB().cpus(2).add_resource(1, 1).user_priority(1))
    env.new_tasks(9, TB().cpus(4))
    return env


def test_owner_partition_covers_every_worker_once():
    ids = np.arange(1, 4097, dtype=np.uint32)
    for world in (1, 2, 3, 8):
        owners = np.asarray([sharded.owner_of(int(i), world) for i in ids])
        assert owners.min() >= 0 and owners.max() < world
        if world > 1:  # FxHash spreads consecutive ids: every shard gets its fair share within 20 %
            counts = np.bincount(owners, minlength=world)
            assert counts.min() > 0.8 * len(ids) / world and counts.max() < 1.2 * len(ids) / world
    # same partition as workloads.shard_workers (the snapshot-level helper)
    snap = workloads.make("c2", n_tasks=10, n_workers=64)
    for r in range(4):
        assert workloads.shard_workers(snap, r, 4).worker_id.tolist() == [int(i) for i in snap.worker_id if sharded.owner_of(int(i), 4) == r]


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_pack_merge_roundtrip(world):
    from oracle.oracle import Oracle

    env = make_env()
    snap = env.snapshot()
    full = Oracle(env.config, canonical=True).tick(snap)
    cap = 256
    total = sharded.sink_layout(len(snap.worker_id), cap)[4]
    merged = np.concatenate([sharded.pack_shard(full, snap.worker_id, r, world, cap) for r in range(world)])
    assert merged.size == world * total
    assert sharded.merge_shards(merged, world, len(snap.worker_id), cap) == full.records
    assert sum(len(r) for r in full.records) > 0


def test_sink_too_small_is_an_error():
    from oracle.oracle import Oracle

    env = make_env()
    snap = env.snapshot()
    full = Oracle(env.config, canonical=True).tick(snap)
    with pytest.raises(ValueError):
        sharded.pack_shard(full, snap.worker_id, 0, 1, 3)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank: int, world: int, port: int, out_q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import Oracle

        env = make_env()
        snap = env.snapshot()
        o = Oracle(env.config, canonical=True)
        st = sharded.ShardedTick(env.config, rank=rank, world=world, records_per_shard=256, backend=o.tick)
        got = st.tick(snap)
        want = o.tick(snap)
        ok = got.records == want.records and got.counts == want.counts and (got.new_free == want.new_free).all()
        mine = sum(len(want.records[w]) for w in range(len(snap.worker_id)) if sharded.owner_of(int(snap.worker_id[w]), world) == rank)
        out_q.put((rank, bool(ok), mine, sum(len(r) for r in want.records)))
    finally:
        dist.destroy_process_group()


def test_two_rank_allgather_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for (_, ok, _, _) in res), res
    total = res[0][3]
    assert sum(m for (_, _, m, _) in res) == total and all(0 < m < total for (_, _, m, _) in res)  # both shards carried records


def test_divergent_placements_are_detected():
    """placement_checksum in the sink headers: shards built from different placements must not merge silently"""
    from oracle.oracle import Oracle

    env = make_env()
    snap = env.snapshot()
    full = Oracle(env.config, canonical=True).tick(snap)
    other = abi.Result(**{**full.__dict__, "counts": full.counts[:-1]})  # another replica's (different) placement
    cap = 256
    merged = np.concatenate([sharded.pack_shard(full, snap.worker_id, 0, 2, cap), sharded.pack_shard(other, snap.worker_id, 1, 2, cap)])
    with pytest.raises(sharded.ShardDivergence):
        sharded.merge_shards(merged, 2, len(snap.worker_id), cap)


def _rank_main_divergent(rank: int, world: int, port: int, out_q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import Oracle

        env = make_env()
        snap = env.snapshot()
        o = Oracle(env.config, canonical=True)
        want = o.tick(snap)

        def replica_tick(s):  # rank 1's solver "ran into its time limit" and holds another incumbent: one task less on the last worker that has any
            r = o.tick(s)
            if rank == 1:
                w = max(i for i, recs in enumerate(r.records) if recs)
                r.records[w] = r.records[w][:-1]
                r.counts = r.counts[:-1] + [(r.counts[-1][0], r.counts[-1][1], r.counts[-1][2], r.counts[-1][3] - 1)]
                r.is_optimal = False
            return r

        st = sharded.ShardedTick(env.config, rank=rank, world=world, records_per_shard=256, backend=replica_tick)
        got = st.tick(snap)
        ok = got.records == want.records and got.counts == want.counts and st.n_divergent == 1  # every rank ends up with rank 0's placement
        out_q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_rank_divergence_falls_back_to_rank0_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main_divergent, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for (_, ok) in res), res


# ---- row f3 sharded: every rank encodes the messages of its own workers (here: the kernels' phase functions through the CPU debug hook), one
# all-gather of the byte buffers, every rank holds the tick's full message list
def _wire_rank_main(rank: int, world: int, port: int, out_q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wire_cases as wc
        from hyperqueue_amd import wire

        sc = wc.random_scenario(12, n_workers=9)  # 59 records, 18 retracts, one multi-node task
        tables, full = wc.tables_and_records(*sc)
        mine = wire.shard_records(full, rank, world)
        res = wire.encode_host_debug(tables, mine, 1 << 20)
        got = wire.all_gather_messages(res, mine, world, 1 << 20)
        want = wc.oracle_messages(*sc)
        own_bytes = res.total_bytes
        out_q.put((rank, got == want, own_bytes, sum(len(b) for _, b in want)))
    finally:
        dist.destroy_process_group()


def test_two_rank_wire_allgather_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wire_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for (_, ok, _, _) in res), res
    total = res[0][3]
    assert sum(b for (_, _, b, _) in res) == total and all(0 < b < total for (_, _, b, _) in res)  # both shards carried message bytes
