"""The multi-GPU path with REAL ranks: one process per GPU, every rank runs libhqtick.so (hqtick_set_shard + device record sink) and the merge is
the library's own RCCL all-gather (hqtick_comm_init / hqtick_shard_allgather).  Needs >= 2 visible GPUs — skipped on the 1-GPU box; the
driver's multi-GPU bench (`bench.py --gpus N`) runs the same code path."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["HQ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["HQ_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
from hyperqueue_amd import abi, sharded, workloads
from hyperqueue_amd.tick import Tick
snap = workloads.make("c4", n_tasks=30000, n_workers=96)
cfg = abi.make_config(time_limit_s=20.0, device_index=rank)
st = sharded.ShardedTick(cfg, rank=rank, world=world, records_per_shard=1 << 15)
assert st.collective == "library"
got = st.tick(snap)
if rank == 0:
    t = Tick(cfg); want = t.tick(snap); t.close()
    assert got.counts == want.counts and got.records == want.records, "sharded != unsharded"
# forced divergence: rank 1 pretends its replica placed differently -> every rank takes rank 0's placement, resident sets stay in step
st.t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
import dataclasses
res = st.tick(snap, resident=True)
st.consume_last()
counts = torch.tensor([st.t.ready_count()], dtype=torch.int64, device="cuda")
allc = [torch.zeros_like(counts) for _ in range(world)]
dist.all_gather(allc, counts)
assert len({int(c.item()) for c in allc}) == 1, "resident ready sets diverged"
dist.barrier(); dist.destroy_process_group()
print("RANK_OK", rank)
'''


WORKER_ONE_GPU = r'''
import os, sys
sys.path.insert(0, os.environ["HQ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["HQ_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
from hyperqueue_amd import abi, sharded, workloads
from hyperqueue_amd.tick import Tick
cfg = abi.make_config(time_limit_s=20.0, device_index=0)
st = sharded.ShardedTick(cfg, rank=rank, world=world, records_per_shard=1 << 15, collective="host")
for name, kw in (("c4", dict(n_tasks=30000, n_workers=96)), ("c3p", dict(n_tasks=4000, n_workers=10))):
    snap = workloads.make(name, **kw)
    got = st.tick(snap)
    if rank == 0:
        t = Tick(cfg); want = t.tick(snap); t.close()
        assert got.counts == want.counts and got.records == want.records and got.retracts == want.retracts, "sharded != unsharded: " + name
# the placement itself split over the two ranks (hqtick_set_exchange, here over gloo): coupled ticks — every rank sweeps its half of the worker blocks with
# k_price_sweep, one small all-gather per sweep — and separable steady-state ticks — every rank solves every second class block with k_block_solve.  The env of this
# process puts the thresholds at 1 so that these small models are split.  Against the unsharded tick on rank 0: equal counts and records, bit for bit.
sys.path.insert(0, os.path.join(os.environ["HQ_ROOT"], "tools"))
from price_fuzz import scenario
cases = [("c3p-64", workloads.make("c3p", n_tasks=160_000, n_workers=64)), ("c3p-100", workloads.make("c3p", n_tasks=250_000, n_workers=100)),
         ("c4-unsat-96", workloads.make("c4", seed=8, n_workers=96, n_tasks=1_400)), ("fuzz-2003", scenario(2003)[0]), ("fuzz-2005", scenario(2005)[0]),
         ("steady-c3-48", workloads.make_steady("c3", seed=0, n_workers=48, n_tasks=60_000)), ("steady-c4-40", workloads.make_steady("c4", seed=2, n_workers=40, n_tasks=60_000))]
for name, snap in cases:
    got = st.tick(snap)
    ks = st.t.kernel_stats()
    assert ks["exchange_calls"] > 0, name + ": the solve was not split"
    if name.startswith("steady"):
        assert ks["exchange_calls"] == 1 and ks["n_classes_device"] > 0, (name, ks["exchange_calls"], ks["n_classes_device"])
    else:
        assert ks["price_sweeps"] > 0 and ks["exchange_calls"] >= ks["price_sweeps"], (name, ks["price_sweeps"], ks["exchange_calls"])
    if rank == 0:
        t = Tick(cfg); want = t.tick(snap); kw = t.kernel_stats(); t.close()
        assert got.status == want.status and got.is_optimal == want.is_optimal and got.batches == want.batches, "sharded solve != unsharded: " + name
        assert got.counts == want.counts and got.records == want.records and (got.new_free == want.new_free).all(), "sharded solve != unsharded: " + name
        assert (ks["price_sweeps"], ks["price_rounds"]) == (kw["price_sweeps"], kw["price_rounds"]), (name, ks["price_sweeps"], kw["price_sweeps"])
        assert kw["exchange_calls"] == 0
print("SOLVE_SPLIT_OK", rank, flush=True)
# resident ready set + consume on every replica, then a forced divergence (rank 1 corrupts its checksum word): every rank takes rank 0's placement
snap = workloads.make("c3", n_tasks=20000, n_workers=32)
st.t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
res = st.tick(snap, resident=True)
st.consume_last()
if rank == 1:
    st._corrupt_next_checksum = True
n0 = st.n_divergent
out = st.tick(snap, resident=True)
assert st.n_divergent == n0 + 1, "the forced divergence was not detected"
st.consume_last()
counts = torch.tensor([st.t.ready_count()], dtype=torch.int64)
allc = [torch.zeros_like(counts) for _ in range(world)]
dist.all_gather(allc, counts)
assert len({int(c.item()) for c in allc}) == 1, "resident ready sets diverged"
dist.barrier(); dist.destroy_process_group()
print("RANK_OK", rank)
'''


def test_two_ranks_on_one_gpu_run_the_library_shards():
    """Two processes, each one rank of a 2-way sharded scheduler running libhqtick.so (hqtick_set_shard + device record sink) on the SAME MI355X;
    the shards are merged across the processes through gloo (host-staged) because RCCL refuses two ranks on one device.  Everything of the
    multi-process path except the RCCL call itself — which test_gpu_parity.py::test_library_allgather_single_rank and the two-GPU test below cover."""
    env = dict(os.environ, HQ_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0",
               HQTICK_SHARD_MIN_BLOCKS="1", HQTICK_SHARD_MIN_CLASSES="1", HQTICK_PRICE_MIN_COLS="16", HQTICK_BLOCK_MIN_CLASSES="1")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER_ONE_GPU], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o and f"SOLVE_SPLIT_OK {r}" in o, o[-3000:]


def test_the_rccl_exchange_of_the_sharded_solve_single_rank():
    """the exchange the sharded solve issues per sweep, through the library's OWN RCCL communicator (one rank: what this box can run): host buffer -> HBM ->
    ncclAllGather -> host, recv == send; and what one such exchange costs (printed: the figure DESIGN.md §7 prices a sharded sweep with)"""
    import ctypes as C
    import time

    import numpy as np

    from hyperqueue_amd import abi
    from hyperqueue_amd.tick import Tick

    t = Tick(abi.make_config(), measure=True)  # libhqtick_test.so: the product objects + include/hqtick_debug.h
    lib = t._lib
    uid = (C.c_ubyte * 128)()
    lib.hqtick_comm_unique_id.argtypes = [C.c_void_p]
    lib.hqtick_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.hqtick_debug_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    assert lib.hqtick_comm_unique_id(uid) == 0
    assert lib.hqtick_comm_init(t._ctx, uid, 0, 1) == 0
    for n in (64, 14_400, 131_072):
        a = np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8); b = np.zeros(n, np.uint8)
        assert lib.hqtick_debug_exchange(t._ctx, a.ctypes.data, b.ctypes.data, n) == 0
        assert (a == b).all()
        t0 = time.perf_counter()
        for _ in range(50):
            lib.hqtick_debug_exchange(t._ctx, a.ctypes.data, b.ctypes.data, n)
        print(f"RCCL exchange, 1 rank, {n} bytes: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call")
    t.close()


def test_two_rank_library_allgather():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = dict(os.environ, HQ_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-3000:]
