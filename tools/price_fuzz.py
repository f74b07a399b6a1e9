#!/usr/bin/env python
"""CPU campaign of the price path (csrc/price.cpp + the emulated k_price_sweep): random coupled ticks of mid size — clusters mid-run with every worker its own
free vector, several priority levels, ready sets that do not saturate — solved with the sweeps forced on from 16 model columns, against the host-only search.
Both claim a 1e-4 certificate, so whenever both are optimal the objectives must agree to 1e-4; the sweeps' point is verified row by row inside the solver
(CompSolver::polish_point).  Prints one line per disagreement and a summary.   python tools/price_fuzz.py [first_seed] [count] [procs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def scenario(seed):
    """(snapshot, W, levels, steady, n_ready) of one random mid-size coupled tick"""
    from hyperqueue_amd import workloads
    from hyperqueue_amd.core import priority_from_user

    rng = np.random.default_rng(seed)
    W = int(rng.choice([12, 16, 24, 32, 48, 64, 96, 128, 192]))
    levels = int(rng.integers(1, 4))
    steady = rng.random() < 0.6
    n_ready = int(rng.integers(W * 2, W * 40))
    if steady:
        snap = workloads.make_steady("c3", seed=seed, n_workers=W, n_tasks=n_ready, release=float(rng.choice([0.1, 0.3, 0.6])))
    else:
        snap = workloads.make("c3", seed=seed, n_workers=W, n_tasks=n_ready)
    snap.task_priority = np.asarray([priority_from_user(int(p)) for p in rng.integers(0, levels, len(snap.task_id))], np.uint64)
    return snap, W, levels, steady, n_ready


def one(seed):
    from hyperqueue_amd import abi
    from test_price import stages
    from test_host_stages import _objective
    from oracle.oracle import Oracle

    snap, W, levels, steady, n_ready = scenario(seed)
    t0 = time.time(); got, sweeps, rounds = stages(snap, True, min_cols=16, tl=5.0); tg = time.time() - t0
    t0 = time.time(); host, _, _ = stages(snap, False, tl=5.0); th = time.time() - t0
    with_highs = os.environ.get("PRICE_FUZZ_HIGHS") == "1"  # also time the reference-configured HiGHS (5 s limit) on the same snapshot
    o = Oracle(abi.make_config(time_limit_s=5.0 if with_highs else 0.05), reference_solver_options=True)
    t0 = time.time(); opt_ref = None
    try:
        opt_ref = bool(o.tick(snap).is_optimal)
    except Exception:
        pass
    tr = time.time() - t0
    m = o.last_model()
    zg, zh = _objective(m, got), _objective(m, host)
    if any(m["ctype"][j] != 0 and m["obj"][j] != 0 for j in range(len(m["obj"]))):
        # flag columns carry part of the objective (blocked workers) and the hook exports placement counts only: compare the COMPLETED objectives
        from test_host_stages import _completed_objective
        zg, zh = _completed_objective(m, got), _completed_objective(m, host)
    bad = got.is_optimal and host.is_optimal and abs(zg - zh) > 1e-4 * max(zg, zh) + 1e-12
    return dict(seed=seed, W=W, levels=levels, steady=steady, n=n_ready, sweeps=sweeps, rounds=rounds, opt_g=bool(got.is_optimal), opt_h=bool(host.is_optimal), zg=zg, zh=zh, tg=tg, th=th, bad=bool(bad), opt_ref=opt_ref if with_highs else None, tr=tr, zr=float(m["objective"]) if with_highs else None, cols=len(m["obj"]))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    from multiprocessing import Pool
    with Pool(procs) as p:
        rows = p.map(one, range(first, first + count), chunksize=1)
    bad = [r for r in rows if r["bad"]]
    for r in bad:
        print("DISAGREE", r)
    ran = [r for r in rows if r["sweeps"] > 0]
    print(f"{len(rows)} scenarios, sweeps ran on {len(ran)}, certified with sweeps {sum(r['opt_g'] for r in ran)}, host-only certified {sum(r['opt_h'] for r in rows)}, "
          f"sweeps better by >1e-4: {sum(1 for r in rows if r['zg'] > r['zh'] * (1 + 1e-4))}, host better by >1e-4: {sum(1 for r in rows if r['zh'] > r['zg'] * (1 + 1e-4))}, disagreements among certified: {len(bad)}")
    if rows and rows[0]["opt_ref"] is not None:
        print(f"reference-configured HiGHS (5 s): certified {sum(1 for r in rows if r['opt_ref'])}; certified by HiGHS but not by the sweeps path: "
              f"{[ (r['seed'], r['W'], r['levels'], r['cols'], round(r['tr'], 2)) for r in rows if r['opt_ref'] and not r['opt_g']]}; by the sweeps path but not HiGHS: {sum(1 for r in rows if r['opt_g'] and not r['opt_ref'])}; "
              f"HiGHS time {sum(r['tr'] for r in rows):.1f} s")
    print(f"time: sweeps path {sum(r['tg'] for r in rows):.1f} s, host path {sum(r['th'] for r in rows):.1f} s")


if __name__ == "__main__":
    main()
