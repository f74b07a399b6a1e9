#!/usr/bin/env python
"""tools/price_fuzz.py's family through the HIP tick (C ABI), the sweeps forced on from 16 columns as tests/test_gpu_price.py does: how many ticks come back certified within the
reference's 5 s limit, against plain HiGHS (the reference's options) on the same snapshots.   python tools/gpu_fuzz_family.py [first_seed] [count] [--default-path]
--default-path: the product's own thresholds (small models stay with the host search) instead of the forced sweeps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: F401
from hyperqueue_amd import abi
from hyperqueue_amd.tick import Tick
from price_fuzz import scenario

first = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000
count = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 60
if "--default-path" not in sys.argv:
    os.environ["HQTICK_PRICE_MIN_COLS"] = "16"
from oracle.oracle import Oracle

rows = []
seeds = [int(a) for a in sys.argv[1:] if a.isdigit()] if "--seeds" in sys.argv else list(range(first, first + count))
for seed in seeds:
    snap = scenario(seed)[0]
    t = Tick(abi.make_config(time_limit_s=5.0))
    try:
        t0 = time.time(); got = t.tick(snap); dt = time.time() - t0
        ks = t.kernel_stats()
    finally:
        t.close()
    o = Oracle(abi.make_config(time_limit_s=5.0), reference_solver_options=True)
    t0 = time.time(); want = o.tick(snap); dr = time.time() - t0
    # two certificates of the same optimum cannot be more than 1e-4 (relative) apart: the product's point in HiGHS's own model against HiGHS's objective
    from test_host_stages import _completed_objective
    hm = o.last_model()
    zp, zr = _completed_objective(hm, got), float(hm["objective"])
    apart = bool(got.is_optimal and want.is_optimal and abs(zp - zr) > 1e-4 * max(abs(zp), abs(zr)) + 1e-12)
    if apart:
        from limits import model_point, rows_hold
        feasible = rows_hold(hm, model_point(hm, got.counts))   # the product's point against every row of the reference's model (as HiGHS got it)
        # ... and who is right: the exact oracle (HiGHS with and without presolve, rel gap 0, the better verified answer — oracle/oracle.py: _highs says why neither mode of
        # scipy's HiGHS 1.8.0 is trusted alone)
        ex = Oracle(abi.make_config(time_limit_s=60.0)); ex.tick(snap); ze = float(ex.last_model()["objective"])
        print("CERTIFICATES APART", seed, "product", zp, "HiGHS", zr, "(HiGHS - product) / HiGHS", (zr - zp) / zr, "| the product's point satisfies every row of the reference's model:", feasible,
              "| exact oracle:", ze, "-> within 1e-4 of it: product", abs(zp - ze) <= 1e-4 * abs(ze), ", reference-configured HiGHS", abs(zr - ze) <= 1e-4 * abs(ze), flush=True)
    rows.append((seed, bool(got.is_optimal), bool(want.is_optimal), dt, dr, int(ks["price_sweeps"]), int(ks["milp_cols"]), apart))
    print(seed, "product", bool(got.is_optimal), f"{dt:.3f}s", "sweeps", int(ks["price_sweeps"]), "cols", int(ks["milp_cols"]), "| HiGHS", bool(want.is_optimal), f"{dr:.3f}s", flush=True)
n = len(rows)
print(f"{n} ticks: product certified {sum(r[1] for r in rows)}, HiGHS certified {sum(r[2] for r in rows)}; HiGHS but not the product: {[r[0] for r in rows if r[2] and not r[1]]}; the product but not HiGHS: {[r[0] for r in rows if r[1] and not r[2]]}; "
      f"product time {sum(r[3] for r in rows):.1f} s, HiGHS time {sum(r[4] for r in rows):.1f} s; both certified and more than 1e-4 apart: {[r[0] for r in rows if r[7]]}")
