"""one price_fuzz seed through the HIP tick with the solver's trace on (HQMILP_TRACE=1 in the environment):  python tools/exp/trace_seed.py <seed> [min_cols]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
if len(sys.argv) > 2:
    os.environ["HQTICK_PRICE_MIN_COLS"] = sys.argv[2]
from hyperqueue_amd import abi
from hyperqueue_amd.tick import Tick
from price_fuzz import scenario
snap = scenario(int(sys.argv[1]))[0]
t = Tick(abi.make_config(time_limit_s=5.0))
got = t.tick(snap)
print("seed", sys.argv[1], "optimal", got.is_optimal, "status", got.status, t.kernel_stats()["price_sweeps"])
t.close()
