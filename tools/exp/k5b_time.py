import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from hyperqueue_amd import abi, workloads
from hyperqueue_amd.tick import Tick
snap = workloads.make("c3")
t = Tick(abi.make_config(time_limit_s=5.0, flags=abi.HQTICK_FLAG_COMPACT_RECORDS))
t.upload_ready(snap.task_id, snap.task_priority, snap.task_rq)
sc = snap.to_c(); t.cluster_upload(sc)
def run(label):
    ks = []
    for i in range(40):
        t.tick_raw(sc, resident=True); ks.append(t.kernel_stats())
    print(label, "expand_mapping us", round(float(np.median([k["other_us"] for k in ks[10:]])), 2), flush=True)
for rep in range(2):
    for v in sys.argv[1:]:
        os.environ.pop("HQK_EXP", None)
        if v != "0": os.environ["HQK_EXP"] = v
        run("exp=" + v)
