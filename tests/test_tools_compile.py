"""bench.py, __graft_entry__.py and every script under tools/ at least compile, and the shell script parses (they run on the GPU box, where a
syntax error would cost a measurement)."""
import glob
import os
import py_compile
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "exp", "*.py")) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py"),
                                                                   os.path.join(ROOT, "profiles", "summarize.py")])


@pytest.mark.parametrize("path", SCRIPTS, ids=lambda p: os.path.basename(p))
def test_script_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


def test_shell_scripts_parse():
    for sh in glob.glob(os.path.join(ROOT, "tools", "*.sh")):
        assert subprocess.run(["bash", "-n", sh]).returncode == 0, sh


def test_bench_cli_and_wire_block_error_path():
    """`bench.py --help` parses; the wire block (a subprocess) reports an error dict instead of raising when it cannot run (no GPU here)"""
    import sys

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "--wire-iters" in p.stdout
    sys.path.insert(0, ROOT)
    import torch

    if not torch.cuda.is_available():
        import bench

        out = bench.wire_block(1)
        assert isinstance(out, dict) and "error" in out


def test_the_bench_line_is_short_and_strict_json():
    """BENCH_r05 came back `parsed: None` because the line was 20 KB: the line bench.py prints is built by ONE function, has exactly the contract's keys, stays below
    4 KB whatever strings go in, and parses under a strict JSON parser (no NaN / Infinity)."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    def strict(text):
        def bad(c):
            raise ValueError(c)
        return json.loads(text, parse_constant=bad)

    kw = dict(value=51234567.891, n_gpus=8, steps=20, warmup=5, ms_per_step=1.0234, scaling="weak", workload="c3p: " + "w" * 3000, priority_levels=3, parallelism="p" * 3000, ranks_in_comm=8, seed=0,
              p50_tick_ms=float("nan"), assigned_per_tick=50900, all_done=True, caches_live="none", p50_warm_ms=float("inf"), model_columns=8205.0, price_sweeps=3.0, gpu_busy_share=0.25,
              roofline={"kernel": "k_level_hist (K1)", "achieved": 2000.1, "peak": 8000.0, "frac": 0.25, "algorithmic_bytes_per_launch": 12_000_000, "avg_launch_us": 5.98, "traffic": None},
              dominant={"kernel": "k_price_sweep", "avg_us": 72.5, "launches_per_tick": 3, "share_of_tick": 0.25},
              cpu={"value": 10182.0, "cores": 1, "cpu": "c" * 500, "kind": "port", "sample": "s" * 5000, "tick_s": 5.289, "is_optimal": False})
    line, text = bench.headline_line(**kw)
    assert len(text) < 4096 and "\n" not in text
    got = strict(text)
    assert got == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in got, k
    assert got["config"]["p50_tick_ms"] is None and got["config"]["p50_tick_ms_identical_ticks_warm_caches"] is None   # non-finite numbers never reach the line
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "dominant_kernel"):
        assert k in got["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    # no CPU baseline (N > 1) and no roofline (timing off): still a valid short line
    line2, text2 = bench.headline_line(**dict(kw, cpu=None, roofline=None, dominant=None))
    assert strict(text2)["cpu_baseline"] is None and len(text2) < 4096
