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
