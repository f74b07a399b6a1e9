"""The C-ABI library loads on a machine without a GPU, exports every symbol include/hqtick.h declares, its struct
layouts match the ctypes mirror, and it fails loudly (no CPU fallback) when asked to run without a gfx950 device."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from hyperqueue_amd import abi, tick

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header: str, prefix: str):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"_[a-z0-9_]+)\s*\(", src)))


@pytest.mark.parametrize("header,prefix", [("hqtick.h", "hqtick"), ("hqwire.h", "hqwire")])
def test_every_declared_symbol_is_exported(header, prefix):
    lib = tick.load()
    names = _declared(header, prefix)
    assert len(names) >= 2
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/{header} but not exported by libhqtick.so"


def test_test_hooks_live_in_the_test_library_only():
    """include/hqtick_debug.h is exported by libhqtick_test.so and by nothing else: the product library has no CPU entry point."""
    from hyperqueue_amd import _testhooks

    names = _declared("hqtick_debug.h", "hqtick") + _declared("hqtick_debug.h", "hqwire")
    assert len(names) >= 8
    test_lib, product = _testhooks.load(), tick.load()
    for n in names:
        assert hasattr(test_lib, n), f"{n} declared in include/hqtick_debug.h but not exported by libhqtick_test.so"
        assert not hasattr(product, n), f"libhqtick.so exports the test hook {n}"
    exported = subprocess.check_output(["nm", "-D", "--defined-only", tick.LIB_PATH]).decode()
    assert "debug" not in exported, [l for l in exported.splitlines() if "debug" in l]


def _dynamic(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return sorted(l.split()[-1] for l in out.splitlines() if l.strip())


def test_dynamic_symbol_tables_are_exactly_the_headers():
    """-fvisibility=hidden + csrc/exports.map: what include/*.h declares is ALL the libraries export — no mangled C++ (libstdc++ instantiations, the host
    model, the solver), no helper with external linkage (285 such symbols leaked)."""
    from hyperqueue_amd import _testhooks, build as b

    product = sorted(set(_declared("hqtick.h", "hqtick") + _declared("hqwire.h", "hqwire")))
    assert _dynamic(tick.LIB_PATH) == product
    hooks = sorted(set(product + _declared("hqtick_debug.h", "hqtick") + _declared("hqtick_debug.h", "hqwire")))
    _testhooks.load()
    assert _dynamic(b.TEST_LIB) == hooks
    assert _dynamic(b.ALLOC_LIB) == sorted(set(_declared("hqalloc.h", "hqalloc")))


def test_rust_binding_file_matches_the_headers():
    """integration/hqtick_sys.rs (the `extern "C"` block a tako maintainer adds; VERDICT r03 missing 5) is generated from include/hqtick.h + include/hqwire.h:
    regenerating it gives the committed file, every declared function is in it, and every struct has the ctypes mirror's field list in the same order
    (the mirror's layout is checked against a compiled C probe in test_struct_layouts_match_header)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "tools", "gen_rust_sys.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    text = open(os.path.join(ROOT, "integration", "hqtick_sys.rs")).read()
    assert text == gen.generate(), "include/*.h changed: run python tools/gen_rust_sys.py"
    for n in _declared("hqtick.h", "hqtick") + _declared("hqwire.h", "hqwire"):
        assert re.search(r"pub fn " + n + r"\(", text), n
    pairs = {"HqtickConfig": abi.Config, "HqtickSnapshot": abi.SnapshotC, "HqtickQueryWorkers": abi.QueryWorkersC, "HqtickResult": abi.ResultC,
             "HqtickQueryResult": abi.QueryResultC, "HqtickKernelStats": abi.KernelStatsC, "HqtickGraphStats": abi.GraphStatsC}
    for rs_name, cls in pairs.items():
        body = re.search(r"pub struct " + rs_name + r" \{[^\n]*\n(.*?)\n\}", text, flags=re.S).group(1)
        fields = re.findall(r"pub (\w+):", body)
        assert fields == [f for f, _ in cls._fields_], rs_name
    shim = open(os.path.join(ROOT, "integration", "shim.rs")).read()
    for sym in set(re.findall(r"(?<![.\w])(hqtick_[a-z_0-9]+)\(", shim)):  # what the shim calls exists in the binding (`.hqtick_ctx()` is tako's own accessor)
        assert re.search(r"pub fn " + sym + r"\(", text), sym
    for fld in set(re.findall(r"\bres\.(\w+)\b(?!\()", shim)):  # ... and the result fields it reads exist
        assert fld in [f for f, _ in abi.ResultC._fields_], fld


def test_versions():
    lib = tick.load()
    assert lib.hqtick_abi_version() == abi.HQTICK_ABI_VERSION
    assert lib.hqtick_build_arch() == b"gfx950"


def test_struct_layouts_match_header():
    """Compile a tiny C program against include/hqtick.h printing sizeof/offsetof; compare with the ctypes mirror."""
    pairs = {
        "hqtick_config": abi.Config, "hqtick_snapshot": abi.SnapshotC, "hqtick_query_workers": abi.QueryWorkersC,
        "hqtick_result": abi.ResultC, "hqtick_query_result": abi.QueryResultC, "hqtick_kernel_stats": abi.KernelStatsC, "hqtick_graph_stats": abi.GraphStatsC,
    }
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "hqtick.h"', "int main(void){"]
    for cname, cls in pairs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _ in cls._fields_:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, src])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(l.split() for l in out if l)
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, f"{cname}.{f}"


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device path cannot be observed")
    with pytest.raises(tick.HqTickError) as e:
        tick.Tick(abi.make_config())
    assert e.value.code == abi.HQTICK_E_NO_DEVICE
    lib = tick.load()
    assert lib.hqtick_create(None, None) == abi.HQTICK_E_INVALID
    assert lib.hqtick_run(None, None, None) == abi.HQTICK_E_INVALID
