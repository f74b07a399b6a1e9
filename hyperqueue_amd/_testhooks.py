"""Loader of libhqtick_test.so: the product objects plus the CPU test hooks of include/hqtick_debug.h (-DHQTICK_TEST_HOOKS).

Test infrastructure: only tests/ and tools/ import this.  The product library (libhqtick.so, hyperqueue_amd/tick.py) exports none of
these symbols and has no CPU path."""
from __future__ import annotations

import ctypes as C
import os

_LIB = None
# HQTICK_TEST_LIB: another build of the same library (tools/host_asan.sh: the AddressSanitizer / UBSan build)
LIB_PATH = os.environ.get("HQTICK_TEST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhqtick_test.so")


def load() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = C.CDLL(LIB_PATH)
    return _LIB
