"""Builds (gcc, once per session) a tiny C shim around include/hqtick_records.h and calls it through ctypes: the header-only record walker a host shim would
use, run against hqtick_result structs from Python.  Test infrastructure."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from hyperqueue_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SRC = r"""
#include "hqtick_records.h"
struct sink { uint64_t *task; uint8_t *variant, *kind; uint32_t *worker; int64_t n; };
static void put(void *u, uint32_t w, uint64_t task, uint8_t variant, uint8_t kind) {
    struct sink *s = (struct sink *)u;
    s->task[s->n] = task; s->variant[s->n] = variant; s->kind[s->n] = kind; s->worker[s->n] = w; s->n++;
}
long long walk_all(const hqtick_result *res, uint32_t n_workers, uint64_t *task, uint8_t *variant, uint8_t *kind, uint32_t *worker) {
    struct sink s = {task, variant, kind, worker, 0};
    return hqtick_all_records(res, n_workers, put, &s);
}
"""
_lib = None


def lib():
    global _lib
    if _lib is None:
        d = tempfile.mkdtemp(prefix="hqrec_")
        src, so = os.path.join(d, "walk.c"), os.path.join(d, "libwalk.so")
        open(src, "w").write(_SRC)
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), "-o", so, src])
        _lib = C.CDLL(so)
        _lib.walk_all.restype = C.c_longlong
        _lib.walk_all.argtypes = [C.POINTER(abi.ResultC), C.c_uint32, abi.u64p, abi.u8p, abi.u8p, abi.u32p]
    return _lib


def walk(res: abi.ResultC, n_workers: int):
    """(n or -1, task ids, variants, kinds, worker indices) in the order hqtick_all_records visits the records"""
    n_rec = int(np.ctypeslib.as_array(res.rec_off, shape=(n_workers + 1,))[n_workers]) if n_workers else 0
    task, var, kind, wk = np.zeros(max(n_rec, 1), np.uint64), np.zeros(max(n_rec, 1), np.uint8), np.zeros(max(n_rec, 1), np.uint8), np.zeros(max(n_rec, 1), np.uint32)
    n = int(lib().walk_all(C.byref(res), n_workers, task.ctypes.data_as(abi.u64p), var.ctypes.data_as(abi.u8p), kind.ctypes.data_as(abi.u8p), wk.ctypes.data_as(abi.u32p)))
    k = max(n, 0)
    return n, task[:k].tolist(), var[:k].tolist(), kind[:k].tolist(), wk[:k].tolist()
