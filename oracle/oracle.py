"""TEST INFRASTRUCTURE — Python driver of the CPU oracle (oracle/hq_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The oracle restates the reference tick in C++ and delegates the placement MILP to HiGHS, exactly as the
reference does (crates/tako/src/internal/solver/highs.rs:65-88).  The HiGHS used here is the one bundled in
scipy (HiGHS 1.8.0, `scipy.optimize.milp`); the reference pins `highs` 2.4.0 / `highs-sys` 1.15.0
(Cargo.lock:1107-1122) — same solver, different release, so ties between equally good optima may differ.

`canonical=True` additionally post-processes the optimum into the tie-break convention the MI355X path
implements: per connected component of the model, among all solutions whose objective is within
1e-9 (relative) of the component optimum, the one that minimises the last column, then the one before it, ...
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
import time
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from hyperqueue_amd import abi  # noqa: E402  (ctypes mirror of include/hqtick.h only)

_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

SOLVE_FN = C.CFUNCTYPE(
    C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8),
    C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_double,
    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int),
)


class ModelView(C.Structure):
    _fields_ = [
        ("ncols", C.c_int), ("nrows", C.c_int),
        ("obj", C.POINTER(C.c_double)), ("col_kind", abi.u8p), ("col_type", abi.u8p), ("col_worker", abi.u32p),
        ("col_rq", abi.u32p), ("col_variant", abi.u8p),
        ("row_type", abi.u8p), ("rhs", C.POINTER(C.c_double)), ("row_off", C.POINTER(C.c_int)),
        ("row_col", C.POINTER(C.c_int)), ("row_coef", C.POINTER(C.c_double)),
        ("x", C.POINTER(C.c_double)), ("objective", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile the C++ restatement (g++ only; no reference sources are copied or compiled)."""
    src = [os.path.join(_HERE, "hq_oracle.cpp"), os.path.join(_HERE, "hb_emul.h"), os.path.join(_ROOT, "include", "hqtick.h")]
    if not force and os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src):
        return _LIB_PATH
    os.makedirs(os.path.dirname(_LIB_PATH), exist_ok=True)
    # built beside its place and moved there: several test processes may find the library stale at once (pytest -n, the CPU campaigns), and none may load a half-written file
    tmp = f"{_LIB_PATH}.{os.getpid()}.tmp"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", tmp, src[0]])
    os.replace(tmp, _LIB_PATH)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_create.restype = C.c_void_p
        _lib.oracle_create.argtypes = [C.POINTER(abi.Config)]
        _lib.oracle_destroy.argtypes = [C.c_void_p]
        _lib.oracle_last_error.restype = C.c_char_p
        _lib.oracle_last_error.argtypes = [C.c_void_p]
        _lib.oracle_batches.argtypes = [C.c_void_p, C.POINTER(abi.SnapshotC), C.POINTER(abi.ResultC)]
        _lib.oracle_tick.argtypes = [C.c_void_p, C.POINTER(abi.SnapshotC), SOLVE_FN, C.c_void_p, C.POINTER(abi.ResultC)]
        _lib.oracle_query.argtypes = [C.c_void_p, C.POINTER(abi.SnapshotC), C.POINTER(abi.QueryWorkersC), SOLVE_FN, C.c_void_p, C.POINTER(abi.QueryResultC)]
        _lib.oracle_last_model.argtypes = [C.c_void_p, C.POINTER(ModelView)]
        _lib.oracle_set_given_counts.argtypes = [C.c_void_p, C.c_uint32, abi.u32p, abi.u8p, abi.u32p, abi.u32p, C.c_int]
        _lib.oracle_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        _lib.oracle_prune_progressive.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, abi.u32p]
        _lib.oracle_gap.argtypes = [C.c_void_p, C.POINTER(abi.SnapshotC), C.c_uint32, C.c_uint32, C.c_uint32, SOLVE_FN, C.c_void_p]
        _lib.oracle_hb_order_u32.argtypes = [abi.u32p, C.c_uint32, abi.u32p]
        _lib.oracle_hb_order_taskid.argtypes = [abi.u64p, C.c_uint32, abi.u64p]
        _lib.oracle_hb_order_rqv.argtypes = [abi.u32p, abi.u8p, C.c_uint32, abi.u32p, abi.u8p]
        _lib.oracle_fx_u32.restype = C.c_uint64
        _lib.oracle_fx_u32.argtypes = [C.c_uint32]
    return _lib


# ------------------------------------------------------------------------------------------------------
# HiGHS (scipy) behind the reference's LpInnerSolver contract: maximise, integer columns 0.. / 0..=1
# ------------------------------------------------------------------------------------------------------
_PRESOLVE_ON = False  # set by Oracle(reference_solver_options=True): HiGHS exactly as the reference configures it (cpu_baseline timing)


def _highs(c_max, A, lo, hi, lb, ub, time_limit=None):
    """maximise c_max.x with HiGHS; returns (x rounded, status) or (None, status)."""
    from scipy.optimize import Bounds, LinearConstraint, milp

    n = len(c_max)
    cons = [LinearConstraint(A, lo, hi)] if A is not None and A.shape[0] else []
    # exact by default (the canonical oracle); as the reference runs it (Oracle(reference_solver_options=True): only `time_limit` is set,
    # solver/highs.rs:65-68) HiGHS keeps its default mip_rel_gap = 1e-4, i.e. "Optimal" means proven within 0.01 %
    opts = {"mip_rel_gap": 1e-4 if _PRESOLVE_ON else 0.0, "disp": False}
    if time_limit is not None and time_limit < 1e20:
        opts["time_limit"] = float(time_limit)
    # HiGHS 1.8.0's PRESOLVE is not reliable on the placement models: tests/test_gpu_fuzz.py found instances it declares infeasible
    # (seed 403) or "optimal" below the true optimum (seed 676: 0.5490 vs 0.5882) although the exact solver's answer checks out row by
    # row and HiGHS itself reaches that optimum with presolve off.  So: no presolve on models small enough for it not to matter, and a
    # retry without it whenever a large model comes back infeasible.  (bench.py's cpu_baseline times the large models as the reference
    # would run them, presolve on.)
    nnz = int(A.nnz) if A is not None else 0
    cobj = -np.asarray(c_max, float)

    def run(extra):
        return milp(cobj, constraints=cons, integrality=np.ones(n), bounds=Bounds(lb, ub), options=dict(opts, **extra))

    def row_ok(x):  # HiGHS's own acceptance: mip_feasibility_tolerance = 1e-6
        if A is None or not A.shape[0]:
            return True
        a = A @ x
        return bool(np.all(a >= np.asarray(lo) - 1e-6) and np.all(a <= np.asarray(hi) + 1e-6))

    if _PRESOLVE_ON or nnz > 200_000:  # the reference's configuration (cpu_baseline timing), and models too large to solve twice
        res = run({})
        if res.status == 2:
            res = run({"presolve": False})
        if res.x is None or res.status not in (0, 1):
            return None, res.status
        return np.round(res.x), res.status
    # Small models are solved TWICE, without and with presolve, and the better verified answer wins.  Neither mode of HiGHS 1.8.0 can be trusted
    # alone: the presolve declares feasible models infeasible / returns "optimal" below the optimum (above), and WITHOUT presolve it does the
    # same on other instances (tools/host_fuzz.py seed 6364: 0.5714 "optimal" vs 0.6071 with presolve; the exact solver's 0.6071 satisfies every row).
    best = None
    statuses = []
    for extra in ({"presolve": False}, {"presolve": True}):
        res = run(extra)
        statuses.append(res.status)
        if res.x is None or res.status not in (0, 1):
            continue
        x = np.round(res.x)
        if not row_ok(x):
            continue
        z = float(np.dot(np.asarray(c_max, float), x))
        if best is None or z > best[0] + 1e-12 * abs(best[0]):
            best = (z, x, res.status)
        elif res.status == 0 and best[2] != 0 and z >= best[0] - 1e-12 * abs(best[0]):
            best = (z, x, res.status)
    if best is None:
        return None, statuses[0]
    # "optimal" only if some run proved it; a run that timed out with a better incumbent than the other's "optimum" cannot happen for a correct
    # solver, and if it does the better point is still the one to compare against
    return best[1], (0 if 0 in statuses and best[2] == 0 else best[2])


def solve_milp(obj, kind, rtype, rhs, roff, rcol, rcoef, time_limit: float = 60.0, canonical: bool = False):
    """Returns (x, objective, is_optimal) or None.  solver/highs.rs:51-88 (maximise; nat = 0.., bool = 0..=1)."""
    from scipy.sparse import csr_matrix

    n, m = len(obj), len(rhs)
    obj = np.asarray(obj, float)
    if n == 0:
        return np.zeros(0), 0.0, True
    ub = np.where(np.asarray(kind) == 1, 1.0, np.inf)
    lb = np.zeros(n)
    A = None
    lo = hi = None
    if m:
        A = csr_matrix((np.asarray(rcoef, float), np.asarray(rcol, np.int32), np.asarray(roff, np.int32)), shape=(m, n))
        A.sum_duplicates()
        lo = np.where(np.asarray(rtype) == 1, -np.inf, np.asarray(rhs, float))  # Max => (-inf, rhs]
        hi = np.where(np.asarray(rtype) == 0, np.inf, np.asarray(rhs, float))  # Min => [rhs, inf)
    if canonical:
        return _solve_canonical(obj, A, lo, hi, lb, ub)
    x, status = _highs(obj, A, lo, hi, lb, ub, time_limit)
    if x is None:
        return None
    return x, float(np.dot(obj, x)), status == 0


_IN_LAZY = [False]


def _column_components(A, n):
    """connected components of the columns of A (two columns are connected when a row holds both), through the bipartite row/column graph:
    nnz(A) edges instead of the column-by-column product, which a single row with thousands of entries blows up quadratically"""
    from scipy.sparse import bmat, csr_matrix
    from scipy.sparse.csgraph import connected_components

    m = A.shape[0]
    pat = csr_matrix((np.ones(A.nnz, np.int8), A.indices, A.indptr), shape=A.shape) if hasattr(A, "indptr") else (A != 0).astype(np.int8)
    g = bmat([[None, pat.T], [pat, None]], format="csr")  # nodes: n columns, then m rows
    _, lab_all = connected_components(g, directed=False)
    lab = lab_all[:n]
    uniq, lab = np.unique(lab, return_inverse=True)
    return len(uniq), lab


def _solve_canonical(obj, A, lo, hi, lb, ub):
    """The tie-break convention of the MI355X path, computed with HiGHS only:
    per connected component of the row/column graph, among the feasible integer vectors whose objective is within 1e-9
    (relative) of the component optimum, minimise the last column, then the one before it, and so on."""
    from scipy.sparse import csr_matrix, vstack
    from scipy.sparse.csgraph import connected_components

    n = len(obj)
    m = A.shape[0] if A is not None else 0
    # Lazy counting rows.  The batch-size rows `sum x <= size` (scheduler/solver.rs:264-271: all-ones, no lower bound) are the only rows that tie the
    # workers of a cut-free tick together; at BASELINE config 4 (4096 workers, no batch saturated) they make the model ONE component of 65 536
    # columns on which HiGHS holds an unproven incumbent after minutes.  A relaxation argument that needs no insight into the product: drop those
    # rows, solve what falls apart component by component, and check the dropped rows on the result — if they hold, the point is optimal for the
    # full model (optimal for a relaxation, feasible) and, every component carrying its own lexicographic minimum, it is the canonical one.
    # If one is violated the full model is solved as before.
    if m and not _IN_LAZY[0]:
        Acsr = A.tocsr()
        nnz_row = np.diff(Acsr.indptr)
        not_one = np.add.reduceat((Acsr.data != 1.0).astype(np.int64), np.minimum(Acsr.indptr[:-1], max(Acsr.nnz - 1, 0))) if Acsr.nnz else np.zeros(m, np.int64)
        ones = (nnz_row > 1) & (np.where(nnz_row > 0, not_one, 1) == 0)
        lazy = np.nonzero(ones & np.isneginf(lo) & np.isfinite(hi))[0]
        if len(lazy):
            keep = np.setdiff1d(np.arange(m), lazy)
            nc_k = _column_components(Acsr[keep], n)[0] if len(keep) else n
            if nc_k > 1:
                _IN_LAZY[0] = True
                try:
                    got = _solve_canonical(obj, Acsr[keep] if len(keep) else None, lo[keep], hi[keep], lb, ub)
                finally:
                    _IN_LAZY[0] = False
                if got is not None and np.all(Acsr[lazy] @ got[0] <= hi[lazy] + 1e-6):
                    return got
    if m:
        ncomp, lab = _column_components(A, n)
    else:
        ncomp, lab = n, np.arange(n)
    x = np.zeros(n)
    memo = {}  # identical components (same rows, bounds and — to 12 digits — normalised costs) are solved once
    order = np.argsort(lab, kind="stable")
    starts = np.searchsorted(lab[order], np.arange(ncomp + 1))
    Acsc = A.tocsc() if m else None
    for k in range(ncomp):
        cols = order[starts[k]:starts[k + 1]]
        if m:
            sub = Acsc[:, cols]
            rows = np.unique(sub.indices)
            Ak = sub.tocsr()[rows]
            lok, hik = lo[rows], hi[rows]
        else:
            Ak, lok, hik = None, None, None
        c = obj[cols]
        cmax = np.abs(c).max()
        cs = c * (1e4 / cmax) if cmax > 0 else c  # O(1e4) costs: keeps HiGHS' absolute gaps (1e-6) far below real differences
        key = None
        if Ak is not None:
            Ak.sort_indices()
            key = (np.round(cs, 8).tobytes(), Ak.indptr.tobytes(), Ak.indices.tobytes(), Ak.data.tobytes(), lok.tobytes(), hik.tobytes(), lb[cols].tobytes(), ub[cols].tobytes())
            if key in memo:
                x[cols] = memo[key]
                continue
        xk, status = _highs(cs, Ak, lok, hik, lb[cols], ub[cols])
        if xk is None or status != 0:
            return None
        z = float(np.dot(cs, xk))
        if cmax > 0:
            cut = csr_matrix(cs.reshape(1, -1))
            Ak2 = vstack([Ak, cut]).tocsr() if Ak is not None and Ak.shape[0] else cut
            lo2 = np.append(lok, z - 1e-9 * abs(z)) if lok is not None else np.array([z - 1e-9 * abs(z)])
            hi2 = np.append(hik, np.inf) if hik is not None else np.array([np.inf])
        else:
            Ak2, lo2, hi2 = Ak, lok, hik
        l, u = lb[cols].copy(), ub[cols].copy()
        for j in range(len(cols) - 1, -1, -1):  # last column first
            if xk[j] <= l[j]:
                l[j] = u[j] = xk[j]
                continue
            e = np.zeros(len(cols))
            e[j] = -1.0  # minimise x_j
            xj, status = _highs(e, Ak2, lo2, hi2, l, u)
            if xj is None or status != 0:
                v = xk[j]
            else:
                v, xk = xj[j], xj
            l[j] = u[j] = v
        x[cols] = xk
        if key is not None:
            memo[key] = xk
    return x, float(np.dot(obj, x)), True


class Oracle:
    """One oracle context (== one tako `Core`'s scheduler config)."""

    def __init__(self, config: Optional[abi.Config] = None, canonical: bool = False, reference_solver_options: bool = False):
        self.cfg = config or abi.make_config()
        self.canonical = canonical
        self.reference_solver_options = reference_solver_options
        self._ctx = lib().oracle_create(C.byref(self.cfg))
        self.solver_s = 0.0
        self.last_models = []
        self._cb = SOLVE_FN(self._solve_cb)

    def __del__(self):
        try:
            if self._ctx:
                lib().oracle_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def _solve_cb(self, user, ncols, obj, kind, nrows, rtype, rhs, roff, rcol, rcoef, tlimit, x_out, obj_out, is_opt):
        t0 = time.perf_counter()
        try:
            o = np.ctypeslib.as_array(obj, shape=(ncols,)) if ncols else np.zeros(0)
            k = np.ctypeslib.as_array(kind, shape=(ncols,)) if ncols else np.zeros(0, np.uint8)
            if nrows:
                rt = np.ctypeslib.as_array(rtype, shape=(nrows,))
                rh = np.ctypeslib.as_array(rhs, shape=(nrows,))
                ro = np.ctypeslib.as_array(roff, shape=(nrows + 1,))
                nnz = int(ro[-1])
                rc = np.ctypeslib.as_array(rcol, shape=(nnz,)) if nnz else np.zeros(0, np.int32)
                rf = np.ctypeslib.as_array(rcoef, shape=(nnz,)) if nnz else np.zeros(0)
            else:
                rt, rh, ro, rc, rf = np.zeros(0, np.uint8), np.zeros(0), np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0)
            global _PRESOLVE_ON
            _PRESOLVE_ON = self.reference_solver_options
            try:
                r = solve_milp(o, k, rt, rh, ro, rc, rf, tlimit, canonical=self.canonical)
            finally:
                _PRESOLVE_ON = False
            if r is None:
                return 0
            x, z, opt = r
            for i in range(ncols):
                x_out[i] = float(x[i])
            obj_out[0] = z
            is_opt[0] = 1 if opt else 0
            return 1
        except Exception as e:  # never let an exception cross the C boundary
            sys.stderr.write(f"[oracle] solver callback failed: {e!r}\n")
            return 0
        finally:
            self.solver_s += time.perf_counter() - t0

    def error(self) -> str:
        return lib().oracle_last_error(self._ctx).decode()

    def batches(self, snap: abi.Snapshot):
        sc, rc = snap.to_c(), abi.ResultC()
        rcode = lib().oracle_batches(self._ctx, C.byref(sc), C.byref(rc))
        if rcode < 0:
            raise RuntimeError(f"oracle_batches failed: {rcode} {self.error()}")
        return abi.parse_batches(rc)

    def tick(self, snap: abi.Snapshot) -> abi.Result:
        sc, rc = snap.to_c(), abi.ResultC()
        rcode = lib().oracle_tick(self._ctx, C.byref(sc), self._cb, None, C.byref(rc))
        if rcode < 0:
            raise RuntimeError(f"oracle_tick failed: {rcode} {self.error()}")
        return abi.parse_result(rc, len(snap.worker_id), snap.n_resources)

    def tick_given(self, snap: abi.Snapshot, counts, is_optimal: bool = True) -> abi.Result:
        """Parity tier T3: the reference's tick with the MILP's answer GIVEN — `counts` = (rq, variant, worker index, count) tuples, e.g. the
        product's — so that decode, create_task_mapping (scheduler/mapping.rs:23-234), proactive filling and the record order are the reference's on
        exactly the product's placement.  The model is still built (column mapping; `last_model()["x"]` holds the given point)."""
        cs = [(int(a), int(b), int(c), int(d)) for a, b, c, d in counts if d]
        rq = np.ascontiguousarray([c[0] for c in cs], np.uint32); v = np.ascontiguousarray([c[1] for c in cs], np.uint8)
        w = np.ascontiguousarray([c[2] for c in cs], np.uint32); val = np.ascontiguousarray([c[3] for c in cs], np.uint32)
        z32, z8 = np.zeros(1, np.uint32), np.zeros(1, np.uint8)
        pick = lambda a, z: a if len(a) else z
        lib().oracle_set_given_counts(self._ctx, len(cs), pick(rq, z32).ctypes.data_as(abi.u32p), pick(v, z8).ctypes.data_as(abi.u8p), pick(w, z32).ctypes.data_as(abi.u32p),
                                      pick(val, z32).ctypes.data_as(abi.u32p), 1 if is_optimal else 0)
        return self.tick(snap)

    def query(self, snap: abi.Snapshot, fake_ids, fake_total, fake_remaining=None, fake_min_util=None):
        sc = snap.to_c()
        n = len(fake_ids)
        ids = np.ascontiguousarray(fake_ids, np.uint32)
        tot = np.ascontiguousarray(np.asarray(fake_total, np.uint64).reshape(-1))
        rem = np.ascontiguousarray(fake_remaining if fake_remaining is not None else np.full(n, abi.HQ_NO_TIME_LIMIT), np.int64)
        mu = np.ascontiguousarray(fake_min_util if fake_min_util is not None else np.zeros(n), np.float32)
        q = abi.QueryWorkersC(n, ids.ctypes.data_as(abi.u32p), tot.ctypes.data_as(abi.u64p), rem.ctypes.data_as(abi.i64p), mu.ctypes.data_as(abi.f32p))
        out = abi.QueryResultC()
        rcode = lib().oracle_query(self._ctx, C.byref(sc), C.byref(q), self._cb, None, C.byref(out))
        if rcode < 0:
            raise RuntimeError(f"oracle_query failed: {rcode} {self.error()}")
        return abi._np(out.is_loaded, n, np.uint8).astype(bool), bool(out.is_optimal)

    def gap(self, snap: abi.Snapshot, high_rq: int, low_rq: int, worker_index: int) -> int:
        sc = snap.to_c()
        return lib().oracle_gap(self._ctx, C.byref(sc), high_rq, low_rq, worker_index, self._cb, None)

    def last_model(self) -> dict:
        v = ModelView()
        lib().oracle_last_model(self._ctx, C.byref(v))
        n, m = v.ncols, v.nrows
        g = abi._np
        roff = g(v.row_off, m + 1, np.int32) if m else np.zeros(1, np.int32)
        nnz = int(roff[-1])
        return dict(
            obj=g(v.obj, n, np.float64), kind=g(v.col_kind, n, np.uint8), ctype=g(v.col_type, n, np.uint8),
            cworker=g(v.col_worker, n, np.uint32), crq=g(v.col_rq, n, np.uint32), cvariant=g(v.col_variant, n, np.uint8),
            rtype=g(v.row_type, m, np.uint8), rhs=g(v.rhs, m, np.float64), roff=roff,
            rcol=g(v.row_col, nnz, np.int32), rcoef=g(v.row_coef, nnz, np.float64),
            x=g(v.x, n, np.float64), objective=v.objective,
        )

    def stage_times_us(self) -> dict:
        t = (C.c_double * 5)()
        lib().oracle_stage_times(self._ctx, t)
        return dict(load=t[0], batches=t[1], model=t[2], solve=t[3], mapping=t[4])


def prune_progressive(n: int, prefix: int, limit: int):
    out = np.zeros(max(n, limit), np.uint32)
    k = lib().oracle_prune_progressive(n, prefix, limit, out.ctypes.data_as(abi.u32p))
    return out[:k].tolist()


def hb_order_u32(keys):
    k = np.ascontiguousarray(keys, np.uint32)
    out = np.zeros(len(k), np.uint32)
    lib().oracle_hb_order_u32(k.ctypes.data_as(abi.u32p), len(k), out.ctypes.data_as(abi.u32p))
    return out.tolist()


def hb_order_taskid(keys):
    k = np.ascontiguousarray(keys, np.uint64)
    out = np.zeros(len(k), np.uint64)
    lib().oracle_hb_order_taskid(k.ctypes.data_as(abi.u64p), len(k), out.ctypes.data_as(abi.u64p))
    return out.tolist()
