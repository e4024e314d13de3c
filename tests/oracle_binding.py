"""ctypes binding of oracle/liboracle.so -- the CPU restatement used as the CHECKER in tests.

Test infrastructure only: the product never imports this module.
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
abi = importlib.import_module("uv-slam_amd.abi")


def _build():
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("uvs_oracle.cpp", "oracle_factors.h", "oracle_math.h", "oracle_marg.h")] + [os.path.join(ROOT, "include", "uvs_solver.h")]
    if (not os.path.exists(so)) or any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        if all(os.path.exists(s) for s in srcs):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return so


class Oracle:
    def __init__(self, lib_path=None):
        """lib_path: another build of the same sources (bench.py's cpu_baseline leg times an -O3 -march=native build made on the box)."""
        self.lib = C.CDLL(lib_path or _build())
        L = self.lib
        L.oracle_solve.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowC), C.c_int, C.POINTER(abi.StateC), C.POINTER(abi.Report)]
        L.oracle_solve.restype = C.c_int
        L.oracle_evaluate.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowC), C.c_int, C.POINTER(abi.EvalC)]
        L.oracle_evaluate.restype = C.c_int
        L.oracle_marginalize.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowC), C.c_int, C.POINTER(abi.Prior)]
        L.oracle_marginalize.restype = C.c_int
        L.oracle_imu_sqrt_info.argtypes = [abi.c_double_p, abi.c_double_p]
        L.oracle_pose_plus.argtypes = [abi.c_double_p, abi.c_double_p, abi.c_double_p]
        L.oracle_sym_eig.argtypes = [C.c_int, abi.c_double_p, abi.c_double_p, abi.c_double_p]

    def solve(self, w, opts=None, linear_mode=0):
        opts = opts or abi.default_options()
        wc, keep = w.to_c()
        st = abi.State(len(w.inv_depth), len(w.line_orth))
        sc = st.alloc_c()
        rep = abi.Report()
        rc = self.lib.oracle_solve(C.byref(opts), C.byref(wc), linear_mode, C.byref(sc), C.byref(rep))
        assert rc in (abi.UVS_OK, abi.UVS_ERR_NUMERIC), rc
        st.from_c(sc)
        return st, rep

    def prepare(self, w, opts=None):
        """Everything of solve() that runs under the GIL (struct conversion, output buffers), so that a thread pool times only the C call."""
        opts = opts or abi.default_options()
        wc, keep = w.to_c()
        st = abi.State(len(w.inv_depth), len(w.line_orth))
        return (opts, wc, keep, st, st.alloc_c(), abi.Report())

    def solve_prepared(self, job):
        opts, wc, keep, st, sc, rep = job
        return self.lib.oracle_solve(C.byref(opts), C.byref(wc), 0, C.byref(sc), C.byref(rep))      # ctypes releases the GIL for the call

    def evaluate(self, w, robust=True, opts=None):
        opts = opts or abi.default_options()
        wc, keep = w.to_c()
        ev = abi.Eval(w)
        ec = ev.alloc_c()
        rc = self.lib.oracle_evaluate(C.byref(opts), C.byref(wc), int(robust), C.byref(ec))
        assert rc == 0, rc
        ev.cost = ec.cost
        return ev

    def marginalize(self, w, flag=0, opts=None):
        opts = opts or abi.default_options()
        wc, keep = w.to_c()
        p = abi.Prior()
        rc = self.lib.oracle_marginalize(C.byref(opts), C.byref(wc), flag, C.byref(p))
        assert rc == 0, rc
        return p

    def imu_sqrt_info(self, cov):
        cov = np.ascontiguousarray(cov, float).reshape(225)
        W = np.zeros(225)
        rc = self.lib.oracle_imu_sqrt_info(cov.ctypes.data_as(abi.c_double_p), W.ctypes.data_as(abi.c_double_p))
        assert rc == 0
        return W.reshape(15, 15)

    def pose_plus(self, x, d):
        x = np.ascontiguousarray(x, float); d = np.ascontiguousarray(d, float); out = np.zeros(7)
        self.lib.oracle_pose_plus(x.ctypes.data_as(abi.c_double_p), d.ctypes.data_as(abi.c_double_p), out.ctypes.data_as(abi.c_double_p))
        return out

    def sym_eig(self, A):
        A = np.ascontiguousarray(A, float); n = A.shape[0]
        V = np.zeros((n, n)); ev = np.zeros(n)
        self.lib.oracle_sym_eig(n, A.ctypes.data_as(abi.c_double_p), V.ctypes.data_as(abi.c_double_p), ev.ctypes.data_as(abi.c_double_p))
        return ev, V
