"""ORACLE (test infrastructure only): ctypes front end of oracle/topk_oracle.c plus a pure-numpy
float64 restatement used to cross-check it.  See topk_oracle.c for the reference citations."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtopk_oracle.so")


def build() -> str:
    src = os.path.join(_HERE, "topk_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libtopk_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def topk(db: np.ndarray, queries: np.ndarray, k: int, metric: str = "l2", group=None, exclude=None, mode: str = "f32chain",
         postfilter: bool = False):
    """returns (rows int32 [Q, k], dist float64 [Q, k]); mode 'f32chain' (bit-comparable with the HIP scan kernel: fewer than 16 queries per
    call), 'f32mfma' (bit-comparable with the HIP fan-out kernel: 16 or more queries per call; metric l2: selection by the expansion, then the 16
    nearest candidates scored again in the f32chain form), 'f32mfma_raw' (the expansion alone: what round 5's kernel returned) or 'f64'.
    `postfilter`: lancedb's `where(..., prefilter=False)` order (topk_oracle.c header)."""
    lib = ctypes.CDLL(build())
    db = np.ascontiguousarray(db, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    n, d = db.shape
    q = queries.shape[0]
    rows = np.empty((q, k), dtype=np.int32)
    dist = np.empty((q, k), dtype=np.float64)
    gp = ep = None
    if exclude is not None:
        group = np.ascontiguousarray(group, dtype=np.int32)
        exclude = np.ascontiguousarray(exclude, dtype=np.int32)
        gp, ep = group.ctypes.data_as(ctypes.c_void_p), exclude.ctypes.data_as(ctypes.c_void_p)
    lib.topk_oracle.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = lib.topk_oracle(db.ctypes.data_as(ctypes.c_void_p), gp, n, d, queries.ctypes.data_as(ctypes.c_void_p), ep, q, k,
                         {"l2": 0, "dot": 1}[metric], {"f32chain": 0, "f64": 1, "f32mfma": 2, "f32mfma_raw": 3}[mode], int(bool(postfilter)), rows.ctypes.data_as(ctypes.c_void_p),
                         dist.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"topk_oracle rc={rc}")
    return rows, dist


def topk_numpy(db: np.ndarray, queries: np.ndarray, k: int, metric: str = "l2", group=None, exclude=None, postfilter: bool = False):
    """float64 numpy restatement (small cases): stable argsort on (dist, row)."""
    if postfilter and exclude is not None:
        rows, d = topk_numpy(db, queries, k, metric)                 # the k nearest, unfiltered ...
        out_r, out_d = np.full_like(rows, -1), np.full_like(d, np.inf)
        for qi in range(rows.shape[0]):                              # ... then the filter; survivors move up
            keep = [j for j in range(k) if rows[qi, j] >= 0 and np.asarray(group)[rows[qi, j]] != np.asarray(exclude)[qi]]
            out_r[qi, :len(keep)], out_d[qi, :len(keep)] = rows[qi, keep], d[qi, keep]
        return out_r, out_d
    db64, q64 = db.astype(np.float64), queries.astype(np.float64)
    if metric == "l2":
        dist = ((q64[:, None, :] - db64[None, :, :]) ** 2).sum(-1)
    else:
        dist = 1.0 - q64 @ db64.T
    if exclude is not None:
        dist = np.where(np.asarray(group)[None, :] == np.asarray(exclude)[:, None], np.inf, dist)
    order = np.argsort(dist, axis=1, kind="stable")[:, :k]
    d = np.take_along_axis(dist, order, axis=1)
    rows = np.where(np.isinf(d), -1, order).astype(np.int32)
    return rows, d
