"""TEST INFRASTRUCTURE — ctypes loader of oracle/libt2l_oracle.so (the plain-C restatement)."""
import ctypes as C
import os
import os.path as osp
import subprocess

import numpy as np

_HERE = osp.dirname(osp.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = osp.join(_HERE, "libt2l_oracle.so")
        if not osp.exists(so):
            subprocess.run(["make", "-C", _HERE], check=True)
        _lib = C.CDLL(so)
        _lib.t2l_oracle_retrieve.restype = C.c_int
        _lib.t2l_oracle_retrieve.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                             C.c_void_p, C.c_void_p]
        _lib.t2l_oracle_contrastive.restype = C.c_double
        _lib.t2l_oracle_contrastive.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_void_p,
                                                C.c_void_p]
    return _lib


def retrieve_topk(cells, queries, k):
    cells = np.ascontiguousarray(cells, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    n, d = cells.shape
    q = queries.shape[0]
    k = min(k, n)
    idx = np.zeros((q, k), dtype=np.int64)
    sc = np.zeros((q, k), dtype=np.float64)
    fn = lib().t2l_oracle_retrieve
    n_thr = min(os.cpu_count() or 1, 32, max(1, q * n // 2_000_000))  # queries are independent: big checks use the host's cores
    if n_thr <= 1:
        assert fn(cells.ctypes.data, n, queries.ctypes.data, q, d, k, idx.ctypes.data, sc.ctypes.data) == 0
        return idx, sc
    from concurrent.futures import ThreadPoolExecutor

    bounds = np.linspace(0, q, n_thr + 1).astype(int)

    def part(i):  # (ctypes releases the GIL; every slice writes its own rows of idx / sc)
        lo, hi = int(bounds[i]), int(bounds[i + 1])
        if hi > lo:
            assert fn(cells.ctypes.data, n, queries[lo:hi].ctypes.data, hi - lo, d, k, idx[lo:hi].ctypes.data, sc[lo:hi].ctypes.data) == 0

    with ThreadPoolExecutor(n_thr) as ex:
        list(ex.map(part, range(n_thr)))
    return idx, sc


def contrastive_loss(im, s, temperature):
    im = np.ascontiguousarray(im, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    b, d = im.shape
    ga = np.zeros((b, d))
    gs = np.zeros((b, d))
    loss = lib().t2l_oracle_contrastive(im.ctypes.data, s.ctypes.data, b, d, float(temperature), ga.ctypes.data,
                                        gs.ctypes.data)
    return loss, ga, gs
