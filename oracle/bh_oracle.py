"""ctypes wrapper around oracle/libbh_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package never does (tests/test_no_oracle_in_product.py enforces it).

Parity status: pinned against the reference's own compiled code through the fixtures in
tests/golden/ (see oracle/bh_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbh_oracle.so")

NODE_DTYPE = np.dtype(
    [
        ("child", "<f8", (4,)),
        ("comx", "<f8"),
        ("comy", "<f8"),
        ("mass", "<f8"),
        ("xmin", "<f8"),
        ("xmax", "<f8"),
        ("ymin", "<f8"),
        ("ymax", "<f8"),
        ("particle", "<f8"),
    ]
)
assert NODE_DTYPE.itemsize == 96


class _WalkStats(C.Structure):
    _fields_ = [("visits", C.c_uint64), ("interactions", C.c_uint64), ("max_stack", C.c_int32)]


@dataclass
class WalkStats:
    visits: int
    interactions: int
    max_stack: int


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "bh_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libbh_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        vp = C.c_void_p
        L.bho_root_bounds.argtypes = [dp, C.c_int64, dp]
        L.bho_root_bounds.restype = None
        L.bho_build_tree.argtypes = [dp, dp, C.c_int64, C.c_int, vp, C.c_int64]
        L.bho_build_tree.restype = C.c_int64
        L.bho_compute_forces_range.argtypes = [vp, dp, dp, C.c_int64, C.c_int64, C.c_double,
                                               C.c_double, C.c_int, dp, C.POINTER(_WalkStats)]
        L.bho_compute_forces_range.restype = None
        L.bho_compute_forces_diag.argtypes = [vp, dp, dp, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int,
                                              C.c_int, C.c_int, dp, C.POINTER(C.c_uint32), dp, dp, dp, dp]
        L.bho_compute_forces_diag.restype = None
        L.bho_direct_forces.argtypes = [dp, dp, C.c_int64, C.c_double, dp]
        L.bho_direct_forces.restype = None
        L.bho_integrate.argtypes = [dp, dp, C.c_int64, C.c_double, dp, dp, dp]
        L.bho_integrate.restype = None
        L.bho_run.argtypes = [dp, dp, dp, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double,
                              C.c_double, C.c_int]
        L.bho_run.restype = C.c_int
        L.bho_export_preorder.argtypes = [vp, C.c_int64, vp, C.POINTER(C.c_int32)]
        L.bho_export_preorder.restype = C.c_int64
        L.bho_write_tree_text.argtypes = [vp, C.c_int64, dp, C.c_char_p]
        L.bho_write_tree_text.restype = C.c_int64
        L.bho_group_union_visits.argtypes = [vp, C.c_int64, dp, C.POINTER(C.c_int64), C.c_int64,
                                             C.c_int, C.c_double]
        L.bho_group_union_visits.restype = C.c_uint64
        _lib = L
    return _lib


def _d(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def root_bounds(pos) -> np.ndarray:
    pos = _f64(pos)
    out = np.empty(4)
    lib().bho_root_bounds(_d(pos), pos.shape[0], _d(out))
    return out


def node_capacity(n: int, max_depth: int) -> int:
    levels = max_depth if max_depth > 0 else 64
    cap = 1 + 4 * n * min(levels, 64)
    if 0 < max_depth < 15:
        cap = min(cap, (4 ** max_depth - 1) // 3)
    return max(1, min(cap, 16 * n + 1024))


def build_tree(pos, mass, max_depth: int = 10) -> np.ndarray:
    """Reference-order node array (NODE_DTYPE).  max_depth<=0: uncapped (ma2)."""
    pos, mass = _f64(pos), _f64(mass)
    n = pos.shape[0]
    cap = node_capacity(n, max_depth)
    nodes = np.zeros(cap, dtype=NODE_DTYPE)
    cnt = lib().bho_build_tree(_d(pos), _d(mass), n, max_depth, nodes.ctypes.data, cap)
    if cnt < 0:
        raise RuntimeError(f"bho_build_tree failed: {cnt}")
    return nodes[:cnt].copy()


def compute_forces(nodes, pos, mass, theta=0.5, G=6.67e-11, compat_self_skip=True, lo=0, hi=None,
                   with_stats=False):
    pos, mass = _f64(pos), _f64(mass)
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    n = pos.shape[0]
    hi = n if hi is None else hi
    f = np.zeros((n, 2))
    st = _WalkStats()
    lib().bho_compute_forces_range(nodes.ctypes.data, _d(pos), _d(mass), lo, hi, theta, G,
                                   1 if compat_self_skip else 0, _d(f), C.byref(st))
    if with_stats:
        return f, WalkStats(st.visits, st.interactions, st.max_stack)
    return f


@dataclass
class WalkDiag:
    """Per-body diagnostics of the oracle walk (bho_compute_forces_diag); rows outside [lo, hi) are zero."""
    forces: np.ndarray       # [n, 2], bit-identical to compute_forces
    counts: np.ndarray       # accepted force evaluations per body
    abs_sum: np.ndarray      # sum of |F_j| over accepted nodes
    coord: np.ndarray        # sum of |F_j| * (|comx| + |comy| (+ |px| + |py|)) / d_j
    flip: np.ndarray         # total multipole error of the borderline cells (0: the node set is unambiguous in fp32)
    cap: np.ndarray          # cap_depth > 0: what summing the depth-cap cells body by body changed (magnitude)


def compute_forces_diag(nodes, pos, mass, theta=0.5, G=6.67e-11, compat_self_skip=True, lo=0, hi=None,
                        pos_rounded=False, threads=0, cap_depth=0) -> WalkDiag:
    """cap_depth > 0 (uncapped tree): cells at that depth are summed body by body like the device's depth-cap
    buckets (forces and counts then follow the device's documented deviation, `cap` says by how much).  threads > 1: the body range is split over that many Python threads (ctypes releases the GIL; bodies are
    independent, the results do not depend on the split)."""
    pos, mass = _f64(pos), _f64(mass)
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    n = pos.shape[0]
    hi = n if hi is None else hi
    f = np.zeros((n, 2))
    cnt = np.zeros(n, dtype=np.uint32)
    asum, coord, flip, cap = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    L = lib()

    def run(a, b):
        L.bho_compute_forces_diag(nodes.ctypes.data, _d(pos), _d(mass), a, b, theta, G, 1 if compat_self_skip else 0,
                                  1 if pos_rounded else 0, cap_depth, _d(f), cnt.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  _d(asum), _d(coord), _d(flip), _d(cap))
    threads = threads or min(16, os.cpu_count() or 1)
    if threads <= 1 or hi - lo < 4096:
        run(lo, hi)
    else:
        import threading
        cuts = [lo + (hi - lo) * k // (4 * threads) for k in range(4 * threads + 1)]
        todo = list(zip(cuts[:-1], cuts[1:]))
        lock = threading.Lock()

        def work():
            while True:
                with lock:
                    if not todo:
                        return
                    a, b = todo.pop()
                run(a, b)
        ts = [threading.Thread(target=work) for _ in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    return WalkDiag(f, cnt, asum, coord, flip, cap)


def direct_forces(pos, mass, G=6.67e-11) -> np.ndarray:
    pos, mass = _f64(pos), _f64(mass)
    f = np.zeros_like(pos)
    lib().bho_direct_forces(_d(pos), _d(mass), pos.shape[0], G, _d(f))
    return f


def integrate(forces, mass, vel, pos, dt=1.0):
    """In-place on copies; returns (acc, vel, pos)."""
    forces, mass = _f64(forces), _f64(mass)
    vel, pos = _f64(vel).copy(), _f64(pos).copy()
    acc = np.zeros_like(pos)
    lib().bho_integrate(_d(forces), _d(mass), pos.shape[0], dt, _d(acc), _d(vel), _d(pos))
    return acc, vel, pos


def run(pos, vel, mass, nsteps, max_depth=10, theta=0.5, G=6.67e-11, dt=1.0, direct=False):
    pos, vel, mass = _f64(pos).copy(), _f64(vel).copy(), _f64(mass)
    rc = lib().bho_run(_d(pos), _d(vel), _d(mass), pos.shape[0], nsteps, max_depth, theta, G, dt,
                       1 if direct else 0)
    if rc != 0:
        raise RuntimeError(f"bho_run failed: {rc}")
    return pos, vel


def export_preorder(nodes):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    out = np.zeros(len(nodes), dtype=NODE_DTYPE)
    depth = np.zeros(len(nodes), dtype=np.int32)
    k = lib().bho_export_preorder(nodes.ctypes.data, len(nodes), out.ctypes.data,
                                  depth.ctypes.data_as(C.POINTER(C.c_int32)))
    return out[:k], depth[:k]


def write_tree_text(nodes, pos, path: str) -> int:
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    pos = _f64(pos)
    k = lib().bho_write_tree_text(nodes.ctypes.data, len(nodes), _d(pos), os.fsencode(path))
    if k < 0:
        raise OSError(f"cannot write {path}")
    return k


def group_union_visits(nodes, pos, order, group=64, theta=0.5) -> int:
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    pos = _f64(pos)
    order = np.ascontiguousarray(order, dtype=np.int64)
    return int(lib().bho_group_union_visits(nodes.ctypes.data, len(nodes), _d(pos),
                                            order.ctypes.data_as(C.POINTER(C.c_int64)),
                                            len(order), group, theta))


def canonical_tree(nodes):
    """(preorder nodes with child indices blanked to 'has child' flags, depth) -- equal for any
    two trees with the same topology and contents, whatever their node numbering."""
    out, depth = export_preorder(nodes)
    out = out.copy()
    out["child"] = np.where(out["child"] == -1, -1.0, 1.0)
    return out, depth
