"""BarnesHutEngine: the Python face of one bh_ctx (include/bhgpu.h).

Mirrors what runSimulationGpu (project.cu:918-1024) does with its device buffers, with the
reference's compile-time constants (project.cu:1-11, 27-35, 60-62) as a runtime BhConfig.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass

import numpy as np

from . import _lib

TREE_NODE_DTYPE = np.dtype(
    [("child", "<f8", (4,)), ("comx", "<f8"), ("comy", "<f8"), ("mass", "<f8"), ("xmin", "<f8"),
     ("xmax", "<f8"), ("ymin", "<f8"), ("ymax", "<f8"), ("particle", "<f8")])

FLAG_WALK_STATS = 1 << 0
FLAG_LDS_STACK = 1 << 1
FLAG_WALK_NO_SPLIT = 1 << 2
FLAG_WALK_PORTABLE = 1 << 3


class Precision(enum.IntEnum):
    F64_EXACT = 0   # bit-identical to the reference CPU path
    F32 = 1         # throughput mode (BASELINE configs "fp32")
    MIXED = 2       # fp64 state, fp32 forces (BASELINE config "fp64 positions / fp32 forces")
    F64 = 3         # fp64 throughout at throughput: the exact mode's tree, a free-order rsqrt walk (<= 1e-12 of the oracle)


class BhError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"bhgpu error {code}: {msg}")
        self.code = code


@dataclass
class BhConfig:
    capacity: int
    theta: float = 0.5              # THETA, project.cu:60
    G: float = 6.67e-11             # project.cu:27
    dt: float = 1.0                 # DELTA_T, project.cu:29
    max_depth: int = 10             # QUADTREE_MAX_DEPTH, project.cu:61
    precision: Precision = Precision.F64_EXACT
    reference_compat: bool = True
    device: int = 0
    n_threads: int = 0              # N_THREADS, project.cu:5-7: bodies walked at a time (passes of whole workgroups); 0 = all
    flags: int = 0
    node_capacity: int = 0


@dataclass
class BhStats:
    n_bodies: int
    n_nodes: int
    n_internal: int
    steps_done: int
    visits: int
    interactions: int
    wave_nodes: int
    last_step_ms: float
    build_ms: float
    walk_ms: float
    device_bytes: int
    keys_ms: float = 0.0        # per kernel group of the last timed step
    sort_ms: float = 0.0
    scan_ms: float = 0.0
    nodes_ms: float = 0.0
    build_bytes: int = 0        # algorithmic bytes of that step
    walk_bytes: int = 0
    wave_quads: int = 0         # FLAG_WALK_STATS: quads loaded, once per wavefront
    sort_spill_buckets: int = 0  # bucket-sort buckets sorted through memory since creation (0 in steady motion)
    let_tree_ms: float = 0.0    # last let_build: global box + local tree
    let_pack_ms: float = 0.0    # last let_build: LET marking / numbering / packing
    sort_rerun_buckets: int = 0  # bucket-sort buckets whose short sort met a long run and was repeated in full (bh_sort.hpp)
    wave_accepts: int = 0       # FLAG_WALK_STATS, fp64 precisions: nodes some lane took a term from, once per wavefront
    walk_launches: int = 0      # walk kernel launches of the last step (1, or the passes of n_threads)


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class BarnesHutEngine:
    def __init__(self, cfg: BhConfig):
        self._lib = _lib.load()
        self.cfg = cfg
        c = _lib.bh_config(cfg.capacity, cfg.theta, cfg.G, cfg.dt, cfg.max_depth, int(cfg.precision),
                           1 if cfg.reference_compat else 0, cfg.device, cfg.n_threads, cfg.flags,
                           cfg.node_capacity)
        h = C.c_void_p()
        rc = self._lib.bh_create(C.byref(c), C.byref(h))
        if rc != 0:
            raise BhError(rc, (self._lib.bh_last_error(None) or b"").decode())
        self._h = h
        self.n = 0

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise BhError(rc, (self._lib.bh_last_error(self._h) or b"").decode())

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.bh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state ------------------------------------------------------------------------------
    def upload(self, positions, velocities, masses) -> None:
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 2)
        vel = np.ascontiguousarray(velocities, dtype=np.float64).reshape(-1, 2)
        m = np.ascontiguousarray(masses, dtype=np.float64).reshape(-1)
        if not (len(pos) == len(vel) == len(m)):
            raise ValueError("positions, velocities and masses must have the same length")
        self._check(self._lib.bh_upload(self._h, _dptr(pos), _dptr(vel), _dptr(m), len(m)))
        self.n = len(m)

    def download(self):
        pos = np.empty((self.n, 2))
        vel = np.empty((self.n, 2))
        self._check(self._lib.bh_download(self._h, _dptr(pos), _dptr(vel)))
        return pos, vel

    def initialize(self, n: int, seed: int = 0, kind: str = "box", lower_m=1e-1, higher_m=5e-1,
                   lower_p=-1e-1, higher_p=1e-1, lower_v=-1e-4, higher_v=1e-4) -> None:
        """On-device initial conditions (initializeGpu, project.cu:304-341; defaults = its
        constants, project.cu:30-35).  kind "plummer": scale lower_p, truncation higher_p, equal
        masses higher_m, zero velocities."""
        k = {"box": 0, "plummer": 1}[kind]
        self._check(self._lib.bh_initialize(self._h, n, seed, k, lower_m, higher_m, lower_p, higher_p,
                                            lower_v, higher_v))
        self.n = n

    def masses(self) -> np.ndarray:
        m = np.empty(self.n)
        self._check(self._lib.bh_download_masses(self._h, _dptr(m)))
        return m

    # -- hot path ---------------------------------------------------------------------------
    def step(self, nsteps: int = 1) -> None:
        self._check(self._lib.bh_step(self._h, nsteps))

    def sync(self) -> None:
        self._check(self._lib.bh_sync(self._h))

    def build_tree(self) -> None:
        self._check(self._lib.bh_build_tree(self._h))

    def compute_forces(self) -> np.ndarray:
        self._check(self._lib.bh_compute_forces(self._h))
        return self.forces()

    def forces(self) -> np.ndarray:
        f = np.empty((self.n, 2))
        self._check(self._lib.bh_get_forces(self._h, _dptr(f)))
        return f

    def accelerations(self) -> np.ndarray:
        a = np.empty((self.n, 2))
        self._check(self._lib.bh_get_accel(self._h, _dptr(a)))
        return a

    def interaction_counts(self) -> np.ndarray:
        """Accepted force evaluations per body of the last walk (FLAG_WALK_STATS; fp32 / mixed / Precision.F64), caller order."""
        c = np.zeros(max(self.n, 1), dtype=np.uint32)
        self._check(self._lib.bh_get_interaction_counts(self._h, c.ctypes.data_as(C.POINTER(C.c_uint32))))
        return c[:self.n]

    # -- tree output ------------------------------------------------------------------------
    def export_tree(self):
        """(nodes in DFS pre-order as TREE_NODE_DTYPE, depth)."""
        n = C.c_int64(0)
        rc = self._lib.bh_export_tree(self._h, None, None, 0, C.byref(n))
        if rc not in (0, -4):
            self._check(rc)
        nodes = np.zeros(max(n.value, 1), dtype=TREE_NODE_DTYPE)
        depth = np.zeros(max(n.value, 1), dtype=np.int32)
        self._check(self._lib.bh_export_tree(self._h, nodes.ctypes.data,
                                             depth.ctypes.data_as(C.POINTER(C.c_int32)), len(nodes),
                                             C.byref(n)))
        return nodes[: n.value], depth[: n.value]

    def write_quadtree_file(self, path: str) -> None:
        self._check(self._lib.bh_write_quadtree_file(self._h, os.fsencode(path)))

    # -- measurement ------------------------------------------------------------------------
    def stats(self) -> BhStats:
        s = _lib.bh_stats_t()
        self._check(self._lib.bh_stats(self._h, C.byref(s)))
        return BhStats(s.n_bodies, s.n_nodes, s.n_internal, s.steps_done, s.visits, s.interactions,
                       s.wave_nodes, s.last_step_ms, s.build_ms, s.walk_ms, s.device_bytes, s.keys_ms, s.sort_ms,
                       s.scan_ms, s.nodes_ms, s.build_bytes, s.walk_bytes, s.wave_quads, s.sort_spill_buckets,
                       s.let_tree_ms, s.let_pack_ms, s.sort_rerun_buckets, s.wave_accepts, s.walk_launches)

    def step_times(self):
        """(step_ms[k], walk_ms[k]) of the steps of the last step() call (at most 4,096): HIP events per step."""
        n = C.c_int32(0)
        self._check(self._lib.bh_step_times(self._h, None, None, 0, C.byref(n)))
        st, wk = np.zeros(max(n.value, 1)), np.zeros(max(n.value, 1))
        self._check(self._lib.bh_step_times(self._h, _dptr(st), _dptr(wk), n.value, C.byref(n)))
        return st[:n.value], wk[:n.value]

    @staticmethod
    def build_info() -> str:
        """What the loaded library was built from (bh_build_info) and whether it is the product library."""
        info = (_lib.load().bh_build_info() or b"").decode()
        return info + ("" if _lib.is_product_library() else f" VARIANT={os.path.basename(_lib.LIB_PATH)}")

    # -- multi-GPU plumbing -----------------------------------------------------------------
    def set_owned_fraction(self, rank: int, world: int) -> None:
        self._check(self._lib.bh_set_owned_fraction(self._h, rank, world))

    def owned_range(self):
        lo, hi = C.c_int64(), C.c_int64()
        self._check(self._lib.bh_owned_range(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def step_local(self) -> None:
        self._check(self._lib.bh_step_local(self._h))

    def scatter_sorted(self) -> None:
        self._check(self._lib.bh_scatter_sorted(self._h))

    def device_sorted(self):
        """Device pointer of the sorted-order exchange buffer: {x, y, vx, vy} float32 per sorted body."""
        p = C.c_void_p()
        self._check(self._lib.bh_device_sorted(self._h, C.byref(p)))
        return p.value

    def device_state(self):
        p, v, m = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n, eb = C.c_int64(), C.c_int32()
        self._check(self._lib.bh_device_state(self._h, C.byref(p), C.byref(v), C.byref(m), C.byref(n),
                                              C.byref(eb)))
        return p.value, v.value, m.value, n.value, eb.value

    # -- distributed step with locally-essential trees ----------------------------------------
    def let_local_quads(self) -> int:
        """Quads this context reserves for its own tree; the forest_base of let_configure must be the
        maximum of this over all ranks."""
        q = C.c_int64()
        self._check(self._lib.bh_let_local_quads(self._h, C.byref(q)))
        return q.value

    def let_configure(self, rank: int, world: int, let_cap: int, forest_base: int | None = None) -> None:
        """forest_base None: this context's own let_local_quads() -- only right when every rank's
        context has the same capacity."""
        if forest_base is None:
            forest_base = self.let_local_quads()
        self._check(self._lib.bh_let_configure(self._h, rank, world, let_cap, forest_base))
        self._let_world = world

    def let_bounds(self) -> None:
        self._check(self._lib.bh_let_bounds(self._h))

    def let_pointers(self):
        """(lbounds, all_bounds, send, recv, block_bytes, boxes_per_rank): device pointers for the two
        collectives; lbounds holds boxes_per_rank x 4 doubles, all_bounds world times that."""
        a, b, s, r = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nb, k = C.c_int64(), C.c_int32()
        self._check(self._lib.bh_let_pointers(self._h, C.byref(a), C.byref(b), C.byref(s), C.byref(r), C.byref(nb),
                                              C.byref(k)))
        return a.value, b.value, s.value, r.value, nb.value, k.value

    def let_build(self) -> None:
        self._check(self._lib.bh_let_build(self._h))

    def let_walk(self) -> None:
        self._check(self._lib.bh_let_walk(self._h))

    def let_forces(self) -> None:
        self._check(self._lib.bh_let_forces(self._h))

    def let_walk_local(self) -> None:
        self._check(self._lib.bh_let_walk_local(self._h))

    def let_walk_remote(self, integrate: bool = True) -> None:
        self._check(self._lib.bh_let_walk_remote(self._h, 1 if integrate else 0))

    def let_counts(self, with_overflow: bool = False):
        """Per peer, the largest LET of any let_build since the previous let_counts / let_configure (waits
        for the stream), and whether any of those builds overflowed -- a LET beyond let_cap, or a local
        tree beyond node_capacity.  Reading starts a new interval.  Raises BhError(-4) on overflow, unless
        with_overflow: then returns (counts, overflow flag)."""
        arr = (C.c_uint32 * self._let_world)()
        if with_overflow:
            ov = C.c_int32()
            self._check(self._lib.bh_let_counts(self._h, arr, C.byref(ov)))
            return list(arr), bool(ov.value)
        self._check(self._lib.bh_let_counts(self._h, arr, None))
        return list(arr)

    # -- device-side migration and re-balancing (LET scheme) ------------------------------------
    def set_ids(self, ids) -> None:
        a = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        if len(a) != self.n:
            raise ValueError("one id per body")
        self._check(self._lib.bh_set_ids(self._h, a.ctypes.data_as(C.POINTER(C.c_int64))))

    def ids(self) -> np.ndarray:
        a = np.empty(self.n, dtype=np.int64)
        self._check(self._lib.bh_get_ids(self._h, a.ctypes.data_as(C.POINTER(C.c_int64))))
        return a

    @staticmethod
    def _cuts_struct(cuts):
        c = _lib.bh_orb_cuts()
        c.world, c.n_cuts = cuts.world, cuts.world - 1
        for k in range(4):
            c.box[k] = float(cuts.box[k])
        for k in range(cuts.world - 1):
            c.axis[k], c.value[k] = int(cuts.axis[k]), float(cuts.value[k])
        return c

    def orb_histogram(self, cuts, level: int):
        """(device pointer, number of uint64 words) of the weighted histograms of the cut tree's regions
        at `level` (row k = the region whose cut is k, ORB_BINS bins across the root box)."""
        c = self._cuts_struct(cuts)
        p, nw = C.c_void_p(), C.c_int64()
        self._check(self._lib.bh_orb_histogram(self._h, C.byref(c), level, C.byref(p), C.byref(nw)))
        return p.value, nw.value

    def migrate_pack(self, cuts):
        """Classify the local bodies by the cut tree and group them by destination in the send buffer;
        returns the number of bodies for every rank (waits for the stream)."""
        c = self._cuts_struct(cuts)
        cnt = (C.c_int64 * cuts.world)()
        self._check(self._lib.bh_migrate_pack(self._h, C.byref(c), cnt))
        return list(cnt)

    def migrate_pointers(self):
        """(send, recv device pointers, capacity in records of 6 doubles)."""
        s, r, cap = C.c_void_p(), C.c_void_p(), C.c_int64()
        self._check(self._lib.bh_migrate_pointers(self._h, C.byref(s), C.byref(r), C.byref(cap)))
        return s.value, r.value, cap.value

    def migrate_unpack(self, n_new: int) -> None:
        self._check(self._lib.bh_migrate_unpack(self._h, n_new))
        self.n = n_new

    def set_stream(self, hip_stream: int) -> None:
        self._check(self._lib.bh_set_stream(self._h, C.c_void_p(hip_stream)))
