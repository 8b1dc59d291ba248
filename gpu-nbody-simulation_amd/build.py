"""Builds libbhgpu.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m gpu_nbody_simulation_amd.build

hipcc cross-compiles without a GPU.  Two translation units: bh_engine.hip (tree build, exact
walk, C-ABI) with -ffp-contract=off so fp64 results stay bit-identical to the reference, and
bh_walk_fast.hip (fp32 walk) with the default contraction.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbhgpu.so")
ARCH = "gfx950"

UNITS = [
    ("bh_engine.hip", ["-ffp-contract=off"]),
    ("bh_walk_fast.hip", []),
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; this library has no non-HIP build)")


def source_digest() -> str:
    """Digest of the device sources (csrc/ + the C-ABI header): what a committed profile is stamped with,
    so that bench.py can tell whether profiles/latest_walk_traffic.json was measured on THESE kernels
    (there is no .git on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h")) and os.path.isfile(os.path.join(CSRC, f))]
    for f in srcs + [os.path.join("..", "..", "include", "bhgpu.h")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h"))] + [
        os.path.join(HERE, "..", "include", "bhgpu.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force: bool = False, verbose: bool = False, variant: str | None = None, extra_flags=()) -> str:
    """variant: build an A/B variant of the library as build/libbhgpu_<variant>.so with extra_flags
    (scripts/ only: BHGPU_LIB selects it at load time; the product is libbhgpu.so)."""
    if variant:
        return _build_to(os.path.join(HERE, "build", f"libbhgpu_{variant}.so"), os.path.join(HERE, "build", variant),
                         list(extra_flags), verbose)
    if not force and not _stale():
        return LIB
    return _build_to(LIB, os.path.join(HERE, "build"), [], verbose)


def _build_to(lib: str, objdir: str, extra_flags, verbose: bool) -> str:
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)
    objs = []
    info = f'-DBHGPU_BUILD_INFO="digest={source_digest()} flags={" ".join(extra_flags) or "-"}"'
    for src, extra in UNITS:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall",
               "-Wno-unused-function", info, *extra, *extra_flags, *os.environ.get("BHGPU_EXTRA_FLAGS", "").split(),
               "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        objs.append(obj)
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:       # python -m gpu_nbody_simulation_amd.build --variant NAME -DFOO=1 ...
        i = sys.argv.index("--variant")
        print(build(variant=sys.argv[i + 1], extra_flags=sys.argv[i + 2:], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
