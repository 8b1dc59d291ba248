"""The C-ABI library loads on a CPU-only host, exports every symbol include/bhgpu.h declares, has
the struct layouts the ctypes binding assumes, and fails loudly (no CPU fallback) without a GPU.
No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from gpu_nbody_simulation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bhgpu.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bh_[a-z_0-9]+)\s*\(", src)))


def test_library_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m gpu_nbody_simulation_amd.build`"
    assert os.path.dirname(_lib.LIB_PATH).startswith(ROOT)


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in bhgpu.h but not exported by libbhgpu.so"
        assert n in _lib.SIGNATURES, f"{n} declared in bhgpu.h but missing from the ctypes binding"
    for n in _lib.SIGNATURES:
        assert n in names, f"binding has {n}, header does not"


def test_abi_version():
    assert _lib.load().bh_abi_version() == _lib.ABI_VERSION == 4


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof/offsetof as gcc sees include/bhgpu.h == what ctypes computes for the binding."""
    prog = tmp_path / "layout.c"
    fields_cfg = [f[0] for f in _lib.bh_config._fields_]
    fields_st = [f[0] for f in _lib.bh_stats_t._fields_]
    fields_oc = [f[0] for f in _lib.bh_orb_cuts._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){",
             'printf("%zu %zu %zu %zu\\n", sizeof(bh_config), sizeof(bh_tree_node), sizeof(bh_stats_t), sizeof(bh_orb_cuts));']
    for f in fields_cfg:
        lines.append(f'printf("%zu\\n", offsetof(bh_config, {f}));')
    for f in fields_st:
        lines.append(f'printf("%zu\\n", offsetof(bh_stats_t, {f}));')
    for f in fields_oc:
        lines.append(f'printf("%zu\\n", offsetof(bh_orb_cuts, {f}));')
    lines.append("return 0;}")
    prog.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(prog)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes, offs = list(map(int, out[:4])), list(map(int, out[4:]))
    assert sizes == [C.sizeof(_lib.bh_config), C.sizeof(_lib.bh_tree_node), C.sizeof(_lib.bh_stats_t),
                     C.sizeof(_lib.bh_orb_cuts)]
    assert sizes[1] == 96                                     # the reference's 12-double Quadrant
    want = [getattr(_lib.bh_config, f).offset for f in fields_cfg] + \
           [getattr(_lib.bh_stats_t, f).offset for f in fields_st] + \
           [getattr(_lib.bh_orb_cuts, f).offset for f in fields_oc]
    assert offs == want
    assert _lib.ORB_BINS == 4096 and _lib.ORB_MAX_CUTS == 63   # BH_ORB_BINS / BH_ORB_MAX_CUTS


def test_header_is_plain_c():
    """The boundary must be bindable from C: compile the header alone as C11, no warnings."""
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", HEADER])


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from gpu_nbody_simulation_amd import BarnesHutEngine, BhConfig, BhError
    with pytest.raises(BhError) as ei:
        BarnesHutEngine(BhConfig(capacity=16))
    assert ei.value.code == -3 and "no CPU fallback" in str(ei.value)


def test_create_rejects_bad_arguments_before_touching_the_device():
    lib = _lib.load()
    h = C.c_void_p()
    for kw in ({"max_depth": 0}, {"max_depth": 33}, {"theta": 0.0}, {"precision": 7}, {"capacity": -1}):
        base = dict(capacity=8, theta=0.5, G=6.67e-11, dt=1.0, max_depth=10, precision=0,
                    reference_compat=1, device=0, n_threads=0, flags=0, node_capacity=0)
        base.update(kw)
        cfg = _lib.bh_config(**base)
        assert lib.bh_create(C.byref(cfg), C.byref(h)) == -1, kw
        assert lib.bh_last_error(None)
    assert lib.bh_create(None, C.byref(h)) == -1
