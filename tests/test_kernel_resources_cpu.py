"""Static resource checks of the hand-scheduled walk kernel (no GPU needed: hipcc cross-compiles).

The assembly traversal loop is designed around two numbers of its code object:
  * at most 80 SGPRs -- a wave's SGPR allocation is its count + 16 rounded up to 16 out of 800 per SIMD, so
    80 is the last value that leaves 8 resident waves per SIMD (measured: 106 -> 6 waves, 94 -> 7), and the
    walk is latency-bound: resident waves are what hides the latency;
  * no scratch: the fixed SGPR block of the loop includes s32, which the compiler reserves as the stack
    pointer of kernels that use private memory."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpu-nbody-simulation_amd", "csrc", "bh_walk_fast.hip")
ASM_KERNEL = "_ZN2bh16walk_fast_kernelILb0ELb0ELi0ELi1ELb1EEEvNS_12WalkFastArgsE"


def test_asm_walk_kernel_fits_eight_waves_and_uses_no_scratch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "walk.s"
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-w",
                           "-o", str(out), SRC], cwd=os.path.dirname(SRC))
    text = out.read_text()
    m = re.search(r"\.name:\s+" + re.escape(ASM_KERNEL) + r"\n(.*?)\n\s+-\s", text + "\n  - ", re.S)
    meta = text[text.index(".name:           " + ASM_KERNEL):][:3000] if m is None else m.group(0)
    val = lambda key: int(re.search(key + r":\s+(\d+)", meta).group(1))
    assert val(r"\.sgpr_count") <= 80, "more than 80 SGPRs: fewer than 8 resident waves per SIMD"
    assert val(r"\.vgpr_count") <= 64
    assert val(r"\.private_segment_fixed_size") == 0 and val(r"\.sgpr_spill_count") == 0 and val(r"\.vgpr_spill_count") == 0
    # the loop really is the hand-written one: its fixed registers and the two-quads-in-flight loads are there
    body = text[text.index(ASM_KERNEL + ":"):]
    body = body[:body.index("s_endpgm", body.index("Ldone_"))]
    assert "s_load_dwordx16 s[48:63]" in body and "s_load_dwordx16 s[24:39]" in body and "v_cmpx_lt_f32_e32" in body
