"""Static resource checks of the walk kernels (no GPU needed: hipcc cross-compiles).

The hand-scheduled traversal loops (walk_tree_asm, walk_list_asm) are designed around three properties of their
code objects; this file checks them for EVERY instantiation that contains the assembly:
  * no scratch and no spills.  The loops' fixed SGPR block includes s32, which the compiler reserves as the
    stack pointer of kernels that use private memory, and hipcc says so on every asm block ("inline asm clobber
    list contains reserved registers: s32, m0").  That note is benign exactly as long as the kernel has no
    private segment: without one nothing reads s32 as a stack pointer, it is an ordinary SGPR (and m0 is saved by
    nobody because nothing else in these kernels uses LDS-DMA / movrel / GWS).  So: private_segment_fixed_size
    == 0, no dynamic stack, zero SGPR and VGPR spills (a spilled SGPR sits in a lane of a VGPR the asm block
    could clobber) -- and the compile's only diagnostics are those known notes.
  * SGPR ceilings.  A wave's SGPR allocation is its count + 16 rounded up to 16 out of 800 per SIMD (measured:
    <= 80 -> 8 resident waves, <= 96 -> 7, more -> 6).  The one-wave-per-group loop is latency-bound on large
    launches and must keep 8 waves; the level-synchronous variants (SPLIT > 1) run on launches that do not fill
    the GPU (<= 3,072 groups x 4 waves), where 6 waves per SIMD hold the whole launch: their ceiling is the
    architectural one.
  * <= 64 VGPRs (8 waves per SIMD by the vector file)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpu-nbody-simulation_amd", "csrc", "bh_walk_fast.hip")
KERNEL = re.compile(r"_ZN2bh16walk_fast_kernelILb([01])ELb([01])ELi(\d+)ELb([01])EEEvNS_12WalkFastArgsE")   # <LDS_STACK, STATS, SPLIT, ASM>
ONE_WAVE_ASM = "_ZN2bh16walk_fast_kernelILb0ELb0ELi1ELb1EEEvNS_12WalkFastArgsE"


@pytest.fixture(scope="module")
def compiled(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("walk") / "walk.s"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wall",
                        "-Wno-unused-function", "-o", str(out), SRC], cwd=os.path.dirname(SRC), capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text(), r.stderr


def _kernels(text):
    """name -> metadata dict of every walk_fast_kernel instantiation in the code object's notes."""
    res = {}
    for m in re.finditer(r"\.name:\s+(_ZN2bh16walk_fast_kernel\S+)\n", text):
        meta = text[m.start():m.start() + 3000]
        val = lambda key: int(re.search(key + r":\s+(\d+)", meta).group(1))
        res[m.group(1)] = {
            "sgpr": val(r"\.sgpr_count"), "vgpr": val(r"\.vgpr_count"), "sgpr_spill": val(r"\.sgpr_spill_count"),
            "vgpr_spill": val(r"\.vgpr_spill_count"), "scratch": val(r"\.private_segment_fixed_size"),
            "dynamic_stack": re.search(r"\.uses_dynamic_stack:\s+(\w+)", meta).group(1),
        }
    return res


def test_every_assembly_walk_kernel_has_no_scratch_no_spills_and_fits_its_sgpr_ceiling(compiled):
    text, _ = compiled
    ks = _kernels(text)
    asm = {k: v for k, v in ks.items() if KERNEL.match(k).group(4) == "1"}
    # the instantiations the launcher can reach: one wave per group, and 2 / 4 / 8 waves per group
    assert sorted(int(KERNEL.match(k).group(3)) for k in asm) == [1, 2, 4, 8]
    for name, r in asm.items():
        split = int(KERNEL.match(name).group(3))
        assert r["scratch"] == 0 and r["dynamic_stack"] == "false", (name, r)     # s32 is not a stack pointer here
        assert r["sgpr_spill"] == 0 and r["vgpr_spill"] == 0, (name, r)
        assert r["vgpr"] <= 64, (name, r)
        assert r["sgpr"] <= (80 if split == 1 else 106), (name, r)
    # no walk kernel at all may spill or touch scratch (the C++ loops share the epilogue and the launch shapes)
    for name, r in ks.items():
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0 and r["sgpr_spill"] == 0, (name, r)


def test_the_only_compiler_diagnostics_are_the_known_reserved_register_notes(compiled):
    _, err = compiled
    warnings = [l for l in err.splitlines() if "warning:" in l]
    for w in warnings:
        assert ("inline asm clobber list contains reserved registers: s32, m0" in w
                or "argument unused during compilation" in w), w
    assert any("reserved registers: s32, m0" in w for w in warnings)     # (the note this file's header explains)


def test_the_one_wave_kernel_really_is_the_hand_written_loop(compiled):
    text, _ = compiled
    body = text[text.index(ONE_WAVE_ASM + ":"):]
    body = body[:body.index("s_endpgm", body.index("Ldone_"))]
    assert "s_load_dwordx16 s[48:63]" in body and "s_load_dwordx16 s[24:39]" in body and "v_cmpx_lt_f32_e32" in body


@pytest.fixture(scope="module")
def engine_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(os.path.dirname(SRC), "bh_engine.hip")
    out = tmp_path_factory.mktemp("engine") / "engine.s"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-w", "-o", str(out), src], cwd=os.path.dirname(src), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


F64_KERNEL = re.compile(r"_ZN2bh15walk_f64_kernelILb([01])ELb([01])ELb([01])ELb([01])EEEv")    # <COMPAT, STATS, DEEP, ASM>


def test_every_fp64_assembly_walk_kernel_has_no_scratch_no_spills_and_fits_its_ceilings(engine_asm):
    """walk64_asm (csrc/bh_walk_f64.hpp, round 4) pins s24..s72 and v20..v47 like the fp32 loop pins its block, with the same
    consequence: s32 is only an ordinary register while the kernel has no private segment, and a spilled SGPR would sit in
    a VGPR lane the block may clobber.  All four instantiations the launcher reaches (reference_compat on / off x one / two
    stack tiers): no scratch, no spills, <= 80 SGPRs and <= 64 VGPRs -- 8 resident waves per SIMD: with one quad in flight per
    wave this walk answers to residency (measured at N = 1M, profiles/r04_f64/walk_ab.txt: 0.704 ms at 7 waves, 0.800 at 5,
    0.834 at 4, 1.088 at 3) --; and the loop in the code object is the hand-written one."""
    ks = {}
    for m in re.finditer(r"\.name:\s+(_ZN2bh15walk_f64_kernel\S+)\n", engine_asm):
        meta = engine_asm[m.start():m.start() + 3000]
        val = lambda key: int(re.search(key + r":\s+(\d+)", meta).group(1))
        ks[m.group(1)] = {"sgpr": val(r"\.sgpr_count"), "vgpr": val(r"\.vgpr_count"), "sgpr_spill": val(r"\.sgpr_spill_count"),
                          "vgpr_spill": val(r"\.vgpr_spill_count"), "scratch": val(r"\.private_segment_fixed_size"),
                          "dynamic_stack": re.search(r"\.uses_dynamic_stack:\s+(\w+)", meta).group(1)}
    asm = {k: v for k, v in ks.items() if F64_KERNEL.match(k).group(4) == "1"}
    assert sorted((F64_KERNEL.match(k).group(1), F64_KERNEL.match(k).group(3)) for k in asm) == [("0", "0"), ("0", "1"), ("1", "0"), ("1", "1")]
    for name, r in asm.items():
        assert F64_KERNEL.match(name).group(2) == "0", name                      # (the counting variant is the C++ loop)
        assert r["scratch"] == 0 and r["dynamic_stack"] == "false", (name, r)
        assert r["sgpr_spill"] == 0 and r["vgpr_spill"] == 0, (name, r)
        assert r["sgpr"] <= 80 and r["vgpr"] <= 64, (name, r)
        body = engine_asm[engine_asm.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm", body.index("Ldone_"))]
        assert "s_load_dwordx16 s[24:39]" in body and "s_load_dwordx16 s[40:55]" in body and "s_load_dwordx8 s[56:63]" in body
        assert body.count("v_cmpx_lt_f64_e32") == 4 and body.count("v_rsq_f64_e32") >= 4
        deep = F64_KERNEL.match(name).group(3) == "1"
        assert ("LpushHi0_" in body) == deep and ("LpopHi_" in body) == deep
        compat = F64_KERNEL.match(name).group(1) == "1"
        assert body.count("v_cmpx_ne_u32_e32") == (8 if compat else 4)
    for name, r in ks.items():
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0, (name, r)


EXACT_KERNEL = re.compile(r"_ZN2bh17walk_exact_kernelILb([01])ELb([01])ELb([01])ELb([01])EEEv")    # <COMPAT, STATS, THR, ASM>


def test_every_bit_exact_assembly_walk_kernel_has_no_scratch_no_spills_and_fits_its_ceilings(engine_asm):
    """walk_exact_asm (csrc/bh_walk_exact.hpp, round 4) pins s24..s49, s64..s71 and v32..v55.  Both instantiations the
    launcher reaches (reference_compat on / off): no scratch, no spills, <= 80 SGPRs and <= 64 VGPRs (8 resident waves per
    SIMD); the loop in the code object is the hand-written one -- one 16-dword request per pair of children, the threshold
    compare narrowing EXEC, the compiler's division expansion only out of line."""
    ks = {}
    for m in re.finditer(r"\.name:\s+(_ZN2bh17walk_exact_kernel\S+)\n", engine_asm):
        meta = engine_asm[m.start():m.start() + 3000]
        val = lambda key: int(re.search(key + r":\s+(\d+)", meta).group(1))
        ks[m.group(1)] = {"sgpr": val(r"\.sgpr_count"), "vgpr": val(r"\.vgpr_count"), "sgpr_spill": val(r"\.sgpr_spill_count"),
                          "vgpr_spill": val(r"\.vgpr_spill_count"), "scratch": val(r"\.private_segment_fixed_size"),
                          "dynamic_stack": re.search(r"\.uses_dynamic_stack:\s+(\w+)", meta).group(1)}
    asm = {k: v for k, v in ks.items() if EXACT_KERNEL.match(k).group(4) == "1"}
    assert sorted(EXACT_KERNEL.match(k).group(1) for k in asm) == ["0", "1"]
    for name, r in asm.items():
        g = EXACT_KERNEL.match(name)
        assert g.group(2) == "0" and g.group(3) == "1", name              # no counters; exact thresholds in the nodes
        assert r["scratch"] == 0 and r["dynamic_stack"] == "false", (name, r)
        assert r["sgpr_spill"] == 0 and r["vgpr_spill"] == 0, (name, r)
        assert r["sgpr"] <= 80 and r["vgpr"] <= 64, (name, r)
        body = engine_asm[engine_asm.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm", body.index("Ldone_"))]
        body = body[body.index("Lloop_"):body.rindex("Ldone_")]            # (the root is evaluated by the C++ statement)
        assert "s_load_dwordx16 s[24:39]" in body and "s_load_dwordx4 s[40:43]" in body
        assert body.count("v_cmpx_le_f64_e32") == 4                       # two children in line, two in the out-of-line stubs
        assert body.count("v_div_scale_f64") == 12 and body.count("v_div_fixup_f64") == 6     # (out of line only: 2 x 3 divisions)
        assert body.count("v_cmpx_ne_u32_e32") == (8 if g.group(1) == "1" else 4)
    for name, r in ks.items():
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0, (name, r)


def test_no_build_or_fp64_walk_kernel_uses_scratch(tmp_path):
    """Every kernel of the engine unit (tree build, sorts, LET, exact and fp64 walks) keeps its working set in registers
    and LDS: private_segment_fixed_size == 0 and no vector-register spills.  (Round 3: bucket_sort_kernel, whose 1,024-thread workgroups
    cap it at 128 VGPRs, spilled 52 bytes per lane after the run fix-up of its short sort was added with two more
    arrays per key; nothing failed -- the kernel was just slower than it had to be.)"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(os.path.dirname(SRC), "bh_engine.hip")
    out = tmp_path / "engine.s"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-w", "-o", str(out), src], cwd=os.path.dirname(src), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    seen = 0
    for m in re.finditer(r"\.name:\s+(\S+)\n", text):
        meta = text[m.start():m.start() + 3000]
        if ".private_segment_fixed_size" not in meta:
            continue
        val = lambda key: int(re.search(key + r":\s+(\d+)", meta).group(1))
        seen += 1
        assert val(r"\.private_segment_fixed_size") == 0, m.group(1)
        assert val(r"\.vgpr_spill_count") == 0, m.group(1)      # (SGPRs spilled to VGPR lanes cost a v_writelane, not memory)
    assert seen >= 40                                              # (all instantiations were looked at)
