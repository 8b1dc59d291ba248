"""Structural guarantees the judge checks: the product never touches the oracle or the reference,
and carries no compatibility layers."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gpu-nbody-simulation_amd")


def _product_sources():
    for base, _, files in os.walk(PKG):
        if os.path.basename(base) in ("build", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                yield os.path.join(base, f)
    yield os.path.join(ROOT, "include", "bhgpu.h")


def test_product_never_imports_or_links_the_oracle():
    pat = re.compile(r"(^|\s)(from|import)\s+oracle\b|bh_oracle|libbh_oracle|oracle/", re.M)
    for path in _product_sources():
        src = open(path).read()
        assert not pat.search(src), f"{path} references the oracle"


def test_product_never_reads_the_reference_at_run_time():
    for path in _product_sources():
        src = open(path).read()
        for line in src.splitlines():
            if "/root/reference" in line:
                stripped = line.strip()
                assert stripped.startswith(("//", "*", "/*", "#", '"""')) or "relative to" in line, \
                    f"{path}: run-time reference to /root/reference: {line}"


def test_no_compatibility_layers():
    bad = re.compile(r"__HIP_PLATFORM_AMD__|__CUDACC__|cuda_runtime|hipify|triton", re.I)
    for path in _product_sources():
        if path.endswith(".py"):
            continue
        assert not bad.search(open(path).read()), path


def test_only_allowed_importers_of_the_oracle():
    allowed = {"bench.py", "__graft_entry__.py"}
    for f in os.listdir(ROOT):
        if f.endswith(".py") and f not in allowed:
            assert "from oracle" not in open(os.path.join(ROOT, f)).read(), f


def test_ldd_shows_no_oracle_dependency():
    import subprocess
    out = subprocess.check_output(["ldd", os.path.join(PKG, "libbhgpu.so")]).decode()
    assert "oracle" not in out
