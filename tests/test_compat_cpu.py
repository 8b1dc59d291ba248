"""SURVEY 8(f) rows that need no GPU: the nvcc stand-in writes a launcher for the reference's own
command line, and the results parser reads the reference's results format."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NVCC = os.path.join(ROOT, "gpu-nbody-simulation_amd", "compat", "nvcc")


def test_nvcc_standin_writes_the_project_launcher(tmp_path):
    # first_scaling_script.sh:30, verbatim shape
    subprocess.check_call([sys.executable, NVCC, "-DN_BODIES=40000", "-DN_THREADS=1024", "-DN_SIMULATIONS=10",
                           "-o", "project", "project.cu"], cwd=tmp_path)
    launcher = tmp_path / "project"
    assert os.access(launcher, os.X_OK)
    text = launcher.read_text()
    assert "gpu_nbody_simulation_amd.project -DN_BODIES=40000 -DN_THREADS=1024 -DN_SIMULATIONS=10" in text
    assert f'PYTHONPATH="{ROOT}' in text


def test_nvcc_standin_refuses_other_inputs(tmp_path):
    r = subprocess.run([sys.executable, NVCC, "-o", "x", "other.cu"], cwd=tmp_path, capture_output=True)
    assert r.returncode != 0 and b"only `project.cu`" in r.stderr
    r = subprocess.run([sys.executable, NVCC, "-DFOO=1", "-o", "x", "project.cu"], cwd=tmp_path, capture_output=True)
    assert r.returncode != 0


def test_results_parser_reads_the_references_format(tmp_path):
    from gpu_nbody_simulation_amd import scaling
    f = tmp_path / "first_scaling_results.txt"
    # header + records as first_scaling_script.sh:15,36 writes them ($runtime spans several lines)
    f.write_text("n_bodies, n_threads, n_simulations, runtime\n"
                 "40000, 1, 10, \n\n\n\nGPU total computation took 5000 milliseconds.\n\n\n"
                 "GPU parallel computation took 4000000 microseconds.\n"
                 "40000, 1024, 10, \n\nGPU total computation took 1200 milliseconds.\n\n"
                 "GPU parallel computation took 8000 microseconds.\n"
                 "40000, 1024, 10, \n\nGPU total computation took 1100 milliseconds.\n\n"
                 "GPU parallel computation took 6000 microseconds.\n")
    recs = scaling.parse_results(str(f))
    assert [r["parallel_us"] for r in recs] == [4000000, 8000, 6000]
    rows = scaling.summarise(recs)
    assert len(rows) == 2 and rows[1]["runs"] == 2 and rows[1]["parallel_us"] == 7000
    assert abs(rows[1]["speedup_parallel"] - 4000000 / 7000) < 1e-9
    assert abs(rows[1]["body_steps_per_s"] - 40000 * 10 / 7e-3) < 1e-3


def test_project_argument_parsing():
    from gpu_nbody_simulation_amd import project
    a, mac = project._parse(["-DN_BODIES=1000 * 40", "-DN_THREADS=32", "-o", "project", "project.cu"])
    assert mac == {"N_BODIES": 40000, "N_THREADS": 32, "N_SIMULATIONS": 10}        # project.cu:1-11
    a, mac = project._parse([])
    assert mac == {"N_BODIES": 40000, "N_THREADS": 0, "N_SIMULATIONS": 10}         # N_THREADS unset: all bodies at once
    assert project._macro_value("(1000 * 40)") == 40000 and project._macro_value(" 7 ") == 7
    # macro values are parsed, never evaluated: anything but integers and `*` is refused
    for bad in ("().__class__", "__import__('os')", "2**3", "1+1", "0x10", ""):
        with pytest.raises(ValueError):
            project._macro_value(bad)
    with pytest.raises(SystemExit):
        project._parse(["-DN_BODIES=().__class__.__base__"])


def test_gpu_sweep_collects_bench_lines_and_computes_efficiency(tmp_path):
    """scaling.gpu_sweep / summarise_gpus (GPUs on the x axis): the command line it
    builds per GPU count is the driver's, the JSON lines are kept verbatim, efficiency follows `scaling`."""
    import json
    from gpu_nbody_simulation_amd import scaling
    seen = []

    def runner(cmd):
        seen.append(cmd)
        g = int(cmd[cmd.index("--gpus") + 1])
        line = {"metric": "body-steps/sec", "value": 1.0e9 * g ** 0.5, "n_gpus": g, "ms_per_step": 1.0 / g ** 0.5,
                "scaling": "strong", "roofline": {"frac": 0.1} if g == 1 else None, "config": {"workload": "w"}}
        return "noise\n" + json.dumps(line) + "\n"

    out = tmp_path / "g.jsonl"
    rows = scaling.gpu_sweep([1, 4], str(out), ["--steps", "5"], runner=runner, bench_path="bench.py")
    assert [r["n_gpus"] for r in rows] == [1, 4]
    assert "torch.distributed.run" not in seen[0] and seen[0][-4:] == ["--gpus", "1", "--steps", "5"]
    assert seen[1][1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node" in seen[1]
    assert seen[1][seen[1].index("--master-addr") + 1] == "127.0.0.1"
    summary = scaling.summarise_gpus([json.loads(l) for l in open(out)])
    assert summary[1]["speedup"] == pytest.approx(2.0) and summary[1]["efficiency"] == pytest.approx(0.5)
    assert summary[0]["roofline_frac"] == 0.1 and summary[1]["roofline_frac"] is None
    png = tmp_path / "g.png"
    scaling.plot_gpus(summary, str(png))
    assert png.stat().st_size > 1000
    assert scaling.main(["show-gpus", str(out)]) == 0
