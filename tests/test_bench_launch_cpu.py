"""bench.py's launch plumbing: `--gpus N` must never print a one-GPU number labelled N.

No device is touched: `--dry-run` stops after the process group has formed (gloo) and the ranks have been all-reduced."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                           "BHGPU_REHEARSE_ON_DEVICE")}
    env.update(kw)
    return env


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "7", "--warmup", "2", "--dry-run"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["sum_of_ranks"] == 1 and j["gpus_arg"] == 2 and (j["steps"], j["warmup"]) == (7, 2)
    assert j["launched_by"] == "torch.distributed.run"
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr      # (the command it ran is on stderr)


def test_gpus_2_under_the_drivers_own_launch_line():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["sum_of_ranks"] == 1


def test_a_world_size_that_contradicts_gpus_is_an_error_not_a_relabelled_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_one_gpu_runs_directly():
    r = subprocess.run([sys.executable, BENCH, "--dry-run"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["launched_by"] == "direct"
