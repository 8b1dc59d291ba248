"""Scaling experiments in the reference's result format.

The reference's drivers append one line per run,

    n_bodies, n_threads, n_simulations, <stdout of ./project>          (first_scaling_script.sh:36)
    n_bodies, n_threads, n_simulations, repetition, <stdout>            (second_scaling_script.sh)

and plot_first_scale.py:55-59 / plot_second_scale.py:19-20 parse them with regexes.  `parse_results`
reads that format (ours or the reference's files); `sweep` produces it with this engine;
`summarise` computes what the reference's plotters plot (median kernel/total time, speed-up and
efficiency against the first row of each size); `plot` draws it when matplotlib is importable.

`gpus` is the 1/2/4/8-GPU sweep the reference has no counterpart for: it runs the repository's bench.py
once per GPU count (through torch.distributed.run for more than one), keeps the JSON lines, and prints
speed-up, efficiency and the walk kernel's roofline fraction; `show-gpus` re-reads such a file.

    python -m gpu_nbody_simulation_amd.scaling sweep --bodies 65536 1048576 --simulations 10 --repeats 3
    python -m gpu_nbody_simulation_amd.scaling show scaling_results.txt
    python -m gpu_nbody_simulation_amd.scaling gpus --gpus 1 2 4 8 --out gpu_scaling.jsonl -- --steps 20
    python -m gpu_nbody_simulation_amd.scaling show-gpus gpu_scaling.jsonl --png gpu_scaling.png
"""
from __future__ import annotations

import argparse
import io
import re
import statistics
import sys
from contextlib import redirect_stdout

RE_HEAD = re.compile(r"^\s*(\d+)\s*,\s*([^,]+)\s*,\s*(\d+)\s*,")                    # plot_first_scale.py:55
RE_PAR = re.compile(r"GPU parallel computation took\s+(\d+)\s+microseconds")        # :58
RE_TOT = re.compile(r"GPU total computation took\s+(\d+)\s+milliseconds\.")         # :59


def parse_results(path: str):
    """[{n_bodies, n_threads, n_simulations, parallel_us, total_ms}, ...]; a record may span lines
    (the program's stdout contains newlines), so the file is split at header matches."""
    text = open(path).read()
    recs, cur = [], None
    for line in text.splitlines():
        m = RE_HEAD.match(line)
        if m and not line.lower().startswith("n_bodies"):
            if cur:
                recs.append(cur)
            cur = {"n_bodies": int(m.group(1)), "n_threads": m.group(2).strip(), "n_simulations": int(m.group(3)),
                   "text": line}
        elif cur:
            cur["text"] += "\n" + line
    if cur:
        recs.append(cur)
    out = []
    for r in recs:
        p, t = RE_PAR.search(r["text"]), RE_TOT.search(r["text"])
        if p and t:
            out.append({k: r[k] for k in ("n_bodies", "n_threads", "n_simulations")} |
                       {"parallel_us": int(p.group(1)), "total_ms": int(t.group(1))})
    return out


def summarise(records):
    """Median per (n_bodies, n_threads); speed-up/efficiency vs the first n_threads of each size."""
    groups: dict = {}
    for r in records:
        groups.setdefault((r["n_bodies"], r["n_threads"]), []).append(r)
    rows = []
    for (nb, nt), rs in groups.items():
        rows.append({"n_bodies": nb, "n_threads": nt, "runs": len(rs),
                     "parallel_us": statistics.median(x["parallel_us"] for x in rs),
                     "total_ms": statistics.median(x["total_ms"] for x in rs),
                     "n_simulations": rs[0]["n_simulations"]})
    base = {}
    for r in rows:
        base.setdefault(r["n_bodies"], r)
    for r in rows:
        b = base[r["n_bodies"]]
        r["speedup_parallel"] = b["parallel_us"] / max(r["parallel_us"], 1e-9)
        r["body_steps_per_s"] = r["n_bodies"] * r["n_simulations"] / max(r["parallel_us"] * 1e-6, 1e-12)
    return rows


def sweep(bodies, threads, simulations, repeats, out_path, extra_args=()):
    from . import project
    with open(out_path, "w") as f:
        f.write("n_bodies, n_threads, n_simulations, runtime\n")
        for nb in bodies:
            for nt in threads:
                for _ in range(repeats):
                    buf = io.StringIO()
                    with redirect_stdout(buf):
                        project.main([f"-DN_BODIES={nb}", f"-DN_THREADS={nt}", f"-DN_SIMULATIONS={simulations}",
                                      *extra_args])
                    f.write(f"{nb}, {nt}, {simulations}, {buf.getvalue()}\n")
                    f.flush()


def plot(rows, png_path):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots()
    for nb in sorted({r["n_bodies"] for r in rows}):
        sel = [r for r in rows if r["n_bodies"] == nb]
        ax.plot(range(len(sel)), [r["body_steps_per_s"] for r in sel], marker="o", label=f"N={nb}")
        ax.set_xticks(range(len(sel)))
        ax.set_xticklabels([r["n_threads"] for r in sel])
    ax.set_xlabel("n_threads (bodies walked at a time: passes of whole 256-thread workgroups)")
    ax.set_ylabel("body-steps / s (device time)")
    ax.set_yscale("log")
    ax.legend()
    fig.savefig(png_path, dpi=120)


def gpu_sweep(gpus, out_path, bench_args=(), runner=None, bench_path=None, port=29600):
    """Run bench.py for every GPU count and append its JSON line to out_path.  runner(cmd) -> stdout
    is injectable for tests; the default runs the command as a child process."""
    import json
    import os
    import subprocess
    bench_path = bench_path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    if runner is None:
        def runner(cmd):
            return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    rows = []
    with open(out_path, "w") as f:
        for k, g in enumerate(gpus):
            cmd = [sys.executable]
            if g > 1:
                cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(g),
                        "--master-addr", "127.0.0.1", "--master-port", str(port + k)]
            cmd += [bench_path, "--gpus", str(g), *bench_args]
            line = [ln for ln in runner(cmd).splitlines() if ln.startswith("{")][-1]
            rows.append(json.loads(line))
            f.write(line + "\n")
            f.flush()
    return rows


def summarise_gpus(rows):
    """Speed-up and efficiency against the smallest GPU count of the file (strong scaling: same work;
    weak: work grows with the GPU count, so efficiency = value / (gpus x value at the base))."""
    rows = sorted(rows, key=lambda r: r["n_gpus"])
    base = rows[0]
    out = []
    for r in rows:
        ratio = r["n_gpus"] / base["n_gpus"]
        sp = r["value"] / base["value"]
        out.append({"n_gpus": r["n_gpus"], "value": r["value"], "ms_per_step": r["ms_per_step"], "speedup": sp,
                    "efficiency": sp / ratio, "scaling": r.get("scaling", "strong"),
                    "roofline_frac": (r.get("roofline") or {}).get("frac"),
                    "workload": (r.get("config") or {}).get("workload")})
    return out


def plot_gpus(rows, png_path):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots()
    g = [r["n_gpus"] for r in rows]
    ax.plot(g, [r["value"] for r in rows], marker="o", label="measured")
    ax.plot(g, [rows[0]["value"] * x / g[0] for x in g], linestyle="--", color="gray", label="ideal")
    ax.set_xscale("log", base=2)
    ax.set_yscale("log")
    ax.set_xticks(g)
    ax.set_xticklabels([str(x) for x in g])
    ax.set_xlabel("GPUs")
    ax.set_ylabel("body-steps / s")
    ax.set_title(rows[0].get("workload") or "")
    ax.legend()
    fig.savefig(png_path, dpi=120)
    plt.close(fig)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("sweep")
    s.add_argument("--bodies", type=int, nargs="+", default=[40000])
    s.add_argument("--threads", type=int, nargs="+", default=[1024])
    s.add_argument("--simulations", type=int, default=10)
    s.add_argument("--repeats", type=int, default=3)
    s.add_argument("--out", default="scaling_results.txt")
    s.add_argument("--precision", choices=["f64", "f32"], default="f32")
    s.add_argument("--max-depth", type=int, default=16)
    h = sub.add_parser("show")
    h.add_argument("file")
    h.add_argument("--png")
    g = sub.add_parser("gpus")
    g.add_argument("--gpus", type=int, nargs="+", default=[1, 2, 4, 8])
    g.add_argument("--out", default="gpu_scaling.jsonl")
    g.add_argument("--png")
    g.add_argument("bench_args", nargs="*", help="passed to bench.py (put them after --)")
    hg = sub.add_parser("show-gpus")
    hg.add_argument("file")
    hg.add_argument("--png")
    a = ap.parse_args(argv)
    if a.cmd in ("gpus", "show-gpus"):
        import json
        if a.cmd == "gpus":
            raw = gpu_sweep(a.gpus, a.out, a.bench_args)
        else:
            raw = [json.loads(ln) for ln in open(a.file) if ln.startswith("{")]
        rows = summarise_gpus(raw)
        print("%6s %16s %12s %9s %11s %14s" % ("gpus", "body-steps/s", "ms/step", "speed-up", "efficiency", "roofline frac"))
        for r in rows:
            print("%6d %16.4e %12.4f %9.2f %11.2f %14s" % (r["n_gpus"], r["value"], r["ms_per_step"], r["speedup"],
                                                            r["efficiency"],
                                                            "-" if r["roofline_frac"] is None else "%.3f" % r["roofline_frac"]))
        if a.png:
            plot_gpus(rows, a.png)
        return 0
    if a.cmd == "sweep":
        sweep(a.bodies, a.threads, a.simulations, a.repeats, a.out,
              ["--init", "gpu", "--precision", a.precision, "--max-depth", str(a.max_depth)])
        a.file, a.png = a.out, None
    rows = summarise(parse_results(a.file))
    print("%10s %10s %5s %14s %10s %16s" % ("n_bodies", "n_threads", "runs", "parallel_us", "total_ms", "body-steps/s"))
    for r in rows:
        print("%10d %10s %5d %14.0f %10.0f %16.3e" % (r["n_bodies"], r["n_threads"], r["runs"], r["parallel_us"],
                                                      r["total_ms"], r["body_steps_per_s"]))
    if getattr(a, "png", None):
        plot(rows, a.png)
    return 0


if __name__ == "__main__":
    sys.exit(main())
