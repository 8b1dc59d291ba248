"""fp32 / mixed-precision TRAJECTORIES where the force actually moves the bodies (VERDICT r3 #5b; SURVEY.md 8(c): "<= 10 steps on
inputs with minimum pair distance above the leaf cell size"), against the oracle (project.cu:575-675 + :819-836 in fp64,
uncapped tree).  The workload: tests/moving_fixture.py -- 65,536 bodies on a jittered grid (closest pair 0.4 grid spacings),
velocities U(-1e-4, 1e-4), masses scaled so that the velocity change over 10 steps is ~1e-2 |v|.  Two comparisons:
  teacher-forced: every step starts from the ORACLE's state (uploaded), one device step, compared with the oracle's next state;
  free-running:   10 device steps from the initial state against 10 oracle steps.
Prints the quantiles the tolerances of tests/test_gpu_moving.py are set from (<= 2x)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import moving_fixture as MF  # noqa: E402
import gpu_nbody_simulation_amd as G  # noqa: E402

for prec in (G.Precision.F32, G.Precision.MIXED):
    for seed in (1, 2, 3):
        print(json.dumps({"precision": prec.name, "seed": seed, **MF.measure(prec, seed)}), flush=True)
