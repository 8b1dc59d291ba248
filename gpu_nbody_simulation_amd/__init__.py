"""Import alias: the package directory is named `gpu-nbody-simulation_amd` (not a valid Python
identifier), so this stub makes it importable as `gpu_nbody_simulation_amd`."""
import pathlib as _pathlib

_real = _pathlib.Path(__file__).resolve().parent.parent / "gpu-nbody-simulation_amd"
__path__ = [str(_real)]
__file__ = str(_real / "__init__.py")
exec(compile((_real / "__init__.py").read_text(), __file__, "exec"))
