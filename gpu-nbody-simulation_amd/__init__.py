"""MI355X-native Barnes-Hut step behind the reference's file-level interface.

The product is libbhgpu.so (hand-written HIP for gfx950, C-ABI in include/bhgpu.h); this package is
the thin Python host above it: a ctypes binding (`_lib`), the engine object (`engine`), the
reference's text formats (`textio`), its step-loop entry points under their own names (`project`)
and the one-process-per-GPU driver (`distributed`).  There is no CPU fallback: without the HIP
library and a GPU, every compute entry point raises.
"""
from .engine import BarnesHutEngine, BhConfig, BhError, Precision  # noqa: F401
from .textio import loadSimulationDataFromText, save_init_files  # noqa: F401

__all__ = ["BarnesHutEngine", "BhConfig", "BhError", "Precision", "loadSimulationDataFromText",
           "save_init_files"]
