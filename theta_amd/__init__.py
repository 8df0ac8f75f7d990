"""
theta_amd -- MI355X-native implementation of THetA's combinatorial likelihood search.

Host side: plain Python + numpy mirroring the reference's operator interface
(Enumerator / Optimizer / CalcAllC.L2,L3 / do_optimization) over a ctypes C ABI
(include/theta_hip.h, theta_amd/libtheta_hip.so).  All arithmetic of the hot path runs in
hand-written HIP kernels for gfx950; nothing here falls back to the CPU.
"""
from ._lib import Comm, Context, Problem, ThetaError, NoCandidates, default_context, load, LIB_PATH  # noqa: F401

__all__ = ["Comm", "Context", "Problem", "ThetaError", "NoCandidates", "default_context", "load", "LIB_PATH"]
