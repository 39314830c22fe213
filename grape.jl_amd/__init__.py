"""grape.jl_amd -- MI355X-native GRAPE gradient evaluator (hot path of JuliaQuantumControl/GRAPE.jl).

Only what the path needs:

* ``csrc/``      hand-written gfx950 HIP kernels + the C ABI of ``include/grape_hip.h``
* ``api``        ctypes binding of that C ABI (``GrapeHip``); fails loudly without the HIP library
* ``grape``      host-side mirror of the reference interface for this path
                 (``Trajectory``, ``GrapeWrk``, ``evaluate_functional``, ``evaluate_gradient_b``,
                 ``optimize``; /root/reference/src/optimize.jl, src/workspace.jl)
* ``sharded``   trajectory shards over ``torch.distributed`` (one process per GPU, RCCL)
* ``synth``      deterministic synthetic problems of BASELINE.json (SplitMix64 + Box-Muller)
"""
from . import synth  # noqa: F401
from .api import (GrapeHip, GrapeHipError, build_library, library_path,  # noqa: F401
                  J_T_SM, J_T_SS, J_T_RE, GRAD_GRADGEN, GRAD_TAYLOR, PROP_EXP, PROP_SERIES)

__all__ = ["GrapeHip", "GrapeHipError", "build_library", "library_path", "synth",
           "J_T_SM", "J_T_SS", "J_T_RE", "GRAD_GRADGEN", "GRAD_TAYLOR", "PROP_EXP", "PROP_SERIES"]
