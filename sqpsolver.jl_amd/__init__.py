"""sqpsolver.jl_amd -- MI355X-native hot path of SqpSolver.jl (QP sub-problem + merit path).

Import as ``import sqpsolver_jl_amd`` (see the shim at the repository root)."""
from . import acopf_synth  # noqa: F401
