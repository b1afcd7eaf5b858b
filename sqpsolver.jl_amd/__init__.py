"""sqpsolver.jl_amd -- MI355X-native hot path of SqpSolver.jl (QP sub-problem + merit path).

Import as ``import sqpsolver_jl_amd`` (see the shim at the repository root).  The numerical work is
done by csrc/libsqphip.so (HIP, gfx950) behind the C ABI of include/sqphip.h; there is no CPU
fallback: every entry point raises if the library has not been built."""
from . import acopf_synth  # noqa: F401
from . import dense_synth  # noqa: F401
from . import _lib  # noqa: F401
from . import host  # noqa: F401
from .host import (Context, QpData, QpHip, default_options, SqpHipError, kkt_order,  # noqa: F401
                   kkt_symbolic, mf_host_solve, mf_host_top2_err, mf_host_spine_err)
