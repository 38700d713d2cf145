"""MI355X-native FLIP hot path: HIP kernels behind the C ABI of include/flip_hip.h, plus the host-side
plumbing the tests and the benchmark use.  The HIP library is the only compute path; importing this
package does not load it (binding.load_library does, and raises when it is missing)."""
from . import scenes  # noqa: F401
from .binding import (FIELDS, FlipContext, FlipError, PRECOND_JACOBI, PRECOND_MULTIGRID, SOLVE_DENSITY,  # noqa: F401
                      SOLVE_DIVERGENCE, load_library)
from .substep import FlipSubstep  # noqa: F401
