"""CPU oracle of the FLIP hot path (test infrastructure; see oracle/flip_oracle.c)."""
from .binding import Oracle, load_reference_shim  # noqa: F401
