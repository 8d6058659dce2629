"""godot_amd: MI355X-native kernels for godot's weighted-photon likelihood and
periodogram hot path, behind the reference's core.py API."""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from . import ops  # noqa: F401

__all__ = ['ops']
