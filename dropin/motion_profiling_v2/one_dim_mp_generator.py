"""Drop-in shim: `import motion_profiling_v2.one_dim_mp_generator` (the reference's import path, src/ on sys.path) resolves to the
MI355X implementation.  Put this repository's dropin/ directory on sys.path ahead of the reference's src/."""
from vexautonomousplanner_amd.motion_profiling_v2.one_dim_mp_generator import *  # noqa: F401,F403
from vexautonomousplanner_amd.motion_profiling_v2 import one_dim_mp_generator as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
