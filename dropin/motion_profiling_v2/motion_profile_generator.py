"""Drop-in shim: `import motion_profiling_v2.motion_profile_generator` (the reference's import path, src/ on sys.path) resolves to the
MI355X implementation.  Put this repository's dropin/ directory on sys.path ahead of the reference's src/."""
from vexautonomousplanner_amd.motion_profiling_v2.motion_profile_generator import *  # noqa: F401,F403
from vexautonomousplanner_amd.motion_profiling_v2 import motion_profile_generator as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
