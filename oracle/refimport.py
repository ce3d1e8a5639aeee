"""TEST INFRASTRUCTURE ONLY — loader for the upstream reference (never shipped, never on the GPU box).

Imports RohitMovva/VexAutonomousPlanner's ``splines`` and ``motion_profiling_v2`` packages from
``/root/reference/src`` exactly as they are, for the sole purpose of generating golden vectors
(``oracle/gen_golden.py``) and validating the C restatement (``oracle/vap_oracle.c``).

``splines/spline_manager.py:8-9`` imports ``gui.node.Node`` / ``gui.action_point.ActionPoint``
(PyQt6 ``QGraphicsItem``s) for type annotations only.  PyQt6 is not installed here, so two
plain-data stand-ins carrying only the fields the hot path reads (``gui/node.py:17-51``,
``gui/action_point.py:16-41``) are registered in ``sys.modules`` before the import.  They are data
holders, not re-implementations of any reference logic.

Nothing here runs when ``/root/reference`` is absent (``available()`` is False).
"""
import logging
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "splines"))


class Node:
    """Plain-data stand-in for gui.node.Node (fields from gui/node.py:17-51)."""

    def __init__(self, **kw):
        self.is_reverse_node = False
        self.turn = 0
        self.wait_time = 0
        self.stop = False
        self.tangent = None
        self.incoming_magnitude = None
        self.outgoing_magnitude = None
        self.max_velocity = 0
        self.max_acceleration = 0
        for k, v in kw.items():
            setattr(self, k, v)


class ActionPoint:
    """Plain-data stand-in for gui.action_point.ActionPoint (fields from gui/action_point.py:16-41)."""

    def __init__(self, t, **kw):
        self.t = t
        self.stop = False
        self.wait_time = 0
        self.max_velocity = 0
        self.max_acceleration = 0
        for k, v in kw.items():
            setattr(self, k, v)


def load():
    """Return (spline_manager_module, quintic_module, motion_profile_generator_module)."""
    if not available():
        raise RuntimeError("reference tree not present; golden vectors can only be generated "
                           "in the build container")
    sys.dont_write_bytecode = True
    logging.disable(logging.CRITICAL)
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    if "gui.node" not in sys.modules:
        gui = types.ModuleType("gui")
        node = types.ModuleType("gui.node")
        ap = types.ModuleType("gui.action_point")
        node.Node = Node
        ap.ActionPoint = ActionPoint
        gui.node = node
        gui.action_point = ap
        sys.modules["gui"] = gui
        sys.modules["gui.node"] = node
        sys.modules["gui.action_point"] = ap
    # the reference's packages are top-level names ("splines", "motion_profiling_v2"); make sure
    # our own drop-in packages of the same name are not shadowing them in this process
    for name in list(sys.modules):
        if name.split(".")[0] in ("splines", "motion_profiling_v2"):
            mod = sys.modules[name]
            f = getattr(mod, "__file__", "") or ""
            if not f.startswith(REFERENCE_SRC):
                del sys.modules[name]
    import splines.spline_manager as sm  # type: ignore
    import splines.quintic_hermite_spline as qh  # type: ignore
    import motion_profiling_v2.motion_profile_generator as mpg  # type: ignore
    assert sm.__file__.startswith(REFERENCE_SRC)
    return sm, qh, mpg
