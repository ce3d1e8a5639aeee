"""Plain-data route elements: the fields of the reference's Qt items that the path code reads
(gui/node.py:17-51, gui/action_point.py:16-41), without Qt.  The GUI's own Node / ActionPoint
objects work unchanged (the managers only read these attributes)."""
from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class Node:
    is_reverse_node: bool = False
    turn: float = 0
    wait_time: float = 0
    stop: bool = False
    tangent: Optional[np.ndarray] = None
    incoming_magnitude: Optional[float] = None
    outgoing_magnitude: Optional[float] = None
    max_velocity: float = 0
    max_acceleration: float = 0


@dataclass
class ActionPoint:
    t: float = 0.0
    stop: bool = False
    wait_time: float = 0
    max_velocity: float = 0
    max_acceleration: float = 0
