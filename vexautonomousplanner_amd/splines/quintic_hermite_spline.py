"""QuinticHermiteSpline with the reference's call surface (splines/quintic_hermite_spline.py:11-748),
fitted and evaluated by the HIP kernels of libvap.so.

What runs where:
  fit / _compute_derivatives / _compute_parameters   -> K1 (vap_fit)                QHS:30-219, 719-736
  get_point / get_derivative / get_second_derivative -> device evaluator (vap_eval_host)  QHS:221-251, 473-504
  get_arc_length (Gauss-Legendre)                    -> device derivative samples + the quadrature sum  QHS:592-644
Quirks kept on purpose (SURVEY.md §8(a) Q-list):
  Q1  fit() before set_all_tangents() returns False (the reference dereferences a None there)
  Q3  set_starting_tangent patches the LAST segment's start-tangent row (QHS:561)
"""
from typing import Optional, Tuple

import numpy as np

from .._device_path import DevicePath
from .spline import Spline


class QuinticHermiteSpline(Spline):
    def __init__(self):
        super().__init__()
        self.first_derivatives: Optional[np.ndarray] = None
        self.second_derivatives: Optional[np.ndarray] = None
        self.starting_tangent: Optional[np.ndarray] = None
        self.ending_tangent: Optional[np.ndarray] = None
        self.set_tangents = None
        self.control_points: Optional[np.ndarray] = None
        self.parameters = np.zeros(0)
        self.segments = []
        self.segment_lengths = []
        self._dev: Optional[DevicePath] = None

    # -- construction -----------------------------------------------------------------------------
    def set_tangent(self, tangent: np.ndarray, index: int):
        if self.set_tangents is None:
            self.set_tangents = np.zeros_like(self.control_points, dtype=float)
        self.set_tangents[index] = tangent

    def set_all_tangents(self, tangents):
        self.set_tangents = tangents

    def _tangent_rows(self, k):
        """set_tangents -> ([k][2] incoming, [k][2] outgoing) with NaN for None (QHS:102-115)."""
        tin = np.full((k, 2), np.nan)
        tout = np.full((k, 2), np.nan)
        for i in range(k):
            pair = self.set_tangents[i]
            if pair is None:
                continue
            if pair[0] is not None:
                tin[i] = pair[0]
            if pair[1] is not None:
                tout[i] = pair[1]
        return tin, tout

    def fit(self, x, y, first_derivatives=None, second_derivatives=None) -> bool:
        """QHS:30-138.  Caller-supplied derivatives are used only when both are present — with one missing the
        reference's _compute_derivatives overwrites both (QHS:66-68, 149-219) — and, as there, derivatives left by
        an earlier fit of this object are reused by the next one (they are attributes, QHS:56, 62, 66)."""
        if len(x) != len(y) or len(x) < 2:
            return False
        k = len(x)
        if first_derivatives is not None:
            if len(first_derivatives) != k:
                return False
            self.first_derivatives = first_derivatives
        if second_derivatives is not None:
            if len(second_derivatives) != k:
                return False
            self.second_derivatives = second_derivatives
        estimate = self.first_derivatives is None or self.second_derivatives is None
        if self.set_tangents is None:
            return False  # Q1: the reference dereferences set_tangents[i] and its blanket except returns False
        pts = np.column_stack((np.asarray(x, dtype=float), np.asarray(y, dtype=float)))
        try:
            tin, tout = self._tangent_rows(k)
            first = second = None
            if not estimate:
                first = np.asarray(self.first_derivatives, dtype=float).reshape(k, 2)
                second = np.asarray(self.second_derivatives, dtype=float).reshape(k, 2)
            stan = etan = None
            if self.starting_tangent is not None:
                # QHS:129-130 -> set_starting_tangent: anything but a (2,) ndarray is refused there, silently
                if isinstance(self.starting_tangent, np.ndarray) and self.starting_tangent.shape == (2,):
                    stan = self.starting_tangent
            if self.ending_tangent is not None:
                if isinstance(self.ending_tangent, np.ndarray) and self.ending_tangent.shape == (2,):
                    etan = self.ending_tangent
        except (TypeError, IndexError, ValueError):
            return False  # the reference's blanket `except Exception: return False` (QHS:136-138)
        dev = DevicePath()
        if not dev.fit(pts, tin, tout, first, second, stan, etan):
            return False
        self._dev = dev
        self.control_points = pts
        self.first_derivatives = dev.first
        self.second_derivatives = dev.second
        # QHS:719-736 chord-length parameters (only [0] and [-1] are ever read); the cumulative
        # chord and parameters[-1] itself come from the device fit
        cum = np.concatenate(([0.0], np.cumsum(dev.segment_lengths)))
        self.parameters = (cum * (k - 1) / cum[-1]) if cum[-1] > 0 else np.linspace(0, k - 1, k)
        self.parameters[-1] = dev.param_last
        self.segments = [dev.segments[i] for i in range(k - 1)]
        self.segment_lengths = [float(v) for v in dev.segment_lengths]
        return True

    @classmethod
    def _from_route(cls, control_points, segments, segment_lengths, param_last):
        """One spline of a route the manager fitted on the device (vap_route_create)."""
        from .._device_path import DevicePath
        self = cls()
        k = len(control_points)
        self.control_points = np.array(control_points, dtype=float)
        self.set_tangents = [[None, None]] * k
        dev = DevicePath.__new__(DevicePath)
        DevicePath.__init__(dev)
        dev.W = k
        dev.segments = np.ascontiguousarray(segments)
        dev.segment_lengths = np.ascontiguousarray(segment_lengths)
        dev.param_last = float(param_last)
        self._dev = dev
        cum = np.concatenate(([0.0], np.cumsum(dev.segment_lengths)))
        self.parameters = (cum * (k - 1) / cum[-1]) if cum[-1] > 0 else np.linspace(0, k - 1, k)
        self.parameters[-1] = dev.param_last
        self.segments = [dev.segments[i] for i in range(k - 1)]
        self.segment_lengths = [float(v) for v in dev.segment_lengths]
        return self

    # -- evaluation -------------------------------------------------------------------------------
    def _require_fit(self):
        if not self.segments:
            raise ValueError("Spline has not been fitted yet")

    def get_point(self, t: float) -> np.ndarray:
        self._require_fit()
        return self._dev.eval(0, t)[0]

    def get_derivative(self, t: float, debug: bool = False) -> np.ndarray:
        self._require_fit()
        return self._dev.eval(1, t)[0]

    def get_second_derivative(self, t: float, debug: bool = False) -> np.ndarray:
        self._require_fit()
        return self._dev.eval(2, t)[0]

    # QHS:288-469: the four basis rows.  The class's own evaluators never come here (the device evaluates point and
    # derivatives whole); they exist for callers of the reference's helpers, the third-derivative one included (in
    # the reference's class, called by nothing there).
    def _basis(self, order: int, t: float) -> np.ndarray:
        from .._device_path import basis_rows
        return basis_rows(order, [t])[0]

    def _get_basis_functions(self, t: float) -> np.ndarray:
        return self._basis(0, t)

    def _get_basis_derivatives(self, t: float) -> np.ndarray:
        return self._basis(1, t)

    def _get_basis_second_derivatives(self, t: float) -> np.ndarray:
        return self._basis(2, t)

    def _get_basis_third_derivatives(self, t: float) -> np.ndarray:
        return self._basis(3, t)

    def get_points(self, ts) -> np.ndarray:
        """Vector form of get_point (one device launch for all parameters)."""
        self._require_fit()
        return self._dev.eval(0, ts)

    def _normalize_parameter(self, t: float) -> Tuple[float, int]:
        if not self.parameters.size:
            raise ValueError("Spline has not been fitted yet")
        t = max(self.parameters[0], min(t, self.parameters[-1]))
        idx = int(t - self.parameters[0])
        if idx == len(self.segments):
            idx = len(self.segments) - 1
        return t - (self.parameters[0] + idx), idx

    def get_magnitude(self, idx):
        return self.segment_lengths[idx]

    def percent_to_parameter(self, percent: float) -> float:
        self._require_fit()
        return self.parameters[0] + self.parameters[-1] * (percent / 100)

    def percent_to_point(self, percent: float) -> np.ndarray:
        return self.get_point(self.percent_to_parameter(percent))

    def get_end_parameter(self) -> float:
        if not self.parameters.size:
            raise ValueError("Spline has not been fitted yet")
        return self.parameters[-1]

    # -- tangent setters (QHS:543-590) --------------------------------------------------------------
    def set_starting_tangent(self, tangent: np.ndarray) -> bool:
        """QHS:543-564.  Quirk Q3 kept: the row that changes is the LAST segment's start tangent.  (The evaluators
        take the segment rows from this host mirror on every call, so the patch is all there is to do.)"""
        if not isinstance(tangent, np.ndarray) or tangent.shape != (2,):
            return False
        self.first_derivatives[0] = tangent      # TypeError before the first fit, as in the reference
        if len(self.segments) > 0:
            self.segments[-1][2] = tangent
        self.starting_tangent = tangent
        return True

    def set_ending_tangent(self, tangent: np.ndarray) -> bool:
        """QHS:566-590."""
        if not isinstance(tangent, np.ndarray) or tangent.shape != (2,):
            return False
        self.first_derivatives[-1] = tangent
        if len(self.segments) > 0:
            self.segments[-1][3] = tangent
        self.ending_tangent = tangent
        return True

    # -- exact-ish arc length API (QHS:592-717; unused by the manager) -------------------------------
    def get_arc_length(self, t_start: float, t_end: float, num_points: int = 20) -> float:
        self._require_fit()
        if t_start >= t_end:
            raise ValueError("t_start must be less than t_end")
        t_min, t_max = self.parameters[0], self.parameters[-1]
        if t_start < t_min or t_end > t_max:
            raise ValueError(f"Parameters must be within range [{t_min}, {t_max}]")
        nodes, weights = np.polynomial.legendre.leggauss(num_points)
        half = (t_end - t_start) / 2
        mid = (t_start + t_end) / 2
        d = self._dev.eval(1, nodes * half + mid)
        return float(half * np.sum(weights * np.hypot(d[:, 0], d[:, 1])))

    def get_total_arc_length(self) -> float:
        self._require_fit()
        return self.get_arc_length(self.parameters[0], self.parameters[-1])

    def get_parameter_by_arc_length(self, arc_length: float, tolerance: float = 1e-6,
                                    max_iterations: int = 50) -> float:
        self._require_fit()
        if arc_length < 0:
            raise ValueError("Arc length must be non-negative")
        total = self.get_total_arc_length()
        if arc_length > total:
            raise ValueError(f"Arc length {arc_length} exceeds total length {total}")
        if arc_length == 0:
            return self.parameters[0]
        if arc_length == total:
            return self.parameters[-1]
        t0 = lo = self.parameters[0]
        hi = self.parameters[-1]
        for _ in range(max_iterations):
            mid = (lo + hi) / 2
            err = self.get_arc_length(t0, mid) - arc_length
            if abs(err) < tolerance:
                return mid
            if err > 0:
                hi = mid
            else:
                lo = mid
        return (lo + hi) / 2
