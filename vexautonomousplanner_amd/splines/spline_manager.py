"""QuinticHermiteSplineManager with the reference's call surface (splines/spline_manager.py:24-594),
backed by the HIP kernels of libvap.so.

What runs where (SM = the reference's splines/spline_manager.py):
  build_path                      -> vap_route_create: splits at reverse / turn nodes, split tangents,
                                     per-spline fit and arc-length tables, all on the device   SM:42-172, 426-475
  precompute_path_properties      -> nothing is materialised: get_curvature / get_heading evaluate
                                     the table entry the reference's step lookup would read    SM:477-580
  get_*_at_parameter, distance_to_time, get_curvature, get_heading -> vap_route_eval / vap_route_lookup
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .. import _lib
from .._device_path import DeviceRoute
from .quintic_hermite_spline import QuinticHermiteSpline


@dataclass
class PathLookupTable:
    """SM:15-21."""
    distances: np.ndarray
    parameters: np.ndarray
    total_length: float


class QuinticHermiteSplineManager:
    def __init__(self):
        self.splines: List[QuinticHermiteSpline] = []
        self.nodes: List = []
        self.action_points: List = []
        self.path_parameters: Dict = {}
        self.arc_length = 0.0
        self.lookup_table: Optional[PathLookupTable] = None
        self._precomputed_properties: Optional[Dict] = None

    # -- construction -----------------------------------------------------------------------------
    def build_path(self, points: np.ndarray, nodes: List, action_points: List) -> bool:
        if len(points) != len(nodes) or len(points) < 2:
            return False
        points = np.asarray(points, dtype=float)
        self.splines = []
        self.nodes = nodes
        self.action_points = action_points
        self._route = DeviceRoute(points, nodes, action_points)   # raises IndexError where the reference does
        r = self._route
        for si in range(r.n_splines):
            a, k = int(r.sp_start[si]), int(r.sp_npts[si])
            seg0 = a   # splines share their split node, so segment i always starts at node i
            self.splines.append(QuinticHermiteSpline._from_route(
                points[a:a + k], r.segments[seg0:seg0 + k - 1], r.segment_lengths[seg0:seg0 + k - 1],
                float(r.sp_param_last[si])))
        self.arc_length = None
        self.lookup_table = None
        return True

    # -- helpers ------------------------------------------------------------------------------------
    def _require(self):
        if not self.splines:
            raise ValueError("No splines have been initialized")

    def _map_parameter_to_spline(self, t: float) -> Tuple[int, float]:
        self._require()
        cumulative = 0
        for i, spline in enumerate(self.splines):
            end = cumulative + len(spline.control_points) - 1
            if t <= end or i == len(self.splines) - 1:
                return i, t - cumulative
            cumulative = end
        raise ValueError("Failed to map parameter to spline segment")

    def _dev(self):
        return self._route

    # -- evaluators (SM:204-241) ----------------------------------------------------------------------
    def get_point_at_parameter(self, t: float) -> np.ndarray:
        self._require()
        return self._route.eval(0, t)[0]

    def get_derivative_at_parameter(self, t: float) -> np.ndarray:
        self._require()
        return self._route.eval(1, t)[0]

    def get_second_derivative_at_parameter(self, t: float) -> np.ndarray:
        self._require()
        return self._route.eval(2, t)[0]

    def get_points_at_parameters(self, ts) -> np.ndarray:
        """Vector form (one launch) of get_point_at_parameter — what a redraw wants (gui/path.py:370-373)."""
        self._require()
        return self._route.eval(0, ts)

    def get_magnitudes_at_parameter(self, idx):
        """SM:174-202."""
        si, lt = self._map_parameter_to_spline(idx)
        node = self.nodes[idx]
        if node.tangent is not None:
            return [node.incoming_magnitude, node.outgoing_magnitude]
        sp = self.splines
        end = sp[si].percent_to_parameter(100)
        if si == 0 and lt == 0:
            return [0, sp[si].get_magnitude(0)]
        if si == len(sp) - 1 and lt == end:
            return [sp[si].get_magnitude(-1), 0]
        if lt == 0:
            return [sp[si - 1].get_magnitude(-1), sp[si].get_magnitude(0)]
        if lt == end:
            return [sp[si].get_magnitude(-1), sp[si + 1].get_magnitude(0)]
        return [sp[si].get_magnitude(round(lt) - 1), sp[si].get_magnitude(round(lt))]

    def percent_to_parameter(self, percent: float):
        """SM:277-289 (note: scales by len(nodes), clamps to len(nodes)-1; quirk Q6)."""
        self._require()
        return min(max(len(self.nodes) * percent, 0), len(self.nodes) - 1)

    # -- tables ---------------------------------------------------------------------------------------
    def build_lookup_table(self, min_samples=1000, max_samples=20000, tolerance=1e-6) -> None:
        """SM:426-475.  The table is built on the device (with the fit for the default size; any other min_samples
        makes the device rebuild it, vap_route_set_table_sizes).  max_samples / tolerance are unused in the reference
        too."""
        if not self.splines:
            raise ValueError("No splines initialized")
        r = self._route
        try:
            r.set_table_sizes(lut_samples=min_samples)
        except _lib.VapError as e:
            if e.status == _lib.VAP_ERR_INVALID and int(min_samples) < 2:   # (other invalid sizes keep their own message)
                raise IndexError("index 1 is out of bounds for axis 0 with size %d" % max(int(min_samples), 0))  # SM:444
            raise
        self.lookup_table = PathLookupTable(distances=r.lut_distances, parameters=r.lut_parameters,
                                            total_length=r.total)

    def precompute_path_properties(self, samples_per_node: int = 1000) -> None:
        """The reference fills samples_per_node*len(nodes) curvature/heading entries here (SM:477-548); the device
        path evaluates the one entry a lookup needs on demand, so only the table geometry is kept."""
        if not self.splines:
            raise ValueError("No splines initialized")
        self._route.set_table_sizes(samples_per_node=samples_per_node)
        self._precomputed_properties = {"samples": len(self.nodes) * samples_per_node, "on_demand": True}

    def rebuild_tables(self):
        self.build_lookup_table()
        self.precompute_path_properties()

    def get_total_arc_length(self) -> float:
        self._require()
        if self.lookup_table is None:
            self.build_lookup_table()
        return self.lookup_table.total_length

    def distance_to_time(self, distance: float) -> float:
        if self.lookup_table is None:
            self.build_lookup_table()
        if distance <= 0:
            return 0
        if distance >= self.lookup_table.total_length:
            return len(self.nodes) - 1
        return float(self._dev().lookup(0, distance)[0])

    def get_heading(self, t: float) -> float:
        if self._precomputed_properties is None:
            self.precompute_path_properties()
        if self.lookup_table is None:
            self.build_lookup_table()
        return float(self._dev().lookup(2, t)[0])

    def get_curvature(self, t: float) -> float:
        if self._precomputed_properties is None:
            self.precompute_path_properties()
        if self.lookup_table is None:
            self.build_lookup_table()
        return float(self._dev().lookup(1, t)[0])

    def _get_heading(self, t: float) -> float:
        d = self.get_derivative_at_parameter(t)
        return float(np.arctan2(d[1], d[0]))

    def _get_curvature(self, t: float) -> float:
        d1 = self.get_derivative_at_parameter(t)
        d2 = self.get_second_derivative_at_parameter(t)
        s2 = d1[0] ** 2 + d1[1] ** 2
        if s2 < 1e-10:
            return 0.0
        return float((d1[0] * d2[1] - d1[1] * d2[0]) / s2 ** 1.5)

    def validate_path_continuity(self) -> bool:
        pass  # an empty stub in the reference too (SM:420-424)
