"""Batched trajectory generation on one MI355X: the new batch-of-paths axis (north_star).

Device memory, streams and the multi-GPU launcher come from PyTorch-ROCm (plumbing only); all
numerics run in the HIP kernels of libvap.so through the C-ABI of include/vap.h.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .synth import DEFAULT_CONSTRAINTS, END_VEL, START_VEL

FIELDS = ("x", "y", "heading", "curvature", "velocity")


def _torch_dtype(dtype):
    if dtype in ("f32", "fp32", torch.float32, np.float32, _lib.VAP_F32):
        return torch.float32, _lib.VAP_F32
    if dtype in ("f64", "fp64", torch.float64, np.float64, _lib.VAP_F64):
        return torch.float64, _lib.VAP_F64
    raise ValueError(f"unsupported dtype {dtype!r}")


class ProfileResult(dict):
    """What profile() / profile_routes() return: a dict of tensors, stamped with the generation of the context rows it
    belongs to (apply_node_limits / time_profile read rows the context kept from that very call)."""
    generation = None


class BatchedTrajectoryGenerator:
    """rebuild_tables + forward_backward_pass (SM:582-594, MPG:70-316) for B independent
    plain-node paths per call, outputs resident in HBM as (B, S) tensors."""

    def __init__(self, device=0, dtype="f32", timing=False, velocity_kernel="auto", recurrence="f64", time_domain_residual=True):
        """dtype "f32" | "f64": type of inputs and outputs.  recurrence (dtype "f32" only): "f64" (default) carries
        the velocity recurrence and its curvature / heading-difference rows in fp64 behind the fp32 outputs — the
        mode that holds 1e-5 against the reference on every path; "f32" is the all-fp32 recurrence (faster,
        ~1.4 % of config-3-shaped paths have a sample above 1e-5).
        time_domain_residual (VAP_OPT_TIME_DOMAIN_RESIDUAL, dtype "f32" with the fp64 recurrence): keep what every stored
        fp32 velocity lost of its fp64 value (4 B per sample-point of extra writes) for a following time_profile();
        False for batches that never go to the time domain."""
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: vexautonomousplanner_amd has no CPU path")
        self.device = torch.device("cuda", device)
        self.tdtype, self.vdtype = _torch_dtype(dtype)
        self.ctx = _lib.Context(device)
        if timing:
            self.ctx.set_timing(True)
        self._L = _lib.lib()
        self.set_velocity_kernel(velocity_kernel)
        self.set_recurrence(recurrence)
        self.time_domain_residual = bool(time_domain_residual)
        self.ctx.set_option(_lib.OPT_TIME_DOMAIN_RESIDUAL, 1 if self.time_domain_residual else 0)
        self._generation = 0

    def set_recurrence(self, which):
        """"f64" | "f32" (VAP_OPT_F32_RECURRENCE; only matters for dtype "f32")."""
        self.recurrence = {"f64": "f64", "f32": "f32", "fp64": "f64", "fp32": "f32"}[which]
        self.ctx.set_option(_lib.OPT_F32_RECURRENCE,
                            _lib.RECURRENCE_F64 if self.recurrence == "f64" else _lib.RECURRENCE_F32)

    def set_time_kernel(self, which):
        """"auto" | "lane" | "quad" | "fused" (VAP_OPT_TIME_KERNEL): lanes per path in the time domain's kinematic
        recurrence; "fused": the quad recurrence with the geometry in the same workgroup."""
        self.ctx.set_option(_lib.OPT_TIME_KERNEL, {"auto": _lib.TIME_KERNEL_AUTO, "lane": _lib.TIME_KERNEL_LANE,
                                                   "quad": _lib.TIME_KERNEL_QUAD, "fused": _lib.TIME_KERNEL_FUSED}[which])

    def set_velocity_kernel(self, which):
        """"auto" | "seq_literal" | "seq_fast" | "relax" | "relax_block" | "relax_wave" | "lanes" | "lanes16" | "lanes32" |
        "lanes64" | "relax_rounds" (VAP_OPT_VELOCITY_KERNEL)."""
        table = {"auto": _lib.VELOCITY_AUTO, "seq_literal": _lib.VELOCITY_SEQ_LITERAL,
                 "seq_fast": _lib.VELOCITY_SEQ_FAST, "relax": _lib.VELOCITY_RELAX,
                 "relax_block": _lib.VELOCITY_RELAX_BLOCK, "relax_wave": _lib.VELOCITY_RELAX_WAVE,
                 "lanes": _lib.VELOCITY_LANES, "lanes16": _lib.VELOCITY_LANES_16, "lanes32": _lib.VELOCITY_LANES_32,
                 "lanes64": _lib.VELOCITY_LANES_64, "relax_rounds": _lib.VELOCITY_RELAX_ROUNDS}
        self.ctx.set_option(_lib.OPT_VELOCITY_KERNEL, table[which])

    def profile(self, waypoints, constraints=DEFAULT_CONSTRAINTS, samples=None, dd=None,
                start_vel=START_VEL, end_vel=END_VEL, want=FIELDS, out=None, capacity=None):
        """waypoints: (B, W, 2) tensor on this device, dtype matching the generator.

        Exactly one of:
          samples=S : fixed grid of S samples per path, dd_b = L_b/(S-1.5)   (bench semantics)
          dd=0.005  : the reference's grid (MPG:112-122); ``capacity`` = row length of the
                      outputs (default: enough for a 64 ft path), n_samples per path in meta[:,3]
        Returns a dict of (B, S) tensors for the requested fields plus
          meta  (B,4) fp64: parameters[-1], total_length, dd, n_samples
          flags (B,)  int32 bit-mask (VAP_FLAG_*)
        """
        if (samples is None) == (dd is None):
            raise ValueError("give exactly one of samples= or dd=")
        wp = waypoints
        if wp.device != self.device or wp.dtype != self.tdtype or wp.dim() != 3 or wp.shape[2] != 2:
            raise ValueError(f"waypoints must be a (B,W,2) {self.tdtype} tensor on {self.device}")
        wp = wp.contiguous()
        B, W, _ = wp.shape
        if samples is not None:
            S, ddv = int(samples), 0.0
        else:
            ddv = float(dd)
            S = int(capacity) if capacity is not None else int(np.ceil(64.0 / ddv)) + 2
        c = _lib.make_constraints(constraints)
        res = self._result_for(out)
        for f in FIELDS:
            if f in want or f == "velocity":
                t = res.get(f)
                if t is None or t.shape != (B, S) or t.dtype != self.tdtype:
                    res[f] = torch.empty((B, S), dtype=self.tdtype, device=self.device)
        if "meta" not in res or res["meta"].shape != (B, 4):
            res["meta"] = torch.empty((B, 4), dtype=torch.float64, device=self.device)
        if "flags" not in res or res["flags"].shape != (B,):
            res["flags"] = torch.empty((B,), dtype=torch.int32, device=self.device)

        def p(name):
            t = res.get(name) if (name in want or name == "velocity") else None
            return C.c_void_p(t.data_ptr()) if t is not None else None

        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        st = self._L.vap_profile_batch(self.ctx.handle, self.vdtype, B, W, S, ddv,
                                       C.c_void_p(wp.data_ptr()), C.byref(c), float(start_vel),
                                       float(end_vel), p("x"), p("y"), p("heading"),
                                       p("curvature"), p("velocity"),
                                       C.c_void_p(res["meta"].data_ptr()),
                                       C.c_void_p(res["flags"].data_ptr()))
        _lib.check(st, "vap_profile_batch")
        self._last_shape = (B, W, S)
        return self._stamp(res, out)

    @staticmethod
    def _result_for(out):
        """The dict the tensors of a call go into: a fresh ProfileResult, the caller's ProfileResult, or — for a plain
        dict passed as out= — a ProfileResult view that starts from its entries (the caller's dict is filled in place
        as well: _stamp copies the entries back)."""
        if out is None:
            return ProfileResult()
        return out if isinstance(out, ProfileResult) else ProfileResult(out)

    def _stamp(self, res, out=None):
        """The context keeps rows of the batch it sampled last (tables, fp64 curvature / heading-difference rows, the
        velocity residual): a result is tied to them by (this generator, generation number), and the follow-up calls
        refuse a result of another generator or one that a later profile()/profile_routes() call has superseded."""
        self._generation += 1
        res.generation = (id(self), self._generation)
        if out is not None and out is not res:      # a plain dict as out=: it receives every tensor of the call too
            out.update(res)
        return res

    def _check_current(self, result, what):
        if getattr(result, "generation", None) != (id(self), self._generation):
            raise ValueError(f"{what} needs the ProfileResult of THIS generator's LAST profile()/profile_routes() call: the "
                             "context rows it reads belong to another batch")

    @staticmethod
    def check_flags(result):
        """Synchronises, then raises RuntimeError if any path of ``result`` carries a flag that makes its rows unusable:
        VAP_FLAG_NOCONVERGE (a bounded wait of the long-row velocity kernel expired — a time-sliced or counter-profiled
        GPU; re-run the batch, or with velocity_kernel="seq_fast") or VAP_FLAG_BAD_ROUTE.  Degenerate / truncated paths
        are reported by their flags only, as the reference reports them by value.  profile() itself never synchronises:
        call this where the rows are consumed."""
        f = result["flags"]
        bad = (f & (_lib.FLAG_NOCONVERGE | _lib.FLAG_BAD_ROUTE)) != 0
        if bool(bad.any().item()):
            idx = torch.nonzero(bad).flatten()[:8].tolist()
            raise RuntimeError(f"{int(bad.sum().item())} paths are flagged no-converge / bad-route (first: {idx}); flags "
                               f"{[int(f[i].item()) for i in idx]}")
        return result

    def profile_routes(self, waypoints, node_reverse=None, node_turn=None, node_tangent=None, node_magnitudes=None,
                       constraints=DEFAULT_CONSTRAINTS, samples=None, dd=None, start_vel=START_VEL, end_vel=END_VEL,
                       want=FIELDS, out=None, capacity=None):
        """``profile`` for B routes whose reverse / turn nodes cut them into several splines (SM:42-172 for a whole
        batch; vap_profile_routes).  Per-node attributes are (B, W) array-likes:
          node_reverse (bool), node_turn (degrees), node_tangent (B, W, 2) with NaN rows for None and
          node_magnitudes (B, W, 2) = [incoming, outgoing] for the nodes that have a tangent.
        Returns the dict of ``profile`` plus "spline_counts" (B,) int32.  ``apply_node_limits`` works on the result
        as for plain paths; the time domain of split routes goes through the drop-in classes."""
        if (samples is None) == (dd is None):
            raise ValueError("give exactly one of samples= or dd=")
        wp = waypoints
        if wp.device != self.device or wp.dtype != self.tdtype or wp.dim() != 3 or wp.shape[2] != 2:
            raise ValueError(f"waypoints must be a (B,W,2) {self.tdtype} tensor on {self.device}")
        wp = wp.contiguous()
        B, W, _ = wp.shape
        if node_tangent is not None and node_magnitudes is None:
            raise ValueError("node_tangent needs node_magnitudes ((B, W, 2): incoming, outgoing) for the nodes that have a tangent")
        rev = np.zeros((B, W), dtype=np.int32) if node_reverse is None else np.asarray(node_reverse).astype(bool).astype(np.int32).reshape(B, W)
        turn = np.zeros((B, W)) if node_turn is None else np.asarray(node_turn, dtype=np.float64).reshape(B, W)
        if not np.all(np.isfinite(turn)):
            raise ValueError("node_turn must be finite (degrees; 0 = no turn)")
        split = ((rev != 0) | (turn != 0))[:, 1:W - 1] if W > 2 else np.zeros((B, 0), dtype=bool)
        max_splines = int(1 + (split.sum(axis=1).max() if split.size else 0))
        dev = self.device
        d_rev = torch.tensor(rev, device=dev)
        d_turn = torch.tensor(turn, device=dev)
        d_tan = d_mag = None
        if node_tangent is not None:
            d_tan = torch.tensor(np.asarray(node_tangent, dtype=np.float64).reshape(B, W, 2), device=dev)
            d_mag = torch.tensor(np.nan_to_num(np.asarray(node_magnitudes, dtype=np.float64)).reshape(B, W, 2), device=dev)
        if samples is not None:
            S, ddv = int(samples), 0.0
        else:
            ddv = float(dd)
            S = int(capacity) if capacity is not None else int(np.ceil(64.0 / ddv)) + 2
        c = _lib.make_constraints(constraints)
        res = self._result_for(out)
        for f in FIELDS:
            if f in want or f == "velocity":
                t = res.get(f)
                if t is None or t.shape != (B, S) or t.dtype != self.tdtype:
                    res[f] = torch.empty((B, S), dtype=self.tdtype, device=dev)
        res["meta"] = torch.empty((B, 4), dtype=torch.float64, device=dev)
        res["flags"] = torch.empty((B,), dtype=torch.int32, device=dev)
        res["spline_counts"] = torch.empty((B,), dtype=torch.int32, device=dev)

        def p(name):
            t = res.get(name) if (name in want or name == "velocity") else None
            return C.c_void_p(t.data_ptr()) if t is not None else None
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        st = self._L.vap_profile_routes(self.ctx.handle, self.vdtype, B, W, S, ddv, max_splines, ptr(wp), ptr(d_rev), ptr(d_turn),
                                        ptr(d_tan), ptr(d_mag), C.byref(c), float(start_vel), float(end_vel), p("x"), p("y"),
                                        p("heading"), p("curvature"), p("velocity"), ptr(res["meta"]), ptr(res["flags"]),
                                        ptr(res["spline_counts"]))
        _lib.check(st, "vap_profile_routes")
        self._last_shape = (B, W, S)
        return self._stamp(res, out)

    def time_profile(self, result, constraints=DEFAULT_CONSTRAINTS, dt=0.01, capacity_rows=None, out=None, node_reverse=None):
        """Time-domain resample (the loop of generate_motion_profile, MPG:413-628) of the batch that
        ``profile`` has just produced with this generator: ``result`` is its return value (the
        velocity rows and meta are read from it, the spline tables from the context).

        Returns a dict with
          rows      (B, capacity_rows, 8) fp64: time, position, velocity, acceleration, heading,
                    angular velocity, x, y per time step of ``dt`` seconds (MPG:558-592)
          counts    (B, 2) int32: rows written, entries of nodes_map
          nodes_map (B, W) int32: row index at which each node is passed (MPG:420, 527-529)
        Paths that need more than capacity_rows rows are cut there and flagged (result["flags"]).
        ``node_reverse`` (B, W): the is_reverse_node flags of routes (``profile_routes``): rows behind an odd number of
        them are reversed (heading - pi, negated velocity / acceleration; MPG:431-433, 540-541, 555, 587-589).
        """
        vel, meta = result["velocity"], result["meta"]
        B, S = vel.shape
        last = getattr(self, "_last_shape", None)
        if last is None or (last[0], last[2]) != (B, S):
            raise ValueError("time_profile needs the result of this generator's last profile() call")
        self._check_current(result, "time_profile")
        W = last[1]
        if capacity_rows is None:
            capacity_rows = 4096
        res = {} if out is None else out
        if "rows" not in res or res["rows"].shape != (B, capacity_rows, 8):
            res["rows"] = torch.empty((B, capacity_rows, 8), dtype=torch.float64, device=self.device)
        if "counts" not in res or res["counts"].shape != (B, 2):
            res["counts"] = torch.empty((B, 2), dtype=torch.int32, device=self.device)
        if "nodes_map" not in res or res["nodes_map"].shape != (B, W):
            res["nodes_map"] = torch.empty((B, W), dtype=torch.int32, device=self.device)
        c = _lib.make_constraints(constraints)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        d_rev = None
        if node_reverse is not None:
            d_rev = torch.tensor(np.asarray(node_reverse).astype(bool).astype(np.int32).reshape(B, W), device=self.device)
        st = self._L.vap_time_profile_routes(self.ctx.handle, self.vdtype, B, W, S, C.c_void_p(meta.data_ptr()),
                                             C.c_void_p(vel.data_ptr()), C.byref(c), float(dt), int(capacity_rows),
                                             C.c_void_p(d_rev.data_ptr()) if d_rev is not None else None,
                                             C.c_void_p(res["rows"].data_ptr()), C.c_void_p(res["counts"].data_ptr()),
                                             C.c_void_p(res["nodes_map"].data_ptr()), C.c_void_p(result["flags"].data_ptr()))
        _lib.check(st, "vap_time_profile_routes")
        return res

    def apply_node_limits(self, result, constraints=DEFAULT_CONSTRAINTS, node_max_velocity=None, node_stop=None,
                          node_max_acceleration=None, action_points=None, start_vel=START_VEL, end_vel=END_VEL):
        """Routes whose nodes / action points carry ``max_velocity``, ``max_acceleration`` and ``stop``
        (motion_utils Node / ActionPoint; MPG:100-176, 194-196, 256-257): recompute the velocity rows of the
        batch ``profile`` has just produced with this generator under those limits.  The geometry rows (x, y,
        heading, curvature) do not depend on them.

          node_max_velocity, node_max_acceleration  (B, W) array-like, 0 = none
          node_stop                                 (B, W) array-like of bool
          action_points  optional list (per path) of lists of dicts {"t", "max_velocity", "max_acceleration", "stop"}
        Returns ``result`` with "velocity" replaced, plus "vcap" (the reference's initial velocity list per
        sample), "node_sample" (B, W) and "action_sample" (B, M): the sample at which each takes effect
        (INT_MAX: never — e.g. an action point the reference skips, see include/vap.h).
        Reverse / turn nodes and waits change more than the limits: those routes go through the drop-in classes
        (vap_route_*)."""
        vel, meta = result["velocity"], result["meta"]
        B, S = vel.shape
        last = getattr(self, "_last_shape", None)
        if last is None or (last[0], last[2]) != (B, S):
            raise ValueError("apply_node_limits needs the result of this generator's last profile() call")
        self._check_current(result, "apply_node_limits")
        W = last[1]
        as2d = lambda a: np.zeros((B, W)) if a is None else np.asarray(a, dtype=np.float64).reshape(B, W)
        mv, ma = as2d(node_max_velocity), as2d(node_max_acceleration)
        stop = np.zeros((B, W), dtype=bool) if node_stop is None else np.asarray(node_stop).astype(bool).reshape(B, W)
        M = max((len(a) for a in action_points), default=0) if action_points is not None else 0
        Mp = max(M, 1)
        ap_t = np.full((B, Mp), np.inf)
        ap_mv, ap_ma = np.zeros((B, Mp)), np.zeros((B, Mp))
        ap_stop = np.zeros((B, Mp), dtype=np.int32)
        if M:                                   # (no per-path Python loop for a batch without action points)
            for b, al in enumerate(action_points):
                for i, a in enumerate(al):
                    ap_t[b, i] = float(a["t"])
                    ap_mv[b, i] = float(a.get("max_velocity", 0.0))
                    ap_ma[b, i] = float(a.get("max_acceleration", 0.0))
                    ap_stop[b, i] = int(bool(a.get("stop", False)))
        with_acc = bool((ma > 0).any() or (ap_ma > 0).any())
        dev = self.device
        # two uploads (one per element type) instead of seven: the arrays one after the other, views of the device block
        fl_dev = torch.tensor(np.concatenate([x.ravel() for x in (mv, ma, ap_t, ap_mv, ap_ma)]), device=dev)
        it_dev = torch.tensor(np.concatenate([stop.astype(np.int32).ravel(), ap_stop.ravel()]), device=dev)
        o = np.cumsum([0, B * W, B * W, B * Mp, B * Mp, B * Mp])
        d = {"mv": fl_dev[o[0]:o[1]], "ma": fl_dev[o[1]:o[2]], "ap_t": fl_dev[o[2]:o[3]], "ap_mv": fl_dev[o[3]:o[4]],
             "ap_ma": fl_dev[o[4]:o[5]], "stop": it_dev[:B * W], "ap_stop": it_dev[B * W:]}
        # the limit rows have the type of the recurrence they enter (vap_limit_rows_dtype): fp64 in the default mode — an
        # fp32-rounded limit (13.9 ft/s^2) is amplified by the recurrence past 1e-5
        ldt = torch.float64 if self._L.vap_limit_rows_dtype(self.ctx.handle, self.vdtype) == _lib.VAP_F64 else torch.float32
        vcap = torch.empty((B, S), dtype=ldt, device=dev)
        acc_f = torch.empty((B, S), dtype=ldt, device=dev) if with_acc else None
        acc_b = torch.empty((B, S), dtype=ldt, device=dev) if with_acc else None
        dec_b = torch.empty((B,), dtype=ldt, device=dev) if with_acc else None
        node_k = torch.empty((B, W), dtype=torch.int32, device=dev)
        ap_k = torch.empty((B, max(M, 1)), dtype=torch.int32, device=dev)
        c = _lib.make_constraints(constraints)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self._L.vap_route_limits(self.ctx.handle, self.vdtype, B, W, M, S, None, ptr(meta), ptr(d["mv"]), ptr(d["ma"]),
                                            ptr(d["stop"]), ptr(d["ap_t"]), ptr(d["ap_mv"]), ptr(d["ap_ma"]), ptr(d["ap_stop"]),
                                            C.byref(c), float(end_vel), ptr(vcap), ptr(acc_f), ptr(acc_b), ptr(dec_b), ptr(node_k),
                                            ptr(ap_k)), "vap_route_limits")
        curv = result.get("curvature")
        ctx_rows = self.vdtype == _lib.VAP_F32 and self.recurrence == "f64"   # the context holds the fp64 rows itself
        if curv is None and not ctx_rows:
            raise ValueError("apply_node_limits needs the curvature rows (profile(want=...) must include 'curvature')")
        _lib.check(self._L.vap_velocity_pass_limits(self.ctx.handle, self.vdtype, B, S, C.byref(c), float(start_vel),
                                                    float(end_vel), ptr(meta), ptr(curv), None, ptr(vcap), ptr(acc_f),
                                                    ptr(acc_b), ptr(dec_b), ptr(vel), ptr(result["flags"])),
                   "vap_velocity_pass_limits")
        result["vcap"] = vcap
        result["node_sample"] = node_k
        result["action_sample"] = ap_k[:, :M]
        return result

    def insert_waits(self, result, tp, node_wait_time=None, action_points=None, dt=0.01, capacity_rows=None,
                     node_turn=None, node_reverse=None, constraints=DEFAULT_CONSTRAINTS):
        """Waits of nodes / action points and ``actions_map`` on top of ``tp = time_profile(result, ...)``
        (MPG:457-476, 509-518, 543-553): returns a new dict with
          rows (B, capacity_rows, 8), counts (B, 3) int32 [rows, nodes_map entries, actions_map entries],
          nodes_map (B, W), actions_map (B, M) int32.
        ``action_points``: list (per path) of lists of dicts {"t", "wait_time"} in route order.
        ``node_turn`` (B, W) degrees: in-place turns (MPG:487-507) inserted where the node is passed, before its wait, with
        the trapezoid of ``constraints``; ``node_reverse`` (B, W): only the heading of a wait at node 0 reads it."""
        rows_in, counts_in, nodes_in = tp["rows"], tp["counts"], tp["nodes_map"]
        B, cap_in, _ = rows_in.shape
        W = nodes_in.shape[1]
        dev = self.device
        wait = np.zeros((B, W)) if node_wait_time is None else np.asarray(node_wait_time, dtype=np.float64).reshape(B, W)
        M = max((len(a) for a in action_points), default=0) if action_points is not None else 0
        ap_t = np.full((B, max(M, 1)), np.inf)
        ap_w = np.zeros((B, max(M, 1)))
        if M:                                   # (no per-path Python loop for a batch without action points)
            for b, al in enumerate(action_points):
                for i, a in enumerate(al):
                    ap_t[b, i] = float(a["t"])
                    ap_w[b, i] = float(a.get("wait_time", 0.0))
        extra = int(np.max(np.floor(wait / dt).sum(axis=1) + np.floor(ap_w / dt).sum(axis=1))) + 1
        c = _lib.make_constraints(constraints)
        d_turn = d_rev = None
        if node_turn is not None:
            turn = np.asarray(node_turn, dtype=np.float64).reshape(B, W)
            d_turn = torch.tensor(turn, device=dev)
            # rows of a turn: the trapezoid's duration / dt + 1 (one_dim_mp_generator.py:4-69), bounded generously
            arc = np.abs(np.radians(turn)) * c.track_width / 2
            dur = 2 * c.max_vel / c.max_acc + arc / c.max_vel
            extra += int(np.max(np.where(turn != 0, np.ceil(dur / dt) + 3, 0).sum(axis=1)))
        if node_reverse is not None:
            d_rev = torch.tensor(np.asarray(node_reverse).astype(bool).astype(np.int32).reshape(B, W), device=dev)
        cap_out = int(capacity_rows) if capacity_rows is not None else cap_in + extra
        out = {"rows": torch.empty((B, cap_out, 8), dtype=torch.float64, device=dev),
               "counts": torch.zeros((B, 3), dtype=torch.int32, device=dev),
               "nodes_map": torch.zeros((B, W), dtype=torch.int32, device=dev),
               "actions_map": torch.zeros((B, max(M, 1)), dtype=torch.int32, device=dev)}
        # one upload for the three float arrays
        fl_dev = torch.tensor(np.concatenate([wait.ravel(), ap_t.ravel(), ap_w.ravel()]), device=dev)
        d_wait, d_apt, d_apw = fl_dev[:B * W], fl_dev[B * W:B * W + ap_t.size], fl_dev[B * W + ap_t.size:]
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(self._L.vap_time_insert_events(self.ctx.handle, B, W, M, cap_in, cap_out, float(dt), C.byref(c),
                                                  ptr(result["meta"]), ptr(rows_in), ptr(counts_in), ptr(nodes_in), ptr(d_wait),
                                                  ptr(d_turn), ptr(d_rev), ptr(d_apt), ptr(d_apw), ptr(out["rows"]), ptr(out["counts"]),
                                                  ptr(out["nodes_map"]), ptr(out["actions_map"]), ptr(result["flags"])),
                   "vap_time_insert_events")
        out["actions_map"] = out["actions_map"][:, :M]
        return out

    def timing(self):
        return self.ctx.last_timing()
