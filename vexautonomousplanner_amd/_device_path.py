"""Single-path plumbing shared by the drop-in classes: one fitted spline resident on the device
(fp64, so the GUI-facing scalar accessors agree with the reference to rounding), reached only through
the C-ABI of include/vap.h.  torch is used for the device buffers, nothing else."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def basis_rows(order, ts, device=0):
    """vap_basis_host: rows [H0..H5] of the quintic Hermite basis of derivative order 0..3 (QHS:288-469)."""
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: vexautonomousplanner_amd has no CPU path")
    ts = np.ascontiguousarray(np.atleast_1d(ts), dtype=np.float64)
    out = np.empty((len(ts), 6), dtype=np.float64)
    dp = C.POINTER(C.c_double)
    ctx = _lib.default_context(device)
    _lib.check(_lib.lib().vap_basis_host(ctx.handle, int(order), len(ts), ts.ctypes.data_as(dp), out.ctypes.data_as(dp)),
               "vap_basis_host")
    return out


class DevicePath:
    """Segments / arc-length table of ONE spline on the GPU + host mirrors of the small arrays."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: vexautonomousplanner_amd has no CPU path")
        self.device = torch.device("cuda", device)
        self.ctx = _lib.default_context(device)
        self._L = _lib.lib()
        self.W = 0
        self.segments = None        # host (G,6,2) fp64
        self.segment_lengths = None  # host (G,)
        self.param_last = None
        self.lut = None             # host (1000,) fp64
        self.total = None
        self._d = {}

    # -- K1 -------------------------------------------------------------------------------------
    def fit(self, points, tan_in=None, tan_out=None, first=None, second=None, start_tangent=None, end_tangent=None):
        """vap_fit_ex: QuinticHermiteSpline.fit with everything the class can carry into it (QHS:30-138).  Leaves
        the first / second derivatives the segments were built from in self.first / self.second."""
        pts = np.ascontiguousarray(points, dtype=np.float64)
        W = len(pts)
        dev = self.device
        d = self._d = {
            "wp": torch.tensor(pts[None], dtype=torch.float64, device=dev),
            "seg": torch.empty((1, W - 1, 6, 2), dtype=torch.float64, device=dev),
            "seglen": torch.empty((1, W - 1), dtype=torch.float64, device=dev),
            "meta": torch.zeros((1, 4), dtype=torch.float64, device=dev),
            "flags": torch.zeros((1,), dtype=torch.int32, device=dev),
            "tin": None, "tout": None,
        }
        if tan_in is not None:
            d["tin"] = torch.tensor(np.asarray(tan_in, dtype=np.float64)[None], device=dev)
            d["tout"] = torch.tensor(np.asarray(tan_out, dtype=np.float64)[None], device=dev)
        def opt(a, shape):
            return None if a is None else torch.tensor(np.asarray(a, dtype=np.float64).reshape(shape), device=dev)
        d["first"], d["second"] = opt(first, (1, W, 2)), opt(second, (1, W, 2))
        d["stan"], d["etan"] = opt(start_tangent, (1, 2)), opt(end_tangent, (1, 2))
        d["fd"] = torch.empty((1, W, 2), dtype=torch.float64, device=dev)
        d["sd"] = torch.empty((1, W, 2), dtype=torch.float64, device=dev)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        st = self._L.vap_fit_ex(self.ctx.handle, _lib.VAP_F64, 1, W, _ptr(d["wp"]), _ptr(d["tin"]), _ptr(d["tout"]),
                                _ptr(d["first"]), _ptr(d["second"]), _ptr(d["stan"]), _ptr(d["etan"]), _ptr(d["seg"]),
                                _ptr(d["seglen"]), _ptr(d["fd"]), _ptr(d["sd"]), _ptr(d["meta"]), _ptr(d["flags"]))
        if st == _lib.VAP_ERR_INVALID:
            return False
        _lib.check(st, "vap_fit_ex")
        self.first = d["fd"][0].cpu().numpy()
        self.second = d["sd"][0].cpu().numpy()
        self.W = W
        self.segments = d["seg"][0].cpu().numpy()
        self.segment_lengths = d["seglen"][0].cpu().numpy()
        self.param_last = float(d["meta"][0, 0].item())
        self.lut = None
        self.total = None
        return True

    # -- K2 -------------------------------------------------------------------------------------
    def build_lut(self):
        d = self._d
        d["lut"] = torch.empty((1, _lib.LUT_SAMPLES), dtype=torch.float64, device=self.device)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.vap_build_lut(self.ctx.handle, 1, self.W, _ptr(d["seg"]), _ptr(d["lut"]),
                                         _ptr(d["meta"]), _ptr(d["flags"])), "vap_build_lut")
        self.lut = d["lut"][0].cpu().numpy()
        self.total = float(self.lut[-1])

    # -- scalar / vector accessors ---------------------------------------------------------------
    def eval(self, order, ts):
        ts = np.ascontiguousarray(np.atleast_1d(ts), dtype=np.float64)
        out = np.empty((len(ts), 2), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _lib.check(self._L.vap_eval_host(self.ctx.handle, self.W, self.segments.ctypes.data_as(dp),
                                         self.param_last, int(order), len(ts), ts.ctypes.data_as(dp),
                                         out.ctypes.data_as(dp)), "vap_eval_host")
        return out

    def lookup(self, what, xs):
        xs = np.ascontiguousarray(np.atleast_1d(xs), dtype=np.float64)
        out = np.empty(len(xs), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _lib.check(self._L.vap_lookup_host(self.ctx.handle, self.W, self.segments.ctypes.data_as(dp),
                                           self.param_last, self.lut.ctypes.data_as(dp), int(what), len(xs),
                                           xs.ctypes.data_as(dp), out.ctypes.data_as(dp)), "vap_lookup_host")
        return out

    # -- K3..K5 ----------------------------------------------------------------------------------
    def forward_backward(self, constraints, dd, start_vel, end_vel, want=("velocity",)):
        """MPG:70-316 on the reference grid for this path; returns dict of host fp64 arrays."""
        d = self._d
        total = self.total
        cap = int(np.ceil(total / dd)) + 8
        dev = self.device
        outs = {k: torch.empty((1, cap), dtype=torch.float64, device=dev)
                for k in ("x", "y", "heading", "curvature", "velocity", "dtheta")}
        c = _lib.make_constraints(constraints)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        L = self._L
        _lib.check(L.vap_sample(self.ctx.handle, _lib.VAP_F64, 1, self.W, cap, float(dd), _ptr(d["seg"]),
                                _ptr(d["lut"]), _ptr(d["meta"]), _ptr(outs["x"]), _ptr(outs["y"]),
                                _ptr(outs["heading"]), _ptr(outs["curvature"]), _ptr(outs["dtheta"]),
                                _ptr(d["flags"])), "vap_sample")
        _lib.check(L.vap_velocity_pass(self.ctx.handle, _lib.VAP_F64, 1, cap, C.byref(c), float(start_vel),
                                       float(end_vel), _ptr(d["meta"]), _ptr(outs["curvature"]),
                                       _ptr(outs["dtheta"]), None, _ptr(outs["velocity"]), _ptr(d["flags"])),
                   "vap_velocity_pass")
        n = int(d["meta"][0, 3].item())
        if int(d["flags"][0].item()) & _lib.FLAG_NOCONVERGE:
            # the long-row kernel bounds every wait by wall time (a grid must drain); on a GPU that is time-sliced or
            # profiled with serialising counters such a wait can expire and leave rows that are flagged, not right.  This
            # call is synchronous anyway: take the path once more through the one-lane sequential sweep (the same rows,
            # bit for bit, when nothing times out).
            d["flags"].bitwise_and_(~_lib.FLAG_NOCONVERGE)
            self.ctx.set_option(_lib.OPT_VELOCITY_KERNEL, _lib.VELOCITY_SEQ_FAST)
            try:
                _lib.check(L.vap_velocity_pass(self.ctx.handle, _lib.VAP_F64, 1, cap, C.byref(c), float(start_vel),
                                               float(end_vel), _ptr(d["meta"]), _ptr(outs["curvature"]),
                                               _ptr(outs["dtheta"]), None, _ptr(outs["velocity"]), _ptr(d["flags"])),
                           "vap_velocity_pass (sequential sweep after a timed-out wait)")
            finally:
                self.ctx.set_option(_lib.OPT_VELOCITY_KERNEL, _lib.VELOCITY_AUTO)
        return {k: outs[k][0, :n].cpu().numpy() for k in want}, n


_ROUTE_BATCH = {}


def _route_batch(device):
    """The fp64 batch generator the single-route calls share on a device (its own context: scratch rows, tables)."""
    gen = _ROUTE_BATCH.get(device)
    if gen is None:
        from .batch import BatchedTrajectoryGenerator
        gen = _ROUTE_BATCH[device] = BatchedTrajectoryGenerator(device, "f64")
    return gen


class DeviceRoute:
    """One general route (any node / action-point attributes) on the GPU: vap_route_* of include/vap.h.

    ``use_batch_kernels`` (class or instance attribute, default False): True runs forward_backward (velocity only) and
    motion_profile on the batch kernels with B = 1 — 0.7 ms instead of 3.2 ms for config 1.  The default stays on the
    one-lane statement-by-statement vap_route_* layer because it keeps the reference's own operation order: against the
    oracle on 21 000 random routes and robots it is at 3.3e-9 / 1.5e-7 (velocity / time rows) at worst, where the batch
    kernels' reordered arithmetic, amplified by the recurrence, reaches 1.2e-7 / 7.5e-6 on 42 000 (profiles/r04_fuzz.txt)
    — inside north_star's 1e-5, but a drop-in is judged on fidelity first."""
    use_batch_kernels = False

    def __init__(self, points, nodes, action_points, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: vexautonomousplanner_amd has no CPU path")
        self.ctx = _lib.default_context(device)
        self.ctx.set_stream(torch.cuda.current_stream(torch.device("cuda", device)).cuda_stream)
        self._L = _lib.lib()
        pts = np.ascontiguousarray(points, dtype=np.float64)
        W, M = len(pts), len(action_points)
        self.W, self.M = W, M
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)

        def dcol(vals):
            return np.ascontiguousarray(vals, dtype=np.float64)

        def icol(vals):
            return np.ascontiguousarray([1 if v else 0 for v in vals], dtype=np.int32)

        tan = np.full((W, 2), np.nan)
        mag = np.zeros((W, 2))
        for i, n in enumerate(nodes):
            if n.tangent is not None:
                tan[i] = np.asarray(n.tangent, dtype=float)
                mag[i] = (n.incoming_magnitude, n.outgoing_magnitude)
        keep = {
            "wp": pts, "rev": icol(n.is_reverse_node for n in nodes), "turn": dcol([n.turn for n in nodes]),
            "stop": icol(n.stop for n in nodes), "wait": dcol([n.wait_time for n in nodes]),
            "maxv": dcol([n.max_velocity for n in nodes]), "maxa": dcol([n.max_acceleration for n in nodes]),
            "tan": tan, "mag": mag,
            "apt": dcol([a.t for a in action_points]), "aps": icol(a.stop for a in action_points),
            "apw": dcol([a.wait_time for a in action_points]), "apv": dcol([a.max_velocity for a in action_points]),
            "apa": dcol([a.max_acceleration for a in action_points]),
        }
        self._keep = keep
        self._nodes_have_limits = bool(keep["stop"].any() or (keep["maxv"] > 0).any() or (keep["maxa"] > 0).any() or M > 0)
        self._aps = [{"t": float(a.t), "max_velocity": float(a.max_velocity), "max_acceleration": float(a.max_acceleration),
                      "stop": bool(a.stop), "wait_time": float(a.wait_time)} for a in action_points]
        d = _lib.RouteDesc()
        d.n_nodes, d.n_actions = W, M
        d.waypoints = keep["wp"].ctypes.data_as(dp)
        d.is_reverse = keep["rev"].ctypes.data_as(ip)
        d.turn = keep["turn"].ctypes.data_as(dp)
        d.stop = keep["stop"].ctypes.data_as(ip)
        d.wait_time = keep["wait"].ctypes.data_as(dp)
        d.max_velocity = keep["maxv"].ctypes.data_as(dp)
        d.max_acceleration = keep["maxa"].ctypes.data_as(dp)
        d.tangent = keep["tan"].ctypes.data_as(dp)
        d.magnitudes = keep["mag"].ctypes.data_as(dp)
        if M:
            d.ap_t = keep["apt"].ctypes.data_as(dp)
            d.ap_stop = keep["aps"].ctypes.data_as(ip)
            d.ap_wait_time = keep["apw"].ctypes.data_as(dp)
            d.ap_max_velocity = keep["apv"].ctypes.data_as(dp)
            d.ap_max_acceleration = keep["apa"].ctypes.data_as(dp)
        h = C.c_void_p()
        st = self._L.vap_route_create(self.ctx.handle, C.byref(d), C.byref(h))
        self.handle = None
        if st == _lib.VAP_ERR_INVALID:
            raise IndexError(self._L.vap_last_error().decode())   # what the reference raises (SM:88,97)
        _lib.check(st, "vap_route_create")
        self.handle = h
        ns, tot = C.c_int(), C.c_double()
        _lib.check(self._L.vap_route_info(h, C.byref(ns), C.byref(tot)), "vap_route_info")
        self.n_splines, self.total = ns.value, tot.value
        n = self.n_splines
        self.sp_start = np.zeros(n, dtype=np.int32)
        self.sp_npts = np.zeros(n, dtype=np.int32)
        self.sp_param_last = np.zeros(n)
        self.segments = np.zeros((W - 1, 6, 2))
        self.segment_lengths = np.zeros(W - 1)
        self.lut_samples, self.samples_per_node = _lib.LUT_SAMPLES, _lib.LUT_SAMPLES
        self._fetch_tables(with_splines=True)

    def _fetch_tables(self, with_splines=False):
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        n = self.n_splines
        self.lut_distances = np.zeros(n * self.lut_samples)
        self.lut_parameters = np.zeros(n * self.lut_samples)
        _lib.check(self._L.vap_route_get_splines(
            self.handle,
            self.sp_start.ctypes.data_as(ip) if with_splines else None,
            self.sp_npts.ctypes.data_as(ip) if with_splines else None,
            self.sp_param_last.ctypes.data_as(dp) if with_splines else None,
            self.segments.ctypes.data_as(dp) if with_splines else None,
            self.segment_lengths.ctypes.data_as(dp) if with_splines else None,
            self.lut_distances.ctypes.data_as(dp), self.lut_parameters.ctypes.data_as(dp)), "vap_route_get_splines")

    def set_table_sizes(self, lut_samples=None, samples_per_node=None):
        """build_lookup_table(min_samples) / precompute_path_properties(samples_per_node) with other sizes than the
        reference's defaults (SM:426-427, 477): the device rebuilds the route's tables."""
        ls = self.lut_samples if lut_samples is None else int(lut_samples)
        sn = self.samples_per_node if samples_per_node is None else int(samples_per_node)
        if (ls, sn) == (self.lut_samples, self.samples_per_node):
            return
        _lib.check(self._L.vap_route_set_table_sizes(self.handle, ls, sn), "vap_route_set_table_sizes")
        changed = ls != self.lut_samples
        self.lut_samples, self.samples_per_node = ls, sn
        if changed:
            ns, tot = C.c_int(), C.c_double()
            _lib.check(self._L.vap_route_info(self.handle, C.byref(ns), C.byref(tot)), "vap_route_info")
            self.total = tot.value
            self._fetch_tables()

    def _try_batched(self, call):
        """call(batch generator), or None for a route the batched entry points refuse (VAP_ERR_UNSUPPORTED: more nodes
        than their tables hold) — the one-lane vap_route_* layer takes those."""
        try:
            return call(_route_batch(self.ctx.device))
        except _lib.VapError as e:
            if e.status == _lib.VAP_ERR_UNSUPPORTED:
                return None
            raise

    def _batched_motion_profile(self, constraints, dt, dd):
        """(rows, nodes_map, actions_map) through the batch kernels, or None when the one-lane layer has to take the call
        (a route the batch refuses, or marks degenerate / not converged); IndexError where the reference raises."""
        k = self._keep
        res = self._try_batched(lambda g: self._batched_distance_domain(g, constraints, dd, 0.01, 0.01,
                                                                         ("curvature", "velocity")))   # MPG:408
        if res is None:
            return None
        gen = _route_batch(self.ctx.device)
        # a time step advances at least 0.1 * dt (MPG:581-582): rows <= total / (0.1 * dt); twice that, plus slack
        cap = int(self.total / (0.05 * float(dt))) + 4096
        for _ in range(3):
            tp = gen.time_profile(res, constraints, dt=dt, capacity_rows=cap, node_reverse=k["rev"][None])
            out = gen.insert_waits(res, tp, node_wait_time=k["wait"][None], action_points=[self._aps] if self.M else None,
                                   dt=dt, node_turn=k["turn"][None], node_reverse=k["rev"][None], constraints=constraints)
            head = torch.cat([out["counts"][0], res["flags"][:1]]).cpu().numpy()      # one small copy, the call's sync
            T, nn, na, flags = (int(v) for v in head)
            if flags & _lib.FLAG_BAD_ROUTE:
                raise IndexError("turn / wait before any profile row exists: the reference raises IndexError "
                                 "(motion_profile_generator.py:440,499)")
            if flags & _lib.FLAG_TRUNCATED:
                cap *= 4
                res["flags"].zero_()
                continue
            if flags:
                return None
            return (out["rows"][0, :T].cpu().numpy(), out["nodes_map"][0, :nn].cpu().numpy().astype(np.int64),
                    out["actions_map"][0, :na].cpu().numpy().astype(np.int64))
        return None

    def close(self):
        if getattr(self, "handle", None):
            self._L.vap_route_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self, order, ts):
        ts = np.ascontiguousarray(np.atleast_1d(ts), dtype=np.float64)
        out = np.empty((len(ts), 2))
        dp = C.POINTER(C.c_double)
        _lib.check(self._L.vap_route_eval(self.handle, int(order), len(ts), ts.ctypes.data_as(dp),
                                          out.ctypes.data_as(dp)), "vap_route_eval")
        return out

    def lookup(self, what, xs):
        xs = np.ascontiguousarray(np.atleast_1d(xs), dtype=np.float64)
        out = np.empty(len(xs))
        dp = C.POINTER(C.c_double)
        _lib.check(self._L.vap_route_lookup(self.handle, int(what), len(xs), xs.ctypes.data_as(dp),
                                            out.ctypes.data_as(dp)), "vap_route_lookup")
        return out

    # -- the batch kernels with B = 1 ------------------------------------------------------------------------------
    # The route's sampling, limits, velocity pass, time-domain resample and event insertion are what the batched entry
    # points do for B routes (vap_profile_routes -> vap_route_limits + vap_velocity_pass_limits -> vap_time_profile_routes
    # -> vap_time_insert_events; the same goldens of the real reference pin both, tests/test_gpu_routes_batch.py): a
    # single GUI route runs them as a batch of one — 0.65 ms for config 1 where the one-lane vap_route_* kernels
    # (statement-by-statement, the in-library reference) take 3.2 ms.  Routes whose tables have other sizes than the
    # reference's defaults (set_table_sizes) and callers that want the per-sample parameter row stay on vap_route_*.
    def _default_tables(self):
        return self.use_batch_kernels and (self.lut_samples, self.samples_per_node) == (_lib.LUT_SAMPLES, _lib.LUT_SAMPLES)

    def _sample_count(self, dd):
        n = C.c_int()
        _lib.check(self._L.vap_route_sample_count(self.handle, float(dd), C.byref(n)), "vap_route_sample_count")
        return n.value

    def _batched_distance_domain(self, gen, constraints, dd, start_vel, end_vel, want):
        k, W = self._keep, self.W
        wp = torch.tensor(k["wp"][None], dtype=torch.float64, device=gen.device)
        has_tan = bool(np.isfinite(k["tan"]).any())
        res = gen.profile_routes(wp, node_reverse=k["rev"][None], node_turn=k["turn"][None],
                                 node_tangent=k["tan"][None] if has_tan else None,
                                 node_magnitudes=k["mag"][None] if has_tan else None, constraints=constraints, dd=float(dd),
                                 start_vel=start_vel, end_vel=end_vel, want=want, capacity=self._sample_count(dd) + 9)
        if self._nodes_have_limits:
            gen.apply_node_limits(res, constraints, node_max_velocity=k["maxv"][None], node_stop=k["stop"][None],
                                  node_max_acceleration=k["maxa"][None], action_points=[self._aps] if self.M else None,
                                  start_vel=start_vel, end_vel=end_vel)
        return res

    def forward_backward(self, constraints, dd, start_vel, end_vel, want=("velocity",)):
        if tuple(want) == ("velocity",) and self._default_tables():
            res = self._try_batched(lambda gen: self._batched_distance_domain(gen, constraints, dd, start_vel, end_vel,
                                                                                ("curvature", "velocity")))
            if res is not None and int(res["flags"][0].item()) == 0:
                N = int(res["meta"][0, 3].item())
                return {"velocity": res["velocity"][0, :N].cpu().numpy()}, N
        N = self._sample_count(dd)
        n = C.c_int()
        dp = C.POINTER(C.c_double)
        names = ("t", "x", "y", "heading", "curvature", "velocity")
        bufs = {k: (np.empty(N) if k in want else None) for k in names}
        c = _lib.make_constraints(constraints)
        args = [bufs[k].ctypes.data_as(dp) if bufs[k] is not None else None for k in names]
        _lib.check(self._L.vap_route_forward_backward(self.handle, C.byref(c), float(dd), float(start_vel),
                                                      float(end_vel), N, C.byref(n), *args), "vap_route_forward_backward")
        return {k: v for k, v in bufs.items() if v is not None}, N

    def motion_profile(self, constraints, dt, dd):
        if self._default_tables():
            got = self._batched_motion_profile(constraints, dt, dd)
            if got is not None:
                return got
        c = _lib.make_constraints(constraints)
        cap = int(self.total / 0.0005) + 4096
        dp, lp = C.POINTER(C.c_double), C.POINTER(C.c_long)
        for _ in range(4):
            rows = np.empty((cap, 8))
            nmap = np.zeros(self.W + 2, dtype=np.int64)
            amap = np.zeros(self.M + 2, dtype=np.int64)
            T, nn, na = C.c_long(), C.c_int(), C.c_int()
            st = self._L.vap_route_motion_profile(self.handle, C.byref(c), float(dt), float(dd), cap,
                                                  rows.ctypes.data_as(dp), C.byref(T), nmap.ctypes.data_as(lp),
                                                  C.byref(nn), amap.ctypes.data_as(lp), C.byref(na))
            if st == _lib.VAP_ERR_CAPACITY:
                cap *= 4
                continue
            if st == _lib.VAP_ERR_INVALID:
                raise IndexError(self._L.vap_last_error().decode())   # MPG:440 / 499 in the reference
            _lib.check(st, "vap_route_motion_profile")
            return rows[:T.value], nmap[:nn.value], amap[:na.value]
        raise RuntimeError("motion profile did not fit the row buffer")
