"""ctypes binding of libvap.so (include/vap.h).  The ONLY compute backend of this package.

There is no CPU fallback: if the shared library is missing or no HIP device is visible, every
entry point raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C vexautonomousplanner_amd/csrc``.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvap.so")

VAP_OK = 0
VAP_ERR_INVALID = -1
VAP_ERR_NO_DEVICE = -2
VAP_ERR_HIP = -3
VAP_ERR_UNFITTED = -4
VAP_ERR_CAPACITY = -5
VAP_ERR_UNSUPPORTED = -6
VAP_F32 = 0
VAP_F64 = 1
FLAG_DEGENERATE = 1
FLAG_TRUNCATED = 2
FLAG_NOCONVERGE = 4
FLAG_BAD_ROUTE = 8
T_FIT, T_LUT, T_SAMPLE, T_VELOCITY, T_TOTAL, T_COUNT = 0, 1, 2, 3, 4, 8
OPT_VELOCITY_KERNEL = 0
OPT_TIME_DOMAIN_RESIDUAL = 3
OPT_F32_RECURRENCE = 1
OPT_TIME_KERNEL = 4
TIME_KERNEL_AUTO, TIME_KERNEL_LANE, TIME_KERNEL_QUAD, TIME_KERNEL_FUSED = 0, 1, 2, 3
RECURRENCE_F64, RECURRENCE_F32 = 0, 1
VELOCITY_AUTO, VELOCITY_SEQ_LITERAL, VELOCITY_SEQ_FAST, VELOCITY_RELAX = 0, 1, 2, 3
VELOCITY_RELAX_BLOCK, VELOCITY_RELAX_WAVE = 4, 5
VELOCITY_LANES, VELOCITY_LANES_16, VELOCITY_LANES_32, VELOCITY_LANES_64 = 6, 7, 8, 9
VELOCITY_RELAX_ROUNDS = 10
LUT_SAMPLES = 1000
SAMPLES_PER_NODE = 1000

# every symbol include/vap.h declares; tests check the library exports exactly these
EXPORTS = (
    "vap_version", "vap_status_string", "vap_last_error", "vap_device_count", "vap_ctx_create",
    "vap_ctx_destroy", "vap_ctx_set_stream", "vap_ctx_synchronize", "vap_ctx_set_option",
    "vap_ctx_set_timing",
    "vap_last_timing", "vap_fit", "vap_build_lut", "vap_sample", "vap_velocity_pass",
    "vap_time_profile", "vap_profile_batch", "vap_profile_batch_host", "vap_eval_host", "vap_basis_host", "vap_lookup_host",
    "vap_route_create", "vap_route_destroy", "vap_route_info", "vap_route_set_table_sizes", "vap_route_table_sizes", "vap_route_get_splines", "vap_route_eval",
    "vap_route_lookup", "vap_route_sample_count", "vap_route_forward_backward", "vap_route_motion_profile",
    "vap_grid_distances", "vap_route_limits", "vap_velocity_pass_limits", "vap_time_insert_waits", "vap_fit_ex",
    "vap_profile_routes", "vap_time_profile_routes", "vap_time_insert_events", "vap_limit_rows_dtype",
)


class Constraints(C.Structure):
    """vap_constraints == motion_profile_generator.Constraints field order (MPG:14-21)."""
    _fields_ = [("max_vel", C.c_double), ("max_acc", C.c_double), ("max_dec", C.c_double),
                ("friction_coef", C.c_double), ("max_jerk", C.c_double), ("track_width", C.c_double)]


ip = C.POINTER(C.c_int)


class RouteDesc(C.Structure):
    """vap_route_desc (include/vap.h)."""
    _fields_ = [("n_nodes", C.c_int), ("waypoints", C.POINTER(C.c_double)), ("is_reverse", C.POINTER(C.c_int)),
                ("turn", C.POINTER(C.c_double)), ("stop", C.POINTER(C.c_int)), ("wait_time", C.POINTER(C.c_double)),
                ("max_velocity", C.POINTER(C.c_double)), ("max_acceleration", C.POINTER(C.c_double)),
                ("tangent", C.POINTER(C.c_double)), ("magnitudes", C.POINTER(C.c_double)),
                ("n_actions", C.c_int), ("ap_t", C.POINTER(C.c_double)), ("ap_stop", C.POINTER(C.c_int)),
                ("ap_wait_time", C.POINTER(C.c_double)), ("ap_max_velocity", C.POINTER(C.c_double)),
                ("ap_max_acceleration", C.POINTER(C.c_double))]


class VapError(RuntimeError):
    def __init__(self, status, where, detail):
        self.status = status
        super().__init__(f"{where}: {detail} (vap_status {status})")


_lib = None
vp = C.c_void_p
dp = C.POINTER(C.c_double)
u32p = C.POINTER(C.c_uint32)


def lib():
    """Load libvap.so once.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `make -C vexautonomousplanner_amd/csrc`).  This package has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; load it FIRST so libvap.so binds to the same HIP
    # runtime instance as the tensors it is handed (two runtimes in one process do not share
    # devices or streams).  A plain C caller links libvap.so against /opt/rocm as usual.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.vap_version.restype = C.c_int
    L.vap_status_string.restype = C.c_char_p
    L.vap_status_string.argtypes = [C.c_int]
    L.vap_last_error.restype = C.c_char_p
    L.vap_device_count.restype = C.c_int
    L.vap_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vap_ctx_destroy.argtypes = [vp]
    L.vap_ctx_set_stream.argtypes = [vp, vp]
    L.vap_ctx_synchronize.argtypes = [vp]
    L.vap_ctx_set_timing.argtypes = [vp, C.c_int]
    L.vap_ctx_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.vap_last_timing.argtypes = [vp, C.POINTER(C.c_float)]
    L.vap_fit.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.vap_fit_ex.argtypes = [vp, C.c_int, C.c_int, C.c_int] + [vp] * 13
    L.vap_build_lut.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.vap_sample.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp, vp,
                             vp, vp, vp, vp, vp, vp]
    L.vap_velocity_pass.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(Constraints),
                                    C.c_double, C.c_double, vp, vp, vp, vp, vp, vp]
    L.vap_profile_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp,
                                    C.POINTER(Constraints), C.c_double, C.c_double,
                                    vp, vp, vp, vp, vp, vp, vp]
    L.vap_profile_batch_host.argtypes = L.vap_profile_batch.argtypes
    L.vap_profile_routes.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp, vp,
                                     C.POINTER(Constraints), C.c_double, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp]
    L.vap_time_profile.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp,
                                   C.POINTER(Constraints), C.c_double, C.c_int, vp, vp, vp, vp]
    L.vap_route_limits.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [vp] * 9 + [C.POINTER(Constraints), C.c_double] + [vp] * 6
    L.vap_velocity_pass_limits.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(Constraints), C.c_double, C.c_double,
                                           vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.vap_limit_rows_dtype.argtypes = [vp, C.c_int]
    L.vap_time_insert_waits.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double] + [vp] * 14
    L.vap_time_profile_routes.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(Constraints), C.c_double,
                                          C.c_int, vp, vp, vp, vp, vp]
    L.vap_time_insert_events.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(Constraints)] + [vp] * 14
    L.vap_eval_host.argtypes = [vp, C.c_int, dp, C.c_double, C.c_int, C.c_int, dp, dp]
    L.vap_basis_host.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.vap_lookup_host.argtypes = [vp, C.c_int, dp, C.c_double, dp, C.c_int, C.c_int, dp, dp]
    lp = C.POINTER(C.c_long)
    L.vap_route_create.argtypes = [vp, C.POINTER(RouteDesc), C.POINTER(vp)]
    L.vap_route_destroy.argtypes = [vp]
    L.vap_route_info.argtypes = [vp, ip, dp]
    L.vap_route_set_table_sizes.argtypes = [vp, C.c_int, C.c_int]
    L.vap_route_table_sizes.argtypes = [vp, ip, ip]
    L.vap_route_get_splines.argtypes = [vp, ip, ip, dp, dp, dp, dp, dp]
    L.vap_route_eval.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.vap_route_lookup.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.vap_route_sample_count.argtypes = [vp, C.c_double, ip]
    L.vap_grid_distances.argtypes = [C.c_double, C.c_double, C.c_long, dp, C.POINTER(C.c_long)]
    L.vap_route_forward_backward.argtypes = [vp, C.POINTER(Constraints), C.c_double, C.c_double, C.c_double,
                                             C.c_int, ip, dp, dp, dp, dp, dp, dp]
    L.vap_route_motion_profile.argtypes = [vp, C.POINTER(Constraints), C.c_double, C.c_double, C.c_long, dp, lp,
                                           lp, ip, lp, ip]
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("vap_version", "vap_device_count"):
            pass
    _lib = L
    return L


def check(status, where):
    if status != VAP_OK:
        L = lib()
        detail = L.vap_last_error().decode() or L.vap_status_string(status).decode()
        raise VapError(status, where, detail)


class Context:
    """One device + stream + scratch arena (vap_ctx).  Not thread-safe; make one per thread."""

    def __init__(self, device=0):
        self._L = lib()
        h = vp()
        check(self._L.vap_ctx_create(int(device), C.byref(h)), "vap_ctx_create")
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self._L.vap_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        """stream_ptr: a hipStream_t as int (0 = HIP's default stream); None = the context's own stream."""
        arg = vp(-1 & 0xFFFFFFFFFFFFFFFF) if stream_ptr is None else vp(int(stream_ptr))
        check(self._L.vap_ctx_set_stream(self.handle, arg), "vap_ctx_set_stream")

    def synchronize(self):
        check(self._L.vap_ctx_synchronize(self.handle), "vap_ctx_synchronize")

    def set_option(self, option, value):
        check(self._L.vap_ctx_set_option(self.handle, int(option), int(value)), "vap_ctx_set_option")

    def set_timing(self, enabled=True):
        check(self._L.vap_ctx_set_timing(self.handle, int(bool(enabled))), "vap_ctx_set_timing")

    def last_timing(self):
        ms = (C.c_float * T_COUNT)()
        check(self._L.vap_last_timing(self.handle, ms), "vap_last_timing")
        return {"fit": ms[T_FIT], "lut": ms[T_LUT], "sample": ms[T_SAMPLE],
                "velocity": ms[T_VELOCITY], "total": ms[T_TOTAL]}


_default_ctx = {}


def default_context(device=0):
    """Process-wide context per device (what the single-path drop-in classes use)."""
    ctx = _default_ctx.get(device)
    if ctx is None or ctx.handle is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx


def make_constraints(c):
    """Accept a Constraints-like object (attributes) or a 6-sequence."""
    if isinstance(c, Constraints):
        return c
    if hasattr(c, "max_vel"):
        return Constraints(float(c.max_vel), float(c.max_acc), float(c.max_dec),
                           float(c.friction_coef), float(c.max_jerk), float(c.track_width))
    v = [float(x) for x in c]
    if len(v) != 6:
        raise ValueError("constraints need 6 values: max_vel,max_acc,max_dec,friction_coef,"
                         "max_jerk,track_width")
    return Constraints(*v)
