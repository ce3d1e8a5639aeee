// vap_api.hip — the C-ABI of libvap.so (declared in include/vap.h).
//
// Host-side runtime: context (device + stream + scratch arena + stage timers) and the entry points
// that sequence the kernels of vap_kernels.hip.  There is deliberately no CPU implementation behind
// any entry point: without a HIP device every call fails with VAP_ERR_NO_DEVICE.
#include <cstdlib>
#include <cstring>

#include "vap_internal.h"
#include "vap_kernels.h"

namespace {
thread_local std::string g_last_error;
}  // namespace

int vap_fail(int status, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

int vap_set_device(vap_ctx *ctx)
{
    if (!ctx) return vap_fail(VAP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return VAP_OK;
}

namespace {

struct StageTimer {
    vap_ctx *c;
    explicit StageTimer(vap_ctx *ctx) : c(ctx)
    {
        if (c->timing) (void)hipEventRecord(c->ev[0], c->stream);
    }
    void mark(int slot)
    {
        if (c->timing) (void)hipEventRecord(c->ev[slot + 1], c->stream);
    }
};

size_t esz(vap_dtype dt) { return dt == VAP_F64 ? 8 : 4; }

int check_shape(int B, int W, int S)
{
    if (B < 1) return vap_fail(VAP_ERR_INVALID, "batch must be >= 1 (got %d)", B);
    if (W < 2) return vap_fail(VAP_ERR_INVALID, "a path needs at least 2 waypoints (got %d)", W);  // SM:50-51
    if (W > vap::kMaxWaypoints) return vap_fail(VAP_ERR_UNSUPPORTED, "W=%d exceeds %d", W, vap::kMaxWaypoints);
    if (S < 2) return vap_fail(VAP_ERR_INVALID, "sample capacity must be >= 2 (got %d)", S);
    return VAP_OK;
}


constexpr int kLanesMinPaths = 2048;   // AUTO: batches from here on take the lane-per-path kernel (fp64 recurrence)

// K5 dispatch.  auto: register-resident relaxation whenever the row fits, else the sequential sweep.
// f64 = arithmetic type of the recurrence = type of the curv / dth rows; io64 = type of the caller's rows (vcap,
// acc, vel).  f64 && !io64: the fp64 recurrence behind fp32 outputs (VAP_F32 with VAP_RECURRENCE_F64).
int run_velocity(vap_ctx *ctx, bool f64, bool io64, int B, int S, const double cc[6], double sv, double ev, const double *meta,
                 const void *curv, const void *dth, const void *vcap, const vap::AccRowsV &acc, void *vel, uint32_t *flags)
{
    int mode = ctx->velocity_kernel;
    ctx->vres_for = nullptr;
    // fp32 rows behind the fp64 recurrence: what the fp32 row lost of the fp64 velocities stays on the context as an fp32
    // residual row (VAP_OPT_TIME_DOMAIN_RESIDUAL, on by default) — the lane-per-path kernel writes it itself, the others
    // leave an fp64 row in scratch, converted here.  Always the residual form: the time domain then integrates the caller's
    // row AS IT IS at that call plus a term below its rounding, so an edited row is integrated as edited and a reused
    // address costs at most that rounding.
    const bool want_hi = f64 && !io64 && ctx->keep_residual;
    auto keep_res = [&]() {
        ctx->vres_for = vel;
        ctx->vres_B = B;
        ctx->vres_S = S;
    };
    auto keep_hi = [&](const void *rows64) -> int {
        VAP_TRY(ctx->ensure(ctx->vres, (size_t)B * S * sizeof(float)));
        HIP_TRY(vap::launch_velocity_residual(ctx->stream, (size_t)B * S, (const double *)rows64, (const float *)vel, (float *)ctx->vres.ptr));
        keep_res();
        return VAP_OK;
    };
    // per-sample initial velocities: the register-resident relaxation kernel takes them, the two-level one for
    // long rows and the wave-per-path variant do not (the sequential sweep does)
    const int relax_limit = acc.fwd ? vap::velocity_relax_acc_max_samples(f64) : vap::velocity_relax_max_samples(f64, vcap != nullptr);
    // fp64 recurrence, many paths: a wavefront of paths (K5w) — its time does not grow with the batch up to 256
    // workgroups, the relaxation kernel's does (one path per CU at a time for rows of ~10^4 samples)
    const bool lanes_by_auto = mode == VAP_VELOCITY_AUTO && f64 && B >= kLanesMinPaths;
    if (lanes_by_auto) mode = VAP_VELOCITY_LANES;
    if (mode >= VAP_VELOCITY_LANES && mode <= VAP_VELOCITY_LANES_64) {
        if (!f64) return vap_fail(VAP_ERR_UNSUPPORTED, "the lane-per-path velocity kernel runs the fp64 recurrence only");
        void *ufwd = nullptr;
        float *vres = nullptr;
        if (!io64) {
            VAP_TRY(ctx->ensure(ctx->ufwd, (size_t)B * S * 8));
            ufwd = ctx->ufwd.ptr;
            if (want_hi) {
                VAP_TRY(ctx->ensure(ctx->vres, (size_t)B * S * sizeof(float)));
                vres = (float *)ctx->vres.ptr;
            }
        }
        const int group = mode == VAP_VELOCITY_LANES ? 0 : (mode == VAP_VELOCITY_LANES_16 ? 16 : (mode == VAP_VELOCITY_LANES_32 ? 32 : 64));
        const hipError_t le = vap::launch_velocity_lanes(ctx->stream, io64, B, S, cc, sv, ev, meta, curv, dth, vcap, acc, vel, ufwd, group, vres);
        if (le == hipSuccess) {
            if (want_hi) keep_res();
            return VAP_OK;
        }
        // AUTO chose the kernel (it needs ~100 KB of dynamic LDS and rows addressable by 32-bit offsets): if it cannot be
        // launched here, the kernels that served such batches before it take the call; a kernel the caller asked for fails
        if (!lanes_by_auto) return vap_fail(VAP_ERR_HIP, "the lane-per-path velocity kernel could not be launched: %s", hipGetErrorString(le));
        (void)hipGetLastError();
        mode = VAP_VELOCITY_AUTO;
    }
    if (mode == VAP_VELOCITY_AUTO)
        mode = (vcap && S > relax_limit) ? VAP_VELOCITY_SEQ_FAST : VAP_VELOCITY_RELAX;
    const int forced = mode;
    if (mode == VAP_VELOCITY_RELAX_BLOCK || mode == VAP_VELOCITY_RELAX_WAVE || mode == VAP_VELOCITY_RELAX_ROUNDS) mode = VAP_VELOCITY_RELAX;
    if (mode == VAP_VELOCITY_RELAX) {
        if (vcap && (forced == VAP_VELOCITY_RELAX_WAVE || S > relax_limit))
            return vap_fail(VAP_ERR_UNSUPPORTED, "per-sample limits: rows up to %d samples in the relaxation kernel, or the sequential sweep",
                            relax_limit);
        if (forced == VAP_VELOCITY_RELAX_WAVE && (f64 || S > vap::velocity_relax_max_samples(false)))
            return vap_fail(VAP_ERR_UNSUPPORTED, "wave-per-path kernel: fp32 recurrence, rows up to %d samples", vap::velocity_relax_max_samples(false));
        // One wave per path (sequential windows) keeps 8 paths resident per CU instead of 2, but measured
        // 2x slower than the workgroup-per-path kernel on config 3 (every wave is then busy every round and
        // two latency-bound waves per SIMD slow each other down): kept selectable, not the default.
        const bool use_wave = forced == VAP_VELOCITY_RELAX_WAVE;
        if (use_wave) {
            // many paths: one wave per path keeps 8 paths resident per CU
            VAP_TRY(ctx->ensure(ctx->ufwd, (size_t)B * S * 4));
            VAP_TRY(ctx->ensure(ctx->lstate, vap::velocity_windows_state_bytes(B, S)));
            VAP_TRY(ctx->ensure(ctx->lcount, sizeof(int) * ((size_t)B + 64)));
            HIP_TRY(vap::launch_velocity_windows(ctx->stream, B, S, cc, sv, ev, meta, curv, dth, vel, flags, ctx->ufwd.ptr,
                                                 ctx->lstate.ptr, (int *)ctx->lcount.ptr));
        } else if (S <= vap::velocity_relax_max_samples(f64, vcap != nullptr)) {
            void *vhi = nullptr;
            if (want_hi) {
                VAP_TRY(ctx->ensure(ctx->ufwd, (size_t)B * S * 8));
                vhi = ctx->ufwd.ptr;
            }
            HIP_TRY(vap::launch_velocity_relax(ctx->stream, f64, io64, B, S, cc, sv, ev, meta, curv, dth, vcap, acc, vel, flags, vhi));
            if (want_hi) VAP_TRY(keep_hi(vhi));
        } else {
            // long rows: two-level relaxation; its scratch row holds the forward values until the backward sweep has
            // read them, so the fp64 velocities get a row of their own.  Interfaces between super-chunks are handed on
            // inside one launch per direction (look-back); RELAX_ROUNDS: one launch per super-round, checked on the host
            const bool rounds = forced == VAP_VELOCITY_RELAX_ROUNDS;
            VAP_TRY(ctx->ensure(ctx->ufwd, (size_t)B * S * (f64 ? 8 : 4)));
            VAP_TRY(ctx->ensure(ctx->lstate, rounds ? vap::velocity_long_state_bytes(f64, B, S) : vap::velocity_chase_state_bytes(f64, B, S)));
            VAP_TRY(ctx->ensure(ctx->lcount, rounds ? vap::velocity_long_counter_bytes(f64, B, S) : vap::velocity_chase_counter_bytes(B)));
            void *vhi = nullptr;
            if (want_hi) {
                VAP_TRY(ctx->ensure(ctx->vhi, (size_t)B * S * 8));
                vhi = ctx->vhi.ptr;
            }
            if (rounds)
                HIP_TRY(vap::launch_velocity_long(ctx->stream, f64, io64, B, S, cc, sv, ev, meta, curv, dth, vel, flags,
                                                  ctx->ufwd.ptr, ctx->lstate.ptr, (int *)ctx->lcount.ptr, vhi));
            else
                HIP_TRY(vap::launch_velocity_chase(ctx->stream, f64, io64, B, S, cc, sv, ev, meta, curv, dth, vel, flags,
                                                   ctx->ufwd.ptr, ctx->lstate.ptr, (int *)ctx->lcount.ptr, vhi));
            if (want_hi) VAP_TRY(keep_hi(vhi));
        }
    } else {
        void *usq = nullptr;
        if (f64 != io64) {   // the sweeps then run in a scratch row of the arithmetic type
            VAP_TRY(ctx->ensure(ctx->ufwd, (size_t)B * S * 8));
            usq = ctx->ufwd.ptr;
        }
        HIP_TRY(vap::launch_velocity_seq(ctx->stream, f64, io64, mode == VAP_VELOCITY_SEQ_FAST, B, S, cc, sv, ev, meta, curv,
                                         dth, vcap, acc, vel, usq));
        if (want_hi) VAP_TRY(keep_hi(usq));
    }
    return VAP_OK;
}

// The velocity row a time-domain entry point integrates: always the caller's, as it is now — plus, for an fp32 row whose
// velocity pass ran the fp64 recurrence in this context, the fp32 residual that pass left behind (row + residual = the
// fp64 velocity to 2^-48; MPG:566-584 integrates positions from the row, and an fp32 row alone moves a position by ~1e-7
// relative, now and then across a boundary of the reference's step lookup, SM:550-580).  The residual is below the
// row's own rounding, so a row the caller has edited since is integrated as edited.
const void *time_domain_velocity(vap_ctx *ctx, vap_dtype dt, int B, int S, const void *d_velocity, bool &is64, const float *&vres)
{
    is64 = dt == VAP_F64;
    vres = nullptr;
    if (dt == VAP_F32 && ctx->vres_for == d_velocity && ctx->vres.ptr && ctx->vres_B == B && ctx->vres_S == S)
        vres = (const float *)ctx->vres.ptr;
    return d_velocity;
}

}  // namespace

extern "C" {

int vap_version(void) { return VAP_VERSION; }

const char *vap_status_string(int status)
{
    switch (status) {
        case VAP_OK: return "ok";
        case VAP_ERR_INVALID: return "invalid argument";
        case VAP_ERR_NO_DEVICE: return "no HIP device";
        case VAP_ERR_HIP: return "HIP runtime error";
        case VAP_ERR_UNFITTED: return "path has not been fitted";
        case VAP_ERR_CAPACITY: return "output capacity too small";
        case VAP_ERR_UNSUPPORTED: return "unsupported size";
        default: return "unknown status";
    }
}

const char *vap_last_error(void) { return g_last_error.c_str(); }

int vap_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vap_ctx_create(int device, vap_ctx **out)
{
    if (!out) return vap_fail(VAP_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        return vap_fail(VAP_ERR_NO_DEVICE, "no HIP device visible: libvap has no CPU path");
    if (device < 0 || device >= n) return vap_fail(VAP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    vap_ctx *c = new vap_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return vap_fail(VAP_ERR_HIP, "hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    for (auto &e : c->ev) {
        if (hipEventCreate(&e) != hipSuccess) {
            (void)vap_ctx_destroy(c);   // releases the stream and the events created so far
            return vap_fail(VAP_ERR_HIP, "hipEventCreate failed");
        }
    }
    *out = c;
    return VAP_OK;
}

int vap_ctx_destroy(vap_ctx *ctx)
{
    if (!ctx) return VAP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    VapBuffer *bufs[] = {&ctx->sptab, &ctx->nspl, &ctx->k64, &ctx->dth64, &ctx->ufwd, &ctx->vhi, &ctx->vres, &ctx->lstate, &ctx->lcount, &ctx->seg, &ctx->power, &ctx->lut, &ctx->slopes, &ctx->aux, &ctx->runs, &ctx->meta, &ctx->dth, &ctx->flags, &ctx->small_in,
                      &ctx->small_out, &ctx->small_seg, &ctx->small_lut};
    for (VapBuffer *b : bufs)
        if (b->ptr) (void)hipFree(b->ptr);
    for (VapBuffer &b : ctx->io)
        if (b.ptr) (void)hipFree(b.ptr);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return VAP_OK;
}

int vap_ctx_set_stream(vap_ctx *ctx, void *hip_stream)
{
    if (!ctx) return vap_fail(VAP_ERR_INVALID, "null context");
    ctx->stream = hip_stream == VAP_STREAM_OWN ? ctx->own_stream : (hipStream_t)hip_stream;
    return VAP_OK;
}

int vap_ctx_synchronize(vap_ctx *ctx)
{
    VAP_TRY(vap_set_device(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VAP_OK;
}

int vap_ctx_set_option(vap_ctx *ctx, int option, int value)
{
    if (!ctx) return vap_fail(VAP_ERR_INVALID, "null context");
    if (option == VAP_OPT_VELOCITY_KERNEL && value >= VAP_VELOCITY_AUTO && value <= VAP_VELOCITY_RELAX_ROUNDS) {
        ctx->velocity_kernel = value;
        return VAP_OK;
    }
    if (option == VAP_OPT_TIME_DOMAIN_RESIDUAL && (value == 0 || value == 1)) {
        ctx->keep_residual = value;
        ctx->vres_for = nullptr;
        return VAP_OK;
    }
    if (option == VAP_OPT_TIME_KERNEL && value >= VAP_TIME_KERNEL_AUTO && value <= VAP_TIME_KERNEL_FUSED) {
        ctx->time_kernel = value;
        return VAP_OK;
    }
    if (option == VAP_OPT_F32_RECURRENCE && (value == VAP_RECURRENCE_F64 || value == VAP_RECURRENCE_F32)) {
        ctx->f32_recurrence = value;
        ctx->rows_valid = false;   // rows left by an earlier call belong to the other mode
        return VAP_OK;
    }
    return vap_fail(VAP_ERR_INVALID, "unknown option %d / value %d", option, value);
}

int vap_ctx_set_timing(vap_ctx *ctx, int enabled)
{
    if (!ctx) return vap_fail(VAP_ERR_INVALID, "null context");
    ctx->timing = enabled != 0;
    return VAP_OK;
}

int vap_last_timing(vap_ctx *ctx, float ms[VAP_T_COUNT])
{
    VAP_TRY(vap_set_device(ctx));
    if (!ms) return vap_fail(VAP_ERR_INVALID, "null output");
    for (int i = 0; i < VAP_T_COUNT; i++) ms[i] = 0.f;
    if (!ctx->timing) return VAP_OK;
    HIP_TRY(hipEventSynchronize(ctx->ev[VAP_T_VELOCITY + 1]));
    float total = 0.f;
    for (int i = 0; i <= VAP_T_VELOCITY; i++) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, ctx->ev[i], ctx->ev[i + 1]));
        ms[i] = t;
        total += t;
    }
    ms[VAP_T_TOTAL] = total;
    return VAP_OK;
}

int vap_fit(vap_ctx *ctx, vap_dtype dt, int B, int W, const void *d_waypoints, const double *d_tangent_in,
            const double *d_tangent_out, double *d_segments, double *d_segment_lengths, double *d_meta,
            uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, 2));
    if (!d_waypoints || !d_segments || !d_meta) return vap_fail(VAP_ERR_INVALID, "null buffer");
    HIP_TRY(vap::launch_fit(ctx->stream, dt == VAP_F64, B, W, d_waypoints, d_tangent_in, d_tangent_out,
                            d_segments, nullptr, d_segment_lengths, d_meta, d_flags));
    return VAP_OK;
}

int vap_fit_ex(vap_ctx *ctx, vap_dtype dt, int B, int W, const void *d_waypoints, const double *d_tangent_in,
               const double *d_tangent_out, const double *d_first_derivatives, const double *d_second_derivatives,
               const double *d_starting_tangent, const double *d_ending_tangent, double *d_segments,
               double *d_segment_lengths, double *d_first_out, double *d_second_out, double *d_meta, uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, 2));
    if (!d_waypoints || !d_segments || !d_meta) return vap_fail(VAP_ERR_INVALID, "null buffer");
    HIP_TRY(vap::launch_fit(ctx->stream, dt == VAP_F64, B, W, d_waypoints, d_tangent_in, d_tangent_out,
                            d_segments, nullptr, d_segment_lengths, d_meta, d_flags, d_first_derivatives,
                            d_second_derivatives, d_starting_tangent, d_ending_tangent, d_first_out, d_second_out));
    return VAP_OK;
}

int vap_build_lut(vap_ctx *ctx, int B, int W, const double *d_segments, double *d_lut, double *d_meta,
                  uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, 2));
    if (!d_segments || !d_lut || !d_meta) return vap_fail(VAP_ERR_INVALID, "null buffer");
    HIP_TRY(vap::launch_lut(ctx->stream, B, W, d_segments, d_lut, nullptr, d_meta, d_flags));
    return VAP_OK;
}

int vap_sample(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, double dd, const double *d_segments,
               const double *d_lut, double *d_meta, void *d_x, void *d_y, void *d_heading, void *d_curvature,
               void *d_dtheta, uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, S));
    if (!d_segments || !d_lut || !d_meta) return vap_fail(VAP_ERR_INVALID, "null buffer");
    const size_t n_seg = (size_t)B * (W - 1);
    VAP_TRY(ctx->ensure(ctx->power, n_seg * vap::kCoefBlockDoubles * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->aux, (size_t)B * 4 * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->runs, (size_t)B * vap::kGridRunBlockDoubles * sizeof(double)));
    HIP_TRY(vap::launch_power(ctx->stream, (int)n_seg, d_segments, (double *)ctx->power.ptr));
    HIP_TRY(vap::launch_grid(ctx->stream, B, W, S, dd, d_meta, (double *)ctx->aux.ptr, (double *)ctx->runs.ptr, d_flags));
    const bool hi = dt == VAP_F32 && ctx->f32_recurrence == VAP_RECURRENCE_F64;
    if (hi) {
        VAP_TRY(ctx->ensure(ctx->k64, (size_t)B * S * sizeof(double)));
        VAP_TRY(ctx->ensure(ctx->dth64, (size_t)B * S * sizeof(double)));
    }
    HIP_TRY(vap::launch_sample(ctx->stream, dt == VAP_F64, B, W, S, (const double *)ctx->power.ptr, d_lut,
                               nullptr, d_meta, (const double *)ctx->aux.ptr,
                               (const double *)ctx->runs.ptr, d_x, d_y, d_heading, d_curvature, d_dtheta,
                               hi ? (double *)ctx->k64.ptr : nullptr, hi ? (double *)ctx->dth64.ptr : nullptr));
    ctx->grid_B = B;
    ctx->grid_W = W;
    ctx->grid_S = S;
    ctx->rows_valid = hi;
    ctx->rows_hi = hi;
    ctx->rows_dt = dt;
    ctx->route_NS = 0;
    return VAP_OK;
}

int vap_velocity_pass_limits(vap_ctx *ctx, vap_dtype dt, int B, int S, const vap_constraints *c, double start_vel,
                             double end_vel, const double *d_meta, const void *d_curvature, const void *d_dtheta,
                             const void *d_vcap, const void *d_acc_forward, const void *d_acc_backward,
                             const void *d_dec_backward, void *d_velocity, uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, 2, S));
    if (!c || !d_meta || !d_velocity) return vap_fail(VAP_ERR_INVALID, "null buffer");
    const bool any_acc = d_acc_forward || d_acc_backward || d_dec_backward;
    if (any_acc && !(d_acc_forward && d_acc_backward && d_dec_backward && d_vcap))
        return vap_fail(VAP_ERR_INVALID, "max_acceleration rows come as a set (forward, backward, dec) together with d_vcap");
    // The limit rows have the type of the recurrence they enter (vap_limit_rows_dtype): a limit such as 13.9 ft/s^2 rounded
    // to fp32 and then amplified by the fp64 recurrence left the 1e-5 bound.  An explicit fp32
    // d_dtheta makes this call an fp32 recurrence, which cannot take the fp64 limit rows of the default mode.
    if (dt == VAP_F32 && d_dtheta && (d_vcap || any_acc) && ctx->f32_recurrence == VAP_RECURRENCE_F64)
        return vap_fail(VAP_ERR_INVALID, "VAP_F32 with VAP_RECURRENCE_F64: the limit rows are fp64 and go with the context's fp64 rows "
                                         "(d_dtheta = NULL); for an fp32 recurrence on caller rows set VAP_OPT_F32_RECURRENCE first");
    bool r64 = dt == VAP_F64;
    if (!d_dtheta) {    // the rows the last sampling call of this shape and dtype left on the context
        if (!ctx->rows_valid || ctx->grid_B != B || ctx->grid_S != S || ctx->rows_dt != (int)dt)
            return vap_fail(VAP_ERR_INVALID, "d_dtheta is NULL and the context holds no rows of this shape and dtype (%d x %d, dtype %d; "
                            "the last sampling call left %s %d x %d, dtype %d)", B, S, (int)dt, ctx->rows_valid ? "rows of" : "no rows;",
                            ctx->grid_B, ctx->grid_S, ctx->rows_dt);
        if (ctx->rows_hi) {   // VAP_F32 with the fp64 recurrence: both rows come from the context, in fp64
            d_curvature = ctx->k64.ptr;
            d_dtheta = ctx->dth64.ptr;
            r64 = true;
        } else {
            d_dtheta = ctx->dth.ptr;
        }
    }
    if (!d_curvature) return vap_fail(VAP_ERR_INVALID, "null curvature row");
    // Quirk Q9: boundary_map always holds sample 0 (MPG:110), so the reference overwrites max_dec with
    // max_accels[0] — max_acc for a plain node — before the first forward step (MPG:194-196) and never
    // restores it until the pass returns: the backward sweep decelerates with max_acc.  (The time loop
    // does see the caller's max_dec, MPG:572-573 — vap_time_profile.)
    const double cc[6] = {c->max_vel, c->max_acc, c->max_acc, c->friction_coef, c->max_jerk, c->track_width};
    vap::AccRowsV acc;
    acc.fwd = d_acc_forward;
    acc.bwd = d_acc_backward;
    acc.dec = d_dec_backward;
    VAP_TRY(run_velocity(ctx, r64, dt == VAP_F64, B, S, cc, start_vel, end_vel, d_meta, d_curvature, d_dtheta, d_vcap, acc,
                         d_velocity, d_flags));
    return VAP_OK;
}

int vap_limit_rows_dtype(vap_ctx *ctx, vap_dtype dt)
{
    if (!ctx) return (int)dt;
    return (dt == VAP_F64 || ctx->f32_recurrence == VAP_RECURRENCE_F64) ? (int)VAP_F64 : (int)VAP_F32;
}

int vap_velocity_pass(vap_ctx *ctx, vap_dtype dt, int B, int S, const vap_constraints *c, double start_vel,
                      double end_vel, const double *d_meta, const void *d_curvature, const void *d_dtheta,
                      const void *d_vcap, void *d_velocity, uint32_t *d_flags)
{
    return vap_velocity_pass_limits(ctx, dt, B, S, c, start_vel, end_vel, d_meta, d_curvature, d_dtheta, d_vcap, nullptr,
                                    nullptr, nullptr, d_velocity, d_flags);
}

int vap_profile_batch(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, double dd, const void *d_waypoints,
                      const vap_constraints *c, double start_vel, double end_vel, void *d_x, void *d_y,
                      void *d_heading, void *d_curvature, void *d_velocity, double *d_meta, uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, S));
    if (!d_waypoints || !c || !d_velocity) return vap_fail(VAP_ERR_INVALID, "null buffer");
    const bool f64 = dt == VAP_F64;
    const size_t n_seg = (size_t)B * (W - 1), n_pts = (size_t)B * S;
    VAP_TRY(ctx->ensure(ctx->seg, n_seg * 12 * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->power, n_seg * vap::kCoefBlockDoubles * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->lut, (size_t)B * VAP_LUT_SAMPLES * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->aux, (size_t)B * 4 * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->runs, (size_t)B * vap::kGridRunBlockDoubles * sizeof(double)));
    // VAP_F32 with the fp64 recurrence (the default): the velocity pass reads fp64 curvature / |dtheta| rows
    const bool hi = !f64 && ctx->f32_recurrence == VAP_RECURRENCE_F64;
    if (hi) {
        VAP_TRY(ctx->ensure(ctx->k64, n_pts * sizeof(double)));
        VAP_TRY(ctx->ensure(ctx->dth64, n_pts * sizeof(double)));
    } else {
        VAP_TRY(ctx->ensure(ctx->dth, n_pts * esz(dt)));
    }
    double *meta = d_meta;
    if (!meta) {
        VAP_TRY(ctx->ensure(ctx->meta, (size_t)B * 4 * sizeof(double)));
        meta = (double *)ctx->meta.ptr;
    }
    uint32_t *flags = d_flags;
    if (!flags) {
        VAP_TRY(ctx->ensure(ctx->flags, (size_t)B * sizeof(uint32_t)));
        flags = (uint32_t *)ctx->flags.ptr;
    }
    void *curv = d_curvature;
    if (!curv && !hi) {
        VAP_TRY(ctx->ensure(ctx->io[7], n_pts * esz(dt)));
        curv = ctx->io[7].ptr;
    }
    // Quirk Q9: boundary_map always holds sample 0 (MPG:110), so the reference overwrites max_dec with
    // max_accels[0] — max_acc for a plain node — before the first forward step (MPG:194-196) and never
    // restores it until the pass returns: the backward sweep decelerates with max_acc.  (The time loop
    // does see the caller's max_dec, MPG:572-573 — vap_time_profile.)
    const double cc[6] = {c->max_vel, c->max_acc, c->max_acc, c->friction_coef, c->max_jerk, c->track_width};
    StageTimer tm(ctx);
    vap::GridArgs grid;     // the distance grid is defined in the tail of the table kernel
    grid.S = S;
    grid.dd = dd;
    grid.aux = (double *)ctx->aux.ptr;
    grid.runs = (double *)ctx->runs.ptr;
    if (vap::fit_lut_fusable(B, W)) {
        // fit and table of a path by one workgroup, one launch (VAP_T_FIT then reads 0, VAP_T_LUT the pair)
        tm.mark(VAP_T_FIT);
        HIP_TRY(vap::launch_fit_lut(ctx->stream, f64, B, W, d_waypoints, (double *)ctx->seg.ptr, (double *)ctx->power.ptr,
                                    (double *)ctx->lut.ptr, meta, flags, grid));
    } else {
        HIP_TRY(vap::launch_fit(ctx->stream, f64, B, W, d_waypoints, nullptr, nullptr, (double *)ctx->seg.ptr,
                                (double *)ctx->power.ptr, nullptr, meta, flags));
        tm.mark(VAP_T_FIT);
        HIP_TRY(vap::launch_lut(ctx->stream, B, W, (const double *)ctx->seg.ptr, (double *)ctx->lut.ptr,
                                nullptr, meta, flags, grid));   // (the sampling kernel forms the interval slopes itself)
    }
    tm.mark(VAP_T_LUT);
    {
        HIP_TRY(vap::launch_sample(ctx->stream, f64, B, W, S, (const double *)ctx->power.ptr, (const double *)ctx->lut.ptr, nullptr,
                                   meta, (const double *)ctx->aux.ptr, (const double *)ctx->runs.ptr, d_x, d_y, d_heading,
                                   curv, hi ? nullptr : ctx->dth.ptr, hi ? (double *)ctx->k64.ptr : nullptr,
                                   hi ? (double *)ctx->dth64.ptr : nullptr));
        tm.mark(VAP_T_SAMPLE);
        if (hi)
            VAP_TRY(run_velocity(ctx, true, false, B, S, cc, start_vel, end_vel, meta, ctx->k64.ptr, ctx->dth64.ptr, nullptr,
                                 vap::AccRowsV(), d_velocity, flags));
        else
            VAP_TRY(run_velocity(ctx, f64, f64, B, S, cc, start_vel, end_vel, meta, curv, ctx->dth.ptr, nullptr, vap::AccRowsV(),
                                 d_velocity, flags));
    }
    tm.mark(VAP_T_VELOCITY);
    ctx->last_B = B;
    ctx->last_W = W;
    ctx->grid_B = B;
    ctx->grid_W = W;
    ctx->grid_S = S;
    ctx->rows_valid = true;
    ctx->rows_hi = hi;
    ctx->rows_dt = dt;
    ctx->route_NS = 0;
    return VAP_OK;
}

int vap_profile_routes(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, double dd, int max_splines, const void *d_waypoints,
                       const int *d_node_reverse, const double *d_node_turn, const double *d_node_tangent,
                       const double *d_node_magnitudes, const vap_constraints *c, double start_vel, double end_vel,
                       void *d_x, void *d_y, void *d_heading, void *d_curvature, void *d_velocity, double *d_meta,
                       uint32_t *d_flags, int *d_spline_counts)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, S));
    if (!d_waypoints || !c || !d_velocity) return vap_fail(VAP_ERR_INVALID, "null buffer");
    if (max_splines < 1 || max_splines > W - 1) return vap_fail(VAP_ERR_INVALID, "max_splines must be in [1, W-1] (got %d)", max_splines);
    if (W > vap::kMaxRouteWaypoints)   // the route fit keeps 13 doubles and an int per node in LDS (160 KB per CU)
        return vap_fail(VAP_ERR_UNSUPPORTED, "batched routes: W=%d exceeds %d nodes (plain paths go to %d through vap_profile_batch)", W,
                        vap::kMaxRouteWaypoints, vap::kMaxWaypoints);
    if (d_node_tangent && !d_node_magnitudes) return vap_fail(VAP_ERR_INVALID, "node tangents come with their magnitudes");
    const bool f64 = dt == VAP_F64;
    const int NS = max_splines;
    const size_t n_seg = (size_t)B * (W - 1), n_pts = (size_t)B * S;
    VAP_TRY(ctx->ensure(ctx->seg, n_seg * 12 * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->power, n_seg * vap::kCoefBlockDoubles * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->lut, (size_t)B * NS * VAP_LUT_SAMPLES * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->sptab, (size_t)B * NS * 4 * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->nspl, (size_t)B * sizeof(int)));
    VAP_TRY(ctx->ensure(ctx->aux, (size_t)B * 4 * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->runs, (size_t)B * vap::kGridRunBlockDoubles * sizeof(double)));
    const bool hi = !f64 && ctx->f32_recurrence == VAP_RECURRENCE_F64;
    if (hi) {
        VAP_TRY(ctx->ensure(ctx->k64, n_pts * sizeof(double)));
        VAP_TRY(ctx->ensure(ctx->dth64, n_pts * sizeof(double)));
    } else {
        VAP_TRY(ctx->ensure(ctx->dth, n_pts * esz(dt)));
    }
    double *meta = d_meta;
    if (!meta) {
        VAP_TRY(ctx->ensure(ctx->meta, (size_t)B * 4 * sizeof(double)));
        meta = (double *)ctx->meta.ptr;
    }
    uint32_t *flags = d_flags;
    if (!flags) {
        VAP_TRY(ctx->ensure(ctx->flags, (size_t)B * sizeof(uint32_t)));
        flags = (uint32_t *)ctx->flags.ptr;
    }
    void *curv = d_curvature;
    if (!curv && !hi) {
        VAP_TRY(ctx->ensure(ctx->io[7], n_pts * esz(dt)));
        curv = ctx->io[7].ptr;
    }
    const double cc[6] = {c->max_vel, c->max_acc, c->max_acc, c->friction_coef, c->max_jerk, c->track_width};   // quirk Q9
    vap::RouteSplitInputs in;
    in.rev = d_node_reverse;
    in.turn = d_node_turn;
    in.tangent = d_node_tangent;
    in.mag = d_node_magnitudes;
    double *sptab = (double *)ctx->sptab.ptr;
    int *nspl = (int *)ctx->nspl.ptr;
    StageTimer tm(ctx);
    HIP_TRY(vap::launch_fit_routes(ctx->stream, f64, B, W, NS, d_waypoints, in, (double *)ctx->seg.ptr, (double *)ctx->power.ptr,
                                   nullptr, sptab, nspl, meta, flags));
    tm.mark(VAP_T_FIT);
    vap::RouteTables rt;
    rt.sptab = sptab;
    rt.nspl = nspl;
    rt.NS = NS;
    // No route of the batch has a split (max_splines == 1: tangent overrides at most): every route is one spline with
    // zero offsets, i.e. a plain path — the persistent sampling kernel of vap_profile_batch takes it, same rows bit for bit as that entry point and 3-4x faster than the
    // thread-per-sample kernel the concatenated tables need.
    const bool single = NS == 1;
    HIP_TRY(vap::launch_lut(ctx->stream, B, W, (const double *)ctx->seg.ptr, (double *)ctx->lut.ptr, nullptr, meta, flags,
                            vap::GridArgs(), rt));
    HIP_TRY(vap::launch_route_offsets(ctx->stream, B, W, NS, S, dd, (const double *)ctx->lut.ptr, sptab, nspl, meta,
                                      (double *)ctx->aux.ptr, (double *)ctx->runs.ptr, flags));
    tm.mark(VAP_T_LUT);
    if (single)
        HIP_TRY(vap::launch_sample(ctx->stream, f64, B, W, S, (const double *)ctx->power.ptr, (const double *)ctx->lut.ptr,
                                   nullptr, meta, (const double *)ctx->aux.ptr,
                                   (const double *)ctx->runs.ptr, d_x, d_y, d_heading, curv, hi ? nullptr : ctx->dth.ptr,
                                   hi ? (double *)ctx->k64.ptr : nullptr, hi ? (double *)ctx->dth64.ptr : nullptr));
    else
        HIP_TRY(vap::launch_sample_routes(ctx->stream, f64, B, W, NS, S, (const double *)ctx->power.ptr, (const double *)ctx->lut.ptr,
                                          sptab, nspl, meta, (const double *)ctx->aux.ptr, (const double *)ctx->runs.ptr, d_x, d_y,
                                          d_heading, curv, hi ? nullptr : ctx->dth.ptr, hi ? (double *)ctx->k64.ptr : nullptr,
                                          hi ? (double *)ctx->dth64.ptr : nullptr));
    tm.mark(VAP_T_SAMPLE);
    if (hi)
        VAP_TRY(run_velocity(ctx, true, false, B, S, cc, start_vel, end_vel, meta, ctx->k64.ptr, ctx->dth64.ptr, nullptr,
                             vap::AccRowsV(), d_velocity, flags));
    else
        VAP_TRY(run_velocity(ctx, f64, f64, B, S, cc, start_vel, end_vel, meta, curv, ctx->dth.ptr, nullptr, vap::AccRowsV(),
                             d_velocity, flags));
    tm.mark(VAP_T_VELOCITY);
    if (d_spline_counts) HIP_TRY(hipMemcpyAsync(d_spline_counts, nspl, sizeof(int) * (size_t)B, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->last_B = B;
    ctx->last_W = W;
    ctx->grid_B = B;
    ctx->grid_W = W;
    ctx->grid_S = S;
    ctx->rows_valid = true;
    ctx->rows_hi = hi;
    ctx->rows_dt = dt;
    ctx->route_NS = NS;
    return VAP_OK;
}

int vap_route_limits(vap_ctx *ctx, vap_dtype dt, int B, int W, int M, int S, const double *d_lut, const double *d_meta,
                     const double *d_node_max_velocity, const double *d_node_max_acceleration, const int *d_node_stop,
                     const double *d_action_t, const double *d_action_max_velocity, const double *d_action_max_acceleration,
                     const int *d_action_stop, const vap_constraints *c, double end_vel, void *d_vcap, void *d_acc_forward,
                     void *d_acc_backward, void *d_dec_backward, int *d_node_sample, int *d_action_sample)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, S));
    if (M < 0 || !d_meta || !d_vcap || !c || !(c->max_vel > 0) || !(c->max_acc > 0)) return vap_fail(VAP_ERR_INVALID, "bad argument");
    if (M > 0 && !d_action_t) return vap_fail(VAP_ERR_INVALID, "null action-point array");
    const bool any_acc = d_acc_forward || d_acc_backward || d_dec_backward;
    if (any_acc && !(d_acc_forward && d_acc_backward && d_dec_backward))
        return vap_fail(VAP_ERR_INVALID, "the max_acceleration outputs come as a set (forward, backward, dec)");
    if (ctx->grid_B != B || ctx->grid_W != W || ctx->grid_S != S || !ctx->runs.ptr || !ctx->aux.ptr)
        return vap_fail(VAP_ERR_INVALID, "no distance grid of this shape on the context (%d x %d x %d; the last sampling call left %d x %d x %d)",
                        B, W, S, ctx->grid_B, ctx->grid_W, ctx->grid_S);
    const double *lut = d_lut;
    if (!lut) {
        if (ctx->last_B != B || ctx->last_W != W || !ctx->lut.ptr)
            return vap_fail(VAP_ERR_INVALID, "d_lut is NULL and the context holds no table of this shape");
        lut = (const double *)ctx->lut.ptr;
    }
    // scratch: node samples, action samples, merged event list {sample, max_velocity, max_acceleration, stop}
    const size_t E = (size_t)(W > 2 ? W - 2 : 0) + M;
    const size_t n_int = (size_t)B * (W + M + 2 * E), n_dbl = (size_t)B * 2 * E;
    VAP_TRY(ctx->ensure(ctx->small_out, n_dbl * sizeof(double) + n_int * sizeof(int) + 64));
    double *ev_mv = (double *)ctx->small_out.ptr, *ev_ma = ev_mv + (size_t)B * E;
    int *node_k = (int *)(ev_ma + (size_t)B * E);
    int *ap_k = node_k + (size_t)B * W;
    int *ev_k = ap_k + (size_t)B * M;
    int *ev_stop = ev_k + (size_t)B * E;
    vap::LimitInputs in;
    in.node_mv = d_node_max_velocity;
    in.node_ma = d_node_max_acceleration;
    in.node_stop = d_node_stop;
    in.ap_t = d_action_t;
    in.ap_mv = d_action_max_velocity;
    in.ap_ma = d_action_max_acceleration;
    in.ap_stop = d_action_stop;
    in.max_vel = c->max_vel;
    in.max_acc = c->max_acc;
    in.end_vel = end_vel;
    vap::RouteTables rt;
    if (ctx->route_NS > 0) {   // the batch on the context is one of routes cut into splines: its own tables only
        if (d_lut) return vap_fail(VAP_ERR_INVALID, "a batch of routes (vap_profile_routes) is on the context: d_lut must be NULL");
        rt.sptab = (const double *)ctx->sptab.ptr;
        rt.nspl = (const int *)ctx->nspl.ptr;
        rt.NS = ctx->route_NS;
    }
    HIP_TRY(vap::launch_route_limits(ctx->stream, vap_limit_rows_dtype(ctx, dt) == VAP_F64, B, W, M, S, lut, d_meta, (const double *)ctx->aux.ptr,
                                     (const double *)ctx->runs.ptr, in, node_k, ap_k, ev_k, ev_mv, ev_ma, ev_stop, d_vcap,
                                     d_acc_forward, d_acc_backward, d_dec_backward, rt));
    if (d_node_sample) HIP_TRY(hipMemcpyAsync(d_node_sample, node_k, sizeof(int) * (size_t)B * W, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_action_sample && M > 0)
        HIP_TRY(hipMemcpyAsync(d_action_sample, ap_k, sizeof(int) * (size_t)B * M, hipMemcpyDeviceToDevice, ctx->stream));
    return VAP_OK;
}

int vap_time_profile(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, const double *d_segments, const double *d_lut,
                     const double *d_meta, const void *d_velocity, const vap_constraints *c, double time_step,
                     int capacity_rows, double *d_rows, int *d_counts, int *d_nodes_map, uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, S));
    if (!d_meta || !d_velocity || !c || !d_rows || !d_counts || !d_nodes_map) return vap_fail(VAP_ERR_INVALID, "null buffer");
    if (!(time_step > 0) || capacity_rows < 1) return vap_fail(VAP_ERR_INVALID, "time_step and capacity_rows must be positive");
    if ((d_segments == nullptr) != (d_lut == nullptr)) return vap_fail(VAP_ERR_INVALID, "pass both d_segments and d_lut, or neither");
    if (!d_segments) {
        if (ctx->last_B != B || ctx->last_W != W || !ctx->seg.ptr || !ctx->lut.ptr)
            return vap_fail(VAP_ERR_UNFITTED, "no tables of a %d x %d batch in this context (last vap_profile_batch: %d x %d)", B,
                            W, ctx->last_B, ctx->last_W);
        if (ctx->route_NS > 0)
            return vap_fail(VAP_ERR_UNSUPPORTED, "the batch on the context is one of split routes (vap_profile_routes): their "
                                                 "time domain goes through vap_time_profile_routes / vap_time_insert_events");
        d_segments = (const double *)ctx->seg.ptr;
        d_lut = (const double *)ctx->lut.ptr;
    }
    bool v64;
    const float *vres;
    const void *vrow = time_domain_velocity(ctx, dt, B, S, d_velocity, v64, vres);
    HIP_TRY(vap::launch_time_profile(ctx->stream, v64, B, W, S, d_segments, d_lut, d_meta, vrow, c->max_acc,
                                     c->max_dec, time_step, capacity_rows, d_rows, d_counts, d_nodes_map, d_flags, vap::RouteTables(),
                                     nullptr, vres, ctx->time_kernel));
    return VAP_OK;
}

int vap_time_insert_waits(vap_ctx *ctx, int B, int W, int M, int capacity_in, int capacity_out, double time_step,
                          const double *d_segments, const double *d_lut, const double *d_meta, const double *d_rows_in,
                          const int *d_counts_in, const int *d_nodes_map_in, const double *d_node_wait,
                          const double *d_action_t, const double *d_action_wait, double *d_rows_out, int *d_counts_out,
                          int *d_nodes_map_out, int *d_actions_map_out, uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, 2));
    if (M < 0 || capacity_in < 1 || capacity_out < 1 || !(time_step > 0)) return vap_fail(VAP_ERR_INVALID, "bad argument");
    if (!d_meta || !d_rows_in || !d_counts_in || !d_nodes_map_in || !d_rows_out || !d_counts_out || !d_nodes_map_out)
        return vap_fail(VAP_ERR_INVALID, "null buffer");
    if (M > 0 && (!d_action_t || !d_actions_map_out)) return vap_fail(VAP_ERR_INVALID, "null action-point array");
    if (d_rows_out == d_rows_in) return vap_fail(VAP_ERR_INVALID, "the rows move: d_rows_out must not be d_rows_in");
    if ((d_segments == nullptr) != (d_lut == nullptr)) return vap_fail(VAP_ERR_INVALID, "pass both d_segments and d_lut, or neither");
    if (!d_segments) {
        if (ctx->last_B != B || ctx->last_W != W || !ctx->seg.ptr || !ctx->lut.ptr)
            return vap_fail(VAP_ERR_UNFITTED, "no tables of a %d x %d batch in this context (last vap_profile_batch: %d x %d)", B,
                            W, ctx->last_B, ctx->last_W);
        if (ctx->route_NS > 0)
            return vap_fail(VAP_ERR_UNSUPPORTED, "the batch on the context is one of split routes (vap_profile_routes): their "
                                                 "time domain goes through vap_time_profile_routes / vap_time_insert_events");
        d_segments = (const double *)ctx->seg.ptr;
        d_lut = (const double *)ctx->lut.ptr;
    }
    HIP_TRY(vap::launch_time_waits(ctx->stream, B, W, M, capacity_in, capacity_out, time_step, d_segments, d_lut, d_meta,
                                   d_rows_in, d_counts_in, d_nodes_map_in, d_node_wait, d_action_t, d_action_wait, d_rows_out,
                                   d_counts_out, d_nodes_map_out, d_actions_map_out, d_flags));
    return VAP_OK;
}

int vap_time_profile_routes(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, const double *d_meta, const void *d_velocity,
                            const vap_constraints *c, double time_step, int capacity_rows, const int *d_node_reverse,
                            double *d_rows, int *d_counts, int *d_nodes_map, uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, S));
    if (!d_meta || !d_velocity || !c || !d_rows || !d_counts || !d_nodes_map) return vap_fail(VAP_ERR_INVALID, "null buffer");
    if (!(time_step > 0) || capacity_rows < 1) return vap_fail(VAP_ERR_INVALID, "time_step and capacity_rows must be positive");
    if (ctx->last_B != B || ctx->last_W != W || !ctx->seg.ptr || !ctx->lut.ptr)
        return vap_fail(VAP_ERR_UNFITTED, "no tables of a %d x %d batch in this context (last profile call: %d x %d)", B, W,
                        ctx->last_B, ctx->last_W);
    vap::RouteTables rt;
    if (ctx->route_NS > 0) {
        rt.sptab = (const double *)ctx->sptab.ptr;
        rt.nspl = (const int *)ctx->nspl.ptr;
        rt.NS = ctx->route_NS;
    }
    bool v64;
    const float *vres;
    const void *vrow = time_domain_velocity(ctx, dt, B, S, d_velocity, v64, vres);
    HIP_TRY(vap::launch_time_profile(ctx->stream, v64, B, W, S, (const double *)ctx->seg.ptr, (const double *)ctx->lut.ptr,
                                     d_meta, vrow, c->max_acc, c->max_dec, time_step, capacity_rows, d_rows, d_counts,
                                     d_nodes_map, d_flags, rt, d_node_reverse, vres, ctx->time_kernel));
    return VAP_OK;
}

int vap_time_insert_events(vap_ctx *ctx, int B, int W, int M, int capacity_in, int capacity_out, double time_step,
                           const vap_constraints *c, const double *d_meta, const double *d_rows_in, const int *d_counts_in,
                           const int *d_nodes_map_in, const double *d_node_wait, const double *d_node_turn,
                           const int *d_node_reverse, const double *d_action_t, const double *d_action_wait, double *d_rows_out,
                           int *d_counts_out, int *d_nodes_map_out, int *d_actions_map_out, uint32_t *d_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, 2));
    if (M < 0 || capacity_in < 1 || capacity_out < 1 || !(time_step > 0) || !c) return vap_fail(VAP_ERR_INVALID, "bad argument");
    if (!d_meta || !d_rows_in || !d_counts_in || !d_nodes_map_in || !d_rows_out || !d_counts_out || !d_nodes_map_out)
        return vap_fail(VAP_ERR_INVALID, "null buffer");
    if (M > 0 && (!d_action_t || !d_actions_map_out)) return vap_fail(VAP_ERR_INVALID, "null action-point array");
    if (d_rows_out == d_rows_in) return vap_fail(VAP_ERR_INVALID, "the rows move: d_rows_out must not be d_rows_in");
    if (d_node_turn && !(c->max_vel > 0 && c->max_acc > 0 && c->track_width > 0))
        return vap_fail(VAP_ERR_INVALID, "in-place turns need max_vel, max_acc and track_width");
    if (ctx->last_B != B || ctx->last_W != W || !ctx->seg.ptr || !ctx->lut.ptr)
        return vap_fail(VAP_ERR_UNFITTED, "no tables of a %d x %d batch in this context (last profile call: %d x %d)", B, W,
                        ctx->last_B, ctx->last_W);
    vap::RouteTables rt;
    if (ctx->route_NS > 0) {
        rt.sptab = (const double *)ctx->sptab.ptr;
        rt.nspl = (const int *)ctx->nspl.ptr;
        rt.NS = ctx->route_NS;
    }
    HIP_TRY(vap::launch_time_waits(ctx->stream, B, W, M, capacity_in, capacity_out, time_step, (const double *)ctx->seg.ptr,
                                   (const double *)ctx->lut.ptr, d_meta, d_rows_in, d_counts_in, d_nodes_map_in, d_node_wait,
                                   d_action_t, d_action_wait, d_rows_out, d_counts_out, d_nodes_map_out, d_actions_map_out, d_flags, rt,
                                   d_node_turn, d_node_reverse, c->max_vel, c->max_acc, c->track_width));
    return VAP_OK;
}

int vap_profile_batch_host(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, double dd, const void *h_waypoints,
                           const vap_constraints *c, double start_vel, double end_vel, void *h_x, void *h_y,
                           void *h_heading, void *h_curvature, void *h_velocity, double *h_meta,
                           uint32_t *h_flags)
{
    VAP_TRY(vap_set_device(ctx));
    VAP_TRY(check_shape(B, W, S));
    if (!h_waypoints || !c || !h_velocity) return vap_fail(VAP_ERR_INVALID, "null buffer");
    const size_t n_pts = (size_t)B * S, e = esz(dt);
    VAP_TRY(ctx->ensure(ctx->io[0], (size_t)B * W * 2 * e));
    void *houts[5] = {h_x, h_y, h_heading, h_curvature, h_velocity};
    void *douts[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 5; i++) {
        if (!houts[i]) continue;
        VAP_TRY(ctx->ensure(ctx->io[1 + i], n_pts * e));
        douts[i] = ctx->io[1 + i].ptr;
    }
    VAP_TRY(ctx->ensure(ctx->meta, (size_t)B * 4 * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->flags, (size_t)B * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(ctx->io[0].ptr, h_waypoints, (size_t)B * W * 2 * e, hipMemcpyHostToDevice, ctx->stream));
    VAP_TRY(vap_profile_batch(ctx, dt, B, W, S, dd, ctx->io[0].ptr, c, start_vel, end_vel, douts[0], douts[1],
                              douts[2], douts[3], douts[4], (double *)ctx->meta.ptr, (uint32_t *)ctx->flags.ptr));
    for (int i = 0; i < 5; i++)
        if (houts[i]) HIP_TRY(hipMemcpyAsync(houts[i], douts[i], n_pts * e, hipMemcpyDeviceToHost, ctx->stream));
    if (h_meta)
        HIP_TRY(hipMemcpyAsync(h_meta, ctx->meta.ptr, (size_t)B * 4 * sizeof(double), hipMemcpyDeviceToHost,
                               ctx->stream));
    if (h_flags)
        HIP_TRY(hipMemcpyAsync(h_flags, ctx->flags.ptr, (size_t)B * sizeof(uint32_t), hipMemcpyDeviceToHost,
                               ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VAP_OK;
}

int vap_eval_host(vap_ctx *ctx, int W, const double *h_segments, double param_last, int order, int n,
                  const double *h_t, double *h_out)
{
    VAP_TRY(vap_set_device(ctx));
    if (W < 2 || !h_segments) return vap_fail(VAP_ERR_UNFITTED, "Spline has not been fitted yet");  // QHS:222-223
    if (order < 0 || order > 2 || n < 0 || (n > 0 && (!h_t || !h_out))) return vap_fail(VAP_ERR_INVALID, "bad argument");
    if (n == 0) return VAP_OK;
    const size_t sb = (size_t)(W - 1) * 12 * sizeof(double);
    VAP_TRY(ctx->ensure(ctx->small_seg, sb));
    VAP_TRY(ctx->ensure(ctx->small_in, (size_t)n * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->small_out, (size_t)n * 2 * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(ctx->small_seg.ptr, h_segments, sb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->small_in.ptr, h_t, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(vap::launch_eval(ctx->stream, W, (const double *)ctx->small_seg.ptr, param_last, order, n,
                             (const double *)ctx->small_in.ptr, (double *)ctx->small_out.ptr));
    HIP_TRY(hipMemcpyAsync(h_out, ctx->small_out.ptr, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VAP_OK;
}

int vap_basis_host(vap_ctx *ctx, int order, int n, const double *h_t, double *h_out)
{
    VAP_TRY(vap_set_device(ctx));
    if (order < 0 || order > 3 || n < 0 || (n > 0 && (!h_t || !h_out))) return vap_fail(VAP_ERR_INVALID, "bad argument");
    if (n == 0) return VAP_OK;
    VAP_TRY(ctx->ensure(ctx->small_in, (size_t)n * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->small_out, (size_t)n * 6 * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(ctx->small_in.ptr, h_t, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(vap::launch_basis(ctx->stream, order, n, (const double *)ctx->small_in.ptr, (double *)ctx->small_out.ptr));
    HIP_TRY(hipMemcpyAsync(h_out, ctx->small_out.ptr, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VAP_OK;
}

int vap_lookup_host(vap_ctx *ctx, int W, const double *h_segments, double param_last, const double *h_lut,
                    int what, int n, const double *h_in, double *h_out)
{
    VAP_TRY(vap_set_device(ctx));
    if (W < 2 || !h_segments || !h_lut) return vap_fail(VAP_ERR_UNFITTED, "No splines have been initialized");
    if (what < 0 || what > 2 || n < 0 || (n > 0 && (!h_in || !h_out))) return vap_fail(VAP_ERR_INVALID, "bad argument");
    if (n == 0) return VAP_OK;
    const size_t sb = (size_t)(W - 1) * 12 * sizeof(double);
    VAP_TRY(ctx->ensure(ctx->small_seg, sb));
    VAP_TRY(ctx->ensure(ctx->small_lut, VAP_LUT_SAMPLES * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->small_in, (size_t)n * sizeof(double)));
    VAP_TRY(ctx->ensure(ctx->small_out, (size_t)n * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(ctx->small_seg.ptr, h_segments, sb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->small_lut.ptr, h_lut, VAP_LUT_SAMPLES * sizeof(double), hipMemcpyHostToDevice,
                           ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->small_in.ptr, h_in, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(vap::launch_lookup(ctx->stream, W, (const double *)ctx->small_seg.ptr, param_last,
                               (const double *)ctx->small_lut.ptr, what, n, (const double *)ctx->small_in.ptr,
                               (double *)ctx->small_out.ptr));
    HIP_TRY(hipMemcpyAsync(h_out, ctx->small_out.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VAP_OK;
}

}  // extern "C"
