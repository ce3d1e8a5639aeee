/* vap.h — C-ABI of libvap.so: MI355X-native batched trajectory generator.
 *
 * Drop-in boundary for the quintic-Hermite spline + 2-D motion-profile hot path of
 * RohitMovva/VexAutonomousPlanner.  The reference has no FFI of its own (it is pure Python); each
 * entry point below names the reference function(s) it replaces (file:line under the reference's
 * src/), and INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 *   QHS = splines/quintic_hermite_spline.py      SM = splines/spline_manager.py
 *   MPG = motion_profiling_v2/motion_profile_generator.py
 *
 * Conventions
 *   - plain C types only; every buffer is caller-owned; the library keeps no pointer after a call.
 *   - "d_" pointers are DEVICE (HBM) pointers valid on the context's device; "h_" pointers are host.
 *   - all entry points return 0 (VAP_OK) or a negative vap_status; none of them throws.
 *   - work is enqueued on the context's HIP stream (vap_ctx_set_stream); "_host" variants and
 *     vap_ctx_synchronize block, the device-pointer variants do not.
 *   - units: feet, seconds, radians.
 *
 * Precision (`vap_dtype`):
 *   VAP_F32  fp32 inputs/outputs.  Parameter/index arithmetic, derivative evaluation, curvature, heading
 *            differences and the velocity recurrence are carried in fp64 on the device (the reference's
 *            table/step-lookup quantisation, its finite-difference angular-acceleration term and the
 *            error amplification of its recurrence in tight curves are not reproducible to 1e-5
 *            otherwise); positions, headings and all stores are fp32.
 *            VAP_OPT_F32_RECURRENCE selects an all-fp32 recurrence instead.
 *   VAP_F64  fp64 inputs/outputs, all arithmetic fp64.  Not every operation is the reference's correctly rounded one:
 *            reciprocals, 1/sqrt and the final square root of a velocity come from the hardware estimates refined by
 *            Newton steps (within an ulp), the recurrence runs in its collapsed four-instruction form,
 *            and atan2 / pow are the device library's, not NumPy's.  Measured against the real reference: velocities
 *            <= 4.1e-11 on the curated fixtures, <= 4e-8 on 60 000 random shapes and robots, and 1.9e-6 on the worst case
 *            found (fixture big_w2048_p2: 2048 waypoints, 4e5 samples) — where the statement-by-statement sweep
 *            (VAP_VELOCITY_SEQ_LITERAL) gives the same 1.9e-6 and the fp64 CPU restatement itself is 7.9e-8 from the
 *            reference: the reference's own recurrence amplifies last-bit differences of its inputs by up to 1e9
 *            there.  The bound that holds for both dtypes on every path tried is north_star's 1e-5.
 */
#ifndef VAP_H
#define VAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAP_VERSION 100

typedef enum {
    VAP_OK = 0,
    VAP_ERR_INVALID = -1,     /* bad argument (NULL, W < 2, S < 2, ...): reference returns False */
    VAP_ERR_NO_DEVICE = -2,   /* no usable HIP device: the product never falls back to the CPU */
    VAP_ERR_HIP = -3,         /* a HIP runtime call failed; vap_last_error() has the text */
    VAP_ERR_UNFITTED = -4,    /* evaluator called before fit/build (reference: ValueError) */
    VAP_ERR_CAPACITY = -5,    /* output capacity S too small for the requested grid */
    VAP_ERR_UNSUPPORTED = -6
} vap_status;

typedef enum { VAP_F32 = 0, VAP_F64 = 1 } vap_dtype;

/* MPG:14-21 Constraints dataclass, same field order. */
typedef struct {
    double max_vel, max_acc, max_dec, friction_coef, max_jerk, track_width;
} vap_constraints;

/* Per-path flag bits written to flags[b]. */
#define VAP_FLAG_DEGENERATE 1u /* zero-length segment or non-finite value met during fit/LUT */
#define VAP_FLAG_TRUNCATED 2u  /* grid needed more than S samples; the first S were produced */
#define VAP_FLAG_NOCONVERGE 4u /* velocity relaxation hit its round limit (never expected) */

/* Timing slots of vap_last_timing (milliseconds, HIP events on the context's stream). */
enum { VAP_T_FIT = 0, VAP_T_LUT = 1, VAP_T_SAMPLE = 2, VAP_T_VELOCITY = 3, VAP_T_TOTAL = 4,
       VAP_T_COUNT = 8 };

#define VAP_LUT_SAMPLES 1000      /* SM:427 min_samples */
#define VAP_SAMPLES_PER_NODE 1000 /* SM:477 samples_per_node */

typedef struct vap_ctx vap_ctx;

/* ---- context ------------------------------------------------------------------------------- */
int vap_version(void);
const char *vap_status_string(int status);
/* Thread-local text of the last failure in this thread ("" if none). */
const char *vap_last_error(void);
/* Number of HIP devices visible (0 when there is none; never initialises a context). */
int vap_device_count(void);
/* One context = one device + one stream + its scratch arena.  Not thread-safe; one per thread. */
int vap_ctx_create(int device, vap_ctx **out);
int vap_ctx_destroy(vap_ctx *ctx);
/* Run subsequent work on `hip_stream` (a hipStream_t, e.g. torch's current stream; NULL is HIP's
 * default stream).  VAP_STREAM_OWN selects the context's own non-blocking stream again (the initial
 * state). */
#define VAP_STREAM_OWN ((void *)(intptr_t)-1)
int vap_ctx_set_stream(vap_ctx *ctx, void *hip_stream);
int vap_ctx_synchronize(vap_ctx *ctx);
/* Tuning / test knobs.  VAP_OPT_VELOCITY_KERNEL selects the K5 implementation: AUTO (default) picks
 * the register-resident relaxation kernel when the row fits and the sequential sweep otherwise;
 * SEQ_LITERAL is the statement-by-statement form of MPG:188-311, SEQ_FAST the same sweep with the
 * collapsed limits (bit-identical to RELAX).  RELAX_BLOCK is the workgroup-per-path kernel RELAX uses;
 * RELAX_WAVE (fp32) walks each path with one wave in stream-ordered windows — exact as well, kept for
 * experiments (slower on MI355X for the sizes measured).  LANES (fp64 recurrence) is "a wavefront of paths": a
 * lane walks a path, 16-64 paths per workgroup, coefficients streamed through LDS by producer waves — every
 * sample evaluated once per direction, bit-identical to SEQ_FAST, any row length; AUTO picks it for batches of
 * 2048 paths and more.  LANES_16 / _32 / _64 force its group size (tests).  Rows too long for the register-resident
 * kernel (config 2) are cut into super-chunks whose interface states are handed on by look-back inside one launch per
 * direction; RELAX_ROUNDS forces the earlier form of that kernel (one launch per super-round, convergence checked
 * on the host) — the same rows bit for bit (tests). */
enum { VAP_OPT_VELOCITY_KERNEL = 0, VAP_OPT_F32_RECURRENCE = 1, /* 2: retired (sampling inside the velocity kernel, rounds 3-4) */
       VAP_OPT_TIME_DOMAIN_RESIDUAL = 3, VAP_OPT_TIME_KERNEL = 4 };
enum { VAP_VELOCITY_AUTO = 0, VAP_VELOCITY_SEQ_LITERAL = 1, VAP_VELOCITY_SEQ_FAST = 2, VAP_VELOCITY_RELAX = 3,
       VAP_VELOCITY_RELAX_BLOCK = 4 /* workgroup per path */, VAP_VELOCITY_RELAX_WAVE = 5 /* wave per path, fp32 */,
       VAP_VELOCITY_LANES = 6 /* lane per path, fp64 recurrence */, VAP_VELOCITY_LANES_16 = 7, VAP_VELOCITY_LANES_32 = 8,
       VAP_VELOCITY_LANES_64 = 9, VAP_VELOCITY_RELAX_ROUNDS = 10 /* long rows: host-checked super-rounds */ };
/* VAP_OPT_F32_RECURRENCE: arithmetic of the forward/backward velocity recurrence in VAP_F32 calls.
 *   VAP_RECURRENCE_F64 (default): the sampling kernel keeps fp64 curvature / heading-difference rows in
 *     context scratch and the recurrence runs in fp64 on them; inputs and every output row stay fp32.  The
 *     reference's recurrence amplifies a rounding error by track_width*curvature/2 per step in curves tighter
 *     than 2/track_width, so only this mode holds 1e-5 against the reference on every path.
 *   VAP_RECURRENCE_F32: rows and recurrence in fp32 — faster, and within 1e-5 on ~98.6 % of config-3-shaped
 *     paths (worst sample 7e-5). */
enum { VAP_RECURRENCE_F64 = 0, VAP_RECURRENCE_F32 = 1 };
/* VAP_OPT_TIME_DOMAIN_RESIDUAL (1 = on, the default; 0 = off): VAP_F32 calls with the fp64 recurrence also leave, in
 * context scratch, what each stored fp32 velocity lost of the fp64 value (an fp32 residual row, 4 B per sample-point of
 * extra writes).  A following vap_time_profile / vap_time_profile_routes that is handed that velocity row integrates
 * row + residual (MPG:566-584) — the caller's row as it is at that moment plus a term below its own rounding — which
 * is what keeps fp32 time-domain rows within 1e-5 of the reference.  Callers that never go to the time domain (pure
 * distance-domain batches, e.g. candidate ranking) switch it off and save the traffic. */
/* VAP_OPT_TIME_KERNEL: the kinematic recurrence of vap_time_profile[_routes] (MPG:566-584).  LANE walks a path with one
 * lane; QUAD with four (one grid index, one velocity sample and one interpolation per lane instead of two, four and two,
 * the 64-byte row stored as four 16-byte pieces) — the same rows bit for bit, a shorter step.  AUTO (default) takes
 * QUAD while four lanes per path still leave at most one wavefront per SIMD (B <= 16384) and LANE above that.
 * FUSED: QUAD's recurrence and the geometry of the rows behind it in one workgroup of 16 plain paths (the geometry runs in
 * the shadow of the recurrence); AUTO takes it while that is at most one workgroup per CU (B <= 16 x CUs), and batches of
 * routes (vap_time_profile_routes) never do. */
enum { VAP_TIME_KERNEL_AUTO = 0, VAP_TIME_KERNEL_LANE = 1, VAP_TIME_KERNEL_QUAD = 2, VAP_TIME_KERNEL_FUSED = 3 };
int vap_ctx_set_option(vap_ctx *ctx, int option, int value);
/* Enable/disable per-stage hipEvent timing (replaces the reference's time.time() log lines,
 * SM:587-594, MPG:398-411).  Off by default. */
int vap_ctx_set_timing(vap_ctx *ctx, int enabled);
int vap_last_timing(vap_ctx *ctx, float ms[VAP_T_COUNT]);

/* ---- staged device API (plain-node paths: one spline of W control points per path) ----------
 * Buffers, for a batch of B paths with W waypoints (G = W-1 segments) and sample capacity S:
 *   waypoints  [B][W][2]                 dtype
 *   segments   [B][G][6][2]   fp64       rows p0,p1,d0*L,d1*L,dd0*L^2,dd1*L^2 (QHS:92-122)
 *   lut        [B][1000]      fp64       lookup_table.distances (SM:448-454); parameters are
 *                                        j * param_last/999 (np.linspace) and are not stored
 *   meta       [B][4]         fp64       {parameters[-1] (QHS:736), total_length, dd, n_samples}
 *   x,y,heading,curvature,velocity [B][S] dtype
 */

/* QHS:30-138 fit + QHS:149-219 _compute_derivatives + QHS:719-736 _compute_parameters, batched;
 * also SM:42-172 build_path for plain nodes.  d_tangent_in/out: optional [B][W][2] fp64 per-node
 * tangent overrides (NaN = None; SM:65-77, QHS:102-115), or NULL. */
int vap_fit(vap_ctx *ctx, vap_dtype dt, int B, int W, const void *d_waypoints,
            const double *d_tangent_in, const double *d_tangent_out, double *d_segments,
            double *d_segment_lengths /* [B][G] fp64, QHS:84-85, may be NULL */, double *d_meta,
            uint32_t *d_flags);

/* vap_fit with the rest of QuinticHermiteSpline.fit's own inputs (QHS:30-138), for callers that use the spline
 * class directly as SM:57-168 does for the splines of a split route:
 *   d_first_derivatives / d_second_derivatives  [B][W][2] fp64: used only when BOTH are given (QHS:52-68: with
 *       one missing, _compute_derivatives overwrites both)
 *   d_starting_tangent / d_ending_tangent       [B][2] fp64, NaN row = not set: QHS:129-132 -> set_starting_tangent /
 *       set_ending_tangent (QHS:543-590) — both write into the LAST segment (quirk Q3) — and the 2-point special
 *       case of _compute_derivatives (QHS:170-172, 181-182: the chord stays un-normalised).
 *   d_first_out / d_second_out                  [B][W][2] fp64, optional: the first_derivatives / second_derivatives
 *       attributes the reference leaves behind (estimates or the caller's arrays, with the setters' writes to
 *       first_derivatives[0] / [-1], QHS:557, 582).
 * Any of the six may be NULL. */
int vap_fit_ex(vap_ctx *ctx, vap_dtype dt, int B, int W, const void *d_waypoints,
               const double *d_tangent_in, const double *d_tangent_out, const double *d_first_derivatives,
               const double *d_second_derivatives, const double *d_starting_tangent,
               const double *d_ending_tangent, double *d_segments, double *d_segment_lengths,
               double *d_first_out, double *d_second_out, double *d_meta, uint32_t *d_flags);

/* SM:426-475 build_lookup_table.  Fills lut and meta[1] (= get_total_arc_length, SM:320-330). */
int vap_build_lut(vap_ctx *ctx, int B, int W, const double *d_segments, double *d_lut,
                  double *d_meta, uint32_t *d_flags);

/* Distance grid + per-sample properties: MPG:112-176 (the sampling loop of forward_backward_pass)
 * with SM:291-318 distance_to_time, SM:477-580 (the curvature/heading table entry the reference's
 * step lookup selects, evaluated on demand — the 1000*W table is never materialised) and
 * SM:204-215 get_point_at_parameter.
 *   dd > 0 : reference grid: s_0 = 0, s_k = fl(s_(k-1) + dd) while s_k < L (the reference's
 *            accumulated current_dist += dd, rounding included), plus the end sample; n_samples varies
 *   dd <= 0: fixed grid of exactly S samples, the same accumulation with dd_b = L_b / (S - 1.5)
 * The accumulated sum is reproduced in closed form (per binade the rounded increment is a constant
 * number of ulps), so sample k needs no scan over its predecessors: vap_grid_distances shows it.
 * Writes meta[2], meta[3]; d_dtheta [B][S] (dtype) receives |heading[k+1]-heading[k]| for the
 * velocity pass (scratch; may be NULL only if the velocity pass is not wanted). */
int vap_sample(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, double dd, const double *d_segments,
               const double *d_lut, double *d_meta, void *d_x, void *d_y, void *d_heading,
               void *d_curvature, void *d_dtheta, uint32_t *d_flags);

/* MPG:188-316 forward + backward pass.  d_velocity receives the final velocities (MPG:316).
 * As in the reference, max_dec does not take part: boundary_map always contains sample 0 (MPG:110), so
 * forward_backward_pass replaces it with max_acc before its first step (MPG:194-196) and decelerates
 * with max_acc; max_dec is used by the time loop only (vap_time_profile, vap_route_motion_profile).
 * d_vcap: optional [B][S] (vap_limit_rows_dtype) per-sample initial velocities — the `velocities` list the reference
 * starts from (MPG:121,127,153,172: node / action-point max_velocity, 0.01 at stops); NULL = the
 * plain-node default max_vel with start/end velocities at the ends.  Entry 0 and the end sample are
 * taken from start_vel / end_vel.  Rows that fit the register-resident relaxation kernel (20 480
 * samples fp32, 10 240 fp64) run there, longer ones in the one-lane sequential sweep.
 * d_dtheta NULL = the rows the last sampling call (vap_sample / vap_profile_batch) of this shape and dtype
 * left on the context; for VAP_F32 with VAP_RECURRENCE_F64 these are fp64 curvature AND |dtheta| rows
 * (d_curvature is then not read and may be NULL) and the recurrence runs in fp64.  With an explicit d_dtheta
 * the recurrence runs in `dt` on the caller's rows. */
int vap_velocity_pass(vap_ctx *ctx, vap_dtype dt, int B, int S, const vap_constraints *c,
                      double start_vel, double end_vel, const double *d_meta,
                      const void *d_curvature, const void *d_dtheta, const void *d_vcap,
                      void *d_velocity, uint32_t *d_flags);

/* Per-sample limits of forward_backward_pass (MPG:100-176, 194-196, 256-257) for B routes whose nodes and
 * action points carry max_velocity, max_acceleration and stop, on the distance grid of the last
 * vap_sample / vap_profile_batch call on this context (same B, W, S):
 *   d_vcap          the `velocities` list the pass starts from (running max_velocity, 0.01 at stops,
 *                   end_vel at the end sample)
 *   d_acc_forward   max_acc (= max_dec) in force for the forward step from each sample  (boundary_map /
 *   d_acc_backward  max_acc the backward sweep has in force for its step from each sample  max_accels,
 *   d_dec_backward  [B] max_dec of the backward sweep: what the forward sweep left behind   incl. their quirks)
 * A node (parameter = its index) or action point (parameter t) takes effect at the first loop sample whose
 * parameter has reached it (MPG:125, 141-145) — a node before an action point on the same sample, and an
 * action point that would fall on its predecessor's sample never does, nor do those after it (the
 * reference looks at one pending action point per sample).
 *   d_node_max_velocity / d_node_max_acceleration [B][W]  (<= 0: none; NULL array: none)  MPG:100-107, 129-137
 *   d_node_stop                                   [B][W]  int32 (NULL: none)              MPG:126-127
 *   d_action_t                                    [B][M]  in route order, > 0; pad with +inf
 *   d_action_max_velocity / _max_acceleration / _stop [B][M]  (NULL arrays: none)         MPG:146-160
 *   d_vcap, d_acc_forward, d_acc_backward [B][S], d_dec_backward [B] out, of type vap_limit_rows_dtype(ctx, dt):
 *                                    the three acceleration outputs are optional as a set (routes that do not
 *                                    change max_acceleration need only d_vcap)
 *   d_node_sample [B][W], d_action_sample [B][M]  int32 out, optional: the sample at which each takes
 *                                    effect (node 0: 0; INT_MAX: never)
 * d_lut NULL = the table of the last vap_profile_batch.  Reverse / turn nodes are not covered here
 * (vap_route_* is the general single-route path); waits act in the time domain (vap_time_insert_waits). */
/* Type of the limit rows (d_vcap, d_acc_forward, d_acc_backward, d_dec_backward) for rows of type dt in this context:
 * the type of the recurrence they enter — VAP_F64 for fp64 rows and for fp32 rows in the default mode
 * (VAP_RECURRENCE_F64: the fp64 recurrence amplifies an fp32-rounded limit such as 13.9 ft/s^2 past 1e-5, MPG:194-196,
 * 256-257 / DESIGN.md section 3), VAP_F32 for fp32 rows with VAP_RECURRENCE_F32. */
int vap_limit_rows_dtype(vap_ctx *ctx, vap_dtype dt);

int vap_route_limits(vap_ctx *ctx, vap_dtype dt, int B, int W, int M, int S, const double *d_lut,
                     const double *d_meta, const double *d_node_max_velocity,
                     const double *d_node_max_acceleration, const int *d_node_stop, const double *d_action_t,
                     const double *d_action_max_velocity, const double *d_action_max_acceleration,
                     const int *d_action_stop, const vap_constraints *c, double end_vel, void *d_vcap,
                     void *d_acc_forward, void *d_acc_backward, void *d_dec_backward, int *d_node_sample,
                     int *d_action_sample);

/* vap_velocity_pass with the limit rows of vap_route_limits (the three acceleration arguments NULL, or
 * all set together with d_vcap).  With acceleration rows the register-resident kernel covers rows up to
 * 10 240 samples (fp32) / 4096 (fp64); longer rows take the sequential sweep. */
int vap_velocity_pass_limits(vap_ctx *ctx, vap_dtype dt, int B, int S, const vap_constraints *c,
                             double start_vel, double end_vel, const double *d_meta,
                             const void *d_curvature, const void *d_dtheta, const void *d_vcap,
                             const void *d_acc_forward, const void *d_acc_backward,
                             const void *d_dec_backward, void *d_velocity, uint32_t *d_flags);

/* MPG:413-628, the time-domain resample that generate_motion_profile runs after
 * forward_backward_pass, for B plain-node paths (no turn / wait / reverse nodes and no action points:
 * those insert rows — use vap_route_motion_profile).  One row per time step of `time_step` seconds
 * (0.01 in the reference, MPG:389):
 *   rows      [B][capacity_rows][8] fp64  {time, position, linear velocity, acceleration, heading,
 *                                          angular velocity, x, y}   (MPG:558-592)
 *   counts    [B][2] int                  {rows written, entries of nodes_map}
 *   nodes_map [B][W] int                  row index at which each node is passed (MPG:420, 527-529;
 *                                          quirk Q5: the last node is never recorded)
 * d_velocity is the [B][S] result of vap_velocity_pass / vap_profile_batch in `dt`; d_meta as above.
 * d_segments / d_lut may both be NULL: the tables this context built in its last vap_profile_batch
 * call (same B and W) are used.  A path needing more than capacity_rows rows is cut there and flagged
 * VAP_FLAG_TRUNCATED. */
int vap_time_profile(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, const double *d_segments,
                     const double *d_lut, const double *d_meta, const void *d_velocity,
                     const vap_constraints *c, double time_step, int capacity_rows, double *d_rows,
                     int *d_counts, int *d_nodes_map, uint32_t *d_flags);

/* Waits and action points in the time domain (MPG:457-476, 509-518, 543-553) on top of the rows of
 * vap_time_profile: a node or action point with wait_time inserts int(wait_time/time_step) rows (zero
 * position / velocity / acceleration / angular velocity, the last heading and point) where it is passed,
 * and every later row moves by as many rows and time steps; actions_map records the row count at which
 * each action point fires (the reference's own test, MPG:546-553: one that lands exactly on a row's
 * parameter, or shares a row interval with its predecessor, never fires and blocks the ones after it).
 *   d_rows_in [B][capacity_in][8], d_counts_in [B][2], d_nodes_map_in [B][W]   from vap_time_profile
 *   d_node_wait   [B][W]  seconds (NULL: none)       d_action_t / d_action_wait [B][M] (pad t with +inf)
 *   d_rows_out    [B][capacity_out][8]  (must not alias d_rows_in)
 *   d_counts_out  [B][3] int32: rows, nodes_map entries, actions_map entries
 *   d_nodes_map_out [B][W], d_actions_map_out [B][M] int32
 * Segments / table NULL = those of the last vap_profile_batch.  Turn and reverse nodes are not covered
 * (vap_route_motion_profile). */
int vap_time_insert_waits(vap_ctx *ctx, int B, int W, int M, int capacity_in, int capacity_out, double time_step,
                          const double *d_segments, const double *d_lut, const double *d_meta,
                          const double *d_rows_in, const int *d_counts_in, const int *d_nodes_map_in,
                          const double *d_node_wait, const double *d_action_t, const double *d_action_wait,
                          double *d_rows_out, int *d_counts_out, int *d_nodes_map_out, int *d_actions_map_out,
                          uint32_t *d_flags);

/* vap_time_profile on the tables this context holds from its last vap_profile_batch / vap_profile_routes call, with
 * the reversed state of routes (MPG:431-433, 540-541): d_node_reverse [B][W] int32 (NULL: none); rows made while an
 * odd number of reverse nodes (node 0's flag included) has been passed carry heading - pi (before the wrap and the
 * sign, MPG:555-563) and negated velocity and acceleration (MPG:587-589).  Rows as vap_time_profile. */
int vap_time_profile_routes(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, const double *d_meta,
                            const void *d_velocity, const vap_constraints *c, double time_step, int capacity_rows,
                            const int *d_node_reverse, double *d_rows, int *d_counts, int *d_nodes_map,
                            uint32_t *d_flags);

/* vap_time_insert_waits plus in-place turns (MPG:487-507 handle_turn over MPG:319-346 motion_profile_angle and
 * one_dim_mp_generator.py:4-69): a node with turn != 0 inserts, where it is passed and before its wait, the rows of a
 * trapezoidal heading profile of max_vel / max_acc on an arc of |turn| * track_width / 2 (zero velocity, the last
 * position and point, headings continuing from the last row, wrapped).  On the context's own tables (plain batch or
 * routes).  d_node_turn [B][W] degrees, d_node_reverse [B][W] (only node 0's wait heading reads it, MPG:463-464);
 * either may be NULL.  A turn at node 0 raises in the reference (quirk Q4): VAP_FLAG_BAD_ROUTE. */
int vap_time_insert_events(vap_ctx *ctx, int B, int W, int M, int capacity_in, int capacity_out, double time_step,
                           const vap_constraints *c, const double *d_meta, const double *d_rows_in,
                           const int *d_counts_in, const int *d_nodes_map_in, const double *d_node_wait,
                           const double *d_node_turn, const int *d_node_reverse, const double *d_action_t,
                           const double *d_action_wait, double *d_rows_out, int *d_counts_out, int *d_nodes_map_out,
                           int *d_actions_map_out, uint32_t *d_flags);

/* ---- fused hot path ------------------------------------------------------------------------ */

/* rebuild_tables (SM:582-594) + forward_backward_pass (MPG:70-316) for B plain-node paths, inputs
 * and outputs resident in HBM.  Any output pointer may be NULL except d_velocity.
 * d_meta: optional [B][4] fp64 (see above); d_flags: optional [B]. */
int vap_profile_batch(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, double dd,
                      const void *d_waypoints, const vap_constraints *c, double start_vel,
                      double end_vel, void *d_x, void *d_y, void *d_heading, void *d_curvature,
                      void *d_velocity, double *d_meta, uint32_t *d_flags);

/* The same for B routes whose reverse / turn nodes cut them into several splines (SM:42-172 build_path for a whole
 * batch): fit with split tangents (SM:84-158) and the tangent setters' quirk (QHS:543-590), one 1000-entry table per
 * spline concatenated with running offsets (SM:436-464), distance -> parameter over that table and parameter ->
 * (spline, local parameter) with a split node belonging to the earlier spline (SM:243-275), then the plain velocity
 * pass — forward_backward_pass treats reverse / turn nodes like any other node (MPG:112-176).
 *   max_splines            upper bound of the splines of any route of the batch (1 + its reverse / turn nodes among
 *                          nodes 1..W-2); table scratch is sized by it
 *   d_node_reverse [B][W]  int32 is_reverse_node, d_node_turn [B][W] degrees (NULL arrays: none)
 *   d_node_tangent [B][W][2] fp64 (NaN row = None) with d_node_magnitudes [B][W][2] {incoming, outgoing} (SM:65-77)
 *   d_spline_counts [B]    int32 out, optional: splines per route
 * A route with a reverse / turn attribute on its LAST node (IndexError in the reference, SM:97) or with more
 * splines than max_splines is flagged VAP_FLAG_BAD_ROUTE; its rows are undefined.  Afterwards the context holds the
 * batch's tables: vap_route_limits + vap_velocity_pass_limits apply node / action-point limits as for plain paths.
 * The time domain of such a batch: vap_time_profile_routes, vap_time_insert_events. */
#define VAP_FLAG_BAD_ROUTE 8u
int vap_profile_routes(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, double dd, int max_splines,
                       const void *d_waypoints, const int *d_node_reverse, const double *d_node_turn,
                       const double *d_node_tangent, const double *d_node_magnitudes, const vap_constraints *c,
                       double start_vel, double end_vel, void *d_x, void *d_y, void *d_heading, void *d_curvature,
                       void *d_velocity, double *d_meta, uint32_t *d_flags, int *d_spline_counts);

/* Same with host buffers (allocates device scratch in the context arena, copies in and out,
 * synchronises).  This is what a single-path GUI call uses. */
int vap_profile_batch_host(vap_ctx *ctx, vap_dtype dt, int B, int W, int S, double dd,
                           const void *h_waypoints, const vap_constraints *c, double start_vel,
                           double end_vel, void *h_x, void *h_y, void *h_heading,
                           void *h_curvature, void *h_velocity, double *h_meta,
                           uint32_t *h_flags);

/* ---- scalar / vector evaluators on a fitted path (host buffers) -----------------------------
 * SM:204-241 get_point / get_derivative / get_second_derivative _at_parameter for n parameters of
 * path 0 of a (1,G,6,2) segment block held on the host.  order = 0,1,2.  out [n][2] fp64. */
int vap_eval_host(vap_ctx *ctx, int W, const double *h_segments, double param_last, int order,
                  int n, const double *h_t, double *h_out);

/* QHS:288-322 _get_basis_functions (order 0), QHS:324-363 _get_basis_derivatives (1), QHS:365-416
 * _get_basis_second_derivatives (2), QHS:418-469 _get_basis_third_derivatives (3) at n local parameters
 * (0..1 inside a segment): out [n][6] fp64 = [H0..H5] of that order, the reference's association order. */
int vap_basis_host(vap_ctx *ctx, int order, int n, const double *h_t, double *h_out);

/* SM:291-318 distance_to_time for n distances; SM:332-346 get_heading / get_curvature (table step
 * lookup) for n parameters.  what: 0 = distance_to_time, 1 = curvature, 2 = heading. */
int vap_lookup_host(vap_ctx *ctx, int W, const double *h_segments, double param_last,
                    const double *h_lut, int what, int n, const double *h_in, double *h_out);

/* ---- one general route (reverse / turn nodes, per-node limits, action points), fp64 -------------
 * The completeness path behind the drop-in classes: everything generate_motion_profile needs for a
 * GUI route, on the device.  Host buffers in, host buffers out; sizes are GUI-sized. */

/* Node / action-point attributes the path code reads (gui/node.py:17-51, gui/action_point.py:16-41).
 * Any attribute array may be NULL (= the GUI defaults: False / 0 / None). */
typedef struct {
    int n_nodes;                    /* W */
    const double *waypoints;        /* [W][2] feet */
    const int *is_reverse;          /* [W] */
    const double *turn;             /* [W] degrees */
    const int *stop;                /* [W] */
    const double *wait_time;        /* [W] seconds */
    const double *max_velocity;     /* [W] 0 = unset */
    const double *max_acceleration; /* [W] 0 = unset */
    const double *tangent;          /* [W][2], NaN row = None */
    const double *magnitudes;       /* [W][2] incoming, outgoing */
    int n_actions;                  /* M */
    const double *ap_t;             /* [M] path parameter */
    const int *ap_stop;
    const double *ap_wait_time, *ap_max_velocity, *ap_max_acceleration;
} vap_route_desc;

typedef struct vap_route vap_route;

/* SM:42-172 build_path (splits at reverse / turn nodes, SM:84-158 split tangents), QHS:30-219 fit per
 * spline (quirk Q3 kept), SM:426-475 lookup table.  VAP_ERR_INVALID where the reference returns
 * False or raises (fewer than 2 nodes; reverse/turn attribute on the last node). */
int vap_route_create(vap_ctx *ctx, const vap_route_desc *desc, vap_route **out);
int vap_route_destroy(vap_route *route);
/* number of splines and lookup_table.total_length (SM:320-330) */
int vap_route_info(vap_route *route, int *n_splines, double *total_length);
/* build_lookup_table(min_samples = lut_samples) and precompute_path_properties(samples_per_node) with sizes other than
 * the defaults every caller in the reference uses (1000 / 1000; SM:426-427, 477): rebuilds the route's arc-length
 * table with lut_samples entries per spline (np.linspace, trapezoid increments, np.cumsum — same operations, same
 * order) and makes the step lookup of get_heading / get_curvature (SM:550-580) read a table of
 * samples_per_node * len(nodes) entries.  Every later call on the route (lookups, forward_backward, motion_profile)
 * uses the new tables.  VAP_ERR_INVALID for lut_samples < 2 (the reference indexes local_params[1], SM:444). */
int vap_route_set_table_sizes(vap_route *route, int lut_samples, int samples_per_node);
int vap_route_table_sizes(vap_route *route, int *lut_samples, int *samples_per_node);
/* Per-spline results for the host mirrors of the drop-in classes; any pointer may be NULL.
 * start/npts/param_last [n_splines]; segments [(W-1)][6][2]; segment_lengths [W-1];
 * lut_distances / lut_parameters [n_splines*1000] = PathLookupTable (SM:466-475). */
int vap_route_get_splines(vap_route *route, int *h_start, int *h_npts, double *h_param_last, double *h_segments,
                          double *h_segment_lengths, double *h_lut_distances, double *h_lut_parameters);
/* SM:204-241 at n global parameters; order 0/1/2; out [n][2]. */
int vap_route_eval(vap_route *route, int order, int n, const double *h_t, double *h_out);
/* what: 0 = SM:291-318 distance_to_time, 1 = SM:340-346 get_curvature, 2 = SM:332-338 get_heading. */
int vap_route_lookup(vap_route *route, int what, int n, const double *h_in, double *h_out);
/* Samples forward_backward_pass produces for spacing dd (MPG:112-122, 172-175). */
int vap_route_sample_count(vap_route *route, double dd, int *n_out);
/* Host only, no device needed: the distance grid of MPG:112-122 for a path of length total_length
 * from the closed form the kernels use (csrc/vap_device.h build_grid_runs / grid_s).  Writes
 * s_k for k < min(*n_out, capacity) into h_s (may be NULL) and the loop count — the number of k with
 * s_k < total_length, i.e. n_samples - 1 — into *n_out.  Bit-identical to the reference's running sum. */
int vap_grid_distances(double dd, double total_length, long capacity, double *h_s, long *n_out);
/* MPG:70-316 with node / action-point limits (MPG:100-163) and boundary_map (MPG:194-196, 256-257).
 * Outputs (capacity each, any may be NULL): parameter t, x, y, heading, curvature, velocity. */
int vap_route_forward_backward(vap_route *route, const vap_constraints *c, double dd, double start_vel,
                               double end_vel, int capacity, int *n_out, double *h_t, double *h_x, double *h_y,
                               double *h_heading, double *h_curvature, double *h_velocity);
/* MPG:389-628 incl. in-place turns (MPG:319-346, one_dim_mp_generator.py:4-69) and waits.
 * rows [capacity_rows][8] = {time, position, linear_vel, acceleration, heading, angular_vel, x, y};
 * nodes_map (capacity W+1) / actions_map (capacity M+1) receive row indices (MPG:420, 528, 550).
 * VAP_ERR_CAPACITY if more rows are needed; VAP_ERR_INVALID where the reference raises (turn at node 0). */
int vap_route_motion_profile(vap_route *route, const vap_constraints *c, double dt, double dd, long capacity_rows,
                             double *h_rows, long *n_rows, long *h_nodes_map, int *n_nodes_map,
                             long *h_actions_map, int *n_actions_map);

#ifdef __cplusplus
}
#endif
#endif /* VAP_H */
