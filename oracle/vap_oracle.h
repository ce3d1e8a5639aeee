/* TEST INFRASTRUCTURE ONLY — not part of the product, never linked into libvap.so.
 *
 * CPU (fp64, scalar, same operation order) restatement of the hot path of
 * RohitMovva/VexAutonomousPlanner: src/splines/quintic_hermite_spline.py,
 * src/splines/spline_manager.py and src/motion_profiling_v2/motion_profile_generator.py.
 * Pinned against golden vectors produced by running the real reference
 * (oracle/gen_golden.py -> the .npz fixtures in tests/golden/; tests/test_oracle_golden.py).
 *
 * Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 */
#ifndef VAP_ORACLE_H
#define VAP_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vapo_path vapo_path;

/* Per-node attributes (all arrays length W, any may be NULL = reference defaults).
 * tangent: W x 2, NaN rows = "tangent is None"; magnitudes: W x 2 = (incoming, outgoing). */
typedef struct {
    const int *is_reverse;
    const double *turn;       /* degrees */
    const int *stop;
    const double *wait_time;
    const double *max_velocity;
    const double *max_acceleration;
    const double *tangent;
    const double *magnitudes;
} vapo_nodes;

/* Action points (arrays length M). */
typedef struct {
    int M;
    const double *t;
    const int *stop;
    const double *wait_time;
    const double *max_velocity;
    const double *max_acceleration;
} vapo_actions;

/* build_path (spline_manager.py:42-172) + fit (quintic_hermite_spline.py:30-138).
 * Returns NULL when the reference would return False or raise. */
vapo_path *vapo_path_create(int W, const double *waypoints /* W x 2 */, const vapo_nodes *nodes,
                            const vapo_actions *actions);
void vapo_path_destroy(vapo_path *p);

int vapo_n_splines(const vapo_path *p);
int vapo_n_segments(const vapo_path *p); /* total over splines == W-1 */
/* copies (W-1) x 6 x 2 segment blocks, W-1 segment lengths, n_splines parameters[-1] values */
void vapo_get_segments(const vapo_path *p, double *seg, double *seglen, double *param_last);

/* scalar evaluators at a GLOBAL parameter (spline_manager.py:204-241) */
/* QHS:288-469: [H0..H5] of the position / first / second / third derivative basis (order 0..3) at local parameter t */
void vapo_basis(int order, double t, double H[6]);
void vapo_point(const vapo_path *p, double t, double out[2]);
void vapo_derivative(const vapo_path *p, double t, double out[2]);
void vapo_second_derivative(const vapo_path *p, double t, double out[2]);

/* rebuild_tables (spline_manager.py:582-594): build_lookup_table + precompute_path_properties */
void vapo_rebuild_tables(vapo_path *p);
/* the two builders with explicit sizes (spline_manager.py:426-427, 477); -1 for min_samples < 2 */
int vapo_build_tables_sized(vapo_path *p, int min_samples, int samples_per_node);
int vapo_lut_size(const vapo_path *p);
void vapo_get_lut(const vapo_path *p, double *dist, double *param, double *total);
int vapo_table_size(const vapo_path *p);
void vapo_get_table(const vapo_path *p, double *param, double *curv, double *head);

double vapo_total_arc_length(const vapo_path *p);             /* spline_manager.py:320-330 */
double vapo_distance_to_time(const vapo_path *p, double s);   /* spline_manager.py:291-318 */
double vapo_curvature(const vapo_path *p, double t);          /* spline_manager.py:340-346,550-580 */
double vapo_heading(const vapo_path *p, double t);            /* spline_manager.py:332-338,550-580 */

/* Number of distance samples forward_backward_pass produces for spacing dd
 * (motion_profile_generator.py:112-176): loop samples + the appended end sample. */
long vapo_count_samples(const vapo_path *p, double dd);

/* Spacing of this build's "fixed sample count" grid: dd = L / (S - 1.5). */
double vapo_dd_for_samples(const vapo_path *p, long S);

/* forward_backward_pass (motion_profile_generator.py:70-316) for constraints
 * c = {max_vel,max_acc,max_dec,friction_coef,max_jerk,track_width}.
 * Outputs (each capacity cap, any may be NULL): parameter t, x, y, heading, curvature, velocity.
 * Returns N (number of samples) or -1 if cap < N. */
long vapo_forward_backward(const vapo_path *p, const double c[6], double dd, double start_vel,
                           double end_vel, long cap, double *t, double *x, double *y,
                           double *heading, double *curvature, double *velocity);

/* Whole hot path for a batch of plain-node paths (the cpu_baseline of bench.py):
 * for each path: create, rebuild_tables, forward_backward with S fixed samples (dd = L/(S-1.5)).
 * Outputs are B x S fp64 (any may be NULL).  Returns 0 or the index+1 of the first failing path.
 * n_threads > 1 distributes paths over pthreads. */
int vapo_profile_batch(int B, int W, long S, const double *waypoints, const double c[6],
                       double start_vel, double end_vel, double *x, double *y, double *heading,
                       double *curvature, double *velocity, double *total_length, int n_threads);

/* generate_motion_profile time-domain resample (motion_profile_generator.py:389-628).
 * Row r of out (capacity cap rows x 8 doubles) = {time, position, linear_vel, accel, heading,
 * angular_vel, x, y}.  nodes_map / actions_map receive sample indices (capacities W and M).
 * Returns T (rows) or -1 on overflow, -2 for inputs on which the reference raises. */
long vapo_generate_motion_profile(vapo_path *p, const double c[6], double dt, double dd, long cap,
                                  double *out, long *nodes_map, int *n_nodes_map,
                                  long *actions_map, int *n_actions_map);

#ifdef __cplusplus
}
#endif
#endif
