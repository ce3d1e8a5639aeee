// vap_kernels.h — launcher declarations shared by vap_kernels.hip and vap_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vap {

constexpr uint32_t VAP_FLAG_DEGENERATE_BIT = 1u;
constexpr uint32_t VAP_FLAG_TRUNCATED_BIT = 2u;
constexpr uint32_t VAP_FLAG_NOCONVERGE_BIT = 4u;

constexpr int kSampleThreads = 256;
constexpr int kSPT = 4;                                   // consecutive samples per thread (one 16-byte store)
constexpr int kSampleChunk = kSampleThreads * kSPT;       // samples evaluated per workgroup
constexpr int kSampleTile = kSampleChunk - kSPT;          // samples written: the last thread only feeds its neighbour
constexpr int kCoefBlockDoubles = 36;                      // doubles per segment coefficient block (scratch sizing)
constexpr int kGridRunBlockDoubles = 3 * 100;            // distance-grid runs per path: 100 entries of {k0, s0, D} (vap_device.h)
constexpr int kLdsCoefSegments = 112;                     // segments whose coefficient blocks are staged in LDS
constexpr int kMaxWaypoints = 2048;              // k_fit LDS: 7*W doubles
constexpr int kMaxRouteWaypoints = 1500;         // k_fit_routes LDS: 13*W doubles + W ints (108 B per node of 160 KB)

// Routes cut into several splines by reverse / turn nodes (vap_routes_batch.hip): the per-route spline table
//   sptab [B][NS][4] = {parameters[-1], distance offset, parameter offset, first node}, nspl [B] = splines per route
struct RouteTables {
    const double *sptab = nullptr;
    const int *nspl = nullptr;
    int NS = 1;
};
struct RouteSplitInputs {
    const int *rev = nullptr;         // [B][W] is_reverse_node
    const double *turn = nullptr;     // [B][W] degrees
    const double *tangent = nullptr;  // [B][W][2], NaN row = None
    const double *mag = nullptr;      // [B][W][2] incoming, outgoing magnitude
};
hipError_t launch_fit(hipStream_t st, bool f64, int B, int W, const void *wp, const double *tin,
                      const double *tout, double *seg, double *pw, double *seglen, double *meta, uint32_t *flags,
                      const double *first = nullptr, const double *second = nullptr, const double *start_tan = nullptr,
                      const double *end_tan = nullptr, double *out_first = nullptr, double *out_second = nullptr);
// When aux is set, k_lut also defines each path's distance grid (what launch_grid does) — the fused call, where
// the spacing is known before the table exists.
struct GridArgs {
    int S = 0;
    double dd = 0.0;
    double *aux = nullptr;
    double *runs = nullptr;
};
hipError_t launch_lut(hipStream_t st, int B, int W, const double *seg, double *lut, double *slopes, double *meta,
                      uint32_t *flags, GridArgs grid = GridArgs(), RouteTables rt = RouteTables());
// K1 + K2 in one launch for the fused call on plain paths (same segments, same table)
bool fit_lut_fusable(int B, int W);
hipError_t launch_fit_lut(hipStream_t st, bool f64, int B, int W, const void *wp, double *seg, double *pw, double *lut,
                          double *meta, uint32_t *flags, GridArgs grid);
hipError_t launch_fit_routes(hipStream_t st, bool f64, int B, int W, int NS, const void *wp, const RouteSplitInputs &in, double *seg,
                             double *pw, double *seglen, double *sptab, int *nspl, double *meta, uint32_t *flags);
hipError_t launch_route_offsets(hipStream_t st, int B, int W, int NS, int S, double dd, const double *lut, double *sptab,
                                const int *nspl, double *meta, double *aux, double *runs, uint32_t *flags);
hipError_t launch_sample_routes(hipStream_t st, bool f64, int B, int W, int NS, int S, const double *pw, const double *lut,
                                const double *sptab, const int *nspl, const double *meta, const double *aux, const double *runs,
                                void *x, void *y, void *h, void *k, void *dth, double *k64, double *dth64);
hipError_t launch_lut_slopes(hipStream_t st, int B, const double *lut, const double *meta, double *slopes);
hipError_t launch_grid(hipStream_t st, int B, int W, int S, double dd, double *meta, double *aux, double *runs,
                       uint32_t *flags);
hipError_t launch_sample(hipStream_t st, bool f64, int B, int W, int S, const double *pw, const double *lut,
                         const double *slopes, const double *meta, const double *aux, const double *runs, void *x,
                         void *y, void *h, void *k, void *dth, double *k64 = nullptr, double *dth64 = nullptr);
// per-sample max_acceleration rows of a batch of routes (vap_limits.hip makes them); all NULL for plain paths
struct AccRowsV {
    const void *fwd = nullptr;   // [B][S] dtype: max_acc (= max_dec) of the forward step from sample i, MPG:194-196
    const void *bwd = nullptr;   // [B][S] dtype: max_acc of the backward step from sample i, MPG:256-257
    const void *dec = nullptr;   // [B]    dtype: max_dec of the backward sweep
};
// Velocity kernels: r64 = arithmetic (and the curvature / dtheta rows) in fp64; io64 = the caller's rows (vcap,
// acc, vel) are fp64.  r64 && !io64 is the fp64 recurrence behind fp32 outputs; every kernel then also leaves the
// velocities in fp64 ([B][S], what the time-domain resample integrates): launch_velocity_seq in `usq`,
// launch_velocity_lanes in `ufwd` (both rows are scratch the sweeps are done with), the others in `vhi`.
hipError_t launch_velocity_seq(hipStream_t st, bool r64, bool io64, bool fast, int B, int S, const double c[6], double sv,
                               double ev, const double *meta, const void *curv, const void *dth, const void *vcap,
                               const AccRowsV &acc, void *vel, void *usq);
int velocity_relax_max_samples(bool f64, bool limits = false);
int velocity_relax_acc_max_samples(bool f64);
// vcap: optional [B][S] per-sample initial velocities (NULL = plain paths)
hipError_t launch_velocity_relax(hipStream_t st, bool r64, bool io64, int B, int S, const double c[6], double sv, double ev,
                                 const double *meta, const void *curv, const void *dth, const void *vcap,
                                 const AccRowsV &acc, void *vel, uint32_t *flags, void *vhi = nullptr);
// rows longer than velocity_relax_max_samples(): two-level relaxation, synchronises the stream once per super-round
size_t velocity_long_state_bytes(bool f64, int B, int S);
size_t velocity_long_counter_bytes(bool f64, int B, int S);
hipError_t launch_velocity_long(hipStream_t st, bool f64, bool io64, int B, int S, const double c[6], double sv, double ev,
                                const double *meta, const void *curv, const void *dth, void *vel, uint32_t *flags,
                                void *ufwd, void *state, int *counters, void *vhi = nullptr);
// the same rows in one launch per direction: super-chunk interfaces handed on by look-back inside the launch (no host
// round trip, any number of super-chunks); the default for long rows
size_t velocity_chase_state_bytes(bool f64, int B, int S);
size_t velocity_chase_counter_bytes(int B);
hipError_t launch_velocity_chase(hipStream_t st, bool f64, bool io64, int B, int S, const double c[6], double sv, double ev,
                                 const double *meta, const void *curv, const void *dth, void *vel, uint32_t *flags,
                                 void *ufwd, void *state, int *counters, void *vhi = nullptr);
// res[i] = (float)(v64[i] - (double)v32[i]): the fp32 residual row the time domain adds back to the caller's fp32 row
hipError_t launch_velocity_residual(hipStream_t st, size_t n, const double *v64, const float *v32, float *res);
// K5w (vap_velocity_lanes.hip), fp64 recurrence only: lane per path, `group` paths per workgroup (0 = by batch size).
// ufwd: [B][S] doubles of scratch for the forward sweep's squared velocities (unused when io64: the rows are used in place)
// vres (fp32 rows): [B][S] floats, v64 - (double)(float)v64 of every velocity written (what the time domain adds back)
int velocity_lanes_group(int B);
hipError_t launch_velocity_lanes(hipStream_t st, bool io64, int B, int S, const double c[6], double sv, double ev,
                                 const double *meta, const void *curv, const void *dth, const void *vcap, const AccRowsV &acc,
                                 void *vel, void *ufwd, int group = 0, float *vres = nullptr);
// fp32, one wave per path, one launch per window of 2560 samples and direction (no host synchronisation)
size_t velocity_windows_state_bytes(int B, int S);
hipError_t launch_velocity_windows(hipStream_t st, int B, int S, const double c[6], double sv, double ev,
                                   const double *meta, const void *curv, const void *dth, void *vel, uint32_t *flags,
                                   void *ufwd, void *state, int *counters);
hipError_t launch_power(hipStream_t st, int n_seg, const double *seg, double *pw);
hipError_t launch_time_profile(hipStream_t st, bool f64, int B, int W, int S, const double *segments, const double *lut,
                               const double *meta, const void *vel, double max_acc, double max_dec, double dt, int cap,
                               double *rows, int *counts, int *nodes_map, uint32_t *flags, RouteTables rt = RouteTables(),
                               const int *node_reverse = nullptr, const float *vres = nullptr,
                               int time_kernel = 0);   // vres: fp32 rows only, see k_time_integrate
// vap_limits.hip: sample of every event (node / action point), then the per-sample limit rows
struct LimitInputs {
    const double *node_mv = nullptr, *node_ma = nullptr;   // [B][W] per-node max_velocity / max_acceleration (<= 0: none)
    const int *node_stop = nullptr;                        // [B][W]
    const double *ap_t = nullptr, *ap_mv = nullptr, *ap_ma = nullptr;   // [B][M] action points in route order (t: +inf = padding)
    const int *ap_stop = nullptr;                          // [B][M]
    double max_vel = 0, max_acc = 0, end_vel = 0;
};
// node_k [B][W], ap_k [B][M] receive the samples (INT_MAX: never); ev_* [B][W-2+M] are scratch for the merged list
hipError_t launch_route_limits(hipStream_t st, bool f64, int B, int W, int M, int S, const double *lut, const double *meta,
                               const double *aux, const double *runs, const LimitInputs &in, int *node_k, int *ap_k,
                               int *ev_k, double *ev_mv, double *ev_ma, int *ev_stop, void *vcap, void *acc_fwd,
                               void *acc_bwd, void *dec_bwd, RouteTables rt = RouteTables());
// waits of nodes / action points and actions_map on top of the rows of launch_time_profile (vap_time.hip)
hipError_t launch_time_waits(hipStream_t st, int B, int W, int M, int cap_in, int cap_out, double dt, const double *segments,
                             const double *lut, const double *meta, const double *rows_in, const int *counts_in,
                             const int *nodes_in, const double *node_wait, const double *ap_t, const double *ap_wait,
                             double *rows_out, int *counts_out, int *nodes_out, int *actions_out, uint32_t *flags,
                             RouteTables rt = RouteTables(), const double *node_turn = nullptr, const int *node_reverse = nullptr,
                             double max_vel = 0, double max_acc = 0, double track_width = 0);
hipError_t launch_eval(hipStream_t st, int W, const double *seg, double t_max, int order, int n, const double *t,
                       double *out);
hipError_t launch_basis(hipStream_t st, int order, int n, const double *t, double *out);   // out [n][6]
hipError_t launch_lookup(hipStream_t st, int W, const double *seg, double t_max, const double *lut, int what,
                         int n, const double *in, double *out);

}  // namespace vap
