// vap_route.hip — one general route (any node / action-point attributes) end to end on the device:
// what the GUI asks for when it saves or graphs a route.  fp64 throughout, reference operation order.
//
//   vap_route_create           SM:42-172 build_path incl. reverse / turn splits, QHS:30-219 fit,
//                              QHS:543-590 split tangents, SM:426-475 lookup table (per spline + offsets)
//   vap_route_eval / _lookup   SM:204-241, 291-346, 550-580 (multi-spline parameter mapping SM:243-275)
//   vap_route_forward_backward MPG:70-316 incl. per-node / action-point limits and boundary_map
//   vap_route_motion_profile   MPG:389-628 incl. turn / wait insertion (MPG:319-346, ODM:4-69)
//
// These routes are GUI-sized (a few thousand samples) and dominated by strictly sequential state
// machines, so the kernels favour fidelity over occupancy: the per-sample properties are evaluated by
// a thread per sample, everything sequential by one lane.  The batched throughput path is
// vap_kernels.hip; this file is the completeness path behind the same drop-in classes.
#include <cmath>
#include <vector>

#include "vap_device.h"
#include "vap_internal.h"
#include "vap_kernels.h"

namespace vap {

struct RouteDev {
    int W, M;
    // inputs
    const double *wp;          // [W][2]
    const int *rev, *stop;     // [W]
    const double *turn, *wait, *maxv, *maxa;   // [W]
    const double *tangent, *mag;               // [W][2] (NaN = None), [W][2] (in, out)
    const double *ap_t, *ap_wait, *ap_maxv, *ap_maxa;   // [M]
    const int *ap_stop;
    // fitted
    int *sp_start, *sp_npts, *sp_seg0;         // [W]
    double *sp_tmax, *sp_dist0, *sp_param0;    // [W]
    double *seg, *seglen;                      // [W-1][12], [W-1]
    double *lut;                               // [n_splines][lut_n] partial distances
    double *lut_mag;                           // [n_splines][lut_n] scratch of the table build (sizes other than 1000)
    int lut_n;                                 // samples per spline of the arc-length table   (SM:427 min_samples)
    int spn;                                   // property-table entries per node               (SM:477 samples_per_node)
    double *info;                              // [0]=n_splines [1]=total [2]=status
};

// ---------------------------------------------------------------------------------------------------
// fit of one spline (QHS:30-138 + 149-219 + 543-590), sequential
// ---------------------------------------------------------------------------------------------------
__device__ void route_fit_spline(int npts, const double *pts, const double *tin, const double *tout,
                                 const double *start_tan, const double *end_tan, double *seg, double *seglen,
                                 double *t_max, double *work /* 5*npts doubles */)
{
    const int G = npts - 1;
    double *dist = work, *fd = work + npts, *sd = work + 3 * npts;
    double cum = 0.0;
    for (int i = 0; i < G; i++) {
        const double dx = pts[2 * (i + 1)] - pts[2 * i], dy = pts[2 * (i + 1) + 1] - pts[2 * i + 1];
        dist[i] = sqrt(dx * dx + dy * dy);
        cum += dist[i];
    }
    *t_max = (cum == 0.0) ? (double)G : cum * (double)G / cum;   // QHS:719-736
    for (int i = 0; i < npts; i++) {   // QHS:163-195
        if (i == 0) {
            const double cx = pts[2] - pts[0], cy = pts[3] - pts[1];
            if (npts == 2 && end_tan) { fd[0] = cx; fd[1] = cy; }
            else { fd[0] = cx / dist[0]; fd[1] = cy / dist[0]; }
        } else if (i == npts - 1) {
            const double cx = pts[2 * i] - pts[2 * (i - 1)], cy = pts[2 * i + 1] - pts[2 * (i - 1) + 1];
            if (npts == 2 && start_tan) { fd[2 * i] = cx; fd[2 * i + 1] = cy; }
            else { fd[2 * i] = cx / dist[G - 1]; fd[2 * i + 1] = cy / dist[G - 1]; }
        } else {
            const double px = (pts[2 * i] - pts[2 * (i - 1)]) / dist[i - 1];
            const double py = (pts[2 * i + 1] - pts[2 * (i - 1) + 1]) / dist[i - 1];
            const double nx = (pts[2 * (i + 1)] - pts[2 * i]) / dist[i];
            const double ny = (pts[2 * (i + 1) + 1] - pts[2 * i + 1]) / dist[i];
            fd[2 * i] = (px + nx) / 2;
            fd[2 * i + 1] = (py + ny) / 2;
        }
    }
    for (int i = 0; i < npts; i++) {   // QHS:197-219
        double sx = 0.0, sy = 0.0;
        if (i > 0 && i < npts - 1) {
            const double avg = (dist[i - 1] + dist[i]) / 2;
            sx = (fd[2 * (i + 1)] - fd[2 * (i - 1)]) / (avg * 0.5);
            sy = (fd[2 * (i + 1) + 1] - fd[2 * (i - 1) + 1]) / (avg * 0.5);
        }
        sd[2 * i] = sx;
        sd[2 * i + 1] = sy;
    }
    for (int i = 0; i < G; i++) {   // QHS:76-127
        const double L = dist[i];
        double *r = seg + (size_t)i * 12;
        seglen[i] = L;
        r[0] = pts[2 * i];       r[1] = pts[2 * i + 1];
        r[2] = pts[2 * (i + 1)]; r[3] = pts[2 * (i + 1) + 1];
        if (L > 0) {
            const double L2 = L * L;
            r[4] = fd[2 * i] * L;          r[5] = fd[2 * i + 1] * L;
            r[6] = fd[2 * (i + 1)] * L;    r[7] = fd[2 * (i + 1) + 1] * L;
            r[8] = sd[2 * i] * L2;         r[9] = sd[2 * i + 1] * L2;
            r[10] = sd[2 * (i + 1)] * L2;  r[11] = sd[2 * (i + 1) + 1] * L2;
            if (!isnan(tout[2 * i])) { r[4] = tout[2 * i]; r[5] = tout[2 * i + 1]; }
            if (!isnan(tin[2 * (i + 1)])) { r[6] = tin[2 * (i + 1)]; r[7] = tin[2 * (i + 1) + 1]; }
        } else {
            r[4] = fd[2 * i];          r[5] = fd[2 * i + 1];
            r[6] = fd[2 * (i + 1)];    r[7] = fd[2 * (i + 1) + 1];
            r[8] = sd[2 * i];          r[9] = sd[2 * i + 1];
            r[10] = sd[2 * (i + 1)];   r[11] = sd[2 * (i + 1) + 1];
        }
    }
    // QHS:129-132 -> 543-590; the start tangent lands in the LAST segment's row 2 (QHS:561, quirk Q3)
    if (start_tan) { seg[(size_t)(G - 1) * 12 + 4] = start_tan[0]; seg[(size_t)(G - 1) * 12 + 5] = start_tan[1]; }
    if (end_tan) { seg[(size_t)(G - 1) * 12 + 6] = end_tan[0]; seg[(size_t)(G - 1) * 12 + 7] = end_tan[1]; }
}

// SM:42-172 build_path.  One lane; `work` holds 9*W doubles.
__global__ void k_route_fit(RouteDev r, double *work)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int W = r.W;
    double *tin = work, *tout = work + 2 * W, *fw = work + 4 * W;
    for (int i = 0; i < W; i++) {
        const bool has = !isnan(r.tangent[2 * i]);
        for (int c = 0; c < 2; c++) {
            tin[2 * i + c] = has ? r.tangent[2 * i + c] * r.mag[2 * i] : NAN;        // SM:65-75
            tout[2 * i + c] = has ? r.tangent[2 * i + c] * r.mag[2 * i + 1] : NAN;
        }
    }
    int n_spl = 0, cur_start = 0, seg0 = 0, status = 0;
    bool have_start = false;
    double start_tan[2] = {0, 0};
    for (int i = 1; i < W && status == 0; i++) {
        const bool split = r.rev[i] || r.turn[i] != 0;
        if (!(split || i == W - 1)) continue;
        double end_tan[2], this_start[2] = {start_tan[0], start_tan[1]};
        const bool this_have_start = have_start;
        bool have_end = false;
        have_start = false;   // SM:79-81
        if (split) {          // SM:84-158
            if (i >= W - 1) { status = 1; break; }   // points[i+1]: IndexError in the reference
            const double *pm = r.wp + 2 * (i - 1), *pi = r.wp + 2 * i, *pn = r.wp + 2 * (i + 1);
            const double prev_len = sqrt((pi[0] - pm[0]) * (pi[0] - pm[0]) + (pi[1] - pm[1]) * (pi[1] - pm[1]));
            const double next_len = sqrt((pn[0] - pi[0]) * (pn[0] - pi[0]) + (pn[1] - pi[1]) * (pn[1] - pi[1]));
            const double ps = prev_len > 0 ? 1.0 / prev_len : 1.0, ns = next_len > 0 ? 1.0 / next_len : 1.0;
            double pv[2] = {(pi[0] - pm[0]) * ps, (pi[1] - pm[1]) * ps};
            const double nv[2] = {(pn[0] - pi[0]) * ns, (pn[1] - pi[1]) * ns};
            const double min_len = prev_len < next_len ? prev_len : next_len;
            const bool has_tan = !isnan(r.tangent[2 * i]);
            if (r.turn[i] != 0) {   // SM:103-132
                double ang = r.turn[i] * (M_PI / 180.0);
                if (r.rev[i]) ang = ang + M_PI;
                const double c = cos(ang), s = sin(ang);
                double nt[2] = {c * pv[0] + (-s) * pv[1], s * pv[0] + c * pv[1]};
                nt[0] *= min_len; nt[1] *= min_len;
                pv[0] *= min_len; pv[1] *= min_len;
                if (has_tan) {
                    const double *tg = r.tangent + 2 * i;
                    const double im = r.mag[2 * i], om = r.mag[2 * i + 1];
                    pv[0] = tg[0] * im; pv[1] = tg[1] * im;
                    nt[0] = (tg[0] * c + tg[1] * s) * -1;
                    nt[1] = (tg[0] * (-s) + tg[1] * c) * -1;
                    nt[0] *= om; nt[1] *= om;
                }
                end_tan[0] = pv[0]; end_tan[1] = pv[1];
                start_tan[0] = nt[0]; start_tan[1] = nt[1];
            } else {   // reverse node, SM:134-158
                double dv[2] = {pv[0] - nv[0], pv[1] - nv[1]};
                const double dn = sqrt(dv[0] * dv[0] + dv[1] * dv[1]);
                if (dn > 0) { dv[0] /= dn; dv[1] /= dn; }
                dv[0] *= min_len; dv[1] *= min_len;
                if (has_tan) { dv[0] = r.tangent[2 * i] * r.mag[2 * i]; dv[1] = r.tangent[2 * i + 1] * r.mag[2 * i]; }
                end_tan[0] = dv[0]; end_tan[1] = dv[1];
                start_tan[0] = -1 * dv[0]; start_tan[1] = -1 * dv[1];
                if (has_tan) {
                    start_tan[0] = -1 * r.tangent[2 * i] * r.mag[2 * i + 1];
                    start_tan[1] = -1 * r.tangent[2 * i + 1] * r.mag[2 * i + 1];
                }
            }
            have_end = true;
            have_start = true;
        }
        const int npts = i - cur_start + 1;
        r.sp_start[n_spl] = cur_start;
        r.sp_npts[n_spl] = npts;
        r.sp_seg0[n_spl] = seg0;
        route_fit_spline(npts, r.wp + 2 * cur_start, tin + 2 * cur_start, tout + 2 * cur_start,
                         this_have_start ? this_start : nullptr, have_end ? end_tan : nullptr,
                         r.seg + (size_t)seg0 * 12, r.seglen + seg0, &r.sp_tmax[n_spl], fw);
        n_spl++;
        seg0 += npts - 1;
        if (split && i < W - 1) cur_start = i;   // SM:165-168
    }
    r.info[0] = (double)n_spl;
    r.info[2] = (double)status;
}

// SM:426-475, one workgroup per spline: partial (un-offset) distances of that spline.  The default 1000-sample table
// is built in LDS; any other min_samples (vap_route_set_table_sizes) goes through the global scratch r.lut_mag —
// same operations in the same order (np.linspace, the trapezoid increments, np.cumsum's sequential sum).
__global__ __launch_bounds__(256) void k_route_lut(RouteDev r)
{
    constexpr int kPad = (kLutN + 15) / 16 * 16;
    __shared__ double mag_s[kPad], cum_s[kPad];
    const int si = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int n = r.lut_n;
    const bool in_lds = n == kLutN;
    double *mag = in_lds ? mag_s : r.lut_mag + (size_t)si * n;
    double *cum = in_lds ? cum_s : r.lut + (size_t)si * n;
    const int G = r.sp_npts[si] - 1;
    const double t_max = r.sp_tmax[si];
    const double *seg = r.seg + (size_t)r.sp_seg0[si] * 12;
    for (int j = tid; j < n; j += nt) {
        const double t = linspace_at(t_max, n, j);
        double dx, dy;
        hermite_eval_ref(seg, t_max, G, 1, t, dx, dy);
        mag[j] = sqrt(dx * dx + dy * dy);
    }
    __syncthreads();
    const double dt = linspace_at(t_max, n, 1) - linspace_at(t_max, n, 0);
    for (int j = tid; j < n; j += nt) cum[j] = (j > 0) ? (mag[j - 1] + mag[j]) * 0.5 * dt : 0.0;
    __syncthreads();
    if (tid == 0) {
        double acc = 0.0;
        for (int j = 0; j < n; j++) { acc += cum[j]; cum[j] = acc; }
    }
    __syncthreads();
    if (in_lds)
        for (int j = tid; j < n; j += nt) r.lut[(size_t)si * n + j] = cum[j];
}

// SM:456-464: distance / parameter offsets of the concatenated table
__global__ void k_route_offsets(RouteDev r)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = (int)r.info[0];
    double current_dist = 0.0, prev_param = 0.0;
    for (int si = 0; si < n; si++) {
        r.sp_dist0[si] = current_dist;
        r.sp_param0[si] = prev_param;
        current_dist = r.lut[(size_t)si * r.lut_n + r.lut_n - 1] + current_dist;   // spline_distances[-1]
        prev_param += r.sp_tmax[si] - 0.0;
    }
    r.info[1] = current_dist;
}

// ---------------------------------------------------------------------------------------------------
// accessors on the fitted route
// ---------------------------------------------------------------------------------------------------
// SM:243-275 + QHS evaluators
__device__ __forceinline__ void route_eval(const RouteDev &r, int n_spl, double t, int order, double &ox, double &oy)
{
    int cumulative = 0, si = n_spl - 1;
    double lt = t;
    for (int i = 0; i < n_spl; i++) {
        const int end = cumulative + r.sp_npts[i] - 1;
        if (t <= (double)end || i == n_spl - 1) { si = i; lt = t - (double)cumulative; break; }
        cumulative = end;
    }
    hermite_eval_ref(r.seg + (size_t)r.sp_seg0[si] * 12, r.sp_tmax[si], r.sp_npts[si] - 1, order, lt, ox, oy);
}

// lookup_table.distances / parameters entry e of the concatenated table (SM:457-462)
__device__ __forceinline__ double route_lut_d(const RouteDev &r, int e)
{
    const int si = e / r.lut_n, j = e % r.lut_n;
    return r.lut[(size_t)si * r.lut_n + j] + r.sp_dist0[si];
}
__device__ __forceinline__ double route_lut_p(const RouteDev &r, int e)
{
    const int si = e / r.lut_n, j = e % r.lut_n;
    return linspace_at(r.sp_tmax[si], r.lut_n, j) + r.sp_param0[si];
}

// SM:291-318
__device__ double route_distance_to_time(const RouteDev &r, int n_spl, double total, double s)
{
    if (s <= 0) return 0.0;
    if (s >= total) return (double)(r.W - 1);
    int lo = 0, hi = n_spl * r.lut_n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (route_lut_d(r, mid) < s) lo = mid + 1;
        else hi = mid;
    }
    if (lo == 0) return route_lut_p(r, 0);
    const double d0 = route_lut_d(r, lo - 1), d1 = route_lut_d(r, lo);
    const double t0 = route_lut_p(r, lo - 1), t1 = route_lut_p(r, lo);
    return t0 + (t1 - t0) * (s - d0) / (d1 - d0);
}

// SM:332-346, 550-580 on the (never materialised) property table: entry jj of linspace(0, W-1, spn*W)
__device__ double route_property_entry(const RouteDev &r, int n_spl, int tab_n, int jj, int which)
{
    const double tp = linspace_at((double)(r.W - 1), tab_n, jj);
    double d1x, d1y;
    route_eval(r, n_spl, tp, 1, d1x, d1y);
    if (which == 2) return atan2(d1y, d1x);
    double d2x, d2y;
    route_eval(r, n_spl, tp, 2, d2x, d2y);
    const double ss = d1x * d1x + d1y * d1y;
    const double num = d1x * d2y - d1y * d2x;
    return (ss >= 1e-10) ? num / (ss * sqrt(ss)) : 0.0;
}
// _interpolate_property in full (SM:550-580).  With the default samples_per_node the two neighbouring table
// parameters never have equal fractional parts (their distance is below 1), so the reference always takes the step
// lookup at SM:577-578 — what table_index returns and what the batched kernels build on.  Only a table with one
// entry per node (samples_per_node = 1: the parameters are the integers) reaches the linear interpolation.
__device__ double route_property(const RouteDev &r, int n_spl, double t, int which /*1 curvature, 2 heading*/)
{
    const double end_param = (double)(r.W - 1);
    const int tab_n = r.W * r.spn;
    const int idx = table_search(t, tab_n, end_param);
    if (idx == 0) return route_property_entry(r, n_spl, tab_n, 0, which);
    const double t0 = linspace_at(end_param, tab_n, idx - 1), t1 = linspace_at(end_param, tab_n, idx);
    if (t0 - floor(t0) != t1 - floor(t1)) {
        const double frac = t - floor(t);   // t % 1 for t >= 0
        return route_property_entry(r, n_spl, tab_n, frac > 0.5 ? idx - 1 : idx, which);
    }
    const double v0 = route_property_entry(r, n_spl, tab_n, idx - 1, which);
    const double v1 = route_property_entry(r, n_spl, tab_n, idx, which);
    return v0 + (v1 - v0) * (t - t0) / (t1 - t0);
}

__global__ void k_route_eval(RouteDev r, int order, int n, const double *__restrict__ t, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x, y;
    route_eval(r, (int)r.info[0], t[i], order, x, y);
    out[2 * i] = x;
    out[2 * i + 1] = y;
}

__global__ void k_route_lookup(RouteDev r, int what, int n, const double *__restrict__ in, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int n_spl = (int)r.info[0];
    out[i] = what == 0 ? route_distance_to_time(r, n_spl, r.info[1], in[i]) : route_property(r, n_spl, in[i], what);
}

// MPG:112-122 / 172-175 sampling: thread per sample
__global__ void k_route_grid(RouteDev r, int N, double dd, const double *__restrict__ runs, int n_runs, double *__restrict__ t,
                             double *__restrict__ kap, double *__restrict__ th, double *__restrict__ x,
                             double *__restrict__ y)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const int n_spl = (int)r.info[0];
    const double total = r.info[1];
    // MPG:112-122: the reference's accumulated distance grid, exactly (vap_device.h, GridRuns)
    int cur = grid_run_hint(dd, (long)k, n_runs);
    const double sk = grid_s(runs, n_runs, (long)k, cur);
    const double s = (k == N - 1) ? total : sk;
    const double tt = route_distance_to_time(r, n_spl, total, s);
    t[k] = tt;
    kap[k] = route_property(r, n_spl, tt, 1);
    th[k] = route_property(r, n_spl, tt, 2);
    double px, py;
    route_eval(r, n_spl, tt, 0, px, py);
    x[k] = px;
    y[k] = py;
}

// Python float % 1 for non-negative values

// MPG:84-176 bookkeeping of the sampling loop (one lane): initial velocities, boundary_map / max_accels
// expanded to per-sample constraint values for the two sweeps.
//   vinit[k]  velocities[k] before the passes
//   acc_f[k]  constraints.max_acc (= max_dec) in force at forward step k      (MPG:194-196)
//   acc_b[k]  constraints.max_acc in force at backward step k                  (MPG:256-257)
//   out[0]    constraints.max_dec during the whole backward sweep (what the forward sweep left)
__global__ void k_route_caps(RouteDev r, int N, const double *__restrict__ t, double max_vel, double max_acc,
                             double max_dec, double end_vel, double *__restrict__ vinit, double *__restrict__ acc_f,
                             double *__restrict__ acc_b, int *__restrict__ bmap, double *__restrict__ max_accels,
                             double *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int n_acc = 0;
    for (int k = 0; k < N; k++) bmap[k] = -1;
    double max_velocity = max_vel;
    max_accels[n_acc++] = r.maxa[0] > 0 ? r.maxa[0] : max_acc;   // MPG:100-104
    if (r.maxv[0] > 0) max_velocity = r.maxv[0];
    bmap[0] = 0;
    const double t_end = (double)(r.W - 1);                       // distance_to_time(total_dist)
    double prev_t = 0;
    int node_num = 0, action_idx = 0;
    for (int i = 0; i < N - 1; i++) {                             // the while loop, MPG:112-167
        const double tt = t[i];
        vinit[i] = max_velocity;
        if (mod1(prev_t) > mod1(tt) && tt < t_end) {              // MPG:124-140
            node_num += 1;
            if (r.stop[node_num]) vinit[i] = 0.01;
            max_velocity = r.maxv[node_num] > 0 ? r.maxv[node_num] : max_vel;
            max_accels[n_acc++] = r.maxa[node_num] > 0 ? r.maxa[node_num] : max_acc;
            if (node_num < r.W - 1) bmap[i] = n_acc - 1;
        }
        if (action_idx < r.M && prev_t < r.ap_t[action_idx] && tt >= r.ap_t[action_idx]) {   // MPG:142-163
            max_velocity = r.ap_maxv[action_idx] > 0 ? r.ap_maxv[action_idx] : max_vel;
            if (r.ap_stop[action_idx]) vinit[i] = 0.01;
            max_accels[n_acc++] = r.ap_maxa[action_idx] > 0 ? r.ap_maxa[action_idx] : max_acc;
            bmap[i] = n_acc - 1;
            action_idx += 1;
        }
        prev_t = tt;
    }
    vinit[N - 1] = end_vel;                                       // MPG:172
    max_accels[n_acc++] = max_acc;                                // MPG:176
    double c_acc = max_acc, c_dec = max_dec;
    for (int k = 0; k < N - 1; k++) {                             // forward, MPG:193-196
        if (bmap[k] >= 0) { c_acc = max_accels[bmap[k]]; c_dec = max_accels[bmap[k]]; }
        acc_f[k] = c_acc;
    }
    acc_f[N - 1] = c_acc;
    out[0] = c_dec;
    for (int k = N - 1; k > 0; k--) {                             // backward, MPG:255-257
        if (bmap[k] >= 0) c_acc = max_accels[bmap[k] + 1];
        acc_b[k] = c_acc;
    }
    acc_b[0] = c_acc;
}

// MPG:188-311, literal statement order (squared-velocity space), one lane
__global__ void k_route_velocity(int N, VelConsts<double> c, double dd, double start_vel, double end_vel,
                                 const double *__restrict__ kap, const double *__restrict__ th,
                                 const double *__restrict__ vinit, const double *__restrict__ acc_f,
                                 const double *__restrict__ acc_b, const double *__restrict__ decb,
                                 double *__restrict__ v)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double twodd = 2.0 * dd;
    double u = start_vel * start_vel, wprev = 0.0;
    v[0] = u;
    for (int i = 0; i < N - 1; i++) {
        const double ca = acc_f[i];
        const SampleLimits<double> L = sample_limits(c, fabs(kap[i]), ca, ca);
        const double un = vinit[i + 1] * vinit[i + 1];
        u = forward_step(c, L, ca, twodd, u, wprev, fabs(th[i + 1] - th[i]), un);
        v[i + 1] = u;
    }
    const double c_dec = decb[0];
    u = end_vel * end_vel;
    wprev = 0.0;
    for (int i = N - 1; i > 0; i--) {
        const double ca = acc_b[i];
        const SampleLimits<double> L = sample_limits(c, fabs(kap[i]), ca, c_dec);
        const double up = backward_step(c, L, ca, twodd, u, wprev, fabs(th[i - 1] - th[i]), v[i - 1]);
        v[i] = sqrt(u);
        u = up;
    }
    v[0] = sqrt(u);
}

// ---------------------------------------------------------------------------------------------------
// MPG:389-628 time-domain resample, one lane.  rows[T][8] = {time, position, linear velocity,
// acceleration, heading, angular velocity, x, y}.
// ---------------------------------------------------------------------------------------------------
#define ROW(i) (rows + (size_t)(i) * 8)

// MPG:487-507 handle_turn (with MPG:319-346 motion_profile_angle and ODM:4-69 inlined); returns rows added or -1
__device__ long emit_turn(double angle, double vmax, double amax, double tw, double dt, double *rows, long T, long cap,
                          double &current_time)
{
    const double arc = fabs(angle) * tw / 2;
    double t_acc = vmax / amax;
    const double d_acc = 0.5 * amax * (t_acc * t_acc);
    double vpeak = vmax, total_time;
    if (2 * d_acc > arc) {
        t_acc = sqrt(arc / amax);
        vpeak = amax * t_acc;
        total_time = 2 * t_acc;
    } else {
        total_time = 2 * t_acc + (arc - 2 * d_acc) / vpeak;
    }
    const long n = (long)ceil((total_time + dt) / dt);   // np.arange(0, total_time + dt, dt)
    if (T + n > cap) return -1;
    const double start_heading = ROW(T - 1)[4];
    const double lastpos = ROW(T - 1)[1], lx = ROW(T - 1)[6], ly = ROW(T - 1)[7];
    double accum = 0, prev_h = 0;
    for (long i = 0; i < n; i++) {
        const double tt = (double)i * dt;
        double vel;
        if (tt <= t_acc) vel = amax * tt;
        else if (tt <= total_time - t_acc) vel = vpeak;
        else vel = vpeak - amax * (tt - (total_time - t_acc));
        double h = accum / (tw / 2) * (angle > 0 ? -1 : 1);
        const double raw = h;
        accum += vel * dt;
        const double w = i == 0 ? 0.0 : (raw - prev_h) / dt;   // differences of the UN-wrapped headings
        prev_h = raw;
        while (h + start_heading > M_PI) h -= 2 * M_PI;
        while (h + start_heading < -M_PI) h += 2 * M_PI;
        double *r = ROW(T + i);
        r[0] = current_time + i * dt; r[1] = lastpos; r[2] = 0; r[3] = 0;
        r[4] = start_heading + h; r[5] = w; r[6] = lx; r[7] = ly;
    }
    current_time = current_time + n * dt;
    return n;
}

// MPG:509-518 handle_wait
__device__ long emit_wait(double wait_time, double dt, double *rows, long T, long cap, double &current_time)
{
    const long steps = (long)(wait_time / dt);
    if (T + steps > cap) return -1;
    const double lh = ROW(T - 1)[4], lx = ROW(T - 1)[6], ly = ROW(T - 1)[7];
    for (long i = 0; i < steps; i++) {
        double *r = ROW(T + i);
        r[0] = current_time + i * dt; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = lh; r[5] = 0; r[6] = lx; r[7] = ly;
    }
    current_time = current_time + steps * dt;
    return steps;
}

__global__ void k_route_time(RouteDev r, int N, const double *__restrict__ vel, double max_vel, double max_acc,
                             double max_dec, double tw, double dt, double dd, long cap, double *__restrict__ rows,
                             long *__restrict__ nodes_map, long *__restrict__ actions_map, long *__restrict__ counts)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n_spl = (int)r.info[0];
    const double total_length = r.info[1];
    long T = 0, nn = 0, na = 0, status = 0;
    nodes_map[nn++] = 0;   // MPG:420
    double current_time = 0, current_pos = 0, current_vel = vel[0];
    bool is_reversed = false;
    if (r.rev[0]) is_reversed = !is_reversed;
    if (r.turn[0] != 0) status = -2;   // MPG:440: headings[-1] of an empty list
    if (status == 0 && r.wait[0] > 0) {   // MPG:459-476
        const long steps = (long)(r.wait[0] / dt);
        double h = -1 * route_property(r, n_spl, 0.0, 2);
        if (is_reversed) h -= M_PI;
        if (h > M_PI) h -= 2 * M_PI;
        if (h < -M_PI) h += 2 * M_PI;
        double px, py;
        route_eval(r, n_spl, 0.0, 0, px, py);
        if (T + steps > cap) status = -1;
        else {
            for (long i = 0; i < steps; i++) {
                double *q = ROW(T + i);
                q[0] = current_time + i * dt; q[1] = 0; q[2] = 0; q[3] = 0; q[4] = h; q[5] = 0; q[6] = px; q[7] = py;
            }
            T += steps;
            current_time += steps * dt;
        }
    }
    double prev_t = 0;
    int action_idx = 0, node_idx = 0;
    const double end_param = (double)(r.W - 1);
    while (status == 0 && current_pos < total_length) {   // MPG:523-600
        const double t = route_distance_to_time(r, n_spl, total_length, current_pos);
        if (mod1(t) < mod1(prev_t) && t < end_param) {   // MPG:527-544
            nodes_map[nn++] = T;
            node_idx += 1;
            if (r.turn[node_idx] != 0) {
                if (T == 0) { status = -2; break; }
                const long n = emit_turn(r.turn[node_idx] * (M_PI / 180.0), max_vel, max_acc, tw, dt, rows, T, cap,
                                         current_time);
                if (n < 0) { status = -1; break; }
                T += n;
            }
            if (r.rev[node_idx]) is_reversed = !is_reversed;
            if (r.wait[node_idx] > 0) {
                if (T == 0) { status = -2; break; }
                const long n = emit_wait(r.wait[node_idx], dt, rows, T, cap, current_time);
                if (n < 0) { status = -1; break; }
                T += n;
            }
        }
        if (action_idx < r.M) {   // MPG:547-553
            const double at = r.ap_t[action_idx];
            if (prev_t < at && at < t) {
                actions_map[na++] = T;
                if (r.ap_wait[action_idx] > 0) {
                    if (T == 0) { status = -2; break; }
                    const long n = emit_wait(r.ap_wait[action_idx], dt, rows, T, cap, current_time);
                    if (n < 0) { status = -1; break; }
                    T += n;
                }
                action_idx += 1;
            }
        }
        prev_t = t;
        const double curvature = route_property(r, n_spl, t, 1);
        double heading = route_property(r, n_spl, t, 2) - (is_reversed ? M_PI : 0);
        heading = py_mod(heading + M_PI, 2 * M_PI) - M_PI;
        heading *= -1;
        double px, py;
        route_eval(r, n_spl, t, 0, px, py);
        double target_vel = lerp_grid(current_pos, dd, 1.0 / dd, vel, N);
        const double next_target_vel = lerp_grid(current_pos + dd, dd, 1.0 / dd, vel, N);
        target_vel = (target_vel + next_target_vel) / 2;
        if (!(target_vel > 0.001)) target_vel = 0.001;
        const double accel = clip((target_vel - current_vel) / dt, -max_dec, max_acc);
        const double angular_vel = target_vel * curvature * -1;
        current_vel = clip(current_vel + accel * dt, 0, target_vel);
        double delta_pos = current_vel * dt + 0.5 * accel * dt * dt;
        if (current_vel <= 0.1) delta_pos = 0.1 * dt + 0.5 * accel * dt * dt;
        current_pos += delta_pos;
        if (T + 1 > cap) { status = -1; break; }
        double *q = ROW(T);
        q[0] = current_time; q[1] = current_pos; q[2] = current_vel * (is_reversed ? -1 : 1);
        q[3] = accel * (is_reversed ? -1 : 1); q[4] = heading; q[5] = angular_vel; q[6] = px; q[7] = py;
        T += 1;
        current_time += dt;
    }
    counts[0] = T;
    counts[1] = nn;
    counts[2] = na;
    counts[3] = status;
}

}  // namespace vap

// =====================================================================================================
// C-ABI
// =====================================================================================================
struct vap_route {
    vap_ctx *ctx = nullptr;
    int W = 0, M = 0, n_splines = 0;
    double total = 0.0;
    void *blob = nullptr;   // one device allocation holding everything below
    void *lut_blob = nullptr;   // arc-length table of a size other than the default (vap_route_set_table_sizes)
    vap::RouteDev d{};
    // cached distance-domain result of the last forward_backward call
    int N = 0;
    double dd = 0.0;
    void *work = nullptr;
    size_t work_cap = 0;
};

namespace {

template <typename T>
T *carve(char *&p, size_t n)
{
    T *r = reinterpret_cast<T *>(p);
    p += (n * sizeof(T) + 63) / 64 * 64;
    return r;
}

int ensure_work(vap_route *rt, size_t bytes)
{
    if (bytes <= rt->work_cap) return VAP_OK;
    if (rt->work) HIP_TRY(hipFree(rt->work));
    rt->work = nullptr;
    rt->work_cap = 0;
    HIP_TRY(hipMalloc(&rt->work, bytes + 4096));
    rt->work_cap = bytes + 4096;
    return VAP_OK;
}

}  // namespace

extern "C" {

int vap_route_create(vap_ctx *ctx, const vap_route_desc *desc, vap_route **out)
{
    VAP_TRY(vap_set_device(ctx));
    if (!desc || !out) return vap_fail(VAP_ERR_INVALID, "null argument");
    *out = nullptr;
    const int W = desc->n_nodes, M = desc->n_actions;
    if (W < 2 || !desc->waypoints) return vap_fail(VAP_ERR_INVALID, "a path needs at least 2 waypoints (got %d)", W);
    if (W > vap::kMaxWaypoints) return vap_fail(VAP_ERR_UNSUPPORTED, "W=%d exceeds %d", W, vap::kMaxWaypoints);
    if (M < 0 || (M > 0 && !desc->ap_t)) return vap_fail(VAP_ERR_INVALID, "bad action points");
    hipStream_t st = ctx->stream;
    // host staging of the inputs (defaults as in gui/node.py:17-51)
    std::vector<double> h_wp(desc->waypoints, desc->waypoints + 2 * (size_t)W);
    auto dcol = [&](const double *src, size_t n, double fill) {
        std::vector<double> v(n, fill);
        if (src) v.assign(src, src + n);
        return v;
    };
    auto icol = [&](const int *src, size_t n) {
        std::vector<int> v(n, 0);
        if (src) v.assign(src, src + n);
        return v;
    };
    std::vector<double> h_turn = dcol(desc->turn, W, 0), h_wait = dcol(desc->wait_time, W, 0),
                        h_maxv = dcol(desc->max_velocity, W, 0), h_maxa = dcol(desc->max_acceleration, W, 0),
                        h_tan = dcol(desc->tangent, 2 * (size_t)W, NAN), h_mag = dcol(desc->magnitudes, 2 * (size_t)W, 0),
                        h_apt = dcol(desc->ap_t, M, 0), h_apw = dcol(desc->ap_wait_time, M, 0),
                        h_apv = dcol(desc->ap_max_velocity, M, 0), h_apa = dcol(desc->ap_max_acceleration, M, 0);
    std::vector<int> h_rev = icol(desc->is_reverse, W), h_stop = icol(desc->stop, W), h_aps = icol(desc->ap_stop, M);

    vap_route *rt = new vap_route();
    rt->ctx = ctx;
    rt->W = W;
    rt->M = M;
    const size_t Mp = M > 0 ? M : 1;
    size_t bytes = 64 * 40 + sizeof(double) * (2 * W + 4 * W + 4 * W + 4 * Mp + 3 * W + 12 * (size_t)W + W + 8 + 9 * (size_t)W) +
                   sizeof(int) * (2 * W + Mp + 3 * W) + sizeof(double) * (size_t)(W - 1) * vap::kLutN;
    if (hipMalloc(&rt->blob, bytes) != hipSuccess) {
        delete rt;
        return vap_fail(VAP_ERR_HIP, "hipMalloc(%zu) failed", bytes);
    }
    char *p = (char *)rt->blob;
    vap::RouteDev &d = rt->d;
    d.W = W;
    d.M = M;
    double *wp = carve<double>(p, 2 * W), *turn = carve<double>(p, W), *wait = carve<double>(p, W),
           *maxv = carve<double>(p, W), *maxa = carve<double>(p, W), *tan = carve<double>(p, 2 * W),
           *mag = carve<double>(p, 2 * W), *apt = carve<double>(p, Mp), *apw = carve<double>(p, Mp),
           *apv = carve<double>(p, Mp), *apa = carve<double>(p, Mp);
    int *rev = carve<int>(p, W), *stop = carve<int>(p, W), *aps = carve<int>(p, Mp);
    d.sp_start = carve<int>(p, W); d.sp_npts = carve<int>(p, W); d.sp_seg0 = carve<int>(p, W);
    d.sp_tmax = carve<double>(p, W); d.sp_dist0 = carve<double>(p, W); d.sp_param0 = carve<double>(p, W);
    d.seg = carve<double>(p, 12 * (size_t)W); d.seglen = carve<double>(p, W);
    d.info = carve<double>(p, 8);
    double *fitwork = carve<double>(p, 9 * (size_t)W);
    d.lut = carve<double>(p, (size_t)(W - 1) * vap::kLutN);
    d.lut_mag = nullptr;
    d.lut_n = vap::kLutN;
    d.spn = vap::kSamplesPerNode;
    d.wp = wp; d.turn = turn; d.wait = wait; d.maxv = maxv; d.maxa = maxa; d.tangent = tan; d.mag = mag;
    d.ap_t = apt; d.ap_wait = apw; d.ap_maxv = apv; d.ap_maxa = apa; d.rev = rev; d.stop = stop; d.ap_stop = aps;
#define UP(dst, vec) if (!(vec).empty() && hipMemcpyAsync(dst, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice, st) != hipSuccess) goto hip_fail
    UP(wp, h_wp); UP(turn, h_turn); UP(wait, h_wait); UP(maxv, h_maxv); UP(maxa, h_maxa); UP(tan, h_tan); UP(mag, h_mag);
    UP(apt, h_apt); UP(apw, h_apw); UP(apv, h_apv); UP(apa, h_apa); UP(rev, h_rev); UP(stop, h_stop); UP(aps, h_aps);
#undef UP
    {
        hipLaunchKernelGGL(vap::k_route_fit, dim3(1), dim3(1), 0, st, d, fitwork);
        double info[8];
        if (hipMemcpyAsync(info, d.info, sizeof(info), hipMemcpyDeviceToHost, st) != hipSuccess) goto hip_fail;
        if (hipStreamSynchronize(st) != hipSuccess) goto hip_fail;
        if (info[2] != 0.0) {
            vap_route_destroy(rt);
            return vap_fail(VAP_ERR_INVALID, "reverse/turn attribute on the last node: the reference raises IndexError "
                                             "(spline_manager.py:88,97)");
        }
        rt->n_splines = (int)info[0];
        hipLaunchKernelGGL(vap::k_route_lut, dim3(rt->n_splines), dim3(256), 0, st, d);
        hipLaunchKernelGGL(vap::k_route_offsets, dim3(1), dim3(1), 0, st, d);
        if (hipMemcpyAsync(info, d.info, sizeof(info), hipMemcpyDeviceToHost, st) != hipSuccess) goto hip_fail;
        if (hipStreamSynchronize(st) != hipSuccess) goto hip_fail;
        rt->total = info[1];
    }
    *out = rt;
    return VAP_OK;
hip_fail:
    vap_route_destroy(rt);
    return vap_fail(VAP_ERR_HIP, "HIP error while building the route: %s", hipGetErrorString(hipGetLastError()));
}

int vap_route_destroy(vap_route *rt)
{
    if (!rt) return VAP_OK;
    if (rt->ctx) (void)hipSetDevice(rt->ctx->device);
    if (rt->blob) (void)hipFree(rt->blob);
    if (rt->lut_blob) (void)hipFree(rt->lut_blob);
    if (rt->work) (void)hipFree(rt->work);
    delete rt;
    return VAP_OK;
}

int vap_route_set_table_sizes(vap_route *rt, int lut_samples, int samples_per_node)
{
    if (!rt) return vap_fail(VAP_ERR_INVALID, "null route");
    if (lut_samples < 2 || samples_per_node < 1)
        return vap_fail(VAP_ERR_INVALID, "table sizes: min_samples >= 2 (np.linspace + the [1] - [0] step, SM:443-444) and "
                                         "samples_per_node >= 1");
    if ((long long)lut_samples * (rt->W - 1) > (1LL << 27) || (long long)samples_per_node * rt->W > (1LL << 30))
        return vap_fail(VAP_ERR_INVALID, "table sizes out of range");
    VAP_TRY(vap_set_device(rt->ctx));
    hipStream_t st = rt->ctx->stream;
    if (lut_samples == rt->d.lut_n) {
        rt->d.spn = samples_per_node;
        rt->N = 0;   // a cached distance-domain result belongs to the old tables
        return VAP_OK;
    }
    // Build the new arc-length table beside the old one and commit (sizes, pointers, the invalidated cache) only when
    // every step has succeeded: a failed rebuild leaves the route as it was.
    HIP_TRY(hipStreamSynchronize(st));
    void *nb = nullptr;
    const size_t per = (size_t)(rt->W - 1) * lut_samples;
    HIP_TRY(hipMalloc(&nb, sizeof(double) * 2 * per));
    auto trial = rt->d;
    trial.spn = samples_per_node;
    trial.lut = (double *)nb;
    trial.lut_mag = (double *)nb + per;
    trial.lut_n = lut_samples;
    hipLaunchKernelGGL(vap::k_route_lut, dim3(rt->n_splines), dim3(256), 0, st, trial);
    hipLaunchKernelGGL(vap::k_route_offsets, dim3(1), dim3(1), 0, st, trial);
    double info[8];
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(info, trial.info, sizeof(info), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        hipLaunchKernelGGL(vap::k_route_offsets, dim3(1), dim3(1), 0, st, rt->d);   // (the device-side total belongs to the old table again)
        (void)hipStreamSynchronize(st);
        (void)hipFree(nb);
        return vap_fail(VAP_ERR_HIP, "rebuilding the arc-length table failed: %s (the route keeps its tables)", hipGetErrorString(e));
    }
    if (rt->lut_blob) (void)hipFree(rt->lut_blob);
    rt->lut_blob = nb;
    rt->d = trial;
    rt->N = 0;   // a cached distance-domain result belongs to the old tables
    rt->total = info[1];
    return VAP_OK;
}

int vap_route_table_sizes(vap_route *rt, int *lut_samples, int *samples_per_node)
{
    if (!rt) return vap_fail(VAP_ERR_INVALID, "null route");
    if (lut_samples) *lut_samples = rt->d.lut_n;
    if (samples_per_node) *samples_per_node = rt->d.spn;
    return VAP_OK;
}

int vap_route_info(vap_route *rt, int *n_splines, double *total_length)
{
    if (!rt) return vap_fail(VAP_ERR_INVALID, "null route");
    if (n_splines) *n_splines = rt->n_splines;
    if (total_length) *total_length = rt->total;
    return VAP_OK;
}

int vap_route_get_splines(vap_route *rt, int *h_start, int *h_npts, double *h_param_last, double *h_segments,
                          double *h_segment_lengths, double *h_lut_distances, double *h_lut_parameters)
{
    if (!rt) return vap_fail(VAP_ERR_INVALID, "null route");
    VAP_TRY(vap_set_device(rt->ctx));
    hipStream_t st = rt->ctx->stream;
    const int n = rt->n_splines, W = rt->W;
    if (h_start) HIP_TRY(hipMemcpyAsync(h_start, rt->d.sp_start, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    if (h_npts) HIP_TRY(hipMemcpyAsync(h_npts, rt->d.sp_npts, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    if (h_param_last) HIP_TRY(hipMemcpyAsync(h_param_last, rt->d.sp_tmax, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    if (h_segments) HIP_TRY(hipMemcpyAsync(h_segments, rt->d.seg, sizeof(double) * 12 * (W - 1), hipMemcpyDeviceToHost, st));
    if (h_segment_lengths) HIP_TRY(hipMemcpyAsync(h_segment_lengths, rt->d.seglen, sizeof(double) * (W - 1), hipMemcpyDeviceToHost, st));
    if (h_lut_distances || h_lut_parameters) {
        // the concatenated table of SM:456-475 (offsets applied exactly as the lookups apply them)
        const int LN = rt->d.lut_n;
        const size_t ne = (size_t)n * LN;
        std::vector<double> lut(ne), dist0(n), par0(n), tmax(n);
        HIP_TRY(hipMemcpyAsync(lut.data(), rt->d.lut, sizeof(double) * ne, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(dist0.data(), rt->d.sp_dist0, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(par0.data(), rt->d.sp_param0, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(tmax.data(), rt->d.sp_tmax, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int si = 0; si < n; si++)
            for (int j = 0; j < LN; j++) {
                const size_t e = (size_t)si * LN + j;
                if (h_lut_distances) h_lut_distances[e] = lut[e] + dist0[si];
                if (h_lut_parameters) {
                    const double step = tmax[si] / (double)(LN - 1);
                    h_lut_parameters[e] = (j == LN - 1 ? tmax[si] : (double)j * step) + par0[si];
                }
            }
    }
    HIP_TRY(hipStreamSynchronize(st));
    return VAP_OK;
}

static int route_vector_call(vap_route *rt, int kind, int arg, int n, const double *h_in, double *h_out, int out_width)
{
    if (!rt) return vap_fail(VAP_ERR_UNFITTED, "No splines have been initialized");
    VAP_TRY(vap_set_device(rt->ctx));
    if (n < 0 || (n > 0 && (!h_in || !h_out))) return vap_fail(VAP_ERR_INVALID, "bad argument");
    if (n == 0) return VAP_OK;
    hipStream_t st = rt->ctx->stream;
    VAP_TRY(rt->ctx->ensure(rt->ctx->small_in, sizeof(double) * n));
    VAP_TRY(rt->ctx->ensure(rt->ctx->small_out, sizeof(double) * n * out_width));
    double *din = (double *)rt->ctx->small_in.ptr, *dout = (double *)rt->ctx->small_out.ptr;
    HIP_TRY(hipMemcpyAsync(din, h_in, sizeof(double) * n, hipMemcpyHostToDevice, st));
    if (kind == 0) hipLaunchKernelGGL(vap::k_route_eval, dim3((n + 127) / 128), dim3(128), 0, st, rt->d, arg, n, din, dout);
    else hipLaunchKernelGGL(vap::k_route_lookup, dim3((n + 127) / 128), dim3(128), 0, st, rt->d, arg, n, din, dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_out, dout, sizeof(double) * n * out_width, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return VAP_OK;
}

int vap_route_eval(vap_route *rt, int order, int n, const double *h_t, double *h_out)
{
    if (order < 0 || order > 2) return vap_fail(VAP_ERR_INVALID, "order must be 0, 1 or 2");
    return route_vector_call(rt, 0, order, n, h_t, h_out, 2);
}

int vap_route_lookup(vap_route *rt, int what, int n, const double *h_in, double *h_out)
{
    if (what < 0 || what > 2) return vap_fail(VAP_ERR_INVALID, "what must be 0, 1 or 2");
    return route_vector_call(rt, 1, what, n, h_in, h_out, 1);
}

// lays out and fills the distance-domain arrays in rt->work; returns pointers through `a`
struct RouteArrays {
    double *t, *kap, *th, *x, *y, *vinit, *acc_f, *acc_b, *maxacc, *scal, *vel;
    int *bmap;
};

static int route_distance_domain(vap_route *rt, const vap_constraints *c, double dd, double sv, double ev, int &N,
                                 RouteArrays &a, size_t extra_bytes, char **extra)
{
    VAP_TRY(vap_set_device(rt->ctx));
    if (!c || !(dd > 0)) return vap_fail(VAP_ERR_INVALID, "bad constraints or spacing");
    hipStream_t st = rt->ctx->stream;
    // MPG:112-122: sample count and grid of the accumulating loop (current_dist += dd), built on the host
    // in closed form (vap_device.h) and handed to the device as a run table
    std::vector<double> h_runs(vap::kGridRunDoubles, 0.0);
    int n_runs = 1;
    if (!(rt->total > 0) || !std::isfinite(rt->total)) return vap_fail(VAP_ERR_INVALID, "path has no length");
    const long n_loop = vap::build_grid_runs(dd, rt->total, (long)1 << 40, h_runs.data(), n_runs);
    if (n_loop + 1 > (long)1 << 28) return vap_fail(VAP_ERR_UNSUPPORTED, "more than 2^28 samples");
    N = (int)(n_loop + 1);
    const size_t Np = (size_t)N + 8;
    const size_t need = sizeof(double) * (9 * Np + (size_t)(rt->W + rt->M + 4) + 8 + vap::kGridRunDoubles) + sizeof(int) * Np + 64 * 17 +
                        extra_bytes;
    VAP_TRY(ensure_work(rt, need));
    char *p = (char *)rt->work;
    a.t = carve<double>(p, Np); a.kap = carve<double>(p, Np); a.th = carve<double>(p, Np); a.x = carve<double>(p, Np);
    a.y = carve<double>(p, Np); a.vinit = carve<double>(p, Np); a.acc_f = carve<double>(p, Np);
    a.acc_b = carve<double>(p, Np); a.vel = carve<double>(p, Np); a.maxacc = carve<double>(p, rt->W + rt->M + 4);
    a.scal = carve<double>(p, 8); a.bmap = carve<int>(p, Np);
    double *d_runs = carve<double>(p, vap::kGridRunDoubles);
    if (extra) *extra = p;
    HIP_TRY(hipMemcpyAsync(d_runs, h_runs.data(), sizeof(double) * vap::kGridRunDoubles, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // h_runs is a local
    hipLaunchKernelGGL(vap::k_route_grid, dim3((N + 127) / 128), dim3(128), 0, st, rt->d, N, dd, d_runs, n_runs, a.t, a.kap, a.th,
                       a.x, a.y);
    hipLaunchKernelGGL(vap::k_route_caps, dim3(1), dim3(1), 0, st, rt->d, N, a.t, c->max_vel, c->max_acc, c->max_dec, ev,
                       a.vinit, a.acc_f, a.acc_b, a.bmap, a.maxacc, a.scal);
    vap::VelConsts<double> vc;
    vc.vmax = c->max_vel; vc.amax = c->max_acc; vc.adec = c->max_dec; vc.tw = c->track_width;
    vc.wmax = 2 * c->max_vel / c->track_width;
    vc.almax = 2 * c->max_acc / c->track_width;
    hipLaunchKernelGGL(vap::k_route_velocity, dim3(1), dim3(1), 0, st, N, vc, dd, sv, ev, a.kap, a.th, a.vinit, a.acc_f,
                       a.acc_b, a.scal, a.vel);
    HIP_TRY(hipGetLastError());
    rt->N = N;
    rt->dd = dd;
    return VAP_OK;
}

int vap_route_forward_backward(vap_route *rt, const vap_constraints *c, double dd, double start_vel, double end_vel,
                               int capacity, int *n_out, double *h_t, double *h_x, double *h_y, double *h_heading,
                               double *h_curvature, double *h_velocity)
{
    if (!rt) return vap_fail(VAP_ERR_UNFITTED, "No splines have been initialized");
    int N = 0;
    RouteArrays a;
    VAP_TRY(route_distance_domain(rt, c, dd, start_vel, end_vel, N, a, 0, nullptr));
    if (n_out) *n_out = N;
    if (capacity < N) return vap_fail(VAP_ERR_CAPACITY, "need %d samples, capacity %d", N, capacity);
    hipStream_t st = rt->ctx->stream;
    double *src[6] = {a.t, a.x, a.y, a.th, a.kap, a.vel};
    double *dst[6] = {h_t, h_x, h_y, h_heading, h_curvature, h_velocity};
    for (int i = 0; i < 6; i++)
        if (dst[i]) HIP_TRY(hipMemcpyAsync(dst[i], src[i], sizeof(double) * N, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return VAP_OK;
}

int vap_route_sample_count(vap_route *rt, double dd, int *n_out)
{
    if (!rt || !n_out || !(dd > 0)) return vap_fail(VAP_ERR_INVALID, "bad argument");
    if (!(rt->total > 0) || !std::isfinite(rt->total)) return vap_fail(VAP_ERR_INVALID, "path has no length");
    std::vector<double> runs(vap::kGridRunDoubles, 0.0);
    int n_runs = 1;
    const long n_loop = vap::build_grid_runs(dd, rt->total, (long)1 << 40, runs.data(), n_runs);
    *n_out = (int)(n_loop + 1);
    return VAP_OK;
}

int vap_grid_distances(double dd, double total_length, long capacity, double *h_s, long *n_out)
{
    if (!n_out || !(dd > 0) || !(total_length > 0) || !std::isfinite(total_length) || capacity < 0)
        return vap_fail(VAP_ERR_INVALID, "bad argument");
    std::vector<double> runs(vap::kGridRunDoubles, 0.0);
    int n_runs = 1;
    const long n_loop = vap::build_grid_runs(dd, total_length, (long)1 << 40, runs.data(), n_runs);
    *n_out = n_loop;
    if (h_s) {
        int cur = 0;
        for (long k = 0; k < n_loop && k < capacity; k++) h_s[k] = vap::grid_s(runs.data(), n_runs, k, cur);
    }
    return VAP_OK;
}

int vap_route_motion_profile(vap_route *rt, const vap_constraints *c, double dt, double dd, long capacity_rows,
                             double *h_rows, long *n_rows, long *h_nodes_map, int *n_nodes_map, long *h_actions_map,
                             int *n_actions_map)
{
    if (!rt) return vap_fail(VAP_ERR_UNFITTED, "No splines have been initialized");
    if (!h_rows || !n_rows || !h_nodes_map || !n_nodes_map || !h_actions_map || !n_actions_map || capacity_rows < 1 || !(dt > 0))
        return vap_fail(VAP_ERR_INVALID, "bad argument");
    int N = 0;
    RouteArrays a;
    char *extra = nullptr;
    const size_t extra_bytes = sizeof(double) * 8 * (size_t)capacity_rows + sizeof(long) * (size_t)(rt->W + rt->M + 8) + 256;
    VAP_TRY(route_distance_domain(rt, c, dd, 0.01, 0.01, N, a, extra_bytes, &extra));   // MPG:408 default velocities
    hipStream_t st = rt->ctx->stream;
    char *p = extra;
    double *rows = carve<double>(p, 8 * (size_t)capacity_rows);
    long *nmap = carve<long>(p, rt->W + 2), *amap = carve<long>(p, rt->M + 2), *counts = carve<long>(p, 4);
    hipLaunchKernelGGL(vap::k_route_time, dim3(1), dim3(1), 0, st, rt->d, N, a.vel, c->max_vel, c->max_acc, c->max_dec,
                       c->track_width, dt, dd, capacity_rows, rows, nmap, amap, counts);
    HIP_TRY(hipGetLastError());
    long hc[4];
    HIP_TRY(hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (hc[3] == -1) return vap_fail(VAP_ERR_CAPACITY, "motion profile needs more than %ld rows", capacity_rows);
    if (hc[3] == -2) return vap_fail(VAP_ERR_INVALID, "turn / wait before any profile row exists: the reference raises "
                                                      "IndexError (motion_profile_generator.py:440,499)");
    *n_rows = hc[0];
    *n_nodes_map = (int)hc[1];
    *n_actions_map = (int)hc[2];
    HIP_TRY(hipMemcpyAsync(h_rows, rows, sizeof(double) * 8 * (size_t)hc[0], hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_nodes_map, nmap, sizeof(long) * hc[1], hipMemcpyDeviceToHost, st));
    if (hc[2] > 0) HIP_TRY(hipMemcpyAsync(h_actions_map, amap, sizeof(long) * hc[2], hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return VAP_OK;
}

}  // extern "C"
