// vap_routes_batch.hip — batches of routes whose reverse / turn nodes cut them into several splines.
//
// The plain-node kernels of vap_kernels.hip treat a path as ONE spline of W control points.  A route with reverse or
// turn nodes is several splines (SM:57-168) that share their split nodes: segment i still joins nodes i and i+1, so
// the segment / coefficient-block layout [B][W-1] is unchanged; what changes is
//   K1  the derivative estimates at a split node (it is the last point of one spline and the first of the next:
//       end rules instead of the interior average, zero second derivative), the split tangents (SM:84-158) and where
//       QuinticHermiteSpline's setters put them (QHS:543-590, quirk Q3: both in the spline's LAST segment), and
//       the 2-point rule (QHS:170-172, 181-182)                                                     -> k_fit_routes
//   K2  one 1000-entry table per spline, concatenated with running offsets (SM:436-464)           -> k_lut (per
//       (route, spline) workgroup, vap_kernels.hip) + k_route_offsets
//   K3+K4  distance -> parameter over the concatenated table, and parameter -> (segment, local t) with a split node
//       belonging to the earlier spline (SM:243-275)                                                -> k_sample_routes
// The velocity pass is the plain one: forward_backward_pass does nothing special at a reverse or turn node
// (MPG:112-176); those act in the time domain (MPG:435-476, 487-507).
#include <type_traits>
#include "vap_device.h"
#include "vap_kernels.h"

namespace vap {

constexpr uint32_t VAP_FLAG_BAD_ROUTE_BIT = 8u;

// K1 for routes.  One workgroup per route, a thread per node / segment.
// LDS (dynamic, doubles): pts[2W] dist[W] fdL[2W] fdR[2W] sd[2W] ET[2W] ST[2W]  + ints: split[W]
template <typename IT>
__global__ __launch_bounds__(256) void k_fit_routes(int W, int NS, const IT *__restrict__ waypoints, RouteSplitInputs in,
                                                    double *__restrict__ segments, double *__restrict__ power,
                                                    double *__restrict__ seglen, double *__restrict__ sptab,
                                                    int *__restrict__ nspl, double *__restrict__ meta,
                                                    uint32_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int G = W - 1;
    double *pts = sh, *dist = pts + 2 * W, *fdL = dist + W, *fdR = fdL + 2 * W, *sd = fdR + 2 * W, *ET = sd + 2 * W,
           *ST = ET + 2 * W;
    int *split = reinterpret_cast<int *>(ST + 2 * W);
    __shared__ uint32_t s_flag;
    if (tid == 0) s_flag = 0;
    const IT *wp = waypoints + (size_t)b * W * 2;
    for (int i = tid; i < 2 * W; i += nt) pts[i] = (double)wp[i];
    for (int k = tid; k < W; k += nt) {
        const bool f = (in.rev && in.rev[(size_t)b * W + k] != 0) || (in.turn && in.turn[(size_t)b * W + k] != 0.0);
        // node 0 never splits (SM:60 starts at 1); a split at the last node indexes points[W] in the reference
        split[k] = (f && k >= 1 && k <= W - 2) ? 1 : 0;
        if (f && k == W - 1) atomicOr(&s_flag, VAP_FLAG_BAD_ROUTE_BIT);
    }
    __syncthreads();
    for (int i = tid; i < G; i += nt) {
        const double dx = pts[2 * (i + 1)] - pts[2 * i], dy = pts[2 * (i + 1) + 1] - pts[2 * i + 1];
        const double d = sqrt(dx * dx + dy * dy);
        dist[i] = d;
        if (!(d > 0.0) || !isfinite(d)) atomicOr(&s_flag, VAP_FLAG_DEGENERATE_BIT);
    }
    __syncthreads();
    auto tangent_of = [&](int k, double &tx, double &ty) {
        if (!in.tangent) return false;
        tx = in.tangent[((size_t)b * W + k) * 2];
        ty = in.tangent[((size_t)b * W + k) * 2 + 1];
        return !isnan(tx);
    };
    // SM:84-158 split tangents: the ending tangent of the spline that ends at node i, the starting tangent of the next
    for (int i = tid; i < W; i += nt) {
        double e0 = NAN, e1 = NAN, s0 = NAN, s1 = NAN;
        if (split[i]) {
            const double *pm = pts + 2 * (i - 1), *pi = pts + 2 * i, *pn = pts + 2 * (i + 1);
            const double prev_len = dist[i - 1], next_len = dist[i];
            const double ps = prev_len > 0 ? 1.0 / prev_len : 1.0, ns = next_len > 0 ? 1.0 / next_len : 1.0;
            double pv[2] = {(pi[0] - pm[0]) * ps, (pi[1] - pm[1]) * ps};
            const double nv[2] = {(pn[0] - pi[0]) * ns, (pn[1] - pi[1]) * ns};
            const double min_len = prev_len < next_len ? prev_len : next_len;
            double tg0 = 0, tg1 = 0;
            const bool has_tan = tangent_of(i, tg0, tg1);
            const double im = has_tan ? in.mag[((size_t)b * W + i) * 2] : 0.0, om = has_tan ? in.mag[((size_t)b * W + i) * 2 + 1] : 0.0;
            const double turn = in.turn ? in.turn[(size_t)b * W + i] : 0.0;
            const bool rev = in.rev && in.rev[(size_t)b * W + i] != 0;
            if (turn != 0) {   // SM:103-132
                double ang = turn * (M_PI / 180.0);
                if (rev) ang = ang + M_PI;
                const double c = cos(ang), s = sin(ang);
                double nt0 = c * pv[0] + (-s) * pv[1], nt1 = s * pv[0] + c * pv[1];
                nt0 *= min_len; nt1 *= min_len;
                pv[0] *= min_len; pv[1] *= min_len;
                if (has_tan) {
                    pv[0] = tg0 * im; pv[1] = tg1 * im;
                    nt0 = (tg0 * c + tg1 * s) * -1;
                    nt1 = (tg0 * (-s) + tg1 * c) * -1;
                    nt0 *= om; nt1 *= om;
                }
                e0 = pv[0]; e1 = pv[1];
                s0 = nt0; s1 = nt1;
            } else {           // reverse node, SM:134-158
                double dv0 = pv[0] - nv[0], dv1 = pv[1] - nv[1];
                const double dn = sqrt(dv0 * dv0 + dv1 * dv1);
                if (dn > 0) { dv0 /= dn; dv1 /= dn; }
                dv0 *= min_len; dv1 *= min_len;
                if (has_tan) { dv0 = tg0 * im; dv1 = tg1 * im; }
                e0 = dv0; e1 = dv1;
                s0 = -1 * dv0; s1 = -1 * dv1;
                if (has_tan) { s0 = -1 * tg0 * om; s1 = -1 * tg1 * om; }
            }
        }
        ET[2 * i] = e0; ET[2 * i + 1] = e1;
        ST[2 * i] = s0; ST[2 * i + 1] = s1;
    }
    // the spline table: {parameters[-1] (QHS:719-736, only [-1] is ever read), 0, 0, first node}; offsets later
    if (tid == 0) {
        int n = 0, first = 0;
        for (int k = 1; k <= W - 1; k++) {
            if (!(split[k] || k == W - 1)) continue;
            if (n < NS) {
                double cum = 0.0;
                for (int i = first; i < k; i++) cum += dist[i];
                const int Gs = k - first;
                double *sp = sptab + ((size_t)b * NS + n) * kSplineStride;
                sp[0] = (cum == 0.0) ? (double)Gs : cum * (double)Gs / cum;
                sp[1] = 0.0;
                sp[2] = 0.0;
                sp[3] = (double)first;
            } else {
                atomicOr(&s_flag, VAP_FLAG_BAD_ROUTE_BIT);     // more splines than the caller allowed for
            }
            n++;
            first = k;
        }
        nspl[b] = n < NS ? n : NS;
    }
    __syncthreads();
    auto is_start = [&](int k) { return k == 0 || split[k]; };
    auto is_end = [&](int k) { return k == W - 1 || split[k]; };
    // QHS:163-195 first derivatives of node k as the spline on its right (fdR, used by segment k) and the spline on
    // its left (fdL, used by segment k-1) see it
    for (int k = tid; k < W; k += nt) {
        double ax = 0, ay = 0, bx = 0, by = 0;
        if (k >= 1 && k <= W - 2) {
            const double px = (pts[2 * k] - pts[2 * (k - 1)]) / dist[k - 1], py = (pts[2 * k + 1] - pts[2 * (k - 1) + 1]) / dist[k - 1];
            const double nx = (pts[2 * (k + 1)] - pts[2 * k]) / dist[k], ny = (pts[2 * (k + 1) + 1] - pts[2 * k + 1]) / dist[k];
            ax = bx = (px + nx) / 2;
            ay = by = (py + ny) / 2;
        }
        if (k <= W - 2 && is_start(k)) {
            // first point of a spline; a 2-point spline with an ending tangent keeps the chord un-normalised (QHS:170-172)
            const bool two = is_end(k + 1);
            const double d = (two && split[k + 1]) ? 1.0 : dist[k];
            bx = (pts[2 * (k + 1)] - pts[2 * k]) / d;
            by = (pts[2 * (k + 1) + 1] - pts[2 * k + 1]) / d;
        }
        if (k >= 1 && is_end(k)) {
            const bool two = is_start(k - 1);
            const double d = (two && split[k - 1]) ? 1.0 : dist[k - 1];   // QHS:181-182 (starting tangent set)
            ax = (pts[2 * k] - pts[2 * (k - 1)]) / d;
            ay = (pts[2 * k + 1] - pts[2 * (k - 1) + 1]) / d;
        }
        fdL[2 * k] = ax; fdL[2 * k + 1] = ay;
        fdR[2 * k] = bx; fdR[2 * k + 1] = by;
    }
    __syncthreads();
    // QHS:197-219 second derivatives: zero at both ends of every spline
    for (int k = tid; k < W; k += nt) {
        double sx = 0.0, sy = 0.0;
        if (k > 0 && k < W - 1 && !split[k]) {
            const double avg = (dist[k - 1] + dist[k]) / 2;
            sx = (fdL[2 * (k + 1)] - fdR[2 * (k - 1)]) / (avg * 0.5);
            sy = (fdL[2 * (k + 1) + 1] - fdR[2 * (k - 1) + 1]) / (avg * 0.5);
        }
        sd[2 * k] = sx;
        sd[2 * k + 1] = sy;
    }
    __syncthreads();
    // QHS:76-132 segment assembly, then the tangent setters
    for (int i = tid; i < G; i += nt) {
        const double L = dist[i];
        double r[12];
        r[0] = pts[2 * i];       r[1] = pts[2 * i + 1];
        r[2] = pts[2 * (i + 1)]; r[3] = pts[2 * (i + 1) + 1];
        if (L > 0) {
            const double L2 = L * L;
            r[4] = fdR[2 * i] * L;           r[5] = fdR[2 * i + 1] * L;
            r[6] = fdL[2 * (i + 1)] * L;     r[7] = fdL[2 * (i + 1) + 1] * L;
            r[8] = sd[2 * i] * L2;           r[9] = sd[2 * i + 1] * L2;
            r[10] = sd[2 * (i + 1)] * L2;    r[11] = sd[2 * (i + 1) + 1] * L2;
            double tx, ty;
            if (tangent_of(i, tx, ty)) {           // SM:65-77: [tangent*incoming, tangent*outgoing]; QHS:102-115
                const double om = in.mag[((size_t)b * W + i) * 2 + 1];
                r[4] = tx * om; r[5] = ty * om;
            }
            if (tangent_of(i + 1, tx, ty)) {
                const double im = in.mag[((size_t)b * W + i + 1) * 2];
                r[6] = tx * im; r[7] = ty * im;
            }
        } else {
            r[4] = fdR[2 * i];           r[5] = fdR[2 * i + 1];
            r[6] = fdL[2 * (i + 1)];     r[7] = fdL[2 * (i + 1) + 1];
            r[8] = sd[2 * i];            r[9] = sd[2 * i + 1];
            r[10] = sd[2 * (i + 1)];     r[11] = sd[2 * (i + 1) + 1];
        }
        if (is_end(i + 1)) {   // the spline's last segment takes both split tangents (QHS:561, 586: quirk Q3)
            int f = i;
            while (!is_start(f)) f--;
            if (f >= 1) { r[4] = ST[2 * f]; r[5] = ST[2 * f + 1]; }          // (is_start(f) && f >= 1  <=>  split[f])
            if (split[i + 1]) { r[6] = ET[2 * (i + 1)]; r[7] = ET[2 * (i + 1) + 1]; }
        }
        double *sg = segments + ((size_t)b * G + i) * 12;
#pragma unroll
        for (int k = 0; k < 12; k++) sg[k] = r[k];
        if (power) make_coef_block(r, power + ((size_t)b * G + i) * kCoefDoubles);
        if (seglen) seglen[(size_t)b * G + i] = L;
    }
    __syncthreads();
    if (tid == 0) {
        if (flags) flags[b] = s_flag;
        meta[(size_t)b * kMetaStride + 0] = 0.0;   // (filled by k_route_offsets)
    }
}

// SM:456-464 + 466-475: distance / parameter offsets of the concatenated table, total length; then the route's
// distance grid (what k_lut's tail does for plain paths).  One thread per route.
__global__ void k_route_offsets(int B, int W, int NS, int S, double dd, const double *__restrict__ lut, double *__restrict__ sptab,
                                const int *__restrict__ nspl, double *__restrict__ meta, double *__restrict__ aux,
                                double *__restrict__ runs, uint32_t *__restrict__ flags)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n = nspl[b];
    double current_dist = 0.0, prev_param = 0.0;
    for (int s = 0; s < n; s++) {
        double *sp = sptab + ((size_t)b * NS + s) * kSplineStride;
        sp[1] = current_dist;
        sp[2] = prev_param;
        current_dist = lut[((size_t)b * NS + s) * kLutN + kLutN - 1] + current_dist;   // spline_distances[-1]
        prev_param += sp[0] - 0.0;
    }
    meta[(size_t)b * kMetaStride + 0] = prev_param;     // lookup_table.parameters[-1]
    meta[(size_t)b * kMetaStride + 1] = current_dist;   // total_length
    if (flags && !(current_dist > 0.0 && isfinite(current_dist))) atomicOr(&flags[b], VAP_FLAG_DEGENERATE_BIT);
    grid_define_route(b, W, S, dd, current_dist, meta, aux, runs, flags);
}

// K3+K4 for routes: four consecutive samples per thread, 1024 per workgroup.  The concatenated table stays in HBM / L2
// (up to W-1 splines x 8 KB per route: too much to stage per workgroup), but a thread searches it once — the spline by
// the distance offsets, then that spline's 1000 entries — and walks on from there for its other three samples; the
// spline table and the linspace step of every spline sit in LDS.  Same arithmetic as the general functions of
// vap_device.h (lutv_distance_to_time, table_index, lutv_map_parameter): np.searchsorted-left over the concatenated
// distances, the reference's lerp, the exact step lookup wherever the fast index is within rounding of a decision.
constexpr int kRouteSPT = 4, kRouteThreads = 256, kRouteChunk = kRouteSPT * kRouteThreads;
template <typename OT, bool HI>
__global__ __launch_bounds__(kRouteThreads) void k_sample_routes(int W, int NS, int S, const double *__restrict__ power,
                                                                 const double *__restrict__ lut, const double *__restrict__ sptab,
                                                                 const int *__restrict__ nspl, const double *__restrict__ meta,
                                                                 const double *__restrict__ aux, const double *__restrict__ runs,
                                                                 OT *__restrict__ ox, OT *__restrict__ oy, OT *__restrict__ oh,
                                                                 OT *__restrict__ ok, OT *__restrict__ odth, double *__restrict__ ok64,
                                                                 double *__restrict__ odth64)
{
    extern __shared__ __attribute__((aligned(16))) double s_sp[];   // NS * (kSplineStride + 1): the spline table + linspace steps
    __shared__ double s_dx[kRouteThreads + 1], s_dy[kRouteThreads + 1];
    __shared__ int s_j[kRouteThreads + 1];
    __shared__ OT s_th[kRouteThreads + 1];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int G = W - 1;
    const double *m = meta + (size_t)b * kMetaStride;
    const double total = m[1];
    const int N = (int)m[3];
    const double *ax = aux + (size_t)b * kAuxStride;
    const double inv_tstep = ax[2];
    const int n_runs = (int)ax[3];
    const double *tab = runs + (size_t)b * kGridRunDoubles;
    const double dd = m[2];
    const int n_spl = nspl[b];
    double *s_step = s_sp + NS * kSplineStride;
    for (int i = tid; i < n_spl * kSplineStride; i += kRouteThreads) s_sp[i] = sptab[(size_t)b * NS * kSplineStride + i];
    for (int i = tid; i < n_spl; i += kRouteThreads)
        s_step[i] = sptab[((size_t)b * NS + i) * kSplineStride + 0] / (double)(kLutN - 1);   // np.linspace's step (SM:443)
    __syncthreads();
    LutView v;
    v.D = lut + (size_t)b * NS * kLutN;
    v.sp = s_sp;
    v.n_spl = n_spl;
    v.total = total;
    v.end_param = (double)(W - 1);
    const double *pw = power + (size_t)b * G * kCoefDoubles;
    const int tab_n = W * kSamplesPerNode;
    const size_t row = (size_t)b * S;
    const int k0 = blockIdx.x * kRouteChunk;
    struct Eval { double ex, ey, kap; OT th, x, y; int jj; };
    // position in the concatenated table: spline si, entry j (1 <= j <= 999) with dist[si][j-1] < s <= dist[si][j] in the
    // searchsorted-left sense over the whole table
    int si = 0, j = 1;
    bool located = false;
    auto dist_at = [&](int sp_i, int e) { return v.D[(size_t)sp_i * kLutN + e] + s_sp[sp_i * kSplineStride + 1]; };
    auto locate = [&](double s) {
        // the first spline whose last entry reaches s, then np.searchsorted-left inside it
        si = 0;
        while (si < n_spl - 1 && dist_at(si, kLutN - 1) < s) si++;
        int lo = 0, hi = kLutN - 1;     // (entry 999 >= s by the choice of the spline, or it is the last spline and s < total)
#pragma unroll 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (dist_at(si, mid) < s) lo = mid + 1;
            else hi = mid;
        }
        j = lo;
        located = true;
    };
    auto eval = [&](int k) {
        Eval e;
        const int kk = k < N - 1 ? k : N - 1;
        int r = grid_run_hint(dd, kk, n_runs);
        const double s = (kk == N - 1) ? total : grid_s(tab, n_runs, kk, r);     // MPG:112-122, 172-175
        double t;
        if (s <= 0) {
            t = 0.0;                                                             // SM:291-318 early returns
        } else if (s >= total) {
            t = v.end_param;
        } else {
            if (!located) {
                locate(s);
            } else {   // samples come in increasing distance: walk on (entries, then splines)
                while (dist_at(si, j) < s) {
                    if (j < kLutN - 1) j++;
                    else { si++; j = 0; }
                }
            }
            // global entry e = si*1000 + j is the first with distances[e] >= s.  e == 0 cannot be (s > 0 = distances[0]);
            // j == 0 of a later spline: its predecessor is the previous spline's last entry (SM:457-462 concatenation)
            const int ps = j > 0 ? si : si - 1, pj = j > 0 ? j - 1 : kLutN - 1;
            const double d0 = dist_at(ps, pj), d1 = dist_at(si, j);
            const double t0 = (pj == kLutN - 1 ? s_sp[ps * kSplineStride + 0] : (double)pj * s_step[ps]) + s_sp[ps * kSplineStride + 2];
            const double t1 = (j == kLutN - 1 ? s_sp[si * kSplineStride + 0] : (double)j * s_step[si]) + s_sp[si * kSplineStride + 2];
            t = t0 + (t1 - t0) * (s - d0) / (d1 - d0);
        }
        bool near;
        int jj = table_index_fast(t, tab_n, inv_tstep, near);                    // SM:340-346 / 550-580
        if (near) jj = table_index(t, tab_n, v.end_param);
        const double tp = linspace_at(v.end_param, tab_n, jj);
        int sg;
        double lt;
        lutv_map_parameter(v, W, tp, sg, lt);
        const double *c = pw + (size_t)sg * kCoefDoubles;
        e.ex = horner4(c + kCoefD1, lt); e.ey = horner4(c + kCoefD1 + 5, lt);
        const double fx = horner3(c + kCoefD2, lt), fy = horner3(c + kCoefD2 + 4, lt);
        const double ss = fma(e.ex, e.ex, e.ey * e.ey);
        const double num = fma(e.ex, fy, -(e.ey * fx));
        e.kap = (ss >= 1e-10) ? curvature_of_r(num, ss) : 0.0;
        e.th = heading_of_r<OT>(e.ey, e.ex);
        e.jj = jj;
        lutv_map_parameter(v, W, t, sg, lt);                                      // SM:204-215 at the sample's own parameter
        c = pw + (size_t)sg * kCoefDoubles;
        if constexpr (sizeof(OT) == 4) {
            const float *cf = reinterpret_cast<const float *>(c + kCoefPf);
            e.x = horner5f(cf, (float)lt);
            e.y = horner5f(cf + 6, (float)lt);
        } else {
            e.x = horner5(c + kCoefP, lt);
            e.y = horner5(c + kCoefP + 6, lt);
        }
        return e;
    };
    const int kb = k0 + tid * kRouteSPT;
    Eval ev[kRouteSPT];
#pragma unroll
    for (int i = 0; i < kRouteSPT; i++) {
        ev[i] = Eval{};
        if (kb + i < S && kb + i < N) ev[i] = eval(kb + i);
    }
    s_dx[tid] = ev[0].ex; s_dy[tid] = ev[0].ey; s_j[tid] = ev[0].jj; s_th[tid] = ev[0].th;
    if (tid == kRouteThreads - 1) {
        Eval nx{};
        if (kb + kRouteSPT < N) nx = eval(kb + kRouteSPT);
        s_dx[kRouteThreads] = nx.ex; s_dy[kRouteThreads] = nx.ey; s_j[kRouteThreads] = nx.jj; s_th[kRouteThreads] = nx.th;
    }
    __syncthreads();
    // a thread's four consecutive samples leave as one 16-byte piece per fp32 row (two per fp64 row) where the row allows
    OT vx[kRouteSPT], vy[kRouteSPT], vh[kRouteSPT], vk[kRouteSPT], vd[kRouteSPT];
    double vk64[kRouteSPT], vd64[kRouteSPT];
#pragma unroll
    for (int i = 0; i < kRouteSPT; i++) {
        const int k = kb + i;
        const Eval &me = ev[i];
        OT dth = (OT)0;
        double dth64 = 0.0;
        if (k < S && k < N - 1) {
            const double nx = i + 1 < kRouteSPT ? ev[(i + 1) % kRouteSPT].ex : s_dx[tid + 1];
            const double ny = i + 1 < kRouteSPT ? ev[(i + 1) % kRouteSPT].ey : s_dy[tid + 1];
            const OT nth = i + 1 < kRouteSPT ? ev[(i + 1) % kRouteSPT].th : s_th[tid + 1];
            const int nj = i + 1 < kRouteSPT ? ev[(i + 1) % kRouteSPT].jj : s_j[tid + 1];
            if constexpr (sizeof(OT) == 8) {
                dth = fabs(nth - me.th);
            } else if constexpr (HI) {
                if (nj != me.jj) dth64 = dtheta_f64(me.ex, me.ey, nx, ny, me.th, nth);
                dth = (OT)dth64;
            } else {
                if (nj != me.jj) dth = dtheta_f32(me.ex, me.ey, nx, ny, me.th, nth);
            }
        }
        const bool in = k < N;
        vx[i] = in ? me.x : (OT)0;
        vy[i] = in ? me.y : (OT)0;
        vh[i] = in ? me.th : (OT)0;
        vk[i] = in ? (OT)me.kap : (OT)0;
        vd[i] = in ? dth : (OT)0;
        vk64[i] = in ? me.kap : 0.0;
        vd64[i] = in ? dth64 : 0.0;
    }
    if (kb < S) {
        const size_t o = row + kb;
        const bool whole = kb + kRouteSPT <= S && (o & 3) == 0;
        auto put = [&](auto *__restrict__ dst, const auto (&x)[kRouteSPT]) {
            using T = std::remove_reference_t<decltype(x[0])>;
            if (whole) {
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4 *>(dst + o) = make_float4(x[0], x[1], x[2], x[3]);
                } else {
                    *reinterpret_cast<double2 *>(dst + o) = make_double2(x[0], x[1]);
                    *reinterpret_cast<double2 *>(dst + o + 2) = make_double2(x[2], x[3]);
                }
            } else {
                for (int i = 0; i < kRouteSPT && kb + i < S; i++) dst[o + i] = x[i];
            }
        };
        if (ox) put(ox, vx);
        if (oy) put(oy, vy);
        if (oh) put(oh, vh);
        if (ok) put(ok, vk);
        if (odth) put(odth, vd);
        if constexpr (HI) {
            put(ok64, vk64);
            put(odth64, vd64);
        }
    }
}

hipError_t launch_fit_routes(hipStream_t st, bool f64, int B, int W, int NS, const void *wp, const RouteSplitInputs &in, double *seg,
                             double *pw, double *seglen, double *sptab, int *nspl, double *meta, uint32_t *flags)
{
    const size_t lds = sizeof(double) * (size_t)(13 * W) + sizeof(int) * (size_t)W + 16;
    const dim3 block(W <= 64 ? 64 : 256);
    if (f64)
        hipLaunchKernelGGL(k_fit_routes<double>, dim3(B), block, lds, st, W, NS, (const double *)wp, in, seg, pw, seglen, sptab, nspl, meta, flags);
    else
        hipLaunchKernelGGL(k_fit_routes<float>, dim3(B), block, lds, st, W, NS, (const float *)wp, in, seg, pw, seglen, sptab, nspl, meta, flags);
    return hipGetLastError();
}

hipError_t launch_route_offsets(hipStream_t st, int B, int W, int NS, int S, double dd, const double *lut, double *sptab,
                                const int *nspl, double *meta, double *aux, double *runs, uint32_t *flags)
{
    hipLaunchKernelGGL(k_route_offsets, dim3((B + 63) / 64), dim3(64), 0, st, B, W, NS, S, dd, lut, sptab, nspl, meta, aux, runs, flags);
    return hipGetLastError();
}

hipError_t launch_sample_routes(hipStream_t st, bool f64, int B, int W, int NS, int S, const double *pw, const double *lut,
                                const double *sptab, const int *nspl, const double *meta, const double *aux, const double *runs,
                                void *x, void *y, void *h, void *k, void *dth, double *k64, double *dth64)
{
    const dim3 grid((S + kRouteChunk - 1) / kRouteChunk, B);
    const bool hi = !f64 && k64 && dth64;
    const size_t lds = sizeof(double) * (size_t)NS * (kSplineStride + 1);
#define VAP_SR(OT_, HI_)                                                                                                     \
    hipLaunchKernelGGL((k_sample_routes<OT_, HI_>), grid, dim3(kRouteThreads), lds, st, W, NS, S, pw, lut, sptab, nspl, meta, aux, runs, \
                       (OT_ *)x, (OT_ *)y, (OT_ *)h, (OT_ *)k, (OT_ *)dth, k64, dth64)
    if (f64) VAP_SR(double, false);
    else if (hi) VAP_SR(float, true);
    else VAP_SR(float, false);
#undef VAP_SR
    return hipGetLastError();
}

}  // namespace vap
