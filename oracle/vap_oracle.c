/* TEST INFRASTRUCTURE ONLY — see vap_oracle.h.
 *
 * Scalar fp64 restatement of the reference hot path, statement by statement, in the reference's own
 * operation order (build with -ffp-contract=off so no FMA is formed where NumPy/CPython round twice).
 * Citations are file:line under /root/reference/src.  QHS = splines/quintic_hermite_spline.py,
 * SM = splines/spline_manager.py, MPG = motion_profiling_v2/motion_profile_generator.py,
 * ODM = motion_profiling_v2/one_dim_mp_generator.py.
 *
 * Parity status: pinned by tests/test_oracle_golden.py against the .npz fixtures in tests/golden/, which were
 * produced by running the real reference in the build container (oracle/gen_golden.py).
 */
#include "vap_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define LUT_MIN_SAMPLES 1000   /* SM:427 */
#define SAMPLES_PER_NODE 1000  /* SM:477 */

typedef struct {
    int start;      /* index of first control point (global node index) */
    int npts;       /* control points in this spline */
    int seg0;       /* index of first segment in path->seg */
    double t_max;   /* parameters[-1], QHS:719-736 */
} spline_t;

struct vapo_path {
    int W;
    int n_splines;
    spline_t *sp;
    double *seg;     /* (W-1) x 6 x 2 */
    double *seglen;  /* W-1 */
    /* node / action attributes the profile reads (MPG:100-163, 428-553) */
    int *rev, *stop;
    double *turn, *wait, *maxv, *maxa;
    int M;
    double *ap_t, *ap_wait, *ap_maxv, *ap_maxa;
    int *ap_stop;
    /* SM:15-21 PathLookupTable */
    int lut_n;
    double *lut_d, *lut_p;
    double total;
    int have_lut;
    /* SM:544-548 _precomputed_properties */
    int tab_n;
    double *tab_p, *tab_k, *tab_h;
    int have_tab;
    /* build_lookup_table(min_samples) / precompute_path_properties(samples_per_node): 0 = the defaults (SM:427, 477) */
    int min_samples, samples_per_node;
};

static double norm2(double dx, double dy) { return sqrt(dx * dx + dy * dy); }

/* np.linspace(0, stop, num)[j] for float stop (numpy/_core/function_base.py): step = stop/(num-1);
 * y = arange(num)*step + 0; y[-1] = stop. */
static double linspace_at(double stop, int num, int j)
{
    if (num == 1) return 0.0;
    if (j == num - 1) return stop;
    double step = stop / (double)(num - 1);
    if (step == 0.0) return ((double)j / (double)(num - 1)) * stop;
    return (double)j * step;
}

/* ---- QHS:288-416 basis polynomials, written exactly as the reference writes them ---- */
static void basis0(double t, double H[6])
{
    double t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
    H[0] = 1 - 10 * t3 + 15 * t4 - 6 * t5;
    H[1] = 10 * t3 - 15 * t4 + 6 * t5;
    H[2] = t - 6 * t3 + 8 * t4 - 3 * t5;
    H[3] = -4 * t3 + 7 * t4 - 3 * t5;
    H[4] = 0.5 * t2 - 1.5 * t3 + 1.5 * t4 - 0.5 * t5;
    H[5] = 0.5 * t3 - t4 + 0.5 * t5;
}
static void basis1(double t, double H[6])
{
    double t2 = t * t, t3 = t2 * t, t4 = t3 * t;
    H[0] = -30 * t2 + 60 * t3 - 30 * t4;
    H[1] = 30 * t2 - 60 * t3 + 30 * t4;
    H[2] = 1 - 18 * t2 + 32 * t3 - 15 * t4;
    H[3] = -12 * t2 + 28 * t3 - 15 * t4;
    H[4] = t - 4.5 * t2 + 6 * t3 - 2.5 * t4;
    H[5] = 1.5 * t2 - 4 * t3 + 2.5 * t4;
}
static void basis2(double t, double H[6])
{
    double t2 = t * t, t3 = t2 * t;
    H[0] = -60 * t + 180 * t2 - 120 * t3;
    H[1] = 60 * t - 180 * t2 + 120 * t3;
    H[2] = -36 * t + 96 * t2 - 60 * t3;
    H[3] = -24 * t + 84 * t2 - 60 * t3;
    H[4] = 1 - 9 * t + 18 * t2 - 10 * t3;
    H[5] = 3 * t - 12 * t2 + 10 * t3;
}

/* QHS:418-469 (in the class, never called by the reference) */
static void basis3(double t, double H[6])
{
    double t2 = t * t;
    H[0] = -60 + 360 * t - 360 * t2;
    H[1] = 60 - 360 * t + 360 * t2;
    H[2] = -36 + 192 * t - 180 * t2;
    H[3] = -24 + 168 * t - 180 * t2;
    H[4] = -9 + 36 * t - 30 * t2;
    H[5] = 3 - 24 * t + 30 * t2;
}
void vapo_basis(int order, double t, double H[6])
{
    if (order == 0) basis0(t, H);
    else if (order == 1) basis1(t, H);
    else if (order == 2) basis2(t, H);
    else basis3(t, H);
}

/* QHS:506-541 _normalize_parameter */
static void normalize_parameter(const spline_t *s, double t, double *local_t, int *idx)
{
    double t_min = 0.0, t_max = s->t_max;
    double tt = t < t_max ? t : t_max; /* min(t, t_max) */
    tt = t_min > tt ? t_min : tt;      /* max(t_min, ...) */
    int nseg = s->npts - 1;
    int i = (int)((tt - t_min) / 1.0);
    if (i == nseg) i = nseg - 1;
    double seg_start = t_min + i * 1.0;
    *local_t = (tt - seg_start) / 1.0;
    *idx = i;
}

/* QHS:221-251 / 473-504: out = sum_i basis_i * segment[idx][i], accumulated from zeros */
static void spline_eval(const vapo_path *p, const spline_t *s, double t, int order, double out[2])
{
    double lt, H[6];
    int idx;
    normalize_parameter(s, t, &lt, &idx);
    if (order == 0) basis0(lt, H);
    else if (order == 1) basis1(lt, H);
    else basis2(lt, H);
    const double *sg = p->seg + (size_t)(s->seg0 + idx) * 12;
    double ax = 0.0, ay = 0.0;
    for (int i = 0; i < 6; i++) {
        ax += H[i] * sg[2 * i];
        ay += H[i] * sg[2 * i + 1];
    }
    out[0] = ax;
    out[1] = ay;
}

/* SM:243-275 _map_parameter_to_spline */
static const spline_t *map_parameter(const vapo_path *p, double t, double *local_t)
{
    int cumulative = 0;
    for (int i = 0; i < p->n_splines; i++) {
        const spline_t *s = &p->sp[i];
        int seg_start = cumulative;
        int seg_end = cumulative + s->npts - 1;
        if (t <= (double)seg_end || i == p->n_splines - 1) {
            *local_t = t - (double)seg_start;
            return s;
        }
        cumulative += s->npts - 1;
    }
    return NULL;
}

void vapo_point(const vapo_path *p, double t, double out[2])
{
    double lt;
    const spline_t *s = map_parameter(p, t, &lt);
    spline_eval(p, s, lt, 0, out);
}
void vapo_derivative(const vapo_path *p, double t, double out[2])
{
    double lt;
    const spline_t *s = map_parameter(p, t, &lt);
    spline_eval(p, s, lt, 1, out);
}
void vapo_second_derivative(const vapo_path *p, double t, double out[2])
{
    double lt;
    const spline_t *s = map_parameter(p, t, &lt);
    spline_eval(p, s, lt, 2, out);
}

/* One spline: QHS:30-138 fit + QHS:149-219 _compute_derivatives + QHS:543-590 tangent setters.
 * pts: npts x 2.  tan_in/tan_out: per control point override rows (NaN = None) = set_tangents.
 * start_tan / end_tan: NULL or 2-vector (starting_tangent / ending_tangent set before fit). */
static int fit_spline(int npts, const double *pts, const double *tan_in, const double *tan_out,
                      const double *start_tan, const double *end_tan, double *seg, double *seglen,
                      double *t_max)
{
    if (npts < 2) return -1;
    int G = npts - 1;
    double *dist = (double *)malloc(sizeof(double) * G);
    double *fd = (double *)calloc((size_t)npts * 2, sizeof(double));
    double *sd = (double *)calloc((size_t)npts * 2, sizeof(double));
    /* QHS:719-736 _compute_parameters: only parameters[-1] is ever read */
    double cum = 0.0;
    for (int i = 0; i < G; i++) {
        double dx = pts[2 * (i + 1)] - pts[2 * i], dy = pts[2 * (i + 1) + 1] - pts[2 * i + 1];
        dist[i] = norm2(dx, dy); /* np.linalg.norm(diffs, axis=1) */
        cum += dist[i];          /* np.cumsum */
    }
    if (cum == 0.0) *t_max = (double)(npts - 1);
    else *t_max = cum * (double)(npts - 1) / cum;

    /* QHS:163-195 first derivatives */
    for (int i = 0; i < npts; i++) {
        if (i == 0) {
            double cx = pts[2] - pts[0], cy = pts[3] - pts[1];
            if (npts == 2 && end_tan) { fd[0] = cx * 1; fd[1] = cy * 1; }
            else { fd[0] = cx * 1 / dist[0]; fd[1] = cy * 1 / dist[0]; }
        } else if (i == npts - 1) {
            double cx = pts[2 * i] - pts[2 * (i - 1)], cy = pts[2 * i + 1] - pts[2 * (i - 1) + 1];
            if (npts == 2 && start_tan) { fd[2 * i] = cx * 1; fd[2 * i + 1] = cy * 1; }
            else { fd[2 * i] = cx * 1 / dist[G - 1]; fd[2 * i + 1] = cy * 1 / dist[G - 1]; }
        } else {
            double px = (pts[2 * i] - pts[2 * (i - 1)]) / dist[i - 1];
            double py = (pts[2 * i + 1] - pts[2 * (i - 1) + 1]) / dist[i - 1];
            double nx = (pts[2 * (i + 1)] - pts[2 * i]) / dist[i];
            double ny = (pts[2 * (i + 1) + 1] - pts[2 * i + 1]) / dist[i];
            fd[2 * i] = (px + nx) * 1 / 2;
            fd[2 * i + 1] = (py + ny) * 1 / 2;
        }
    }
    /* QHS:197-219 second derivatives */
    for (int i = 1; i < npts - 1; i++) {
        double avg = (dist[i - 1] + dist[i]) / 2;
        sd[2 * i] = (fd[2 * (i + 1)] - fd[2 * (i - 1)]) / (avg * 0.5);
        sd[2 * i + 1] = (fd[2 * (i + 1) + 1] - fd[2 * (i - 1) + 1]) / (avg * 0.5);
    }
    /* QHS:76-127 segment assembly */
    for (int i = 0; i < G; i++) {
        const double *p0 = pts + 2 * i, *p1 = pts + 2 * (i + 1);
        double dx = p1[0] - p0[0], dy = p1[1] - p0[1];
        /* np.linalg.norm of a 1-D vector = sqrt(dot(x,x)) */
        double L = sqrt(dx * dx + dy * dy);
        seglen[i] = L;
        double *s = seg + (size_t)i * 12;
        s[0] = p0[0]; s[1] = p0[1];
        s[2] = p1[0]; s[3] = p1[1];
        if (L > 0) {
            double L2 = L * L;
            s[4] = fd[2 * i] * L;           s[5] = fd[2 * i + 1] * L;
            s[6] = fd[2 * (i + 1)] * L;     s[7] = fd[2 * (i + 1) + 1] * L;
            s[8] = sd[2 * i] * L2;          s[9] = sd[2 * i + 1] * L2;
            s[10] = sd[2 * (i + 1)] * L2;   s[11] = sd[2 * (i + 1) + 1] * L2;
            if (tan_out && !isnan(tan_out[2 * i])) { s[4] = tan_out[2 * i]; s[5] = tan_out[2 * i + 1]; }
            if (tan_in && !isnan(tan_in[2 * (i + 1)])) { s[6] = tan_in[2 * (i + 1)]; s[7] = tan_in[2 * (i + 1) + 1]; }
        } else {
            s[4] = fd[2 * i];           s[5] = fd[2 * i + 1];
            s[6] = fd[2 * (i + 1)];     s[7] = fd[2 * (i + 1) + 1];
            s[8] = sd[2 * i];           s[9] = sd[2 * i + 1];
            s[10] = sd[2 * (i + 1)];    s[11] = sd[2 * (i + 1) + 1];
        }
    }
    /* QHS:129-132 -> QHS:543-590.  The start tangent lands in the LAST segment's row 2 (QHS:561). */
    if (start_tan) { seg[(size_t)(G - 1) * 12 + 4] = start_tan[0]; seg[(size_t)(G - 1) * 12 + 5] = start_tan[1]; }
    if (end_tan) { seg[(size_t)(G - 1) * 12 + 6] = end_tan[0]; seg[(size_t)(G - 1) * 12 + 7] = end_tan[1]; }
    free(dist); free(fd); free(sd);
    return 0;
}

static double *dup_d(const double *src, int n, double fill)
{
    double *d = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) d[i] = src ? src[i] : fill;
    return d;
}
static int *dup_i(const int *src, int n)
{
    int *d = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) d[i] = src ? src[i] : 0;
    return d;
}

vapo_path *vapo_path_create(int W, const double *wp, const vapo_nodes *nodes,
                            const vapo_actions *actions)
{
    if (W < 2) return NULL; /* SM:50-51 */
    vapo_path *p = (vapo_path *)calloc(1, sizeof(vapo_path));
    p->W = W;
    p->rev = dup_i(nodes ? nodes->is_reverse : NULL, W);
    p->stop = dup_i(nodes ? nodes->stop : NULL, W);
    p->turn = dup_d(nodes ? nodes->turn : NULL, W, 0.0);
    p->wait = dup_d(nodes ? nodes->wait_time : NULL, W, 0.0);
    p->maxv = dup_d(nodes ? nodes->max_velocity : NULL, W, 0.0);
    p->maxa = dup_d(nodes ? nodes->max_acceleration : NULL, W, 0.0);
    p->M = actions ? actions->M : 0;
    p->ap_t = dup_d(actions ? actions->t : NULL, p->M, 0.0);
    p->ap_wait = dup_d(actions ? actions->wait_time : NULL, p->M, 0.0);
    p->ap_maxv = dup_d(actions ? actions->max_velocity : NULL, p->M, 0.0);
    p->ap_maxa = dup_d(actions ? actions->max_acceleration : NULL, p->M, 0.0);
    p->ap_stop = dup_i(actions ? actions->stop : NULL, p->M);
    p->sp = (spline_t *)calloc((size_t)W, sizeof(spline_t));
    p->seg = (double *)calloc((size_t)(W - 1) * 12, sizeof(double));
    p->seglen = (double *)calloc((size_t)(W - 1), sizeof(double));

    /* per-node [tangent*in_mag, tangent*out_mag] (SM:65-75) */
    double *tin = (double *)malloc(sizeof(double) * 2 * W);
    double *tout = (double *)malloc(sizeof(double) * 2 * W);
    for (int i = 0; i < W; i++) {
        int has = nodes && nodes->tangent && !isnan(nodes->tangent[2 * i]);
        for (int c = 0; c < 2; c++) {
            tin[2 * i + c] = has ? nodes->tangent[2 * i + c] * nodes->magnitudes[2 * i] : NAN;
            tout[2 * i + c] = has ? nodes->tangent[2 * i + c] * nodes->magnitudes[2 * i + 1] : NAN;
        }
    }
    int cur_start = 0, have_start_tan = 0, fail = 0, seg0 = 0;
    double start_tan[2] = {0, 0};
    for (int i = 1; i < W && !fail; i++) {
        int split = p->rev[i] || p->turn[i] != 0;
        if (!(split || i == W - 1)) continue;
        double end_tan[2];
        int have_end_tan = 0;
        double this_start[2] = {start_tan[0], start_tan[1]};
        int this_have_start = have_start_tan;
        have_start_tan = 0; /* SM:79-81 */
        if (split) { /* SM:84-158 */
            if (i >= W - 1) { fail = 1; break; } /* points[i+1] raises IndexError in the reference */
            const double *pm = wp + 2 * (i - 1), *pi = wp + 2 * i, *pn = wp + 2 * (i + 1);
            double prev_len = sqrt((pi[0] - pm[0]) * (pi[0] - pm[0]) + (pi[1] - pm[1]) * (pi[1] - pm[1]));
            double next_len = sqrt((pn[0] - pi[0]) * (pn[0] - pi[0]) + (pn[1] - pi[1]) * (pn[1] - pi[1]));
            double ps = prev_len > 0 ? 1.0 / prev_len : 1.0, ns = next_len > 0 ? 1.0 / next_len : 1.0;
            double pv[2] = {(pi[0] - pm[0]) * ps, (pi[1] - pm[1]) * ps};
            double nv[2] = {(pn[0] - pi[0]) * ns, (pn[1] - pi[1]) * ns};
            double min_len = prev_len < next_len ? prev_len : next_len;
            int has_tan = nodes && nodes->tangent && !isnan(nodes->tangent[2 * i]);
            if (p->turn[i] != 0) { /* SM:103-132 */
                double ang = p->turn[i] * (M_PI / 180.0); /* np.radians */
                if (p->rev[i]) ang = ang + M_PI;
                double c = cos(ang), s = sin(ang);
                /* rotation_matrix @ prev_vector */
                double nt[2] = {c * pv[0] + (-s) * pv[1], s * pv[0] + c * pv[1]};
                nt[0] *= min_len; nt[1] *= min_len;
                pv[0] *= min_len; pv[1] *= min_len;
                if (has_tan) {
                    const double *tg = nodes->tangent + 2 * i;
                    double im = nodes->magnitudes[2 * i], om = nodes->magnitudes[2 * i + 1];
                    pv[0] = tg[0] * im; pv[1] = tg[1] * im;
                    /* tangent @ rotation_matrix * -1 */
                    nt[0] = (tg[0] * c + tg[1] * s) * -1;
                    nt[1] = (tg[0] * (-s) + tg[1] * c) * -1;
                    nt[0] *= om; nt[1] *= om;
                }
                end_tan[0] = pv[0]; end_tan[1] = pv[1];
                start_tan[0] = nt[0]; start_tan[1] = nt[1];
            } else { /* reverse node, SM:134-158 */
                double dv[2] = {pv[0] - nv[0], pv[1] - nv[1]};
                double dn = sqrt(dv[0] * dv[0] + dv[1] * dv[1]);
                if (dn > 0) { dv[0] /= dn; dv[1] /= dn; }
                dv[0] *= min_len; dv[1] *= min_len;
                if (has_tan) {
                    dv[0] = nodes->tangent[2 * i] * nodes->magnitudes[2 * i];
                    dv[1] = nodes->tangent[2 * i + 1] * nodes->magnitudes[2 * i];
                }
                end_tan[0] = dv[0]; end_tan[1] = dv[1];
                start_tan[0] = -1 * dv[0]; start_tan[1] = -1 * dv[1];
                if (has_tan) {
                    start_tan[0] = -1 * nodes->tangent[2 * i] * nodes->magnitudes[2 * i + 1];
                    start_tan[1] = -1 * nodes->tangent[2 * i + 1] * nodes->magnitudes[2 * i + 1];
                }
            }
            have_end_tan = 1;
            have_start_tan = 1;
        }
        int npts = i - cur_start + 1;
        spline_t *s = &p->sp[p->n_splines];
        s->start = cur_start; s->npts = npts; s->seg0 = seg0;
        if (fit_spline(npts, wp + 2 * cur_start, tin + 2 * cur_start, tout + 2 * cur_start,
                       this_have_start ? this_start : NULL, have_end_tan ? end_tan : NULL,
                       p->seg + (size_t)seg0 * 12, p->seglen + seg0, &s->t_max) != 0) { fail = 1; break; }
        p->n_splines++;
        seg0 += npts - 1;
        if (split && i < W - 1) cur_start = i; /* SM:165-168 */
    }
    free(tin); free(tout);
    if (fail) { vapo_path_destroy(p); return NULL; }
    return p;
}

void vapo_path_destroy(vapo_path *p)
{
    if (!p) return;
    free(p->sp); free(p->seg); free(p->seglen);
    free(p->rev); free(p->stop); free(p->turn); free(p->wait); free(p->maxv); free(p->maxa);
    free(p->ap_t); free(p->ap_wait); free(p->ap_maxv); free(p->ap_maxa); free(p->ap_stop);
    free(p->lut_d); free(p->lut_p); free(p->tab_p); free(p->tab_k); free(p->tab_h);
    free(p);
}

int vapo_n_splines(const vapo_path *p) { return p->n_splines; }
int vapo_n_segments(const vapo_path *p) { return p->W - 1; }
void vapo_get_segments(const vapo_path *p, double *seg, double *seglen, double *param_last)
{
    if (seg) memcpy(seg, p->seg, sizeof(double) * 12 * (size_t)(p->W - 1));
    if (seglen) memcpy(seglen, p->seglen, sizeof(double) * (size_t)(p->W - 1));
    if (param_last) for (int i = 0; i < p->n_splines; i++) param_last[i] = p->sp[i].t_max;
}

/* SM:426-475 build_lookup_table */
static void build_lookup_table(vapo_path *p)
{
    int n = p->min_samples > 0 ? p->min_samples : LUT_MIN_SAMPLES;
    free(p->lut_d); free(p->lut_p);
    p->lut_n = n * p->n_splines;
    p->lut_d = (double *)malloc(sizeof(double) * p->lut_n);
    p->lut_p = (double *)malloc(sizeof(double) * p->lut_n);
    double current_dist = 0.0, prev_param = 0.0;
    double *mag = (double *)malloc(sizeof(double) * n);
    for (int si = 0; si < p->n_splines; si++) {
        const spline_t *s = &p->sp[si];
        double param_end = s->t_max;
        double dt = linspace_at(param_end, n, 1) - linspace_at(param_end, n, 0);
        for (int j = 0; j < n; j++) {
            double d[2];
            spline_eval(p, s, linspace_at(param_end, n, j), 1, d);
            mag[j] = norm2(d[0], d[1]);
        }
        double acc = 0.0; /* np.cumsum */
        for (int j = 0; j < n; j++) {
            double partial = 0.0;
            if (j > 0) { acc += (mag[j - 1] + mag[j]) * 0.5 * dt; partial = acc; }
            p->lut_d[si * n + j] = partial + current_dist;
            p->lut_p[si * n + j] = linspace_at(param_end, n, j) + prev_param;
        }
        current_dist = p->lut_d[si * n + n - 1];
        prev_param += param_end - 0.0;
    }
    free(mag);
    p->total = current_dist;
    p->have_lut = 1;
}

/* SM:477-548 precompute_path_properties */
static void precompute_path_properties(vapo_path *p)
{
    int n = p->W * (p->samples_per_node > 0 ? p->samples_per_node : SAMPLES_PER_NODE);
    free(p->tab_p); free(p->tab_k); free(p->tab_h);
    p->tab_n = n;
    p->tab_p = (double *)malloc(sizeof(double) * n);
    p->tab_k = (double *)malloc(sizeof(double) * n);
    p->tab_h = (double *)malloc(sizeof(double) * n);
    for (int j = 0; j < n; j++) {
        double t = linspace_at((double)(p->W - 1), n, j), d1[2], d2[2];
        vapo_derivative(p, t, d1);
        vapo_second_derivative(p, t, d2);
        double ss = d1[0] * d1[0] + d1[1] * d1[1];
        double num = d1[0] * d2[1] - d1[1] * d2[0];
        p->tab_p[j] = t;
        p->tab_k[j] = (ss >= 1e-10) ? num / pow(ss, 1.5) : 0.0;
        p->tab_h[j] = atan2(d1[1], d1[0]);
    }
    p->have_tab = 1;
}

void vapo_rebuild_tables(vapo_path *p)
{
    p->min_samples = p->samples_per_node = 0;   /* SM:582-594: both builders with their defaults */
    build_lookup_table(p);
    precompute_path_properties(p);
}

/* build_lookup_table(min_samples=...) and precompute_path_properties(samples_per_node=...) with explicit sizes */
int vapo_build_tables_sized(vapo_path *p, int min_samples, int samples_per_node)
{
    if (min_samples < 2 || samples_per_node < 1) return -1;   /* SM:444 indexes local_params[1] */
    p->min_samples = min_samples;
    p->samples_per_node = samples_per_node;
    build_lookup_table(p);
    precompute_path_properties(p);
    return 0;
}

int vapo_lut_size(const vapo_path *p) { return p->lut_n; }
void vapo_get_lut(const vapo_path *p, double *dist, double *param, double *total)
{
    if (dist) memcpy(dist, p->lut_d, sizeof(double) * p->lut_n);
    if (param) memcpy(param, p->lut_p, sizeof(double) * p->lut_n);
    if (total) *total = p->total;
}
int vapo_table_size(const vapo_path *p) { return p->tab_n; }
void vapo_get_table(const vapo_path *p, double *param, double *curv, double *head)
{
    if (param) memcpy(param, p->tab_p, sizeof(double) * p->tab_n);
    if (curv) memcpy(curv, p->tab_k, sizeof(double) * p->tab_n);
    if (head) memcpy(head, p->tab_h, sizeof(double) * p->tab_n);
}

double vapo_total_arc_length(const vapo_path *p) { return p->total; }

/* np.searchsorted(a, v, side="left"): first i with a[i] >= v */
static int searchsorted_left(const double *a, int n, double v)
{
    int lo = 0, hi = n;
    while (lo < hi) { int mid = lo + (hi - lo) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
/* side="right": first i with a[i] > v */
static long searchsorted_right_grid(double dd, long n, double v)
{
    long lo = 0, hi = n;
    while (lo < hi) { long mid = lo + (hi - lo) / 2; if (!(v < (double)mid * dd)) lo = mid + 1; else hi = mid; }
    return lo;
}

/* SM:291-318 */
double vapo_distance_to_time(const vapo_path *p, double s)
{
    if (s <= 0) return 0.0;
    if (s >= p->total) return (double)(p->W - 1);
    int idx = searchsorted_left(p->lut_d, p->lut_n, s);
    if (idx == 0) return p->lut_p[0];
    double d0 = p->lut_d[idx - 1], d1 = p->lut_d[idx];
    double t0 = p->lut_p[idx - 1], t1 = p->lut_p[idx];
    return t0 + (t1 - t0) * (s - d0) / (d1 - d0);
}

/* SM:550-580 _interpolate_property.  For two distinct table parameters t0 % 1 != t1 % 1 always
 * holds, so the branch at SM:577-578 is the one taken (a step lookup); SM:580 is kept for form. */
static double interpolate_property(const vapo_path *p, double t, const double *vals)
{
    int idx = searchsorted_left(p->tab_p, p->tab_n, t);
    if (idx == 0) return vals[0];
    if (idx >= p->tab_n) return vals[p->tab_n - 1];
    double t0 = p->tab_p[idx - 1], t1 = p->tab_p[idx];
    if (fmod(t0, 1.0) != fmod(t1, 1.0)) return fmod(t, 1.0) > 0.5 ? vals[idx - 1] : vals[idx];
    return vals[idx - 1] + (vals[idx] - vals[idx - 1]) * (t - t0) / (t1 - t0);
}
double vapo_curvature(const vapo_path *p, double t) { return interpolate_property(p, t, p->tab_k); }
double vapo_heading(const vapo_path *p, double t) { return interpolate_property(p, t, p->tab_h); }

long vapo_count_samples(const vapo_path *p, double dd)
{
    long n = 0;
    double s = 0;
    while (s < p->total) { n++; s += dd; }
    return n + 1;
}

double vapo_dd_for_samples(const vapo_path *p, long S) { return p->total / ((double)S - 1.5); }

/* Python min(a, b): keeps a unless b < a (a NaN in b is skipped, a NaN in a sticks) */
static double pymin(double a, double b) { return b < a ? b : a; }

/* MPG:23-33 */
static double max_speed_at_curvature(double max_vel, double tw, double curvature)
{
    if (fabs(curvature) < 1e-6) return max_vel;
    double m = ((2 * max_vel / tw) * max_vel) / (fabs(curvature) * max_vel + (2 * max_vel / tw));
    return pymin(m, max_vel);
}
/* MPG:52-59 */
static double max_accels_at_turn(double max_acc, double tw, double angular_accel)
{
    double left = max_acc + angular_accel * tw / 2;
    double right = max_acc - angular_accel * tw / 2;
    return fabs(left) < fabs(right) ? left : right;
}

/* MPG:70-316 */
long vapo_forward_backward(const vapo_path *p, const double c[6], double dd, double start_vel,
                           double end_vel, long cap, double *ot, double *ox, double *oy,
                           double *oh, double *ok, double *ov)
{
    double max_vel = c[0], max_acc = c[1], max_dec = c[2], tw = c[5];
    double max_angular_vel = 2 * max_vel / tw;       /* MPG:81 */
    double max_angular_accel = 2 * max_acc / tw;     /* MPG:82 */
    long N = vapo_count_samples(p, dd);
    if (cap < N) return -1;
    double *v = (double *)malloc(sizeof(double) * N);
    double *K = (double *)malloc(sizeof(double) * N);
    double *H = (double *)malloc(sizeof(double) * N);
    /* boundary_map as a dense array: -1 = no entry */
    long *bmap = (long *)malloc(sizeof(long) * N);
    double *max_accels = (double *)malloc(sizeof(double) * (size_t)(p->W + p->M + 2));
    int n_acc = 0;
    for (long i = 0; i < N; i++) bmap[i] = -1;

    double total = p->total, cur = 0;
    double prev_t = 0;
    int node_num = 0, action_idx = 0;
    double max_velocity = max_vel;
    max_accels[n_acc++] = p->maxa[0] > 0 ? p->maxa[0] : max_acc; /* MPG:100-104 */
    if (p->maxv[0] > 0) max_velocity = p->maxv[0];
    bmap[0] = 0;
    double t_end = vapo_distance_to_time(p, total);
    long i = 0;
    while (cur < total) { /* MPG:112-167 */
        double t = vapo_distance_to_time(p, cur);
        K[i] = vapo_curvature(p, t);
        H[i] = vapo_heading(p, t);
        if (ot) ot[i] = t;
        if (ox || oy) { double q[2]; vapo_point(p, t, q); if (ox) ox[i] = q[0]; if (oy) oy[i] = q[1]; }
        v[i] = max_velocity;
        cur += dd;
        if (fmod(prev_t, 1.0) > fmod(t, 1.0) && t < t_end) { /* MPG:124-140 */
            node_num += 1;
            if (p->stop[node_num]) v[i] = 0.01;
            max_velocity = p->maxv[node_num] > 0 ? p->maxv[node_num] : max_vel;
            max_accels[n_acc++] = p->maxa[node_num] > 0 ? p->maxa[node_num] : max_acc;
            if (node_num < p->W - 1) bmap[i] = n_acc - 1;
        }
        if (action_idx < p->M && prev_t < p->ap_t[action_idx] && t >= p->ap_t[action_idx]) { /* MPG:142-163 */
            max_velocity = p->ap_maxv[action_idx] > 0 ? p->ap_maxv[action_idx] : max_vel;
            if (p->ap_stop[action_idx]) v[i] = 0.01;
            max_accels[n_acc++] = p->ap_maxa[action_idx] > 0 ? p->ap_maxa[action_idx] : max_acc;
            bmap[i] = n_acc - 1;
            action_idx += 1;
        }
        i += 1;
        prev_t = t;
    }
    /* MPG:171-176 final point */
    v[i] = end_vel;
    {
        double t = vapo_distance_to_time(p, total);
        H[i] = vapo_heading(p, t);
        K[i] = vapo_curvature(p, t);
        if (ot) ot[i] = t;
        if (ox || oy) { double q[2]; vapo_point(p, t, q); if (ox) ox[i] = q[0]; if (oy) oy[i] = q[1]; }
    }
    max_accels[n_acc++] = max_acc;
    /* (MPG:178-186 curvature_derivs: computed by the reference, never read) */

    /* forward pass MPG:188-249 */
    double c_acc = max_acc, c_dec = max_dec; /* constraints.max_acc / max_dec as mutated */
    v[0] = start_vel;
    double prev_ang_vel = 0, accel_ang = 0;
    for (long k = 0; k < N - 1; k++) {
        if (bmap[k] >= 0) { c_acc = max_accels[bmap[k]]; c_dec = max_accels[bmap[k]]; }
        double current_vel = v[k], curvature = K[k];
        double ang_vel = v[k] * fabs(curvature);
        double max_linear_vel, max_accel;
        if (fabs(curvature) < 1e-6) {
            max_linear_vel = max_vel;
            max_accel = c_acc;
        } else {
            double delta_theta = H[k + 1] - H[k];
            accel_ang = (ang_vel * ang_vel - prev_ang_vel * prev_ang_vel) / (2 * fabs(delta_theta));
            double max_vel_ang = max_angular_vel / fabs(curvature);
            double max_vel_kin = 2 * max_vel / (tw * fabs(curvature) + 2);
            double max_curve_vel = max_speed_at_curvature(max_vel, tw, fabs(curvature));
            max_linear_vel = pymin(pymin(max_vel_ang, max_vel_kin), max_curve_vel);
            double max_accel_ang = max_angular_accel / fabs(curvature);
            double max_accel_kin = 2 * c_acc / (tw * fabs(curvature) + 2);
            double max_accel_wheel = max_accels_at_turn(c_acc, tw, fabs(accel_ang));
            if (max_accel_wheel < 0) max_accel_wheel = 0;
            max_accel = pymin(pymin(pymin(max_accel_ang, max_accel_kin), max_accel_wheel), c_acc);
        }
        double next_vel = pymin(max_linear_vel, sqrt(current_vel * current_vel + 2 * max_accel * dd));
        v[k + 1] = pymin(v[k + 1], next_vel);
        prev_ang_vel = ang_vel;
        v[k + 1] = pymin(v[k + 1], fabs(max_vel / (1 + (tw * fabs(curvature) / 2))));
    }
    /* backward pass MPG:251-311 */
    v[N - 1] = end_vel;
    prev_ang_vel = 0;
    for (long k = N - 1; k > 0; k--) {
        if (bmap[k] >= 0) c_acc = max_accels[bmap[k] + 1];
        double current_vel = v[k], curvature = K[k];
        double ang_vel = v[k] * fabs(curvature);
        double max_linear_vel, max_decel;
        if (fabs(curvature) < 1e-6) {
            max_linear_vel = max_vel;
            max_decel = c_dec;
        } else {
            double delta_theta = H[k - 1] - H[k];
            accel_ang = (ang_vel * ang_vel - prev_ang_vel * prev_ang_vel) / (2 * fabs(delta_theta));
            double max_vel_ang = max_angular_vel / fabs(curvature);
            double max_vel_kin = 2 * max_vel / (tw * fabs(curvature) + 2);
            double max_curve_vel = max_speed_at_curvature(max_vel, tw, curvature);
            max_linear_vel = pymin(pymin(max_vel_ang, max_vel_kin), max_curve_vel);
            double max_decel_ang = max_angular_accel / fabs(curvature);
            double max_decel_kin = 2 * c_dec / (tw * fabs(curvature) + 2);
            double max_accel_wheel = max_accels_at_turn(c_acc, tw, accel_ang);
            if (max_accel_wheel < 0) max_accel_wheel = 0;
            max_decel = pymin(pymin(pymin(max_decel_ang, max_decel_kin), max_accel_wheel), c_dec);
        }
        double prev_vel = sqrt(current_vel * current_vel + 2 * max_decel * dd);
        prev_vel = pymin(pymin(prev_vel, v[k - 1]), max_linear_vel);
        v[k - 1] = prev_vel;
        prev_ang_vel = ang_vel;
        v[k - 1] = pymin(v[k - 1], fabs(max_vel / (1 + (tw * fabs(curvature) / 2))));
    }
    for (long k = 0; k < N; k++) {
        if (ov) ov[k] = v[k];
        if (ok) ok[k] = K[k];
        if (oh) oh[k] = H[k];
    }
    free(v); free(K); free(H); free(bmap); free(max_accels);
    return N;
}

/* ---------------- batch driver (cpu_baseline) ---------------- */
typedef struct {
    int B, W; long S; const double *wp; const double *c; double sv, ev;
    double *x, *y, *h, *k, *v, *L;
    int first, stride; int err;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *j = (batch_job *)arg;
    for (int b = j->first; b < j->B; b += j->stride) {
        vapo_path *p = vapo_path_create(j->W, j->wp + (size_t)b * j->W * 2, NULL, NULL);
        if (!p) { if (!j->err) j->err = b + 1; continue; }
        vapo_rebuild_tables(p);
        double dd = vapo_dd_for_samples(p, j->S);
        size_t o = (size_t)b * (size_t)j->S;
        long n = vapo_forward_backward(p, j->c, dd, j->sv, j->ev, j->S, NULL, j->x ? j->x + o : NULL,
                                       j->y ? j->y + o : NULL, j->h ? j->h + o : NULL,
                                       j->k ? j->k + o : NULL, j->v ? j->v + o : NULL);
        if (n != j->S && !j->err) j->err = b + 1;
        if (j->L) j->L[b] = p->total;
        vapo_path_destroy(p);
    }
    return NULL;
}

int vapo_profile_batch(int B, int W, long S, const double *waypoints, const double c[6],
                       double start_vel, double end_vel, double *x, double *y, double *heading,
                       double *curvature, double *velocity, double *total_length, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    batch_job jobs[256];
    pthread_t th[256];
    for (int t = 0; t < n_threads; t++) {
        batch_job j = {B, W, S, waypoints, c, start_vel, end_vel, x, y, heading, curvature, velocity,
                       total_length, t, n_threads, 0};
        jobs[t] = j;
    }
    if (n_threads == 1) batch_worker(&jobs[0]);
    else {
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    for (int t = 0; t < n_threads; t++) if (jobs[t].err) return jobs[t].err;
    return 0;
}

/* ---------------- time-domain resample: MPG:319-628, ODM:4-69 ---------------- */

/* Python float % for a positive divisor */
static double pymod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0.0) { if ((b < 0) != (r < 0)) r += b; }
    else r = copysign(0.0, b);
    return r;
}

/* ODM:4-69 generate_trapezoidal_profile -> velocity array; returns length, fills vel (cap) */
static long trapezoidal_profile(double max_velocity, double max_acceleration, double total_distance,
                                double time_step, double **out)
{
    double time_to_max_vel = max_velocity / max_acceleration;
    double dist_accel = 0.5 * max_acceleration * (time_to_max_vel * time_to_max_vel);
    double total_time;
    if (2 * dist_accel > total_distance) {
        time_to_max_vel = sqrt(total_distance / max_acceleration);
        max_velocity = max_acceleration * time_to_max_vel;
        total_time = 2 * time_to_max_vel;
    } else {
        double dist_constant_vel = total_distance - 2 * dist_accel;
        double time_constant_vel = dist_constant_vel / max_velocity;
        total_time = 2 * time_to_max_vel + time_constant_vel;
    }
    /* np.arange(0, total_time + time_step, time_step) */
    double stop = total_time + time_step;
    long n = (long)ceil((stop - 0.0) / time_step);
    if (n < 0) n = 0;
    double *vel = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
    for (long i = 0; i < n; i++) {
        double t = 0.0 + (double)i * time_step;
        if (t <= time_to_max_vel) vel[i] = max_acceleration * t;
        else if (t <= total_time - time_to_max_vel) vel[i] = max_velocity;
        else {
            double time_in_decel = t - (total_time - time_to_max_vel);
            vel[i] = max_velocity - max_acceleration * time_in_decel;
        }
    }
    *out = vel;
    return n;
}

/* MPG:319-346 motion_profile_angle -> headings, angular velocities */
static long motion_profile_angle(double angle, const double c[6], double dt, double **heads,
                                 double **angvels)
{
    double tw = c[5];
    double arc_length = fabs(angle) * tw / 2;
    double *vel;
    long n = trapezoidal_profile(c[0], c[1], arc_length, dt, &vel);
    double *h = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
    double *w = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
    double accum = 0;
    for (long i = 0; i < n; i++) {
        double current_angle = accum / (tw / 2);
        h[i] = current_angle * (angle > 0 ? -1 : 1);
        accum += vel[i] * dt;
    }
    /* angular_velocities = [0] + diffs: has length max(n,1) in the reference; with n >= 1 always */
    if (n > 0) w[0] = 0;
    for (long i = 1; i < n; i++) w[i] = (h[i] - h[i - 1]) / dt;
    free(vel);
    *heads = h; *angvels = w;
    return n;
}

/* MPG:349-386 lerp over x_array[i] = i*dd (MPG:484), y_array = velocities */
static double lerp_grid(double x, double dd, const double *ys, long n)
{
    long idx = searchsorted_right_grid(dd, n, x) - 1;
    if (idx < 0) return ys[0];
    if (idx >= n - 1) return ys[n - 1];
    double x0 = (double)idx * dd, x1 = (double)(idx + 1) * dd;
    double y0 = ys[idx], y1 = ys[idx + 1];
    return y0 + (x - x0) * (y1 - y0) / (x1 - x0);
}

static double clipd(double x, double lo, double hi) { double m = x < lo ? lo : x; return m > hi ? hi : m; }

#define ROW(r) (out + (size_t)(r) * 8)

long vapo_generate_motion_profile(vapo_path *p, const double c[6], double dt, double dd, long cap,
                                  double *out, long *nodes_map, int *n_nodes_map,
                                  long *actions_map, int *n_actions_map)
{
    vapo_rebuild_tables(p); /* MPG:402 */
    long N = vapo_count_samples(p, dd);
    double *vel = (double *)malloc(sizeof(double) * N);
    vapo_forward_backward(p, c, dd, 0.01, 0.01, N, NULL, NULL, NULL, NULL, NULL, vel); /* MPG:408 */
    long T = 0;
    int nn = 0, na = 0;
    long ret = 0;
    nodes_map[nn++] = 0; /* MPG:420 */
    double current_time = 0, current_pos = 0, current_vel = vel[0];
    double total_length = p->total;
    int is_reversed = 0;
    int node_idx = 0;
    if (p->rev[0]) is_reversed = !is_reversed;
    if (p->turn[0] != 0) { free(vel); return -2; } /* MPG:440 headings[-1] on an empty list */
    if (p->wait[0] > 0) { /* MPG:459-476 */
        long steps = (long)(p->wait[0] / dt);
        double h = -1 * vapo_heading(p, 0);
        if (is_reversed) h -= M_PI;
        if (h > M_PI) h -= 2 * M_PI;
        if (h < -M_PI) h += 2 * M_PI;
        double q[2];
        vapo_point(p, 0, q);
        if (T + steps > cap) { free(vel); return -1; }
        for (long i = 0; i < steps; i++) {
            double *r = ROW(T + i);
            r[0] = current_time + i * dt; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = h; r[5] = 0; r[6] = q[0]; r[7] = q[1];
        }
        T += steps;
        current_time += steps * dt;
    }
    double prev_t = 0;
    int action_idx = 0;
    node_idx = 0;
    double end_param = vapo_distance_to_time(p, total_length);
    while (current_pos < total_length) { /* MPG:523-600 */
        double t = vapo_distance_to_time(p, current_pos);
        if (fmod(t, 1.0) < fmod(prev_t, 1.0) && t < end_param) { /* MPG:527-544 */
            nodes_map[nn++] = T;
            node_idx += 1;
            if (p->turn[node_idx] != 0) { /* handle_turn MPG:487-507 */
                if (T == 0) { ret = -2; break; }
                double angle = p->turn[node_idx] * (M_PI / 180.0);
                double start_heading = ROW(T - 1)[4];
                double *ih, *iw;
                long n = motion_profile_angle(angle, c, dt, &ih, &iw);
                for (long i = 0; i < n; i++) {
                    while (ih[i] + start_heading > M_PI) ih[i] -= 2 * M_PI;
                    while (ih[i] + start_heading < -M_PI) ih[i] += 2 * M_PI;
                }
                if (T + n > cap) { free(ih); free(iw); ret = -1; break; }
                double lastpos = ROW(T - 1)[1], lx = ROW(T - 1)[6], ly = ROW(T - 1)[7];
                for (long i = 0; i < n; i++) {
                    double *r = ROW(T + i);
                    r[0] = current_time + i * dt; r[1] = lastpos; r[2] = 0; r[3] = 0;
                    r[4] = start_heading + ih[i]; r[5] = iw[i]; r[6] = lx; r[7] = ly;
                }
                T += n;
                current_time = current_time + n * dt;
                free(ih); free(iw);
            }
            if (p->rev[node_idx]) is_reversed = !is_reversed;
            if (p->wait[node_idx] > 0) { /* handle_wait MPG:509-518 */
                if (T == 0) { ret = -2; break; }
                long steps = (long)(p->wait[node_idx] / dt);
                if (T + steps > cap) { ret = -1; break; }
                double lh = ROW(T - 1)[4], lx = ROW(T - 1)[6], ly = ROW(T - 1)[7];
                for (long i = 0; i < steps; i++) {
                    double *r = ROW(T + i);
                    r[0] = current_time + i * dt; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = lh; r[5] = 0; r[6] = lx; r[7] = ly;
                }
                T += steps;
                current_time = current_time + steps * dt;
            }
        }
        if (action_idx < p->M) { /* MPG:547-553 */
            double at = p->ap_t[action_idx];
            if (prev_t < at && at < t) {
                actions_map[na++] = T;
                if (p->ap_wait[action_idx] > 0) {
                    if (T == 0) { ret = -2; break; }
                    long steps = (long)(p->ap_wait[action_idx] / dt);
                    if (T + steps > cap) { ret = -1; break; }
                    double lh = ROW(T - 1)[4], lx = ROW(T - 1)[6], ly = ROW(T - 1)[7];
                    for (long i = 0; i < steps; i++) {
                        double *r = ROW(T + i);
                        r[0] = current_time + i * dt; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = lh; r[5] = 0; r[6] = lx; r[7] = ly;
                    }
                    T += steps;
                    current_time = current_time + steps * dt;
                }
                action_idx += 1;
            }
        }
        prev_t = t;
        double curvature = vapo_curvature(p, t);
        double heading = vapo_heading(p, t) - (is_reversed ? M_PI : 0);
        heading = pymod(heading + M_PI, 2 * M_PI) - M_PI;
        heading *= -1;
        double q[2];
        vapo_point(p, t, q);
        double target_vel = lerp_grid(current_pos, dd, vel, N);
        double next_target_vel = lerp_grid(current_pos + dd, dd, vel, N);
        target_vel = (target_vel + next_target_vel) / 2;
        if (!(target_vel > 0.001)) target_vel = 0.001; /* max(x, 0.001) */
        double accel = clipd((target_vel - current_vel) / dt, -c[2], c[1]);
        double angular_vel = target_vel * curvature * -1;
        current_vel = clipd(current_vel + accel * dt, 0, target_vel);
        double delta_pos = current_vel * dt + 0.5 * accel * dt * dt;
        if (current_vel <= 0.1) delta_pos = 0.1 * dt + 0.5 * accel * dt * dt;
        current_pos += delta_pos;
        if (T + 1 > cap) { ret = -1; break; }
        double *r = ROW(T);
        r[0] = current_time; r[1] = current_pos; r[2] = current_vel * (is_reversed ? -1 : 1);
        r[3] = accel * (is_reversed ? -1 : 1); r[4] = heading; r[5] = angular_vel; r[6] = q[0]; r[7] = q[1];
        T += 1;
        current_time += dt;
    }
    free(vel);
    *n_nodes_map = nn;
    *n_actions_map = na;
    return ret < 0 ? ret : T;
}
