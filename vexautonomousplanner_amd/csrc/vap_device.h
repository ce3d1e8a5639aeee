// vap_device.h — device-side building blocks shared by the kernels (gfx950).
//
// Built with -ffp-contract=off: the parameter / index arithmetic must round exactly like the
// reference's NumPy fp64; FMAs appear only where written as fma().
//
// Reference citations: QHS = splines/quintic_hermite_spline.py, SM = splines/spline_manager.py,
// MPG = motion_profiling_v2/motion_profile_generator.py (under the reference's src/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

namespace vap {

constexpr int kLutN = 1000;        // SM:427
constexpr int kSamplesPerNode = 1000;  // SM:477

// meta[b][4] = {param_last, total_length, dd, n_samples}             (public, include/vap.h)
constexpr int kMetaStride = 4;
// aux[b][4]  = {lut_step, table_step, 1/table_step, spare}              (internal scratch)
constexpr int kAuxStride = 4;

// np.linspace(0, stop, num)[j] (numpy/_core/function_base.py): j*step, endpoint forced to stop.
__device__ __forceinline__ double linspace_at(double stop, int num, int j)
{
    if (j == num - 1) return stop;
    const double step = stop / (double)(num - 1);
    return (double)j * step;
}

// QHS:506-541 _normalize_parameter for a spline with G segments and parameters[-1] = t_max.
__device__ __forceinline__ void normalize_parameter(double t, double t_max, int G, double &lt, int &idx)
{
    double tt = t < t_max ? t : t_max;
    tt = 0.0 > tt ? 0.0 : tt;
    int i = (int)tt;
    if (i == G) i = G - 1;
    lt = tt - (double)i;
    idx = i;
}

// QHS:324-363 first-derivative basis, in the reference's own association order (no FMA): used where
// bit-identical arc-length tables matter (SM:447).
__device__ __forceinline__ void hermite_d1_basis_ref(double t, double H[6])
{
    const double t2 = t * t, t3 = t2 * t, t4 = t3 * t;
    H[0] = -30 * t2 + 60 * t3 - 30 * t4;
    H[1] = 30 * t2 - 60 * t3 + 30 * t4;
    H[2] = 1 - 18 * t2 + 32 * t3 - 15 * t4;
    H[3] = -12 * t2 + 28 * t3 - 15 * t4;
    H[4] = t - 4.5 * t2 + 6 * t3 - 2.5 * t4;
    H[5] = 1.5 * t2 - 4 * t3 + 2.5 * t4;
}
// ... and the sum over the segment's rows, accumulated from zero in row order (QHS:482-484)
__device__ __forceinline__ void hermite_combine_ref(const double *__restrict__ sg, const double H[6], double &ox, double &oy)
{
    double ax = 0.0, ay = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        ax += H[i] * sg[2 * i];
        ay += H[i] * sg[2 * i + 1];
    }
    ox = ax;
    oy = ay;
}
__device__ __forceinline__ void hermite_d1_ref(const double *__restrict__ sg, double t, double &ox, double &oy)
{
    double H[6];
    hermite_d1_basis_ref(t, H);
    hermite_combine_ref(sg, H, ox, oy);
}

// QHS:288-322 / 324-363 / 365-416 / 418-469 the position, first-, second- and third-derivative bases in the reference's
// association order.  (The third-derivative basis is in the reference's class but none of its callers: here it is
// reachable through vap_basis_host, order 3.)
__device__ __forceinline__ void hermite_basis_ref(int order, double t, double H[6])
{
    const double t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
    if (order == 3) {
        H[0] = -60 + 360 * t - 360 * t2;
        H[1] = 60 - 360 * t + 360 * t2;
        H[2] = -36 + 192 * t - 180 * t2;
        H[3] = -24 + 168 * t - 180 * t2;
        H[4] = -9 + 36 * t - 30 * t2;
        H[5] = 3 - 24 * t + 30 * t2;
    } else if (order == 0) {
        H[0] = 1 - 10 * t3 + 15 * t4 - 6 * t5;
        H[1] = 10 * t3 - 15 * t4 + 6 * t5;
        H[2] = t - 6 * t3 + 8 * t4 - 3 * t5;
        H[3] = -4 * t3 + 7 * t4 - 3 * t5;
        H[4] = 0.5 * t2 - 1.5 * t3 + 1.5 * t4 - 0.5 * t5;
        H[5] = 0.5 * t3 - t4 + 0.5 * t5;
    } else if (order == 1) {
        H[0] = -30 * t2 + 60 * t3 - 30 * t4;
        H[1] = 30 * t2 - 60 * t3 + 30 * t4;
        H[2] = 1 - 18 * t2 + 32 * t3 - 15 * t4;
        H[3] = -12 * t2 + 28 * t3 - 15 * t4;
        H[4] = t - 4.5 * t2 + 6 * t3 - 2.5 * t4;
        H[5] = 1.5 * t2 - 4 * t3 + 2.5 * t4;
    } else {
        H[0] = -60 * t + 180 * t2 - 120 * t3;
        H[1] = 60 * t - 180 * t2 + 120 * t3;
        H[2] = -36 * t + 96 * t2 - 60 * t3;
        H[3] = -24 * t + 84 * t2 - 60 * t3;
        H[4] = 1 - 9 * t + 18 * t2 - 10 * t3;
        H[5] = 3 * t - 12 * t2 + 10 * t3;
    }
}

// QHS:221-251 / 473-504: sum_i basis_i * segment[idx][i], accumulated from zero in row order.
__device__ __forceinline__ void hermite_eval_ref(const double *__restrict__ seg, double t_max, int G, int order,
                                                 double t, double &ox, double &oy)
{
    double lt, H[6];
    int idx;
    normalize_parameter(t, t_max, G, lt, idx);
    hermite_basis_ref(order, lt, H);
    const double *sg = seg + (size_t)idx * 12;
    double ax = 0.0, ay = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        ax += H[i] * sg[2 * i];
        ay += H[i] * sg[2 * i + 1];
    }
    ox = ax;
    oy = ay;
}

// Hermite rows [p0,p1,d0,d1,dd0,dd1] -> monomial coefficients c0..c5 of P(t) = sum c_j t^j, for one
// coordinate.  (Expansion of the basis polynomials at QHS:293-298; p1-p0 is formed first so the
// near-cancelling high-order terms of a gently curved segment stay small and accurate.)
__device__ __forceinline__ void hermite_to_power(double r0, double r1, double r2, double r3, double r4,
                                                 double r5, double c[6])
{
    const double d = r1 - r0;
    c[0] = r0;
    c[1] = r2;
    c[2] = 0.5 * r4;
    c[3] = 10 * d - 6 * r2 - 4 * r3 - 1.5 * r4 + 0.5 * r5;
    c[4] = -15 * d + 8 * r2 + 7 * r3 + 1.5 * r4 - r5;
    c[5] = 6 * d - 3 * r2 - 3 * r3 - 0.5 * r4 + 0.5 * r5;
}

// Per-segment coefficient block used by the sampling kernel (kCoefDoubles doubles):
//   [0..4]   x: j*c_j (j=1..5)        [5..9]   y          P'   (fp64)
//   [10..13] x: j(j-1)*c_j (j=2..5)   [14..17] y          P''  (fp64)
//   [18..23] x: c0..c5                [24..29] y          P    (fp64)
//   [30..35] the same 12 position coefficients as fp32 (x then y), for the fp32 output mode
constexpr int kCoefDoubles = 36;
constexpr int kCoefD1 = 0, kCoefD2 = 10, kCoefP = 18, kCoefPf = 30;
__device__ __forceinline__ void make_coef_block(const double r[12], double *__restrict__ o)
{
    double cx[6], cy[6];
    hermite_to_power(r[0], r[2], r[4], r[6], r[8], r[10], cx);
    hermite_to_power(r[1], r[3], r[5], r[7], r[9], r[11], cy);
#pragma unroll
    for (int k = 1; k < 6; k++) { o[kCoefD1 + k - 1] = (double)k * cx[k]; o[kCoefD1 + 5 + k - 1] = (double)k * cy[k]; }
#pragma unroll
    for (int k = 2; k < 6; k++) { o[kCoefD2 + k - 2] = (double)(k * (k - 1)) * cx[k]; o[kCoefD2 + 4 + k - 2] = (double)(k * (k - 1)) * cy[k]; }
#pragma unroll
    for (int k = 0; k < 6; k++) { o[kCoefP + k] = cx[k]; o[kCoefP + 6 + k] = cy[k]; }
    float *f = reinterpret_cast<float *>(o + kCoefPf);
#pragma unroll
    for (int k = 0; k < 6; k++) { f[k] = (float)cx[k]; f[6 + k] = (float)cy[k]; }
}
__device__ __forceinline__ float horner5f(const float *__restrict__ c, float t)
{
    return fmaf(fmaf(fmaf(fmaf(fmaf(c[5], t, c[4]), t, c[3]), t, c[2]), t, c[1]), t, c[0]);
}
__device__ __forceinline__ double horner5(const double *__restrict__ c, double t)   // c[0..5]
{
    return fma(fma(fma(fma(fma(c[5], t, c[4]), t, c[3]), t, c[2]), t, c[1]), t, c[0]);
}
__device__ __forceinline__ double horner4(const double *__restrict__ c, double t)   // c[0..4]
{
    return fma(fma(fma(fma(c[4], t, c[3]), t, c[2]), t, c[1]), t, c[0]);
}
__device__ __forceinline__ double horner3(const double *__restrict__ c, double t)   // c[0..3]
{
    return fma(fma(fma(c[3], t, c[2]), t, c[1]), t, c[0]);
}

// P, P', P'' of one coordinate from monomial coefficients (Horner, explicit FMA).
__device__ __forceinline__ double poly_p(const double *__restrict__ c, double t)
{
    return fma(fma(fma(fma(fma(c[5], t, c[4]), t, c[3]), t, c[2]), t, c[1]), t, c[0]);
}
__device__ __forceinline__ double poly_d1(const double *__restrict__ c, double t)
{
    return fma(fma(fma(fma(5.0 * c[5], t, 4.0 * c[4]), t, 3.0 * c[3]), t, 2.0 * c[2]), t, c[1]);
}
__device__ __forceinline__ double poly_d2(const double *__restrict__ c, double t)
{
    return fma(fma(fma(20.0 * c[5], t, 12.0 * c[4]), t, 6.0 * c[3]), t, 2.0 * c[2]);
}

// np.searchsorted(D, s, side="left") over the 1000-entry distance table: first j with D[j] >= s.
__device__ __forceinline__ int lut_search_left(const double *__restrict__ D, double s)
{
    int lo = 0, hi = kLutN;
#pragma unroll 1
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (D[mid] < s) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// SM:291-318 distance_to_time.  D = lookup_table.distances (LDS or global), parameters are
// linspace(0, t_max, 1000) (SM:443).  `end_param` = len(nodes)-1.
__device__ __forceinline__ double distance_to_time(const double *__restrict__ D, double total, double t_max,
                                                    double end_param, double s)
{
    if (s <= 0) return 0.0;
    if (s >= total) return end_param;
    const int idx = lut_search_left(D, s);
    if (idx == 0) return 0.0;
    const double d0 = D[idx - 1], d1 = D[idx];
    const double t0 = linspace_at(t_max, kLutN, idx - 1), t1 = linspace_at(t_max, kLutN, idx);
    return t0 + (t1 - t0) * (s - d0) / (d1 - d0);
}

// ------------------------------------------------------------------------------------------------
// Routes that reverse / turn nodes cut into several splines (SM:57-168): the lookup table is the concatenation of
// the splines' 1000-entry tables with running offsets (SM:436-464), and a global parameter maps to the spline that
// holds it, a split node belonging to the EARLIER spline (SM:243-275: t <= segment_end).  Segment i always joins
// nodes i and i+1 (splines share their split node), so a spline is {first node, point count, parameters[-1]}.
//   sp[s][4] = {parameters[-1] of spline s, distance offset, parameter offset, first node}      (kSplineStride)
// A plain path is the case of one spline with zero offsets: adding 0.0 changes no bit, so the functions below give
// the single-spline numbers (distance_to_time above) exactly.
// ------------------------------------------------------------------------------------------------
constexpr int kSplineStride = 4;
struct LutView {
    const double *D;     // [n_spl][1000] partial distances of each spline (SM:448-454)
    const double *sp;    // [n_spl][kSplineStride]
    int n_spl;
    double total;        // lookup_table.total_length
    double end_param;    // len(nodes) - 1
};
__device__ __forceinline__ double lutv_d(const LutView &v, int e)      // lookup_table.distances[e] (SM:457)
{
    const int s = e / kLutN, j = e - s * kLutN;
    return v.D[(size_t)s * kLutN + j] + v.sp[s * kSplineStride + 1];
}
__device__ __forceinline__ double lutv_p(const LutView &v, int e)      // lookup_table.parameters[e] (SM:461)
{
    const int s = e / kLutN, j = e - s * kLutN;
    return linspace_at(v.sp[s * kSplineStride + 0], kLutN, j) + v.sp[s * kSplineStride + 2];
}
// SM:291-318 distance_to_time on the concatenated table
__device__ __forceinline__ double lutv_distance_to_time(const LutView &v, double s)
{
    if (s <= 0) return 0.0;
    if (s >= v.total) return v.end_param;
    int lo = 0, hi = v.n_spl * kLutN;
#pragma unroll 1
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (lutv_d(v, mid) < s) lo = mid + 1;
        else hi = mid;
    }
    if (lo == 0) return lutv_p(v, 0);
    const double d0 = lutv_d(v, lo - 1), d1 = lutv_d(v, lo);
    const double t0 = lutv_p(v, lo - 1), t1 = lutv_p(v, lo);
    return t0 + (t1 - t0) * (s - d0) / (d1 - d0);
}
// SM:243-275 _map_parameter_to_spline + QHS:506-541 _normalize_parameter: global parameter -> (segment, local t)
__device__ __forceinline__ void lutv_map_parameter(const LutView &v, int W, double t, int &seg, double &lt)
{
    int si = v.n_spl - 1;
#pragma unroll 1
    for (int i = 0; i < v.n_spl - 1; i++) {
        const double end = v.sp[(i + 1) * kSplineStride + 3];     // this spline's last node = the next one's first
        if (t <= end) { si = i; break; }
    }
    const int first = (int)v.sp[si * kSplineStride + 3];
    const int last = si + 1 < v.n_spl ? (int)v.sp[(si + 1) * kSplineStride + 3] : W - 1;
    int idx;
    normalize_parameter(t - (double)first, v.sp[si * kSplineStride + 0], last - first, lt, idx);
    seg = first + idx;
}

// SM:550-580 _interpolate_property reduced to what it always does (a step lookup):
// returns the index into the linspace(0, W-1, 1000*W) property table that the reference reads for
// parameter t.  tab_n = 1000*W, end_param = W-1.
// np.searchsorted(linspace(0, end_param, tab_n), t) (side "left"; t <= end_param): first j with tp[j] >= t
__device__ __forceinline__ int table_search(double t, int tab_n, double end_param)
{
    const double step = end_param / (double)(tab_n - 1);
    long j = (long)ceil(t / step);
    if (j < 0) j = 0;
    if (j > tab_n - 1) j = tab_n - 1;
    while (j > 0 && linspace_at(end_param, tab_n, (int)(j - 1)) >= t) j--;
    while (j < tab_n - 1 && linspace_at(end_param, tab_n, (int)j) < t) j++;
    return (int)j;
}
__device__ __forceinline__ int table_index(double t, int tab_n, double end_param)
{
    const int j = table_search(t, tab_n, end_param);
    if (j == 0) return 0;
    const double frac = t - floor(t);  // t % 1 for t >= 0
    return (int)(frac > 0.5 ? j - 1 : j);
}

// Fast form of the same index: with x = t * (1/step), ceil(x) is the searchsorted-left position
// unless x sits within 1e-8 of an integer (x is within ~1e-9 of t/step and of the NumPy-rounded
// table parameters), and the "t % 1 > 0.5" decision is safe unless t is within 1e-11 of an integer
// or of a segment midpoint.  In those rare cases `near` is set and the caller takes the exact path
// (reference rounding of t, table_index above), so the selected entry always equals the reference's.
__device__ __forceinline__ int table_index_fast(double t, int tab_n, double inv_step, bool &near)
{
    const double x = t * inv_step;
    const double cx = ceil(x);
    int j = (int)cx;
    j = j > tab_n - 1 ? tab_n - 1 : j;
    const double dxi = cx - x;                 // in [0,1): distance (in entries) up to the chosen entry
    const double frac = t - floor(t);          // t % 1 for t >= 0
    const double fh = fabs(frac - 0.5);
    near = dxi < 1e-8 || dxi > 1.0 - 1e-8 || fh < 1e-11 || fh > 0.5 - 1e-11;
    return (j > 0 && frac > 0.5) ? j - 1 : j;
}

// QHS:506-541 for parameters already inside [0, len(nodes)-1]: segment = min(int(t), G-1).  (When
// parameters[-1] rounds one ulp below G the reference's clamp moves the end sample's local parameter
// from 1 to 1-2^-53; that is the only difference.)
__device__ __forceinline__ void normalize_inside(double t, int G, double &lt, int &idx)
{
    int i = (int)t;
    i = i > G - 1 ? G - 1 : i;
    lt = t - (double)i;
    idx = i;
}

// atan2 for fp32 headings: octant reduction + the classic degree-4 (in z^2) minimax for |z| <= tan(pi/8),
// branch-free; absolute error ~1.5e-7, relative error ~2e-7 for small angles (what |dtheta| needs).
__device__ __forceinline__ float atan2_f32(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a = mn * __builtin_amdgcn_rcpf(mx);
    a = mx == 0.0f ? 0.0f : a;
    const bool big = a > 0.41421356237f;
    const float z = big ? (a - 1.0f) * __builtin_amdgcn_rcpf(a + 1.0f) : a;
    const float z2 = z * z;
    const float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z2, -1.38776856032e-1f), z2, 1.99777106478e-1f), z2,
                         -3.33329491539e-1f);
    float r = fmaf(p * z2, z, z);
    r = big ? r + 0.78539816339744831f : r;
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = x < 0.0f ? 3.14159265358979324f - r : r;
    return copysignf(r, y);
}

// |heading[k+1]-heading[k]| from the two fp64 derivative vectors and the two fp32 headings: the angle
// between the vectors (small-angle accurate) plus the 2*pi multiple that the raw difference of the
// reference's un-unwrapped atan2 values carries.  Consecutive samples are almost always within the
// series' range (|tan| < 0.41, same half-plane); the general atan2 handles the rest.
__device__ __forceinline__ float dtheta_f32(double ax, double ay, double bx, double by, float tha, float thb)
{
    const float cr = (float)fma(ax, by, -(ay * bx));
    const float dt = (float)fma(ax, bx, ay * by);
    float dl;
    const float z = cr * __builtin_amdgcn_rcpf(dt);
    if (dt > 0.0f && fabsf(z) < 0.41421356237f) {
        const float z2 = z * z;
        const float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z2, -1.38776856032e-1f), z2, 1.99777106478e-1f), z2,
                             -3.33329491539e-1f);
        dl = fmaf(p * z2, z, z);
    } else {
        dl = atan2_f32(cr, dt);
    }
    const float n = rintf(((thb - tha) - dl) * 0.15915494309189535f);
    return fabsf(fmaf(n, 6.283185307179586f, dl));
}

// The same difference in fp64 for the fp64 recurrence behind fp32 outputs ("strict" VAP_F32):
// the angle between the two fp64 derivative vectors from atan's series in z = cross/dot (|z| < 0.06 for any
// pair of neighbouring table entries of a smooth path: truncation < 1e-17), plus the 2*pi multiple of the raw
// headings.  Relative error ~1e-16, i.e. tighter than the reference's own difference of two rounded atan2 values.
// (the general atan2 is the rare branch of dtheta_f64: kept out of line so that its register needs do not weigh on
// the sampling kernel's straight path)
__device__ __attribute__((noinline)) double atan2_out_of_line(double y, double x) { return atan2(y, x); }

__device__ __forceinline__ double dtheta_f64(double ax, double ay, double bx, double by, float tha, float thb)
{
    const double cr = fma(ax, by, -(ay * bx));
    const double dt = fma(ax, bx, ay * by);
    double dl;
    double r = __builtin_amdgcn_rcp(dt);
    r = fma(r, fma(-dt, r, 1.0), r);
    r = fma(r, fma(-dt, r, 1.0), r);
    const double z = cr * r;
    if (dt > 0.0 && fabs(z) < 0.06) {
        const double z2 = z * z;
        // z - z^3/3 + z^5/5 - z^7/7 + z^9/9 - z^11/11
        const double p = fma(fma(fma(fma(fma(-1.0 / 11.0, z2, 1.0 / 9.0), z2, -1.0 / 7.0), z2, 1.0 / 5.0), z2, -1.0 / 3.0), z2, 1.0);
        dl = z * p;
    } else {
        dl = atan2_out_of_line(cr, dt);
    }
    const double n = rint(((double)(thb - tha) - dl) * 0.15915494309189535);
    return fabs(fma(n, 6.283185307179586, dl));
}

// dtheta_f64 in two halves, for callers that evaluate several samples side by side without a branch per sample: the
// series part for every sample (`general` says whether the sample is outside its range and needs dtheta_f64_general),
// then the 2*pi multiple.  The same operations in the same order as dtheta_f64.
__device__ __forceinline__ double dtheta_f64_series(double ax, double ay, double bx, double by, bool &general)
{
    const double cr = fma(ax, by, -(ay * bx));
    const double dt = fma(ax, bx, ay * by);
    double r = __builtin_amdgcn_rcp(dt);
    r = fma(r, fma(-dt, r, 1.0), r);
    r = fma(r, fma(-dt, r, 1.0), r);
    const double z = cr * r;
    general = !(dt > 0.0 && fabs(z) < 0.06);
    const double z2 = z * z;
    const double p = fma(fma(fma(fma(fma(-1.0 / 11.0, z2, 1.0 / 9.0), z2, -1.0 / 7.0), z2, 1.0 / 5.0), z2, -1.0 / 3.0), z2, 1.0);
    return z * p;
}
__device__ __forceinline__ double dtheta_f64_general(double ax, double ay, double bx, double by)
{
    return atan2_out_of_line(fma(ax, by, -(ay * bx)), fma(ax, bx, ay * by));
}
__device__ __forceinline__ double dtheta_f64_wrap(double dl, float tha, float thb)
{
    const double n = rint(((double)(thb - tha) - dl) * 0.15915494309189535);
    return fabs(fma(n, 6.283185307179586, dl));
}

// Python's min(a, b): keeps a unless b < a (a NaN in b is skipped).
template <typename R>
__device__ __forceinline__ R pymin(R a, R b) { return b < a ? b : a; }

template <typename R>
struct VelConsts {
    R vmax, amax, adec, tw;
    R wmax;    // max_angular_vel   = 2*max_vel/track_width   (MPG:81)
    R almax;   // max_angular_accel = 2*max_acc/track_width   (MPG:82)
};

// Per-sample max_acceleration rows for routes whose nodes / action points change it (all NULL: the constraints'):
//   fwd [B][S]  max_acc (= max_dec) in force for the forward step FROM sample i     MPG:194-196
//   bwd [B][S]  max_acc the backward sweep has in force for its step FROM sample i  MPG:256-257
//   dec [B]     max_dec of the whole backward sweep (what the forward sweep left)
template <typename R>
struct AccRows {
    const R *fwd = nullptr;
    const R *bwd = nullptr;
    const R *dec = nullptr;
};

// Curvature-only limits of one sample (MPG:204-233 / 264-293), in squared-velocity space.
template <typename R>
struct SampleLimits {
    R q;       // kappa^2
    R cap_u;   // min(max_linear_vel, max_vel/(1+tw*|k|/2))^2   (MPG:218-220, 243-249)
    R a_acc;   // min(max_accel_ang, max_accel_kin, max_acc)    (MPG:222-233 without the wheel term)
    R a_dec;   // same with max_dec                             (MPG:283-293)
    bool straight;
};

template <typename R>
__device__ __forceinline__ SampleLimits<R> sample_limits(const VelConsts<R> &c, R kabs, R cur_acc, R cur_dec)
{
    SampleLimits<R> o;
    o.q = kabs * kabs;
    o.straight = kabs < (R)1e-6;
    const R vtw = fabs(c.vmax / ((R)1 + (c.tw * kabs / (R)2)));
    if (o.straight) {
        o.cap_u = pymin(c.vmax, vtw);
        o.cap_u = o.cap_u * o.cap_u;
        o.a_acc = cur_acc;
        o.a_dec = cur_dec;
    } else {
        const R max_vel_ang = c.wmax / kabs;
        const R max_vel_kin = (R)2 * c.vmax / (c.tw * kabs + (R)2);
        // MPG:23-33 max_speed_at_curvature
        R mts = (((R)2 * c.vmax / c.tw) * c.vmax) / (kabs * c.vmax + ((R)2 * c.vmax / c.tw));
        mts = pymin(mts, c.vmax);
        R vl = pymin(pymin(max_vel_ang, max_vel_kin), mts);
        vl = pymin(vl, vtw);
        o.cap_u = vl * vl;
        const R a_ang = c.almax / kabs;
        o.a_acc = pymin(pymin(a_ang, (R)2 * cur_acc / (c.tw * kabs + (R)2)), cur_acc);
        o.a_dec = pymin(pymin(a_ang, (R)2 * cur_dec / (c.tw * kabs + (R)2)), cur_dec);
    }
    return o;
}

// One forward step i -> i+1 (MPG:193-249) in u = v^2 space.
//   u      : u_i (already final for the forward pass)
//   wprev  : (v_{i-1}|k_{i-1}|)^2, updated to (v_i|k_i|)^2
//   dth    : |heading[i+1]-heading[i]|
//   u_next : current content of u_{i+1} (the initial velocity cap squared)
template <typename R>
__device__ __forceinline__ R forward_step(const VelConsts<R> &c, const SampleLimits<R> &L, R cur_acc, R twodd,
                                          R u, R &wprev, R dth, R u_next)
{
    const R w = u * L.q;
    R a;
    if (L.straight) {
        a = L.a_acc;
    } else {
        const R accel_ang = (w - wprev) / ((R)2 * dth);
        // MPG:52-59 max_accels_at_turn(abs(accel_ang)); MPG:228-229 clamp at 0
        const R x = fabs(accel_ang) * c.tw / (R)2;
        const R left = cur_acc + x, right = cur_acc - x;
        R aw = fabs(left) < fabs(right) ? left : right;
        if (aw < (R)0) aw = (R)0;
        a = pymin(pymin(L.a_acc, aw), cur_acc);
    }
    wprev = w;
    const R nu = u + twodd * a;
    return pymin(pymin(u_next, nu), L.cap_u);
}

// One backward step i -> i-1 (MPG:255-311).  dth = |heading[i-1]-heading[i]|, u_prev = u_{i-1}.
template <typename R>
__device__ __forceinline__ R backward_step(const VelConsts<R> &c, const SampleLimits<R> &L, R cur_acc, R twodd,
                                           R u, R &wprev, R dth, R u_prev)
{
    const R w = u * L.q;
    R a;
    if (L.straight) {
        a = L.a_dec;
    } else {
        const R accel_ang = (w - wprev) / ((R)2 * dth);
        const R x = accel_ang * c.tw / (R)2;  // signed (MPG:288)
        const R left = cur_acc + x, right = cur_acc - x;
        R aw = fabs(left) < fabs(right) ? left : right;
        if (aw < (R)0) aw = (R)0;
        a = pymin(L.a_dec, aw);
    }
    wprev = w;
    const R nu = u + twodd * a;
    return pymin(pymin(nu, u_prev), L.cap_u);
}

// ------------------------------------------------------------------------------------------------
// "Fast" form of the velocity-pass step (what the production kernels use).
//
// In real arithmetic the reference's limits collapse (MPG:210-224, 23-33):
//   max_vel_kin = max_curve_vel = max_vel/(1+tw*k/2) =: vmax*r   and  max_vel_ang = 2vmax/(tw*k) > vmax*r
//   max_accel_kin = max_acc*r                                    and  max_accel_ang = 2amax/(tw*k) > amax*r
// with r = 1/(1 + tw*k/2), so   cap = (vmax*r)^2,  A = amax*r   (straight samples: A = amax exactly,
// MPG:204-206).  All accelerations are pre-multiplied by 2*dd ("p" suffix) so a step is
//   u' = min(u_init', cap, u + clamp(amaxp - |dw|*g, 0, Ap))
// (the offset form u + clamp(..) is also the reference's own order: v^2 + 2*a*dd with a clamped).
// Heading-difference zero (two samples on one table entry) makes the reference produce +-inf / NaN
// (SURVEY §8(a) "Edge semantics"); g is clamped to kHuge so the same decisions fall out without
// NaNs: forward  dw==0 -> A, dw!=0 -> 0;  backward  dw>0 -> 0, dw<=0 -> A.
// ------------------------------------------------------------------------------------------------
template <typename R> struct Huge;
template <> struct Huge<float> { static constexpr float v = 1e30f; };
template <> struct Huge<double> { static constexpr double v = 1e300; };

template <typename R>
struct FastConsts {
    R vmax;
    R amaxp;   // 2*dd*max_acc
    R adecp;   // 2*dd*max_dec
    R h;       // track_width/2
    R gk;      // 2*dd*track_width/4
    R aangp;   // 2*dd*max_angular_accel = 2*dd * 2*max_acc/track_width (MPG:82, the constraints' max_acc)
};

template <typename R>
__device__ __forceinline__ FastConsts<R> make_fast(const VelConsts<R> &c, R twodd)
{
    FastConsts<R> f;
    f.vmax = c.vmax;
    f.amaxp = twodd * c.amax;
    f.adecp = twodd * c.adec;
    f.h = c.tw / (R)2;
    f.gk = twodd * c.tw / (R)4;
    f.aangp = twodd * c.almax;
    return f;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// fp64 reciprocal for the step coefficients: the hardware estimate (v_rcp_f64, ~2^-26) and two Newton steps
// — five instructions, relative error ~2^-52 — instead of the IEEE division's fifteen.  Every velocity kernel
// derives its coefficients through this one function, so they stay bit-identical to each other; against the
// oracle the difference (one ulp of a coefficient) is far inside the fp64 bound.
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
// a / b for operands and quotient well inside the normal range (the arc-length table's interval slopes): the core of the
// IEEE division sequence — reciprocal estimate, two Newton steps, quotient, one residual correction — without its
// scaling and fix-up instructions (9 instead of 15).  The same quotient as `/` on such operands; every place that forms
// the interval slopes (SM:311-317) goes through this one function, so they agree with each other whatever the case.
__device__ __forceinline__ double div_inrange(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double vmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float vmax_(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double vmax_(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float clamp0(float x, float hi) { return __builtin_amdgcn_fmed3f(x, 0.0f, hi); }
__device__ __forceinline__ double clamp0(double x, double hi) { return fmin(fmax(x, 0.0), hi); }

// A wave-uniform double moved to scalar registers (v_readfirstlane of both halves): a VALU instruction reads it from
// there, and it stops competing with the per-sample arrays for vector registers.
__device__ __forceinline__ double uniform(double x)
{
    const uint64_t v = __builtin_bit_cast(uint64_t, x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// ... and kept there: the value passes through a scalar register operand of an (empty) asm statement, so the compiler
// cannot fold the move back into the vector arithmetic that produced it
__device__ __forceinline__ double uniform_pinned(double x)
{
    asm volatile("" : "+v"(x));   // (a value the compiler knows to be uniform would have its v_readfirstlane folded away)
    const uint64_t v = __builtin_bit_cast(uint64_t, x);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// Hides a value from code motion (loop-invariant hoisting, sinking into branches).
__device__ __forceinline__ float opaque(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double opaque(double x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

// Per-sample step coefficients.  With w_i = u_i*q_i (q = k^2, the squared angular velocity) the
// reference's dw = w_i - w_{i-1} is formed as  q_i*(u_i - rho_i*u_{i-1}),  rho_i = q_{i-1}/q_i, so the
// recurrence carries the last two squared velocities and a step is five dependent operations:
//   t = u - rho*u_prev;  y = amaxp - |t|*gq;  u' = min(u + clamp(y, 0, A), cap [, u_init'])
// with   gq  = g*q,  g = 2dd*tw/(4*dtheta)   (0 on straight samples, k < 1e-6, MPG:204-206)
//        A   = base_p*r,  cap = (vmax*r)^2,  r = 1/(1 + tw*k/2)   (straight samples: A = base_p)
//        rho = 0 on straight samples and on the first step of a sweep (MPG:190, 253: previous
//              angular velocity 0).
// A sample whose heading difference is zero has g clamped to kHuge; it is stored as gq = -kHuge
// (ordinary samples are kept in [0, kHuge/10]): the sign tells the backward step to be "sign-aware".
template <typename R>
__device__ __forceinline__ void fast_derive_k(const FastConsts<R> &c, R kabs, R kprev_abs, R base_p, R &rho, R &q,
                                              R &A, R &cap)
{
    q = kabs * kabs;
    const bool straight = kabs < (R)1e-6;
    const R r = fast_rcp((R)1 + c.h * kabs);
    const R vr = c.vmax * r;
    cap = vr * vr;
    A = straight ? base_p : base_p * r;
    // rho must be exactly 1 when both samples read the same table entry: then t = u - u_prev is an exact
    // zero for an unchanged velocity, the reference's 0/0 case (MPG:209 with dtheta == 0).
    // Straight-line selects (opaque keeps the reciprocal out of a branch): this runs 80 times per thread.
    const R qp = kprev_abs * kprev_abs;
    R x = opaque(qp * fast_rcp(q));
    x = qp == q ? (R)1 : x;
    rho = straight ? (R)0 : x;
}

// Routes whose nodes raise max_acceleration above the constraints' (per-sample amaxp): then max_accel_ang =
// max_angular_accel/|k| (MPG:222, 283 — fixed by the constraints' max_acc, MPG:82) can be the smaller limit, which
// it never is for the constraints' own value (2a/(tw k) > 2a/(2 + tw k)).  Folded into the clamp A.
template <typename R>
__device__ __forceinline__ R fast_cap_A(const FastConsts<R> &c, R kabs, R A)
{
    const R a = vmin(A, c.aangp * fast_rcp(kabs));
    return kabs < (R)1e-6 ? A : a;
}

template <typename R>
__device__ __forceinline__ R fast_gg(const FastConsts<R> &c, R dth) { return vmin(c.gk * fast_rcp(dth), Huge<R>::v); }

// q = k^2 of the same step (straight <=> q < 1e-12).  Branch-free: selects only.  A zero heading
// difference (gg clamped to kHuge) is returned as -kHuge: the magnitude is what both sweeps multiply
// by, the sign is the marker the backward sweep reads.
template <typename R>
__device__ __forceinline__ R fast_gq(R gg, R q)
{
    R r = vmin(gg * q, Huge<R>::v * (R)0.1);
    r = gg >= Huge<R>::v ? -Huge<R>::v : r;
    return q < (R)1e-12 ? (R)0 : r;
}

// The same two with kHuge handed in (a kernel that keeps it in scalar registers: as a 64-bit literal the compiler
// parks it in a vector register pair for the selects and, under pressure, spills and reloads it at every use).
template <typename R>
__device__ __forceinline__ R fast_gg(const FastConsts<R> &c, R dth, R hugev) { return vmin(c.gk * fast_rcp(dth), hugev); }
template <typename R>
__device__ __forceinline__ R fast_gq(R gg, R q, R hugev)
{
    R r = vmin(gg * q, Huge<R>::v * (R)0.1);
    r = gg >= hugev ? -hugev : r;
    return q < (R)1e-12 ? (R)0 : r;
}

template <typename R>
__device__ __forceinline__ void fast_derive(const FastConsts<R> &c, R kabs, R kprev_abs, R dth, R base_p, R &rho, R &gq,
                                            R &A, R &cap)
{
    R q;
    fast_derive_k(c, kabs, kprev_abs, base_p, rho, q, A, cap);
    gq = fast_gq(fast_gg(c, dth), q);
}

// Coefficients of a slot that holds no step (past the end of the path, or the fixed end sample in
// the backward sweep): the huge limits let the step return min(u + amaxp, u_init) = u_init, and the
// first real step after it has rho = 0 — i.e. walking through it restarts the chain.
template <typename R>
__device__ __forceinline__ void idle_coef(R &rho, R &gq, R &A, R &cap)
{
    rho = (R)0;
    gq = (R)0;
    A = Huge<R>::v;
    cap = Huge<R>::v;
}

// Velocity from its square, the last operation of every velocity kernel (MPG:316 returns v, the kernels
// carry u = v^2).  fp32: the hardware square root (v_sqrt_f32, 1 ulp) instead of the correctly rounded
// sequence — one instruction instead of ten, 40 times per thread; all kernels use this one function (either
// precision), so they stay bit-identical to each other.
__device__ __forceinline__ float vel_sqrt(float u) { return __builtin_amdgcn_sqrtf(u); }
// fp64: the hardware estimate of 1/sqrt(u) (v_rsq_f64), one coupled Newton step on (sqrt, 1/(2 sqrt)) and a final
// residual correction — eight instructions, within an ulp of the correctly rounded root (the library sqrt is ~22
// instructions with scaling for denormals, which a squared velocity never is).  u = 0 (a zero start / end velocity)
// gives 0.
__device__ __forceinline__ double vel_sqrt(double u)
{
    const double y = __builtin_amdgcn_rsq(u);
    double s = u * y, h = 0.5 * y;
    const double r = fma(-h, s, 0.5);
    s = fma(s, r, s);
    h = fma(h, r, h);
    const double d = fma(-s, s, u);
    s = fma(d, h, s);
    return u > 0.0 ? s : 0.0;
}

__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ double med3(double a, double lo, double hi) { return fmin(fmax(a, lo), hi); }

// Forward step, MPG:193-249.  A zero heading difference (|gq| = kHuge) gives the reference's
// dw == 0 -> A, dw != 0 -> 0 without a special case: t is exactly zero when the table entry and the
// velocity are unchanged (rho = 1), and any other t times kHuge exceeds amaxp.
// fp32:  am = 2dd*max_acc of the step, g = gq:   u' = min(u + med3(am - |t|*|g|, 0, A), cap, u_next)   (5 instructions)
// fp64:  the same step scaled by 1/A, so that the clamp to [0, A] becomes the free clamp-to-[0,1] output modifier of
//        the FMA (there is no v_med3_f64, and fmin / fmax cost a canonicalising v_max each on top — nine instructions
//        for the unscaled form):  am = 2dd*max_acc/A, g = gq/A (fast_scale below; the sign marker of g survives),
//            c = clamp01(am - |t|*|g|);   u' = min(A*c + u, cap [, u_next])                              (4 instructions)
//        c is exactly 0 / exactly 1 where the unscaled form clamps, so the clamped branches are bit-identical to it;
//        in between the two differ by one rounding of the increment (~1e-16 of u).
__device__ __forceinline__ double fma_nabs_clamp(double t, double g, double a)   // clamp01(a - |t|*|g|)
{
    double c;
    asm("v_fma_f64 %0, -|%1|, |%2|, %3 clamp" : "=v"(c) : "v"(t), "v"(g), "v"(a));
    return c;
}
__device__ __forceinline__ double fma_n_clamp(double p, double g, double a)      // clamp01(a - p*|g|)
{
    double c;
    asm("v_fma_f64 %0, -%1, |%2|, %3 clamp" : "=v"(c) : "v"(p), "v"(g), "v"(a));
    return c;
}
// v_min_f64 / v_max_f64 as they are: fmin() / fmax() put a canonicalising v_max in front of every operand the
// compiler cannot prove quiet (anything that went through opaque() or memory), doubling their cost
__device__ __forceinline__ double min_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// what the step functions take as (am, g) for a slot with acceleration budget amaxp, wheel term gq and clamp A
__device__ __forceinline__ void fast_scale(float amaxp, float gq, float A, float &am, float &g)
{
    (void)A;
    am = amaxp;
    g = gq;
}
__device__ __forceinline__ void fast_scale(double amaxp, double gq, double A, double &am, double &g)
{
    const double inv = fast_rcp(A);
    am = amaxp * inv;
    g = gq * inv;      // (A > 0: a negative gq, the zero-heading-difference marker, stays negative)
}

__device__ __forceinline__ float fast_forward_a(float amaxp, float rho, float gq, float A, float cap, float u, float &uprev, float u_next)
{
    const float t = fma(-rho, uprev, u);
    const float y = fma(-fabs(t), fabs(gq), amaxp);
    uprev = u;
    return vmin(vmin(u + med3(y, 0.0f, A), cap), u_next);
}
// u_next = +huge (no initial-velocity row): pass HAS_NEXT = false and the last min disappears
// (the four instructions of a step as ONE asm statement: between separate ones hipcc puts an s_nop, which costs a
// wave on a dependent chain as much as half an instruction)
__device__ __forceinline__ double step4(double am, double rho, double g, double A, double cap, double u, double uprev)
{
    double r;
    asm("v_fma_f64 %0, -%2, %3, %1\n\t"
        "v_fma_f64 %0, -|%0|, |%4|, %5 clamp\n\t"
        "v_fma_f64 %0, %6, %0, %1\n\t"
        "v_min_f64 %0, %0, %7"
        : "=&v"(r)
        : "v"(u), "v"(rho), "v"(uprev), "v"(g), "v"(am), "v"(A), "v"(cap));
    return r;
}
template <bool HAS_NEXT = true>
__device__ __forceinline__ double fast_forward_a(double am, double rho, double g, double A, double cap, double u, double &uprev, double u_next)
{
    const double r = step4(am, rho, g, A, cap, u, uprev);
    uprev = u;
    if constexpr (HAS_NEXT) return min_raw(r, u_next);
    else return r;
}

// Backward step, MPG:255-311.  DUP = the path has samples with a zero heading difference (gq < 0): the
// reference's signed +-inf handling there differs from the forward one (dw > 0 -> 0, dw <= 0 -> A), i.e.
// the penalty is max(t, 0)*kHuge where an ordinary sample has |t|*gq = max(t, -t)*gq.
// (amaxp as an argument: routes whose nodes change max_acceleration carry it per sample, MPG:194-196, 256-257.
// There the two limits differ — the wheel limit y comes from max_acc, the clamp A from max_dec — and a straight
// sample, whose limit is max_dec alone (MPG:270-272), is given amaxp = A so that the clamp decides.)
template <bool DUP>
__device__ __forceinline__ float fast_backward_a(float amaxp, float rho, float gq, float A, float cap, float u, float &uprev, float u_prev)
{
    const float t = fma(-rho, uprev, u);
    float y;
    if constexpr (DUP) {
        gq = opaque(gq);   // keep the select below inside the round loop (one register per sample otherwise)
        const float other = gq < 0.0f ? 0.0f : -t;
        y = fma(-vmax_(t, other), fabs(gq), amaxp);
    } else {
        y = fma(-fabs(t), gq, amaxp);
    }
    uprev = u;
    return vmin(vmin(u + med3(y, 0.0f, A), cap), u_prev);
}
// HAS_PREV = false: the caller has folded the forward value of the sample into cap (min(cap, u_prev)) already
template <bool DUP, bool HAS_PREV = true>
__device__ __forceinline__ double fast_backward_a(double am, double rho, double g, double A, double cap, double u, double &uprev, double u_prev)
{
    double r;
    if constexpr (DUP) {
        const double t = fma(-rho, uprev, u);
        g = opaque(g);
        const double other = g < 0.0 ? 0.0 : -t;
        const double c = fma_n_clamp(max_raw(t, other), g, am);
        r = min_raw(fma(A, c, u), cap);
    } else {
        r = step4(am, rho, g, A, cap, u, uprev);
    }
    uprev = u;
    if constexpr (HAS_PREV) return min_raw(r, u_prev);
    else return r;
}

// The step functions as the kernels call them.  kScaledStep<R>: the arithmetic type takes the scaled form, so every
// kernel keeps the per-sample `am` of fast_scale next to g (fp32 kernels only when nodes change max_acceleration).
template <typename R> struct ScaledStep { static constexpr bool value = false; };
template <> struct ScaledStep<double> { static constexpr bool value = true; };
__device__ __forceinline__ float step_fwd(float am, float rho, float g, float A, float cap, float u, float &uprev)
{
    return fast_forward_a(am, rho, g, A, cap, u, uprev, Huge<float>::v);
}
__device__ __forceinline__ double step_fwd(double am, double rho, double g, double A, double cap, double u, double &uprev)
{
    return fast_forward_a<false>(am, rho, g, A, cap, u, uprev, 0.0);
}

// Cooperative copy of n doubles from HBM/L2 into LDS: up to ITER loads per thread are issued before
// the first LDS write, so a workgroup pays the memory latency once instead of once per element.
template <int ITER>
__device__ __forceinline__ void lds_fill(double *__restrict__ dst, const double *__restrict__ src, int n, int tid, int nt)
{
    for (int base = 0; base < n; base += ITER * nt) {
        double v[ITER];
#pragma unroll
        for (int it = 0; it < ITER; it++) {
            const int i = base + tid + it * nt;
            v[it] = i < n ? src[i] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < ITER; it++) {
            const int i = base + tid + it * nt;
            if (i < n) dst[i] = v[it];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The distance grid of forward_backward_pass (MPG:112-122): the reference ACCUMULATES,
//     current_dist = 0;  while current_dist < total: ...; current_dist += delta_dist
// so sample k sits at s_k = fl(s_{k-1} + dd), which is not fl(k*dd): the two drift apart by up to
// k*ulp(s)/2, enough to move a sample across a table-entry boundary now and then (~3e-7 of the samples
// at 10^4 samples per path, more on long rows).  The sum has a closed form inside a binade
// [2^e, 2^(e+1)): every s there is a multiple of ulp_e = 2^(e-52), so fl(s + dd) = s + m*ulp_e with
// m = round(dd/ulp_e) — constant from the second element of the binade on (a tie rounds to even; the
// first result is even, after which the choice no longer depends on s).  A path's grid is therefore at
// most two arithmetic runs per binade; GridRuns lists them and grid_s() evaluates s_k exactly:
//   run r covers k in [k0[r], k0[r+1]):  s_k = s0[r] + (k - k0[r]) * D[r]     (exact in fp64)
// ------------------------------------------------------------------------------------------------
constexpr int kMaxGridRuns = 96;   // two per binade from dd to the path length: 2^-25 ft steps on a 2^22 ft path
constexpr int kGridRunStride = kMaxGridRuns + 4;      // + end markers, so a reader may load three entries ahead
constexpr int kGridRunDoubles = 3 * kGridRunStride;   // per path: 100 entries of {k0 (int64 bits), s0, D}

__host__ __device__ inline long grid_run_k0(const double *tab, int r)
{
    long v;
    __builtin_memcpy(&v, tab + 3 * r, sizeof v);
    return v;
}
__host__ __device__ inline void grid_run_set_k0(double *tab, int r, long k) { __builtin_memcpy(tab + 3 * r, &k, sizeof k); }

// 2^e and floor(log2(x)) for normal doubles, by their bits (ldexp / ilogb cost ~40 instructions each on the
// device); out-of-range exponents take the library route
__host__ __device__ inline double grid_pow2(int e)
{
    if (e < -1000 || e > 1000) return ldexp(1.0, e);
    const unsigned long long b = (unsigned long long)(e + 1023) << 52;
    double v;
    __builtin_memcpy(&v, &b, sizeof v);
    return v;
}
__host__ __device__ inline int grid_exponent(double x)
{
    unsigned long long b;
    __builtin_memcpy(&b, &x, sizeof b);
    const int be = (int)((b >> 52) & 0x7ff);
    return (be == 0 || be == 0x7ff) ? ilogb(x) : be - 1023;
}
// a reciprocal good enough to seed a floor() that is corrected afterwards
__host__ __device__ inline double grid_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rcp(x);
    return r * (2.0 - x * r);           // one Newton step: the corrections after floor() stay at a step or two
#else
    return 1.0 / x;
#endif
}

// Builds the runs for indices 0 .. k_limit (inclusive) or until s_k >= total, whichever comes first, into
// tab (kGridRunStride entries of {first index k0 — an int64 stored in the double's bits —, its distance s0,
// increment D}) and returns the number of indices k with s_k < total among
// those built (the reference's loop count; the appended end sample is not included).  dd > 0, total > 0.
__host__ __device__ inline long build_grid_runs(double dd, double total, long k_limit, double *__restrict__ tab,
                                                int &n_runs)
{
    int R = 0;
    long k = 0;
    double s = 0.0;
    long n_below = 0;            // indices with s_k < total found so far
    auto emit = [&](long kk, double ss, double dd_run) {
        grid_run_set_k0(tab, R, kk);
        tab[3 * R + 1] = ss;
        tab[3 * R + 2] = dd_run;
        R++;
    };
    // k = 0: s = 0, then s_1 = fl(0 + dd) = dd
    emit(0, 0.0, dd);
    n_below = 1;                 // s_0 = 0 < total
    k = 1;
    s = dd;
    while (k <= k_limit && s < total && R < kMaxGridRuns - 2) {
        const int e = grid_exponent(s);                   // s in [2^e, 2^(e+1))
        const double top = grid_pow2(e + 1);
        const double up = grid_pow2(52 - e);              // 1 / ulp
        const double ulp = grid_pow2(e - 52);
        const double s1 = s + dd;                         // hardware rounding, whichever binade it lands in
        emit(k, s, s1 - s);                               // the binade's first element: a run of one
        n_below = k + 1;
        if (s1 >= top) {                                  // the next element already left the binade:
            emit(k + 1, s1, 0.0);                         // an empty second run keeps two runs per binade
            k += 1;
            s = s1;
            continue;
        }
        // constant increment from s1 on: dd in ulps (exact power-of-two scaling), rounded half to even —
        // every partial sum from s1 on has an even last place when dd/ulp ends in one half
        const double m = rint(dd * up);
        const double Dc = m * ulp;
        // elements s1 + j*Dc, j = 0..J, stay below top: J = floor(((top - s1)/ulp - 1) / m)
        double J = 0.0;
        if (m > 0.0) {
            const double a = (top - s1) * up - 1.0;       // integer-valued, exact
            J = floor(a * grid_rcp(m));
            while ((J + 1.0) * m <= a) J += 1.0;
            while (J > 0.0 && J * m > a) J -= 1.0;
        } else {
            J = (double)(k_limit - k);                    // dd below half an ulp: s no longer moves
        }
        long Jl = (J > 4.0e15) ? (long)4e15 : (long)J;
        if (k + 1 + Jl > k_limit) Jl = k_limit - (k + 1) < 0 ? 0 : k_limit - (k + 1);
        emit(k + 1, s1, Dc);
        if (s1 >= total) break;
        const double s_last = s1 + (double)Jl * Dc;       // exact
        if (s_last >= total) {
            // the grid ends inside this run: the first j with s1 + j*Dc >= total (Dc > 0 here)
            double q = floor((total - s1) * grid_rcp(Dc));
            while (s1 + q * Dc < total) q += 1.0;
            while (q > 0.0 && s1 + (q - 1.0) * Dc >= total) q -= 1.0;
            n_below = k + 1 + (long)q;
            break;
        }
        n_below = k + 1 + Jl + 1;
        k = k + 1 + Jl + 1;
        s = s_last + dd;                                  // crossing step: hardware rounding in the next binade
    }
    for (int j = 0; j < 4; j++) {                         // the last run extends as far as anyone asks
        grid_run_set_k0(tab, R + j, k_limit + 2);
        tab[3 * (R + j) + 1] = 0.0;
        tab[3 * (R + j) + 2] = 0.0;
    }
    n_runs = R;
    return n_below;
}

// s_k from the runs (k within what build_grid_runs covered); r is a hint / cursor: the run of the previous
// lookup, moved forward or back as needed.
__host__ __device__ inline double grid_s(const double *__restrict__ tab, int n_runs, long k, int &r)
{
    while (r + 1 < n_runs && k >= grid_run_k0(tab, r + 1)) r++;
    while (r > 0 && k < grid_run_k0(tab, r)) r--;
    return tab[3 * r + 1] + (double)(k - grid_run_k0(tab, r)) * tab[3 * r + 2];
}

// First guess of the run that holds sample k: run 0 is k = 0 and binade i above dd's (i = 0, 1, ...)
// owns runs 1 + 2i (its first element) and 2 + 2i (the arithmetic rest), so the exponent of k*dd lands
// within a run or two of the right one; grid_s() walks the rest.
__host__ __device__ inline int grid_run_hint(double dd, long k, int n_runs)
{
    if (k <= 0) return 0;
    int r = 2 + 2 * (ilogb((double)k * dd) - ilogb(dd));
    r = r < 1 ? 1 : r;
    return r > n_runs - 1 ? n_runs - 1 : r;
}

// ------------------------------------------------------------------------------------------------
// Scalar helpers of the time-domain resample (MPG:389-628), shared by the single-route kernel
// (vap_route.hip) and the batched one (vap_time.hip).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double mod1(double v) { return fmod(v, 1.0); }

__device__ __forceinline__ double py_mod(double a, double b)   // Python float % for b > 0
{
    double m = fmod(a, b);
    if (m != 0.0) { if (m < 0) m += b; }
    else m = copysign(0.0, b);
    return m;
}

// np.searchsorted(xs, x, side="right") - 1 over xs[i] = i*dd, i < n: the largest i with i*dd <= x
// (-1 if none).  The predicate is monotone in i, so an estimate from x * (1/dd) is corrected by at most
// a step or two instead of bisecting; the result is the bisection's.
__device__ __forceinline__ int grid_index(double x, double dd, double inv_dd, int n)
{
    double e = floor(x * inv_dd);
    if (!(e >= -1.0)) e = -1.0;            // also NaN
    if (e > (double)(n - 1)) e = (double)(n - 1);
    int i = (int)e;
    // The answer is the i in [-1, n-1] with i*dd <= x < (i+1)*dd (products as rounded).  x*inv_dd is within 2^-52 of
    // x/dd relatively, so the floor is off by at most one: one step up, one step down, selects only — the two
    // `while` loops this replaces compiled to ~50 instructions each and were most of a time step of k_time_integrate.
    // The loops remain behind a check of the two conditions (never taken for finite input).
    // (bitwise & and |: the products are evaluated unconditionally, no short-circuit branches)
    const bool up = (i + 1 < n) & ((double)(i + 1) * dd <= x);
    i += up ? 1 : 0;
    const bool down = (i >= 0) & !((double)i * dd <= x);
    i -= down ? 1 : 0;
    const bool ok = ((i + 1 >= n) | !((double)(i + 1) * dd <= x)) & ((i < 0) | ((double)i * dd <= x));
    if (__builtin_expect(!ok, 0)) {
        while (i + 1 < n && (double)(i + 1) * dd <= x) i++;
        while (i >= 0 && !((double)i * dd <= x)) i--;
    }
    return i;
}

// grid_index from a guess (the time loop: the position one grid step ahead lands one sample further) — the same
// answer by the same two conditions, without the multiplication / floor / clamps of the general entry.
__device__ __forceinline__ int grid_index_from(double x, double dd, double inv_dd, int n, int guess)
{
    int i = guess < -1 ? -1 : (guess > n - 1 ? n - 1 : guess);
    const bool up = (i + 1 < n) & ((double)(i + 1) * dd <= x);
    i += up ? 1 : 0;
    const bool down = (i >= 0) & !((double)i * dd <= x);
    i -= down ? 1 : 0;
    const bool ok = ((i + 1 >= n) | !((double)(i + 1) * dd <= x)) & ((i < 0) | ((double)i * dd <= x));
    if (__builtin_expect(!ok, 0)) return grid_index(x, dd, inv_dd, n);
    return i;
}

// MPG:349-386 lerp over x_array[i] = i*dd, given idx = grid_index(x) and the two samples it selects
// (y0 = ys[clamp(idx)], y1 = ys[clamp(idx + 1)]).
__device__ __forceinline__ double lerp_at(double x, double dd, int idx, int n, double y0, double y1)
{
    if (idx < 0 || idx >= n - 1) return y0;   // below the grid: ys[0]; at or past its end: ys[n-1]
    const double x0 = (double)idx * dd, x1 = (double)(idx + 1) * dd;
    return y0 + (x - x0) * (y1 - y0) / (x1 - x0);
}

__device__ __forceinline__ int clamp_index(int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }

template <typename T>
__device__ __forceinline__ double lerp_grid(double x, double dd, double inv_dd, const T *__restrict__ ys, int n)
{
    const int idx = grid_index(x, dd, inv_dd, n);
    return lerp_at(x, dd, idx, n, (double)ys[clamp_index(idx, n)], (double)ys[clamp_index(idx + 1, n)]);
}

__device__ __forceinline__ double clip(double x, double lo, double hi)
{
    const double m = x < lo ? lo : x;
    return m > hi ? hi : m;
}

// ------------------------------------------------------------------------------------------------
// Grid definition of one path (MPG:112-122 sample count, or this build's fixed-S grid) and the evaluations the
// sampling kernels share.
// ------------------------------------------------------------------------------------------------
__device__ inline void grid_define(int b, int W, int S, double dd_in, double total, double t_max, double *__restrict__ meta,
                            double *__restrict__ aux, double *__restrict__ runs, uint32_t *__restrict__ flags)
{
    double *tab = runs + (size_t)b * kGridRunDoubles;
    double dd, n;
    int n_runs = 1;
    const bool usable = total > 0.0 && isfinite(total);
    if (dd_in > 0) {
        dd = dd_in;
    } else {
        dd = total / ((double)S - 1.5);
    }
    long n_loop = 1;
    if (usable && dd > 0.0) {
        // the reference's accumulated grid (current_dist += dd, MPG:112-122), exactly: vap_device.h
        n_loop = build_grid_runs(dd, total, (long)S, tab, n_runs);
    } else {
        for (int j = 0; j < 5; j++) {     // one run that never moves, and the end markers
            grid_run_set_k0(tab, j, j == 0 ? 0 : (long)S + 2);
            tab[3 * j + 1] = 0.0;
            tab[3 * j + 2] = 0.0;
        }
    }
    if (dd_in > 0) {
        long N = n_loop + 1;  // + appended end sample, MPG:172-175
        if (N > S) {
            N = S;
            if (flags) atomicOr(&flags[b], 2u /* VAP_FLAG_TRUNCATED */);
        }
        n = (double)N;
    } else {
        n = (double)S;      // dd = total/(S-1.5): s_(S-2) < total <= s_(S-1) with half a step of margin
    }
    meta[(size_t)b * kMetaStride + 2] = dd;
    meta[(size_t)b * kMetaStride + 3] = n;
    const double tstep = (double)(W - 1) / (double)(W * kSamplesPerNode - 1);  // np.linspace step, SM:487
    aux[(size_t)b * kAuxStride + 0] = t_max / (double)(kLutN - 1);             // SM:443
    aux[(size_t)b * kAuxStride + 1] = tstep;
    aux[(size_t)b * kAuxStride + 2] = 1.0 / tstep;
    aux[(size_t)b * kAuxStride + 3] = (double)n_runs;
}


// the same for a route of several splines (the per-spline table steps live in its spline table)
__device__ inline void grid_define_route(int b, int W, int S, double dd_in, double total, double *__restrict__ meta,
                                         double *__restrict__ aux, double *__restrict__ runs, uint32_t *__restrict__ flags)
{
    grid_define(b, W, S, dd_in, total, meta[(size_t)b * kMetaStride + 0], meta, aux, runs, flags);
}

template <typename OT>
__device__ __forceinline__ OT heading_of(double dy, double dx);
template <>
__device__ __forceinline__ float heading_of<float>(double dy, double dx) { return atan2_f32((float)dy, (float)dx); }
template <>
__device__ __forceinline__ double heading_of<double>(double dy, double dx) { return atan2(dy, dx); }

// 1/sqrt(x)^3 * num without fp64 sqrt/div: hardware estimate + two Newton steps (~1e-16 relative)
__device__ __forceinline__ double curvature_of(double num, double ss)
{
    double r = __builtin_amdgcn_rsq(ss);
    r = r * fma(-0.5 * ss * r, r, 1.5);
    r = r * fma(-0.5 * ss * r, r, 1.5);
    return num * r * r * r;
}
template <typename OT>
__device__ __forceinline__ OT heading_of_r(double dy, double dx) { return heading_of<OT>(dy, dx); }
__device__ __forceinline__ double curvature_of_r(double num, double ss) { return curvature_of(num, ss); }

}  // namespace vap
