// vap_kernels.hip — HIP kernels (gfx950 / CDNA4) of the batched trajectory generator.
//
// Stage map: K1 fit -> K2 arc-length LUT -> K3+K4 sample -> K5 velocity pass.
// Data layout in HBM (B paths, W waypoints, G = W-1 segments, S sample capacity):
//   segments [B][G][6][2] fp64   reference row order (QHS:120-122)
//   power    [B][G][2][6] fp64   monomial coefficients c0..c5 of x then y (scratch)
//   lut      [B][1000]    fp64   cumulative trapezoid distances (SM:448-454)
//   meta     [B][4]       fp64   {param_last, total_length, dd, n_samples}
//   x,y,heading,curvature,dtheta,velocity [B][S]  fp32 or fp64, sample-major per path so that a
//   wavefront of consecutive samples reads/writes 256 (fp32) or 512 (fp64) contiguous bytes.
#include "vap_device.h"
#include "vap_kernels.h"

namespace vap {

// ------------------------------------------------------------------------------------------------
// K1: fit.  One workgroup per path.  QHS:30-138, 149-219, 719-736; SM:65-77 tangent overrides.
// LDS (dynamic): pts[W][2], dist[G], fd[W][2], sd[W][2]  (fp64)
// ------------------------------------------------------------------------------------------------
template <typename IT>
__global__ __launch_bounds__(256) void k_fit(int W, const IT *__restrict__ waypoints,
                                             const double *__restrict__ tan_in,
                                             const double *__restrict__ tan_out,
                                             double *__restrict__ segments, double *__restrict__ power,
                                             double *__restrict__ meta, uint32_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int G = W - 1;
    double *pts = sh;             // 2W
    double *dist = pts + 2 * W;   // G (W slots)
    double *fd = dist + W;        // 2W
    double *sd = fd + 2 * W;      // 2W
    __shared__ uint32_t s_flag;
    if (tid == 0) s_flag = 0;
    const IT *wp = waypoints + (size_t)b * W * 2;
    for (int i = tid; i < 2 * W; i += nt) pts[i] = (double)wp[i];
    __syncthreads();
    for (int i = tid; i < G; i += nt) {
        const double dx = pts[2 * (i + 1)] - pts[2 * i], dy = pts[2 * (i + 1) + 1] - pts[2 * i + 1];
        const double d = sqrt(dx * dx + dy * dy);  // np.linalg.norm(diffs, axis=1), QHS:160
        dist[i] = d;
        if (!(d > 0.0) || !isfinite(d)) atomicOr(&s_flag, VAP_FLAG_DEGENERATE_BIT);
    }
    __syncthreads();
    if (tid == 0) {
        // QHS:719-736: only parameters[-1] is ever read; sequential np.cumsum order
        double cum = 0.0;
        for (int i = 0; i < G; i++) cum += dist[i];
        const double t_max = (cum == 0.0) ? (double)G : cum * (double)G / cum;
        meta[(size_t)b * kMetaStride + 0] = t_max;
    }
    // QHS:163-195 first derivatives
    for (int i = tid; i < W; i += nt) {
        double fx, fy;
        if (i == 0) {
            fx = (pts[2] - pts[0]) / dist[0];
            fy = (pts[3] - pts[1]) / dist[0];
        } else if (i == W - 1) {
            fx = (pts[2 * i] - pts[2 * (i - 1)]) / dist[G - 1];
            fy = (pts[2 * i + 1] - pts[2 * (i - 1) + 1]) / dist[G - 1];
        } else {
            const double px = (pts[2 * i] - pts[2 * (i - 1)]) / dist[i - 1];
            const double py = (pts[2 * i + 1] - pts[2 * (i - 1) + 1]) / dist[i - 1];
            const double nx = (pts[2 * (i + 1)] - pts[2 * i]) / dist[i];
            const double ny = (pts[2 * (i + 1) + 1] - pts[2 * i + 1]) / dist[i];
            fx = (px + nx) / 2;
            fy = (py + ny) / 2;
        }
        fd[2 * i] = fx;
        fd[2 * i + 1] = fy;
    }
    __syncthreads();
    // QHS:197-219 second derivatives (zero at both ends)
    for (int i = tid; i < W; i += nt) {
        double sx = 0.0, sy = 0.0;
        if (i > 0 && i < W - 1) {
            const double avg = (dist[i - 1] + dist[i]) / 2;
            sx = (fd[2 * (i + 1)] - fd[2 * (i - 1)]) / (avg * 0.5);
            sy = (fd[2 * (i + 1) + 1] - fd[2 * (i - 1) + 1]) / (avg * 0.5);
        }
        sd[2 * i] = sx;
        sd[2 * i + 1] = sy;
    }
    __syncthreads();
    // QHS:76-127 segment assembly
    for (int i = tid; i < G; i += nt) {
        const double L = dist[i];  // == np.linalg.norm(p1 - p0)
        double r[12];
        r[0] = pts[2 * i];       r[1] = pts[2 * i + 1];
        r[2] = pts[2 * (i + 1)]; r[3] = pts[2 * (i + 1) + 1];
        if (L > 0) {
            const double L2 = L * L;
            r[4] = fd[2 * i] * L;          r[5] = fd[2 * i + 1] * L;
            r[6] = fd[2 * (i + 1)] * L;    r[7] = fd[2 * (i + 1) + 1] * L;
            r[8] = sd[2 * i] * L2;         r[9] = sd[2 * i + 1] * L2;
            r[10] = sd[2 * (i + 1)] * L2;  r[11] = sd[2 * (i + 1) + 1] * L2;
            if (tan_out) {
                const double *t = tan_out + ((size_t)b * W + i) * 2;
                if (!isnan(t[0])) { r[4] = t[0]; r[5] = t[1]; }
            }
            if (tan_in) {
                const double *t = tan_in + ((size_t)b * W + i + 1) * 2;
                if (!isnan(t[0])) { r[6] = t[0]; r[7] = t[1]; }
            }
        } else {
            r[4] = fd[2 * i];          r[5] = fd[2 * i + 1];
            r[6] = fd[2 * (i + 1)];    r[7] = fd[2 * (i + 1) + 1];
            r[8] = sd[2 * i];          r[9] = sd[2 * i + 1];
            r[10] = sd[2 * (i + 1)];   r[11] = sd[2 * (i + 1) + 1];
        }
        double *sg = segments + ((size_t)b * G + i) * 12;
#pragma unroll
        for (int k = 0; k < 12; k++) sg[k] = r[k];
        if (power) {
            double cx[6], cy[6];
            hermite_to_power(r[0], r[2], r[4], r[6], r[8], r[10], cx);
            hermite_to_power(r[1], r[3], r[5], r[7], r[9], r[11], cy);
            double *pw = power + ((size_t)b * G + i) * 12;
#pragma unroll
            for (int k = 0; k < 6; k++) { pw[k] = cx[k]; pw[6 + k] = cy[k]; }
        }
    }
    __syncthreads();
    if (tid == 0 && flags) flags[b] = s_flag;
}

// ------------------------------------------------------------------------------------------------
// K2: arc-length lookup table.  One workgroup per path.  SM:426-475.
// 1000 uniform-parameter samples of |P'(t)| (reference basis, reference order), trapezoid, and a
// SEQUENTIAL cumulative sum (np.cumsum order) so the table is bit-identical to the reference's.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lut(int W, const double *__restrict__ segments,
                                             double *__restrict__ lut, double *__restrict__ meta,
                                             uint32_t *__restrict__ flags)
{
    __shared__ double mag[kLutN];
    __shared__ double cum[kLutN];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int G = W - 1;
    const double t_max = meta[(size_t)b * kMetaStride + 0];
    const double *seg = segments + (size_t)b * G * 12;
    for (int j = tid; j < kLutN; j += nt) {
        const double t = linspace_at(t_max, kLutN, j);
        double lt;
        int idx;
        normalize_parameter(t, t_max, G, lt, idx);
        double dx, dy;
        hermite_d1_ref(seg + (size_t)idx * 12, lt, dx, dy);
        mag[j] = sqrt(dx * dx + dy * dy);  // np.linalg.norm(derivatives, axis=1), SM:448
    }
    __syncthreads();
    if (tid == 0) {
        const double dt = linspace_at(t_max, kLutN, 1) - linspace_at(t_max, kLutN, 0);  // SM:444
        double acc = 0.0;
        cum[0] = 0.0 + 0.0;
        for (int j = 1; j < kLutN; j++) {
            acc += (mag[j - 1] + mag[j]) * 0.5 * dt;  // SM:452-454
            cum[j] = acc + 0.0;                        // + current_dist (single spline), SM:457
        }
        const double total = cum[kLutN - 1];
        meta[(size_t)b * kMetaStride + 1] = total;
        if (flags && !(total > 0.0 && isfinite(total))) atomicOr(&flags[b], VAP_FLAG_DEGENERATE_BIT);
    }
    __syncthreads();
    for (int j = tid; j < kLutN; j += nt) lut[(size_t)b * kLutN + j] = cum[j];
}

// ------------------------------------------------------------------------------------------------
// Grid definition: one thread per path.  MPG:112-122 sample count, or this build's fixed-S grid.
// ------------------------------------------------------------------------------------------------
__global__ void k_grid(int B, int S, double dd_in, double *__restrict__ meta, uint32_t *__restrict__ flags)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double total = meta[(size_t)b * kMetaStride + 1];
    double dd, n;
    if (dd_in > 0) {
        dd = dd_in;
        // loop samples: k >= 0 with k*dd < total (the reference accumulates current_dist += dd)
        long nl = (long)ceil(total / dd);
        if (nl < 1) nl = 1;
        while (nl > 1 && (double)(nl - 1) * dd >= total) nl--;
        while ((double)nl * dd < total) nl++;
        long N = nl + 1;  // + appended end sample, MPG:172-175
        if (N > S) {
            N = S;
            if (flags) atomicOr(&flags[b], VAP_FLAG_TRUNCATED_BIT);
        }
        n = (double)N;
    } else {
        dd = total / ((double)S - 1.5);
        n = (double)S;
    }
    meta[(size_t)b * kMetaStride + 2] = dd;
    meta[(size_t)b * kMetaStride + 3] = n;
}

// ------------------------------------------------------------------------------------------------
// K3+K4: sampling.  grid = (tiles, B); a tile is kSampleTile consecutive samples of one path.
//   thread i evaluates sample k0+i (the last thread's sample is the next tile's first one: it only
//   supplies the neighbour needed for |dtheta|).
// LDS: the path's distance table (8 KB) + per-sample derivative/table index for the neighbour
// exchange.  Coefficients are read from HBM/L2: a wavefront's 64 consecutive samples touch one or
// two 96-byte segment blocks, so those loads are broadcasts.
// ------------------------------------------------------------------------------------------------
template <typename OT>
__device__ __forceinline__ OT heading_of(double dy, double dx);
template <>
__device__ __forceinline__ float heading_of<float>(double dy, double dx) { return atan2f((float)dy, (float)dx); }
template <>
__device__ __forceinline__ double heading_of<double>(double dy, double dx) { return atan2(dy, dx); }

template <typename OT>
__global__ __launch_bounds__(kSampleThreads) void k_sample(int W, int S, const double *__restrict__ power,
                                                           const double *__restrict__ lut,
                                                           const double *__restrict__ meta,
                                                           OT *__restrict__ ox, OT *__restrict__ oy,
                                                           OT *__restrict__ oh, OT *__restrict__ ok,
                                                           OT *__restrict__ odth)
{
    __shared__ double sD[kLutN];
    __shared__ double s_dx[kSampleThreads], s_dy[kSampleThreads];
    __shared__ int s_j[kSampleThreads];
    __shared__ OT s_th[kSampleThreads];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int G = W - 1;
    const double *m = meta + (size_t)b * kMetaStride;
    const double t_max = m[0], total = m[1], dd = m[2];
    const int N = (int)m[3];
    const int k0 = blockIdx.x * kSampleTile;
    if (k0 >= S) return;
    const size_t row = (size_t)b * S;
    if (k0 >= N) {
        // past this path's grid (ragged dd mode): zero-fill so every output element is defined
        const int k = k0 + tid;
        if (tid < kSampleTile && k < S) {
            if (ox) ox[row + k] = (OT)0;
            if (oy) oy[row + k] = (OT)0;
            if (oh) oh[row + k] = (OT)0;
            if (ok) ok[row + k] = (OT)0;
            if (odth) odth[row + k] = (OT)0;
        }
        return;
    }
    for (int j = tid; j < kLutN; j += kSampleThreads) sD[j] = lut[(size_t)b * kLutN + j];
    __syncthreads();

    const int k = k0 + tid;
    const double end_param = (double)(W - 1);
    const int tab_n = W * kSamplesPerNode;
    const double *pw = power + (size_t)b * G * 12;
    OT th = (OT)0, kap = (OT)0, px = (OT)0, py = (OT)0;
    double d1x = 0.0, d1y = 0.0;
    int jj = -1;
    if (k < N) {
        // MPG:112-122 distance grid; the reference accumulates s += dd, we form k*dd
        const double s = (k == N - 1) ? total : (double)k * dd;
        const double t = distance_to_time(sD, total, t_max, end_param, s);
        // SM:340-346 / 550-580: table entry selected by the step lookup, evaluated on demand
        jj = table_index(t, tab_n, end_param);
        const double tp = linspace_at(end_param, tab_n, jj);
        double lt;
        int sg;
        normalize_parameter(tp, t_max, G, lt, sg);
        const double *cx = pw + (size_t)sg * 12, *cy = cx + 6;
        d1x = poly_d1(cx, lt);
        d1y = poly_d1(cy, lt);
        const double d2x = poly_d2(cx, lt), d2y = poly_d2(cy, lt);
        const double ss = d1x * d1x + d1y * d1y;                       // SM:517
        const double num = d1x * d2y - d1y * d2x;                      // SM:523
        const double kd = (ss >= 1e-10) ? num / (ss * sqrt(ss)) : 0.0;  // SM:526-527
        kap = (OT)kd;
        th = heading_of<OT>(d1y, d1x);                                 // SM:536
        // SM:204-215 get_point_at_parameter(t) at the sample's own parameter
        normalize_parameter(t, t_max, G, lt, sg);
        cx = pw + (size_t)sg * 12;
        cy = cx + 6;
        px = (OT)poly_p(cx, lt);
        py = (OT)poly_p(cy, lt);
    }
    s_dx[tid] = d1x;
    s_dy[tid] = d1y;
    s_j[tid] = jj;
    s_th[tid] = th;
    __syncthreads();
    if (tid < kSampleTile && k < S) {
        OT dth = (OT)0;
        if (k < N - 1) {
            // |heading[k+1] - heading[k]| of the reference's raw (un-unwrapped) atan2 values
            if constexpr (sizeof(OT) == 8) {
                dth = fabs(s_th[tid + 1] - th);
            } else {
                if (s_j[tid + 1] != jj) {
                    // small-angle accurate: angle between the two fp64 derivative vectors, then the
                    // 2*pi multiple that the raw difference of the two atan2 values carries
                    const double nx = s_dx[tid + 1], ny = s_dy[tid + 1];
                    const float cr = (float)(d1x * ny - d1y * nx);
                    const float dt = (float)(d1x * nx + d1y * ny);
                    const float dl = atan2f(cr, dt);
                    const float raw = s_th[tid + 1] - th;
                    const float n = rintf((raw - dl) * 0.15915494309189535f);
                    dth = fabsf(fmaf(n, 6.283185307179586f, dl));
                }
            }
        }
        if (k >= N) { px = py = th = kap = (OT)0; }
        if (ox) ox[row + k] = px;
        if (oy) oy[row + k] = py;
        if (oh) oh[row + k] = th;
        if (ok) ok[row + k] = kap;
        if (odth) odth[row + k] = dth;
    }
}

// ------------------------------------------------------------------------------------------------
// K5 (v0): forward/backward velocity pass, one lane per path, strictly sequential — the exact
// statement of MPG:188-311 in squared-velocity space.  Used as the in-library reference for the
// relaxation kernel and for batches with very many short paths.
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(64) void k_velocity_seq(int B, int S, VelConsts<R> c, R start_u, R end_u,
                                                     const double *__restrict__ meta,
                                                     const R *__restrict__ curv, const R *__restrict__ dtheta,
                                                     const R *__restrict__ vcap, R *__restrict__ vel)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *m = meta + (size_t)b * kMetaStride;
    const R twodd = (R)2 * (R)m[2];
    const int N = (int)m[3];
    const size_t row = (size_t)b * S;
    const R *K = curv + row, *DT = dtheta + row;
    R *V = vel + row;
    const R vmax2 = c.vmax * c.vmax;
    // forward, MPG:188-249
    R u = start_u, wprev = (R)0;
    V[0] = u;
    for (int i = 0; i < N - 1; i++) {
        const SampleLimits<R> L = sample_limits(c, (R)fabs(K[i]), c.amax, c.adec);
        R un = (i + 1 == N - 1) ? end_u : (vcap ? vcap[row + i + 1] * vcap[row + i + 1] : vmax2);
        u = forward_step(c, L, c.amax, twodd, u, wprev, DT[i], un);
        V[i + 1] = u;
    }
    // backward, MPG:251-311
    u = end_u;
    wprev = (R)0;
    for (int i = N - 1; i > 0; i--) {
        const SampleLimits<R> L = sample_limits(c, (R)fabs(K[i]), c.amax, c.adec);
        const R up = backward_step(c, L, c.amax, twodd, u, wprev, DT[i - 1], V[i - 1]);
        V[i] = sqrt(u);
        u = up;
    }
    V[0] = sqrt(u);
    for (int i = N; i < S; i++) V[i] = (R)0;
}

// ------------------------------------------------------------------------------------------------
// Segment blocks -> monomial coefficients (scratch used by k_sample).
// ------------------------------------------------------------------------------------------------
__global__ void k_power(int n_seg, const double *__restrict__ segments, double *__restrict__ power)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seg) return;
    const double *r = segments + (size_t)i * 12;
    double cx[6], cy[6];
    hermite_to_power(r[0], r[2], r[4], r[6], r[8], r[10], cx);
    hermite_to_power(r[1], r[3], r[5], r[7], r[9], r[11], cy);
    double *pw = power + (size_t)i * 12;
#pragma unroll
    for (int k = 0; k < 6; k++) { pw[k] = cx[k]; pw[6 + k] = cy[k]; }
}

// ------------------------------------------------------------------------------------------------
// Scalar/vector accessors of one fitted path (the calls the GUI and L2 make one value at a time):
// SM:204-241 point / derivative / second derivative at a parameter.
// ------------------------------------------------------------------------------------------------
__global__ void k_eval(int W, const double *__restrict__ seg, double t_max, int order, int n,
                       const double *__restrict__ t, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x, y;
    hermite_eval_ref(seg, t_max, W - 1, order, t[i], x, y);
    out[2 * i] = x;
    out[2 * i + 1] = y;
}

// what: 0 = SM:291-318 distance_to_time, 1 = SM:340-346 get_curvature, 2 = SM:332-338 get_heading
__global__ void k_lookup(int W, const double *__restrict__ seg, double t_max, const double *__restrict__ lut,
                         int what, int n, const double *__restrict__ in, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double end_param = (double)(W - 1);
    if (what == 0) {
        out[i] = distance_to_time(lut, lut[kLutN - 1], t_max, end_param, in[i]);
        return;
    }
    const int tab_n = W * kSamplesPerNode;
    const int jj = table_index(in[i], tab_n, end_param);
    const double tp = linspace_at(end_param, tab_n, jj);
    double d1x, d1y;
    hermite_eval_ref(seg, t_max, W - 1, 1, tp, d1x, d1y);
    if (what == 2) {
        out[i] = atan2(d1y, d1x);  // SM:536
    } else {
        double d2x, d2y;
        hermite_eval_ref(seg, t_max, W - 1, 2, tp, d2x, d2y);
        const double ss = d1x * d1x + d1y * d1y;
        const double num = d1x * d2y - d1y * d2x;
        out[i] = (ss >= 1e-10) ? num / (ss * sqrt(ss)) : 0.0;  // SM:517-527
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <typename IT>
static hipError_t launch_fit_t(hipStream_t st, int B, int W, const void *wp, const double *tin,
                               const double *tout, double *seg, double *pw, double *meta, uint32_t *flags)
{
    const size_t lds = sizeof(double) * (size_t)(7 * W);
    hipLaunchKernelGGL(k_fit<IT>, dim3(B), dim3(W <= 64 ? 64 : 256), lds, st, W, (const IT *)wp, tin, tout,
                       seg, pw, meta, flags);
    return hipGetLastError();
}

hipError_t launch_fit(hipStream_t st, bool f64, int B, int W, const void *wp, const double *tin,
                      const double *tout, double *seg, double *pw, double *meta, uint32_t *flags)
{
    return f64 ? launch_fit_t<double>(st, B, W, wp, tin, tout, seg, pw, meta, flags)
               : launch_fit_t<float>(st, B, W, wp, tin, tout, seg, pw, meta, flags);
}

hipError_t launch_lut(hipStream_t st, int B, int W, const double *seg, double *lut, double *meta,
                      uint32_t *flags)
{
    hipLaunchKernelGGL(k_lut, dim3(B), dim3(256), 0, st, W, seg, lut, meta, flags);
    return hipGetLastError();
}

hipError_t launch_grid(hipStream_t st, int B, int S, double dd, double *meta, uint32_t *flags)
{
    hipLaunchKernelGGL(k_grid, dim3((B + 255) / 256), dim3(256), 0, st, B, S, dd, meta, flags);
    return hipGetLastError();
}

hipError_t launch_sample(hipStream_t st, bool f64, int B, int W, int S, const double *pw, const double *lut,
                         const double *meta, void *x, void *y, void *h, void *k, void *dth)
{
    const dim3 grid((S + kSampleTile - 1) / kSampleTile, B);
    if (f64)
        hipLaunchKernelGGL(k_sample<double>, grid, dim3(kSampleThreads), 0, st, W, S, pw, lut, meta, (double *)x,
                           (double *)y, (double *)h, (double *)k, (double *)dth);
    else
        hipLaunchKernelGGL(k_sample<float>, grid, dim3(kSampleThreads), 0, st, W, S, pw, lut, meta, (float *)x,
                           (float *)y, (float *)h, (float *)k, (float *)dth);
    return hipGetLastError();
}

template <typename R>
static VelConsts<R> make_consts(const double c[6])
{
    VelConsts<R> v;
    v.vmax = (R)c[0];
    v.amax = (R)c[1];
    v.adec = (R)c[2];
    v.tw = (R)c[5];
    v.wmax = (R)2 * v.vmax / v.tw;
    v.almax = (R)2 * v.amax / v.tw;
    return v;
}

hipError_t launch_velocity_seq(hipStream_t st, bool f64, int B, int S, const double c[6], double sv, double ev,
                               const double *meta, const void *curv, const void *dth, const void *vcap,
                               void *vel)
{
    const dim3 grid((B + 63) / 64);
    if (f64) {
        hipLaunchKernelGGL(k_velocity_seq<double>, grid, dim3(64), 0, st, B, S, make_consts<double>(c), sv * sv,
                           ev * ev, meta, (const double *)curv, (const double *)dth, (const double *)vcap,
                           (double *)vel);
    } else {
        const float svf = (float)sv, evf = (float)ev;
        hipLaunchKernelGGL(k_velocity_seq<float>, grid, dim3(64), 0, st, B, S, make_consts<float>(c), svf * svf,
                           evf * evf, meta, (const float *)curv, (const float *)dth, (const float *)vcap,
                           (float *)vel);
    }
    return hipGetLastError();
}

hipError_t launch_power(hipStream_t st, int n_seg, const double *seg, double *pw)
{
    hipLaunchKernelGGL(k_power, dim3((n_seg + 255) / 256), dim3(256), 0, st, n_seg, seg, pw);
    return hipGetLastError();
}

hipError_t launch_eval(hipStream_t st, int W, const double *seg, double t_max, int order, int n, const double *t,
                       double *out)
{
    hipLaunchKernelGGL(k_eval, dim3((n + 255) / 256), dim3(256), 0, st, W, seg, t_max, order, n, t, out);
    return hipGetLastError();
}

hipError_t launch_lookup(hipStream_t st, int W, const double *seg, double t_max, const double *lut, int what,
                         int n, const double *in, double *out)
{
    hipLaunchKernelGGL(k_lookup, dim3((n + 255) / 256), dim3(256), 0, st, W, seg, t_max, lut, what, n, in, out);
    return hipGetLastError();
}

}  // namespace vap
