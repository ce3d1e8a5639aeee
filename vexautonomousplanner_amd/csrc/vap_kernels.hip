// vap_kernels.hip — HIP kernels (gfx950 / CDNA4) of the batched trajectory generator.
//
// Stage map: K1 fit -> K2 arc-length LUT -> K3+K4 sample -> K5 velocity pass.
// Data layout in HBM (B paths, W waypoints, G = W-1 segments, S sample capacity):
//   segments [B][G][6][2] fp64   reference row order (QHS:120-122)
//   power    [B][G][30]   fp64   monomial coefficients of P, P', P'' (x then y each; scratch)
//   lut      [B][1000]    fp64   cumulative trapezoid distances (SM:448-454)
//   meta     [B][4]       fp64   {param_last, total_length, dd, n_samples}
//   x,y,heading,curvature,dtheta,velocity [B][S]  fp32 or fp64, sample-major per path so that a
//   wavefront of consecutive samples reads/writes 256 (fp32) or 512 (fp64) contiguous bytes.
#include "vap_device.h"
#include "vap_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace vap {

static_assert(kCoefBlockDoubles == kCoefDoubles, "scratch sizing and block layout disagree");
static_assert(kGridRunBlockDoubles == kGridRunDoubles, "scratch sizing and run-table layout disagree");

// ------------------------------------------------------------------------------------------------
// K1: fit.  One workgroup per path.  QHS:30-138, 149-219, 719-736; SM:65-77 tangent overrides.
// LDS (dynamic): pts[W][2], dist[G], fd[W][2], sd[W][2]  (fp64)
// ------------------------------------------------------------------------------------------------
// ex (optional, QuinticHermiteSpline's own call surface): caller-supplied derivatives — used only when BOTH arrays are
// given (QHS:52-68: with one of them missing _compute_derivatives overwrites both) — and the starting / ending
// tangent of a split spline (QHS:129-132, 543-590; quirk Q3: both patch the LAST segment).
struct FitExtras {
    const double *first = nullptr, *second = nullptr;    // [B][W][2]
    const double *start_tan = nullptr, *end_tan = nullptr;   // [B][2], NaN = not set
    double *out_first = nullptr, *out_second = nullptr;      // [B][W][2]: the derivatives the segments were built from
};

// The fit of path b by the calling workgroup (sh: 7*W doubles of LDS).  Ends with a workgroup barrier; the segments,
// coefficient blocks, meta[0] and flags of the path are then written (visible to this workgroup after a fence).
template <typename IT>
__device__ __forceinline__ void fit_path(int b, int W, const IT *__restrict__ waypoints, const double *__restrict__ tan_in,
                                         const double *__restrict__ tan_out, const FitExtras &ex, double *__restrict__ segments,
                                         double *__restrict__ power, double *__restrict__ seglen, double *__restrict__ meta,
                                         uint32_t *__restrict__ flags, double *sh, int tid = threadIdx.x, int nt = blockDim.x,
                                         uint32_t *flag_word = nullptr)
{
    // (tid, nt: the threads that work on THIS path — the whole workgroup, or a sub-group of it when k_fit_many packs
    // several short paths into one; every __syncthreads below is reached by all threads of the workgroup either way)
    const int G = W - 1;
    double *pts = sh;             // 2W
    double *dist = pts + 2 * W;   // G (W slots)
    double *fd = dist + W;        // 2W
    double *sd = fd + 2 * W;      // 2W
    __shared__ uint32_t s_flag_one;
    uint32_t &s_flag = flag_word ? *flag_word : s_flag_one;
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
    const bool have_d = ex.first && ex.second;
    const bool has_st = ex.start_tan && !isnan(ex.start_tan[(size_t)b * 2]);
    const bool has_en = ex.end_tan && !isnan(ex.end_tan[(size_t)b * 2]);
    // QHS:163-195 first derivatives
    for (int i = tid; i < W; i += nt) {
        double fx, fy;
        if (have_d) {
            fx = ex.first[((size_t)b * W + i) * 2];
            fy = ex.first[((size_t)b * W + i) * 2 + 1];
        } else if (i == 0) {
            // QHS:170-172: a 2-point spline with an ending tangent keeps the chord un-normalised
            const double d = (W == 2 && has_en) ? 1.0 : dist[0];
            fx = (pts[2] - pts[0]) / d;
            fy = (pts[3] - pts[1]) / d;
        } else if (i == W - 1) {
            const double d = (W == 2 && has_st) ? 1.0 : dist[G - 1];   // QHS:181-182
            fx = (pts[2 * i] - pts[2 * (i - 1)]) / d;
            fy = (pts[2 * i + 1] - pts[2 * (i - 1) + 1]) / d;
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
        if (have_d) {
            sx = ex.second[((size_t)b * W + i) * 2];
            sy = ex.second[((size_t)b * W + i) * 2 + 1];
        } else if (i > 0 && i < W - 1) {
            const double avg = (dist[i - 1] + dist[i]) / 2;
            sx = (fd[2 * (i + 1)] - fd[2 * (i - 1)]) / (avg * 0.5);
            sy = (fd[2 * (i + 1) + 1] - fd[2 * (i - 1) + 1]) / (avg * 0.5);
        }
        sd[2 * i] = sx;
        sd[2 * i + 1] = sy;
    }
    __syncthreads();
    if (ex.out_first && ex.out_second) {
        // the attributes the reference leaves behind: estimates (or the caller's arrays), with the tangent setters'
        // writes to first_derivatives[0] / [-1] (QHS:557, 582)
        for (int i = tid; i < 2 * W; i += nt) {
            double f = fd[i];
            if (has_st && i < 2) f = ex.start_tan[(size_t)b * 2 + i];
            if (has_en && i >= 2 * (W - 1)) f = ex.end_tan[(size_t)b * 2 + i - 2 * (W - 1)];   // (applied second, as QHS:131-132)
            ex.out_first[(size_t)b * W * 2 + i] = f;
            ex.out_second[(size_t)b * W * 2 + i] = sd[i];
        }
    }
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
        if (i == G - 1) {   // QHS:129-132 -> 543-590: both setters write into segments[-1] (quirk Q3)
            if (has_st) { r[4] = ex.start_tan[(size_t)b * 2]; r[5] = ex.start_tan[(size_t)b * 2 + 1]; }
            if (has_en) { r[6] = ex.end_tan[(size_t)b * 2]; r[7] = ex.end_tan[(size_t)b * 2 + 1]; }
        }
        double *sg = segments + ((size_t)b * G + i) * 12;
#pragma unroll
        for (int k = 0; k < 12; k++) sg[k] = r[k];
        if (power) make_coef_block(r, power + ((size_t)b * G + i) * kCoefDoubles);
        if (seglen) seglen[(size_t)b * G + i] = L;
    }
    __syncthreads();
    if (tid == 0 && flags) flags[b] = s_flag;
}

template <typename IT>
__global__ __launch_bounds__(256) void k_fit(int W, const IT *__restrict__ waypoints,
                                             const double *__restrict__ tan_in,
                                             const double *__restrict__ tan_out, FitExtras ex,
                                             double *__restrict__ segments, double *__restrict__ power,
                                             double *__restrict__ seglen, double *__restrict__ meta,
                                             uint32_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    fit_path<IT>(blockIdx.x, W, waypoints, tan_in, tan_out, ex, segments, power, seglen, meta, flags, sh);
}

// K1 for very many short paths (config 5: 131 072 paths of 8 waypoints): k_fit gives a path a workgroup of 64 threads of
// which W work; here a workgroup of 256 threads takes 256 / L paths, L = the power of two >= W lanes each — the same
// fit_path, the same rows.
template <typename IT>
__global__ __launch_bounds__(256) void k_fit_many(int B, int W, int L, const IT *__restrict__ waypoints, double *__restrict__ segments,
                                                  double *__restrict__ power, double *__restrict__ meta, uint32_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];   // (256 / L) paths x 7W doubles
    __shared__ uint32_t s_flags[64];
    const int per = 256 / L, sub = threadIdx.x / L, lane = threadIdx.x % L;
    int b = blockIdx.x * per + sub;
    // (paths past the batch: the last path once more — same values into the same places, so every thread reaches the barriers)
    b = b < B ? b : B - 1;
    fit_path<IT>(b, W, waypoints, nullptr, nullptr, FitExtras(), segments, power, nullptr, meta, flags, sh + (size_t)sub * 7 * W, lane, L,
                 &s_flags[sub]);
}

__global__ void k_grid(int B, int W, int S, double dd_in, double *__restrict__ meta, double *__restrict__ aux,
                       double *__restrict__ runs, uint32_t *__restrict__ flags)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    grid_define(b, W, S, dd_in, meta[(size_t)b * kMetaStride + 1], meta[(size_t)b * kMetaStride + 0], meta, aux, runs, flags);
}

// ------------------------------------------------------------------------------------------------
// K2: arc-length lookup table.  One workgroup per path.  SM:426-475.
// 1000 uniform-parameter samples of |P'(t)| (reference basis, reference order), trapezoid, and a
// SEQUENTIAL cumulative sum (np.cumsum order) so the table is bit-identical to the reference's.
// ------------------------------------------------------------------------------------------------
// Routes cut into several splines (rt.sptab set): grid = (routes, spline slots); the workgroup builds the partial
// (un-offset) table of one spline, SM:436-454, and k_route_offsets (vap_routes_batch.hip) does the rest.
constexpr int kLutPad = (kLutN + 31) / 32 * 32;
template <bool SEG_LDS>
__device__ __forceinline__ void lut_path(int W, const double *__restrict__ segments, double *__restrict__ lut,
                                         double *__restrict__ slopes, double *__restrict__ meta, uint32_t *__restrict__ flags,
                                         const GridArgs &grid, const RouteTables &rt, long long *__restrict__ stats, double *s_seg,
                                         double *cum)
{
    const long long tl0 = stats ? __builtin_amdgcn_s_memtime() : 0;
    constexpr int kPad = kLutPad;   // the sequential sum walks whole groups of 32
    // cum (LDS, kLutPad doubles, 16-byte aligned): magnitudes, then (in place) trapezoid increments, then cumulative distances
    double *mag = cum;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    int G = W - 1;
    double t_max;
    const double *seg = segments + (size_t)b * (W - 1) * 12;
    size_t row = (size_t)b;                 // table row: the path, or (route, spline slot)
    if (rt.sptab) {
        const int sl = blockIdx.y, n = rt.nspl[b];
        if (sl >= n) return;
        const double *sp = rt.sptab + ((size_t)b * rt.NS + sl) * kSplineStride;
        const int first = (int)sp[3], last = sl + 1 < n ? (int)sp[kSplineStride + 3] : W - 1;
        G = last - first;
        t_max = sp[0];
        seg += (size_t)first * 12;
        row = (size_t)b * rt.NS + sl;
    } else {
        t_max = meta[(size_t)b * kMetaStride + 0];
    }
    if constexpr (SEG_LDS) {
        lds_fill<4>(s_seg, seg, G * 12, tid, nt);
        __syncthreads();
        seg = s_seg;
    }
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
    const long long tl1 = stats ? __builtin_amdgcn_s_memtime() : 0;
    // trapezoid increments in parallel (SM:452-454: (m[j-1] + m[j]) * 0.5 * dt, that association) ...
    const double dt = linspace_at(t_max, kLutN, 1) - linspace_at(t_max, kLutN, 0);  // SM:444
    {
        constexpr int kPer = kPad / 128 + 1;   // increments per thread for the smallest launch (128 threads)
        double inc[kPer];
#pragma unroll
        for (int it = 0; it < kPer; it++) {
            const int j = tid + it * nt;
            inc[it] = (j > 0 && j < kLutN) ? (mag[j - 1] + mag[j]) * 0.5 * dt : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kPer; it++) {
            const int j = tid + it * nt;
            if (j < kPad) cum[j] = inc[it];
        }
    }
    __syncthreads();
    // ... and np.cumsum's strictly left-to-right sum.  Lane 0 walks the whole chain but keeps only the
    // running sum at every 32nd element (no stores on the chain); then lane g replays group g from its
    // exact starting sum — the same adds in the same order, so every prefix is bit-identical to the
    // one-lane result — and stores it.
    __shared__ double s_start[kPad / 32];
    __shared__ double s_total;
    if (tid == 0) {
        double acc = 0.0;   // cum[0] = 0: the first add is the exact 0 + 0 of partial_distances[0]
        // (+ current_dist of SM:457 is + 0.0 for a single spline: a no-op on these non-negative sums)
        const double2 *cum2 = reinterpret_cast<const double2 *>(cum);
#pragma unroll 1
        for (int g0 = 0; g0 < kPad / 32; g0++) {
            s_start[g0] = acc;
            double2 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = cum2[g0 * 16 + k];
#pragma unroll
            for (int k = 0; k < 16; k++) { acc += v[k].x; acc += v[k].y; }
        }
        if (!rt.sptab) meta[(size_t)b * kMetaStride + 1] = acc;   // elements past kLutN-1 are zero increments
        s_total = acc;
        if (flags && !(acc > 0.0 && isfinite(acc))) atomicOr(&flags[b], VAP_FLAG_DEGENERATE_BIT);
    }
    __syncthreads();
    // the fused call knows the grid spacing already: the last thread (a wave with nothing to do during the
    // replay below) defines the path's distance grid, which would otherwise be a launch of its own
    if (grid.aux && !rt.sptab && tid == nt - 1) grid_define(b, W, grid.S, grid.dd, s_total, t_max, meta, grid.aux, grid.runs, flags);
    if (tid < kPad / 32) {
        double acc = s_start[tid];
        double2 *cum2 = reinterpret_cast<double2 *>(cum) + tid * 16;
        double2 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = cum2[k];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            acc += v[k].x;
            v[k].x = acc;
            acc += v[k].y;
            v[k].y = acc;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) cum2[k] = v[k];
    }
    __syncthreads();
    const long long tl2 = stats ? __builtin_amdgcn_s_memtime() : 0;
    const double lstep = t_max / (double)(kLutN - 1);
    for (int j = tid; j < kLutN; j += nt) {
        lut[row * kLutN + j] = cum[j];
        if (slopes) {
            // (t1 - t0)/(d1 - d0) of SM:311-317 for the interval ending at entry j
            double w = 0.0;
            if (j > 0) {
                const double t0 = (double)(j - 1) * lstep, t1 = (j == kLutN - 1) ? t_max : (double)j * lstep;
                w = div_inrange(t1 - t0, cum[j] - cum[j - 1]);
            }
            slopes[row * kLutN + j] = w;
        }
    }
    if (stats && tid == 0) {
        stats[(size_t)b * 4 + 0] = tl1 - tl0;
        stats[(size_t)b * 4 + 1] = tl2 - tl1;
        stats[(size_t)b * 4 + 2] = __builtin_amdgcn_s_memtime() - tl2;
    }
}

template <bool SEG_LDS>
__global__ __launch_bounds__(256) void k_lut(int W, const double *__restrict__ segments,
                                             double *__restrict__ lut, double *__restrict__ slopes,
                                             double *__restrict__ meta, uint32_t *__restrict__ flags,
                                             GridArgs grid, RouteTables rt, long long *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) double s_seg[];   // G * 12 when SEG_LDS
    __shared__ __attribute__((aligned(16))) double cum[kLutPad];
    lut_path<SEG_LDS>(W, segments, lut, slopes, meta, flags, grid, rt, stats, s_seg, cum);
}

// K1 + K2 in one launch for the fused call on plain paths (vap_profile_batch): the workgroup that fits a path builds its
// table right away — one launch and its ramp less per step (24 + 57 us as two kernels at config 3).  The same two
// bodies: same segments, same table.  Sized so that a CU holds sixteen of these workgroups at once — config 3's 4096
// paths are then ONE round over the 256 CUs, not a full one and a third: at most 64 registers, and one LDS array for both
// phases (the fit's arrays, then the table: max(1024, 7 W) doubles), the segment rows read back through L1.
template <typename IT>
__global__ __launch_bounds__(128, 8) void k_fit_lut(int W, const IT *__restrict__ waypoints, double *__restrict__ segments,
                                                    double *__restrict__ power, double *__restrict__ lut,
                                                    double *__restrict__ meta, uint32_t *__restrict__ flags, GridArgs grid,
                                                    long long *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];     // max(kLutPad, 7 * W) doubles
    const long long t0 = stats ? __builtin_amdgcn_s_memtime() : 0;
    fit_path<IT>(blockIdx.x, W, waypoints, nullptr, nullptr, FitExtras(), segments, power, nullptr, meta, flags, sh);
    __threadfence_block();      // the segments and meta[0] this workgroup wrote, read back below
    __syncthreads();
    if (stats && threadIdx.x == 0) stats[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - t0;
    lut_path<false>(W, segments, lut, nullptr, meta, flags, grid, RouteTables(), stats, nullptr, sh);
}

// K2 for very many short paths (config 5: 131 072 paths x 8 waypoints): one workgroup builds the tables of 64 paths.
// The 1000-entry sequential sum — np.cumsum's order, kept for bit-identity — costs a one-lane chain of 1000
// dependent fp64 adds per path in k_lut; here the 64 lanes of one wavefront each walk their own path's chain, so
// the chain is paid once per 64 paths.  Entries go through LDS in tiles of 32 per path: all threads evaluate
// magnitudes and trapezoid increments (lanes = consecutive entries of a path, so segment rows are read as
// broadcasts), the scanner wave adds, all threads store distances and slopes (256-byte rows).
// Same expressions in the same order as k_lut: the tables are bit-identical.
// Two group sizes: 64 paths per workgroup for very many short paths (W <= 9: config 5), 8 for large batches of paths of up
// to 64 waypoints (config 4's share: 1000 entries of 8 paths are one entry per thread and tile, the eight chains run in
// eight lanes).
constexpr int kLutManyTile = 32;
constexpr int kLutManyMinPaths = 32768;   // P = 64: batches from here on (and W <= 9)
constexpr int kLutGroupMinPaths = 8192;   // P = 8: batches from here on (and W <= 64); measured at 4096 x 32: 59 us against
                                          // k_lut's 57 (two workgroups per CU are too few to hide the barriers), at 8192: 75 against 93
template <int kLutManyPaths, int kLutManyThreads>
__global__ __launch_bounds__(kLutManyThreads, kLutManyThreads == 512 ? 4 : 1) void k_lut_many(int B, int W, const double *__restrict__ segments,
                                                               double *__restrict__ lut, double *__restrict__ slopes,
                                                               double *__restrict__ meta, uint32_t *__restrict__ flags,
                                                               GridArgs grid)
{
    constexpr int P = kLutManyPaths, TE = kLutManyTile, ST = TE + 1;   // row stride 33: lane = path reads hit 64 banks
    extern __shared__ __attribute__((aligned(16))) double s_seg[];     // the 64 paths' segment rows: P * G * 12
    __shared__ double s_mag[P * ST], s_inc[P * ST];
    __shared__ double s_prev_mag[P], s_carry[P], s_prev_cum[P], s_tmax[P], s_step[P];
    // Paths whose parameters[-1] is exactly W-1 (most: 78 % of config 5's — the rest are an ulp off, QHS:719-736) share
    // their table parameters linspace(0, W-1, 1000), hence segment index, local parameter and the six basis values of
    // every entry: those are evaluated ONCE per entry and workgroup (the reference's expressions, so the same bits), one
    // tile ahead, and a path of that class reads them instead of repeating ~40 of its ~130 operations per entry.
    // (one buffer, rewritten for the next tile right after the barrier that ends a tile's magnitude phase; bytes where
    // bytes do: two of these workgroups share a CU's 160 KB only while each stays under 80 KB)
    __shared__ double s_H[TE][6];
    __shared__ unsigned char s_hidx[TE], s_std[P];
    const int tid = threadIdx.x, b0 = blockIdx.x * P;
    const int G = W - 1;
    auto shared_basis = [&](int tile) {      // threads 0..TE-1: the class's entries of `tile`
        const int j = tile * TE + tid;
        if (tid < TE && j < kLutN) {
            const double t_max = (double)G;
            const double step = t_max / (double)(kLutN - 1);
            const double t = (j == kLutN - 1) ? t_max : (double)j * step;
            double lt, H[6];
            int idx;
            normalize_parameter(t, t_max, G, lt, idx);
            hermite_d1_basis_ref(lt, H);
#pragma unroll
            for (int i = 0; i < 6; i++) s_H[tid][i] = H[i];
            s_hidx[tid] = (unsigned char)idx;    // (W <= 64 in both group sizes)
        }
    };
    const int n_paths = B - b0 < P ? B - b0 : P;
    lds_fill<4>(s_seg, segments + (size_t)b0 * G * 12, n_paths * G * 12, tid, kLutManyThreads);
    if (tid < P) {
        s_prev_mag[tid] = 0.0; s_carry[tid] = 0.0; s_prev_cum[tid] = 0.0;
        const double tm = tid < n_paths ? meta[(size_t)(b0 + tid) * kMetaStride + 0] : 1.0;
        s_tmax[tid] = tm;
        // np.linspace's step (SM:443), once per path: the per-entry expressions below used to repeat this division —
        // and its twin in dt and in the slope — for every one of the 1000 entries (five IEEE divisions per entry
        // where one is needed)
        s_step[tid] = tm / (double)(kLutN - 1);
        s_std[tid] = tm == (double)G;
    }
    shared_basis(0);
    __syncthreads();
    constexpr int kTiles = (kLutN + TE - 1) / TE;
#pragma unroll 1
    for (int tl = 0; tl < kTiles; tl++) {
        const int j0 = tl * TE;
        // magnitudes |P'(t_j)| (SM:447-448): item = (path, entry), a wavefront covers two paths' 32 entries
#pragma unroll 2
        for (int r = 0; r < P * TE / kLutManyThreads; r++) {
            const int it = tid + r * kLutManyThreads, p = it / TE, e = it % TE, j = j0 + e, b = b0 + p;
            double m = 0.0;
            if (b < B && j < kLutN) {
                double H[6];
                int idx;
                if (s_std[p]) {
#pragma unroll
                    for (int i = 0; i < 6; i++) H[i] = s_H[e][i];
                    idx = s_hidx[e];
                } else {
                    const double t_max = s_tmax[p];
                    const double t = (j == kLutN - 1) ? t_max : (double)j * s_step[p];   // linspace_at(t_max, kLutN, j)
                    double lt;
                    normalize_parameter(t, t_max, G, lt, idx);
                    hermite_d1_basis_ref(lt, H);
                }
                double dx, dy;
                hermite_combine_ref(s_seg + ((size_t)p * G + idx) * 12, H, dx, dy);
                m = sqrt(dx * dx + dy * dy);
            }
            s_mag[p * ST + e] = m;
        }
        __syncthreads();
        if (tl + 1 < kTiles) shared_basis(tl + 1);     // (this tile's readers are past the barrier; the next tile's come after two more)
        // trapezoid increments (SM:452-454: (m[j-1] + m[j]) * 0.5 * dt, that association)
#pragma unroll
        for (int r = 0; r < P * TE / kLutManyThreads; r++) {
            const int it = tid + r * kLutManyThreads, p = it / TE, e = it % TE, j = j0 + e, b = b0 + p;
            double inc = 0.0;
            if (b < B && j > 0 && j < kLutN) {
                const double dt = (double)1 * s_step[p] - (double)0 * s_step[p];   // SM:444: local_params[1] - local_params[0]
                const double mp = e > 0 ? s_mag[p * ST + e - 1] : s_prev_mag[p];
                inc = (mp + s_mag[p * ST + e]) * 0.5 * dt;
            }
            s_inc[p * ST + e] = inc;
        }
        __syncthreads();
        // np.cumsum's strictly left-to-right sum, one lane per path
        if (tid < P) {
            double acc = s_carry[tid];
            const double last_mag = s_mag[tid * ST + TE - 1];
#pragma unroll
            for (int h = 0; h < TE; h += 8) {      // eight reads in flight, eight dependent adds, eight writes
                double v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = s_inc[tid * ST + h + e];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    acc += v[e];
                    v[e] = acc;
                }
#pragma unroll
                for (int e = 0; e < 8; e++) s_mag[tid * ST + h + e] = v[e];   // the tile's distances
            }
            s_prev_mag[tid] = last_mag;
            s_carry[tid] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < P * TE / kLutManyThreads; r++) {
            const int it = tid + r * kLutManyThreads, p = it / TE, e = it % TE, j = j0 + e, b = b0 + p;
            if (b < B && j < kLutN) {
                const double c = s_mag[p * ST + e];
                lut[(size_t)b * kLutN + j] = c;
                if (slopes) {
                    // (t1 - t0)/(d1 - d0) of SM:311-317 for the interval ending at entry j
                    double w = 0.0;
                    if (j > 0) {
                        const double t_max = s_tmax[p];
                        const double lstep = s_step[p];
                        const double t0 = (double)(j - 1) * lstep, t1 = (j == kLutN - 1) ? t_max : (double)j * lstep;
                        const double cp = e > 0 ? s_mag[p * ST + e - 1] : s_prev_cum[p];
                        w = div_inrange(t1 - t0, c - cp);
                    }
                    slopes[(size_t)b * kLutN + j] = w;
                }
            }
        }
        __syncthreads();
        if (tid < P) s_prev_cum[tid] = s_mag[tid * ST + TE - 1];
        // (the next tile's first barrier orders this write before its readers)
    }
    if (tid < P && b0 + tid < B) {
        const int b = b0 + tid;
        const double acc = s_carry[tid];
        meta[(size_t)b * kMetaStride + 1] = acc;
        if (flags && !(acc > 0.0 && isfinite(acc))) atomicOr(&flags[b], VAP_FLAG_DEGENERATE_BIT);
        if (grid.aux) grid_define(b, W, grid.S, grid.dd, acc, meta[(size_t)b * kMetaStride + 0], meta, grid.aux, grid.runs, flags);
    }
}

// Slopes for a distance table that came in through the staged API.
__global__ void k_lut_slopes(int B, const double *__restrict__ lut, const double *__restrict__ meta,
                             double *__restrict__ slopes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * kLutN) return;
    const int b = i / kLutN, j = i % kLutN;
    const double t_max = meta[(size_t)b * kMetaStride + 0];
    const double lstep = t_max / (double)(kLutN - 1);
    double w = 0.0;
    if (j > 0) {
        const double t0 = (double)(j - 1) * lstep, t1 = (j == kLutN - 1) ? t_max : (double)j * lstep;
        w = div_inrange(t1 - t0, lut[i] - lut[i - 1]);
    }
    slopes[i] = w;
}

// ------------------------------------------------------------------------------------------------
// Grid definition: one thread per path.  MPG:112-122 sample count, or this build's fixed-S grid.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// K3+K4: sampling.  grid = (tiles, B); a workgroup evaluates kSampleChunk consecutive samples of one
// path, each thread kSPT consecutive ones (so every output leaves as one 16-byte store per lane and
// the arc-length table is walked, not searched, after the thread's first sample).  The last thread's
// samples belong to the next tile: they only supply the neighbour for |dtheta| (rows that fit one pass,
// S <= kSampleChunk, need no such neighbour and are written by all threads).
// LDS: distance table + interval slopes (16 KB), the path's coefficient blocks when they fit
// (G <= kLdsCoefSegments; otherwise they are read through L1/L2, where a wavefront's consecutive
// samples hit one or two blocks), and a 256-entry neighbour exchange.
// Arithmetic: parameter/index path, derivative evaluation and curvature in fp64 (DESIGN.md
// §Numerics); no fp64 division on the per-sample path.
// ------------------------------------------------------------------------------------------------
// HI (fp32 outputs only): also write the curvature and |dtheta| rows in fp64 (ok64, odth64) for the fp64
// velocity recurrence behind fp32 outputs; the fp32 |dtheta| row is then optional (odth may be NULL).
template <typename OT, bool COEF_LDS, bool HI, int PPB = 1>
__global__ __launch_bounds__(kSampleThreads, 3) void k_sample(int B, int W, int S, int tile, int tiles_per_block,
                                                           const double *__restrict__ power,
                                                           const double *__restrict__ lut,
                                                           const double *__restrict__ slopes,
                                                           const double *__restrict__ meta,
                                                           const double *__restrict__ aux,
                                                           const double *__restrict__ runs,
                                                           OT *__restrict__ ox, OT *__restrict__ oy,
                                                           OT *__restrict__ oh, OT *__restrict__ ok,
                                                           OT *__restrict__ odth, double *__restrict__ ok64,
                                                           double *__restrict__ odth64, long long *__restrict__ stats)
{
    static_assert(!HI || sizeof(OT) == 4, "the fp64 side rows belong to the fp32 output mode");
    // PPB paths per workgroup (blockIdx.y * PPB + pp).  Rows that are one tile long (config 5: 1024 samples) leave a
    // workgroup as much time in staging — a chain of dependent memory round trips: meta, aux, tables, run table — as in
    // the tile itself; with PPB = 2 the loads of both paths are in flight together and the chain is paid once for two.
    extern __shared__ __attribute__((aligned(16))) double s_coef[];   // PPB * G * kCoefDoubles when COEF_LDS
    __shared__ double sD_all[PPB * kLutN], sWt_all[PPB * kLutN];
    // neighbour exchange, double-buffered by tile parity (one barrier per tile)
    __shared__ double s_dx[2][kSampleThreads], s_dy[2][kSampleThreads];
    __shared__ int s_j[2][kSampleThreads];
    __shared__ OT s_th[2][kSampleThreads];
    const long long ts0 = stats ? __builtin_amdgcn_s_memtime() : 0;
    const int tid = threadIdx.x;
    const int G = W - 1;
    const int tile0 = blockIdx.x * tiles_per_block;
    if (tile0 * tile >= S) return;
    // tile == kSampleChunk: the whole row is one tile, so the last thread's last sample is the row's last
    // and needs no neighbour — every thread writes
    const bool writer = tile == kSampleChunk || tid < kSampleThreads - 1;
    constexpr int VW = 16 / sizeof(OT);   // elements per 16-byte store
    const bool aligned = (S % VW) == 0;   // rows (and tile starts, multiples of kSPT) then start 16-byte aligned

    // per path: the scalars of its grid and tables
    int p_b[PPB], p_N[PPB], p_nruns[PPB];
    bool p_live[PPB];
    double p_tmax[PPB], p_total[PPB], p_dd[PPB], p_lstep[PPB], p_tstep[PPB], p_inv[PPB], p_sk[PPB];
    double run_touch = 0.0;
#pragma unroll
    for (int pp = 0; pp < PPB; pp++) {
        const int bq = blockIdx.y * PPB + pp;
        p_live[pp] = bq < B;
        const int b = p_live[pp] ? bq : B - 1;
        p_b[pp] = b;
        const double *m = meta + (size_t)b * kMetaStride;
        p_tmax[pp] = m[0]; p_total[pp] = m[1]; p_dd[pp] = m[2];
        p_N[pp] = (int)m[3];
        const double *ax = aux + (size_t)b * kAuxStride;
        p_lstep[pp] = ax[0]; p_tstep[pp] = ax[1]; p_inv[pp] = ax[2];
        p_nruns[pp] = (int)ax[3];
        // (touch the head of the run table now: its lookup below depends on meta / aux and would otherwise wait
        // for memory a second time)
        run_touch += runs[(size_t)b * kGridRunDoubles + (tid < 120 ? tid : 0)];
    }
    // the paths' tables are staged once and serve every tile of this workgroup; all loads first
    {
        constexpr int ITER = 4;
        double vD[PPB][ITER], vW[PPB][ITER];
#pragma unroll
        for (int pp = 0; pp < PPB; pp++) {
            // (not a function of the path's sample count: the table loads then leave with the meta loads instead of a memory
            // round trip after them — the row is in bounds either way, p_b is clamped)
            const bool go = p_live[pp];
#pragma unroll
            for (int it = 0; it < ITER; it++) {
                const int i = tid + it * kSampleThreads;
                vD[pp][it] = (go && i < kLutN) ? lut[(size_t)p_b[pp] * kLutN + i] : 0.0;
                vW[pp][it] = (go && slopes && i < kLutN) ? slopes[(size_t)p_b[pp] * kLutN + i] : 0.0;
            }
        }
        static_assert(ITER * kSampleThreads >= kLutN, "one pass covers the table");
        if constexpr (COEF_LDS) {
#pragma unroll
            for (int pp = 0; pp < PPB; pp++)
                if (p_live[pp])     // (as the tables: not behind the meta loads)
                    lds_fill<4>(s_coef + (size_t)pp * G * kCoefDoubles, power + (size_t)p_b[pp] * G * kCoefDoubles, G * kCoefDoubles, tid,
                                kSampleThreads);
        }
#pragma unroll
        for (int pp = 0; pp < PPB; pp++) {
#pragma unroll
            for (int it = 0; it < ITER; it++) {
                const int i = tid + it * kSampleThreads;
                if (i < kLutN) {
                    sD_all[pp * kLutN + i] = vD[pp][it];
                    if (slopes) sWt_all[pp * kLutN + i] = vW[pp][it];
                }
            }
        }
    }
    // MPG:112-122 distance grid: the reference accumulates current_dist += dd.  A thread's first sample of a
    // tile comes from the path's run table (that sum in closed form, vap_device.h), the following ones by the
    // reference's own addition.
    auto grid_first_of = [&](const double *run_tab, double dd, int N, int n_runs, int kb0) {
        const int dd_exp = (__double2hiint(dd) >> 20) & 0x7ff;
        const int kf = kb0 < N - 1 ? kb0 : N - 1;
        // the run from the exponent of kf*dd: right except next to a binade boundary
        int r = 2 + 2 * (((__double2hiint((double)kf * dd) >> 20) & 0x7ff) - dd_exp);
        r = kf <= 0 ? 0 : r < 1 ? 1 : r;
        r = r > n_runs - 1 ? n_runs - 1 : r;
        const double *e = run_tab + 3 * r;
        int ka = (int)grid_run_k0(e, 0);
        const int kb = (int)grid_run_k0(e, 1);
        double sa = e[1], da = e[2];
        if (kf < ka || kf >= kb) {
            while ((int)grid_run_k0(run_tab, r + 1) <= kf) r++;                      // the end markers stop this
            while (r > 0 && (int)grid_run_k0(run_tab, r) > kf) r--;
            ka = (int)grid_run_k0(run_tab, r);
            sa = run_tab[3 * r + 1];
            da = run_tab[3 * r + 2];
        }
        return sa + (double)(kf - ka) * da;
    };
    // the first tile's lookup goes out together with the staging loads
#pragma unroll
    for (int pp = 0; pp < PPB; pp++)
        p_sk[pp] = (p_live[pp] && tile0 * tile < p_N[pp])
                       ? grid_first_of(runs + (size_t)p_b[pp] * kGridRunDoubles, p_dd[pp], p_N[pp], p_nruns[pp], tile0 * tile + tid * kSPT)
                       : 0.0;
    // Rows of one tile (PPB == 2 serves only those): a path's 1000 interval slopes would serve ~1000 samples once each, so a
    // sample forms the slope of ITS interval on the spot — the same expression, the same bits — and the slope array, its
    // 1000 divisions per path and a workgroup barrier go (config 5: staging is 45 % of a workgroup's life).
    constexpr bool kSlopeOnDemand = PPB == 2;
    if (!slopes && !kSlopeOnDemand) {
        // the interval slopes (t1 - t0)/(d1 - d0) of SM:311-317 from the staged distances — the expression k_lut uses, so
        // the same numbers, without 8 KB per path going to HBM and back (config 5: a fifth of the step's traffic)
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < PPB; pp++) {
            if (p_live[pp] && tile0 * tile < p_N[pp]) {
                const double *sDp = sD_all + pp * kLutN;
                for (int j = tid; j < kLutN; j += kSampleThreads) {
                    double w = 0.0;
                    if (j > 0) {
                        const double t0 = (double)(j - 1) * p_lstep[pp], t1 = (j == kLutN - 1) ? p_tmax[pp] : (double)j * p_lstep[pp];
                        w = div_inrange(t1 - t0, sDp[j] - sDp[j - 1]);
                    }
                    sWt_all[pp * kLutN + j] = w;
                }
            }
        }
    }
    __syncthreads();
    asm volatile("" ::"v"(run_touch));
    const long long ts1 = stats ? __builtin_amdgcn_s_memtime() : 0;
    const double end_param = (double)(W - 1);
    const int tab_n = W * kSamplesPerNode;
    int tiles_done = 0;   // parity of the neighbour exchange

#pragma unroll 1
    for (int pp = 0; pp < PPB; pp++) {
    // (the path's scalars again, from the scalar cache: carrying them through the tile loop as arrays indexed by pp
    // costs registers the tile body does not have)
    const int bq = blockIdx.y * PPB + pp;
    if (bq >= B) break;
    const int b = bq;
    const double *m = meta + (size_t)b * kMetaStride;
    const double t_max = m[0], total = m[1], dd = m[2];
    const int N = (int)m[3];
    const double *ax = aux + (size_t)b * kAuxStride;
    const double lstep = ax[0], tstep = ax[1], inv_tstep = ax[2];
    const int n_runs = (int)ax[3];
    const size_t row = (size_t)b * S;
    const double *sD = sD_all + pp * kLutN, *sWt = sWt_all + pp * kLutN;
    const double *pw = power + (size_t)b * G * kCoefDoubles;
    const double *coef = COEF_LDS ? s_coef + (size_t)pp * G * kCoefDoubles : pw;
    const double *run_tab = runs + (size_t)b * kGridRunDoubles;
    const double sk_first = PPB == 1 ? p_sk[0] : (pp == 0 ? p_sk[0] : p_sk[PPB - 1]);
    auto grid_first = [&](int kb0) { return grid_first_of(run_tab, dd, N, n_runs, kb0); };
    for (int tl = 0; tl < tiles_per_block; tl++) {
        const int k0 = (tile0 + tl) * tile;
        if (k0 >= S) break;
        const int kbase = k0 + tid * kSPT;
        auto store_vec = [&](OT *dst, const OT v[kSPT]) {
            if (!dst || !writer) return;
            if (aligned && kbase + kSPT <= S) {
                if constexpr (sizeof(OT) == 4) {
                    if constexpr (HI) {
                        // the caller's rows leave non-temporally: nothing on the device reads them again before the whole
                        // batch has been written, and the fp64 side rows the velocity kernel is about to read keep the
                        // cache (MALL) to themselves
                        using f4 = float __attribute__((ext_vector_type(4)));
                        f4 w = {v[0], v[1], v[2], v[3]};
                        __builtin_nontemporal_store(w, reinterpret_cast<f4 *>(dst + row + kbase));
                    } else {
                        *reinterpret_cast<float4 *>(dst + row + kbase) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else {
                    *reinterpret_cast<double2 *>(dst + row + kbase) = make_double2(v[0], v[1]);
                    *reinterpret_cast<double2 *>(dst + row + kbase + 2) = make_double2(v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < kSPT; i++)
                    if (kbase + i < S) dst[row + kbase + i] = v[i];
            }
        };
        auto store_hi = [&](double *dst, const double v[kSPT]) {   // rows of S doubles: 16-byte aligned when S is even
            if (!writer) return;
            if ((S & 1) == 0 && kbase + kSPT <= S) {
                *reinterpret_cast<double2 *>(dst + row + kbase) = make_double2(v[0], v[1]);
                *reinterpret_cast<double2 *>(dst + row + kbase + 2) = make_double2(v[2], v[3]);
            } else {
#pragma unroll
                for (int i = 0; i < kSPT; i++)
                    if (kbase + i < S) dst[row + kbase + i] = v[i];
            }
        };
        if (k0 >= N) {
            // past this path's grid (ragged dd mode): zero-fill so every output element is defined
            const OT z[kSPT] = {(OT)0, (OT)0, (OT)0, (OT)0};
            store_vec(ox, z); store_vec(oy, z); store_vec(oh, z); store_vec(ok, z); store_vec(odth, z);
            if constexpr (HI) {
                const double zd[kSPT] = {0.0, 0.0, 0.0, 0.0};
                store_hi(ok64, zd); store_hi(odth64, zd);
            }
            continue;
        }
        // interior tiles (every sample and its right-hand neighbour before the path's last sample, not the first tile, rows
        // 16-byte aligned): the same body without the end-of-path selects, chosen per tile — wave-uniform
        auto tile_body = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
        OT vx[kSPT], vy[kSPT], vh[kSPT], vk[kSPT], vd[kSPT];
        double vd64[HI ? kSPT : 1];
        [[maybe_unused]] double kap_even = 0.0;   // HI: the fp64 curvature row leaves in pairs as soon as a pair exists
        // MPG:112-122 distance grid: the reference accumulates current_dist += dd.  The thread's first
        // sample comes from the path's run table (that sum in closed form), the following ones by the
        // reference's own addition.  The wave's kSPT*64 consecutive samples almost always lie in one run,
        // or in three (a binade boundary: the old run, the boundary element, the new run).
        double sk = tl == 0 ? sk_first : grid_first(kbase);
        double d1x[kSPT], d1y[kSPT];
        int jjv[kSPT];
        int idx = 0;
        // Phase A, the four samples' parameters and table entries side by side; the redo of a sample that sits a few ulps
        // from a decision point (rare) is ONE block after them instead of a divergent branch inside every sample
        double tA[kSPT], sA[kSPT];
        int idxA[kSPT];
        bool redo[kSPT], any_redo = false;
#pragma unroll
        for (int i = 0; i < kSPT; i++) {
            const int k = INTERIOR ? kbase + i : (kbase + i < N - 1 ? kbase + i : N - 1);
            if (i > 0) sk = sk + dd;
            const double s = INTERIOR ? sk : ((k == N - 1) ? total : sk);
            if (i == 0) idx = lut_search_left(sD, s);
            else while (idx < kLutN - 1 && sD[idx] < s) idx++;
            if constexpr (!INTERIOR) idx = idx < 1 ? 1 : idx;
            const double d0 = sD[idx - 1];
            const double t0 = (double)(idx - 1) * lstep;
            const bool exact = INTERIOR ? false : (s <= 0.0 || s >= total);
            double wslope;
            if (kSlopeOnDemand && !slopes) {
                const double t1 = (idx == kLutN - 1) ? t_max : (double)idx * lstep;
                wslope = div_inrange(t1 - t0, sD[idx] - d0);
            } else {
                wslope = sWt[idx];
            }
            double t = fma(wslope, s - d0, t0);
            if constexpr (!INTERIOR) t = s >= total ? end_param : t;
            bool near;
            jjv[i] = table_index_fast(t, tab_n, inv_tstep, near);
            tA[i] = t;
            sA[i] = s;
            idxA[i] = idx;
            redo[i] = near && !exact;
            any_redo |= redo[i];
        }
        if (__builtin_expect(any_redo, 0)) {
#pragma unroll
            for (int i = 0; i < kSPT; i++) {
                if (redo[i]) {   // the reference's own rounding (SM:311-317)
                    const int ix = idxA[i];
                    const double d0 = sD[ix - 1], t0 = (double)(ix - 1) * lstep;
                    const double t1 = (ix == kLutN - 1) ? t_max : (double)ix * lstep;
                    tA[i] = t0 + (t1 - t0) * (sA[i] - d0) / (sD[ix] - d0);
                    jjv[i] = table_index(tA[i], tab_n, end_param);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kSPT; i++) {
            const double t = tA[i];
            const int jj = jjv[i];
            const double tp = (jj == tab_n - 1) ? end_param : (double)jj * tstep;
            double lt;
            int sg;
            normalize_inside(tp, G, lt, sg);
            const double *c = coef + sg * kCoefDoubles;
            const double ex = horner4(c + kCoefD1, lt), ey = horner4(c + kCoefD1 + 5, lt);     // P'
            const double fx = horner3(c + kCoefD2, lt), fy = horner3(c + kCoefD2 + 4, lt);     // P''
            const double ss = fma(ex, ex, ey * ey);                               // SM:517
            const double num = fma(ex, fy, -(ey * fx));                           // SM:523
            const double kap = (ss >= 1e-10) ? curvature_of(num, ss) : 0.0;       // SM:526-527
            vk[i] = (OT)kap;
            if constexpr (HI) {
                vd64[i] = 0.0;
                if ((i & 1) == 0) {
                    kap_even = kap;
                } else if (writer) {
                    const int ke = kbase + i - 1;
                    const double v0 = (INTERIOR || ke < N) ? kap_even : 0.0, v1 = (INTERIOR || ke + 1 < N) ? kap : 0.0;
                    if (INTERIOR || ((S & 1) == 0 && ke + 2 <= S)) {
                        *reinterpret_cast<double2 *>(ok64 + row + ke) = make_double2(v0, v1);
                    } else {
                        if (ke < S) ok64[row + ke] = v0;
                        if (ke + 1 < S) ok64[row + ke + 1] = v1;
                    }
                }
            }
            vh[i] = heading_of<OT>(ey, ex);                                       // SM:536
            d1x[i] = ex;
            d1y[i] = ey;
            jjv[i] = jj;
            // SM:204-215 get_point_at_parameter(t) at the sample's own parameter
            normalize_inside(t, G, lt, sg);
            c = coef + sg * kCoefDoubles;
            if constexpr (sizeof(OT) == 4) {
                const float *cf = reinterpret_cast<const float *>(c + kCoefPf);
                const float ltf = (float)lt;
                vx[i] = horner5f(cf, ltf);
                vy[i] = horner5f(cf + 6, ltf);
            } else {
                vx[i] = horner5(c + kCoefP, lt);
                vy[i] = horner5(c + kCoefP + 6, lt);
            }
            vd[i] = (OT)0;
        }
        const int pb = tiles_done & 1;
        tiles_done++;
        s_dx[pb][tid] = d1x[0];
        s_dy[pb][tid] = d1y[0];
        s_j[pb][tid] = jjv[0];
        s_th[pb][tid] = vh[0];
        __syncthreads();
        if (writer) {
            // |heading[k+1] - heading[k]| of the reference's raw (un-unwrapped) atan2 values
            if constexpr (HI && sizeof(OT) == 4) {
                // the four samples side by side, no branch per sample: the series for all of them, ONE rarely taken block
                // for samples outside its range, then the 2*pi multiples (the same operations as dtheta_f64)
                double dl[kSPT];
                bool act[kSPT], gen[kSPT];
                double nxv[kSPT], nyv[kSPT];
                OT nthv[kSPT];
                bool any_gen = false;
#pragma unroll
                for (int i = 0; i < kSPT; i++) {
                    const int k = kbase + i;
                    nxv[i] = (i + 1 < kSPT) ? d1x[(i + 1) % kSPT] : s_dx[pb][tid + 1];
                    nyv[i] = (i + 1 < kSPT) ? d1y[(i + 1) % kSPT] : s_dy[pb][tid + 1];
                    const int nj = (i + 1 < kSPT) ? jjv[(i + 1) % kSPT] : s_j[pb][tid + 1];
                    nthv[i] = (i + 1 < kSPT) ? vh[(i + 1) % kSPT] : s_th[pb][tid + 1];
                    act[i] = (INTERIOR || k < N - 1) && nj != jjv[i];
                    dl[i] = dtheta_f64_series(d1x[i], d1y[i], nxv[i], nyv[i], gen[i]);
                    gen[i] = gen[i] && act[i];
                    any_gen |= gen[i];
                }
                if (__builtin_expect(any_gen, 0)) {
#pragma unroll
                    for (int i = 0; i < kSPT; i++)
                        if (gen[i]) dl[i] = dtheta_f64_general(d1x[i], d1y[i], nxv[i], nyv[i]);
                }
#pragma unroll
                for (int i = 0; i < kSPT; i++) {
                    const double v = dtheta_f64_wrap(dl[i], vh[i], nthv[i]);
                    vd64[i] = act[i] ? v : vd64[i];
                    vd[i] = (INTERIOR || kbase + i < N - 1) ? (OT)vd64[i] : vd[i];
                }
            } else {
#pragma unroll
            for (int i = 0; i < kSPT; i++) {
                const int k = kbase + i;
                if (INTERIOR || k < N - 1) {
                    const double nx = (i + 1 < kSPT) ? d1x[(i + 1) % kSPT] : s_dx[pb][tid + 1];
                    const double ny = (i + 1 < kSPT) ? d1y[(i + 1) % kSPT] : s_dy[pb][tid + 1];
                    const int nj = (i + 1 < kSPT) ? jjv[(i + 1) % kSPT] : s_j[pb][tid + 1];
                    const OT nth = (i + 1 < kSPT) ? vh[(i + 1) % kSPT] : s_th[pb][tid + 1];
                    if constexpr (sizeof(OT) == 8) {
                        vd[i] = fabs(nth - vh[i]);
                    } else if constexpr (HI) {
                        if (nj != jjv[i]) vd64[i] = dtheta_f64(d1x[i], d1y[i], nx, ny, vh[i], nth);
                        vd[i] = (OT)vd64[i];   // (only stored when the staged API asked for the fp32 row as well)
                    } else {
                        if (nj != jjv[i]) vd[i] = dtheta_f32(d1x[i], d1y[i], nx, ny, vh[i], nth);
                    }
                }
            }
            }
            if (!INTERIOR && k0 + kSampleChunk > N) {   // only the path's last tile has samples to blank
#pragma unroll
                for (int i = 0; i < kSPT; i++)
                    if (kbase + i >= N) {
                        vx[i] = vy[i] = vh[i] = vk[i] = vd[i] = (OT)0;
                        if constexpr (HI) vd64[i] = 0.0;
                    }
            }
            store_vec(ox, vx); store_vec(oy, vy); store_vec(oh, vh); store_vec(ok, vk);
            if constexpr (HI) store_hi(odth64, vd64);
            store_vec(odth, vd);
        }
            };
        const bool interior = k0 > 0 && k0 + kSampleThreads * kSPT < N - 1 && aligned && (S & 1) == 0 && k0 + kSampleThreads * kSPT <= S;
        if (interior) tile_body(std::true_type{});
        else tile_body(std::false_type{});
    }
    }   // paths of this workgroup
    if (stats && (tid & 63) == 0) {
        long long *st = stats + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6)) * 4;
        st[0] = ts1 - ts0;                                 // staging + barrier
        st[1] = __builtin_amdgcn_s_memtime() - ts1;        // all tiles
        st[2] = 0;
        st[3] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// K5a: forward/backward velocity pass, one lane per path, strictly sequential — MPG:188-311 in
// squared-velocity space.  FAST = false is the literal statement (every min/max of the reference);
// FAST = true uses the collapsed limits of vap_device.h and is bit-identical to the relaxation
// kernel below, which makes it the in-library check of that kernel.
// ------------------------------------------------------------------------------------------------
// Per-sample max_acceleration for routes whose nodes / action points change it (all NULL: the constraints'):
//   fwd [B][S]  max_acc (= max_dec) in force for the forward step FROM sample i     MPG:194-196
//   bwd [B][S]  max_acc the backward sweep has in force for its step FROM sample i  MPG:256-257
//   dec [B]     max_dec of the whole backward sweep (what the forward sweep left)
// (AccRows<R>: vap_device.h)

// R = arithmetic type (and type of the curvature / dtheta rows), IO = type of the caller's rows (initial
// velocities, max_acceleration rows, the velocity output): IO = float with R = double is the fp64 recurrence
// behind fp32 outputs.
template <typename R, typename IO, bool FAST>
__global__ __launch_bounds__(64) void k_velocity_seq(int B, int S, VelConsts<R> c, R start_u, R end_u,
                                                     const double *__restrict__ meta,
                                                     const R *__restrict__ curv, const R *__restrict__ dtheta,
                                                     const R *__restrict__ vcap, AccRows<R> acc, IO *__restrict__ vel,
                                                     R *__restrict__ usq)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *m = meta + (size_t)b * kMetaStride;
    const R twodd = (R)2 * (R)m[2];
    const int N = (int)m[3];
    const size_t row = (size_t)b * S;
    const R *K = curv + row, *DT = dtheta + row;
    // the sweeps run on squared velocities in the arithmetic type: in place when the output row has that type,
    // otherwise in the scratch row `usq`
    R *V;
    if constexpr (std::is_same<R, IO>::value) V = vel + row;
    else V = usq + row;
    const R vmax2 = c.vmax * c.vmax;
    const FastConsts<R> fc = make_fast(c, twodd);
    // same per-path decision as the relaxation kernel: does any step have a zero heading difference?
    bool dup = false;
    if constexpr (FAST) {
        for (int i = 0; i < N - 1; i++) {
            const R kabs = (R)fabs(K[i]);
            dup |= fast_gq(fast_gg(fc, DT[i]), kabs * kabs) < (R)0;
        }
    }
    // Both sweeps read the rows of step i + 1 while step i is computed (a lane waits a memory round trip per step
    // otherwise: the chain of a step is a few hundred cycles, a load from L2 about as long again).
    // forward, MPG:188-249
    R u = start_u, wprev = (R)0;   // FAST: wprev carries the previous squared velocity instead
    V[0] = u;
    auto load_fwd = [&](int j, R &k, R &dt, R &cap, R &a) {      // what forward step j reads
        k = K[j];
        dt = DT[j];
        cap = vcap ? (R)vcap[row + j + 1] : (R)0;
        a = acc.fwd ? (R)acc.fwd[row + j] : c.amax;
    };
    R k1 = (R)0, d1 = (R)0, c1 = (R)0, a1 = c.amax, kprev = (R)0;
    if (N > 1) load_fwd(0, k1, d1, c1, a1);
    for (int i = 0; i < N - 1; i++) {
        const R k0 = k1, d0 = d1, c0 = c1, cur = a1;
        load_fwd(i + 1 < N - 1 ? i + 1 : i, k1, d1, c1, a1);
        const R un = (i + 1 == N - 1) ? end_u : (vcap ? c0 * c0 : vmax2);
        // MPG:194-196: max_acc = max_dec = the value in force from the last boundary at or before sample i (`cur`)
        if constexpr (FAST) {
            R rho, gq, A, cap;
            const R amaxp = acc.fwd ? twodd * cur : fc.amaxp;
            fast_derive(fc, (R)fabs(k0), i > 0 ? (R)fabs(kprev) : (R)0, d0, amaxp, rho, gq, A, cap);
            if (acc.fwd) A = fast_cap_A(fc, (R)fabs(k0), A);
            R am, g;
            fast_scale(amaxp, gq, A, am, g);
            u = fast_forward_a(am, rho, g, A, cap, u, wprev, un);
        } else {
            const SampleLimits<R> L = sample_limits(c, (R)fabs(k0), cur, acc.fwd ? cur : c.adec);
            u = forward_step(c, L, cur, twodd, u, wprev, d0, un);
        }
        kprev = k0;
        V[i + 1] = u;
    }
    // backward, MPG:251-311
    u = end_u;
    wprev = (R)0;
    // MPG:256-257: max_acc as the backward sweep finds it at sample i; max_dec is what the forward sweep left
    const R cur_dec = acc.dec ? (R)acc.dec[b] : c.adec;
    auto load_bwd = [&](int j, R &k, R &dt, R &vf, R &a) {      // what backward step j (>= 1) reads
        k = K[j];
        dt = DT[j - 1];
        vf = V[j - 1];
        a = acc.bwd ? (R)acc.bwd[row + j] : c.amax;
    };
    R v1 = (R)0, knext = (R)0;
    if (N > 1) load_bwd(N - 1, k1, d1, v1, a1);
    for (int i = N - 1; i > 0; i--) {
        R up;
        const R k0 = k1, d0 = d1, vf = v1, cur_acc = a1;
        load_bwd(i - 1 > 0 ? i - 1 : i, k1, d1, v1, a1);
        if constexpr (FAST) {
            R rho, gq, A, cap;
            const R kabs = (R)fabs(k0);
            fast_derive(fc, kabs, i + 1 <= N - 1 ? (R)fabs(knext) : (R)0, d0, acc.bwd ? twodd * cur_dec : fc.adecp, rho, gq, A, cap);
            if (acc.bwd) A = fast_cap_A(fc, kabs, A);
            // a straight sample is limited by max_dec alone (MPG:270-272), and so is a sample with a zero heading
            // difference whose angular velocity does not rise (the wheel limit is then +inf, MPG:52-59)
            // (amaxp = A there: the clamp decides, and kHuge times any non-zero rise of the angular velocity still wins)
            const R amaxp = acc.bwd ? ((kabs < (R)1e-6 || gq < (R)0) ? A : twodd * cur_acc) : fc.amaxp;
            R am, g;
            fast_scale(amaxp, gq, A, am, g);
            up = dup ? fast_backward_a<true>(am, rho, g, A, cap, u, wprev, vf)
                     : fast_backward_a<false>(am, rho, g, A, cap, u, wprev, vf);
        } else {
            const SampleLimits<R> L = sample_limits(c, (R)fabs(k0), cur_acc, cur_dec);
            up = backward_step(c, L, cur_acc, twodd, u, wprev, d0, vf);
        }
        knext = k0;
        // (R != IO: V[i] — the scratch row, the forward value of sample i — is dead from here on and receives the
        // velocity in the arithmetic type: the fp64 row the time-domain resample integrates behind fp32 rows)
        const R vv = vel_sqrt(u);
        vel[row + i] = (IO)vv;
        if constexpr (!std::is_same<R, IO>::value) V[i] = vv;
        u = up;
    }
    {
        const R vv = vel_sqrt(u);
        vel[row] = (IO)vv;
        if constexpr (!std::is_same<R, IO>::value) V[0] = vv;
    }
    for (int i = N; i < S; i++) {
        vel[row + i] = (IO)0;
        if constexpr (!std::is_same<R, IO>::value) V[i] = (R)0;
    }
}

// ------------------------------------------------------------------------------------------------
// K5b: velocity pass by speculative chunk relaxation.  One workgroup per path; thread c owns the L
// consecutive samples [c*L, c*L+L) and keeps their step coefficients (rho, g*k^2, A, cap — vap_device.h)
// and squared velocities in registers.  The recurrence is sequential only through the 2-word state
// (u_i, u_{i-1}) that crosses a chunk boundary, so a thread whose incoming state changed re-runs its L
// steps from that state and hands its outgoing state on; a thread whose incoming state is bit-identical
// to the one it last used is already final.  Chunk 0's incoming state is exact, so by induction chunk c
// is exact after at most c+1 evaluations, and the fixed point (no state changed) is bit-identical to
// the sequential sweep.  Two trajectories started from different states merge (bitwise) after ~240
// samples on average, so config 3 needs ~29 + 23 evaluations by the busiest wavefront, not 250
// (DESIGN.md §5, K5).
// States move lane to lane by DPP inside a wavefront and through a 2-word LDS record per wavefront
// across wavefronts, one workgroup barrier per 8 evaluations.  HBM traffic: one read of (curvature,
// dtheta) per sweep direction and the final velocity store.
// ------------------------------------------------------------------------------------------------
template <typename R> struct BitsOf;
template <> struct BitsOf<float> { using type = uint32_t; };
template <> struct BitsOf<double> { using type = uint64_t; };
template <typename R>
__device__ __forceinline__ bool same_bits(R a, R b)
{
    using U = typename BitsOf<R>::type;
    return __builtin_bit_cast(U, a) == __builtin_bit_cast(U, b);
}

// Cooperative, fully coalesced copy of one row (n elements) between HBM and the padded LDS stage:
// element i lives at stage[i + i / L], so thread c's chunk [c*L, c*L+L) is read with lane stride
// L+1 words (odd => no bank conflicts) while HBM sees 16 bytes per lane.
template <typename R, int L>
__device__ __forceinline__ int stage_pos(int i) { return i + i / L; }

// One thread's share of a row on its way from HBM to the stage.  Fetch and put are separate so that a
// row's loads can be in flight while the previous row is being consumed.
template <typename R, int L>
struct RowRegs {
    static constexpr int V = 16 / sizeof(R);
    static constexpr int ITER = (L + V - 1) / V + 1;   // covers cap_n <= T*L + V elements
    using VT = typename std::conditional<sizeof(R) == 4, float4, double2>::type;
    VT v[ITER];      // rows that start 16-byte aligned: 16 bytes per lane
    R s[L + 1];      // other rows: one element per lane
};

template <typename R, int L>
__device__ __forceinline__ void row_fetch(RowRegs<R, L> &r, const R *__restrict__ src, int n, bool aligned, int tid, int T)
{
    using RR = RowRegs<R, L>;
    constexpr int V = RR::V;
    if (aligned) {
        const typename RR::VT *src4 = reinterpret_cast<const typename RR::VT *>(src);
        // all loads first (independent, so the memory latency is paid once); the one group that straddles
        // the end of the row is assembled element by element, zero-padded
#pragma unroll
        for (int it = 0; it < RR::ITER; it++) {
            const int i = tid + it * T;
            if ((i + 1) * V <= n) {
                r.v[it] = src4[i];
            } else {
                R *e = reinterpret_cast<R *>(&r.v[it]);
#pragma unroll
                for (int k = 0; k < V; k++) e[k] = (i * V + k < n) ? src[i * V + k] : (R)0;
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < L + 1; it++) {
            const int i = tid + it * T;
            r.s[it] = i < n ? src[i] : (R)0;
        }
    }
}

template <typename R, int L>
__device__ __forceinline__ void stage_put(R *__restrict__ stage, const RowRegs<R, L> &r, int cap_n, bool aligned, int tid, int T)
{
    using RR = RowRegs<R, L>;
    constexpr int V = RR::V;
    if (aligned) {
#pragma unroll
        for (int it = 0; it < RR::ITER; it++) {
            const int i = tid + it * T;
            const R *e = reinterpret_cast<const R *>(&r.v[it]);
            if (i * V < cap_n) {   // the stage has room for a whole group past cap_n (T*L + T + 8 words)
                const int p0 = stage_pos<R, L>(i * V);
#pragma unroll
                for (int k = 0; k < V; k++)   // L % V == 0: the V elements share a chunk
                    stage[(L % V == 0) ? p0 + k : stage_pos<R, L>(i * V + k)] = e[k];
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < L + 1; it++) {
            const int i = tid + it * T;
            if (i < cap_n) stage[stage_pos<R, L>(i)] = r.s[it];
        }
    }
}

template <typename R, int L>
__device__ __forceinline__ void stage_load(R *__restrict__ stage, const R *__restrict__ src, int n, int cap_n,
                                           bool aligned, int tid, int T)
{
    RowRegs<R, L> r;
    row_fetch<R, L>(r, src, n, aligned, tid, T);
    stage_put<R, L>(stage, r, cap_n, aligned, tid, T);
}

// Boundary state of the recurrence between two chunks: the last two squared velocities.
template <typename R>
struct alignas(2 * sizeof(R)) BoundaryState {
    R u, w;
};

// Whole-wave lane shifts by DPP (no LDS round trip): lane i receives lane i-1 (up) / i+1 (down);
// lane 0 (up) / lane 63 (down) keep their own value.
__device__ __forceinline__ float wave_shift_up(float x)
{
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shift_down(float x)
{
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_shift_up(double x)
{
    const uint64_t v = __builtin_bit_cast(uint64_t, x);
    const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((uint64_t)h2 << 32) | l2);
}
__device__ __forceinline__ double wave_shift_down(double x)
{
    const uint64_t v = __builtin_bit_cast(uint64_t, x);
    const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((uint64_t)h2 << 32) | l2);
}

// VCAP: the caller gave per-sample initial velocities (MPG:121,127,153,172: node / action-point max_velocity and
// stops); the forward step into sample j is then also limited by vcap[j]^2 — folded into that slot's cap.
// R = arithmetic type and type of the curvature / dtheta rows; IO = type of the caller's rows (vcap, acc, vel).
// backward step of either form: FOLDED = the forward value of the sample is already folded into cap (commit mode)
template <bool DUP, bool FOLDED>
__device__ __forceinline__ float bwd_step(float am, float rho, float g, float A, float cap, float u, float &uprev, float u_prev)
{
    return fast_backward_a<DUP>(am, rho, g, A, cap, u, uprev, FOLDED ? Huge<float>::v : u_prev);   // (one v_min3 either way)
}
template <bool DUP, bool FOLDED>
__device__ __forceinline__ double bwd_step(double am, double rho, double g, double A, double cap, double u, double &uprev, double u_prev)
{
    return fast_backward_a<DUP, !FOLDED>(am, rho, g, A, cap, u, uprev, u_prev);
}

template <typename R, typename IO, int L, int MAXT, int MINW, bool VCAP, bool ACC>
__global__ __launch_bounds__(MAXT, MINW) void k_velocity_relax(int S, VelConsts<R> c, R start_u, R end_u,
                                                               const double *__restrict__ meta,
                                                               const R *__restrict__ curv,
                                                               const R *__restrict__ dtheta,
                                                               const R *__restrict__ vcap, AccRows<R> acc,
                                                               IO *__restrict__ vel, uint32_t *__restrict__ flags,
                                                               long long *__restrict__ stats, R *__restrict__ vhi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *stage = reinterpret_cast<R *>(smem_raw);   // (T*L + T + 2) elements
    // boundary states, double-buffered by round parity so one barrier per round suffices
    __shared__ BoundaryState<R> s_bs[2][MAXT / 64 + 2];   // states crossing a wavefront boundary
    __shared__ int s_any[3];   // "some chunk's incoming state changed" per round, rotating slots
    __shared__ int s_dup;
    const long long t_start = stats ? __builtin_amdgcn_s_memtime() : 0;
    const int b = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    const double *m = meta + (size_t)b * kMetaStride;
    const R twodd = (R)2 * (R)m[2];
    const int N = (int)m[3];
    const size_t row = (size_t)b * S;
    const R *K = curv + row, *DT = dtheta + row;
    const FastConsts<R> fc = make_fast(c, twodd);
    const R adecp_b = ACC ? twodd * (R)acc.dec[b] : fc.adecp;   // ACC: the backward sweep's max_dec (AccRows)
    const int lo = tid * L;
    const int TL = T * L;
    const bool aligned = (S % (16 / (int)sizeof(R))) == 0;
    // the step's `am` (fast_scale, vap_device.h) per slot: fp32 only for routes whose nodes change max_acceleration
    // (ACC), the scaled fp64 step always
    constexpr bool PSA = ACC || ScaledStep<R>::value;
    // CM ("commit mode", the fp64 instantiations — registers): no array of squared velocities.  The forward rounds
    // only move boundary states and one commit evaluation writes the forward result over cp[] (the cap is dead by
    // then); the backward sweep folds it into its own cap, min(cap, forward value), and commits into cp[] again.
    constexpr bool CM = ScaledStep<R>::value;
    R q[L], g[L], A[L], cp[L], u[CM ? 1 : L];
    R am[PSA ? L : 1];
    if (tid == 0) { s_any[0] = 0; s_any[1] = 0; s_any[2] = 0; s_dup = 0; }

    // ---------------- forward sweep: the step (j-1 -> j) into owned sample j uses k[j-1], dth[j-1]
    // (and k[j-2] for rho).  q[] holds rho, g[] holds g*k^2 (vap_device.h) once both rows are consumed.
    // Stage positions of this thread's window: sample lo + k sits at cbase + k for 0 <= k < L, one word
    // further for the next chunk's samples and one word nearer for the previous chunk's.
    const int cbase = tid * (L + 1);
    auto cpos = [&](int k) { return k < 0 ? cbase + k - 1 : (k < L ? cbase + k : cbase + k + 1); };
    // rows are read up to the capacity S, not the sample count N (samples past N hold zeros): the loads
    // then do not wait for the meta record
    stage_load<R, L>(stage, K, S < TL ? S : TL, TL, aligned, tid, T);
    RowRegs<R, L> rd;   // the dtheta row is fetched while the curvature row is consumed
    row_fetch<R, L>(rd, DT, S < TL ? S : TL, aligned, tid, T);
    __syncthreads();
    // Samples are handled BK at a time: BK unconditional LDS reads in one batch (one wait), then
    // straight-line arithmetic, one sample after the other (the opaque() keeps the scheduler from
    // interleaving all L of them, which costs registers the rounds need).
    constexpr int BK = (L % 8 == 0) ? 8 : ((L % 5 == 0) ? 5 : 4);
    {
        // k[lo-2], k[lo-1] first (tid 0 has no previous chunk: clamped to word 0 and masked)
        R kp = (R)fabs(stage[(lo >= 2) ? cpos(-2) : 0]);
        R kc = (R)fabs(stage[(lo >= 1) ? cpos(-1) : 0]);
#pragma unroll
        for (int s0 = 0; s0 < L; s0 += BK) {
            R kn[BK];
#pragma unroll
            for (int i = 0; i < BK; i++) kn[i] = (R)fabs(stage[cpos(s0 + i)]);
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int s = s0 + i, j = lo + s;
                const bool valid = j >= 1 && j <= N - 1;
                R base = fc.amaxp;
                if constexpr (ACC) {   // the step into j starts at sample j-1: its max_acc (= max_dec), MPG:194-196
                    base = valid ? twodd * (R)acc.fwd[row + j - 1] : fc.amaxp;
                    am[s] = base;
                }
                fast_derive_k(fc, kc, (j >= 2) ? kp : (R)0, base, q[s], g[s], A[s], cp[s]);   // g[s] = k^2 for now
                if constexpr (ACC) A[s] = fast_cap_A(fc, kc, A[s]);
                if (!valid) idle_coef(q[s], g[s], A[s], cp[s]);
                q[s] = opaque(q[s]);
                kp = kc;
                kc = kn[i];
            }
        }
    }
    __syncthreads();
    stage_put<R, L>(stage, rd, TL, aligned, tid, T);
    __syncthreads();
    bool dup = false;
#pragma unroll
    for (int s0 = 0; s0 < L; s0 += BK) {
        R dn[BK];
#pragma unroll
        for (int i = 0; i < BK; i++) dn[i] = stage[(lo + s0 + i >= 1) ? cpos(s0 + i - 1) : 0];
#pragma unroll
        for (int i = 0; i < BK; i++) {
            const int s = s0 + i, j = lo + s;
            const bool valid = j >= 1 && j <= N - 1;
            const R gq = fast_gq(fast_gg(fc, dn[i]), g[s]);
            R amv, gv;
            fast_scale(ACC ? am[ACC ? s : 0] : fc.amaxp, valid ? gq : (R)0, A[s], amv, gv);
            g[s] = opaque(gv);
            if constexpr (PSA) am[s] = amv;
            dup |= g[s] < (R)0;
            if constexpr (!CM) u[s] = start_u;
        }
    }
    if constexpr (VCAP) {
        // the forward step into sample j (1 <= j <= N-2; the end sample is fixed by the backward sweep) also
        // honours the sample's initial velocity: min(.., cap, vcap^2) — one number per slot
        const R *VC = vcap + row;
#pragma unroll
        for (int s0 = 0; s0 < L; s0 += BK) {    // BK loads in flight, then BK selects (as the phases above)
            R vc[BK];
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int j = lo + s0 + i;
                vc[i] = (j >= 1 && j <= N - 2) ? (R)VC[j] : Huge<R>::v * (R)1e-20;   // (squares without overflow)
            }
#pragma unroll
            for (int i = 0; i < BK; i++) cp[s0 + i] = opaque(vmin(cp[s0 + i], vc[i] * vc[i]));
        }
    }
    // boundary state = the last two squared velocities (u, u_prev)
    R in_u, in_w;
    if (tid == 0) { in_u = start_u; in_w = (R)0; }
    else { in_u = cp[0]; in_w = in_u; }
    const bool fwd_active = lo <= N - 1;       // the chunk holds at least one real sample
    bool need = fwd_active;
    R out_u = in_u, out_w = in_w;
    int rounds = 0;
    if (dup) s_dup = 1;
    __syncthreads();
    const bool any_dup = s_dup != 0;
    const long long t_fwd0 = stats ? __builtin_amdgcn_s_memtime() : 0;
    // A round: up to kInner times { lanes whose incoming state changed re-run their chunk; the outgoing
    // states move one lane up inside each wavefront by DPP }, then the states that cross a wavefront
    // boundary go through LDS (buffer r&1) behind one workgroup barrier, picked up together with the
    // "someone still has work" flag of round r-1.  The flags rotate over three slots (this round's, the
    // next one's being cleared, the previous one's being read), so termination lags one cheap round and
    // needs no second barrier.  Chains are ~6 chunks long on average: most end inside one round.
    constexpr int kInner = 8;
    const int wv = tid >> 6, lane = tid & 63;
    int f_cur = 0, f_nxt = 1, f_prv = 2;
    const bool fwd_nb = lane > 0 && fwd_active;
    while (true) {
        if constexpr (CM) {
            // Commit mode keeps no per-sample result, so re-running a chunk whose incoming state has not changed is
            // harmless: the whole wavefront evaluates whenever any of its lanes has to (no per-lane branch, no exec
            // bookkeeping between an evaluation and the next — ~35 instructions on the chain of a ramp otherwise), and
            // the loop is left on the ballot of the lanes whose incoming state moved.
            if (__ballot(need) != 0) {
#pragma unroll 1
                for (int k = 0; k < kInner; k++) {
                    R uu = in_u, wp = in_w;
#pragma unroll
                    for (int s = 0; s < L; s++) {
                        const R nx = step_fwd(am[PSA ? s : 0], q[s], g[s], A[s], cp[s], uu, wp);
                        if (s == 0) {   // sample 0 is the given start velocity (MPG:189): thread 0 skips its first slot
                            const R w1 = wp;               // (step_fwd has moved uu into wp)
                            wp = tid == 0 ? in_w : w1;
                            uu = tid == 0 ? in_u : nx;
                        } else {
                            uu = nx;
                        }
                    }
                    out_u = uu;
                    out_w = wp;
                    const R nu = wave_shift_up(out_u), nw = wave_shift_up(out_w);
                    need = fwd_nb && !(same_bits(nu, in_u) && same_bits(nw, in_w));
                    in_u = fwd_nb ? nu : in_u;
                    in_w = fwd_nb ? nw : in_w;
                    if (__ballot(need) == 0) break;
                }
            }
        } else {
#pragma unroll 1
        for (int k = 0; k < kInner; k++) {
            if (need) {
                R uu = in_u, wp = in_w;
#pragma unroll
                for (int s = 0; s < L; s++) {
                    if (s == 0 && tid == 0) continue;   // sample 0 is the given start velocity (MPG:189)
                    uu = step_fwd(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp);
                    if constexpr (!CM) u[s] = uu;
                }
                out_u = uu;
                out_w = wp;
            }
            const R nu = wave_shift_up(out_u), nw = wave_shift_up(out_w);
            need = false;
            if (lane > 0 && fwd_active) {
                need = !(same_bits(nu, in_u) && same_bits(nw, in_w));
                in_u = nu;
                in_w = nw;
            }
            if (__ballot(need) == 0) break;
        }
        }
        const int pb = rounds & 1;
        if (lane == 63) s_bs[pb][wv + 1] = BoundaryState<R>{out_u, out_w};
        if (tid == 0) s_any[f_nxt] = 0;
        __syncthreads();
        const int changed_last = s_any[f_prv];
        const BoundaryState<R> nb = s_bs[pb][wv];
        if (rounds > 0 && changed_last == 0) break;   // nobody had work left last round
        if (lane == 0 && tid > 0 && fwd_active) {
            need = !(same_bits(nb.u, in_u) && same_bits(nb.w, in_w));
            in_u = nb.u;
            in_w = nb.w;
        }
        if (need) s_any[f_cur] = 1;
        { const int t = f_prv; f_prv = f_cur; f_cur = f_nxt; f_nxt = t; }
        rounds++;
        if (rounds > 2 * T + 8) {
            if (tid == 0 && flags) atomicOr(&flags[b], VAP_FLAG_NOCONVERGE_BIT);
            break;
        }
    }
    if constexpr (CM) {
        // forward commit: the converged incoming states are final; write the sweep's result over the caps
        R uu = in_u, wp = in_w;
#pragma unroll
        for (int s = 0; s < L; s++) {
            if (s == 0 && tid == 0) { cp[0] = start_u; continue; }
            uu = step_fwd(am[PSA ? s : 0], q[s], g[s], A[s], cp[s], uu, wp);
            cp[s] = uu;
        }
    }
    const int fwd_rounds = rounds;
    const long long t_fwd1 = stats ? __builtin_amdgcn_s_memtime() : 0;

    // ---------------- backward sweep: the step (j+1 -> j) into owned sample j uses k[j+1], dth[j]
    // (and k[j+2] for rho).  Slots at or past the fixed end sample N-1 are idle slots holding
    // u = end_u: walking through them restarts the chain exactly as MPG:252-253 does.
    // (The stage still holds dtheta.)
    RowRegs<R, L> rk;   // the curvature row comes back while dtheta is read out of the stage
    row_fetch<R, L>(rk, K, S < TL + 2 ? S : TL + 2, aligned, tid, T);
#pragma unroll
    for (int s = 0; s < L; s++) g[s] = stage[cpos(s)];   // dtheta[j] for now
    __syncthreads();
    stage_put<R, L>(stage, rk, TL + 2, aligned, tid, T);
    __syncthreads();
    {
        R kc = (R)fabs(stage[cpos(1)]);
#pragma unroll
        for (int s0 = 0; s0 < L; s0 += BK) {
            R kn[BK];
#pragma unroll
            for (int i = 0; i < BK; i++) kn[i] = (R)fabs(stage[cpos(s0 + i + 2)]);
#pragma unroll
            for (int i = 0; i < BK; i++) {
                const int s = s0 + i, j = lo + s;
                const bool valid = j <= N - 2;
                R qq;
                R base = fc.adecp;
                if constexpr (ACC) {
                    // the step into j starts at sample j+1: the clamp comes from the sweep's max_dec, the wheel limit
                    // from the max_acc the sweep has at j+1 (MPG:256-257); a straight sample has max_dec alone
                    base = adecp_b;
                }
                [[maybe_unused]] const R ufwd_s = cp[s];   // CM: the forward result sits where the new cap goes
                fast_derive_k(fc, kc, (j + 2 <= N - 1) ? kn[i] : (R)0, base, q[s], qq, A[s], cp[s]);
                g[s] = fast_gq(fast_gg(fc, g[s]), qq);
                if constexpr (ACC) {
                    A[s] = fast_cap_A(fc, kc, A[s]);
                    // (a zero heading difference, g < 0: the wheel limit is +inf unless the angular velocity rises)
                    // — there the clamp decides: amaxp = A (kHuge times any non-zero rise still wins against it)
                    am[s] = (valid && !(kc < (R)1e-6) && !(g[s] < (R)0)) ? twodd * (R)acc.bwd[row + j + 1] : A[s];
                }
                if constexpr (CM) {
                    if (!valid) { idle_coef(q[s], g[s], A[s], cp[s]); cp[s] = end_u; }
                    else cp[s] = vmin(cp[s], ufwd_s);
                } else {
                    if (!valid) { idle_coef(q[s], g[s], A[s], cp[s]); u[s] = end_u; }
                }
                {
                    R amv, gv;
                    fast_scale(ACC ? am[ACC ? s : 0] : fc.amaxp, g[s], A[s], amv, gv);
                    g[s] = gv;
                    if constexpr (PSA) am[s] = amv;
                }
                q[s] = opaque(q[s]);
                kc = kn[i];
            }
        }
    }
    const int last_chunk = (N - 1) / L;     // chunk that owns the fixed end sample
    if (tid >= last_chunk) { in_u = end_u; in_w = (R)0; }
    else { in_u = CM ? cp[L - 1] : u[CM ? 0 : L - 1]; in_w = in_u; }
    // The backward step reads the forward value of the sample it overwrites, so a chunk cannot be
    // re-run in place: the rounds only propagate boundary states (u[] stays the forward result) and
    // one commit evaluation with the final incoming state stores the backward velocities.
    const bool bwd_active = tid <= last_chunk;
    need = bwd_active;
    out_u = in_u;
    out_w = in_w;
    rounds = 0;
    if (tid == 0) { s_any[0] = 0; s_any[1] = 0; s_any[2] = 0; }
    __syncthreads();
    f_cur = 0; f_nxt = 1; f_prv = 2;
    const bool bwd_nb = lane < 63 && tid < last_chunk;
    while (true) {   // mirror image: states move one lane down, wave w+1 hands its first chunk's state to wave w
        if constexpr (CM) {
            if (__ballot(need) != 0) {   // (as in the forward sweep: the wavefront evaluates as one)
#pragma unroll 1
                for (int k = 0; k < kInner; k++) {
                    R uu = in_u, wp = in_w;
                    if (any_dup) {
#pragma unroll
                        for (int s = L - 1; s >= 0; s--) uu = bwd_step<true, true>(am[PSA ? s : 0], q[s], g[s], A[s], cp[s], uu, wp, (R)0);
                    } else {
#pragma unroll
                        for (int s = L - 1; s >= 0; s--) uu = bwd_step<false, true>(am[PSA ? s : 0], q[s], g[s], A[s], cp[s], uu, wp, (R)0);
                    }
                    out_u = uu;
                    out_w = wp;
                    const R nu = wave_shift_down(out_u), nw = wave_shift_down(out_w);
                    need = bwd_nb && !(same_bits(nu, in_u) && same_bits(nw, in_w));
                    in_u = bwd_nb ? nu : in_u;
                    in_w = bwd_nb ? nw : in_w;
                    if (__ballot(need) == 0) break;
                }
            }
        } else
#pragma unroll 1
        for (int k = 0; k < kInner; k++) {
            if (need) {
                R uu = in_u, wp = in_w;
                if (any_dup) {
#pragma unroll
                    for (int s = L - 1; s >= 0; s--) uu = bwd_step<true, CM>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, u[CM ? 0 : s]);
                } else {
#pragma unroll
                    for (int s = L - 1; s >= 0; s--) uu = bwd_step<false, CM>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, u[CM ? 0 : s]);
                }
                out_u = uu;
                out_w = wp;
            }
            const R nu = wave_shift_down(out_u), nw = wave_shift_down(out_w);
            need = false;
            if (lane < 63 && tid < last_chunk) {
                need = !(same_bits(nu, in_u) && same_bits(nw, in_w));
                in_u = nu;
                in_w = nw;
            }
            if (__ballot(need) == 0) break;
        }
        const int pb = rounds & 1;
        if (lane == 0) s_bs[pb][wv] = BoundaryState<R>{out_u, out_w};
        if (tid == 0) s_any[f_nxt] = 0;
        __syncthreads();
        const int changed_last = s_any[f_prv];
        const BoundaryState<R> nb = s_bs[pb][wv + 1];
        if (rounds > 0 && changed_last == 0) break;
        if (lane == 63 && tid < last_chunk) {
            need = !(same_bits(nb.u, in_u) && same_bits(nb.w, in_w));
            in_u = nb.u;
            in_w = nb.w;
        }
        if (need) s_any[f_cur] = 1;
        { const int t = f_prv; f_prv = f_cur; f_cur = f_nxt; f_nxt = t; }
        rounds++;
        if (rounds > 2 * T + 8) {
            if (tid == 0 && flags) atomicOr(&flags[b], VAP_FLAG_NOCONVERGE_BIT);
            break;
        }
    }
    const long long t_bwd1 = stats ? __builtin_amdgcn_s_memtime() : 0;
    if (bwd_active) {
        R uu = in_u, wp = in_w;
        if (any_dup) {
#pragma unroll
            for (int s = L - 1; s >= 0; s--) (CM ? cp[s] : u[CM ? 0 : s]) = uu = bwd_step<true, CM>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, u[CM ? 0 : s]);
        } else {
#pragma unroll
            for (int s = L - 1; s >= 0; s--) (CM ? cp[s] : u[CM ? 0 : s]) = uu = bwd_step<false, CM>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, u[CM ? 0 : s]);
        }
    }
    // velocities leave through the stage (as IO elements, same padded positions) so the row is written with
    // 16 bytes per lane
    __syncthreads();
    IO *ostage = reinterpret_cast<IO *>(smem_raw);
#pragma unroll
    for (int s = 0; s < L; s++) {
        const int j = lo + s;
        ostage[cpos(s)] = j < N ? (IO)vel_sqrt(CM ? cp[s] : u[CM ? 0 : s]) : (IO)0;
    }
    __syncthreads();
    IO *V = vel + row;
    {
        constexpr int VW = 16 / sizeof(IO);
        const int n = S < TL ? S : TL;
        if ((S % VW) == 0) {
            using VT = typename std::conditional<sizeof(IO) == 4, float4, double2>::type;
            VT *dst = reinterpret_cast<VT *>(V);
            for (int i = tid; i < n / VW; i += T) {
                VT v;
                IO *e = reinterpret_cast<IO *>(&v);
                const int p0 = stage_pos<IO, L>(i * VW);
#pragma unroll
                for (int k = 0; k < VW; k++)   // L % VW == 0: the VW elements share a chunk
                    e[k] = ostage[(L % VW == 0) ? p0 + k : stage_pos<IO, L>(i * VW + k)];
                dst[i] = v;
            }
            for (int i = (n / VW) * VW + tid; i < n; i += T) V[i] = ostage[stage_pos<IO, L>(i)];
        } else {
            for (int i = tid; i < n; i += T) V[i] = ostage[stage_pos<IO, L>(i)];
        }
    }
    for (int j = TL + tid; j < S; j += T) V[j] = (IO)0;
    if constexpr (!std::is_same<R, IO>::value) {
        // the velocities in the arithmetic type as well (vhi: [B][S], what the time-domain resample integrates behind
        // fp32 rows); straight from the registers — this kernel serves the small batches
        if (vhi) {
#pragma unroll
            for (int s = 0; s < L; s++) {
                const int j = lo + s;
                if (j < S) vhi[row + j] = j < N ? vel_sqrt(CM ? cp[s] : u[CM ? 0 : s]) : (R)0;
            }
            for (int j = TL + tid; j < S; j += T) vhi[row + j] = (R)0;
        }
    }
    if (stats && tid == 0) {
        long long *st = stats + (size_t)b * 8;
        st[0] = fwd_rounds;
        st[1] = rounds;
        st[2] = t_fwd0 - t_start;                       // load + derive
        st[3] = t_fwd1 - t_fwd0;                        // forward rounds
        st[4] = t_bwd1 - t_fwd1;                        // backward derive + rounds
        st[5] = __builtin_amdgcn_s_memtime() - t_bwd1;  // commit + store
    }
}

// ------------------------------------------------------------------------------------------------
// K5c: rows longer than the register-resident kernel covers (config 2: one path, 10^6 samples).
// Same relaxation, two levels: a row is cut into super-chunks of SC = MAXT*L samples, one workgroup
// each, which relax internally exactly like K5b for a given incoming interface state; the interface
// states between super-chunks live in HBM and are iterated by re-launching ("super-rounds") until no
// interface changes.  A super-chunk whose incoming state is bit-identical to the one it last used
// returns at once.  Forward velocities go through an HBM scratch row (the backward sweep of a
// super-chunk may be re-run), final velocities are written by the backward kernel.
//   bnd   [2][B][nsc+1][2]  interface states by super-round parity
//   used  [B][nsc][2]       incoming state of the last evaluation      (NaN pattern = never)
//   outst [B][nsc][2]       outgoing state of the last evaluation
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ void k_dup_scan(int B, int S, VelConsts<R> c, const double *__restrict__ meta, const R *__restrict__ curv,
                           const R *__restrict__ dth, int *__restrict__ dup)
{
    const int b = blockIdx.y;
    const int N = (int)meta[(size_t)b * kMetaStride + 3];
    const FastConsts<R> fc = make_fast(c, (R)2 * (R)meta[(size_t)b * kMetaStride + 2]);
    bool any = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N - 1; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)b * S + i;
        const R kabs = (R)fabs(curv[o]);
        any |= fast_gq(fast_gg(fc, dth[o]), kabs * kabs) < (R)0;   // the kernels' own predicate
    }
    if (__syncthreads_or(any ? 1 : 0) && threadIdx.x == 0) atomicOr(&dup[b], 1);
}

template <typename R, typename IO, int L, int MAXT, int MINW, bool BWD>
__global__ __launch_bounds__(MAXT, MINW) void k_velocity_long(int S, int nsc, int round, int seq_sc, VelConsts<R> c, R start_u,
                                                              R end_u, const double *__restrict__ meta,
                                                              const R *__restrict__ curv,
                                                              const R *__restrict__ dtheta, R *__restrict__ ufwd,
                                                              IO *__restrict__ vel, R *__restrict__ bnd,
                                                              R *__restrict__ used, R *__restrict__ outst,
                                                              const int *__restrict__ dupflag,
                                                              int *__restrict__ changed, uint32_t *__restrict__ flags,
                                                              R *__restrict__ vhi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *stage = reinterpret_cast<R *>(smem_raw);
    __shared__ BoundaryState<R> s_bs[2][MAXT / 64 + 2];   // states crossing a wavefront boundary
    __shared__ int s_any[3];
    constexpr int T = MAXT;
    constexpr int SC = T * L;
    // seq_sc >= 0: "sequential windows" mode — this launch handles super-chunk seq_sc of every path and
    // its incoming interface state (published by the previous launch) is final, so it is evaluated once.
    const bool seq = seq_sc >= 0;
    const int sc = seq ? seq_sc : blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int B = gridDim.y;
    const double *m = meta + (size_t)b * kMetaStride;
    const R twodd = (R)2 * (R)m[2];
    const int N = (int)m[3];
    const int base = sc * SC;
    const int last_sc = (N - 1) / SC;
    if (sc > last_sc) return;
    const size_t row = (size_t)b * S;
    const R *K = curv + row, *DT = dtheta + row;
    const FastConsts<R> fc = make_fast(c, twodd);
    const bool any_dup = dupflag[b] != 0;
    const int lo = tid * L;
    // interface arrays
    const size_t ifs = (size_t)(nsc + 1) * 2;                          // per path
    R *bnd_prev = bnd + ((size_t)(seq ? 0 : ((round + 1) & 1)) * B + b) * ifs;
    R *bnd_cur = bnd + ((size_t)(seq ? 0 : (round & 1)) * B + b) * ifs;
    R *my_used = used + ((size_t)b * nsc + sc) * 2;
    R *my_out = outst + ((size_t)b * nsc + sc) * 2;
    const int if_in = BWD ? sc + 1 : sc, if_out = BWD ? sc : sc + 1;
    const bool exact_in = BWD ? (sc == last_sc) : (sc == 0);
    R in0_u = exact_in ? (BWD ? end_u : start_u) : (R)0, in0_w = (R)0;
    if (seq && !exact_in) { in0_u = bnd_prev[if_in * 2]; in0_w = bnd_prev[if_in * 2 + 1]; }
    if (!seq && round > 0) {
        if (!exact_in) { in0_u = bnd_prev[if_in * 2]; in0_w = bnd_prev[if_in * 2 + 1]; }
        if (exact_in || (same_bits(in0_u, my_used[0]) && same_bits(in0_w, my_used[1]))) {
            // nothing new came in: republish the last outgoing state for the next super-round
            if (tid == 0) { bnd_cur[if_out * 2] = my_out[0]; bnd_cur[if_out * 2 + 1] = my_out[1]; }
            return;
        }
    }
    if (tid == 0) { s_any[0] = 0; s_any[1] = 0; s_any[2] = 0; }
    R q[L], g[L], A[L], cp[L], u[L];   // q[] = rho, g[] = g*k^2 (vap_device.h)
    constexpr bool PSA = ScaledStep<R>::value;   // the scaled fp64 step keeps its `am` per slot (fast_scale)
    R am[PSA ? L : 1];
    const R base_p = BWD ? fc.adecp : fc.amaxp;
    // element e of the stage = global sample g0 + e; a forward super-chunk needs k[base-2 ..]
    constexpr int VW = 16 / (int)sizeof(R);
    const int g0 = BWD ? base : (base > 0 ? base - VW : 0);
    const int n_k = BWD ? SC + 2 : SC + VW;
    const bool aligned = (S % VW) == 0;
    {
        int n = N - g0;
        n = n < 0 ? 0 : (n > n_k ? n_k : n);
        stage_load<R, L>(stage, K + g0, n, n_k, aligned, tid, T);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < L; s++) {
        const int j = base + lo + s;                                  // global owned sample
        const bool valid = BWD ? (j <= N - 2) : (j >= 1 && j <= N - 1);
        const int src = BWD ? j + 1 : j - 1;                          // curvature sample of the step
        const int prv = BWD ? j + 2 : j - 2;                          // ... and of the step before it
        const bool has_prv = valid && prv >= 0 && prv <= N - 1;
        const R kabs = (R)fabs(stage[stage_pos<R, L>(valid ? src - g0 : 0)]);
        const R kpv = (R)fabs(stage[stage_pos<R, L>(has_prv ? prv - g0 : 0)]);
        fast_derive_k(fc, kabs, has_prv ? kpv : (R)0, base_p, q[s], g[s], A[s], cp[s]);   // g[s] = k^2 for now
        if (!valid) idle_coef(q[s], g[s], A[s], cp[s]);
        u[s] = BWD ? end_u : start_u;
    }
    __syncthreads();
    {
        int n = N - g0;
        n = n < 0 ? 0 : (n > n_k ? n_k : n);
        stage_load<R, L>(stage, DT + g0, n, n_k, aligned, tid, T);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < L; s++) {
        const int j = base + lo + s;
        const bool valid = BWD ? (j <= N - 2) : (j >= 1 && j <= N - 1);
        const int src = BWD ? j : j - 1;                              // dtheta sample of the step
        const R dth = stage[stage_pos<R, L>(valid ? src - g0 : 0)];
        const R gq = fast_gq(fast_gg(fc, dth), g[s]);
        R amv, gv;
        fast_scale(fc.amaxp, valid ? gq : (R)0, A[s], amv, gv);
        g[s] = gv;
        if constexpr (PSA) am[s] = amv;
    }
    if constexpr (BWD) {
        // forward velocities of the owned samples (idle slots hold end_u)
        __syncthreads();
        int n = N - base;
        n = n < 0 ? 0 : (n > SC ? SC : n);
        stage_load<R, L>(stage, ufwd + row + base, n, SC, (S % (16 / (int)sizeof(R))) == 0, tid, T);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < L; s++) {
            const int j = base + lo + s;
            if (j <= N - 2) u[s] = stage[stage_pos<R, L>(lo + s)];
        }
    }
    // chunk-local incoming states
    R in_u, in_w;
    const int local_last = BWD ? ((N - 1 - base) / L) : 0;   // BWD: chunk holding the fixed end sample (if in range)
    const bool bwd_has_end = BWD && sc == last_sc;
    if constexpr (!BWD) {
        if (tid == 0) {
            if (!exact_in && round == 0 && !seq) { in0_u = cp[0]; in0_w = in0_u; }
            in_u = in0_u; in_w = in0_w;
        } else { in_u = cp[0]; in_w = in_u; }
    } else {
        const bool tail = bwd_has_end ? tid >= local_last : false;
        if (tail) { in_u = end_u; in_w = (R)0; }
        else if (tid == T - 1) {
            if (round == 0 && !seq) { in0_u = u[L - 1]; in0_w = in0_u; }
            in_u = in0_u; in_w = in0_w;
        } else { in_u = u[L - 1]; in_w = in_u; }
    }
    const bool active = BWD ? (!bwd_has_end || tid <= local_last) : (base + lo <= N - 1);
    bool need = active;
    R out_u = in_u, out_w = in_w;
    int rounds = 0;
    __syncthreads();
    // same round structure as k_velocity_relax: up to kInner evaluations per workgroup barrier, states
    // handed lane to lane by DPP inside a wavefront, through LDS across wavefronts
    constexpr int kInner = 8;
    const int wv = tid >> 6, lane = tid & 63;
    const bool in_wave_nb = BWD ? (lane < 63 && tid < T - 1 && (!bwd_has_end || tid < local_last)) : (lane > 0 && active);
    const bool edge_nb = BWD ? (lane == 63 && tid < T - 1 && (!bwd_has_end || tid < local_last)) : (lane == 0 && tid > 0 && active);
    int f_cur = 0, f_nxt = 1, f_prv = 2;
    while (true) {
#pragma unroll 1
        for (int k = 0; k < kInner; k++) {
            if (need) {
                R uu = in_u, wp = in_w;
                if constexpr (!BWD) {
#pragma unroll
                    for (int s = 0; s < L; s++) {
                        if (s == 0 && tid == 0 && base == 0) continue;   // sample 0 is the given start velocity
                        uu = step_fwd(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp);
                        u[s] = uu;
                    }
                } else if (any_dup) {
#pragma unroll
                    for (int s = L - 1; s >= 0; s--) uu = fast_backward_a<true>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, u[s]);
                } else {
#pragma unroll
                    for (int s = L - 1; s >= 0; s--) uu = fast_backward_a<false>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, u[s]);
                }
                out_u = uu;
                out_w = wp;
            }
            const R nu = BWD ? wave_shift_down(out_u) : wave_shift_up(out_u);
            const R nw = BWD ? wave_shift_down(out_w) : wave_shift_up(out_w);
            need = false;
            if (in_wave_nb) {
                need = !(same_bits(nu, in_u) && same_bits(nw, in_w));
                in_u = nu;
                in_w = nw;
            }
            if (__ballot(need) == 0) break;
        }
        const int pb = rounds & 1;
        if (lane == (BWD ? 0 : 63)) s_bs[pb][wv + (BWD ? 0 : 1)] = BoundaryState<R>{out_u, out_w};
        if (tid == 0) s_any[f_nxt] = 0;
        __syncthreads();
        const int changed_last = s_any[f_prv];
        const BoundaryState<R> nb = s_bs[pb][wv + (BWD ? 1 : 0)];
        if (rounds > 0 && changed_last == 0) break;
        if (edge_nb) {
            need = !(same_bits(nb.u, in_u) && same_bits(nb.w, in_w));
            in_u = nb.u;
            in_w = nb.w;
        }
        if (need) s_any[f_cur] = 1;
        { const int t = f_prv; f_prv = f_cur; f_cur = f_nxt; f_nxt = t; }
        rounds++;
        if (rounds > 2 * T + 8) {
            if (tid == 0 && flags) atomicOr(&flags[b], VAP_FLAG_NOCONVERGE_BIT);
            break;
        }
    }
    if constexpr (BWD) {
        if (active) {
            R uu = in_u, wp = in_w;
            if (any_dup) {
#pragma unroll
                for (int s = L - 1; s >= 0; s--) u[s] = uu = fast_backward_a<true>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, u[s]);
            } else {
#pragma unroll
                for (int s = L - 1; s >= 0; s--) u[s] = uu = fast_backward_a<false>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, u[s]);
            }
        }
    }
    // publish the super-chunk's outgoing interface state: the out-state of its last (forward) / first
    // (backward) chunk, which that thread holds
    const int owner = BWD ? 0 : T - 1;
    if (tid == owner) {
        const R ou = out_u, ow = out_w;
        const bool diff = !seq && (round == 0 || !(same_bits(ou, bnd_prev[if_out * 2]) && same_bits(ow, bnd_prev[if_out * 2 + 1])));
        bnd_cur[if_out * 2] = ou;
        bnd_cur[if_out * 2 + 1] = ow;
        my_out[0] = ou;
        my_out[1] = ow;
        // a guessed incoming state (super-round 0) is recorded as "never": the next super-round
        // re-evaluates with whatever the neighbour published
        const bool guessed = round == 0 && !exact_in && !seq;
        my_used[0] = guessed ? (R)NAN : in0_u;
        my_used[1] = guessed ? (R)NAN : in0_w;
        if (diff) atomicAdd(changed, 1);
    }
    // rows leave through the stage: forward -> u (squared velocity) scratch, backward -> final velocity
    __syncthreads();
    int n = S - base;
    n = n > SC ? SC : n;
    if constexpr (BWD) {
        IO *ostage = reinterpret_cast<IO *>(smem_raw);
#pragma unroll
        for (int s = 0; s < L; s++) {
            const int j = base + lo + s;
            ostage[stage_pos<IO, L>(lo + s)] = j < N ? (IO)vel_sqrt(u[s]) : (IO)0;
        }
        __syncthreads();
        IO *dst = vel + row + base;
        for (int i = tid; i < n; i += T) dst[i] = ostage[stage_pos<IO, L>(i)];
        if (sc == last_sc)
            for (int j = (last_sc + 1) * SC + tid; j < S; j += T) vel[row + j] = (IO)0;
        if constexpr (!std::is_same<R, IO>::value) {
            if (vhi) {   // the velocities in the arithmetic type as well (the time-domain resample behind fp32 rows)
#pragma unroll
                for (int s = 0; s < L; s++) {
                    const int j = base + lo + s;
                    if (j < S) vhi[row + j] = j < N ? vel_sqrt(u[s]) : (R)0;
                }
                if (sc == last_sc)
                    for (int j = (last_sc + 1) * SC + tid; j < S; j += T) vhi[row + j] = (R)0;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < L; s++) stage[stage_pos<R, L>(lo + s)] = u[s];
        __syncthreads();
        R *dst = ufwd + row + base;
        for (int i = tid; i < n; i += T) dst[i] = stage[stage_pos<R, L>(i)];
    }
}

// ------------------------------------------------------------------------------------------------
// K5c': the same two-level relaxation with the interfaces handed on INSIDE a launch — decoupled
// look-back between super-chunks, one launch per direction, no host round trip (config 2).
// A workgroup is one wavefront and owns the super-chunk of 64*L samples its TICKET names (tickets are
// drawn in dispatch order, so a workgroup only ever waits for workgroups that already run: no
// residency assumption, any number of super-chunks).  It relaxes its super-chunk from a guessed
// incoming state and publishes a record {incoming state used, outgoing state}; then it looks back at
// the records of its (up to) 64 predecessors at once, one per lane:
//   * predecessor's outgoing state differs from the state it used -> relax again from that state
//     (only the entry lane starts; the change ripples as far as it has to) and publish again;
//   * it is FINAL as soon as some predecessor j is final and every link between j and itself is
//     consistent (record k used exactly what record k-1 published): a record's outgoing state is a
//     function of the incoming state it names, so such a chain carries the sequential sweep's values
//     whatever the moments the records were read at.  Finality therefore travels 64 super-chunks per
//     hop where nothing changes, and at the speed of the recurrence where a ramp is being walked.
// Super-chunk 0 (backwards: the one holding the end sample) knows its incoming state and is final
// after one evaluation.  Records are 8-byte {tag, half-word} granules written by single agent-scope
// stores; a record is taken only when all its tags agree (tag = version, top bit = final; zeroed
// before every launch).  Every spin is bounded: a workgroup that waits longer than kChaseTimeout
// raises VAP_FLAG_NOCONVERGE, publishes what it has as final and goes on, so the grid always drains.
// The fixed point is the sequential sweep bit for bit, like k_velocity_long's.
// ------------------------------------------------------------------------------------------------
using gu64 = __attribute__((address_space(1))) unsigned long long;
using gu32 = __attribute__((address_space(1))) unsigned int;
constexpr int kChaseRecGranules = 8;                 // 64-byte records
constexpr uint32_t kChaseFinal = 0x80000000u;
constexpr long long kChaseTimeout = 100000000ll;     // wall_clock64 ticks (100 MHz): 1 s

template <typename R>
__device__ __forceinline__ R wave_bcast(R x, int src)
{
    if constexpr (sizeof(R) == 8) {
        const uint64_t v = __builtin_bit_cast(uint64_t, x);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
        return __builtin_bit_cast(R, ((uint64_t)hi << 32) | lo);
    } else {
        return __builtin_bit_cast(R, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src));
    }
}

// the four values of a record, wave-uniform, leave as G = 4 * sizeof(R) / 4 granules from lanes 0..G-1
template <typename R>
__device__ __forceinline__ void chase_publish(gu64 *rec, uint32_t tag, R in_u, R in_w, R out_u, R out_w, int lane)
{
    constexpr int H = sizeof(R) / 4, G = 4 * H;
    const int vi = lane / H, h = lane % H;
    const R sel = vi == 0 ? in_u : (vi == 1 ? in_w : (vi == 2 ? out_u : out_w));
    uint32_t half;
    if constexpr (sizeof(R) == 8) {
        const uint64_t b = __builtin_bit_cast(uint64_t, sel);
        half = h ? (uint32_t)(b >> 32) : (uint32_t)b;
    } else {
        half = __builtin_bit_cast(uint32_t, sel);
    }
    if (lane < G) __hip_atomic_store(rec + lane, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The sign-aware backward step (fast_backward_a<true, false>, vap_device.h) with its compare / select / max taken off the
// dependent chain: the penalty max(t, other)*|g| with other = (g < 0 ? 0 : -t) is the larger of t*|g| and -t*gn,
// gn = (g < 0 ? 0 : |g|), and rounding is monotone, so
//     clamp01(am - max(t, other)*|g|) = clamp01(min(fma(-t, |g|, am), fma(t, gn, am)))        bit for bit
// (g >= 0: the two FMAs are am -+ |t||g|, the smaller one is the old value; g < 0: t > 0 gives am - t|g| <= am = the
// second, t <= 0 gives max(t, 0) = 0 -> am = the second exactly).  Two independent FMAs and a min with the clamp modifier
// instead of xor / cndmask / cndmask / max / fma: the chain is fma, fma, min, fma, min.
__device__ __forceinline__ double bwd_step_dup2(double am, double rho, double g, double gn, double A, double cap, double u,
                                                double &uprev)
{
    double r, c2;
    asm("v_fma_f64 %0, -%3, %4, %2\n\t"
        "v_fma_f64 %1, %0, %6, %7\n\t"
        "v_fma_f64 %0, -%0, |%5|, %7\n\t"
        "v_min_f64 %0, %0, %1 clamp\n\t"
        "v_fma_f64 %0, %8, %0, %2\n\t"
        "v_min_f64 %0, %0, %9"
        : "=&v"(r), "=&v"(c2)
        : "v"(u), "v"(rho), "v"(uprev), "v"(g), "v"(gn), "v"(am), "v"(A), "v"(cap));
    uprev = u;
    return r;
}
__device__ __forceinline__ float bwd_step_dup2(float am, float rho, float g, float gn, float A, float cap, float u, float &uprev)
{
    (void)gn;
    return bwd_step<true, true>(am, rho, g, A, cap, u, uprev, 0.0f);
}

template <typename R, typename IO, int L, bool BWD>
__global__ __launch_bounds__(64, 2) void k_velocity_chase(int B, int S, int nsc, VelConsts<R> c, R start_u, R end_u,
                                                          const double *__restrict__ meta, const R *__restrict__ curv,
                                                          const R *__restrict__ dtheta, R *__restrict__ ufwd,
                                                          IO *__restrict__ vel, unsigned long long *records,
                                                          unsigned int *ticket, int *dupflag, uint32_t *__restrict__ flags,
                                                          R *__restrict__ vhi, long long *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *stage = reinterpret_cast<R *>(smem_raw);
    constexpr int T = 64;
    constexpr int SC = T * L;
    constexpr int H = sizeof(R) / 4, G = 4 * H;
    const int tid = threadIdx.x, lane = tid;
    // dispatch-ordered ticket -> (path, position in the sweep's order)
    unsigned int tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add((gu32 *)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
    const int b = (int)(tk / (unsigned int)nsc), idx = (int)(tk % (unsigned int)nsc);
    if (b >= B) return;
    const double *m = meta + (size_t)b * kMetaStride;
    const R twodd = (R)2 * (R)m[2];
    const int N = (int)m[3];
    const int last_sc = (N - 1) / SC;
    const int sc = BWD ? last_sc - idx : idx;
    if (idx > last_sc) return;                          // nobody looks back at these
    const int base = sc * SC;
    const size_t row = (size_t)b * S;
    const R *K = curv + row, *DT = dtheta + row;
    const FastConsts<R> fc = make_fast(c, twodd);
    // zero heading differences anywhere on the path (the sign-aware backward step): the forward launch finds them,
    // the backward launch reads the flag
    const bool any_dup = BWD ? dupflag[b] != 0 : false;
    const long long t_start = stats ? wall_clock64() : 0;
    const int lo = tid * L;
    gu64 *path_recs = (gu64 *)records + (size_t)b * nsc * kChaseRecGranules;
    gu64 *my_rec = path_recs + (size_t)idx * kChaseRecGranules;
    const bool exact_in = idx == 0;
    R in0_u = exact_in ? (BWD ? end_u : start_u) : (R)0, in0_w = (R)0;

    // q[] = rho, g[] = g*k^2 (vap_device.h).  Backwards the forward value of a sample is folded into its cap once
    // (min is exact, so min(min(x, cap), u_fwd) = min(x, min(cap, u_fwd)): k_velocity_relax's commit mode) and the
    // results are committed into cp[] — one dependent instruction less per step and no u[] to keep in registers
    R q[L], g[L], A[L], cp[L], u[BWD ? 1 : L];
    constexpr bool PSA = ScaledStep<R>::value;
    R am[PSA ? L : 1];
    R gn[(PSA && BWD) ? L : 1];   // backwards: g with the zero-heading-difference slots zeroed (the sign-aware step)
    const R base_p = BWD ? fc.adecp : fc.amaxp;
    constexpr int VW = 16 / (int)sizeof(R);
    const int g0 = BWD ? base : (base > 0 ? base - VW : 0);
    const int n_k = BWD ? SC + 2 : SC + VW;
    const bool aligned = (S % VW) == 0;
    {
        int n = N - g0;
        n = n < 0 ? 0 : (n > n_k ? n_k : n);
        stage_load<R, L>(stage, K + g0, n, n_k, aligned, tid, T);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < L; s++) {
        const int j = base + lo + s;
        const bool valid = BWD ? (j <= N - 2) : (j >= 1 && j <= N - 1);
        const int src = BWD ? j + 1 : j - 1;
        const int prv = BWD ? j + 2 : j - 2;
        const bool has_prv = valid && prv >= 0 && prv <= N - 1;
        const R kabs = (R)fabs(stage[stage_pos<R, L>(valid ? src - g0 : 0)]);
        const R kpv = (R)fabs(stage[stage_pos<R, L>(has_prv ? prv - g0 : 0)]);
        fast_derive_k(fc, kabs, has_prv ? kpv : (R)0, base_p, q[s], g[s], A[s], cp[s]);
        if (!valid) idle_coef(q[s], g[s], A[s], cp[s]);
        if constexpr (!BWD) u[s] = start_u;
    }
    __syncthreads();
    {
        int n = N - g0;
        n = n < 0 ? 0 : (n > n_k ? n_k : n);
        stage_load<R, L>(stage, DT + g0, n, n_k, aligned, tid, T);
    }
    __syncthreads();
    bool saw_dup = false;
#pragma unroll
    for (int s = 0; s < L; s++) {
        const int j = base + lo + s;
        const bool valid = BWD ? (j <= N - 2) : (j >= 1 && j <= N - 1);
        const int src = BWD ? j : j - 1;
        const R dth = stage[stage_pos<R, L>(valid ? src - g0 : 0)];
        const R gq = fast_gq(fast_gg(fc, dth), g[s]);
        if constexpr (!BWD) saw_dup |= valid && gq < (R)0;   // k_dup_scan's predicate, on the same values
        R amv, gv;
        fast_scale(fc.amaxp, valid ? gq : (R)0, A[s], amv, gv);
        g[s] = gv;
        if constexpr (PSA) am[s] = amv;
        if constexpr (PSA && BWD) gn[s] = gv < (R)0 ? (R)0 : gv;
    }
    if constexpr (!BWD) {
        if (__ballot(saw_dup) != 0ull && tid == 0) atomicOr(&dupflag[b], 1);
    }
    if constexpr (BWD) {
        __syncthreads();
        int n = N - base;
        n = n < 0 ? 0 : (n > SC ? SC : n);
        stage_load<R, L>(stage, ufwd + row + base, n, SC, aligned, tid, T);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < L; s++) {
            const int j = base + lo + s;
            const R uf = j <= N - 2 ? stage[stage_pos<R, L>(lo + s)] : end_u;   // (a slot that holds no step restarts the chain at end_u)
            cp[s] = vmin(cp[s], uf);
            if (s == L - 1) u[0] = uf;
        }
    }
    // chunk-local incoming states (k_velocity_long's, for one wavefront)
    R in_u, in_w;
    const int local_last = BWD ? ((N - 1 - base) / L) : 0;
    const bool bwd_has_end = BWD && sc == last_sc;
    constexpr int entry = BWD ? T - 1 : 0;              // the lane a neighbour's state comes in at
    constexpr int owner = BWD ? 0 : T - 1;              // the lane whose outgoing state leaves the super-chunk
    if constexpr (!BWD) {
        if (!exact_in) { in0_u = wave_bcast(cp[0], 0); in0_w = in0_u; }
        if (tid == 0) { in_u = in0_u; in_w = in0_w; }
        else { in_u = cp[0]; in_w = in_u; }
    } else {
        if (!exact_in) { in0_u = wave_bcast(u[0], T - 1); in0_w = in0_u; }
        const bool tail = bwd_has_end ? tid >= local_last : false;
        if (tail) { in_u = end_u; in_w = (R)0; }
        else if (tid == T - 1) { in_u = in0_u; in_w = in0_w; }
        else { in_u = u[0]; in_w = in_u; }
    }
    const bool active = BWD ? (!bwd_has_end || tid <= local_last) : (base + lo <= N - 1);
    const bool in_wave_nb = BWD ? (lane < 63 && (!bwd_has_end || tid < local_last)) : (lane > 0 && active);
    R out_u = in_u, out_w = in_w;
    bool need = active;
    uint32_t version = 0;
    bool timed_out = false;
    const long long t_begin = wall_clock64();
    long long t_first = 0;
    int n_evals = 0, n_iters = 0, n_polls = 0;
    while (true) {
        // relax the super-chunk for the incoming state in0: states move lane to lane by DPP
        int it = 0;
        while (true) {
            if (need) {
                R uu = in_u, wp = in_w;
                if constexpr (!BWD) {
#pragma unroll
                    for (int s = 0; s < L; s++) {
                        if (s == 0 && tid == 0 && base == 0) continue;   // sample 0 is the given start velocity
                        uu = step_fwd(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp);
                        u[s] = uu;
                    }
                } else if (any_dup) {
#pragma unroll
                    for (int s = L - 1; s >= 0; s--) uu = bwd_step_dup2(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], gn[(PSA && BWD) ? s : 0], A[s], cp[s], uu, wp);
                } else {
#pragma unroll
                    for (int s = L - 1; s >= 0; s--) uu = bwd_step<false, true>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, (R)0);
                }
                out_u = uu;
                out_w = wp;
            }
            const R nu = BWD ? wave_shift_down(out_u) : wave_shift_up(out_u);
            const R nw = BWD ? wave_shift_down(out_w) : wave_shift_up(out_w);
            need = false;
            if (in_wave_nb) {
                need = !(same_bits(nu, in_u) && same_bits(nw, in_w));
                in_u = nu;
                in_w = nw;
            }
            n_iters++;
            if (__ballot(need) == 0) break;
            if (++it > 2 * T + 8) {
                if (tid == 0 && flags) atomicOr(&flags[b], VAP_FLAG_NOCONVERGE_BIT);
                break;
            }
        }
        const R pub_u = wave_bcast(out_u, owner), pub_w = wave_bcast(out_w, owner);
        if (stats && n_evals == 0) t_first = wall_clock64();
        n_evals++;
        if (exact_in) {
            chase_publish<R>(my_rec, kChaseFinal | 1u, in0_u, in0_w, pub_u, pub_w, lane);
            break;
        }
        // look back: lane i reads the record of predecessor idx-1-i
        bool final_now = false, again = false;
        bool published = false;
        while (true) {
            n_polls++;
            const int p = idx - 1 - lane;
            unsigned long long gr[G];
            if (p >= 0) {
                gu64 *pr = path_recs + (size_t)p * kChaseRecGranules;
#pragma unroll
                for (int k = 0; k < G; k++) gr[k] = __hip_atomic_load(pr + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
#pragma unroll
                for (int k = 0; k < G; k++) gr[k] = 0ull;
            }
            const uint32_t tag = (uint32_t)(gr[0] >> 32);
            bool ok = p >= 0 && tag != 0u;
#pragma unroll
            for (int k = 1; k < G; k++) ok = ok && (uint32_t)(gr[k] >> 32) == tag;
            R pv[4];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                if constexpr (sizeof(R) == 8)
                    pv[v] = __builtin_bit_cast(R, ((gr[2 * v + 1] & 0xffffffffull) << 32) | (gr[2 * v] & 0xffffffffull));
                else
                    pv[v] = __builtin_bit_cast(R, (uint32_t)gr[v]);
            }
            const unsigned long long okmask = __ballot(ok);
            const bool ok0 = (okmask & 1ull) != 0;
            const R cand_u = wave_bcast(pv[2], 0), cand_w = wave_bcast(pv[3], 0);
            if (ok0 && !(same_bits(cand_u, in0_u) && same_bits(cand_w, in0_w))) {
                in0_u = cand_u;
                in0_w = cand_w;
                again = true;
            }
            // links: record k (lane i) used what record k-1 (lane i+1) published
            const R nxt_u = wave_shift_down(pv[2]), nxt_w = wave_shift_down(pv[3]);
            const bool nxt_ok = lane < 63 && ((okmask >> (lane + 1)) & 1ull) != 0;
            const bool link = ok && nxt_ok && same_bits(pv[0], nxt_u) && same_bits(pv[1], nxt_w);
            const unsigned long long fin = __ballot(ok && (tag & kChaseFinal) != 0u);
            const unsigned long long links = __ballot(link);
            bool chain = false;
            if (ok0 && fin != 0ull) {
                const int j = __builtin_ctzll(fin);
                const unsigned long long mask = (j == 0) ? 0ull : (~0ull >> (64 - j));
                chain = (links & mask) == mask;
            }
            if (again) { final_now = false; break; }   // relax from the new state first; the next look-back decides
            if (chain) { final_now = true; break; }
            if (!published) {
                // what this evaluation produced, for the successors to work with meanwhile
                version++;
                chase_publish<R>(my_rec, version, in0_u, in0_w, pub_u, pub_w, lane);
                published = true;
            }
            if (wall_clock64() - t_begin > kChaseTimeout) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        if (timed_out) {
            if (tid == 0 && flags) atomicOr(&flags[b], VAP_FLAG_NOCONVERGE_BIT);
            version++;
            chase_publish<R>(my_rec, kChaseFinal | version, in0_u, in0_w, pub_u, pub_w, lane);
            break;
        }
        if (final_now) {
            version++;
            chase_publish<R>(my_rec, kChaseFinal | version, in0_u, in0_w, pub_u, pub_w, lane);
            break;
        }
        // again: the entry lane restarts from the new incoming state
        need = false;
        if (lane == entry) {
            need = true;
            in_u = in0_u;
            in_w = in0_w;
        }
    }
    const long long t_final = stats ? wall_clock64() : 0;
    if constexpr (BWD) {
        if (active) {
            R uu = in_u, wp = in_w;
            if (any_dup) {
#pragma unroll
                for (int s = L - 1; s >= 0; s--) cp[s] = uu = bwd_step_dup2(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], gn[(PSA && BWD) ? s : 0], A[s], cp[s], uu, wp);
            } else {
#pragma unroll
                for (int s = L - 1; s >= 0; s--) cp[s] = uu = bwd_step<false, true>(PSA ? am[PSA ? s : 0] : fc.amaxp, q[s], g[s], A[s], cp[s], uu, wp, (R)0);
            }
        }
    }
    // rows leave through the stage: forward -> u (squared velocity) scratch, backward -> final velocity
    __syncthreads();
    int n = S - base;
    n = n > SC ? SC : n;
    if constexpr (BWD) {
        IO *ostage = reinterpret_cast<IO *>(smem_raw);
#pragma unroll
        for (int s = 0; s < L; s++) {
            const int j = base + lo + s;
            ostage[stage_pos<IO, L>(lo + s)] = j < N ? (IO)vel_sqrt(cp[s]) : (IO)0;
        }
        __syncthreads();
        IO *dst = vel + row + base;
        for (int i = tid; i < n; i += T) dst[i] = ostage[stage_pos<IO, L>(i)];
        if (sc == last_sc)
            for (int j = (last_sc + 1) * SC + tid; j < S; j += T) vel[row + j] = (IO)0;
        if constexpr (!std::is_same<R, IO>::value) {
            if (vhi) {   // the velocities in the arithmetic type as well (the time-domain resample behind fp32 rows)
#pragma unroll
                for (int s = 0; s < L; s++) {
                    const int j = base + lo + s;
                    if (j < S) vhi[row + j] = j < N ? vel_sqrt(cp[s]) : (R)0;
                }
                if (sc == last_sc)
                    for (int j = (last_sc + 1) * SC + tid; j < S; j += T) vhi[row + j] = (R)0;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < L; s++) stage[stage_pos<R, L>(lo + s)] = u[s];
        __syncthreads();
        R *dst = ufwd + row + base;
        for (int i = tid; i < n; i += T) dst[i] = stage[stage_pos<R, L>(i)];
    }
    if (stats && tid == 0) {
        long long *my = stats + ((size_t)b * nsc + idx) * 8;
        my[0] = t_start;
        my[1] = t_begin;
        my[2] = t_first;
        my[3] = t_final;
        my[4] = wall_clock64();
        my[5] = n_evals;
        my[6] = n_iters;
        my[7] = n_polls;
    }
}

// ------------------------------------------------------------------------------------------------
// Segment blocks -> monomial coefficients (scratch used by k_sample).
// ------------------------------------------------------------------------------------------------
__global__ void k_power(int n_seg, const double *__restrict__ segments, double *__restrict__ power)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seg) return;
    double r[12];
#pragma unroll
    for (int k = 0; k < 12; k++) r[k] = segments[(size_t)i * 12 + k];
    make_coef_block(r, power + (size_t)i * kCoefDoubles);
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

// QHS:288-469 _get_basis_functions / _derivatives / _second_derivatives / _third_derivatives at n LOCAL parameters
__global__ void k_basis(int order, int n, const double *__restrict__ t, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double H[6];
    hermite_basis_ref(order, t[i], H);
#pragma unroll
    for (int k = 0; k < 6; k++) out[(size_t)i * 6 + k] = H[k];
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
                               const double *tout, const FitExtras &ex, double *seg, double *pw, double *seglen, double *meta,
                               uint32_t *flags)
{
    const size_t lds = sizeof(double) * (size_t)(7 * W);
    const bool plain = !tin && !tout && !seglen && !ex.first && !ex.second && !ex.start_tan && !ex.end_tan && !ex.out_first && !ex.out_second;
    if (plain && W <= 16 && B >= 8192) {
        // very many short paths: 256 / L paths per workgroup, L lanes each (k_fit would leave 56 of its 64 lanes idle at W = 8)
        int L = 4;
        while (L < W) L *= 2;
        const int per = 256 / L;
        hipLaunchKernelGGL(k_fit_many<IT>, dim3((B + per - 1) / per), dim3(256), lds * per, st, B, W, L, (const IT *)wp, seg, pw, meta, flags);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_fit<IT>, dim3(B), dim3(W <= 64 ? 64 : 256), lds, st, W, (const IT *)wp, tin, tout, ex,
                       seg, pw, seglen, meta, flags);
    return hipGetLastError();
}

hipError_t launch_fit(hipStream_t st, bool f64, int B, int W, const void *wp, const double *tin,
                      const double *tout, double *seg, double *pw, double *seglen, double *meta, uint32_t *flags,
                      const double *first, const double *second, const double *start_tan, const double *end_tan,
                      double *out_first, double *out_second)
{
    FitExtras ex;
    ex.first = first;
    ex.second = second;
    ex.start_tan = start_tan;
    ex.end_tan = end_tan;
    ex.out_first = out_first;
    ex.out_second = out_second;
    return f64 ? launch_fit_t<double>(st, B, W, wp, tin, tout, ex, seg, pw, seglen, meta, flags)
               : launch_fit_t<float>(st, B, W, wp, tin, tout, ex, seg, pw, seglen, meta, flags);
}

// K1 + K2 as one launch (k_fit_lut) where the fused call can take it: plain paths, segments that fit the table
// kernel's LDS staging, batches below the grouped table kernel's threshold.
bool fit_lut_fusable(int B, int W) { return W - 1 <= 512 && !(B >= kLutGroupMinPaths && W <= 64) && !(B >= kLutManyMinPaths && W <= 9); }
hipError_t launch_fit_lut(hipStream_t st, bool f64, int B, int W, const void *wp, double *seg, double *pw, double *lut,
                          double *meta, uint32_t *flags, GridArgs grid)
{
    const size_t n = (size_t)(7 * W > kLutPad ? 7 * W : kLutPad);
    // developer knob: VAP_LUT_STATS=1 prints in-kernel cycle shares (synchronises!)
    static const bool want_stats = getenv("VAP_LUT_STATS") != nullptr;
    long long *stats = nullptr;
    if (want_stats) (void)hipMalloc(&stats, (size_t)B * 4 * sizeof(long long));
    if (f64)
        hipLaunchKernelGGL(k_fit_lut<double>, dim3(B), dim3(128), sizeof(double) * n, st, W, (const double *)wp, seg, pw, lut, meta, flags, grid, stats);
    else
        hipLaunchKernelGGL(k_fit_lut<float>, dim3(B), dim3(128), sizeof(double) * n, st, W, (const float *)wp, seg, pw, lut, meta, flags, grid, stats);
    if (stats) {
        std::vector<long long> h((size_t)B * 4);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), stats, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(stats);
        double sum[4] = {0, 0, 0, 0};
        for (int b = 0; b < B; b++)
            for (int k = 0; k < 4; k++) sum[k] += (double)h[(size_t)b * 4 + k];
        fprintf(stderr, "[fit+lut] mean ticks per workgroup: fit %.0f  magnitudes %.0f  increments+sum %.0f  store %.0f\n", sum[3] / B,
                sum[0] / B, sum[1] / B, sum[2] / B);
    }
    return hipGetLastError();
}

hipError_t launch_lut(hipStream_t st, int B, int W, const double *seg, double *lut, double *slopes, double *meta,
                      uint32_t *flags, GridArgs grid, RouteTables rt)
{
    static const bool want_stats = getenv("VAP_LUT_STATS") != nullptr;
    long long *stats = nullptr;
    if (want_stats) (void)hipMalloc(&stats, (size_t)B * 4 * sizeof(long long));
    // (the 64 paths' segment rows are staged in LDS: 6 KB per segment column, so short paths only)
    if (!want_stats && !rt.sptab && W <= 9 && B >= kLutManyMinPaths) {
        // very many paths: 64 per workgroup, the sequential sums of 64 paths in the lanes of one wavefront
        // (512 threads: two of these workgroups share a CU — LDS: 80 KB each — and eight waves each give its SIMDs four)
        hipLaunchKernelGGL((k_lut_many<64, 512>), dim3((B + 63) / 64), dim3(512), sizeof(double) * 64 * (W - 1) * 12, st, B, W, seg, lut,
                           slopes, meta, flags, grid);
        return hipGetLastError();
    }
    if (!want_stats && !rt.sptab && W <= 64 && B >= kLutGroupMinPaths) {
        hipLaunchKernelGGL((k_lut_many<8, 256>), dim3((B + 7) / 8), dim3(256), sizeof(double) * 8 * (W - 1) * 12, st, B, W, seg, lut,
                           slopes, meta, flags, grid);
        return hipGetLastError();
    }
    // 128 threads: the sequential sum keeps one lane busy, so residency (16 workgroups per CU) is what hides it
    const dim3 lgrid(B, rt.sptab ? rt.NS : 1);
    if (W - 1 <= 512) hipLaunchKernelGGL(k_lut<true>, lgrid, dim3(128), sizeof(double) * 12 * (W - 1), st, W, seg, lut, slopes, meta, flags, grid, rt, stats);
    else hipLaunchKernelGGL(k_lut<false>, lgrid, dim3(128), 0, st, W, seg, lut, slopes, meta, flags, grid, rt, stats);
    if (stats) {
        std::vector<long long> h((size_t)B * 4);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), stats, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(stats);
        double sum[3] = {0, 0, 0};
        for (int b = 0; b < B; b++)
            for (int k = 0; k < 3; k++) sum[k] += (double)h[(size_t)b * 4 + k];
        fprintf(stderr, "[lut] mean ticks: magnitudes %.0f  increments+sum %.0f  store+slopes %.0f\n", sum[0] / B, sum[1] / B, sum[2] / B);
    }
    return hipGetLastError();
}

hipError_t launch_lut_slopes(hipStream_t st, int B, const double *lut, const double *meta, double *slopes)
{
    hipLaunchKernelGGL(k_lut_slopes, dim3((B * kLutN + 255) / 256), dim3(256), 0, st, B, lut, meta, slopes);
    return hipGetLastError();
}

hipError_t launch_grid(hipStream_t st, int B, int W, int S, double dd, double *meta, double *aux, double *runs,
                       uint32_t *flags)
{
    hipLaunchKernelGGL(k_grid, dim3((B + 63) / 64), dim3(64), 0, st, B, W, S, dd, meta, aux, runs, flags);
    return hipGetLastError();
}

hipError_t launch_sample(hipStream_t st, bool f64, int B, int W, int S, const double *pw, const double *lut,
                         const double *slopes, const double *meta, const double *aux, const double *runs, void *x,
                         void *y, void *h, void *k, void *dth, double *k64, double *dth64)
{
    const bool hi = !f64 && k64 && dth64;
    // one workgroup stages a path's tables once and walks tiles_per_block consecutive tiles; paths are
    // split over several workgroups only when the batch alone cannot fill the chip
    // a tile is kSampleTile samples (the last thread only feeds its neighbour's |dtheta|) unless the whole
    // row fits one workgroup pass (S <= kSampleChunk, e.g. 1024-sample rows), where no neighbour is needed
    const int tile = S <= kSampleChunk ? kSampleChunk : kSampleTile;
    const int n_tiles = (S + tile - 1) / tile;
    int split = (2048 + B - 1) / B;
    split = split < 1 ? 1 : (split > n_tiles ? n_tiles : split);
    const int tiles_per_block = (n_tiles + split - 1) / split;
    const bool in_lds = (W - 1) <= kLdsCoefSegments;
    // rows of one tile in large batches: two paths per workgroup (their staging loads in flight together)
    const int ppb = (n_tiles == 1 && B >= 8192 && in_lds && (W - 1) <= 16) ? 2 : 1;
    const dim3 grid((n_tiles + tiles_per_block - 1) / tiles_per_block, (B + ppb - 1) / ppb);
    const size_t lds = in_lds ? sizeof(double) * (size_t)ppb * (W - 1) * kCoefDoubles : 0;
    // developer knob: VAP_SAMPLE_STATS=1 prints in-kernel cycle shares per wave (synchronises!)
    static const bool want_stats = getenv("VAP_SAMPLE_STATS") != nullptr;
    long long *stats = nullptr;
    const size_t n_waves = (size_t)grid.x * grid.y * 4;
    if (want_stats) (void)hipMalloc(&stats, n_waves * 4 * sizeof(long long));
#define VAP_SAMPLE(OT_, LDS_, HI_, PPB_)                                                                            \
    hipLaunchKernelGGL((k_sample<OT_, LDS_, HI_, PPB_>), grid, dim3(kSampleThreads), lds, st, B, W, S, tile, tiles_per_block, pw, lut, \
                       slopes, meta, aux, runs, (OT_ *)x, (OT_ *)y, (OT_ *)h, (OT_ *)k, (OT_ *)dth, k64, dth64, stats)
    if (ppb == 2) {
        if (f64) VAP_SAMPLE(double, true, false, 2);
        else if (hi) VAP_SAMPLE(float, true, true, 2);
        else VAP_SAMPLE(float, true, false, 2);
    } else if (f64) { if (in_lds) VAP_SAMPLE(double, true, false, 1); else VAP_SAMPLE(double, false, false, 1); }
    else if (hi) { if (in_lds) VAP_SAMPLE(float, true, true, 1); else VAP_SAMPLE(float, false, true, 1); }
    else { if (in_lds) VAP_SAMPLE(float, true, false, 1); else VAP_SAMPLE(float, false, false, 1); }
#undef VAP_SAMPLE
    if (stats) {
        std::vector<long long> h(n_waves * 4);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), stats, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(stats);
        double sum[4] = {0, 0, 0, 0};
        for (size_t w = 0; w < n_waves; w++)
            for (int k = 0; k < 4; k++) sum[k] += (double)h[w * 4 + k];
        fprintf(stderr, "[sample grid %ux%u, %d tiles/block] per-wave mean ticks: stage %.0f tiles %.0f\n", grid.x,
                grid.y, tiles_per_block, sum[0] / n_waves, sum[1] / n_waves);
    }
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

template <typename R, typename IO>
static void launch_seq_t(hipStream_t st, bool fast, int B, int S, const double c[6], double sv, double ev, const double *meta,
                         const void *curv, const void *dth, const void *vcap, const AccRowsV &accv, void *vel, void *usq)
{
    const dim3 grid((B + 63) / 64);
    AccRows<R> acc;   // (limit rows have the recurrence's arithmetic type)
    acc.fwd = (const R *)accv.fwd; acc.bwd = (const R *)accv.bwd; acc.dec = (const R *)accv.dec;
    const R s = (R)sv, e = (R)ev;
    auto k = fast ? k_velocity_seq<R, IO, true> : k_velocity_seq<R, IO, false>;
    hipLaunchKernelGGL(k, grid, dim3(64), 0, st, B, S, make_consts<R>(c), s * s, e * e, meta, (const R *)curv,
                       (const R *)dth, (const R *)vcap, acc, (IO *)vel, (R *)usq);
}

// r64: arithmetic (and curvature / dtheta rows) in fp64; io64: the caller's rows (vcap, acc, vel) are fp64.
// usq: [B][S] scratch of the arithmetic type, needed when the two differ.
hipError_t launch_velocity_seq(hipStream_t st, bool r64, bool io64, bool fast, int B, int S, const double c[6], double sv,
                               double ev, const double *meta, const void *curv, const void *dth, const void *vcap,
                               const AccRowsV &accv, void *vel, void *usq)
{
    if (r64 && io64) launch_seq_t<double, double>(st, fast, B, S, c, sv, ev, meta, curv, dth, vcap, accv, vel, usq);
    else if (r64) launch_seq_t<double, float>(st, fast, B, S, c, sv, ev, meta, curv, dth, vcap, accv, vel, usq);
    else launch_seq_t<float, float>(st, fast, B, S, c, sv, ev, meta, curv, dth, vcap, accv, vel, usq);
    return hipGetLastError();
}

// Largest sample capacity the register-resident relaxation kernel covers.
// (plain fp64 rows are walked in windows and could be of any length; beyond 64 windows the two-level kernel, which
// spreads a row over many workgroups, is the better tool)
int velocity_relax_max_samples(bool f64, bool limits) { (void)limits; return f64 ? 512 * 20 : 512 * 40; }
// ... and with per-sample max_acceleration rows (one more register array per thread)
int velocity_relax_acc_max_samples(bool f64) { return f64 ? 512 * 8 : 512 * 20; }

template <typename R, typename IO, int L, int MAXT, int MINW, bool ACC = false>
static void launch_relax_t(hipStream_t st, int B, int S, const double c[6], double sv, double ev,
                           const double *meta, const void *curv, const void *dth, const void *vcap, const AccRowsV &accv,
                           void *vel, uint32_t *flags, void *vhi)
{
    AccRows<R> acc;   // (limit rows have the recurrence's arithmetic type)
    acc.fwd = (const R *)accv.fwd;
    acc.bwd = (const R *)accv.bwd;
    acc.dec = (const R *)accv.dec;
    int T = (S + L - 1) / L;
    T = (T + 63) / 64 * 64;
    const R s = (R)sv, e = (R)ev;
    // developer knob: VAP_RELAX_STATS=1 prints rounds and in-kernel cycle shares (synchronises!)
    static const bool want_stats = getenv("VAP_RELAX_STATS") != nullptr;
    long long *stats = nullptr;
    if (want_stats) (void)hipMalloc(&stats, (size_t)B * 8 * sizeof(long long));
    const size_t lds = sizeof(R) * ((size_t)T * L + T + 8);
    if constexpr (ACC)
        hipLaunchKernelGGL((k_velocity_relax<R, IO, L, MAXT, MINW, true, true>), dim3(B), dim3(T), lds, st, S, make_consts<R>(c), s * s,
                           e * e, meta, (const R *)curv, (const R *)dth, (const R *)vcap, acc, (IO *)vel, flags, stats, (R *)vhi);
    else if (vcap)
        hipLaunchKernelGGL((k_velocity_relax<R, IO, L, MAXT, MINW, true, false>), dim3(B), dim3(T), lds, st, S, make_consts<R>(c), s * s,
                           e * e, meta, (const R *)curv, (const R *)dth, (const R *)vcap, acc, (IO *)vel, flags, stats, (R *)vhi);
    else
        hipLaunchKernelGGL((k_velocity_relax<R, IO, L, MAXT, MINW, false, false>), dim3(B), dim3(T), lds, st, S, make_consts<R>(c), s * s,
                           e * e, meta, (const R *)curv, (const R *)dth, (const R *)nullptr, acc, (IO *)vel, flags, stats, (R *)vhi);
    if (stats) {
        std::vector<long long> h((size_t)B * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), stats, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(stats);
        double sum[6] = {0, 0, 0, 0, 0, 0};
        long long mx[6] = {0, 0, 0, 0, 0, 0};
        for (int b = 0; b < B; b++)
            for (int k = 0; k < 6; k++) {
                sum[k] += (double)h[(size_t)b * 8 + k];
                if (h[(size_t)b * 8 + k] > mx[k]) mx[k] = h[(size_t)b * 8 + k];
            }
        fprintf(stderr, "[relax L=%d T=%d] rounds fwd mean %.1f max %lld | bwd mean %.1f max %lld | ticks mean: load %.0f fwd %.0f bwd %.0f commit %.0f | max: %lld %lld %lld %lld\n",
                L, T, sum[0] / B, mx[0], sum[1] / B, mx[1], sum[2] / B, sum[3] / B, sum[4] / B, sum[5] / B, mx[2], mx[3], mx[4], mx[5]);
    }
}

hipError_t launch_velocity_relax(hipStream_t st, bool r64, bool io64, int B, int S, const double c[6], double sv, double ev,
                                 const double *meta, const void *curv, const void *dth, const void *vcap,
                                 const AccRowsV &acc, void *vel, uint32_t *flags, void *vhi)
{
    if (acc.fwd) {
        // per-sample max_acceleration: one more register array per thread, so shorter chunks
        // (velocity_relax_acc_max_samples() is the limit the caller checks)
#define VAP_RELAX_ACC(R_, IO_, L_, MAXT_, W_) launch_relax_t<R_, IO_, L_, MAXT_, W_, true>(st, B, S, c, sv, ev, meta, curv, dth, vcap, acc, vel, flags, vhi)
        if (r64 && io64) {
            if (S <= 64 * 4) VAP_RELAX_ACC(double, double, 4, 256, 4);
            else VAP_RELAX_ACC(double, double, 8, 512, 4);
        } else if (r64) {
            if (S <= 64 * 4) VAP_RELAX_ACC(double, float, 4, 256, 4);
            else VAP_RELAX_ACC(double, float, 8, 512, 4);
        } else {
            if (S <= 64 * 4) VAP_RELAX_ACC(float, float, 4, 1024, 8);
            else if (S <= 512 * 16) VAP_RELAX_ACC(float, float, 16, 512, 4);
            else VAP_RELAX_ACC(float, float, 20, 512, 2);
        }
#undef VAP_RELAX_ACC
        return hipGetLastError();
    }
#define VAP_RELAX(R_, IO_, L_, MAXT_, W_) launch_relax_t<R_, IO_, L_, MAXT_, W_>(st, B, S, c, sv, ev, meta, curv, dth, vcap, acc, vel, flags, vhi)
    // chunk length: the longest instantiated L whose thread count still covers the row — fewer, longer
    // chunks mean fewer rounds (rounds ~ longest unclamped run / L) and fewer waves to synchronise
    const bool one_wave = S > 64 * 4 && S <= 64 * 16;
    if (r64 && io64) {
        if (S <= 64 * 4) VAP_RELAX(double, double, 4, 256, 4);
        else if (one_wave) VAP_RELAX(double, double, 16, 64, 2);   // one wavefront per path: no workgroup barrier at all
        else if (S <= 512 * 8) VAP_RELAX(double, double, 8, 512, 4);
        else VAP_RELAX(double, double, 20, 512, 2);
        return hipGetLastError();
    }
    if (r64) {
        if (S <= 64 * 4) VAP_RELAX(double, float, 4, 256, 4);
        else if (one_wave) VAP_RELAX(double, float, 16, 64, 2);
        else if (S <= 512 * 8) VAP_RELAX(double, float, 8, 512, 4);
        else VAP_RELAX(double, float, 20, 512, 2);
        return hipGetLastError();
    }
    if (S <= 64 * 4) VAP_RELAX(float, float, 4, 1024, 8);
    else if (S <= 256 * 16) VAP_RELAX(float, float, 16, 512, 4);
    else VAP_RELAX(float, float, 40, 512, 2);
#undef VAP_RELAX
    return hipGetLastError();
}

template <typename R, typename IO, int L, int MAXT, int MINW>
static hipError_t velocity_long_t(hipStream_t st, int B, int S, const double c[6], double sv, double ev,
                                  const double *meta, const void *curv, const void *dth, void *vel, uint32_t *flags,
                                  void *ufwd, void *state, int *counters, void *vhi)
{
    constexpr int SC = MAXT * L;
    const int nsc = (S + SC - 1) / SC;
    const R s = (R)sv, e = (R)ev;
    const size_t lds = sizeof(R) * ((size_t)MAXT * L + MAXT + 8);
    // state layout: bnd [2][B][nsc+1][2] | used [B][nsc][2] | outst [B][nsc][2]
    R *bnd = (R *)state;
    R *used = bnd + (size_t)2 * B * (nsc + 1) * 2;
    R *outst = used + (size_t)B * nsc * 2;
    int *dup = counters;            // [B]
    int *changed = counters + B;    // [2 * (nsc + 2)] one counter per super-round and direction
    hipError_t err;
    if ((err = hipMemsetAsync(counters, 0, sizeof(int) * ((size_t)B + 2 * (nsc + 2)), st)) != hipSuccess) return err;
    hipLaunchKernelGGL(k_dup_scan<R>, dim3(64, B), dim3(256), 0, st, B, S, make_consts<R>(c), meta, (const R *)curv,
                       (const R *)dth, dup);
    // Super-round convergence is decided on the host, one 4-byte read per check — but a super-round in
    // which nothing changes costs a few microseconds (every workgroup returns at once), a host round trip
    // ~35: rounds are launched three at a time and only the last one's counter is read.
    // The first check of a sweep comes after as many rounds as the previous call of this shape needed (a caller that
    // profiles batch after batch of similar routes — the bench loop — then pays one round trip per sweep instead of
    // one per three rounds); purely a launch-count heuristic, the result does not depend on it.
    constexpr int kRoundsPerCheck = 3;
    static thread_local int expect_rounds[2] = {kRoundsPerCheck, kRoundsPerCheck};
    static thread_local int expect_nsc = -1;
    if (expect_nsc != nsc) { expect_rounds[0] = expect_rounds[1] = kRoundsPerCheck; expect_nsc = nsc; }
    for (int dir = 0; dir < 2; dir++) {
        int round = 0;
        int batch = expect_rounds[dir];
        while (round <= nsc + 1) {
            int *ch = nullptr;
            for (int k = 0; k < batch && round <= nsc + 1; k++, round++) {
                ch = changed + dir * (nsc + 2) + round;
                if (dir == 0)
                    hipLaunchKernelGGL((k_velocity_long<R, IO, L, MAXT, MINW, false>), dim3(nsc, B), dim3(MAXT), lds, st, S, nsc,
                                       round, -1, make_consts<R>(c), s * s, e * e, meta, (const R *)curv, (const R *)dth,
                                       (R *)ufwd, (IO *)vel, bnd, used, outst, dup, ch, flags, (R *)vhi);
                else
                    hipLaunchKernelGGL((k_velocity_long<R, IO, L, MAXT, MINW, true>), dim3(nsc, B), dim3(MAXT), lds, st, S, nsc,
                                       round, -1, make_consts<R>(c), s * s, e * e, meta, (const R *)curv, (const R *)dth,
                                       (R *)ufwd, (IO *)vel, bnd, used, outst, dup, ch, flags, (R *)vhi);
                if ((err = hipGetLastError()) != hipSuccess) return err;
            }
            int h = 0;
            if ((err = hipMemcpyAsync(&h, ch, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return err;
            if ((err = hipStreamSynchronize(st)) != hipSuccess) return err;
            if (h == 0) break;   // round >= 1 here: the last launched round saw no interface change
            batch = kRoundsPerCheck;
        }
        expect_rounds[dir] = round < kRoundsPerCheck ? kRoundsPerCheck : (round > 48 ? 48 : round);
    }
    return hipSuccess;
}

size_t velocity_long_state_bytes(bool f64, int B, int S)
{
    const int SC = f64 ? 64 * 16 : 64 * 40;   // the smallest super-chunk launch_velocity_long may pick
    const int nsc = (S + SC - 1) / SC;
    return (f64 ? 8 : 4) * ((size_t)2 * B * (nsc + 1) * 2 + (size_t)2 * B * nsc * 2) + 64;
}
size_t velocity_long_counter_bytes(bool f64, int B, int S)
{
    const int SC = f64 ? 64 * 16 : 64 * 40;   // as above: the smallest super-chunk
    const int nsc = (S + SC - 1) / SC;
    return sizeof(int) * ((size_t)B + 2 * (nsc + 2)) + 64;
}

hipError_t launch_velocity_long(hipStream_t st, bool f64, bool io64, int B, int S, const double c[6], double sv, double ev,
                                const double *meta, const void *curv, const void *dth, void *vel, uint32_t *flags,
                                void *ufwd, void *state, int *counters, void *vhi)
{
    // super-chunk = 512, 128 or 64 threads x 16 samples (fp64): few long rows (config 2: one) are cut finer so that the
    // chip has more workgroups to run and a super-round is shorter
    if (f64) {
        const long blocks512 = (long)B * ((S + 512 * 16 - 1) / (512 * 16));
        const int t64 = blocks512 < 256 ? 64 : (blocks512 < 1024 ? 128 : 512);
#define VAP_LONG64(IO_, T_) velocity_long_t<double, IO_, 16, T_, 2>(st, B, S, c, sv, ev, meta, curv, dth, vel, flags, ufwd, state, counters, vhi)
        if (io64) return t64 == 64 ? VAP_LONG64(double, 64) : (t64 == 128 ? VAP_LONG64(double, 128) : VAP_LONG64(double, 512));
        return t64 == 64 ? VAP_LONG64(float, 64) : (t64 == 128 ? VAP_LONG64(float, 128) : VAP_LONG64(float, 512));
#undef VAP_LONG64
    }
    // super-chunk = 256 or 64 threads x 40 samples: few long rows (config 2: one) are cut finer so that the
    // chip has more workgroups to run and a super-round is shorter
    const long blocks256 = (long)B * ((S + 256 * 40 - 1) / (256 * 40));
    if (blocks256 < 512)
        return velocity_long_t<float, float, 40, 64, 2>(st, B, S, c, sv, ev, meta, curv, dth, vel, flags, ufwd, state, counters, nullptr);
    return velocity_long_t<float, float, 40, 256, 2>(st, B, S, c, sv, ev, meta, curv, dth, vel, flags, ufwd, state, counters, nullptr);
}

// K5c': one launch per direction, interfaces handed on by look-back (k_velocity_chase).
//   state    [2][B][nsc] records of 64 bytes (zeroed per call: tag 0 = nothing published)
//   counters [B] dup flags | [2] tickets
template <typename R, typename IO, int L>
static hipError_t velocity_chase_t(hipStream_t st, int B, int S, const double c[6], double sv, double ev, const double *meta,
                                   const void *curv, const void *dth, void *vel, uint32_t *flags, void *ufwd, void *state,
                                   int *counters, void *vhi)
{
    constexpr int SC = 64 * L;
    const int nsc = (S + SC - 1) / SC;
    const R s = (R)sv, e = (R)ev;
    const size_t lds = sizeof(R) * ((size_t)64 * L + 64 + 8);
    const size_t rec_bytes = (size_t)B * nsc * kChaseRecGranules * 8;
    unsigned long long *rec = (unsigned long long *)state;
    int *dup = counters;
    unsigned int *ticket = (unsigned int *)(counters + B);
    hipError_t err;
    if ((err = hipMemsetAsync(state, 0, 2 * rec_bytes, st)) != hipSuccess) return err;
    if ((err = hipMemsetAsync(counters, 0, sizeof(int) * ((size_t)B + 2), st)) != hipSuccess) return err;
    const unsigned int grid = (unsigned int)((size_t)B * nsc);
    // developer knob: VAP_CHASE_STATS=1 prints the timeline of the super-chunks (synchronises!)
    static const bool want_stats = getenv("VAP_CHASE_STATS") != nullptr;
    long long *stats = nullptr;
    if (want_stats) {
        (void)hipMalloc(&stats, (size_t)2 * grid * 8 * sizeof(long long));
        (void)hipMemsetAsync(stats, 0, (size_t)2 * grid * 8 * sizeof(long long), st);
    }
    hipLaunchKernelGGL((k_velocity_chase<R, IO, L, false>), dim3(grid), dim3(64), lds, st, B, S, nsc, make_consts<R>(c), s * s,
                       e * e, meta, (const R *)curv, (const R *)dth, (R *)ufwd, (IO *)vel, rec, ticket, dup, flags, (R *)vhi,
                       stats);
    if ((err = hipGetLastError()) != hipSuccess) return err;
    hipLaunchKernelGGL((k_velocity_chase<R, IO, L, true>), dim3(grid), dim3(64), lds, st, B, S, nsc, make_consts<R>(c), s * s,
                       e * e, meta, (const R *)curv, (const R *)dth, (R *)ufwd, (IO *)vel, rec + rec_bytes / 8, ticket + 1, dup,
                       flags, (R *)vhi, stats ? stats + (size_t)grid * 8 : nullptr);
    if (stats) {
        std::vector<long long> h((size_t)2 * grid * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), stats, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(stats);
        for (int dir = 0; dir < 2; dir++) {
            const long long *d = h.data() + (size_t)dir * grid * 8;
            long long t0 = 0, t_end = 0;
            for (unsigned int i = 0; i < grid; i++)
                if (d[i * 8]) {
                    if (!t0 || d[i * 8] < t0) t0 = d[i * 8];
                    if (d[i * 8 + 4] > t_end) t_end = d[i * 8 + 4];
                }
            double ev = 0, it = 0, po = 0;
            long long mev = 0, n = 0;
            for (unsigned int i = 0; i < grid; i++)
                if (d[i * 8]) {
                    n++;
                    ev += (double)d[i * 8 + 5];
                    it += (double)d[i * 8 + 6];
                    po += (double)d[i * 8 + 7];
                    if (d[i * 8 + 5] > mev) mev = d[i * 8 + 5];
                }
            fprintf(stderr, "[chase %s L=%d] %lld super-chunks, span %.1f us | per super-chunk: evaluations mean %.2f max %lld, inner rounds %.1f, look-backs %.1f\n",
                    dir ? "bwd" : "fwd", L, n, (double)(t_end - t0) * 0.01, ev / (double)n, mev, it / (double)n, po / (double)n);
            // the first path's timeline (us from the first start): where finality advanced by more than 3 us from one
            // super-chunk to the next, and every 1/8 of the row
            const int step = nsc >= 8 ? nsc / 8 : 1;
            long long prev_final = 0;
            for (int i = 0; i < nsc; i++) {
                const long long *r = d + (size_t)i * 8;
                if (!r[0]) continue;
                if (i % step == 0 || r[3] - prev_final > 300)
                    fprintf(stderr, "   idx %5d: start %7.1f derived %7.1f relaxed %7.1f final %7.1f (+%5.1f) end %7.1f | evals %lld rounds %lld polls %lld\n", i,
                            (double)(r[0] - t0) * 0.01, (double)(r[1] - t0) * 0.01, (double)(r[2] - t0) * 0.01, (double)(r[3] - t0) * 0.01,
                            (double)(r[3] - prev_final) * 0.01, (double)(r[4] - t0) * 0.01, r[5], r[6], r[7]);
                prev_final = r[3];
            }
        }
    }
    return hipGetLastError();
}

size_t velocity_chase_state_bytes(bool f64, int B, int S)
{
    const int SC = f64 ? 64 * 16 : 64 * 40;
    const int nsc = (S + SC - 1) / SC;
    return 2 * (size_t)B * nsc * kChaseRecGranules * 8 + 64;
}
size_t velocity_chase_counter_bytes(int B) { return sizeof(int) * ((size_t)B + 2) + 64; }

hipError_t launch_velocity_chase(hipStream_t st, bool f64, bool io64, int B, int S, const double c[6], double sv, double ev,
                                 const double *meta, const void *curv, const void *dth, void *vel, uint32_t *flags,
                                 void *ufwd, void *state, int *counters, void *vhi)
{
    if (f64) {
        if (io64) return velocity_chase_t<double, double, 16>(st, B, S, c, sv, ev, meta, curv, dth, vel, flags, ufwd, state, counters, vhi);
        return velocity_chase_t<double, float, 16>(st, B, S, c, sv, ev, meta, curv, dth, vel, flags, ufwd, state, counters, vhi);
    }
    return velocity_chase_t<float, float, 40>(st, B, S, c, sv, ev, meta, curv, dth, vel, flags, ufwd, state, counters, nullptr);
}

// K5b': many paths, fp32: one wave per path walks the row in windows of 64*L samples, one launch per
// window and direction (stream-ordered, no host synchronisation).  A window's incoming interface state
// is final when its launch starts, so every window is relaxed exactly once; a row of 5 registers per
// sample no longer has to stay resident, so eight paths share a CU instead of two.
size_t velocity_windows_state_bytes(int B, int S)
{
    const int nsc = (S + 64 * 40 - 1) / (64 * 40);
    return 4 * ((size_t)2 * B * (nsc + 1) * 2 + (size_t)2 * B * nsc * 2) + 64;
}

hipError_t launch_velocity_windows(hipStream_t st, int B, int S, const double c[6], double sv, double ev,
                                   const double *meta, const void *curv, const void *dth, void *vel, uint32_t *flags,
                                   void *ufwd, void *state, int *counters)
{
    using R = float;
    constexpr int L = 40, T = 64, SC = T * L;
    const int nsc = (S + SC - 1) / SC;
    const R s = (R)sv, e = (R)ev;
    const size_t lds = sizeof(R) * ((size_t)T * L + T + 8);
    R *bnd = (R *)state;
    R *used = bnd + (size_t)2 * B * (nsc + 1) * 2;
    R *outst = used + (size_t)B * nsc * 2;
    int *dup = counters, *changed = counters + B;
    hipError_t err;
    if ((err = hipMemsetAsync(counters, 0, sizeof(int) * ((size_t)B + 8), st)) != hipSuccess) return err;
    hipLaunchKernelGGL(k_dup_scan<R>, dim3(8, B), dim3(256), 0, st, B, S, make_consts<R>(c), meta, (const R *)curv,
                       (const R *)dth, dup);
    for (int sc = 0; sc < nsc; sc++)
        hipLaunchKernelGGL((k_velocity_long<R, R, L, T, 2, false>), dim3(1, B), dim3(T), lds, st, S, nsc, 0, sc,
                           make_consts<R>(c), s * s, e * e, meta, (const R *)curv, (const R *)dth, (R *)ufwd, (R *)vel, bnd,
                           used, outst, dup, changed, flags, (R *)nullptr);
    for (int sc = nsc - 1; sc >= 0; sc--)
        hipLaunchKernelGGL((k_velocity_long<R, R, L, T, 2, true>), dim3(1, B), dim3(T), lds, st, S, nsc, 0, sc,
                           make_consts<R>(c), s * s, e * e, meta, (const R *)curv, (const R *)dth, (R *)ufwd, (R *)vel, bnd,
                           used, outst, dup, changed, flags, (R *)nullptr);
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

hipError_t launch_basis(hipStream_t st, int order, int n, const double *t, double *out)
{
    hipLaunchKernelGGL(k_basis, dim3((n + 255) / 256), dim3(256), 0, st, order, n, t, out);
    return hipGetLastError();
}

hipError_t launch_lookup(hipStream_t st, int W, const double *seg, double t_max, const double *lut, int what,
                         int n, const double *in, double *out)
{
    hipLaunchKernelGGL(k_lookup, dim3((n + 255) / 256), dim3(256), 0, st, W, seg, t_max, lut, what, n, in, out);
    return hipGetLastError();
}

}  // namespace vap
