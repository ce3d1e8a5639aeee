// vap_sample_lane.h — K3+K4 (SM:291-318, 340-346, 550-580, 204-215, MPG:112-176) with LANE = SAMPLE, for the fused
// kernel: a wavefront evaluates 64 consecutive samples of one path, tile after tile, inside k_velocity_lanes' forward
// producers (vap_velocity_lanes.hip), so that the fp64 curvature / heading-difference rows reach the forward sweep
// without crossing HBM.
//
// k_sample (vap_kernels.hip) stages a path's whole arc-length table, its slopes and its coefficient blocks in LDS
// (25 KB per path): a workgroup of 16 paths cannot.  But the samples of a path are visited in order, so a tile of 64
// needs only a WINDOW of the table — the entries between the previous tile's last hit and ~30 further on — which the
// wave keeps in 256 bytes of LDS and refills one tile ahead; the coefficient blocks (one or two segments per tile) and
// the distance-grid runs are read through L1 (the 64 lanes mostly read the same addresses).  The slope of a table
// interval is formed per sample with the expression k_sample forms it with per entry.
//
// Same expressions, same order, same helper functions as k_sample: the rows are bit-identical to it
// (tests/test_gpu_fused.py).
#pragma once
#include "vap_device.h"

namespace vap {

constexpr int kLaneWindow = 32;   // table entries a wave keeps in LDS per path

struct LanePath {        // per path, wave-uniform
    const double *D;     // arc-length table [kLutN] (global)
    const double *coef;  // coefficient blocks [G][kCoefDoubles] (global)
    const double *runs;  // distance-grid runs (global)
    double t_max, total, lstep, tstep, inv_tstep, end_param;
    int N, n_runs, G, tab_n;
};

// what sample j-1 hands to sample j (lane to lane by DPP, lane 63 -> the next tile's lane 0 through these)
struct LaneCarry {
    double ex = 1.0, ey = 0.0, kap = 0.0, kap_prev = 0.0;
    float th = 0.0f;
    int jj = -1;
};

struct LaneSample {
    float x, y, th, kapf;      // the caller's rows at sample j
    double kap;                // fp64 curvature of sample j
    double dth_prev;           // |theta_j - theta_{j-1}| in fp64 (0 for j = 0 and past the end): row index j-1
    double kap_m1, kap_m2;     // curvatures of samples j-1, j-2 (what the forward step into j uses)
};

__device__ __forceinline__ double lane_read(double v, int lane)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float lane_read(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ int lane_read(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int lane_shift_up(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ float lane_shift_up(float x) { return __builtin_bit_cast(float, lane_shift_up(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ double lane_shift_up(double x)
{
    const uint64_t v = __builtin_bit_cast(uint64_t, x);
    const uint32_t l2 = (uint32_t)lane_shift_up((int)(uint32_t)v), h2 = (uint32_t)lane_shift_up((int)(uint32_t)(v >> 32));
    return __builtin_bit_cast(double, ((uint64_t)h2 << 32) | l2);
}

// The table entry this lane prefetches for the window that starts at entry w0 (lanes past the window load entry 999).
__device__ __forceinline__ double lane_window_fetch(const LanePath &p, int w0, int lane)
{
    int j = w0 + (lane < kLaneWindow ? lane : kLaneWindow - 1);
    j = j > kLutN - 1 ? kLutN - 1 : j;
    return p.D[j];
}

// Sample j = tile*64 + lane of path p.  `win` (LDS, kLaneWindow doubles, this wave's own) holds table entries
// w0 .. w0+31 (put there by the caller from lane_window_fetch of the previous step); `run_cur` is the lane's cursor
// into the distance-grid runs, `carry` the hand-over from the previous tile.  Returns in `next_w0` the window start for
// the next tile (wave-uniform).
__device__ __forceinline__ LaneSample lane_sample(const LanePath &p, int j, int lane, const double *win, int w0, int &run_cur,
                                                  LaneCarry &carry, int &next_w0, long long *ph = nullptr)
{
    const long long ph0 = ph ? __builtin_amdgcn_s_memtime() : 0;
    const int N = p.N;
    // samples past the end of the grid are evaluated at the end sample and blanked by the caller: the body stays
    // straight-line (MPG:112-122, 172-176)
    const int k = j < N - 1 ? j : N - 1;
    const double sk = grid_s(p.runs, p.n_runs, (long)(k < 0 ? 0 : k), run_cur);
    const double s = (k == N - 1) ? p.total : sk;
    // SM:291-318 distance_to_time: np.searchsorted(D, s, "left") inside the window when it covers the wave's samples
    const double s_hi = lane_read(s, 63);
    const long long ph1 = ph ? __builtin_amdgcn_s_memtime() : 0;
    const int w_last = w0 + kLaneWindow - 1;
    const bool covered = w_last >= kLutN - 1 || win[kLaneWindow - 1] >= s_hi;      // wave-uniform
    int idx;
    double d0, d1;
    if (covered) {
        int lo = 0, hi = kLaneWindow;                 // offsets into the window; entries before w0 are < every s of this tile
#pragma unroll
        for (int it = 0; it < 6; it++) {              // 32 candidates + "none": six halvings
            const int mid = (lo + hi) >> 1;
            const bool below = (w0 + mid <= kLutN - 1) && win[mid] < s;
            lo = below ? mid + 1 : lo;
            hi = below ? hi : mid;
        }
        idx = w0 + lo;
        idx = idx > kLutN ? kLutN : idx;
        idx = idx < 1 ? 1 : idx;
        idx = idx > kLutN - 1 ? kLutN - 1 : idx;      // (s <= total = D[999]: the search cannot pass the last entry)
        const int o = idx - w0;
        // idx-1 can lie one entry before the window only when idx == w0, i.e. w0 == idx: the caller starts the window
        // one entry below the previous tile's last hit, so o >= 1 except for the very first window (w0 = 0, idx = 1)
        d0 = o >= 1 ? win[o - 1] : p.D[idx - 1];
        d1 = win[o < kLaneWindow ? o : kLaneWindow - 1];
    } else {
        idx = lut_search_left(p.D, s);
        idx = idx < 1 ? 1 : idx;
        idx = idx > kLutN - 1 ? kLutN - 1 : idx;
        d0 = p.D[idx - 1];
        d1 = p.D[idx];
    }
    next_w0 = lane_read(idx, 63) - 1;
    next_w0 = next_w0 < 0 ? 0 : next_w0;
    const long long ph2 = ph ? __builtin_amdgcn_s_memtime() : 0;
    const double t0 = (double)(idx - 1) * p.lstep;
    const double t1 = (idx == kLutN - 1) ? p.t_max : (double)idx * p.lstep;
    const double wt = (t1 - t0) / (d1 - d0);          // the interval slope k_sample / k_lut form per entry (SM:311-317)
    const bool exact = s <= 0.0 || s >= p.total;      // the reference's early returns: t is exact
    double t = fma(wt, s - d0, t0);
    t = s >= p.total ? p.end_param : t;
    // SM:340-346 / 550-580: the table entry the reference's step lookup selects
    bool near;
    int jj = table_index_fast(t, p.tab_n, p.inv_tstep, near);
    if (near && !exact) {   // a few ulps from a decision point: redo with the reference's own rounding
        t = t0 + (t1 - t0) * (s - d0) / (d1 - d0);
        jj = table_index(t, p.tab_n, p.end_param);
    }
    const long long ph3 = ph ? __builtin_amdgcn_s_memtime() : 0;
    const double tp = (jj == p.tab_n - 1) ? p.end_param : (double)jj * p.tstep;
    double lt;
    int sg;
    normalize_inside(tp, p.G, lt, sg);
    const double *c = p.coef + (size_t)sg * kCoefDoubles;
    const double ex = horner4(c + kCoefD1, lt), ey = horner4(c + kCoefD1 + 5, lt);     // P'
    const double fx = horner3(c + kCoefD2, lt), fy = horner3(c + kCoefD2 + 4, lt);     // P''
    const double ss = fma(ex, ex, ey * ey);                               // SM:517
    const double num = fma(ex, fy, -(ey * fx));                           // SM:523
    const double kap = (ss >= 1e-10) ? curvature_of(num, ss) : 0.0;       // SM:526-527
    LaneSample o;
    o.kap = kap;
    o.kapf = (float)kap;
    o.th = heading_of<float>(ey, ex);                                     // SM:536
    // SM:204-215 get_point_at_parameter(t) at the sample's own parameter
    normalize_inside(t, p.G, lt, sg);
    const float *cf = reinterpret_cast<const float *>(p.coef + (size_t)sg * kCoefDoubles + kCoefPf);
    const float ltf = (float)lt;
    o.x = horner5f(cf, ltf);
    o.y = horner5f(cf + 6, ltf);
    const long long ph4 = ph ? __builtin_amdgcn_s_memtime() : 0;
    // the neighbour below: lane - 1, or the previous tile's last lane
    double pex = lane_shift_up(ex), pey = lane_shift_up(ey), pk = lane_shift_up(kap);
    float pth = lane_shift_up(o.th);
    int pjj = lane_shift_up(jj);
    if (lane == 0) { pex = carry.ex; pey = carry.ey; pk = carry.kap; pth = carry.th; pjj = carry.jj; }
    double pk2 = lane_shift_up(pk);
    if (lane == 0) pk2 = carry.kap_prev;
    // |heading[j] - heading[j-1]| of the reference's raw atan2 values (row index j-1): zero when the two samples share a
    // table entry, and for the samples past the end
    double dth = 0.0;
    if (j >= 1 && j <= N - 1 && pjj != jj) dth = dtheta_f64(pex, pey, ex, ey, pth, o.th);
    o.dth_prev = dth;
    o.kap_m1 = pk;
    o.kap_m2 = pk2;
    carry.ex = lane_read(ex, 63);
    carry.ey = lane_read(ey, 63);
    carry.kap = lane_read(kap, 63);
    carry.kap_prev = lane_read(pk, 63);
    carry.th = lane_read(o.th, 63);
    carry.jj = lane_read(jj, 63);
    if (ph) {
        const long long ph5 = __builtin_amdgcn_s_memtime();
        ph[0] += ph1 - ph0;   // grid distance
        ph[1] += ph2 - ph1;   // table search
        ph[2] += ph3 - ph2;   // slope, parameter, table index
        ph[3] += ph4 - ph3;   // coefficient loads, derivatives, curvature, heading, position
        ph[4] += ph5 - ph4;   // neighbour exchange, heading difference
    }
    return o;
}

}  // namespace vap
