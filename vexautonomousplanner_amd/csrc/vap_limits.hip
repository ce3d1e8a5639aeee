// Initial velocities of forward_backward_pass for batches of routes whose nodes / action points carry
// max_velocity and stop (MPG:100-176): the `velocities` list the velocity pass starts from, i.e. d_vcap of
// vap_velocity_pass.
//
// The reference walks the distance grid once and, at the first sample whose parameter has passed a node
// (the wrap test MPG:125) or an action point (MPG:141-145), switches the running max_velocity and, for a
// stop, overwrites that sample's entry with 0.01.  Both tests pick "the first loop sample k with
// t(s_k) >= T" (T = the node's index / the action point's parameter), so the list is a step function of
// the sample index with one event per node / action point:
//   k_event_samples   one thread per (path, node / action point): the sample it would take effect on — a guess
//                     from the inverse of the arc-length table, then settled with the reference's own numbers
//                     (the path's running-sum grid and SM:291-318 distance_to_time) on the neighbouring samples;
//   k_event_merge     one thread per path: the action points the reference never reaches (one pending action
//                     point per sample), and one event list ordered by sample, nodes first;
//   k_limit_fill      one thread per sample: the limit in force (events at earlier samples), 0.01 where
//                     an event with stop falls on the sample, end_vel at the end sample;
//   k_limit_fill      also the per-sample max_acceleration rows the two sweeps see (boundary_map / max_accels,
//                     MPG:194-196, 256-257) when the route changes max_acceleration.
// Routes cut into several splines by reverse / turn nodes go through the same kernels: forward_backward_pass treats
// their nodes like any other (MPG:112-176), only distance_to_time runs over the concatenated table (LutView).
// Waits and turns act in the time domain (vap_time.hip).
#include "vap_device.h"
#include "vap_kernels.h"

namespace vap {

constexpr int kNever = 0x7fffffff;

// first loop sample k (1 <= k <= N-2) whose parameter has reached T, or kNever
__device__ int first_sample_reaching(double T, int W, const LutView &v, const double *__restrict__ m,
                                     const double *__restrict__ tab, int n_runs)
{
    const double total = m[1], dd = m[2];
    const int N = (int)m[3];
    const double end_param = (double)(W - 1);
    if (!(T > 0.0 && T < end_param && N >= 3 && total > 0.0 && dd > 0.0)) return kNever;
    // where the table's parameter reaches T: the spline whose parameter range holds it (its parameters are
    // linspace(0, parameters[-1], 1000) + offset, SM:443, 461) — a guess only, settled below
    int si = v.n_spl - 1;
    for (int i = 0; i < v.n_spl - 1; i++)
        if (T <= v.sp[(i + 1) * kSplineStride + 2]) { si = i; break; }
    const double t_max = v.sp[si * kSplineStride + 0], lt = T - v.sp[si * kSplineStride + 2];
    const double *D = v.D + (size_t)si * kLutN;
    const double lstep = t_max / (double)(kLutN - 1);
    int j = (int)floor(lt / lstep);
    j = j < 0 ? 0 : (j > kLutN - 2 ? kLutN - 2 : j);
    const double t0 = linspace_at(t_max, kLutN, j), t1 = linspace_at(t_max, kLutN, j + 1);
    const double s_star = v.sp[si * kSplineStride + 1] + D[j] + (lt - t0) / (t1 - t0) * (D[j + 1] - D[j]);
    long k = (long)floor(s_star / dd);
    k = k < 1 ? 1 : (k > N - 2 ? N - 2 : k);
    // settle it with the reference's own parameter of the neighbouring samples
    auto t_of = [&](long kk) {
        int r = grid_run_hint(dd, kk, n_runs);
        return lutv_distance_to_time(v, grid_s(tab, n_runs, kk, r));
    };
    while (k > 1 && t_of(k - 1) >= T) k--;
    while (k <= N - 2 && t_of(k) < T) k++;
    return k <= N - 2 ? (int)k : kNever;
}

// One thread per (path, node 1..W-2 or action point): the sample at which it would take effect on its own.
__global__ void k_event_samples(int B, int W, int M, const double *__restrict__ lut, const double *__restrict__ meta,
                                const double *__restrict__ aux, const double *__restrict__ runs, RouteTables rt,
                                const double *__restrict__ ap_t, int *__restrict__ node_k, int *__restrict__ ap_k)
{
    const int per = W + M;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per) return;
    const int b = i / per, e = i - b * per;
    const double *m = meta + (size_t)b * kMetaStride;
    const double *tab = runs + (size_t)b * kGridRunDoubles;
    const int n_runs = (int)aux[(size_t)b * kAuxStride + 3];
    // the table: one spline with zero offsets for a plain path (bit for bit distance_to_time of SM:291-318 on it)
    const double plain_sp[kSplineStride] = {m[0], 0.0, 0.0, 0.0};
    LutView v;
    v.D = lut + (size_t)b * rt.NS * kLutN;
    v.sp = rt.sptab ? rt.sptab + (size_t)b * rt.NS * kSplineStride : plain_sp;
    v.n_spl = rt.sptab ? rt.nspl[b] : 1;
    v.total = m[1];
    v.end_param = (double)(W - 1);
    if (e < W) {
        // node e: index 0 is the start (sample 0), the last node is never passed inside the loop (MPG:125)
        node_k[(size_t)b * W + e] = e == 0 ? 0 : (e == W - 1 ? kNever : first_sample_reaching((double)e, W, v, m, tab, n_runs));
    } else {
        const double T = ap_t[(size_t)b * M + (e - W)];
        ap_k[(size_t)b * M + (e - W)] = (T == T && T != INFINITY) ? first_sample_reaching(T, W, v, m, tab, n_runs) : kNever;
    }
}

// One thread per path: the reference looks at one pending action point per sample (MPG:141-145, 163), so an action
// point that would take effect on the sample of its predecessor (or earlier) never does, and neither do those after
// it; then nodes and action points merge into one list by sample, a node before an action point on the same sample
// (the order the reference handles them in, MPG:125 then 141 — whatever their parameters).
__global__ void k_event_merge(int B, int W, int M, const LimitInputs in, const int *__restrict__ node_k, int *__restrict__ ap_k,
                              int *__restrict__ ev_k, double *__restrict__ ev_mv, double *__restrict__ ev_ma,
                              int *__restrict__ ev_stop)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int E = (W > 2 ? W - 2 : 0) + M;
    const int *NK = node_k + (size_t)b * W;
    int *AK = ap_k + (size_t)b * M;
    int last = -1;
    bool blocked = false;
    for (int a = 0; a < M; a++) {
        if (blocked || AK[a] == kNever || AK[a] <= last) { blocked = true; AK[a] = kNever; }
        else last = AK[a];
    }
    int n = 1, a = 0, o = 0;
    const size_t base = (size_t)b * E;
    while (o < E) {
        const int kn = n <= W - 2 ? NK[n] : kNever, ka = a < M ? AK[a] : kNever;
        const bool take_node = n <= W - 2 && (a >= M || kn <= ka);
        if (take_node) {
            ev_k[base + o] = kn;
            ev_mv[base + o] = in.node_mv ? in.node_mv[(size_t)b * W + n] : 0.0;
            ev_ma[base + o] = in.node_ma ? in.node_ma[(size_t)b * W + n] : 0.0;
            ev_stop[base + o] = in.node_stop ? in.node_stop[(size_t)b * W + n] : 0;
            n++;
        } else {
            ev_k[base + o] = ka;
            ev_mv[base + o] = in.ap_mv ? in.ap_mv[(size_t)b * M + a] : 0.0;
            ev_ma[base + o] = in.ap_ma ? in.ap_ma[(size_t)b * M + a] : 0.0;
            ev_stop[base + o] = in.ap_stop ? in.ap_stop[(size_t)b * M + a] : 0;
            a++;
        }
        o++;
    }
}

// The reference's lists in terms of the events (sorted by sample; R of them are reached, i.e. fall on a loop
// sample):  max_accels = [a0, acc(e_0), ..., acc(e_{R-1}), max_acc]  (MPG:100-104, 134-137, 155-160, 176) and
// boundary_map = {0: 0} + {sample of e: 1 + index of the LAST event on that sample} (MPG:139-140, 162 — an
// action point on a node's sample replaces the node's entry).
//   forward  (MPG:194-196)  at a boundary: max_acc = max_dec = max_accels[boundary_map[i]]
//                           -> step from sample i: the last event with sample <= i (a0 before the first)
//   backward (MPG:256-257)  at a boundary: max_acc = max_accels[boundary_map[i] + 1] — the NEXT list entry, i.e. the
//                           event after the last one on the nearest boundary at or above i (max_acc past the end);
//                           above every boundary: what the forward sweep left; max_dec: what the forward sweep left
template <typename R, bool EV_LDS>
__global__ void k_limit_fill(int B, int W, int S, int E, const double *__restrict__ meta, LimitInputs in,
                             const int *__restrict__ ev_k, const double *__restrict__ ev_mv,
                             const double *__restrict__ ev_ma, const int *__restrict__ ev_stop, R *__restrict__ vcap,
                             R *__restrict__ acc_fwd, R *__restrict__ acc_bwd, R *__restrict__ dec_bwd)
{
    // the path's event list in LDS (a few dozen entries): every sample searches it, and from global memory the five
    // dependent probes of a search were most of the kernel (252 -> ~90 us for config 3's 4.1e7 samples)
    extern __shared__ __attribute__((aligned(16))) double s_ev[];     // [E] max_velocity, [E] max_acceleration, then ints
    const int b = blockIdx.y;
    const int N = (int)meta[(size_t)b * kMetaStride + 3];
    const double *MV = ev_mv + (size_t)b * E, *MA = ev_ma + (size_t)b * E;
    const int *K = ev_k + (size_t)b * E, *ST = ev_stop + (size_t)b * E;
    if constexpr (EV_LDS) {       // (lists too long for LDS — thousands of action points — are searched where they lie)
        double *mv = s_ev, *ma = s_ev + E;
        int *kk = reinterpret_cast<int *>(s_ev + 2 * E), *st = kk + E;
        for (int e = threadIdx.x; e < E; e += blockDim.x) {
            mv[e] = MV[e];
            ma[e] = MA[e];
            kk[e] = K[e];
            st[e] = ST[e];
        }
        __syncthreads();
        MV = mv; MA = ma; K = kk; ST = st;
    }
    const double mv0 = in.node_mv ? in.node_mv[(size_t)b * W] : 0.0, ma0 = in.node_ma ? in.node_ma[(size_t)b * W] : 0.0;
    const double m0 = mv0 > 0.0 ? mv0 : in.max_vel;     // MPG:100-107: node 0
    const double a0 = ma0 > 0.0 ? ma0 : in.max_acc;
    auto acc_of = [&](int e) { return MA[e] > 0.0 ? MA[e] : in.max_acc; };
    auto count_le = [&](int k) {    // events with sample <= k (the samples ascend; unreached ones hold INT_MAX)
        int lo = 0, hi = E;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (K[mid] <= k) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int reached = count_le(N - 2);
    const double left = reached == 0 ? a0 : acc_of(reached - 1);   // what the forward sweep leaves behind
    if (dec_bwd && blockIdx.x == 0 && threadIdx.x == 0) dec_bwd[b] = (R)left;
    // four consecutive samples per thread: one search for the first, a walk along the (sparse) event list for the others,
    // and the rows leave as 16- / 32-byte pieces
    for (int k0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4; k0 < S; k0 += gridDim.x * blockDim.x * 4) {
        R ov[4], oaf[4], oab[4];
        int before = count_le(k0 - 1);                  // events that have switched the limit before sample k0
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = k0 + j;
            if (j) while (before < E && K[before] <= k - 1) before++;
            double v = 0.0, af = 0.0, ab = 0.0;
            if (k < N - 1) {
                v = before == 0 ? m0 : (MV[before - 1] > 0.0 ? MV[before - 1] : in.max_vel);
                int upto = before;
                for (; upto < E && K[upto] == k; upto++)
                    if (ST[upto]) v = 0.01;              // MPG:127, 153
                if (acc_fwd) af = upto == 0 ? a0 : acc_of(upto - 1);
            } else if (k == N - 1) {
                v = in.end_vel;                          // MPG:172
            }
            if (acc_fwd && k <= N - 1 && k >= 1) {
                // nearest boundary at or above k: the first event with sample >= k
                const int first = before;
                if (first >= reached) {
                    ab = left;
                } else {
                    const int last_there = count_le(K[first]) - 1;     // the last event on that sample
                    ab = last_there + 1 < reached ? acc_of(last_there + 1) : in.max_acc;
                }
            }
            ov[j] = (R)v; oaf[j] = (R)af; oab[j] = (R)ab;
        }
        const size_t o = (size_t)b * S + k0;
        auto put = [&](R *__restrict__ row, const R (&x)[4]) {
            if (k0 + 3 < S && (o & 3) == 0) {
                if constexpr (sizeof(R) == 8) {
                    *reinterpret_cast<double2 *>(row + o) = make_double2(x[0], x[1]);
                    *reinterpret_cast<double2 *>(row + o + 2) = make_double2(x[2], x[3]);
                } else {
                    *reinterpret_cast<float4 *>(row + o) = make_float4(x[0], x[1], x[2], x[3]);
                }
            } else {
                for (int j = 0; j < 4 && k0 + j < S; j++) row[o + j] = x[j];
            }
        };
        put(vcap, ov);
        if (acc_fwd) {
            put(acc_fwd, oaf);
            put(acc_bwd, oab);
        }
    }
}

hipError_t launch_route_limits(hipStream_t st, bool f64, int B, int W, int M, int S, const double *lut, const double *meta,
                               const double *aux, const double *runs, const LimitInputs &in, int *node_k, int *ap_k,
                               int *ev_k, double *ev_mv, double *ev_ma, int *ev_stop, void *vcap, void *acc_fwd,
                               void *acc_bwd, void *dec_bwd, RouteTables rt)
{
    const int E = (W > 2 ? W - 2 : 0) + M;
    const int per = W + M;
    hipLaunchKernelGGL(k_event_samples, dim3((B * per + 127) / 128), dim3(128), 0, st, B, W, M, lut, meta, aux, runs, rt, in.ap_t, node_k, ap_k);
    if (E > 0) hipLaunchKernelGGL(k_event_merge, dim3((B + 63) / 64), dim3(64), 0, st, B, W, M, in, node_k, ap_k, ev_k, ev_mv, ev_ma, ev_stop);
    const dim3 grid((unsigned)((S + 1023) / 1024 < 64 ? (S + 1023) / 1024 : 64), (unsigned)B);
    const size_t lds = (sizeof(double) * 2 + sizeof(int) * 2) * (size_t)E + 16;
#define VAP_FILL(R_, LDS_)                                                                                                      \
    hipLaunchKernelGGL((k_limit_fill<R_, LDS_>), grid, dim3(256), LDS_ ? lds : 0, st, B, W, S, E, meta, in, ev_k, ev_mv, ev_ma, ev_stop, \
                       (R_ *)vcap, (R_ *)acc_fwd, (R_ *)acc_bwd, (R_ *)dec_bwd)
    if (lds <= 48 * 1024) { if (f64) VAP_FILL(double, true); else VAP_FILL(float, true); }
    else { if (f64) VAP_FILL(double, false); else VAP_FILL(float, false); }
#undef VAP_FILL
    return hipGetLastError();
}

}  // namespace vap
