// Initial velocities of forward_backward_pass for batches of routes whose nodes / action points carry
// max_velocity and stop (MPG:100-176): the `velocities` list the velocity pass starts from, i.e. d_vcap of
// vap_velocity_pass.
//
// The reference walks the distance grid once and, at the first sample whose parameter has passed a node
// (the wrap test MPG:125) or an action point (MPG:141-145), switches the running max_velocity and, for a
// stop, overwrites that sample's entry with 0.01.  Both tests pick "the first loop sample k with
// t(s_k) >= T" (T = the node's index / the action point's parameter), so the list is a step function of
// the sample index with one event per node / action point:
//   k_event_samples   one thread per (path, event): the sample of the event — a guess from the inverse of
//                     the arc-length table, then settled with the reference's own numbers (the path's
//                     running-sum grid and SM:291-318 distance_to_time) on the neighbouring samples;
//   k_vcap_fill       one thread per sample: the limit in force (events at earlier samples), 0.01 where
//                     an event with stop falls on the sample, end_vel at the end sample.
// Not covered here (single-route path only, vap_route_*): per-node max_acceleration (boundary_map,
// MPG:194-196), reverse / turn nodes, waits.
#include "vap_device.h"
#include "vap_kernels.h"

namespace vap {

constexpr int kNever = 0x7fffffff;

__global__ void k_event_samples(int B, int W, int E, const double *__restrict__ lut, const double *__restrict__ meta,
                                const double *__restrict__ aux, const double *__restrict__ runs,
                                const double *__restrict__ ev_t, int *__restrict__ ev_k)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * E) return;
    const int b = i / E;
    const double T = ev_t[i];
    const double *m = meta + (size_t)b * kMetaStride;
    const double t_max = m[0], total = m[1], dd = m[2];
    const int N = (int)m[3];
    const double end_param = (double)(W - 1);
    int out = kNever;
    if (T > 0.0 && T < end_param && N >= 3 && total > 0.0 && dd > 0.0) {
        const double *D = lut + (size_t)b * kLutN;
        const double *tab = runs + (size_t)b * kGridRunDoubles;
        const int n_runs = (int)aux[(size_t)b * kAuxStride + 3];
        // where the table's parameter reaches T (its parameters are linspace(0, t_max, 1000), SM:443)
        const double lstep = t_max / (double)(kLutN - 1);
        int j = (int)floor(T / lstep);
        j = j < 0 ? 0 : (j > kLutN - 2 ? kLutN - 2 : j);
        const double t0 = linspace_at(t_max, kLutN, j), t1 = linspace_at(t_max, kLutN, j + 1);
        const double s_star = D[j] + (T - t0) / (t1 - t0) * (D[j + 1] - D[j]);
        long k = (long)floor(s_star / dd);
        k = k < 1 ? 1 : (k > N - 2 ? N - 2 : k);
        // settle it with the reference's own parameter of the neighbouring samples
        auto t_of = [&](long kk) {
            int r = grid_run_hint(dd, kk, n_runs);
            return distance_to_time(D, total, t_max, end_param, grid_s(tab, n_runs, kk, r));
        };
        while (k > 1 && t_of(k - 1) >= T) k--;
        while (k <= N - 2 && t_of(k) < T) k++;
        if (k <= N - 2) out = (int)k;
    }
    ev_k[i] = out;
}

template <typename R>
__global__ void k_vcap_fill(int B, int S, int E, const double *__restrict__ meta, const double *__restrict__ first_mv,
                            const double *__restrict__ ev_mv, const int *__restrict__ ev_stop,
                            const int *__restrict__ ev_k, double max_vel, double end_vel, R *__restrict__ vcap)
{
    const int b = blockIdx.y;
    const int N = (int)meta[(size_t)b * kMetaStride + 3];
    const int *K = ev_k + (size_t)b * E;
    const double *MV = ev_mv + (size_t)b * E;
    const int *ST = ev_stop + (size_t)b * E;
    const double m0 = (first_mv && first_mv[b] > 0.0) ? first_mv[b] : max_vel;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < S; k += gridDim.x * blockDim.x) {
        double v = 0.0;
        if (k < N - 1) {
            // events at samples <= k-1 have switched the limit; the sorted event samples make that a count
            int lo = 0, hi = E;   // first event with K > k-1
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (K[mid] <= k - 1) lo = mid + 1; else hi = mid;
            }
            v = lo == 0 ? m0 : (MV[lo - 1] > 0.0 ? MV[lo - 1] : max_vel);
            for (int e = lo; e < E && K[e] == k; e++)
                if (ST[e]) v = 0.01;                    // MPG:127, 153
        } else if (k == N - 1) {
            v = end_vel;                                // MPG:172
        }
        vcap[(size_t)b * S + k] = (R)v;
    }
}

hipError_t launch_initial_velocities(hipStream_t st, bool f64, int B, int W, int S, int E, const double *lut,
                                     const double *meta, const double *aux, const double *runs, const double *first_mv,
                                     const double *ev_t, const double *ev_mv, const int *ev_stop, double max_vel,
                                     double end_vel, int *ev_k, void *vcap)
{
    if (E > 0) hipLaunchKernelGGL(k_event_samples, dim3((B * E + 127) / 128), dim3(128), 0, st, B, W, E, lut, meta, aux, runs, ev_t, ev_k);
    const dim3 grid((unsigned)((S + 255) / 256 < 64 ? (S + 255) / 256 : 64), (unsigned)B);
    if (f64)
        hipLaunchKernelGGL(k_vcap_fill<double>, grid, dim3(256), 0, st, B, S, E, meta, first_mv, ev_mv, ev_stop, ev_k, max_vel, end_vel, (double *)vcap);
    else
        hipLaunchKernelGGL(k_vcap_fill<float>, grid, dim3(256), 0, st, B, S, E, meta, first_mv, ev_mv, ev_stop, ev_k, max_vel, end_vel, (float *)vcap);
    return hipGetLastError();
}

}  // namespace vap
