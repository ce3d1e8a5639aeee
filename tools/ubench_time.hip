// Microbenchmark (developer tool, not part of the product): where a step of the time loop (k_time_integrate,
// MPG:523-584) spends its time.  One lane per path, 4096 paths of 10^4 samples with a smooth synthetic velocity row;
// variants of the same loop:
//   0  the loop as the library has it
//   1  no memory: the four velocity samples are computed from their indices (what the row holds), nothing is loaded
//   2  the three divisions replaced by multiplications with a reciprocal (NOT the reference's rounding: cost only)
//   3  1 + 2
//   4  the grid indices replaced by one truncation each (cost of the index corrections)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I vexautonomousplanner_amd/csrc tools/ubench_time.hip -o tools/bin/ubench_time
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "vap_device.h"

using namespace vap;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__host__ __device__ inline float synth_v(int b, int j, int N)
{
    const float x = (float)j / (float)N;
    const float ramp = fminf(1.0f, fminf(x, 1.0f - x) * 12.0f);
    return 0.05f + ramp * (1.5f + 2.0f * (0.5f + 0.5f * __builtin_sinf(40.0f * x + 0.37f * (float)(b % 97))));
}

template <int MODE>
__global__ __launch_bounds__(64) void k_loop(int B, int S, const float *__restrict__ vel, double max_acc, double max_dec, double dt,
                                             int cap, double *__restrict__ rows, int *__restrict__ counts)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = S;
    const double total = 20.0 + 0.01 * (b % 50), dd = total / ((double)S - 1.5), inv_dd = 1.0 / dd;
    const double inv_dt = 1.0 / dt;
    const float *v = vel + (size_t)b * S;
    double *out = rows + (size_t)b * cap * 8;
    double current_time = 0, current_pos = 0, current_vel = (MODE & 1) ? (double)synth_v(b, 0, N) : (double)v[0];
    int T = 0;
    while (current_pos < total) {
        if (T >= cap) break;
        const double ahead = current_pos + dd;
        int i0, i1;
        if constexpr (MODE & 4) {
            i0 = (int)(current_pos * inv_dd);
            i1 = (int)(ahead * inv_dd);
            i0 = i0 > N - 1 ? N - 1 : i0;
            i1 = i1 > N - 1 ? N - 1 : i1;
        } else {
            i0 = grid_index(current_pos, dd, inv_dd, N);
            i1 = grid_index_from(ahead, dd, inv_dd, N, i0 + 1);
        }
        double a0, a1, c0, c1;
        if constexpr (MODE & 1) {
            a0 = (double)synth_v(b, clamp_index(i0, N), N); a1 = (double)synth_v(b, clamp_index(i0 + 1, N), N);
            c0 = (double)synth_v(b, clamp_index(i1, N), N); c1 = (double)synth_v(b, clamp_index(i1 + 1, N), N);
        } else {
            a0 = (double)v[clamp_index(i0, N)]; a1 = (double)v[clamp_index(i0 + 1, N)];
            c0 = (double)v[clamp_index(i1, N)]; c1 = (double)v[clamp_index(i1 + 1, N)];
        }
        double target_vel, next_target_vel;
        if constexpr (MODE & 2) {
            target_vel = (i0 < 0 || i0 >= N - 1) ? a0 : a0 + (current_pos - (double)i0 * dd) * (a1 - a0) * inv_dd;
            next_target_vel = (i1 < 0 || i1 >= N - 1) ? c0 : c0 + (ahead - (double)i1 * dd) * (c1 - c0) * inv_dd;
        } else {
            target_vel = lerp_at(current_pos, dd, i0, N, a0, a1);
            next_target_vel = lerp_at(ahead, dd, i1, N, c0, c1);
        }
        target_vel = (target_vel + next_target_vel) / 2;
        if (!(target_vel > 0.001)) target_vel = 0.001;
        double accel;
        if constexpr (MODE & 2) accel = clip((target_vel - current_vel) * inv_dt, -max_dec, max_acc);
        else accel = clip((target_vel - current_vel) / dt, -max_dec, max_acc);
        current_vel = clip(current_vel + accel * dt, 0, target_vel);
        double delta_pos = current_vel * dt + 0.5 * accel * dt * dt;
        if (current_vel <= 0.1) delta_pos = 0.1 * dt + 0.5 * accel * dt * dt;
        current_pos += delta_pos;
        double *q = out + (size_t)T * 8;
        q[0] = current_time; q[1] = current_pos; q[2] = current_vel; q[3] = accel; q[5] = target_vel;
        T += 1;
        current_time += dt;
    }
    counts[b] = T;
}


// ---- the candidate: same rows, bit for bit (checked below against variant 0) ----
// F & 1: grid indices from candidates evaluated side by side (no dependent up-then-down correction)
// F & 2: divisions as reciprocal + one residual correction (the hardware expansion without its scaling / fix-up steps; the
//        reciprocal of a grid interval is refined from 1/dd off the critical path, 1/dt is formed once)
// F & 4: the three samples i0, i0+1, i0+2 in one load when i1 == i0 + 1 inside the row (almost always)
// F & 8: a helper wave (same workgroup, another SIMD) reads the paths' positions and advances from LDS and touches the
//        cache lines they will reach during the next steps (its own vmcnt: the loop's loads never wait for them)
__device__ __forceinline__ int grid_index_par(double x, double dd, double inv_dd, int n, double &ed)
{
    double e = floor(x * inv_dd);
    if (!(e >= -1.0)) e = -1.0;
    const double top = (double)(n - 1);
    if (e > top) e = top;
    const double p0 = e * dd, p1 = (e + 1.0) * dd, pp = (e + 2.0) * dd, pm = (e - 1.0) * dd;
    const bool up = (e + 1.0 <= top) & (p1 <= x);
    const bool down = !up & (e >= 0.0) & !(p0 <= x);
    // the two conditions of the definition at the chosen index
    const bool ok_up = (e + 2.0 > top) | !(pp <= x);
    const bool ok_down = (e - 1.0 < 0.0) | (pm <= x);
    const bool ok = up ? ok_up : (down ? ok_down : true);
    double r = up ? e + 1.0 : (down ? e - 1.0 : e);
    if (__builtin_expect(!ok, 0)) r = (double)grid_index(x, dd, inv_dd, n);
    ed = r;
    return (int)r;
}
__device__ __forceinline__ double div_by(double a, double b, double y)   // a / b given y ~ 1/b to working precision
{
    const double q = a * y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);
}
__device__ __forceinline__ double recip_near(double b, double y0)   // 1/b from y0 within ~2^-39 of it
{
    double e = fma(-b, y0, 1.0);
    double y = fma(y0, e, y0);
    e = fma(-b, y, 1.0);
    return fma(y, e, y);
}

template <int F, typename R>
__global__ __launch_bounds__((F & 8) ? 128 : 64) void k_fast(int B, int S, const R *__restrict__ vel, double max_acc, double max_dec, double dt,
                                             int cap, double *__restrict__ rows, int *__restrict__ counts)
{
    __shared__ double2 s_pa[64];
    __shared__ int s_done;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 64 + lane;
    const int N = S;
    const int bb = b < B ? b : B - 1;
    const double total = 20.0 + 0.01 * (bb % 50), dd = total / ((double)S - 1.5), inv_dd = 1.0 / dd;
    const R *v = vel + (size_t)bb * S;
    if constexpr (F & 8) {
        if (threadIdx.x < 64) s_pa[lane] = make_double2(0.0, 0.0);
        if (threadIdx.x == 0) s_done = 0;
        __syncthreads();
        if (threadIdx.x >= 64) {
            // the helper: lines from one advance ahead to five advances ahead of where the path stands
            constexpr int kPer = 128 / sizeof(R);
            R sink = 0;
            for (int it = 0; it < cap * 8; it++) {
                if (*(volatile int *)&s_done) break;
                double2 pa;
                pa.x = *(volatile double *)&s_pa[lane].x;
                pa.y = *(volatile double *)&s_pa[lane].y;
                int lo = (int)((pa.x + pa.y) * inv_dd), hi = (int)((pa.x + 5.0 * pa.y) * inv_dd) + 2;
                lo = lo < 0 ? 0 : (lo > N - 1 ? N - 1 : lo);
                hi = hi < lo ? lo : (hi > N - 1 ? N - 1 : hi);
                hi = hi > lo + 8 * kPer ? lo + 8 * kPer : hi;
#pragma unroll
                for (int k = 0; k <= 8; k++) {
                    const int i = lo + k * kPer;
                    if (i <= hi) sink += *(volatile const R *)(v + i);
                }
            }
            if (sink == (R)-12345.678) counts[0] = 1;   // (keeps the loads)
            return;
        }
    }
    const double inv_dt = recip_near(dt, 1.0 / dt);
    double *out = rows + (size_t)b * cap * 8;
    double current_time = 0, current_pos = 0, current_vel = (double)v[0];
    double last_delta = 0.0;
    int T = 0;
    while (b < B && current_pos < total) {
        if (T >= cap) break;
        const double ahead = current_pos + dd;
        int i0, i1;
        double e0 = 0, e1 = 0;
        if constexpr (F & 1) {
            i0 = grid_index_par(current_pos, dd, inv_dd, N, e0);
            i1 = grid_index_par(ahead, dd, inv_dd, N, e1);
        } else {
            i0 = grid_index(current_pos, dd, inv_dd, N);
            i1 = grid_index_from(ahead, dd, inv_dd, N, i0 + 1);
        }
        if constexpr (F & 8) s_pa[lane] = make_double2(current_pos, last_delta);   // for the helper wave
        double a0, a1, c0, c1;
        bool fast = false;
        if constexpr (F & 4) fast = (i0 >= 0) & (i0 + 2 <= N - 1) & (i1 == i0 + 1);
        if (fast) {
            if constexpr (sizeof(R) == 4) {
                struct __attribute__((packed, aligned(4))) f3 { float a, b, c; };
                const f3 t = *reinterpret_cast<const f3 *>(v + i0);
                a0 = (double)t.a; a1 = (double)t.b; c0 = a1; c1 = (double)t.c;
            } else {
                struct __attribute__((packed, aligned(8))) d3 { double a, b, c; };
                const d3 t = *reinterpret_cast<const d3 *>(v + i0);
                a0 = t.a; a1 = t.b; c0 = a1; c1 = t.c;
            }
        } else {
            a0 = (double)v[clamp_index(i0, N)]; a1 = (double)v[clamp_index(i0 + 1, N)];
            c0 = (double)v[clamp_index(i1, N)]; c1 = (double)v[clamp_index(i1 + 1, N)];
        }
        double target_vel, next_target_vel;
        if constexpr (F & 2) {
            auto lerp_fast = [&](double x, int idx, double ed, double y0, double y1) {
                if (idx < 0 || idx >= N - 1) return y0;
                const double x0 = ((F & 1) ? ed : (double)idx) * dd, x1 = (((F & 1) ? ed : (double)idx) + 1.0) * dd;
                const double den = x1 - x0;
                return y0 + div_by((x - x0) * (y1 - y0), den, recip_near(den, inv_dd));
            };
            target_vel = lerp_fast(current_pos, i0, e0, a0, a1);
            next_target_vel = lerp_fast(ahead, i1, e1, c0, c1);
        } else {
            target_vel = lerp_at(current_pos, dd, i0, N, a0, a1);
            next_target_vel = lerp_at(ahead, dd, i1, N, c0, c1);
        }
        target_vel = (target_vel + next_target_vel) / 2;
        if (!(target_vel > 0.001)) target_vel = 0.001;
        double accel;
        if constexpr (F & 2) accel = clip(div_by(target_vel - current_vel, dt, inv_dt), -max_dec, max_acc);
        else accel = clip((target_vel - current_vel) / dt, -max_dec, max_acc);
        current_vel = clip(current_vel + accel * dt, 0, target_vel);
        double delta_pos = current_vel * dt + 0.5 * accel * dt * dt;
        if (current_vel <= 0.1) delta_pos = 0.1 * dt + 0.5 * accel * dt * dt;
        current_pos += delta_pos;
        last_delta = delta_pos;
        double *q = out + (size_t)T * 8;
        q[0] = current_time; q[1] = current_pos; q[2] = current_vel; q[3] = accel; q[5] = target_vel;
        T += 1;
        current_time += dt;
    }
    if (b < B) counts[b] = T;
    if constexpr (F & 8) {
        if (lane == 0) *(volatile int *)&s_done = 1;
    }
}

static std::vector<double> g_ref_rows;
static std::vector<int> g_ref_counts;

template <int F, typename R>
void run_fast(const char *what, int B, int S, const R *dv, double *rows, int *counts, int cap, bool is_ref = false)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    CK(hipMemset(rows, 0, (size_t)B * cap * 8 * 8));
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_fast<F, R>), dim3((B + 63) / 64), dim3((F & 8) ? 128 : 64), 0, 0, B, S, dv, 12.0, 12.0, 0.01, cap, rows, counts);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<int> c(B);
    std::vector<double> r((size_t)B * cap * 8);
    CK(hipMemcpy(c.data(), counts, B * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(r.data(), rows, r.size() * 8, hipMemcpyDeviceToHost));
    int mx = 0;
    for (int x : c) if (x > mx) mx = x;
    long long bad = 0;
    if (is_ref) { g_ref_rows = r; g_ref_counts = c; }
    else {
        for (int b = 0; b < B; b++) {
            if (c[b] != g_ref_counts[b]) { bad++; continue; }
            if (memcmp(&r[(size_t)b * cap * 8], &g_ref_rows[(size_t)b * cap * 8], (size_t)c[b] * 64)) bad++;
        }
    }
    printf("%-58s %8.3f ms  longest %d -> %7.1f ns per step   paths whose rows differ from the loop as it is: %lld\n", what, best, mx,
           best * 1e6 / mx, bad);
}

template <int MODE>
void run(const char *what, int B, int S, const float *dv, double *rows, int *counts, int cap)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_loop<MODE>, dim3((B + 63) / 64), dim3(64), 0, 0, B, S, dv, 12.0, 12.0, 0.01, cap, rows, counts);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<int> c(B);
    CK(hipMemcpy(c.data(), counts, B * sizeof(int), hipMemcpyDeviceToHost));
    long long sum = 0; int mx = 0;
    for (int x : c) { sum += x; if (x > mx) mx = x; }
    printf("%-58s %8.3f ms   rows %lld  longest %d  -> %7.1f ns per step of the longest path\n", what, best, sum, mx, best * 1e6 / mx);
}

int main()
{
    const int B = 4096, S = 10000, cap = 4096;
    std::vector<float> hv((size_t)B * S);
    for (int b = 0; b < B; b++) for (int j = 0; j < S; j++) hv[(size_t)b * S + j] = synth_v(b, j, S);
    float *dv; double *rows; int *counts;
    CK(hipMalloc(&dv, hv.size() * 4)); CK(hipMalloc(&rows, (size_t)B * cap * 8 * 8)); CK(hipMalloc(&counts, B * 4));
    CK(hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    run<0>("0 the library's loop", B, S, dv, rows, counts, cap);
    run<1>("1 no memory (samples computed from their indices)", B, S, dv, rows, counts, cap);
    run<2>("2 divisions -> multiplications (cost only)", B, S, dv, rows, counts, cap);
    run<3>("3 no memory, no divisions", B, S, dv, rows, counts, cap);
    run<4>("4 truncating grid indices", B, S, dv, rows, counts, cap);
    run<7>("7 all three", B, S, dv, rows, counts, cap);
    printf("fp32 rows\n");
    run_fast<0, float>("F0 the loop as it is", B, S, dv, rows, counts, cap, true);
    run_fast<1, float>("F1 side-by-side grid indices", B, S, dv, rows, counts, cap);
    run_fast<2, float>("F2 divisions by reciprocal + correction", B, S, dv, rows, counts, cap);
    run_fast<4, float>("F4 three samples in one load", B, S, dv, rows, counts, cap);
    run_fast<8, float>("F8 touch the line two steps on", B, S, dv, rows, counts, cap);
    run_fast<12, float>("F12 = 4 + 8", B, S, dv, rows, counts, cap);
    run_fast<7, float>("F7 = 1 + 2 + 4", B, S, dv, rows, counts, cap);
    run_fast<15, float>("F15 all", B, S, dv, rows, counts, cap);
    // fp64 rows (what the default mode hands to the time loop)
    std::vector<double> hd(hv.begin(), hv.end());
    double *dvd;
    CK(hipMalloc(&dvd, hd.size() * 8));
    CK(hipMemcpy(dvd, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
    printf("fp64 rows\n");
    run_fast<0, double>("F0 the loop as it is", B, S, dvd, rows, counts, cap, true);
    run_fast<7, double>("F7 = 1 + 2 + 4", B, S, dvd, rows, counts, cap);
    run_fast<15, double>("F15 all", B, S, dvd, rows, counts, cap);
    return 0;
}
